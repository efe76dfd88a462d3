// pnvo_internal.h — declarations shared by the HIP translation units of libpnvo.so (gfx950 only).
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnvo {

constexpr int PNVO_ABSMAX_UINTS = 64 * 16;      // an absolute-maximum record: 64 atomicMax slots, one cache line apart

// A 2-channel slice of one observation tensor feeding the fused stem (MODE 2 of conv_mfma_kernel).
struct SrcPiece {
  const float *base;   // tensor base (nullptr: zero padding piece)
  int nch;             // channels per pixel of that tensor
  int choff;           // first channel of the slice
};

// One convolution / linear layer as an implicit GEMM:  M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.
struct ConvArgs {
  const float *x;        // [B,H,W,CIN] NHWC, CIN % 8 == 0 (channel-padded)
  const float *wpk;      // packed weights, see pack_conv_weight()
  float *y;              // [M, y_cstride] raw output
  const float *in_scale; // [B,CIN] per-(sample,channel) scale of the fused input transform, or nullptr
  const float *in_shift; // [B,CIN] shift; transform = relu(x*scale+shift) (GroupNorm+ReLU of the producer)
  float *stats;          // [B,slots,COUTP,2] per-(sample,slot,channel) partial (sum, sumsq) or nullptr
  const float *bias;     // [bias_rows,COUT] epilogue bias or nullptr (linear layers)
  const int64_t *bias_row; // [B] row of `bias` per sample (act-embed variants) or nullptr (row 0)
  int B, H, W, CIN;
  int Ho, Wo, COUT, COUTP; // COUTP = COUT rounded up to 32
  int KH, KW, stride, pad;
  int y_cstride;         // channel stride of y (COUT or COUTP)
  int relu_out;          // epilogue ReLU (after bias)
  int slots;             // stats slots per sample
  int lds_floats;        // dynamic LDS available for staging the input transform tables
  int MT, NT;            // wave tile: MT*32 pixels x NT*32 output channels
  int src_mode;          // 1: A is gathered from the observation tensors through `pieces` (x unused)
  int up;                // 2: transposed (backward-data of a stride-2 conv): source = (pos - pad + k) / 2 when even; else 1
  int accum;             // epilogue adds into y instead of overwriting it
  const float *zero_page;            // >= 16 B of zeros (target of masked gathers)
  int y_sh, y_sw, y_oh, y_ow, y_H, y_W;   // y_sh != 0: output pixel (n,i,j) is stored at (n, i*y_sh + y_oh, j*y_sw + y_ow) of a
                                          // [B, y_H, y_W, y_cstride] tensor (one parity phase of a stride-2 backward-data conv)
  int wsplit;            // 1: the 4 waves of a workgroup share ONE 32-pixel tile and split its reduction (summed through LDS in
                         // a fixed order): 4x finer work items for layers whose grid divides badly over the 256 CUs
  int ksplit;            // > 1: blockIdx.z owns a slice of the (tap, channel-group) stages and writes a raw partial
  float *kpart;          // [ksplit][M][y_cstride] partials (bias / ReLU / y are left to ksplit_reduce)
  SrcPiece pieces[8][2][2];          // [j][lane half h][q]: channels 8j+4h+2q, +1 of the stem's K order
};

// The LDS-staged fused stem (stem_lds.hip).
struct StemArgs {
  SrcPiece pieces[8][2][2];   // [j][h][q] -> channels 8j+4h+2q, +1 of the stem's K order (slot g = 2j+h)
  const float *sc, *sh;       // [CPL] whitening x*sc+sh in the stem's channel order (0 for pad channels)
  const float *wpk;           // pack_stem_weight()
  const float *zero_page;     // >= 16 B of zeros (target of masked gathers)
  float *y;                   // [B,Ho,Wo,COUT] raw conv output
  float *stats;               // [B,slots,COUT,2]
  int B, H, W, Ho, Wo, CPL, slots, tiles_x, tiles_y;
  int dbg, lds_pad;           // experiment knobs (PNVO_STEM_DBG="<flags>,<lds_pad_bytes>"): 1 skip staging, 2 skip epilogue
  int paired;                 // 1: stand in for an 8 x 16-tile stem — slots = ceil(Ho/8) * ceil(Wo/16), partials of a tile pair summed
  const int *only_if;         // paired: predicate read on the DEVICE (nullptr: always run) — the launch is a no-op while *only_if == 0
  int *publish;               // paired + only_if: host-mapped copy of the flag, set by this launch when the flag is up
};
int stem_tiles_x(int Wo);
int stem_tiles_y(int Ho);
hipError_t launch_stem_lds(const StemArgs &a, int cout, hipStream_t s);
void pack_stem_weight(const float *oihw_new, int cout, int cinp, int cpl, float *out);

// The one-hot-aware fused stem (stem_dd.hip): dense channels on the matrix cores, one-hot depth bins as a table gather.
struct StemDDArgs {
  SrcPiece pieces[8][2][2];   // [slot][0][q]: dense channels 4*slot + 2q (rgb | depth | tdv | indicator | pad) of 12
  const float *sc, *sh;       // [12] whitening of the dense channels; indicator: sc = 0, sh = 1
  const float *wpk;           // pack_stem_dd_weight of the dense + indicator weights
  const float *table;         // [7 kh][slice]: [7 kw][bins + 1][2 frames][32] one-hot weight rows (row `bins` = 0)
  const float *dd;            // discretised-depth tensor [B,H,W,2*bins]
  const float *zero_page;
  int *bad_onehot;            // device flag: a depth pixel was not one-hot
  float *y, *stats;
  int B, H, W, Ho, Wo, bins, slots, tiles_x, tiles_y, slice_floats;
  int dbg;                    // experiment variants (PNVO_STEM_DBG); 9 = per-phase cycle counters into prof[4]
  unsigned long long *prof;
};
int stem_dd_slice_floats(int bins);
bool stem_dd_supported(int bins);
int stem_dd_slots(int Ho, int Wo);                  // GroupNorm partial-sum slots per sample (tiles)
void pack_stem_dd_weight(const float *w_o12t, int cout, float *out);
hipError_t launch_stem_dd(const StemDDArgs &a, hipStream_t s);
hipError_t launch_stem_dd_repack(const float *w_oihw, int cin, const float *sc_new, const float *sh_new, const int *dense_ref,
                                 const int *dense_new, int nd, const int *dd_ref, const int *dd_new, int bins,
                                 float *table, float *wpk, float *sc12, float *sh12, hipStream_t s);

// The stem on the bf16 matrix cores (stem_mx.hip): float32 results from exact three-piece bf16 weights, or native bf16.
struct StemMXArgs {
  const float *src[4];        // rgb, depth, dd, tdv observation tensors (nullptr if absent)
  const unsigned short *wpk;  // pack_stem_mx_weight(): [49 taps][fragments][N-tiles][64 lanes][8 bf16]
  const float *zero_page;     // >= 128 B of zeros (target of out-of-image reads)
  int *bad_input;             // host-visible flag: a value that must be exact in bf16 was not (PIECES = 3)
  void *y[4];                 // per N-tile (32 output channels): output tensor [B,Ho,Wo,y_cstride] float or bf16
  float *stats[4];            // per N-tile: [B,slots,stats_cstride,2]
  int y_coff[4];              // per N-tile: first channel inside y / stats
  int y_cstride, stats_cstride;
  int B, H, W, Ho, Wo, slots, tiles_x, tiles_y;
  unsigned long long *prof;   // [8 waves][8] phase cycle sums (PNVO_STEM_DBG=9) or nullptr
  // pooled output (float32 results only): [B,Hp,Wp,y_cstride] order-preserving integer keys of max over the 3x3/2 window of
  // sgn(pool_gamma[c]) * x, pre-set to STEM_POOL_INIT; y is not written.  nullptr: the raw output goes to y
  int *pool;
  const float *pool_gamma;    // GroupNorm weight of the stem [cout] (its sign decides max or min)
  int Hp, Wp;
  float oscale;               // PIECES = 2: inverse of the power-of-two scale folded into the packed weights
  const float *oscale_ptr;    //   ... or where it lives on the device (training: the scale follows the weights)
  // RAW staging (pnvo_forward_raw): sensor frames instead of src[0..2]; src[3] stays the top-down view pair tensor
  const unsigned char *raw_rgb;   // [B][2][H][W][3] uint8 or nullptr (model without rgb)
  const float *raw_depth;         // [B][2][H][W] float32; non-null selects the RAW stager
  int raw_flags;                  // bit 0: the model has the depth modality, bit 1: discretised depth
  int *raw_err;                   // device flag: a depth outside [0, 1]
  float edges[12];                // bin edges e_0 .. e_10 of the one-hot depth (float32(i / 10))
  // developer ablations of stem_rs_kernel (option stem_dbg = 16 + bits; WRONG RESULTS, timing only)
  int dbg;
  // grouped forward (see ConvX3Args): operands of models 1 / 2 (stem_mx_kernel only)
  int grp_end0, grp_end1;
  const unsigned short *wpk_g[2];
  float oscale_g[2];
  const float *pool_gamma_g[2];
};
constexpr int STEM_POOL_INIT = (int)0x807fffffu;   // key of -inf
int stem_mx_slots(int Ho, int Wo);
size_t stem_mx_packed_u16(int pieces, int ntiles);
void pack_stem_mx_weight(const float *wk, int cout, int pieces, const int *xslot, unsigned short *out);
float pack_stem_mx_weight_h(const float *wk, int cout, const int *xslot, unsigned short *out);   // two float16 pieces -> oscale
hipError_t launch_stem_mx(const StemMXArgs &a, int pieces, int ntiles_n, bool bf16_out, hipStream_t s);
bool stem_rs_takes(const StemMXArgs &a, int pieces, int ntiles_n, bool bf16_out, int wgs);   // persistent, weights resident in registers
hipError_t launch_stem_rs(const StemMXArgs &a, int pieces, bool fast, int wgs, hipStream_t s);   // fast: see stem_rs.hip (FAST)                      //   (stem_rs.hip)
hipError_t launch_stem_mx_repack_h(const float *w_oihw, int cin, const float *sc_new, const float *sh_new, const int *slot_ref,
                                   const int *slot_new, const int *xslot, float *scale2, unsigned short *wpk2, hipStream_t s);
hipError_t launch_stem_mx_repack(const float *w_oihw, int cin, const float *sc_new, const float *sh_new, const int *slot_ref,
                                 const int *slot_new, const int *xslot, unsigned short *wpk3, hipStream_t s);

// float32 convs on the bf16 matrix cores by three-piece operand splitting (conv_x3.hip)
struct ConvX3Args {
  const float *x;                    // [B,H,W,CIN] float32 NHWC
  const unsigned short *wpk;         // pack_conv_x3_weight()
  float *y;                          // [B,Ho,Wo,COUTP] raw output
  const float *in_scale, *in_shift;  // [B,CIN] MODE 1: relu(x*scale+shift) applied while staging
  // MODE 2 (fused BasicBlock tail, resnet.py:47-55): the conv's input is relu(x*scale+shift + r), r = res (final activations)
  // or res*res_scale+res_shift (downsample branch); computed while staging and written to xout by the tile that owns the pixel
  const float *res, *res_scale, *res_shift;
  float *xout;
  float *stats;                      // [B,slots,COUTP,2] GroupNorm partial sums or nullptr
  int B, H, W, CIN, Ho, Wo, COUTP;
  int TR, TC, tiles_r, tiles_c, PR, PC, CK, MT, wn, slots;   // filled by conv_x3_plan
  unsigned long long *prof;          // PNVO_X3_PROF=1: phase cycles of one workgroup (nullptr otherwise)
  int force;                         // conv_x3_plan: take the layer at any launch size (option conv=x3)
  // np == 2 on a GRADIENT input (backward-data): float bits of max |x| over the input tensor (gn_bwd_apply's absmax); the stager
  // multiplies by 2^(14 - e) before the float16 split, the epilogue divides again (both exact)
  const unsigned *in_absmax;         //   (64 slots of 16 uints: readers take the maximum, absmax_of())
  int rs_bands, rs_rows;             // conv_rows32_plan (conv_rows.hip): bands per sample, rows per band
  int rs_dbg;                        // developer ablations (env PNVO_ROWS_DBG; WRONG RESULTS, timing only): 1 no loads, 2 no stores, 4 no MFMAs, 8 no conversion
  int np;                            // operand pieces: 3 = bf16 (six exact product terms; 0 means 3), 2 = float16 (three terms)
  float oscale;                      // np == 2: inverse of the power-of-two scale folded into the packed weights
  const float *oscale_ptr;           //   ... or where it lives on the device (training: the scale follows the weights)
  int persist_wgs;                   // > 0: eligible launches take conv_x3p_kernel with this many workgroups (3 per CU); 0: never
  int w8_ok, w8;                     // option x3_w8 / conv_x3_plan's decision: eight waves per workgroup (two per SIMD) — 256-channel 6 x 11 maps, one tile per sample
  int ksw_ok, ksw;                   // option x3_ksplit / conv_x3_plan's decision: fine-plan tiles of three / four M-tiles split K over the four waves
  int fine;                          // conv_x3_plan: allow the fine plan (one N-tile per workgroup) for launches below 224 workgroups (option x3_fine)
  int strip;                         // conv_x3_plan: wide strip tiles with the N-tiles split over blockIdx.y for 64 / 128 output channels
  // GroupNorm finalisation inside the conv (slots == 1: the workgroup that wrote a sample's only partial sums holds the complete sums
  // of its channels): scale / shift [B,COUTP] as gn_finalize_kernel would write them, bit for bit; nullptr: the separate launch
  const float *gn_gamma, *gn_beta;
  float *gn_scale, *gn_shift;
  float *gn_mu, *gn_rstd;            // [B,groups] mean / reciprocal standard deviation for the backward pass, or nullptr
  int gn_cpg;                        // channels per group (divides 32)
  float gn_eps;
  long gn_P;                         // pixels per sample
  // slots > 1 with gn_scale set: [B][gridDim.y] arrival counters (zero between launches) — the LAST tile of a sample (and N-tile
  // group) finalises (gn_finalize_group_wave: gn_finalize_kernel's arithmetic bit for bit); nullptr with slots > 1: the launch
  unsigned *gn_ctr;
  // The block's 1x1 stride-2 downsample conv riding on a 3x3 stride-2 launch (conv_x3_kernel<.., DSF = true>; ds_wpk == nullptr: none):
  // its float16-piece operand (pack_conv_x2_weight of the 1x1 weight), raw output [B,Ho,Wo,COUTP], partial sums (the layout of `stats`,
  // a buffer of its own), weight scale, and its GroupNorm (same groups as the conv's; scale / shift written when the conv's are)
  const unsigned short *ds_wpk;
  float *ds_y, *ds_stats;
  float ds_oscale;
  const float *ds_oscale_ptr;
  const float *ds_gamma, *ds_beta;
  float *ds_scale, *ds_shift;
  float *ds_mu, *ds_rstd;            // [B,groups] for a backward pass, or nullptr
  // GROUPED forward (pnvo_forward_grouped_raw: pairs of up to three action models in one launch chain, sorted by model): sample n
  // belongs to model (n >= grp_end0) + (n >= grp_end1), an end of 0 meaning "no such model" (the structs are zero-filled); models 1 / 2
  // take their operands from [0] / [1]
  // DEFERRED GroupNorm finalisation (round 6, option gn_defer): the producer of this conv's input skipped its gn_finalize launch; this
  // kernel turns the producer's partial sums into the scale / shift table of ITS sample in its prologue (fin_in: the input's GroupNorm,
  // replaces in_scale / in_shift; fin_res: the skip branch's, replaces res_scale / res_shift) — gn_finalize_kernel's arithmetic bit for
  // bit (gn_finalize_wave16), overlapped with the first patch loads.  stats == nullptr: the tables come from memory as before.
  struct Fin {
    const float *stats;              // [B][slots][CIN][2] partial sums of the producer
    int slots, cpg;
    const float *gamma, *beta, *gamma_g[2], *beta_g[2];   // affine parameters (and those of models 1 / 2 of a grouped forward)
  } fin_in, fin_res;
  int grp_end0, grp_end1;
  const unsigned short *wpk_g[2], *ds_wpk_g[2];
  float oscale_g[2], ds_oscale_g[2];
  const float *gn_gamma_g[2], *gn_beta_g[2], *ds_gamma_g[2], *ds_beta_g[2];
};
#if defined(__HIPCC__)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// One lane = one channel of a sample, lanes of a group adjacent and aligned: GroupNorm scale / shift from the channel's complete
// sums (s1, s2), in gn_finalize_kernel's arithmetic — fp64, the butterfly over the group's lanes in its order (the wider offsets
// of that kernel's 64-lane butterfly add exact zeros), no a*b+c contraction (elementwise.hip is compiled without it).
__device__ __forceinline__ void gn_finalize_lane(float s1, float s2, int cpg, long P, float eps, float gamma, float beta, float *scale, float *shift,
                                                 float *mu_out = nullptr, float *rstd_out = nullptr) {
#pragma clang fp contract(off)
  double d1 = (double)s1, d2 = (double)s2;
  for (int o = cpg >> 1; o >= 1; o >>= 1) {
    d1 += __shfl_xor(d1, o);
    d2 += __shfl_xor(d2, o);
  }
  const double cnt = (double)P * cpg;
  const double mu = d1 / cnt;
  double var = d2 / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (mu_out != nullptr) {                                   // (the group's first lane, as thread 0 of gn_finalize_kernel's block)
    *mu_out = (float)mu;
    *rstd_out = (float)rstd;
  }
  const double sc = rstd * (double)gamma;
  *scale = (float)sc;
  *shift = (float)((double)beta - mu * sc);
}
// ---- GroupNorm finalisation by the LAST workgroup of a sample (multi-slot layers: no finalisation launch) ------------------------
// Every workgroup that wrote GroupNorm partial sums of sample n bumps a device-scope counter; the one that brings it to `expect`
// knows that all partials of the sample are in memory and turns them into scale / shift — gn_finalize_kernel's arithmetic, bit for
// bit (one wave per group: the same lane -> (slot, channel) walk, fp64, the same 64-lane butterfly), so which workgroup arrives
// last changes nothing.  The partials cross workgroups INSIDE a launch: they are written and read with agent-scope accesses (sc1:
// coherent across the per-XCD L2s, smallnet.hip's protocol), the counter is bumped after the writer's stores are acknowledged
// (s_waitcnt vmcnt(0) + workgroup barrier) — no L2 write-back or invalidate.  The last arriver also clears the counter for the next launch.
__device__ __forceinline__ void gn_stats_store(float *dst, float s1, float s2) {
  __hip_atomic_store(reinterpret_cast<unsigned *>(dst), __builtin_bit_cast(unsigned, s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned *>(dst) + 1, __builtin_bit_cast(unsigned, s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One full wave = one (sample, group): `stats` points at the sample's [slots][CP][2] partials, ns slots are summed.
__device__ __forceinline__ void gn_finalize_group_wave(const float *stats, int ns, int CP, int g, int cpg, long P, float eps, const float *gamma,
                                                       const float *beta, float *scale_n, float *shift_n, float *mu_out, float *rstd_out) {
#pragma clang fp contract(off)
  const int lane = (int)(threadIdx.x & 63);
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < ns * cpg; k += 64) {
    const int slot = k / cpg, c = g * cpg + k % cpg;
    const unsigned *src = reinterpret_cast<const unsigned *>(stats + ((long)slot * CP + c) * 2);
    s1 += (double)__builtin_bit_cast(float, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    s2 += (double)__builtin_bit_cast(float, __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  const double cnt = (double)P * cpg;
  const double mu = s1 / cnt;
  double var = s2 / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (lane == 0 && mu_out != nullptr) {
    *mu_out = (float)mu;
    *rstd_out = (float)rstd;
  }
  for (int k = lane; k < cpg; k += 64) {
    const int c = g * cpg + k;
    const double sc = rstd * (double)gamma[c];
    scale_n[c] = (float)sc;
    shift_n[c] = (float)((double)beta[c] - mu * sc);
  }
}
// Arrive (whole workgroup, behind its partial-sum stores); true for every thread of the workgroup that arrived last.
// `flag` = one int of LDS scratch the caller no longer reads.
__device__ __forceinline__ bool gn_last_arrival(unsigned *ctr, unsigned expect, int *flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old + 1u == expect;
    if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
// One WAVE finalises SIXTEEN groups at once — gn_finalize_kernel's result bit for bit with four lanes per group instead of a block
// of 64.  That kernel: lane k of 64 adds elements k, k + 64, ... (element e = (slot e / cpg, channel g cpg + e % cpg)) in fp64, then a
// xor-butterfly over 32, 16, ..., 1 — after every step the partner lanes hold the same value, so the result is the fixed tree
// ((x[v] + x[v ^ 32]) + ...) whatever lane reads it.  Here lane 4 g' + q holds the sixteen "virtual lanes" v = 16 q + i of group g0 + g':
// steps 32 and 16 are xor-shuffles across the four lanes (q ^ 2, q ^ 1), steps 8 ... 1 run inside the lane.  Same operands, same
// order (a + b == b + a), no a*b+c contraction.  Groups >= G are skipped.  scale_tab / shift_tab: the sample's tables (LDS).
__device__ __forceinline__ void gn_finalize_wave16(const float *stats_n, int ns, int CP, int g0, int G, int cpg, long P, float eps,
                                                   const float *gamma, const float *beta, float *scale_tab, float *shift_tab) {
#pragma clang fp contract(off)
  const int lane = (int)(threadIdx.x & 63), q = lane & 3, g = g0 + (lane >> 2);
  const bool live = g < G;
  const int ne = ns * cpg;
  double x1[16], x2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x1[i] = 0.0;
    x2[i] = 0.0;
  }
  if (live) {
    for (int e0 = 0; e0 < ne; e0 += 64) {                       // (one pass whenever slots * cpg <= 64)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = e0 + 16 * q + i;
        if (e < ne) {
          const int slot = e / cpg, c = g * cpg + e % cpg;
          const f32x2_t v = *reinterpret_cast<const f32x2_t *>(stats_n + ((long)slot * CP + c) * 2);
          x1[i] += (double)v[0];
          x2[i] += (double)v[1];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {                                 // butterfly steps 32 and 16: across the group's four lanes
    x1[i] += __shfl_xor(x1[i], 2);
    x2[i] += __shfl_xor(x2[i], 2);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x1[i] += __shfl_xor(x1[i], 1);
    x2[i] += __shfl_xor(x2[i], 1);
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1)                               // steps 8, 4, 2, 1: inside the lane
#pragma unroll
    for (int i = 0; i < o; ++i) {
      x1[i] += x1[i + o];
      x2[i] += x2[i + o];
    }
  if (!live) return;
  const double cnt = (double)P * cpg;
  const double mu = x1[0] / cnt;
  double var = x2[0] / cnt - mu * mu;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int t = q; t < cpg; t += 4) {
    const int c = g * cpg + t;
    const double sc = rstd * (double)gamma[c];
    scale_tab[c] = (float)sc;
    shift_tab[c] = (float)((double)beta[c] - mu * sc);
  }
}
#endif
bool conv_x3_plan(ConvX3Args &a, int ks, int stride, int *mw, int *nw, size_t *lds_bytes);
// Row-streaming form of the 32 -> 32 channel 3x3 stride-1 convs (conv_rows.hip): plan fills rs_bands / rs_rows / slots.
bool conv_rows32_plan(ConvX3Args &a, int ks, int stride, int mode, int num_cus);
hipError_t launch_conv_rows32(const ConvX3Args &a, int mode, int num_cus, hipStream_t s);
bool conv_x3_persistent(const ConvX3Args &a, int ks, int stride, int mode, int mw, int nw);   // would launch_conv_x3 take conv_x3p_kernel?
hipError_t launch_conv_x3(const ConvX3Args &a, int ks, int stride, int mode, int mw, int nw, size_t lds_bytes, hipStream_t s);
hipError_t launch_conv_x3_repack(const float *w_oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, int transposed,
                                 unsigned short *out, hipStream_t s);
void pack_conv_x3_weight(const float *oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, unsigned short *out);
hipError_t launch_conv_x2_scales(const float *params, const long *seg_dev, int nseg, float *out, hipStream_t s);   // [nseg][2]: scale, 1/scale
hipError_t launch_conv_x2_repack(const float *w_oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, const float *scale_dev,
                                 unsigned short *out, hipStream_t s, int transposed = 0);
float pack_conv_x2_weight(const float *oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, unsigned short *out);   // -> oscale

// Native-bf16 convs of the residual stages (conv_bf16.hip); index [z] = model of the launch (dual forward: two).
struct ConvBArgs {
  const unsigned short *x[2];        // [B,H,W,CIN] bf16 NHWC
  const unsigned short *wpk[2];      // pack_conv_bf16_weight()
  void *y[2];                        // [B,Ho,Wo,COUTP] raw output, bf16 (or float for the compression conv)
  const float *in_scale[2], *in_shift[2];   // [B,CIN] MODE 1: relu(x*scale+shift) applied while staging
  // MODE 2 (fused BasicBlock tail, resnet.py:47-55): the conv's input is relu(x*scale+shift + r), r = x2 (final activations)
  // or x2*scale2+shift2 (downsample branch); computed while staging and written to xout by the tile that owns the pixel
  const unsigned short *x2[2];
  const float *in_scale2[2], *in_shift2[2];
  unsigned short *xout[2];
  float *stats[2];                   // [B,slots,COUTP,2] GroupNorm partial sums
  int B, H, W, CIN, Ho, Wo, COUTP;
  int TR, TC, tiles_r, tiles_c, PR, PC, CK, MT, wn, slots;   // filled by conv_bf16_plan
  int persist_wgs;                   // > 0: 32-input-channel layers run persistent with resident weights on this many workgroups
  // the block's 1x1 stride-2 downsample conv riding on a 3x3 stride-2 launch (conv_bf16_kernel DSF; ds_wpk[0] == nullptr: none)
  const unsigned short *ds_wpk[2];
  void *ds_y[2];
  float *ds_stats[2];
  const float *ds_gamma[2], *ds_beta[2];
  float *ds_scale[2], *ds_shift[2];
  // GroupNorm finalisation inside the conv (slots == 1, as ConvX3Args): scale / shift [B,COUTP] per model; nullptr: the separate launch
  const float *gn_gamma[2], *gn_beta[2];
  float *gn_scale[2], *gn_shift[2];
  int gn_cpg;
  float gn_eps;
  long gn_P;
};
bool conv_bf16_plan(ConvBArgs &a, int ks, int stride, int *mw, int *nw, size_t *lds_bytes);
hipError_t launch_conv_bf16(const ConvBArgs &a, int ks, int stride, int mode, bool f32out, int mw, int nw, size_t lds_bytes,
                            int nmodels, hipStream_t s);
void pack_conv_bf16_weight(const float *oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, unsigned short *out);
hipError_t launch_gn_relu_maxpool_bf16(const unsigned short *const *x, const float *const *scale, const float *const *shift,
                                       int B, int H, int W, int C, unsigned short *const *out, int nmodels, hipStream_t s);
hipError_t launch_residual_bf16(const unsigned short *const *a, const float *const *sa, const float *const *ta,
                                const unsigned short *const *b, const float *const *sb, const float *const *tb, int B, long P,
                                int C, unsigned short *const *y, int nmodels, hipStream_t s);

int conv_slots(int P, int MT);                       // stats slots per sample for a given wave tile
void choose_tile(long M, int COUTP, int *MT, int *NT);
hipError_t launch_conv(const ConvArgs &a, hipStream_t s);
// y = [relu](sum_z kpart[z] + bias[row]) for a split-K linear layer (a.ksplit > 1)
hipError_t launch_ksplit_reduce(const ConvArgs &a, hipStream_t s);
// ... with the output head (Linear COUT -> out_dim <= 4, plain [out_dim][COUT] weight + bias) computed on the reduced row: out [B][out_dim]
// fc_rows.hip: the hidden layer (+ GroupNorm / ReLU of its input) and the output head for <= 32 samples, every action model of a grouped
// forward in one launch each
constexpr int FC_ROWS_MAXV = 12;          // 16-byte vectors of a weight row per lane: Kp <= 3072
struct FcRowsArgs {
  const float *x, *sc, *sh;               // compression conv's raw output [B][Kp] (Kp = fh * fw * cp), its GroupNorm scale / shift [B][cp]
  float *hid, *out;                       // [B][hidden], [B][out_dim]
  const int64_t *actions;                 // act-embed models: the bias row of a sample (nullptr: row 0)
  int B, Kp, cp, hidden, out_dim, ngroups;
  int end[3];                             // one past the last sample of each action model
  const float *w[3], *bias[3], *head_w[3], *head_b[3];   // per model: [hidden][Kp] rows in the activation's (h, w, padded c) order, bias rows, head
};
hipError_t launch_fc_rows(const FcRowsArgs &a, bool with_head, hipStream_t s);
hipError_t launch_ksplit_reduce_head(const ConvArgs &a, const float *w2, const float *b2, int out_dim, float *out, hipStream_t s);
int conv_ksplit(const ConvArgs &a);   // how many K slices launch_conv would use for `a` when a.kpart is set (1: none)

// LDS-staged 3x3 stride-1 kernel (conv3_lds.hip): same arguments / packing / epilogue contract as launch_conv.
bool conv3_lds_supported(const ConvArgs &a);
int conv3_lds_slots(const ConvArgs &a);                       // statistics slots per sample (tiles x waves)
hipError_t launch_conv3_lds(const ConvArgs &a, int nt, hipStream_t s);

size_t packed_conv_floats(int cout, int cin, int kh, int kw);
void pack_conv_weight(const float *oihw, int cout, int cin, int kh, int kw, float *out);

// GroupNorm statistics -> per-(sample,channel) scale/shift.
// fixed_ns > 0: every sample has exactly fixed_ns slots (stem tiles); else slots follow the flattened wave tiles.
// mu_out / rstd_out (optional, [B,G]) keep the statistics for the backward pass.
struct GnGroup {                     // grouped forward: affine parameters of models 1 / 2 (sample n -> model as in ConvX3Args)
  int end0, end1;
  const float *gamma[2], *beta[2];
};
hipError_t launch_gn_finalize(const float *stats, int B, int slots, int CP, int C, int G, long P, int WM,
                              const float *gamma, const float *beta, float eps, float *scale, float *shift,
                              hipStream_t s, int fixed_ns = 0, float *mu_out = nullptr, float *rstd_out = nullptr,
                              const GnGroup *grp = nullptr);

// Two GroupNorms of one geometry in one launch (fixed slots per sample), each in gn_finalize_kernel's arithmetic; mu / rstd of set 0.
hipError_t launch_gn_finalize_pair(const float *const *stats, int B, int slots, int CP, int C, int G, long P, const float *const *gamma,
                                   const float *const *beta, float eps, float *const *scale, float *const *shift, float *const *mu_out,
                                   float *const *rstd_out, hipStream_t s, const GnGroup *grp0 = nullptr, const GnGroup *grp1 = nullptr);
hipError_t launch_gn_finalize2(const float *const *stats, int B, int slots, int CP, int C, int G, long P, const float *const *gamma,
                               const float *const *beta, float eps, float *const *scale, float *const *shift, int nmodels,
                               hipStream_t s);

// Input assembly + whitening (vo_cnn.py:110-176) into channel-padded NHWC.
struct AssembleArgs {
  const float *src[4];   // rgb, depth, dd, tdv (nullptr if absent)
  int nsrc[4];           // pair channel counts
  const float *mean;     // [C] or nullptr
  const float *stdev;    // [C]
  int C, CP;
  long npix;             // B*H*W
  float *out;            // [npix, CP]
};
hipError_t launch_assemble(const AssembleArgs &a, hipStream_t s);

// relu(gn(x)) then MaxPool 3x3 s2 p1.
hipError_t launch_gn_relu_maxpool(const float *x, const float *scale, const float *shift, int B, int H, int W,
                                  int C, float *out, hipStream_t s);

// *host_flag = 1 if *dev_flag != 0 (one lane): the input-contract flag for the handle's host-side decisions.
hipError_t launch_flag_publish(const int *dev_flag, int *host_flag, hipStream_t s);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: true exactly once per (device, `seen` mask) — the caller then sets the
// attributes of its kernels for that device.  Serialised by `mu` (two handles on two GPUs may launch from two threads).
inline bool pnvo_first_launch_on_device(std::mutex &mu, unsigned long long &seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return true;
  const unsigned long long bit = 1ull << (dev & 63);
  std::lock_guard<std::mutex> lk(mu);
  if (seen & bit) return false;
  seen |= bit;
  return true;
}

// Pooled order-preserving keys (the fused stems' POOL output) from a raw stem output; a no-op while *only_if == 0 (device-side).
hipError_t launch_pool_keys_from_raw(const float *x, const float *gamma, int B, int H, int W, int C, int *keys, const int *only_if,
                                     hipStream_t s);

// y = relu(a*sa+ta + r)  with r = b (plain) or b*sb+tb.  a,b,y: [B,P,C].
hipError_t launch_residual(const float *a, const float *sa, const float *ta, const float *b, const float *sb,
                           const float *tb, int B, long P, int C, float *y, hipStream_t s);

// y = relu(x*s+t)  (materialise a normalised activation; used for taps only)
hipError_t launch_apply_ss_relu(const float *x, const float *s, const float *t, int B, long P, int C, float *y,
                                hipStream_t st);

hipError_t launch_discretize_depth(const float *depth, int64_t n, int64_t in_stride, int bins, float *onehot,
                                   int64_t out_stride, int32_t *err_flag, hipStream_t s, const float *edges = nullptr);
hipError_t launch_topdown_f64(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                              const double *consts, int rows_around_center, float *out, int64_t out_fstride,
                              int64_t out_pstride, void *work, hipStream_t s);
hipError_t launch_half_to_float(const unsigned short *src, long n, float *dst, hipStream_t s);
hipError_t launch_dataset_pairs(const unsigned char *prev_rgb, const unsigned char *cur_rgb, const unsigned short *prev_depth,
                                const unsigned short *cur_depth, const float *tdv_frames, const int *src, const int *swap,
                                int N, int M, int H, int W, int bins, const float *edges, float *o_rgb, float *o_depth,
                                float *o_dd, float *o_tdv, int *err_flag, hipStream_t s);
hipError_t launch_ring_assemble(const unsigned char *up_rgb, const float *up_dep, const float *up_tdv, unsigned char *ring_rgb,
                                float *ring_dep, float *ring_tdv, const int *idx, int n, int H, int W, unsigned char *pair_rgb,
                                float *pair_dep, float *pair_tdv, hipStream_t s);
hipError_t launch_frame_pairs(const unsigned char *rgb, const float *depth, int n, int H, int W, int bins, float *o_rgb,
                              float *o_depth, float *o_dd, int *err_flag, hipStream_t s);
size_t topdown_workspace_bytes(int N, int H, int W);
hipError_t launch_topdown(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                          const float *consts_host, int rows_around_center, float *out, int64_t out_fstride,
                          int64_t out_pstride, void *work, hipStream_t s, int64_t out_pair = 0);

// ---- training step (train_kernels.hip) ---------------------------------------------------------------------------
struct SrcLane {            // stem weight gradient: the observation-tensor slot feeding input channel (lane) i
  const float *base;
  int nch, choff;
  float sc, sh;
};

struct WgradArgs {
  const float *x;           // input activation [B,H,W,CIN] (mode 0/1)
  const float *in_scale, *in_shift;   // [B,CIN] (mode 1: relu(x*scale+shift))
  const float *dy;          // [B,Ho,Wo,DYC] gradient of the conv's raw output
  float *partial;           // [units][TG][32][32]
  const float *zero_page;
  SrcLane src[32];          // mode 2
  int B, H, W, CIN, Ho, Wo, COUT, DYC, KH, KW, stride, pad;
  int mode;                 // 0 plain, 1 producer GN+ReLU recomputed, 2 gathered+whitened observation tensors
  int TG, groups, ci_tiles, pairs, chunks;
  long pix_per_chunk;
  // LDS-staged path (wgrad3_lds_kernel: 3x3 stride-1 convs with 32-channel multiples), chosen by wgrad_plan
  int lds3, TH, TW, tiles_x, tiles_y, tiles_per_chunk;
  // twelve-wave path (wgrad3_mw_kernel, lds3 = 4 / 5): units per workgroup = cs ci-tiles x os co-tiles x rs row groups (= 4),
  // workgroups = pair groups x wg_chunks, LDS buffers
  int cs, os, rs, wg_chunks, nbuf;
  long grad_pitch;          // row pitch of the OIHW gradient tensor (0: cin_out * KH * KW)
  // bf16-matrix-core path (wgrad_x3.hip, lds3 = 6): row segments per strip; use_x3 = 0 keeps wgrad_plan off it
  int xr_rsegs, use_x3;
  // np = 2: two float16 pieces per operand, three terms (X bounded by the forward's range check; dY scaled by 2^(14 - e) from
  // dy_absmax, un-scaled on the accumulators); np = 3 (or 0): three bf16 pieces, six terms
  int np;
  const unsigned *dy_absmax;
};
bool wgrad_x3_plan(WgradArgs &a);
hipError_t launch_wgrad_x3(const WgradArgs &a, hipStream_t s);
// Weight gradient of the 7x7 stem on the bf16 matrix cores (wgrad_stem_mx.hip): exact three-piece bf16 on both operands.
struct WgradStemMXArgs {
  const float *src[4];        // rgb, depth, discretised depth, top-down view observation tensors (nullptr if absent)
  const float *dy;            // [B,Ho,Wo,32] gradient of the stem's raw output
  const float *zero_page;     // >= 16 B of zeros (source of the DMA pieces outside the image)
  float *partial;             // [nwg][49][48][32] (set by the launcher)
  int B, H, W, Ho, Wo;
  int tiles_x, tiles_y, tiles_per_wg, nwg;   // filled by wgrad_stem_mx_plan
  unsigned long long *prof;   // PNVO_WSM_PROF: phase cycle counters of one workgroup, or nullptr
};
void wgrad_stem_mx_plan(WgradStemMXArgs &a);
size_t wgrad_stem_mx_scratch_floats(const WgradStemMXArgs &a);
hipError_t launch_wgrad_stem_mx(const WgradStemMXArgs &a, float *scratch, const float *sc_new, const float *sh_new, const int *slot_ref,
                                const int *slot_new, int cin, float *grad, hipStream_t s);
void wgrad_plan(WgradArgs &a);
size_t wgrad_partial_floats(const WgradArgs &a);
hipError_t launch_wgrad(const WgradArgs &a, float *grad, const int *ci_perm, int cin_out, hipStream_t s);

hipError_t launch_gn_bwd(const float *x, const float *dout, const float *scale, const float *shift, const float *mu,
                         const float *rstd, const float *gamma, int B, long P, int C, int Creal, int G, int mask,
                         float *part, float *coef, float *dgamma, float *dbeta, float *dx, hipStream_t s, unsigned *absmax = nullptr);
hipError_t launch_gn_bwd_pool(const float *x, const float *dpool, const unsigned char *idx, int Hs, int Ws, int Hp, int Wp,
                              const float *scale, const float *shift, const float *mu, const float *rstd, const float *gamma, int B, int C,
                              int G, float *part, float *coef, float *dgamma, float *dbeta, float *dx, hipStream_t s);
hipError_t launch_relu_mask(const float *dy, const float *y, const float *add, long n, float *g, hipStream_t s);
hipError_t launch_add(const float *a, const float *b, long n, float *o, hipStream_t s);
hipError_t launch_maxpool_train(const float *x, const float *scale, const float *shift, int B, int H, int W, int C,
                                float *out, unsigned char *idx, hipStream_t s);
hipError_t launch_maxpool_bwd(const float *dpool, const unsigned char *idx, int B, int H, int W, int C, float *dact,
                              hipStream_t s);
hipError_t launch_rmv_merge(const float *m12, int C, int B, float *mean, float *var, float *count, hipStream_t s);
hipError_t launch_colsum(const float *x, int rows, int cols, int ld, float *out, hipStream_t s);
hipError_t launch_padcopy(const float *src, int rows, int cols, int ldd, float *dst, hipStream_t s);
// y = [relu(x*scale[n,c]+shift[n,c]) or x] * keep/(1-p); keep = hash(seed, step, layer, element) >= p.  x viewed as
// [B][row] with row = P*C elements (C = channel count for the scale index); x == nullptr writes the bare scaled mask.
hipError_t launch_dropout(const float *x, const float *scale, const float *shift, int B, long P, int C, float p,
                          uint64_t seed, uint64_t step, int layer, float *y, hipStream_t s);
hipError_t launch_embed_gather(const float *emb, const long long *actions, int B, int rows, float *out, int *err, hipStream_t s);
hipError_t launch_embed_bias(const float *efeat, const float *w1, long pitch, int flat, const float *b1, int B, int hidden,
                             float *bias, hipStream_t s);
hipError_t launch_embed_backward(const float *gh, const float *efeat, const float *w1, long pitch, int flat, int B, int hidden,
                                 float *dw1, float *dfeat, hipStream_t s);
hipError_t launch_embed_scatter(const float *dfeat, const long long *actions, int B, int rows, float *demb, hipStream_t s);
hipError_t launch_mse_loss(const float *pred, const float *target, int B, int D, float *loss, float *grad, hipStream_t s);
hipError_t launch_mse_loss_coef(const float *pred, const float *target, const float *coef, int n, float *loss, float *grad,
                                hipStream_t s);
hipError_t launch_geo_inverse_loss(const float *deltas, const int *actions, int P, int move_forward, float weight, float *out,
                                   float *grad, hipStream_t s);
hipError_t launch_adam(float *p, const float *g, float *m, float *v, long n, float lr, float b1, float b2, float eps,
                       int step, hipStream_t s);
hipError_t launch_gather(const float *src, const int *map, long n, float *dst, hipStream_t s);
struct GatherSeg {
  const int *map;
  float *dst;
  long start;
};
hipError_t launch_gather_all(const float *src, const GatherSeg *segs, int nseg, long total, hipStream_t s);
hipError_t launch_whiten_table(const float *mean, const float *var, const int *ref_of_new, const int *tensor_of_new, int CPL,
                               float *sc, float *sh, hipStream_t s);
struct MomentsArgs {
  const float *src[4];
  int nch[4];
  int tensor[64], ch[64];   // reference channel c -> (observation tensor, channel inside it)
  const float *center;      // [C] or nullptr
  long npix;
  int pw;                   // 1: mean of (x - center), 2: mean of (x - center)^2, 3: both (out[0..C), out[C..2C))
  int nblk[4];              // filled by launch_moments: blocks of each tensor (proportional to its channel count)
};
constexpr int MOMENTS_BLOCKS = 2048;   // partial slots per row: `part` holds 2 * C * MOMENTS_BLOCKS doubles
hipError_t launch_moments(const MomentsArgs &a, int C, double *part, float *out, hipStream_t s);

}  // namespace pnvo
