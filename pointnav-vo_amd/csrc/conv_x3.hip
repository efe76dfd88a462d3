// conv_x3.hip — float32 convs of the residual stages on the bf16 matrix cores, gfx950 only.
//
// The 3x3 convs after the stem (resnet.py:29-55,189-212; 16 of them = 41.6 % of the MACs) are bound
// by the fp32 matrix pipe (157 TFLOP/s; the fp32 kernels of conv3_lds.hip / conv_mfma.hip run at 100-119).  The bf16 pipe is 16
// times faster, and a float32 number is EXACTLY the sum of three bf16 numbers (hi + mid + lo: 3 x 8 significand bits), so
//   a * w = (a0 + a1 + a2)(w0 + w1 + w2) ~ a0 w0 + a0 w1 + a1 w0 + a0 w2 + a2 w0 + a1 w1
// with every kept product exact in the float32 accumulator and the three dropped ones (a1 w2, a2 w1, a2 w2) below 2^-23 of
// a * w — the size of one float32 rounding.  Six bf16 MFMAs replace sixteen fp32-MFMA issue slots of the same K: 2.7 x the
// fp32 rate at equal pipe utilisation, float32-grade results (measured against the fp64 reference in tests/test_gpu_parity.py,
// same tolerances as the fp32 kernels).  The stem (stem_mx.hip) uses the exact special case (inputs exact in bf16).
//
// Structure = conv_bf16.hip (implicit GEMM on v_mfma_f32_32x32x16_bf16, input patch of the workgroup's output tile staged in
// LDS, producer's GroupNorm + ReLU applied once per element while staging, zero padding after it, pixel table for ragged
// tiles, raw float32 output + per-slot GroupNorm partial sums) with float32 activations in HBM: the stager splits every
// element into its three pieces (three LDS planes), the weights are split at load (three B fragments per step).
//
// Input modes of the stager (template MODE):
//   0  final activations
//   1  relu(x * scale + shift): the producer's GroupNorm + ReLU
//   2  the previous BasicBlock's tail (resnet.py:47-55): relu(x * scale + shift + r), r = res or res * res_scale + res_shift
//      (downsample branch); the tile that owns an input pixel also writes the result to xout (the block output the next skip
//      branch / downsample conv reads).  Same float operations as residual_kernel: bit-identical to the separate pass.
//   3  pooled stem keys (stem_mx.hip POOL): decode, * |scale| + shift, ReLU; pooled activations written to xout.
// Also covers the 1x1 stride-2 downsample convs (KS = 1: the stager reads only the pixels the conv uses).
// conv_x3_plan() takes a layer only when its launch has >= 192 workgroups (PNVO_CONV=x3 forces it).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 r = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pack2h(float a, float b) {
  const f16x2 r = __builtin_convertvector(f32x2{a, b}, f16x2);    // v_cvt_pk_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
}  // namespace

// NP = 3: three bf16 pieces per operand, six product terms (every kept product exact).  NP = 2: two float16 pieces per operand
// (22 significant bits each), three terms a1 w0 + a0 w1 + a0 w0 on v_mfma_f32_32x32x16_f16: each product within 3 * 2^-22 of a * w —
// the size of a few float32 roundings of the accumulation it feeds — at half the matrix-pipe time, two LDS planes instead of
// three and 5 instead of 11 VALU per channel pair in the stager.  The weights are pre-scaled by a power of two (p.oscale undoes
// it exactly in the epilogue) so that their second piece stays in float16's normal range.
// Waves per SIMD: the two-plane patch of the small wave tiles (32- / 64-channel stages) is <= 54 KB, so three workgroups fit a
// CU; their accumulators are few, so 168 registers do.
// (DSF — the downsample conv riding on a stride-2 launch — doubles the accumulators: (2,1) tiles drop to two waves per SIMD, the
//  96 + 96-accumulator (3,2) tile to one: its launches are one workgroup per CU anyway, the 6 x 11 maps)
constexpr int x3_wpe(int mw, int nw, int np, bool dsf = false) {
  return dsf ? (mw * nw >= 6 ? 1 : (mw * nw >= 2 ? 2 : 3)) : ((np == 2 && mw * nw <= 2) ? 3 : 2);
}

// DSF (KS = 3, STRIDE = 2, NP = 2: the first conv of a stride-2 BasicBlock, resnet.py:189-212): the block's 1x1 stride-2 DOWNSAMPLE conv
// (resnet.py:192-195) rides on this launch.  It reads the same input pixels as this conv's CENTRE tap — input (2 i, 2 j) for output
// (i, j) — so its A fragments are the ones the K loop already holds in registers during the centre tap's steps; it only needs its own
// B fragments (p.ds_wpk, 1/9 of this conv's), a second set of accumulators and a second epilogue (p.ds_y, p.ds_stats, its own
// GroupNorm).  One launch and one GroupNorm finalisation less per stride-2 block, the block input is read once instead of twice, and
// in the block-tail mode (MODE 2) the block input need not be written to HBM at all (p.xout == nullptr): both of its readers are here.
// Same MFMA terms in the same order as the separate 1x1 launch (k-chunks ascending, a1 w0, a0 w1, a0 w0): bit-identical raw output.
// W8: eight waves per workgroup instead of four (wave grid (8 / wn) x wn).  The 256-channel convs on 6 x 11 maps are ONE tile per sample
// = one workgroup per CU at 256 pairs: with four waves of (3,2) tiles each SIMD holds a single wave and its K loop issues MFMAs 75 % of
// the time (in-order issue: the fragment waits are not covered); eight waves of (3,1) tiles are two per SIMD and issue 92 %
// (profiles/r6_experiments.md), staging the patch once with twice the threads.
// KSW (fine plan, round 6): the tile's MT = MW M-tiles are ALL multiplied by every wave, each wave over a quarter of the K steps of every
// staged chunk (B fragments fetched once per workgroup instead of once per wave, MW MFMA triples per fragment pair instead of one); the
// four partial accumulator sets meet in LDS in wave order and wave w finishes M-tile w as before.  For the deep stages of small batches:
// a (1,1) wave tile walks 432 dependent-latency-bound MFMAs alone, a quarter of the K walk with three or four tiles is half of that.
template <int KS, int STRIDE, int MODE, int MW, int NW, int NP, bool DSF = false, bool W8 = false, bool KSW = false>
__global__ __launch_bounds__(W8 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(W8 ? 2 : x3_wpe(MW, NW, NP, DSF), W8 ? 2 : x3_wpe(MW, NW, NP, DSF)))) void conv_x3_kernel(const ConvX3Args p) {
  constexpr int NTH = W8 ? 512 : 256, NWV = NTH / 64;
  constexpr int EMW = KSW ? 1 : MW;                 // M-tiles a wave finishes in the epilogue
  static_assert(!KSW || (NW == 1 && NP == 2 && KS == 3 && !DSF && !W8), "K split over the waves: float16-piece 3x3 convs, one N-tile");
  static_assert(!DSF || (KS == 3 && STRIDE == 2 && NP == 2 && (MODE == 0 || MODE == 2)), "the downsample rides on a float16-piece 3x3 stride-2 conv");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PSTEP = KS == 1 ? STRIDE : 1;      // input pixels per patch pixel (a 1x1 conv stages only what it reads)
  constexpr int CS = KS == 1 ? 1 : STRIDE;         // patch pixels per output pixel
  constexpr int PAD = KS / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  const int ntiles = p.B * p.tiles_r * p.tiles_c;
  const int chunk = (ntiles + 7) >> 3;
  int bid = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);     // consecutive tiles of an XCD are neighbours
  if (bid >= ntiles) return;
  const int tci = bid % p.tiles_c;
  bid /= p.tiles_c;
  const int tri = bid % p.tiles_r;
  const int n = bid / p.tiles_r;
  const int r0 = tri * p.TR, c0 = tci * p.TC;
  const int hi0 = r0 * STRIDE - PAD, wi0 = c0 * STRIDE - PAD;
  // grouped forward: the sample's action model picks the operands (wave-uniform; one model: both ends are INT_MAX)
  const int mdl = (p.grp_end0 > 0 && n >= p.grp_end0) + (p.grp_end1 > 0 && n >= p.grp_end1);   // (0: no such model)
  const unsigned short *g_wpk = mdl == 0 ? p.wpk : p.wpk_g[mdl - 1];
  const unsigned short *g_ds_wpk = mdl == 0 ? p.ds_wpk : p.ds_wpk_g[mdl - 1];
  const float g_oscale = mdl == 0 ? p.oscale : p.oscale_g[mdl - 1], g_ds_oscale = mdl == 0 ? p.ds_oscale : p.ds_oscale_g[mdl - 1];
  const float *g_gamma = mdl == 0 ? p.gn_gamma : p.gn_gamma_g[mdl - 1], *g_beta = mdl == 0 ? p.gn_beta : p.gn_beta_g[mdl - 1];
  const float *g_ds_gamma = mdl == 0 ? p.ds_gamma : p.ds_gamma_g[mdl - 1], *g_ds_beta = mdl == 0 ? p.ds_beta : p.ds_beta_g[mdl - 1];
  // Deferred GroupNorm finalisation (round 6, small launches): the producer of this conv's input (fin_in) / of the skip branch (fin_res)
  // left its partial sums un-finalised; waves 0 and 1 turn them into this sample's scale / shift tables before anything else holds
  // registers (gn_finalize_wave16: gn_finalize_kernel's arithmetic, bit for bit).  One memory round trip + ~0.5 us of fp64 per
  // workgroup: cheaper than a launch while launches are the bound (8-48 pairs), dearer at 256 pairs — the host defers accordingly.
  if ((MODE == 1 || MODE == 2) && (p.fin_in.stats != nullptr || p.fin_res.stats != nullptr)) {
    const int planes0 = NP * ((p.PR * p.PC) * (p.CK * 2 + 16));
    float *ftab0 = reinterpret_cast<float *>(lds + (KSW ? max(planes0, 4 * MW * 4096) : planes0) + (size_t)p.MT * 32 * 8);
    const long P_in = (long)p.H * p.W;
    if (p.fin_in.stats != nullptr && wave == 0) {
      const int Gn = p.CIN / p.fin_in.cpg;
      const float *ga = mdl == 0 ? p.fin_in.gamma : p.fin_in.gamma_g[mdl - 1], *be = mdl == 0 ? p.fin_in.beta : p.fin_in.beta_g[mdl - 1];
      for (int g0 = 0; g0 < Gn; g0 += 16)
        gn_finalize_wave16(p.fin_in.stats + (long)n * p.fin_in.slots * p.CIN * 2, p.fin_in.slots, p.CIN, g0, Gn, p.fin_in.cpg, P_in, 1e-5f, ga, be,
                           ftab0, ftab0 + p.CIN);
    }
    if (MODE == 2 && p.fin_res.stats != nullptr && wave == 1) {
      const int Gn = p.CIN / p.fin_res.cpg;
      const float *ga = mdl == 0 ? p.fin_res.gamma : p.fin_res.gamma_g[mdl - 1], *be = mdl == 0 ? p.fin_res.beta : p.fin_res.beta_g[mdl - 1];
      for (int g0 = 0; g0 < Gn; g0 += 16)
        gn_finalize_wave16(p.fin_res.stats + (long)n * p.fin_res.slots * p.CIN * 2, p.fin_res.slots, p.CIN, g0, Gn, p.fin_res.cpg, P_in, 1e-5f, ga, be,
                           ftab0 + 2 * p.CIN, ftab0 + 3 * p.CIN);
    }
    __syncthreads();
  }
  const int PR = p.PR, PC = p.PC, CK = p.CK;
  const int pitch = CK * 2 + 16;                   // bytes per patch pixel in one piece plane (odd number of 16-byte units)
  const int plane = PR * PC * pitch;               // bytes of one piece plane
  const int npix = p.TR * p.TC;
  // pixel table behind the three planes: [0] patch byte offset of tile pixel q's top-left tap; [1] element offset of its output
  // pixel inside the sample's output plane, bit 31 set when the pixel does not exist
  // (KSW: the tables sit behind the partial-sum area of the K-split reduction as well: the epilogue reads them after it)
  const int tab_off = KSW ? max(NP * plane, 4 * MW * 4096) : NP * plane;
  unsigned *qtab = reinterpret_cast<unsigned *>(lds + tab_off);
  unsigned *otab = qtab + p.MT * 32;
  for (int e = (int)threadIdx.x; e < p.MT * 32; e += NTH) {               // (strip tiles: nine M-tiles = 288 entries)
    const int q = min(e, npix - 1);
    const int tr = q / p.TC, tc = q - tr * p.TC;
    qtab[e] = (unsigned)(((tr * CS) * PC + tc * CS) * pitch);
    const bool ok = e < npix && r0 + tr < p.Ho && c0 + tc < p.Wo;
    otab[e] = (unsigned)((tr * p.Wo + tc) * p.COUTP) | (ok ? 0u : 0x80000000u);
  }

  const int wn = p.wn;                                                   // wave grid: (NWV / wn) x wn
  const int wave_m = wave / wn;
  const int wave_n = (wave & (wn - 1)) + (int)blockIdx.y * wn;            // blockIdx.y = group of wn * NW N-tiles
  const int ntt = p.COUTP >> 5, kct = p.CIN >> 4;                        // N-tiles, 16-channel k-chunks of the layer

  f32x16 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 accd[DSF ? MW : 1][DSF ? NW : 1];                               // DSF: the downsample conv's accumulators
  if (DSF) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[DSF ? i : 0][DSF ? j : 0][r] = 0.f;
  }

  // a gradient input (backward-data, NP = 2): power-of-two scale 2^(14 - e) from the tensor's absolute maximum = f * 2^e
  float in_mul = 1.f, in_div = 1.f;
  const bool in_scaled = NP == 2 && p.in_absmax != nullptr;
  if (in_scaled) {
    unsigned mb = p.in_absmax[(threadIdx.x & 63) * 16];                 // maximum of the 64 slots: one per lane, butterfly
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d));
    const int e = mb != 0u ? (int)((mb >> 23) & 0xffu) - 126 : 14;
    in_mul = __builtin_bit_cast(float, (unsigned)(14 - e + 127) << 23);
    in_div = __builtin_bit_cast(float, (unsigned)(e - 14 + 127) << 23);
  }
  // staging role: thread -> (pixel lane, 8-channel group); G groups per pixel, NTH / G pixels per step
  const int G = CK >> 3;
  const int cg = threadIdx.x & (G - 1), pl = threadIdx.x / G, PS = NTH / G;
  const int dr = PS / PC, dc = PS - dr * PC;
  const int nppix = PR * PC;

  unsigned long long c_stage = 0, c_mm = 0, t_0 = __builtin_readcyclecounter(), r_0 = wall_clock64();
  for (int ck0 = 0; ck0 < p.CIN; ck0 += CK) {
    unsigned long long t_a = __builtin_readcyclecounter();
    if (ck0 > 0) __syncthreads();                                        // the previous chunk's patch is no longer read
    {
      f32x4 sc0, sc1, sh0, sh1, rs0, rs1, rt0, rt1;
      const bool fin_i = (MODE == 1 || MODE == 2) && p.fin_in.stats != nullptr;
      const bool fin_r = MODE == 2 && p.fin_res.stats != nullptr;
      const bool res_ss = MODE == 2 && (p.res_scale != nullptr || fin_r);
      // Deferred GroupNorm finalisation (p.fin_in / p.fin_res): the sample's scale / shift tables were built at the top of the kernel,
      // in LDS behind the pixel tables
      float *ftab = reinterpret_cast<float *>(lds + tab_off + (size_t)p.MT * 32 * 8);   // [in scale | in shift | res scale | res shift][CIN]
      auto load_ss = [&]() {
        if (MODE >= 1) {
          const float *ps = fin_i ? ftab + ck0 + 8 * cg : p.in_scale + (long)n * p.CIN + ck0 + 8 * cg;
          const float *pt = fin_i ? ftab + p.CIN + ck0 + 8 * cg : p.in_shift + (long)n * p.CIN + ck0 + 8 * cg;
          sc0 = *reinterpret_cast<const f32x4 *>(ps);
          sc1 = *reinterpret_cast<const f32x4 *>(ps + 4);
          sh0 = *reinterpret_cast<const f32x4 *>(pt);
          sh1 = *reinterpret_cast<const f32x4 *>(pt + 4);
        }
        if (res_ss) {
          const float *ps = fin_r ? ftab + 2 * p.CIN + ck0 + 8 * cg : p.res_scale + (long)n * p.CIN + ck0 + 8 * cg;
          const float *pt = fin_r ? ftab + 3 * p.CIN + ck0 + 8 * cg : p.res_shift + (long)n * p.CIN + ck0 + 8 * cg;
          rs0 = *reinterpret_cast<const f32x4 *>(ps);
          rs1 = *reinterpret_cast<const f32x4 *>(ps + 4);
          rt0 = *reinterpret_cast<const f32x4 *>(pt);
          rt1 = *reinterpret_cast<const f32x4 *>(pt + 4);
        }
      };
      load_ss();
      int pr = pl / PC, pc = pl - pr * PC;
      const long img = ((long)n * p.H * p.W) * p.CIN + ck0 + 8 * cg;
      const float *xb = p.x + img;
      // two pixels per thread and round; the loads of the next round are in flight while this one is split and stored
      struct Round {
        f32x4 v[2][2], w[2][2];
        unsigned off[2];
        int gofs[2];
        bool ok[2], in[2], own[2];
      };
      auto fetch = [&](int pix, Round &r) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int hi = hi0 + pr * PSTEP, wi = wi0 + pc * PSTEP;
          r.ok[k] = pix + k * PS < nppix;
          r.in[k] = r.ok[k] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
          r.off[k] = (unsigned)((pr * PC + pc) * pitch + 16 * cg);
          r.v[k][0] = r.v[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (MODE >= 2) {
            // the tile owns the input pixels under its own outputs (stride 2: the 2 x 2 block of each): every pixel once
            r.own[k] = r.in[k] && p.xout != nullptr && blockIdx.y == 0 && pr >= PAD && pr < PAD + p.TR * CS && pc >= PAD && pc < PAD + p.TC * CS;
            r.gofs[k] = (hi * p.W + wi) * p.CIN;
          }
          if (r.in[k]) {
            const float *src = xb + ((long)hi * p.W + wi) * p.CIN;
            r.v[k][0] = *reinterpret_cast<const f32x4 *>(src);
            r.v[k][1] = *reinterpret_cast<const f32x4 *>(src + 4);
            if (MODE == 2) {
              const float *rsrc = p.res + img + ((long)hi * p.W + wi) * p.CIN;
              r.w[k][0] = *reinterpret_cast<const f32x4 *>(rsrc);
              r.w[k][1] = *reinterpret_cast<const f32x4 *>(rsrc + 4);
            }
          }
          pr += dr;
          pc += dc;
          if (pc >= PC) {
            pc -= PC;
            pr += 1;
          }
        }
      };
      auto split_store = [&](const Round &r) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (!r.ok[k]) continue;
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = r.v[k][e >> 2][e & 3];
            if (MODE == 1) {                                             // the producer's GroupNorm + ReLU; zero padding AFTER it
              const f32x4 &sc = e < 4 ? sc0 : sc1, &sh = e < 4 ? sh0 : sh1;
              t = r.in[k] ? fmaxf(__builtin_fmaf(t, sc[e & 3], sh[e & 3]), 0.f) : 0.f;
            }
            if (MODE == 3) {                                             // pooled stem keys (stem_mx.hip POOL): decode, |scale|, shift, ReLU
              const f32x4 &sc = e < 4 ? sc0 : sc1, &sh = e < 4 ? sh0 : sh1;
              int key = __builtin_bit_cast(int, t);
              key = key >= 0 ? key : key ^ 0x7fffffff;
              t = r.in[k] ? fmaxf(__builtin_fmaf(__builtin_bit_cast(float, key), __builtin_fabsf(sc[e & 3]), sh[e & 3]), 0.f) : 0.f;
            }
            if (MODE == 2) {                                             // block tail: the same operations as residual_kernel
              const f32x4 &sc = e < 4 ? sc0 : sc1, &sh = e < 4 ? sh0 : sh1;
              float u = r.w[k][e >> 2][e & 3];
              if (res_ss) u = __builtin_fmaf(u, (e < 4 ? rs0 : rs1)[e & 3], (e < 4 ? rt0 : rt1)[e & 3]);
              t = r.in[k] ? fmaxf(__builtin_fmaf(t, sc[e & 3], sh[e & 3]) + u, 0.f) : 0.f;
            }
            f[e] = t;
          }
          if (MODE >= 2 && r.own[k]) {
            float *dst = p.xout + img + r.gofs[k];
            *reinterpret_cast<f32x4 *>(dst) = f32x4{f[0], f[1], f[2], f[3]};
            *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{f[4], f[5], f[6], f[7]};
          }
          if (NP == 2) {                                                 // two float16 pieces of the eight channels
            if (in_scaled) {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= in_mul;
            }
            u32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = f[2 * e], b = f[2 * e + 1];
              const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
              o0[e] = __builtin_bit_cast(unsigned, h);
              o1[e] = pack2h(a - (float)h[0], b - (float)h[1]);
            }
            *reinterpret_cast<u32x4 *>(lds + r.off[k]) = o0;
            *reinterpret_cast<u32x4 *>(lds + plane + r.off[k]) = o1;
            continue;
          }
          u32x4 o0, o1, o2;                                              // the three bf16 pieces of the eight channels
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = f[2 * e], b = f[2 * e + 1];
            const unsigned h = pack2(a, b);
            const float ra = a - lo_f(h), rb = b - hi_f(h);
            const unsigned m = pack2(ra, rb);
            o0[e] = h;
            o1[e] = m;
            o2[e] = pack2(ra - lo_f(m), rb - hi_f(m));
          }
          *reinterpret_cast<u32x4 *>(lds + r.off[k]) = o0;
          *reinterpret_cast<u32x4 *>(lds + plane + r.off[k]) = o1;
          *reinterpret_cast<u32x4 *>(lds + 2 * plane + r.off[k]) = o2;
        }
      };
      constexpr bool PIPE = !((MODE == 2 && (MW * NW >= 6 || x3_wpe(MW, NW, NP) == 3 || DSF)) || (DSF && MW * NW == 4));   // (the block-tail stager of the 96-accumulator tile /
                                                                                         //  of the 168-register small tiles would spill)
      if (MW * NW <= 2 && !(MODE == 2 && (x3_wpe(MW, NW, NP) == 3 || DSF))) {
        // few accumulators: registers for three rounds in flight — the whole patch of the 32- / 64-channel stages is ONE memory
        // latency instead of three (their staging phase is as long as their MFMA phase)
        for (int pix = pl; pix < nppix; pix += 6 * PS) {
          Round r0, r1, r2;
          fetch(pix, r0);
          fetch(pix + 2 * PS, r1);
          fetch(pix + 4 * PS, r2);
          split_store(r0);
          split_store(r1);
          split_store(r2);
        }
      } else if (PIPE) {
        Round cur, nxt;
        fetch(pl, cur);
        for (int pix = pl; pix < nppix; pix += 2 * PS) {
          const bool more = pix + 2 * PS < nppix;
          if (more) fetch(pix + 2 * PS, nxt);
          split_store(cur);
          if (more) cur = nxt;
        }
      } else {
        for (int pix = pl; pix < nppix; pix += 2 * PS) {
          Round cur;
          fetch(pix, cur);
          split_store(cur);
        }
      }
    }
    __syncthreads();
    unsigned long long t_b = __builtin_readcyclecounter();
    c_stage += t_b - t_a;

    // ---- compute: steps s = (tap, 16-channel chunk).  B fragments (three weight pieces per N-tile) stream from L2 one
    // step ahead (two register sets); A fragments (three planes per M-tile) come from LDS at the start of the step.
    unsigned aoff[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = min(KSW ? i : wave_m * MW + i, p.MT - 1);
      aoff[i] = qtab[mt * 32 + (lane & 31)] + (unsigned)((lane >> 5) * 16);
    }
    const int kcc = CK >> 4;                                             // k-chunks per staged chunk
    const int nsteps = KS * KS * kcc;
    // The step being fetched (one ahead of the one being multiplied), kept as running offsets: a division per step costs the
    // wave ~50 scalar instructions between two MFMA groups, and with one wave per SIMD (6 x 11 maps) nobody fills that gap.
    const unsigned kstep = (unsigned)ntt * (NP * 1024u);                 // bytes of one k-chunk of B (all N-tiles, NP pieces)
    const char *wb_n = reinterpret_cast<const char *>(g_wpk) + (long)(ck0 >> 4) * kstep;
    unsigned toff_n = 0;                                                 // patch byte offset of the step's (tap, k-chunk)
    int kc_n = 0, kw_n = 0;
    auto advance = [&]() {
      ++kc_n;
      toff_n += 32;
      wb_n += kstep;
      if (kc_n == kcc) {
        kc_n = 0;
        toff_n += (unsigned)(pitch - kcc * 32);
        wb_n += (long)(kct - kcc) * kstep;
        if (++kw_n == KS) {
          kw_n = 0;
          toff_n += (unsigned)((PC - KS) * pitch);
        }
      }
    };
    // RING (float16 pieces, no riding downsample conv): B fragments are fetched TWO steps ahead into a ring of three register sets —
    // one step (6-18 MFMAs, 200-600 cycles) is shorter than an L2 round trip under load, and the ISA showed every step waiting for the
    // fragments it had asked for one step earlier.  A (LDS) stays one step ahead.  The two fetches run on trackers of their own.
    // Small wave tiles only (one or two accumulator tiles: the fine plan's launches, the 32- / 64-channel stages): on the larger tiles
    // the ring was neutral at 256 pairs (2.275 against 2.270 ms) and cost the 128-pair training step 0.4 % (15-28 registers per variant).
    constexpr bool RING = NP == 2 && KS == 3 && MW * NW <= 2;
    const char *wb_b = wb_n;                                             // RING: the B step being fetched
    int kc_b = 0;
    auto advanceB = [&]() {
      wb_b += kstep;
      if (++kc_b == kcc) {
        kc_b = 0;
        wb_b += (long)(kct - kcc) * kstep;
      }
    };
    unsigned voff[NW];                                                   // lane's byte offset inside a k-chunk of B
#pragma unroll
    for (int j = 0; j < NW; ++j) voff[j] = (unsigned)min(wave_n * NW + j, ntt - 1) * (NP * 1024u) + (unsigned)lane * 16u;   // (N-tiles past the layer's repeat the last one)
    auto loadBr = [&](u32x4 (*b)[NW]) {
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) b[pc][j] = *reinterpret_cast<const u32x4 *>(wb_b + (size_t)voff[j] + pc * 1024);
    };
    // A fragments of M-tile i (three planes) / B fragments (three weight pieces per N-tile) of the step being fetched
    auto loadA = [&](int i, u32x4 (*a)[MW]) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) a[pc][i] = *reinterpret_cast<const u32x4 *>(lds + pc * plane + aoff[i] + toff_n);
    };
    auto loadB = [&](u32x4 (*b)[NW]) {
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) b[pc][j] = *reinterpret_cast<const u32x4 *>(wb_n + (size_t)voff[j] + pc * 1024);
    };
    // One step: M-tile by M-tile; as soon as an M-tile's MFMAs are issued its A registers take the NEXT step's fragments, so
    // the LDS latency hides behind the other M-tiles' MFMAs
    auto step = [&](u32x4 (*a)[MW], const u32x4 (*b)[NW]) {
#pragma unroll
      for (int i = 0; i < MW; ++i) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          if (NP == 2) {                                                 // a1 w0, a0 w1, a0 w0 (float16 pieces)
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[TA[t]][i]),
                                                                 __builtin_bit_cast(f16x8, b[TB[t]][j]), acc[i][j], 0, 0, 0);
          } else {
            // smallest terms first: a1 w1, a2 w0, a0 w2, a1 w0, a0 w1, a0 w0
            constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[TA[t]][i]),
                                                                  __builtin_bit_cast(bf16x8, b[TB[t]][j]), acc[i][j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        loadA(i, a);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    u32x4 a[NP][MW], b0[NP][NW], b1[NP][NW];
    // DSF: the riding downsample conv multiplies BEHIND the nine taps of the chunk (ds_tail below): the centre tap's A fragments are read
    // from LDS once more — kcc steps — so the main K loop is the plain one (and takes the B ring).  Its B fragments: a 1x1 conv's
    // operand, [k-chunk][N-tile][piece][lane][8]; those of the chunk's first k-chunk are fetched before the main loop.
    u32x4 bd0[DSF ? NP : 1][NW], bd1[DSF ? NP : 1][NW];
    auto loadBd = [&](u32x4 (*b)[NW], int kc) {
      const char *wd = reinterpret_cast<const char *>(g_ds_wpk) + (long)((ck0 >> 4) + kc) * kstep;
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) b[pc][j] = *reinterpret_cast<const u32x4 *>(wd + (size_t)voff[j] + pc * 1024);
    };
    if constexpr (KSW) {
      // this wave's quarter of the chunk's steps, B through a ring of three register sets (two steps ahead), A one step ahead
      const int s0 = (wave * nsteps) >> 2, s1 = ((wave + 1) * nsteps) >> 2;
      {
        const int tap = s0 / kcc, kc = s0 - tap * kcc, kh = tap / KS, kw = tap - kh * KS;
        kc_n = kc;
        kw_n = kw;
        toff_n = (unsigned)((kh * PC + kw) * pitch + kc * 32);
        wb_n += ((long)tap * kct + kc) * kstep;
        kc_b = kc;
        wb_b = wb_n;
      }
      u32x4 br[3][NP][NW];
      int fb = s0;                                                       // the next step whose B fragments are fetched
      auto fetchB = [&](u32x4 (*bq)[NW]) {
        loadBr(bq);
        if (++fb < s1) advanceB();                                       // (past the last step: harmless re-fetches of it)
      };
      fetchB(br[0]);
      fetchB(br[1]);
#pragma unroll
      for (int i = 0; i < MW; ++i) loadA(i, a);
      if (s0 + 1 < s1) advance();
#pragma unroll 1
      for (int s = s0; s < s1; s += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (s + u >= s1) break;
          fetchB(br[(u + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
          step(a, br[u]);                                                // multiplies step s + u, fetches A of step s + u + 1
          if (s + u + 2 < s1) advance();
        }
      }
    } else {
    if (DSF) loadBd(bd0, 0);
    loadB(b0);
#pragma unroll
    for (int i = 0; i < MW; ++i) loadA(i, a);
    advance();                                                           // -> step 1
    if (RING) {
      // nsteps = 9 taps x kcc (kcc = 2, 4, 8) is a multiple of 3 and of 6.  Ring of R register sets, B fetched R - 1 steps ahead: small
      // wave tiles have short steps (a (1,1) tile: three MFMAs = 96 cycles against an L2 round trip of several hundred — the fine plan's
      // launches and the compression conv ran at a third of the pipe rate waiting for fragments), so they look further ahead.
      // Entering step t: toff_n = A offset of step t + 1 (advance() moves it; its B pointer is not used here), wb_b = B pointer of step
      // t + R - 1.  Past the last step the trackers stay put (harmless re-fetches of the last step).
      constexpr int R = MW * NW <= 2 ? 6 : 3;
      u32x4 br[R][NP][NW];
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
#pragma unroll
        for (int j = 0; j < NW; ++j) br[0][pc][j] = b0[pc][j];           // step 0 came through loadB (wb_n == wb_b there)
#pragma unroll
      for (int q = 1; q < R - 1; ++q) {
        advanceB();
        loadBr(br[q]);                                                   // steps 1 .. R - 2
      }
      advanceB();                                                        // -> step R - 1
      // Small wave tiles also fetch A (LDS) TWO steps ahead, into two register sets used by the even / odd steps: with one M-tile a step
      // is three MFMAs = 96 cycles, less than an LDS read under load (measured: 63 cycles per MFMA with A one step ahead).
      constexpr bool A2 = MW * NW <= 2 && !(MODE == 2 && MW * NW == 2 && !DSF) && !(MODE == 2 && DSF && MW * NW == 1);   // (168-register tiles: three waves per SIMD)
      u32x4 a1[A2 ? NP : 1][MW];
      if (A2) {
#pragma unroll
        for (int i = 0; i < MW; ++i) loadA(i, a1);                        // step 1 (toff_n is there); `a` holds step 0
        advance();                                                       // -> step 2
      }
#pragma unroll 1
      for (int s = 0; s < nsteps; s += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          loadBr(br[(u + R - 1) % R]);                                   // step s + u + R - 1 into the set step s + u - 1 just left
          if (s + u + R < nsteps) advanceB();
          __builtin_amdgcn_sched_barrier(0);
          if (A2) {
            step((u & 1) ? a1 : a, br[u]);                               // multiplies step s + u, refills ITS A set with step s + u + 2
            if (s + u + 3 < nsteps) advance();
          } else {
            step(a, br[u]);                                              // multiplies step s + u, fetches A of step s + u + 1
            if (s + u + 2 < nsteps) advance();
          }
        }
      }
    } else {
#pragma unroll 1
      for (int s = 0; s < nsteps; s += 2) {                              // nsteps is even for every supported shape
        const bool more = s + 2 < nsteps;                                // (the last iteration re-fetches step s + 1: unused)
        loadB(b1);
        __builtin_amdgcn_sched_barrier(0);
        step(a, b0);                                                     // multiplies step s, fetches A of step s + 1
        if (more) advance();
        loadB(b0);
        __builtin_amdgcn_sched_barrier(0);
        step(a, b1);                                                     // multiplies step s + 1, fetches A of step s + 2
        if (more) advance();
      }
    }
    }
    if constexpr (DSF) {
      // ds_tail: the downsample conv = this conv's centre tap (patch offset (PC + 1) pixels) on its own weights, k-chunks ascending,
      // terms a1 w0, a0 w1, a0 w0 — the order of the separate 1x1 launch (bit-identical raw output)
      const unsigned tc0 = (unsigned)((PC + 1) * pitch);
      u32x4 ad[NP][MW];
      auto loadAd = [&](int kc) {
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int pc = 0; pc < NP; ++pc) ad[pc][i] = *reinterpret_cast<const u32x4 *>(lds + pc * plane + aoff[i] + tc0 + (unsigned)kc * 32u);
      };
      auto mm = [&](const u32x4 (*bd)[NW]) {
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
              accd[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ad[TA[t]][i]), __builtin_bit_cast(f16x8, bd[TB[t]][j]),
                                                                  accd[i][j], 0, 0, 0);
      };
      if constexpr (MW * NW == 4) {                                        // (one fragment set: 16 registers this tile does not have twice)
#pragma unroll 1
        for (int kc = 0; kc < kcc; ++kc) {
          loadAd(kc);
          mm(bd0);
          if (kc + 1 < kcc) loadBd(bd0, kc + 1);
        }
      } else {
#pragma unroll 1
        for (int kc = 0; kc < kcc; kc += 2) {                              // (kcc is even)
          loadAd(kc);
          loadBd(bd1, kc + 1);
          mm(bd0);
          loadAd(kc + 1);
          if (kc + 2 < kcc) loadBd(bd0, kc + 2);
          mm(bd1);
        }
      }
    }
    c_mm += __builtin_readcyclecounter() - t_b;
  }
  if constexpr (KSW) {
    // the four waves' partial sums of the tile meet in LDS ([wave][M-tile][register][lane]: conflict-free), added in wave order;
    // wave w goes on with M-tile w in acc[0]
    __syncthreads();                                                     // the patch is no longer read
    float *part = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[((wave * MW + i) * 16 + r) * 64 + lane] = acc[i][0][r];
    __syncthreads();
    const int mt = min(wave, MW - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = part[((0 * MW + mt) * 16 + r) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += part[((w * MW + mt) * 16 + r) * 64 + lane];
      acc[0][0][r] = v;
    }
    __syncthreads();                                                     // (the statistics scratch of the epilogue reuses this memory)
  }
  unsigned long long t_e = __builtin_readcyclecounter();

  // ---- epilogue: raw output + per-(sample, tile, channel) GroupNorm partial sums (one writer per slot).  Runs once for the conv and
  // (DSF) once more for the downsample conv that rode on it: accumulators, output tensor, statistics and GroupNorm outputs of each.
  const int rr16 = lane >> 5;
  const long ybase = (((long)n * p.Ho + r0) * p.Wo + c0) * p.COUTP;
  auto emit = [&](f32x16 (*ac)[NW], float *yout, float *stats, const float *oscale_ptr, float oscale, const float *gamma, const float *beta,
                  float *gscale, float *gshift, float *gmu, float *grstd, bool again) {
    if (NP == 2) {                                                         // undo the weights' power-of-two scale (exact)
      const float os = (oscale_ptr != nullptr ? *oscale_ptr : oscale) * in_div;
#pragma unroll
      for (int i = 0; i < EMW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) ac[i][j][r] *= os;
    }
    float t1[NW], t2[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) t1[j] = t2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < EMW; ++i) {
      const int mt = KSW ? wave : wave_m * MW + i;
      if (mt >= p.MT) continue;
      u32x4 ent[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) ent[g4] = *reinterpret_cast<const u32x4 *>(otab + mt * 32 + 8 * g4 + 4 * rr16);
      unsigned flags = 0;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) flags |= ent[g4][0] | ent[g4][1] | ent[g4][2] | ent[g4][3];
      const bool whole = !__any((int)(flags >> 31));                       // wave-uniform
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int nt = wave_n * NW + j;
        if (nt >= ntt) continue;
        const int co = nt * 32 + (lane & 31);
        float s1 = 0.f, s2 = 0.f;
        if (whole) {                                                       // every pixel of the M-tile exists: plain stores
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = ac[i][j][r];
            (yout + ybase + co)[ent[r >> 2][r & 3]] = v;
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned e = ent[r >> 2][r & 3];
            const bool ok = (int)e >= 0;
            const float v = ok ? ac[i][j][r] : 0.f;
            if (ok) (yout + ybase + co)[e] = v;
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);
          }
        }
        t1[j] += s1 + __shfl_xor(s1, 32);                                  // M-tiles of this wave, in order
        t2[j] += s2 + __shfl_xor(s2, 32);
      }
    }
    if (stats != nullptr) {                                                // one slot per tile: the waves along M meet in LDS
      const int rows = NWV / wn;
      float *red = reinterpret_cast<float *>(lds);                        // [wave][NW][32 channels][2]
      if (rows > 1) {
        __syncthreads();                                                   // the patch (the first pass's scratch) is no longer read
        if (lane < 32)
#pragma unroll
          for (int j = 0; j < NW; ++j) *reinterpret_cast<f32x2 *>(red + ((wave * NW + j) * 32 + lane) * 2) = f32x2{t1[j], t2[j]};
        __syncthreads();
      }
      if (wave_m == 0 && lane < 32) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int nt = wave_n * NW + j;
          if (nt >= ntt) continue;
          float s1 = t1[j], s2 = t2[j];
          for (int w = 1; w < rows; ++w) {                                 // fixed order: bit-reproducible
            const f32x2 o = *reinterpret_cast<const f32x2 *>(red + (((w * wn + (wave & (wn - 1))) * NW + j) * 32 + lane) * 2);
            s1 += o[0];
            s2 += o[1];
          }
          float *dst = stats + (((long)n * p.slots + tri * p.tiles_c + tci) * p.COUTP + nt * 32 + lane) * 2;
          if (p.gn_ctr != nullptr) {                                       // read by the sample's last workgroup inside this launch
            gn_stats_store(dst, s1, s2);
            continue;
          }
          dst[0] = s1;
          dst[1] = s2;
          if (gscale != nullptr) {                                         // the sample's only slot: finalise here (no launch)
            const int c = nt * 32 + lane;
            const bool first = gmu != nullptr && c % p.gn_cpg == 0;
            const long gi = (long)n * (p.COUTP / p.gn_cpg) + c / p.gn_cpg;
            gn_finalize_lane(s1, s2, p.gn_cpg, p.gn_P, p.gn_eps, gamma[c], beta[c], gscale + (long)n * p.COUTP + c,
                             gshift + (long)n * p.COUTP + c, first ? gmu + gi : nullptr, first ? grstd + gi : nullptr);
          }
        }
      }
    }
    (void)again;
  };
  emit(acc, p.y, p.stats, p.oscale_ptr, g_oscale, g_gamma, g_beta, p.gn_scale, p.gn_shift, p.gn_mu, p.gn_rstd, false);
  if constexpr (DSF) emit(accd, p.ds_y, p.ds_stats, p.ds_oscale_ptr, g_ds_oscale, g_ds_gamma, g_ds_beta, p.ds_scale, p.ds_shift, p.ds_mu, p.ds_rstd, true);
  if (p.stats != nullptr && p.gn_ctr != nullptr) {                        // several tiles per sample: the last one to arrive finalises
    if (gn_last_arrival(p.gn_ctr + (long)n * gridDim.y + blockIdx.y, (unsigned)p.slots, reinterpret_cast<int *>(lds + 4096))) {
      const int nt0 = (int)blockIdx.y * wn * NW, nt1 = min(nt0 + wn * NW, ntt);
      const int NG = p.COUTP / p.gn_cpg;
      for (int g = (nt0 * 32) / p.gn_cpg + wave; g < (nt1 * 32) / p.gn_cpg; g += NWV) {
        gn_finalize_group_wave(p.stats + (long)n * p.slots * p.COUTP * 2, p.slots, p.COUTP, g, p.gn_cpg, p.gn_P, p.gn_eps, g_gamma, g_beta,
                               p.gn_scale + (long)n * p.COUTP, p.gn_shift + (long)n * p.COUTP, p.gn_mu ? p.gn_mu + (long)n * NG + g : nullptr,
                               p.gn_mu ? p.gn_rstd + (long)n * NG + g : nullptr);
        if (DSF)
          gn_finalize_group_wave(p.ds_stats + (long)n * p.slots * p.COUTP * 2, p.slots, p.COUTP, g, p.gn_cpg, p.gn_P, p.gn_eps, g_ds_gamma,
                                 g_ds_beta, p.ds_scale + (long)n * p.COUTP, p.ds_shift + (long)n * p.COUTP, p.ds_mu ? p.ds_mu + (long)n * NG + g : nullptr,
                                 p.ds_mu ? p.ds_rstd + (long)n * NG + g : nullptr);
      }
    }
  }
  if (p.prof && lane == 0 && blockIdx.x == 13 && blockIdx.y == 0) {
    const unsigned long long t_f = __builtin_readcyclecounter();
    unsigned long long *d = p.prof + wave * 8;
    d[0] = c_stage;
    d[1] = c_mm;
    d[2] = t_f - t_e;
    d[3] = t_f - t_0;
    d[4] = wall_clock64() - r_0;
  }
}


// =====================================================================================================================
// conv_x3p_kernel — the PERSISTENT form of the float16-piece 3x3 stride-1 conv for the shallow stages (32 / 64 input channels:
// the whole K dimension is one staged chunk, wave tiles (1,1) / (2,1)), input modes 0, 1 and 3.
//
// conv_x3_kernel's workgroup loads its patch, converts it, multiplies, stores, and exits; the next workgroup starts with a cold
// memory round trip.  On the 48x86 and 24x43 maps the layers are HBM-bound in float32 activations (300 MB per conv at 256 pairs)
// and ran at ~2.4 TB/s because every workgroup's loads come as one burst followed by ~20 k cycles without any.  Here a workgroup
// walks tiles, and the NEXT tile's patch (and its sample's GroupNorm scale / shift) is fetched into registers right behind the
// barrier that publishes the current patch: those loads are in flight during the whole K loop and epilogue of the current tile.
// Same MFMA order, same statistics order as conv_x3_kernel: bit-identical outputs (tests/test_gpu_knobs.py).
// BRES (32 -> 32 channels: layer1): the layer's whole B operand — 9 taps x 2 k-chunks x 2 pieces = 36 fragments, 144 registers — is
// loaded ONCE per workgroup and stays in registers for every tile it walks.  Round 4 measured what bounds these layers: not the
// patch loads (prefetching them changed nothing) but the CU's vector-memory pipe, ~20 B/clk, of which the weight fragments were two
// thirds (4 waves x 36 KB per tile against 41 KB of patch + 32 KB of output).  Two waves per SIMD (256 registers), no patch prefetch.
template <int MODE, int MW, int NW, bool BRES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BRES ? 2 : 3, BRES ? 2 : 3))) void conv_x3p_kernel(const ConvX3Args p) {
  static_assert(MODE == 0 || MODE == 1 || MODE == 3, "the block-tail stager has no registers for a second patch");
  static_assert(!BRES || NW == 1, "resident weights: one N-tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int KS = 3, PAD = 1, NP = 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = p.B * p.tiles_r * p.tiles_c;
  const int chunk = (ntiles + 7) >> 3;
  const int PR = p.PR, PC = p.PC, CK = p.CK;
  const int pitch = CK * 2 + 16;
  const int plane = PR * PC * pitch;
  const int npix = p.TR * p.TC;
  unsigned *qtab = reinterpret_cast<unsigned *>(lds + NP * plane);
  unsigned *otab = qtab + p.MT * 32;
  const int wn = p.wn;
  const int wave_m = wave / wn;
  const int wave_n = (wave & (wn - 1)) + (int)blockIdx.y * wn;
  const int ntt = p.COUTP >> 5;
  float in_mul = 1.f, in_div = 1.f;
  const bool in_scaled = p.in_absmax != nullptr;
  if (in_scaled) {
    unsigned mb = p.in_absmax[(threadIdx.x & 63) * 16];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d));
    const int e = mb != 0u ? (int)((mb >> 23) & 0xffu) - 126 : 14;
    in_mul = __builtin_bit_cast(float, (unsigned)(14 - e + 127) << 23);
    in_div = __builtin_bit_cast(float, (unsigned)(e - 14 + 127) << 23);
  }
  const int G = CK >> 3;
  const int cg = threadIdx.x & (G - 1), pl = threadIdx.x / G, PS = 256 / G;
  const int nppix = PR * PC;
  // this thread's (at most six) patch pixels: tile-independent
  constexpr int NPX = 6;
  int ppr[NPX], ppc[NPX];
  unsigned poff[NPX];
  bool pok[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int pix = pl + k * PS;
    pok[k] = pix < nppix;
    const int q = pok[k] ? pix : 0;
    ppr[k] = q / PC;
    ppc[k] = q - ppr[k] * PC;
    poff[k] = (unsigned)((ppr[k] * PC + ppc[k]) * pitch + 16 * cg);
  }
  if ((int)threadIdx.x < p.MT * 32) {
    const int q = min((int)threadIdx.x, npix - 1);
    const int tr = q / p.TC, tc = q - tr * p.TC;
    qtab[threadIdx.x] = (unsigned)((tr * PC + tc) * pitch);
  }
  struct Tile {
    int n, tri, tci, r0, c0, hi0, wi0;
    bool valid;
  };
  auto decode = [&](int vb) {
    Tile t;
    int bid = (vb & 7) * chunk + (vb >> 3);
    t.valid = vb < 8 * chunk && bid < ntiles;
    if (!t.valid) bid = 0;
    t.tci = bid % p.tiles_c;
    bid /= p.tiles_c;
    t.tri = bid % p.tiles_r;
    t.n = bid / p.tiles_r;
    t.r0 = t.tri * p.TR;
    t.c0 = t.tci * p.TC;
    t.hi0 = t.r0 - PAD;
    t.wi0 = t.c0 - PAD;
    return t;
  };
  // ---- the patch of one tile in registers
  f32x4 v[NPX][2], sc0, sc1, sh0, sh1;
  bool inb[NPX];
  auto fetch = [&](const Tile &t) {
    const long img = ((long)t.n * p.H * p.W) * p.CIN + 8 * cg;
    if (MODE >= 1) {
      const float *ps = p.in_scale + (long)t.n * p.CIN + 8 * cg;
      const float *pt = p.in_shift + (long)t.n * p.CIN + 8 * cg;
      sc0 = *reinterpret_cast<const f32x4 *>(ps);
      sc1 = *reinterpret_cast<const f32x4 *>(ps + 4);
      sh0 = *reinterpret_cast<const f32x4 *>(pt);
      sh1 = *reinterpret_cast<const f32x4 *>(pt + 4);
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int hi = t.hi0 + ppr[k], wi = t.wi0 + ppc[k];
      inb[k] = pok[k] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
      v[k][0] = v[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (inb[k]) {
        const float *src = p.x + img + ((long)hi * p.W + wi) * p.CIN;
        v[k][0] = *reinterpret_cast<const f32x4 *>(src);
        v[k][1] = *reinterpret_cast<const f32x4 *>(src + 4);
      }
    }
  };
  auto split_store = [&](const Tile &t) {
    const long img = ((long)t.n * p.H * p.W) * p.CIN + 8 * cg;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      if (!pok[k]) continue;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[k][e >> 2][e & 3];
        if (MODE == 1) {
          const f32x4 &sc = e < 4 ? sc0 : sc1, &sh = e < 4 ? sh0 : sh1;
          x = inb[k] ? fmaxf(__builtin_fmaf(x, sc[e & 3], sh[e & 3]), 0.f) : 0.f;
        }
        if (MODE == 3) {
          const f32x4 &sc = e < 4 ? sc0 : sc1, &sh = e < 4 ? sh0 : sh1;
          int key = __builtin_bit_cast(int, x);
          key = key >= 0 ? key : key ^ 0x7fffffff;
          x = inb[k] ? fmaxf(__builtin_fmaf(__builtin_bit_cast(float, key), __builtin_fabsf(sc[e & 3]), sh[e & 3]), 0.f) : 0.f;
        }
        f[e] = x;
      }
      if (MODE == 3) {                                                   // the tile owns the input pixels under its own outputs
        const bool own = inb[k] && blockIdx.y == 0 && ppr[k] >= PAD && ppr[k] < PAD + p.TR && ppc[k] >= PAD && ppc[k] < PAD + p.TC;
        if (own) {
          float *dst = p.xout + img + ((long)(t.hi0 + ppr[k]) * p.W + t.wi0 + ppc[k]) * p.CIN;
          *reinterpret_cast<f32x4 *>(dst) = f32x4{f[0], f[1], f[2], f[3]};
          *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{f[4], f[5], f[6], f[7]};
        }
      }
      if (in_scaled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= in_mul;
      }
      u32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = f[2 * e], b = f[2 * e + 1];
        const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
        o0[e] = __builtin_bit_cast(unsigned, h);
        o1[e] = pack2h(a - (float)h[0], b - (float)h[1]);
      }
      *reinterpret_cast<u32x4 *>(lds + poff[k]) = o0;
      *reinterpret_cast<u32x4 *>(lds + plane + poff[k]) = o1;
    }
  };

  Tile cur = decode((int)blockIdx.x);
  if (!cur.valid) return;
  constexpr int RB = BRES ? 18 : 1;                                       // resident B: [step = tap * 2 + k-chunk][piece]
  u32x4 bres[RB][NP];
  if (BRES) {
#pragma unroll
    for (int st = 0; st < RB; ++st)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
        bres[st][pc] = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.wpk) + (size_t)st * (NP * 1024u) + pc * 1024 + lane * 16);
  }
  fetch(cur);
  for (int vb = (int)blockIdx.x;; vb += (int)gridDim.x) {
    if ((int)threadIdx.x < p.MT * 32) {
      const int q = min((int)threadIdx.x, npix - 1);
      const int tr = q / p.TC, tc = q - tr * p.TC;
      const bool ok = (int)threadIdx.x < npix && cur.r0 + tr < p.Ho && cur.c0 + tc < p.Wo;
      otab[threadIdx.x] = (unsigned)((tr * p.Wo + tc) * p.COUTP) | (ok ? 0u : 0x80000000u);
    }
    split_store(cur);
    __syncthreads();
    const Tile nxt = decode(vb + (int)gridDim.x);
    if (!BRES && nxt.valid) fetch(nxt);                                  // in flight during this tile's K loop and epilogue

    // ---- compute (conv_x3_kernel's K loop, one staged chunk)
    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    unsigned aoff[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = min(wave_m * MW + i, p.MT - 1);
      aoff[i] = qtab[mt * 32 + (lane & 31)] + (unsigned)((lane >> 5) * 16);
    }
    const int kcc = CK >> 4;
    const int nsteps = KS * KS * kcc;
    const unsigned kstep = (unsigned)ntt * (NP * 1024u);
    const char *wb_n = reinterpret_cast<const char *>(p.wpk);
    unsigned toff_n = 0;
    int kc_n = 0, kw_n = 0;
    auto advance = [&]() {
      ++kc_n;
      toff_n += 32;
      wb_n += kstep;
      if (kc_n == kcc) {
        kc_n = 0;
        toff_n += (unsigned)(pitch - kcc * 32);
        if (++kw_n == KS) {
          kw_n = 0;
          toff_n += (unsigned)((PC - KS) * pitch);
        }
      }
    };
    unsigned voff[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) voff[j] = (unsigned)min(wave_n * NW + j, ntt - 1) * (NP * 1024u) + (unsigned)lane * 16u;
    auto loadA = [&](int i, u32x4 (*a)[MW]) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) a[pc][i] = *reinterpret_cast<const u32x4 *>(lds + pc * plane + aoff[i] + toff_n);
    };
    auto loadB = [&](u32x4 (*b)[NW]) {
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) b[pc][j] = *reinterpret_cast<const u32x4 *>(wb_n + (size_t)voff[j] + pc * 1024);
    };
    auto step = [&](u32x4 (*a)[MW], const u32x4 (*b)[NW]) {
#pragma unroll
      for (int i = 0; i < MW; ++i) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
          for (int t = 0; t < 3; ++t)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[TA[t]][i]),
                                                               __builtin_bit_cast(f16x8, b[TB[t]][j]), acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        loadA(i, a);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    u32x4 a[NP][MW], b0[NP][NW], b1[NP][NW];
    if (BRES) {
      // steps in conv_x3_kernel's order (tap-major, k-chunk inner); A offsets are compile-time: tap (kh, kw), chunk kc
#pragma unroll
      for (int i = 0; i < MW; ++i) loadA(i, a);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int nx = st + 1, ntap = nx >> 1, nkc = nx & 1;
        toff_n = (unsigned)(((ntap / 3) * PC + (ntap % 3)) * pitch + nkc * 32);   // the NEXT step's A fragments (past the end: unused)
        if (st == 17) toff_n = 0;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
          for (int t = 0; t < 3; ++t)
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[TA[t]][i]),
                                                               __builtin_bit_cast(f16x8, bres[BRES ? st : 0][TB[t]]), acc[i][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          loadA(i, a);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      loadB(b0);
#pragma unroll
      for (int i = 0; i < MW; ++i) loadA(i, a);
      advance();
#pragma unroll 1
      for (int s = 0; s < nsteps; s += 2) {
        const bool more = s + 2 < nsteps;
        loadB(b1);
        __builtin_amdgcn_sched_barrier(0);
        step(a, b0);
        if (more) advance();
        loadB(b0);
        __builtin_amdgcn_sched_barrier(0);
        step(a, b1);
        if (more) advance();
      }
    }

    // ---- epilogue (conv_x3_kernel's)
    {
      const float os = (p.oscale_ptr != nullptr ? *p.oscale_ptr : p.oscale) * in_div;
#pragma unroll
      for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= os;
    }
    const int rr16 = lane >> 5;
    const long ybase = (((long)cur.n * p.Ho + cur.r0) * p.Wo + cur.c0) * p.COUTP;
    float t1[NW], t2[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) t1[j] = t2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = wave_m * MW + i;
      if (mt >= p.MT) continue;
      u32x4 ent[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) ent[g4] = *reinterpret_cast<const u32x4 *>(otab + mt * 32 + 8 * g4 + 4 * rr16);
      unsigned flags = 0;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) flags |= ent[g4][0] | ent[g4][1] | ent[g4][2] | ent[g4][3];
      const bool whole = !__any((int)(flags >> 31));
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int nt = wave_n * NW + j;
        if (nt >= ntt) continue;
        const int co = nt * 32 + (lane & 31);
        float s1 = 0.f, s2 = 0.f;
        if (whole) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = acc[i][j][r];
            (p.y + ybase + co)[ent[r >> 2][r & 3]] = x;
            s1 += x;
            s2 = __builtin_fmaf(x, x, s2);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned e = ent[r >> 2][r & 3];
            const bool ok = (int)e >= 0;
            const float x = ok ? acc[i][j][r] : 0.f;
            if (ok) (p.y + ybase + co)[e] = x;
            s1 += x;
            s2 = __builtin_fmaf(x, x, s2);
          }
        }
        t1[j] += s1 + __shfl_xor(s1, 32);
        t2[j] += s2 + __shfl_xor(s2, 32);
      }
    }
    if (p.stats != nullptr) {
      const int rows = 4 / wn;
      float *red = reinterpret_cast<float *>(lds);
      if (rows > 1) {
        __syncthreads();
        if (lane < 32)
#pragma unroll
          for (int j = 0; j < NW; ++j) *reinterpret_cast<f32x2 *>(red + ((wave * NW + j) * 32 + lane) * 2) = f32x2{t1[j], t2[j]};
        __syncthreads();
      }
      if (wave_m == 0 && lane < 32) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int nt = wave_n * NW + j;
          if (nt >= ntt) continue;
          float s1 = t1[j], s2 = t2[j];
          for (int w = 1; w < rows; ++w) {
            const f32x2 o = *reinterpret_cast<const f32x2 *>(red + (((w * wn + (wave & (wn - 1))) * NW + j) * 32 + lane) * 2);
            s1 += o[0];
            s2 += o[1];
          }
          float *dst = p.stats + (((long)cur.n * p.slots + cur.tri * p.tiles_c + cur.tci) * p.COUTP + nt * 32 + lane) * 2;
          dst[0] = s1;
          dst[1] = s2;
        }
      }
    }
    if (!nxt.valid) break;
    if (BRES) fetch(nxt);
    __syncthreads();                                                     // planes, tables and the reduction scratch are free again
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {
template <int KS, int STRIDE, int NP>
hipError_t launch_ks(const ConvX3Args &a, int mode, int mw, int nw, dim3 grid, size_t ldsb, hipStream_t s) {
#define PNVO_X3(MODE_, MW_, NW_)                                                                          \
  if (mode == MODE_ && mw == MW_ && nw == NW_) {                                                          \
    hipLaunchKernelGGL((conv_x3_kernel<KS, STRIDE, MODE_, MW_, NW_, NP>), grid, dim3(256), ldsb, s, a);   \
    return hipGetLastError();                                                                             \
  }
  if constexpr (KS == 3 && STRIDE == 2 && NP == 2) {
    if (a.ds_wpk != nullptr) {                                           // the block's downsample conv rides on this launch (DSF)
#define PNVO_X3D(MODE_, MW_, NW_)                                                                               \
  if (mode == MODE_ && mw == MW_ && nw == NW_) {                                                                \
    hipLaunchKernelGGL((conv_x3_kernel<KS, STRIDE, MODE_, MW_, NW_, NP, true>), grid, dim3(256), ldsb, s, a);   \
    return hipGetLastError();                                                                                   \
  }
      if (a.w8) {                                                         // eight waves: the 256-channel head on 6 x 11 maps
#define PNVO_X3DW(MODE_)                                                                                                  \
  if (mode == MODE_ && mw == 3 && nw == 1) {                                                                             \
    hipLaunchKernelGGL((conv_x3_kernel<KS, STRIDE, MODE_, 3, 1, NP, true, true>), grid, dim3(512), ldsb, s, a);          \
    return hipGetLastError();                                                                                            \
  }
        PNVO_X3DW(0) PNVO_X3DW(2)
#undef PNVO_X3DW
        return hipErrorInvalidValue;
      }
      PNVO_X3D(0, 1, 1) PNVO_X3D(2, 1, 1) PNVO_X3D(0, 2, 1) PNVO_X3D(2, 2, 1) PNVO_X3D(0, 2, 2) PNVO_X3D(2, 2, 2) PNVO_X3D(0, 3, 2) PNVO_X3D(2, 3, 2)
#undef PNVO_X3D
      return hipErrorInvalidValue;
    }
  }
  PNVO_X3(0, 1, 1) PNVO_X3(1, 1, 1) PNVO_X3(0, 2, 1) PNVO_X3(1, 2, 1) PNVO_X3(0, 2, 2) PNVO_X3(1, 2, 2) PNVO_X3(0, 3, 2) PNVO_X3(1, 3, 2)      // what conv_x3_plan picks
  PNVO_X3(2, 1, 1) PNVO_X3(2, 2, 1) PNVO_X3(2, 2, 2) PNVO_X3(2, 3, 2) PNVO_X3(3, 1, 1) PNVO_X3(3, 2, 1) PNVO_X3(3, 2, 2) PNVO_X3(3, 3, 2)

  if (KS == 3 && STRIDE == 1 && NP == 2) {                                                      // wide strips (conv_x3_plan)
    PNVO_X3(0, 5, 1) PNVO_X3(1, 5, 1) PNVO_X3(2, 5, 1) PNVO_X3(3, 5, 1)   // (mode 3 on a strip plan: a 128-channel first block)
    if constexpr (KS == 3 && STRIDE == 1 && NP == 2) if (a.ksw) {          // fine plan, K split over the waves (three / four M-tiles)
#define PNVO_X3K(MODE_, MW_)                                                                                             \
  if (mode == MODE_ && mw == MW_ && nw == 1) {                                                                          \
    hipLaunchKernelGGL((conv_x3_kernel<KS, STRIDE, MODE_, MW_, 1, NP, false, false, true>), grid, dim3(256), ldsb, s, a); \
    return hipGetLastError();                                                                                           \
  }
      PNVO_X3K(0, 3) PNVO_X3K(1, 3) PNVO_X3K(2, 3) PNVO_X3K(0, 4) PNVO_X3K(1, 4) PNVO_X3K(2, 4)
#undef PNVO_X3K
      return hipErrorInvalidValue;
    }
  }
  if constexpr (KS == 3 && NP == 2) if (a.w8) {          // eight waves: the 256-channel 6 x 11 maps
#define PNVO_X3W(MODE_)                                                                                                  \
  if (mode == MODE_ && mw == 3 && nw == 1) {                                                                            \
  hipLaunchKernelGGL((conv_x3_kernel<KS, STRIDE, MODE_, 3, 1, NP, false, true>), grid, dim3(512), ldsb, s, a);        \
  return hipGetLastError();                                                                                           \
  }
    PNVO_X3W(0) PNVO_X3W(1) PNVO_X3W(2) PNVO_X3W(3)
#undef PNVO_X3W
    return hipErrorInvalidValue;
  }
#undef PNVO_X3
  return hipErrorInvalidValue;
}
}  // namespace

// Fills the plan fields of `a` from its shape fields; false: the layer is outside what the kernel covers.
namespace {
bool conv_x3_plan_impl(ConvX3Args &a, int ks, int stride, int *mw, int *nw, size_t *lds_bytes, bool fine);
long plan_wgs(const ConvX3Args &a, int nw) {
  const int ntt = a.COUTP / 32;
  return (long)a.B * a.tiles_r * a.tiles_c * ((ntt + a.wn * nw - 1) / (a.wn * nw));
}
}  // namespace

// The plan of a launch.  Round 6: when the regular plan would give the launch fewer than 224 workgroups (small batches: the deep
// stages have one or two tiles per sample) a FINE plan is tried — one N-tile per workgroup (blockIdx.y = N-tile), the four waves
// along M, no strips — which multiplies the workgroups by the layer's N-tiles at the price of staging the patch once per N-tile
// (L2 serves it).  It keeps the deep stages of 8-48-pair batches on the float16 matrix pipe, with block tails, riding
// downsample convs and in-kernel GroupNorm finalisation, where they used to fall back to the fp32-pipe kernels plus separate
// residual / normalisation passes (option x3_fine; same MFMA order per output: bit-identical raw outputs to the regular plan).
bool conv_x3_plan(ConvX3Args &a, int ks, int stride, int *mw, int *nw, size_t *lds_bytes) {
  ConvX3Args reg = a;
  int rmw = 0, rnw = 0;
  size_t rlds = 0;
  const int force0 = a.force;
  reg.force = 1;                                                       // (the regular plan's shape, whatever its size)
  const bool reg_ok = conv_x3_plan_impl(reg, ks, stride, &rmw, &rnw, &rlds, false);
  const long reg_wgs = reg_ok ? plan_wgs(reg, rnw) : 0;
  const int ntt = a.COUTP / 32;
  static const long fine_below = std::getenv("PNVO_FINE_BELOW") ? std::atol(std::getenv("PNVO_FINE_BELOW")) : 224;   // (developer sweep; 224 measured best)
  if (a.fine && a.np == 2 && ntt >= 2 && (!reg_ok || reg_wgs < fine_below)) {
    ConvX3Args f = a;
    int fmw = 0, fnw = 0;
    size_t flds = 0;
    f.force = 1;
    if (conv_x3_plan_impl(f, ks, stride, &fmw, &fnw, &flds, true)) {
      const long fw = plan_wgs(f, fnw);
      if (fw > reg_wgs && (fw >= 48 || force0)) {
        f.force = force0;
        a = f;
        *mw = fmw;
        *nw = fnw;
        *lds_bytes = flds;
        return true;
      }
    }
  }
  return conv_x3_plan_impl(a, ks, stride, mw, nw, lds_bytes, false);
}

namespace {
bool conv_x3_plan_impl(ConvX3Args &a, int ks, int stride, int *mw, int *nw, size_t *lds_bytes, bool fine) {
  if (a.CIN % 32 || a.COUTP % 32 || a.COUTP > 1024 || a.CIN > 1024) return false;
  if (!((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 2))) return false;
  a.w8 = 0;
  a.ksw = 0;
  const int ntt = a.COUTP / 32;
  // Operand bandwidth decides the wave tile: per wave and cycle the MFMAs want 16/NW bytes of A (LDS, 128 B/clk per CU) and
  // 16/MW bytes of B (L1, 64 B/clk per CU) at full rate, eight waves per CU.  (MW, NW) = (3, 2) keeps both under their limits
  // (64 and 43 B/clk); (2, 2) sits on the L1 limit, (2, 1) on both.  A layer with one N-tile cannot reuse A at all (16 B per
  // cycle and wave from LDS alone): such layers stay on the fp32 kernels.
  int TR, TC;
  if (stride == 2 && ks == 3) {
    // the patch of a stride-2 conv is four times its output tile: the tile is what 76 KB of three-piece patch allow at 32-channel
    // chunks (PR x PC <= 324 pixels of 240 B): 4 x 16 outputs on the wide maps, else whole rows
    TC = a.Wo >= 32 ? 16 : a.Wo;
    const int pc = 2 * TC + 1;
    TR = (324 / pc - 1) / 2;                                     // (two-piece patches would allow 486 pixels: 12 x 22 outputs in three tiles of four rows
                                                                 //  instead of four of three — measured neutral, 93.2 vs 93.8 us: that head is not MFMA-bound)
    if (TR < 1) return false;
    if (TR > a.Ho) TR = a.Ho;
    while (TR > 1 && TR * TC > (ntt == 2 ? 64 : (ntt == 4 ? 128 : 96))) --TR;    // M-tiles the wave grid below covers
  } else if (ntt == 1 && stride == 1 && ks == 3 && a.CIN > 32 && a.np == 2 && a.Wo < 32 && a.Ho * a.Wo <= 128) {
    // one N-tile behind MANY input channels on a small map (the compression conv, vo_cnn.py:76-101: 256 -> 31 channels on 6 x 11):
    // the whole map is one tile, the four waves along M with one M-tile each (round 6; it ran on the fp32 pipe at 60 TFLOP/s —
    // no A reuse with a single N-tile, but three float16 MFMAs per K = 16 are still five times less pipe time than eight fp32 ones)
    TC = a.Wo;
    TR = a.Ho;
  } else if (ntt == 1) {
    // one N-tile (32 output channels): all four waves along M, two M-tiles each -> 16 x 16 outputs (18 x 18 patch = 76 KB)
    if (stride != 1 || ks != 3 || a.CIN != 32) return false;
    TC = a.Wo >= 16 ? 16 : a.Wo;
    TR = 256 / TC;
    while (TR > 1 && (TR + 2) * (TC + 2) > 324) --TR;
    if (TR > a.Ho) TR = a.Ho;
    if (TR * TC <= 96) return false;                               // (small maps: not worth it)
  } else if (a.strip && !fine && a.np == 2 && stride == 1 && ks == 3 && ntt == 4 && a.Wo >= 16 && a.Ho >= 8) {
    // Round 4: what bounds these layers is the CU's vector-memory pipe, and most of its traffic is weight fragments — every workgroup
    // streams the layer's whole B operand for ITS pixels.  Wide strips (up to 288 pixels = nine M-tiles: the 12 x 22 map whole, the
    // 24 x 43 map in four) with the N-tiles split over blockIdx.y (two per workgroup, one per wave column, five M-tiles per wave)
    // halve the weight bytes per pixel of the 8 x 22 tiles.  Measured at 256 pairs: the 128-channel stage 0.074 -> 0.067 ms per conv;
    // the 64-channel stage (24 x 43 in four strips) 3 % SLOWER than its 8 x 16 tiles and left alone.
    const int ncol = (a.Wo + 21) / 22;
    TC = (a.Wo + ncol - 1) / ncol;
    TR = 288 / TC;
    if (TR > a.Ho) TR = a.Ho;
    const int nrow = (a.Ho + TR - 1) / TR;
    TR = (a.Ho + nrow - 1) / nrow;
  } else if (a.Wo >= 32) {
    TR = 8;
    TC = 16;
  } else {
    TC = a.Wo;
    int target = ntt == 4 ? 192 : (ntt >= 8 ? 96 : 128);           // pixels per tile: six / three / four M-tiles
    // fine plan with the K split over the waves: three M-tiles per tile (12 x 22 maps: three balanced tiles of four rows instead of 8 + 4)
    if (fine && a.ksw_ok && ks == 3 && stride == 1 && a.np == 2 && a.CIN >= 128) target = 96;
    TR = target / TC;
    if (TR < 1) TR = 1;
    if (TR > a.Ho) TR = a.Ho;
  }
  // (fine plan with the 12 x 22 map whole — nine M-tiles on (3,1) wave tiles, GroupNorm finalised in the kernel — was measured: three
  //  launches fewer but 32 pairs 0.626 -> 0.640 ms, and below 12 pairs its 4 B workgroups miss the threshold: not kept)
  a.TR = TR;
  a.TC = TC;
  a.tiles_r = (a.Ho + TR - 1) / TR;
  a.tiles_c = (a.Wo + TC - 1) / TC;
  a.MT = (TR * TC + 31) / 32;
  const int cs = ks == 1 ? 1 : stride;
  a.PR = (TR - 1) * cs + ks;
  a.PC = (TC - 1) * cs + ks;
  // wave grid (4 / wn) x wn and accumulators per wave (accumulators + one A set + two B sets within 256 registers)
  const bool strip = a.strip && !fine && a.np == 2 && stride == 1 && ks == 3 && ntt == 4 && a.Wo >= 16 && a.Ho >= 8 && a.MT > 6;
  if (fine) {                                                      // one N-tile per workgroup, the four waves along M
    if (a.MT > 8) return false;
    a.wn = 1;
    *mw = (a.MT + 3) / 4;
    *nw = 1;
    // three / four M-tiles behind many input channels (the 128- / 256-channel stages): the waves split K instead of M (KSW)
    if (a.ksw_ok && (a.MT == 3 || a.MT == 4) && ks == 3 && stride == 1 && a.np == 2 && a.CIN >= 128) {
      a.ksw = 1;
      *mw = a.MT;
    }
  } else if (strip) {
    a.wn = 2;
    *mw = (a.MT + 1) / 2;                                          // two wave rows
    *nw = 1;
    if (*mw > 5) return false;
    if (*mw < 5) *mw = 5;                                          // (one instantiation)
  } else if (ntt == 1) {
    a.wn = 1;
    *mw = a.CIN > 32 ? 1 : 2;
    *nw = 1;
    if (a.ksw_ok && a.CIN >= 128 && (a.MT == 3 || a.MT == 4) && ks == 3 && stride == 1 && a.np == 2) {   // the compression conv: K over the waves
      a.ksw = 1;
      *mw = a.MT;
    }
  } else if (ntt == 2) {
    a.wn = 2;
    *mw = a.MT > 2 ? 2 : 1;
    *nw = 1;
  } else if (ntt == 4) {
    a.wn = 2;
    *mw = a.MT > 4 ? 3 : 2;
    *nw = 2;
  } else {
    a.wn = 4;
    *mw = a.MT > 2 ? 3 : 2;
    *nw = 2;
    // fewer tiles than CUs (6 x 11 maps: one tile per image; 128 pairs): four N-tiles per workgroup instead of eight doubles
    // the workgroups (measured: 9.23 -> 9.03 ms per training step at 128 pairs; at 256 tiles it costs 0.127 -> 0.170 ms per conv)
    const long ntiles = (long)a.B * a.tiles_r * a.tiles_c;
    if (ntiles < 200 && a.MT <= 4) {
      a.wn = 2;
      *mw = 2;
    }
    // one tile per sample and a tile per CU or more: eight waves of (3,1) tiles, all eight N-tiles in one workgroup (W8)
    if (a.w8_ok && !fine && ntiles >= 200 && a.MT == 3 && ntt == 8 && a.np == 2 && ks == 3) {   // (stride 1 and the stride-2 block head)
      a.w8 = 1;
      a.wn = 8;
      *mw = 3;
      *nw = 1;
    }
  }
  if (!a.ksw && ((a.w8 ? 8 : 4) / a.wn) * *mw < a.MT) return false;
  // channel chunk: the largest multiple-of-32 divisor of CIN (power-of-two steps) whose three planes fit 72 KB
  // (a power of two: the stager's thread -> (pixel, 8-channel group) split uses masks; 32 always divides CIN)
  const size_t np = a.np == 2 ? 2 : 3;                          // operand pieces = LDS planes (tiles are sized for three: same plan)
  int ck = 32;
  while (ck < 256 && a.CIN % (2 * ck) == 0) ck *= 2;
  // (W8 with the whole K = 256 in one 110 KB chunk — its launches are one workgroup per CU anyway: staging 14.5 k -> 7.8 k cycles per
  //  workgroup, the conv's time unchanged, 77-82 us on either form: not kept)
  static const long w8_s2_lds = std::getenv("PNVO_X3_W8_LDS") ? std::atol(std::getenv("PNVO_X3_W8_LDS")) : 72;   // (developer sweep)
  const size_t cap = (size_t)(a.w8 && stride == 2 ? w8_s2_lds : 72) * 1024;
  while (ck > 32 && np * a.PR * a.PC * (ck * 2 + 16) > cap) ck /= 2;
  if (np * a.PR * a.PC * (ck * 2 + 16) > cap + 4096) return false;
  if (a.CIN % ck) return false;
  a.CK = ck;
  // Few workgroups (small batches: the reference's navigation loop calls with ONE pair): a workgroup walks its whole K loop alone
  // (≈ 60-110 us on the deep stages) while the fp32 kernels split small problems over the chip — they keep those launches
  // (measured: batch 1 0.41 ms against 0.72 with conv_x3 everywhere; break-even per layer at ~200 workgroups).  Option conv=x3 (a.force) takes them anyway.
  {
    const bool force = a.force != 0;
    const long wgs = (long)a.B * a.tiles_r * a.tiles_c * ((ntt + a.wn * *nw - 1) / (a.wn * *nw));
    // (break-even measured at ~200 workgroups for the six-term form; the three-term float16 form halves the K loop: batch 64
    //  — 128 workgroups on the deep stages — 1.09 ms with it against 1.18 without, batch 32 — 64 workgroups — 0.82 against 0.78)
    if (wgs < (a.np == 2 ? 112 : 192) && !force) return false;
  }
  a.slots = a.tiles_r * a.tiles_c;                               // one GroupNorm partial per tile
  *lds_bytes = np * a.PR * a.PC * (ck * 2 + 16) + (size_t)a.MT * 32 * 4 * 2;
  if (a.ksw) *lds_bytes = std::max(np * a.PR * a.PC * (ck * 2 + 16), (size_t)4 * a.MT * 4096) + (size_t)a.MT * 32 * 4 * 2;   // planes / the four waves' partial accumulator tiles, then the tables   // planes, pixel tables (a launch with a deferred GroupNorm adds 16 B per input
                                                                                // channel behind them: pnvo_run_conv — 1 KB more would cost the 64-channel convs their third workgroup per CU)
  return true;
}
}  // namespace

// Does launch_conv_x3 take the persistent resident-weight form (conv_x3p_kernel) for this planned launch?
bool conv_x3_persistent(const ConvX3Args &a, int ks, int stride, int mode, int mw, int nw) {
  const long ntiles = (long)a.B * a.tiles_r * a.tiles_c;
  const int ntt = a.COUTP / 32, pwgs = a.persist_wgs;
  const bool bres = ntt == 1 && a.CIN == 32 && nw == 1 && (mode == 0 || mode == 1);   // the layer's weights stay in registers
  return a.np == 2 && ks == 3 && stride == 1 && bres && mw * nw <= 2 && a.CK == a.CIN && pwgs >= 8 &&
         (long)a.PR * a.PC * (a.CK / 8) <= 6 * 256 && ntiles >= 2L * pwgs;
}

hipError_t launch_conv_x3(const ConvX3Args &a0, int ks, int stride, int mode, int mw, int nw, size_t lds_bytes, hipStream_t s) {
  ConvX3Args a = a0;
  static unsigned long long *prof = nullptr;             // PNVO_X3_PROF=1: phase cycles of workgroup 13, printed per launch (syncs)
  static const bool want = std::getenv("PNVO_X3_PROF") != nullptr;
  if (want && !prof) {
    (void)hipMalloc((void **)&prof, 512);
    (void)hipMemset(prof, 0, 512);
  }
  a.prof = prof;
  struct Dump {
    const ConvX3Args &a; int mw, nw; hipStream_t s; unsigned long long *prof;
    ~Dump() {
      static int calls = 0;
      if (!prof || ++calls > 40) return;
      unsigned long long h[64];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h, prof, 512, hipMemcpyDeviceToHost);
      std::fprintf(stderr, "[pnvo] conv_x3 B %d %dx%d cin %d cout %d tile %dx%d MT %d CK %d (%d,%d) wn %d: wave0 stage %llu mfma %llu epi %llu total %llu "
                           "(100MHz ticks %llu) | wave3 stage %llu mfma %llu epi %llu total %llu\n", a.B, a.Ho, a.Wo, a.CIN, a.COUTP, a.TR, a.TC,
                   a.MT, a.CK, mw, nw, a.wn, h[0], h[1], h[2], h[3], h[4], h[24], h[25], h[26], h[27]);
    }
  } dump{a, mw, nw, s, prof};
  const long ntiles = (long)a.B * a.tiles_r * a.tiles_c;
  const int ntt = a.COUTP / 32, per_wg = a.wn * nw;   // N-tiles one workgroup covers
  dim3 grid((unsigned)(((ntiles + 7) / 8) * 8), (unsigned)((ntt + per_wg - 1) / per_wg), 1u);
  // persistent form (conv_x3p_kernel): float16 pieces, 3x3 stride 1, the whole K in one staged chunk, small wave tiles, no block
  // tail in the stager, six patch pixels per thread at most, and enough tiles to walk
  const int pwgs = a.persist_wgs;
  // (measured at 256 pairs: with the weights resident the 32 -> 32 convs go 0.107 -> 0.085 ms; the prefetching form WITHOUT resident
  //  weights — 64-channel stage — is 9 % slower than one tile per workgroup, and the pooled-key input mode, which also writes the
  //  pooled activations, loses more from two workgroups per CU than it gains: both stay on conv_x3_kernel)
  if (conv_x3_persistent(a, ks, stride, mode, mw, nw)) {
    const bool bres = true;
    dim3 pg((unsigned)(pwgs & ~7), grid.y, 1u);
    if (bres) pg.x = (unsigned)(((pwgs / 3) * 2) & ~7);                 // two workgroups per CU (256 registers per lane)
#define PNVO_X3P(MODE_, MW_, NW_)                                                                        \
  if (mode == MODE_ && mw == MW_ && nw == NW_) {                                                         \
    hipLaunchKernelGGL((conv_x3p_kernel<MODE_, MW_, NW_, true>), pg, dim3(256), lds_bytes, s, a);       \
    return hipGetLastError();                                                                            \
  }
    PNVO_X3P(0, 1, 1) PNVO_X3P(1, 1, 1) PNVO_X3P(0, 2, 1) PNVO_X3P(1, 2, 1)     // (bres only: resident weights, plain / GN-input modes)
#undef PNVO_X3P
  }
  if (a.np == 2) {
    if (ks == 3 && stride == 1) return launch_ks<3, 1, 2>(a, mode, mw, nw, grid, lds_bytes, s);
    if (ks == 3 && stride == 2) return launch_ks<3, 2, 2>(a, mode, mw, nw, grid, lds_bytes, s);
    if (ks == 1 && stride == 2) return launch_ks<1, 2, 2>(a, mode, mw, nw, grid, lds_bytes, s);
    return hipErrorInvalidValue;
  }
  if (ks == 3 && stride == 1) return launch_ks<3, 1, 3>(a, mode, mw, nw, grid, lds_bytes, s);
  if (ks == 3 && stride == 2) return launch_ks<3, 2, 3>(a, mode, mw, nw, grid, lds_bytes, s);
  if (ks == 1 && stride == 2) return launch_ks<1, 2, 3>(a, mode, mw, nw, grid, lds_bytes, s);
  return hipErrorInvalidValue;
}

// B operand: out[tap][k-chunk (cin/16)][N-tile (coutp/32)][piece 3][lane = kh*32 + n][8 bf16]
//            = piece of W[N-tile*32 + n][16 kc + 8 kh + j][tap]   (hi + mid + lo == the float32 weight exactly)
void pack_conv_x3_weight(const float *oihw, int cout, int cin, int cinp, int coutp, int kh_, int kw_, unsigned short *out) {
  const int T = kh_ * kw_, kct = cinp / 16, ntt = coutp / 32;
  auto bf = [](float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  auto tof = [](unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  for (int tap = 0; tap < T; ++tap)
    for (int kc = 0; kc < kct; ++kc)
      for (int nt = 0; nt < ntt; ++nt)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int co = nt * 32 + (ln & 31), ci = 16 * kc + 8 * (ln >> 5) + j;
            float v = 0.f;
            if (co < cout && ci < cin) v = oihw[((size_t)co * cin + ci) * T + tap];
            const unsigned short h = bf(v);
            const float r1 = v - tof(h);
            const unsigned short m = bf(r1);
            const unsigned short l = bf(r1 - tof(m));
            const size_t base = ((((size_t)tap * kct + kc) * ntt + nt) * 3) * 64 * 8 + (size_t)ln * 8 + j;
            out[base] = h;
            out[base + 64 * 8] = m;
            out[base + 2 * 64 * 8] = l;
          }
}

// Two float16 pieces of scale * W (scale = a power of two that puts the layer's largest weight near 2^12: the second piece of
// any weight within 2^-9 of the largest stays a normal float16; smaller ones keep an absolute resolution of 2^-36 of the
// largest).  Layout as above with two pieces per N-tile.  Returns the scale's inverse (the kernel's p.oscale).
static unsigned short f32_to_f16_rne(float f) {
  const _Float16 h = (_Float16)f;                                  // host conversion: round to nearest even
  unsigned short u;
  std::memcpy(&u, &h, 2);
  return u;
}
static float f16_to_f32(unsigned short u) {
  _Float16 h;
  std::memcpy(&h, &u, 2);
  return (float)h;
}
float pack_conv_x2_weight(const float *oihw, int cout, int cin, int cinp, int coutp, int kh_, int kw_, unsigned short *out) {
  const int T = kh_ * kw_, kct = cinp / 16, ntt = coutp / 32;
  float mx = 0.f;
  for (size_t k = 0; k < (size_t)cout * cin * T; ++k) mx = std::fmax(mx, std::fabs(oihw[k]));
  int e = 0;
  if (mx > 0.f) std::frexp(mx, &e);                                // mx = f * 2^e, f in [0.5, 1)
  const float scale = std::ldexp(1.0f, 12 - e), inv = std::ldexp(1.0f, e - 12);
  for (int tap = 0; tap < T; ++tap)
    for (int kc = 0; kc < kct; ++kc)
      for (int nt = 0; nt < ntt; ++nt)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int co = nt * 32 + (ln & 31), ci = 16 * kc + 8 * (ln >> 5) + j;
            float v = 0.f;
            if (co < cout && ci < cin) v = oihw[((size_t)co * cin + ci) * T + tap] * scale;
            const unsigned short h = f32_to_f16_rne(v);
            const unsigned short m = f32_to_f16_rne(v - f16_to_f32(h));
            const size_t base = ((((size_t)tap * kct + kc) * ntt + nt) * 2) * 64 * 8 + (size_t)ln * 8 + j;
            out[base] = h;
            out[base + 64 * 8] = m;
          }
  return inv;
}

// The same packing on the device from an OIHW float32 weight (the training step's flat parameter buffer: after an optimiser
// step the eval forward must see the new weights without a host round trip).  One thread per packed element triple.
// transposed = 1: the operand of the backward-data conv of the layer whose OIHW weight is `w` [cin][cout][T] as seen from
// here (output channels = the layer's inputs, input channels = its outputs, taps flipped).
__global__ __launch_bounds__(256) void conv_x3_repack_kernel(const float *w, int cout, int cin, int cinp, int coutp, int T,
                                                           int transposed, unsigned short *out, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e & 7), ln = (int)((e >> 3) & 63);
  long r = e >> 9;
  const int ntt = coutp / 32, kct = cinp / 16;
  const int nt = (int)(r % ntt);
  r /= ntt;
  const int kc = (int)(r % kct), tap = (int)(r / kct);
  const int co = nt * 32 + (ln & 31), ci = 16 * kc + 8 * (ln >> 5) + j;
  const float v = (co < cout && ci < cin) ? (transposed ? w[((long)ci * cout + co) * T + (T - 1 - tap)] : w[((long)co * cin + ci) * T + tap])
                                          : 0.f;
  const unsigned h = pack2(v, 0.f) & 0xffffu;
  const float r1 = v - lo_f(h);
  const unsigned m = pack2(r1, 0.f) & 0xffffu;
  const unsigned l = pack2(r1 - lo_f(m), 0.f) & 0xffffu;
  const long base = ((((long)tap * kct + kc) * ntt + nt) * 3) * 512 + (long)ln * 8 + j;
  out[base] = (unsigned short)h;
  out[base + 512] = (unsigned short)m;
  out[base + 1024] = (unsigned short)l;
}

// ---- the two-piece float16 operand on the device (training forward: the weights move every optimiser step)
// max |w| of every segment of the flat parameter buffer -> out[2 seg] = scale = 2^(12 - e), out[2 seg + 1] = 1 / scale, mx = f * 2^e
// with f in [0.5, 1)  (pack_conv_x2_weight's rule, on the device).  32 workgroups per segment fold their part into an integer
// maximum of the float bits (bits[seg], behind the 2 nseg floats of `out`); a one-block kernel turns the maxima into the scale pairs
// and clears them for the next step.  (One workgroup per segment walked the 2.4 MB of a 256-channel conv alone: 103 us per step.)
constexpr int X2_SCALE_CHUNKS = 32;
__global__ __launch_bounds__(256) void conv_x2_absmax_kernel(const float *params, const long *seg, unsigned *bits) {
  const long off = seg[2 * blockIdx.x], n = seg[2 * blockIdx.x + 1];
  const long per = (n + X2_SCALE_CHUNKS - 1) / X2_SCALE_CHUNKS;
  const long lo = per * blockIdx.y, hi = lo + per < n ? lo + per : n;
  float mx = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 1024) {
    const float a = fabsf(params[off + i]);
    const float b = i + 256 < hi ? fabsf(params[off + i + 256]) : 0.f;
    const float c = i + 512 < hi ? fabsf(params[off + i + 512]) : 0.f;
    const float d = i + 768 < hi ? fabsf(params[off + i + 768]) : 0.f;
    mx = fmaxf(fmaxf(mx, fmaxf(a, b)), fmaxf(c, d));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(&bits[blockIdx.x], __builtin_bit_cast(unsigned, mx));   // (order of floats >= 0 = order of their bits)
}

__global__ __launch_bounds__(256) void conv_x2_scale_finish_kernel(unsigned *bits, float *out, int nseg) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nseg) return;
  const unsigned mb = bits[k];
  bits[k] = 0u;
  int e = 0;
  if (mb != 0u) e = (int)((mb >> 23) & 0xffu) - 126;
  out[2 * k] = __builtin_bit_cast(float, (unsigned)(12 - e + 127) << 23);
  out[2 * k + 1] = __builtin_bit_cast(float, (unsigned)(e - 12 + 127) << 23);
}

// out: [2 nseg] floats followed by [nseg] unsigned maxima (zero between calls)
hipError_t launch_conv_x2_scales(const float *params, const long *seg_dev, int nseg, float *out, hipStream_t s) {
  if (nseg <= 0) return hipSuccess;
  unsigned *bits = reinterpret_cast<unsigned *>(out + 2 * nseg);
  hipLaunchKernelGGL(conv_x2_absmax_kernel, dim3((unsigned)nseg, X2_SCALE_CHUNKS), dim3(256), 0, s, params, seg_dev, bits);
  hipLaunchKernelGGL(conv_x2_scale_finish_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, s, bits, out, nseg);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void conv_x2_repack_kernel(const float *w, int cout, int cin, int cinp, int coutp, int T, int transposed,
                                                           const float *scale, unsigned short *out, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e & 7), ln = (int)((e >> 3) & 63);
  long r = e >> 9;
  const int ntt = coutp / 32, kct = cinp / 16;
  const int nt = (int)(r % ntt);
  r /= ntt;
  const int kc = (int)(r % kct), tap = (int)(r / kct);
  const int co = nt * 32 + (ln & 31), ci = 16 * kc + 8 * (ln >> 5) + j;
  // transposed = 1: the operand of the backward-data conv (w is [cin][cout][T] as seen from here, taps flipped)
  const float v = (co < cout && ci < cin) ? (transposed ? w[((long)ci * cout + co) * T + (T - 1 - tap)] : w[((long)co * cin + ci) * T + tap]) * scale[0]
                                          : 0.f;
  const _Float16 h = (_Float16)v;
  const _Float16 m = (_Float16)(v - (float)h);
  const long base = ((((long)tap * kct + kc) * ntt + nt) * 2) * 512 + (long)ln * 8 + j;
  out[base] = __builtin_bit_cast(unsigned short, h);
  out[base + 512] = __builtin_bit_cast(unsigned short, m);
}

hipError_t launch_conv_x2_repack(const float *w_oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, const float *scale_dev,
                                 unsigned short *out, hipStream_t s, int transposed) {
  const long total = (long)kh * kw * (cinp / 16) * (coutp / 32) * 512;
  hipLaunchKernelGGL(conv_x2_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, cout, cin, cinp, coutp,
                     kh * kw, transposed, scale_dev, out, total);
  return hipGetLastError();
}

hipError_t launch_conv_x3_repack(const float *w_oihw, int cout, int cin, int cinp, int coutp, int kh, int kw, int transposed,
                                 unsigned short *out, hipStream_t s) {
  const long total = (long)kh * kw * (cinp / 16) * (coutp / 32) * 512;
  hipLaunchKernelGGL(conv_x3_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w_oihw, cout, cin, cinp, coutp,
                     kh * kw, transposed, out, total);
  return hipGetLastError();
}

}  // namespace pnvo
