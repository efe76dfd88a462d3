// stem_dd.hip — the fused 7x7 stride-2 stem for models whose input carries one-hot discretised-depth channels
// (gfx950 only).  Same role / epilogue contract as stem_lds.hip, but the K loop is split by the STRUCTURE of the input:
//
//   * DENSE channels (rgb, depth, top-down view: 10 of the default model's 30) + one INDICATOR channel (12 with padding)
//     go through the fp32 matrix cores: 3 x v_mfma_f32_16x16x4_f32 per tap and 16-channel output tile instead of 8;
//   * the 2 x BINS ONE-HOT channels contribute, per tap and frame, exactly one pre-scaled weight row
//         sum_c W[co][c][tap] * (onehot_c - mean_c)/std_c  =  W[co][bin][tap]/std_bin  -  sum_c W[co][c][tap]*mean_c/std_c
//     The first term is a GATHER from a table T[tap][bin][frame][co] (LDS-resident per kernel row kh) added on the VALU,
//     concurrently with the MFMAs; the second term is constant wherever the tap lies inside the image and 0 in the conv's
//     zero padding (padding is applied AFTER whitening, vo_cnn.py:176-177), i.e. it is the weight of an "inside the
//     image" indicator channel — which is how it rides the matrix-core path and stays exact at the borders.
//   Every product the reference computes is either computed here or is an exact zero; only the fp32 summation order
//   differs.  SURVEY.md §7 "hard parts" names this design choice; roofline figures still use the algorithmic FLOPs.
//
// Workgroup = 8 waves, tile = 8 output rows x 16 columns, wave w = output row w (persistent workgroups, 2 per CU).
// The K loop is a sequence of BLOCKS (kernel row kh, pixel half HP, kernel column kw), 98 per tile and wave:
//   MFMA   : block (kh,HP,kw) multiplies the wave's 16 pixels with tap (kh,kw) for output channels 16*HP.. (K = 12:
//            3 MFMAs); weights stream from L2 (b96 buffer loads, prefetched 4 blocks ahead), the whitened dense patch
//            sits in LDS as 12 channel planes of 21 x 37 pixels (odd plane pitch: the A reads of a half-wave cover the
//            32 banks exactly once; a pixel-major [777][12] layout was 4-way conflicted and cost half the LDS time);
//   gather : block (kh,HP,kw) adds the table rows of pixels 8*HP.. of the wave's row.  lane = (frame, channel): one
//            ds_read_b32 fetches both frames' rows of one pixel and the 64 lanes hit distinct banks whatever the bins are
//            (a table row is [2 frames][32 channels] = 256 B).  The row offsets (bin * 256) of the 21 patch columns a
//            half-row needs are held in registers for its 7 taps.  The gathers of block g+1 are issued between the
//            MFMAs of block g (software pipeline inside a half-row), their adds follow one block later;
//   table  : one kernel-row slice (7 taps, 19.25 KiB) is resident, the next one arrives by LDS-DMA (global_load_lds) while
//            the current one is used: one workgroup barrier per kernel row.  The last slice's buffer doubles as the
//            epilogue's exchange area;
//   LDS    : 2 x 20 KiB table + 12 x 777 x 4 B dense patch + 2 x 21 x 38 x 2 B row offsets = 79.5 KiB -> 2 WGs per CU.
// The blocks are inline asm: left to itself the compiler hoists every gather of a kernel row above the adds and spills
// them (840 B scratch/lane, 4x slower).  Measured (B=256, 341x192): 2.29 ms vs 3.73 ms for the dense stem; phases per
// tile in cycles (PNVO_STEM_DBG=9): staging 10.8k, K loop 51.6k (MFMA-bound would be 37.6k), epilogue 3.7k.
//
// Contract: the discretised-depth input must be one-hot per frame (what the reference's _discretize_depth_func
// produces and asserts, base_trainer_with_vo.py:163).  A pixel whose BINS values are not exactly one 1 and zeros raises
// the host-visible flag `p.bad_onehot` (pnvo_check_inputs); callers with soft depth codes select PNVO_STEM=dense.
#include <type_traits>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

namespace {
constexpr int TH = 8, TW = 16;
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;   // 21 x 37
constexpr int NPIX = PH * PW;                     // 777
constexpr int NTHREADS = 512;
constexpr int CD = 12;                            // dense channels in LDS: 12 planes of NPIX floats (odd plane pitch:
                                                  //   the MFMA A reads of a half-wave then cover all 32 banks once)
constexpr int OW = 38;                            // row pitch of the u16 row-offset planes (even: pairs are dwords)
constexpr int NT16 = 2, COUT = 32;
constexpr int GP = 36;                            // pitch of the gather-result exchange rows (bank spread)

constexpr int slice_floats_c(int bins) { return (7 * (bins + 1) * 64 + 255) / 256 * 256; }   // whole KiB (LDS-DMA)

__device__ __forceinline__ f32x3 wload3(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0));
}
}  // namespace

template <int BINS, bool PROF, int ABL>   // PROF: per-phase s_memtime brackets into p.prof (PNVO_STEM_DBG=9)
                                          // ABL (timing experiments, wrong results): 1 = no gathers/adds, 2 = no MFMAs
__global__ __launch_bounds__(NTHREADS, 4) void stem_dd_kernel(const StemDDArgs p) {
  constexpr int BROWS = BINS + 1;                 // + the all-zero row for padding pixels
  constexpr int KWB = BROWS * 256;                // bytes of one kernel-column block of the table
  constexpr int SLICE_F = slice_floats_c(BINS), SLICE_B = SLICE_F * 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tab = lds;                                                         // 2 x [7 kw][BROWS][2 f][32 c]  (offset 0:
  float *dense = lds + 2 * SLICE_F;                                         //   gather immediates stay < 64 KiB)
  unsigned short *offs = reinterpret_cast<unsigned short *>(dense + NPIX * CD);   // [2 f][PH][OW] byte offsets bin*256

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // = output row of the tile
  const int i = lane & 15, kq = lane >> 4;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, 49u * NT16 * 768u, 0x00020000);
  const unsigned wlane = (unsigned)lane * 12u;
  constexpr unsigned SB = NT16 * 768u;

  // Table slice: global -> LDS by LDS-DMA (16 B per lane, 1 KiB per wave instruction, no VGPRs).  Issued through inline
  // asm on purpose: the compiler would otherwise order every later LDS read behind the copy (s_waitcnt vmcnt(0) right
  // after issue).  The waves wait for it explicitly (row_end) one kernel row later.
  const unsigned tab_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)tab;
  auto dma_slice = [&](int kh, int buf) {
    const char *g = reinterpret_cast<const char *>(p.table + (long)kh * SLICE_F) + lane * 16;
    for (int c = wave; c < SLICE_B / 1024; c += NTHREADS / 64) {
      const char *gc = g + c * 1024;
      const unsigned lc = tab_lds + (unsigned)(buf * SLICE_B + c * 1024);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gc), "s"(lc) : "memory");   // (m0 is not tracked by the compiler in this kernel)
    }
  };

  long long pt[5] = {0, 0, 0, 0, 0};
  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int ho0 = ty * TH, wo0 = tx * TW;
    const int hi_base = 2 * ho0 - 3, wi_base = 2 * wo0 - 3;

    const long long tp0 = PROF ? clock64() : 0;
    dma_slice(0, 0);                              // the previous tile's last barrier freed both table buffers
    // dense staging role: thread -> (pixel, 16-byte slot); 510 threads cover 170 pixels per pass.  Re-derived per tile
    // (the empty asm stops the compiler from keeping ~40 loop-invariant staging registers alive across the K loop).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int sg = tid % 3, pp0 = tid / 3;
    const bool stager = tid < 510;
    const SrcPiece e0 = p.pieces[sg][0][0], e1 = p.pieces[sg][0][1];
    const f32x4 wsc = *reinterpret_cast<const f32x4 *>(p.sc + 4 * sg);
    const f32x4 wsh = *reinterpret_cast<const f32x4 *>(p.sh + 4 * sg);
    // ---- stage 1/2: dense patch (whitened; indicator channel = 1 inside the image), loads batched for MLP
    {
      constexpr int PSTEP = 170;
      constexpr int NB = (NPIX + PSTEP - 1) / PSTEP;        // 5
      f32x2 x0[NB], x1[NB];
      bool ok[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = pp0 + b * PSTEP;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int hi = hi_base + pr, wi = wi_base + pc;
        ok[b] = stager && pp < NPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const long pix = ((long)n * p.H + hi) * p.W + wi;
        const float *a0 = (ok[b] && e0.base != nullptr) ? e0.base + pix * e0.nch + e0.choff : p.zero_page;
        const float *a1 = (ok[b] && e1.base != nullptr) ? e1.base + pix * e1.nch + e1.choff : p.zero_page;
        x0[b] = *reinterpret_cast<const f32x2 *>(a0);
        x1[b] = *reinterpret_cast<const f32x2 *>(a1);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = pp0 + b * PSTEP;
        if (stager && pp < NPIX) {
          f32x4 v = {x0[b][0], x0[b][1], x1[b][0], x1[b][1]};
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = ok[b] ? __builtin_fmaf(v[t], wsc[t], wsh[t]) : 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) dense[(4 * sg + t) * NPIX + pp] = v[t];
        }
      }
    }
    // ---- stage 2/2: depth bins of the patch pixels: thread -> pixel (both frames), bin = position of the single 1
    {
      constexpr int NB = (NPIX + NTHREADS - 1) / NTHREADS;  // 2
      constexpr int NV = 2 * BINS / 4;                      // float4 per pixel
      f32x4 dv[NB][NV];
      bool okp[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = tid + b * NTHREADS;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int hi = hi_base + pr, wi = wi_base + pc;
        okp[b] = pp < NPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const long pix = ((long)n * p.H + hi) * p.W + wi;
        const float *src = okp[b] ? p.dd + pix * (2 * BINS) : p.zero_page;
#pragma unroll
        for (int k = 0; k < NV; ++k) dv[b][k] = *reinterpret_cast<const f32x4 *>(src + (okp[b] ? 4 * k : 0));
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = tid + b * NTHREADS;
        if (pp < NPIX) {
          // branch-free: sum = 1 and every value in {0, 1} (v*v == v)  <=>  exactly one 1; bin = sum of k * v[k]
          int bin[2] = {BINS, BINS};
          if (okp[b]) {
            float dev = 0.f;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              float sum = 0.f, pos = 0.f;
#pragma unroll
              for (int k = 0; k < BINS; ++k) {
                const float v = dv[b][(f * BINS + k) >> 2][(f * BINS + k) & 3];
                sum += v;
                pos = __builtin_fmaf(v, (float)k, pos);
                dev = __builtin_fmaxf(dev, __builtin_fabsf(__builtin_fmaf(v, v, -v)));
              }
              dev = __builtin_fmaxf(dev, __builtin_fabsf(sum - 1.0f));
              bin[f] = (int)pos;
            }
            if (!(dev == 0.f)) {                            // also catches NaN
              *p.bad_onehot = 1;
              bin[0] = bin[1] = BINS;
            }
          }
          const int pr = pp / PW, pc = pp - pr * PW;
          offs[pr * OW + pc] = (unsigned short)(bin[0] * 256);
          offs[(PH + pr) * OW + pc] = (unsigned short)(bin[1] * 256);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of table slice 0 is in LDS
    __syncthreads();

    const long long tp1 = PROF ? clock64() : 0;
    // ---- K loop.  One BLOCK = (kernel row kh, pixel half HP, kernel column kw): 8 gathers (pixels 8*HP.. of this wave's
    //      row, both frames) + the 3 MFMAs of tap (kh,kw) for output channels 16*HP..  Blocks are written as inline asm:
    //      left to itself the compiler hoists every gather of a kernel row above the adds and spills them to scratch.
    //      In-order issue does the overlap: 8 ds_reads go out, the 3 MFMAs cover their latency, then the 8 adds.
    // one accumulator per (n-tile, k-step): the MFMAs of a block are independent, so neither the 40-cycle dependent
    // latency of 16x16x4 nor the same-accumulator issue cliff (MI355X_MICROARCH.md) sits between them
    f32x4 acc3[NT16][3];
    f32x2 g2[8];                                                        // gathered sums: [half-row][pixel pair], 2 channels
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt)
#pragma unroll
      for (int t = 0; t < 3; ++t) acc3[nt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) g2[q] = f32x2{0.f, 0.f};
    const float *abase = dense + (3 * kq) * NPIX + (2 * wave) * PW + 2 * i;      // MFMA A: pixel i, channels 3kq..3kq+2
    // gather lanes: lane = (pixel parity P, frame f, channel pair cp): one ds_read_b64 serves TWO pixels (lanes 0-31 / 32-63)
    // and lands in an aligned register pair that v_pk_add_f32 accumulates -- half the VALU instructions of one channel per
    // lane for the same LDS bytes (VALU and MFMA instructions of a SIMD do not overlap freely: tools/ubench/mfma_valu.hip)
    const int gP = lane >> 5, gf = (lane >> 4) & 1, gcp = lane & 15;
    const unsigned *obase = reinterpret_cast<const unsigned *>(offs + (gf * PH + 2 * wave) * OW) + gP;   // this lane's frame, +2 columns for P
    const unsigned lane8 = tab_lds + (unsigned)(gf * 128 + gcp * 8);    // LDS address of (frame, channel pair) in row 0
    const unsigned abase_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)abase;

    auto ld_b = [&](int tap, int nt) -> f32x3 {                         // tap >= 49: out of range -> zeros
      return wload3(rw, wlane + (unsigned)nt * 768u, (unsigned)tap * SB);
    };
    auto ld_a = [&](int kh, int kw) -> f32x3 {                          // (kh = 7 reads past the patch: unused)
      const float *a = abase + kh * PW + kw;
      return f32x3{a[0], a[NPIX], a[2 * NPIX]};
    };
    constexpr int BD = 4;                                               // weight prefetch distance in blocks
    f32x3 bq[BD], a_cur = ld_a(0, 0);
#pragma unroll
    for (int d = 0; d < BD; ++d) bq[d] = ld_b(d, 0);
    dma_slice(1, 1);

    auto grow = [&](int kh, auto bufc) {
      constexpr int BUF = decltype(bufc)::value;
      const unsigned arow = abase_lds + (unsigned)(kh * PW) * 4u;       // this kernel row of the dense patch
#pragma unroll
      for (int HP = 0; HP < 2; ++HP) {
        unsigned o[19];
        f32x2 tg[2][4];
        const unsigned *orow = obase + (kh * OW + 16 * HP) / 2;        // 10 dwords = 20 columns (19 used)
#pragma unroll
        for (int j = 0; j < 10; ++j) {
          const unsigned pr = orow[j];
          o[2 * j] = (pr & 0xffffu) + lane8;
          if (2 * j + 1 < 19) o[2 * j + 1] = (pr >> 16) + lane8;
        }
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
          const int x2 = HP * 7 + kw + BD, x1 = HP * 7 + kw + 1;      // blocks g+BD (weights) and g+1 (pixels)
          const f32x3 b_pre = ld_b((kh + x2 / 14) * 7 + (x2 % 14) % 7, (x2 % 14) / 7);
          // the next block's MFMA A operands, issued from asm BEFORE this block's gathers: loaded by C++ they are ds_reads
          // the compiler tracks, and its lgkmcnt(3) before the next MFMAs drained the gathers in flight at every block
          // (the gather latency was never hidden).  Being older than the gathers, the adds' lgkmcnt(4) covers them.
          f32x3 a_nxt;
          asm volatile("ds_read_b32 %[a0], %[ab] offset:%[i0]\n\t"
                       "ds_read_b32 %[a1], %[ab] offset:%[i1]\n\t"
                       "ds_read_b32 %[a2], %[ab] offset:%[i2]"
                       : [a0] "=&v"(a_nxt[0]), [a1] "=&v"(a_nxt[1]), [a2] "=&v"(a_nxt[2])
                       : [ab] "v"(arow), [i0] "i"((((x1 / 14) * PW + (x1 % 14) % 7)) * 4),
                         [i1] "i"((((x1 / 14) * PW + (x1 % 14) % 7) + NPIX) * 4),
                         [i2] "i"((((x1 / 14) * PW + (x1 % 14) % 7) + 2 * NPIX) * 4));
          f32x2 *g = g2 + 4 * HP;
          f32x2(&tc)[4] = tg[kw & 1];                                   // this block's gathered rows
          f32x2(&tn)[4] = tg[(kw & 1) ^ 1];                             // next block's (in flight)
          // pixel pair q of the half-row = pixels 2q + P: patch columns 4q + kw (+2 for P, folded into obase)
          if (kw == 0 && ABL != 1)                                      // first block of the half-row: own gathers
            asm volatile(
                "ds_read_b64 %[t0], %[o0] offset:%[imm]\n\t"
                "ds_read_b64 %[t1], %[o1] offset:%[imm]\n\t"
                "ds_read_b64 %[t2], %[o2] offset:%[imm]\n\t"
                "ds_read_b64 %[t3], %[o3] offset:%[imm]"
                : [t0] "=&v"(tc[0]), [t1] "=&v"(tc[1]), [t2] "=&v"(tc[2]), [t3] "=&v"(tc[3])
                : [o0] "v"(o[0]), [o1] "v"(o[4]), [o2] "v"(o[8]), [o3] "v"(o[12]), [imm] "i"(BUF * SLICE_B));
          if (kw < 6 && ABL == 2) {                                     // ablation: the gathers alone
            asm volatile(
                "ds_read_b64 %[t0], %[o0] offset:%[imm]\n\t"
                "ds_read_b64 %[t1], %[o1] offset:%[imm]\n\t"
                "ds_read_b64 %[t2], %[o2] offset:%[imm]\n\t"
                "ds_read_b64 %[t3], %[o3] offset:%[imm]"
                : [t0] "=&v"(tn[0]), [t1] "=&v"(tn[1]), [t2] "=&v"(tn[2]), [t3] "=&v"(tn[3])
                : [o0] "v"(o[kw + 1]), [o1] "v"(o[kw + 5]), [o2] "v"(o[kw + 9]), [o3] "v"(o[kw + 13]),
                  [imm] "i"(BUF * SLICE_B + (kw + 1) * KWB));
          } else if (ABL == 2) {
          } else if (kw < 6 && ABL == 0) {                              // MFMAs of this block + gathers of the next one
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %[c0], %[a0], %[b0], %[c0]\n\t"
                "ds_read_b64 %[t0], %[o0] offset:%[imm]\n\t"
                "ds_read_b64 %[t1], %[o1] offset:%[imm]\n\t"
                "v_mfma_f32_16x16x4_f32 %[c1], %[a1], %[b1], %[c1]\n\t"
                "ds_read_b64 %[t2], %[o2] offset:%[imm]\n\t"
                "ds_read_b64 %[t3], %[o3] offset:%[imm]\n\t"
                "v_mfma_f32_16x16x4_f32 %[c2], %[a2], %[b2], %[c2]"
                : [c0] "+v"(acc3[HP][0]), [c1] "+v"(acc3[HP][1]), [c2] "+v"(acc3[HP][2]), [t0] "=&v"(tn[0]), [t1] "=&v"(tn[1]),
                  [t2] "=&v"(tn[2]), [t3] "=&v"(tn[3])
                : [o0] "v"(o[kw + 1]), [o1] "v"(o[kw + 5]), [o2] "v"(o[kw + 9]), [o3] "v"(o[kw + 13]), [a0] "v"(a_cur[0]),
                  [a1] "v"(a_cur[1]), [a2] "v"(a_cur[2]), [b0] "v"(bq[0][0]), [b1] "v"(bq[0][1]), [b2] "v"(bq[0][2]),
                  [imm] "i"(BUF * SLICE_B + (kw + 1) * KWB));
          } else {
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %[c0], %[a0], %[b0], %[c0]\n\t"
                "v_mfma_f32_16x16x4_f32 %[c1], %[a1], %[b1], %[c1]\n\t"
                "v_mfma_f32_16x16x4_f32 %[c2], %[a2], %[b2], %[c2]"
                : [c0] "+v"(acc3[HP][0]), [c1] "+v"(acc3[HP][1]), [c2] "+v"(acc3[HP][2])
                : [a0] "v"(a_cur[0]), [a1] "v"(a_cur[1]), [a2] "v"(a_cur[2]), [b0] "v"(bq[0][0]), [b1] "v"(bq[0][1]),
                  [b2] "v"(bq[0][2]));
          }
          // this block's rows were requested one block ago: LDS returns in order, so "at most 4 outstanding" = landed
#define PNVO_DD_ADDS(WAIT)                                                                                              \
  asm volatile(WAIT "\n\t"                                                                                              \
               "v_pk_add_f32 %[g0], %[g0], %[t0]\n\t"                                                                   \
               "v_pk_add_f32 %[g1], %[g1], %[t1]\n\t"                                                                   \
               "v_pk_add_f32 %[g2], %[g2], %[t2]\n\t"                                                                   \
               "v_pk_add_f32 %[g3], %[g3], %[t3]\n\t"                                                                   \
               "s_nop 1"                                                                                                \
               : [g0] "+v"(g[0]), [g1] "+v"(g[1]), [g2] "+v"(g[2]), [g3] "+v"(g[3])                                   \
               : [t0] "v"(tc[0]), [t1] "v"(tc[1]), [t2] "v"(tc[2]), [t3] "v"(tc[3]))
          if (ABL == 1) {
          } else if (kw < 6)
            PNVO_DD_ADDS("s_waitcnt lgkmcnt(4)");
          else
            PNVO_DD_ADDS("s_waitcnt lgkmcnt(0)");
#undef PNVO_DD_ADDS
#pragma unroll
          for (int d = 0; d + 1 < BD; ++d) bq[d] = bq[d + 1];
          bq[BD - 1] = b_pre;
          a_cur = a_nxt;
        }
      }
    };
    long long tbar = 0;
    auto row_end = [&](int kh) {                  // slice kh+1 has landed for everyone; slice kh's buffer is free
      const long long b0 = PROF ? clock64() : 0;
      // vmcnt retires in order: "at most BD outstanding" leaves only the weight prefetches of the next BD blocks in
      // flight — the table DMA of this row's start (>= 10 loads older) has landed.  vmcnt(0) would drain the prefetches
      // (an L2 round trip) at every row end.
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BD) : "memory");
      __syncthreads();
      if (PROF) tbar += clock64() - b0;
      if (kh + 2 < 7) dma_slice(kh + 2, kh & 1);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    for (int kh = 0; kh < 6; kh += 2) {
      grow(kh, C0{});
      row_end(kh);
      grow(kh + 1, C1{});
      row_end(kh + 1);
    }
    grow(6, C0{});

    const long long tp2 = PROF ? clock64() : 0;
    // ---- epilogue: gathered sums (lane = frame x channel) -> MFMA C layout through LDS, store, per-tile GroupNorm
    //      partials.  Fixed order: mfma + (prev-frame row + cur-frame row).  The exchange lives in table buffer 1 (free
    //      since the barrier after kernel row 5), so no barrier is needed here: a wave only reads back its own rows.
    static_assert(TH * 16 * GP + TH * COUT * 2 <= SLICE_F, "exchange area must fit one table buffer");
    f32x4 acc[NT16];
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) acc[nt] = (acc3[nt][0] + acc3[nt][1]) + acc3[nt][2];   // fixed order
    float *gx = tab + SLICE_F;                              // [8 rows][16 pixels][GP]
    float *red = gx + TH * 16 * GP;
#pragma unroll
    for (int q = 0; q < 8; ++q) {                           // prev-frame + cur-frame rows: lanes 16 apart (ds_swizzle xor 0x10)
      f32x2 v = g2[q];
      float o0, o1;                                         // (inline asm: the compiler folded the two builtin swizzles of a
      asm volatile("ds_swizzle_b32 %0, %2 offset:swizzle(SWAP,16)\n\t"    //  register pair into one and reused its result)
                   "ds_swizzle_b32 %1, %3 offset:swizzle(SWAP,16)\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(o0), "=&v"(o1)
                   : "v"(v[0]), "v"(v[1]));
      if (gf == 0) {                                        // own = prev frame: prev + cur, as before
        const int px = 8 * (q >> 2) + 2 * (q & 3) + gP;
        *reinterpret_cast<f32x2 *>(gx + (wave * 16 + px) * GP + 2 * gcp) = f32x2{v[0] + o0, v[1] + o1};
      }
    }
    {
      const int ho = ho0 + wave;
      const bool rvalid = ho < p.Ho;
      float *yrow = p.y + (((long)n * p.Ho + ho) * p.Wo) * COUT;
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wo = wo0 + 4 * kq + r;
          const bool ok = rvalid && wo < p.Wo;
          const float g = gx[(wave * 16 + 4 * kq + r) * GP + nt * 16 + i];
          const float v = ok ? acc[nt][r] + g : 0.f;
          if (ok) yrow[(long)wo * COUT + nt * 16 + i] = v;
          s1 += v;
          s2 = __builtin_fmaf(v, v, s2);
        }
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kq == 0) {
          red[(wave * COUT + nt * 16 + i) * 2] = s1;
          red[(wave * COUT + nt * 16 + i) * 2 + 1] = s2;
        }
      }
    }
    const long long tp3 = PROF ? clock64() : 0;
    __syncthreads();                                        // every wave is past the K loop: dense / offs / table buffer 0
    if ((int)threadIdx.x < COUT) {                          //   may be overwritten by the next tile's staging
      const int c = threadIdx.x;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < TH; ++w) {
        s1 += red[(w * COUT + c) * 2];
        s2 += red[(w * COUT + c) * 2 + 1];
      }
      const int slot = ty * p.tiles_x + tx;
      float *dst = p.stats + (((long)n * p.slots + slot) * COUT + c) * 2;
      dst[0] = s1;
      dst[1] = s2;
    }
    if (PROF) {
      pt[0] += tp1 - tp0;
      pt[1] += tp2 - tp1;
      pt[2] += tp3 - tp2;
      pt[3] += 1;
      pt[4] += tbar;
    }
  }
  if (PROF && threadIdx.x == 0 && p.prof != nullptr)
    for (int k = 0; k < 5; ++k) atomicAdd(p.prof + k, (unsigned long long)pt[k]);
}

int stem_dd_slice_floats(int bins) { return slice_floats_c(bins); }

bool stem_dd_supported(int bins) { return bins == 10; }

int stem_dd_slots(int Ho, int Wo) { return ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW); }

// Dense + indicator weights [cout][12][49] (OIHW over the 12 dense channels) -> MFMA B operand order
// [tap][n-tile][lane = kq*16 + n][t] = W[n-tile*16 + n][3*kq + t][tap]   (12 B per lane, b96 loads).
void pack_stem_dd_weight(const float *w, int cout, float *out) {
  for (int tap = 0; tap < 49; ++tap)
    for (int nt = 0; nt < cout / 16; ++nt)
      for (int kq = 0; kq < 4; ++kq)
        for (int n = 0; n < 16; ++n)
          for (int t = 0; t < 3; ++t)
            out[(((size_t)tap * (cout / 16) + nt) * 64 + kq * 16 + n) * 3 + t] = w[((size_t)(nt * 16 + n) * CD + 3 * kq + t) * 49 + tap];
}

// Training: rebuild the kernel's operands (table, packed dense + indicator weights, dense whitening) ON THE DEVICE from
// the current OIHW stem weight (inside the flat parameter buffer) and the current whitening tables sc/sh of the dense
// stem (new channel order; sc = 1/(div*std), sh = -mean/std) — both change every optimisation step.  Same arithmetic as
// the host-side construction in pnvo_load_weights (double, rounded to float once).
__global__ __launch_bounds__(256) void stem_dd_repack_kernel(const float *w, int cin, const float *sc_new, const float *sh_new,
                                                           const int *dense_ref, const int *dense_new, int nd,
                                                           const int *dd_ref, const int *dd_new, int bins, int slice,
                                                           float *table, float *wpk, float *sc12, float *sh12) {
  const int brows = bins + 1;
  const int n_tab = 49 * bins * 2 * 32, n_pk = 49 * NT16 * 64 * 3;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n_tab) {                                   // table[kh][kw][b][f][o] = W[o][ref(f,b)][tap] / std
    const int o = e & 31;
    int r = e >> 5;
    const int f = r & 1;
    r >>= 1;
    const int b = r % bins;
    const int tap = r / bins;
    const int kh = tap / 7, kw = tap - 7 * kh;
    const int k = f * bins + b;
    const double inv_std = (double)sc_new[dd_new[k]];            // div = 1 for the depth modalities
    table[(long)kh * slice + (((long)kw * brows + b) * 2 + f) * 32 + o] =
        (float)((double)w[((long)o * cin + dd_ref[k]) * 49 + tap] * inv_std);
  } else if (e < n_tab + n_pk) {                     // packed dense + indicator weights
    int r = e - n_tab;
    const int t = r % 3;
    r /= 3;
    const int lane = r & 63;
    r >>= 6;
    const int nt = r % NT16;
    const int tap = r / NT16;
    const int co = nt * 16 + (lane & 15), d = 3 * (lane >> 4) + t;
    float v = 0.f;
    if (d < nd) {
      v = w[((long)co * cin + dense_ref[d]) * 49 + tap];
    } else if (d == nd) {                            // "inside the image" indicator: -sum_c W * mean_c / std_c
      double ind = 0.0;
      for (int k = 0; k < 2 * bins; ++k)
        ind += (double)w[((long)co * cin + dd_ref[k]) * 49 + tap] * (double)sh_new[dd_new[k]];   // sh = -mean/std
      v = (float)ind;
    }
    wpk[e - n_tab] = v;
  } else if (e < n_tab + n_pk + CD) {
    const int d = e - n_tab - n_pk;
    sc12[d] = d < nd ? sc_new[dense_new[d]] : 0.f;
    sh12[d] = d < nd ? sh_new[dense_new[d]] : (d == nd ? 1.f : 0.f);
  }
}

hipError_t launch_stem_dd_repack(const float *w, int cin, const float *sc_new, const float *sh_new, const int *dense_ref,
                                 const int *dense_new, int nd, const int *dd_ref, const int *dd_new, int bins,
                                 float *table, float *wpk, float *sc12, float *sh12, hipStream_t s) {
  const int total = 49 * bins * 2 * 32 + 49 * NT16 * 64 * 3 + CD;
  hipLaunchKernelGGL(stem_dd_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, cin, sc_new, sh_new,
                     dense_ref, dense_new, nd, dd_ref, dd_new, bins, stem_dd_slice_floats(bins), table, wpk, sc12, sh12);
  return hipGetLastError();
}

hipError_t launch_stem_dd(const StemDDArgs &a, hipStream_t s) {
  StemDDArgs p = a;
  p.tiles_x = (a.Wo + TW - 1) / TW;
  p.tiles_y = (a.Ho + TH - 1) / TH;
  p.slice_floats = stem_dd_slice_floats(a.bins);
  if (!stem_dd_supported(a.bins)) return hipErrorInvalidValue;
  const size_t lds = (size_t)(2 * p.slice_floats + NPIX * CD) * 4 + (size_t)2 * PH * OW * 2;
  const long ntiles = (long)a.B * p.tiles_x * p.tiles_y;
  const long resident = (long)(160 * 1024 / lds) * 256;
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  if (a.dbg == 9)
    hipLaunchKernelGGL((stem_dd_kernel<10, true, 0>), grid, dim3(NTHREADS), lds, s, p);
  else if (a.dbg == 11)
    hipLaunchKernelGGL((stem_dd_kernel<10, false, 1>), grid, dim3(NTHREADS), lds, s, p);
  else if (a.dbg == 12)
    hipLaunchKernelGGL((stem_dd_kernel<10, false, 2>), grid, dim3(NTHREADS), lds, s, p);
  else
    hipLaunchKernelGGL((stem_dd_kernel<10, false, 0>), grid, dim3(NTHREADS), lds, s, p);
  return hipGetLastError();
}

}  // namespace pnvo
