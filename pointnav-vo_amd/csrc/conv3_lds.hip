// conv3_lds.hip — 3x3 stride-1 convolution of the residual stages with the input patch staged in LDS (gfx950 only).
//
// Same math, weight packing and epilogue contract as conv_mfma.hip (implicit GEMM on v_mfma_f32_32x32x2_f32, raw output
// + deterministic GroupNorm partial sums), but the A operand no longer comes from L2 per lane and per tap:
//   * a workgroup owns a TH x TW rectangle of ONE sample's output and stages the (TH+2) x (TW+2) input patch, 32 input
//     channels at a time, into LDS with fully coalesced 16-byte loads.  The producer's GroupNorm+ReLU (MODE 1:
//     relu(x*scale[n,c]+shift[n,c]), resnet.py:50-67 order conv -> GN -> ReLU) and the zero padding (applied AFTER it)
//     are done ONCE per patch element while staging instead of once per tap in the operand fetch (9x fewer VALU ops);
//   * a wave owns a 4x8 (or 8x4) block of output pixels = the 32 rows of the MFMA tile and NT x 32 output channels;
//     lane (i, h) reads 16 B = channels [8j+4h, 8j+4h+4) of its pixel for tap (kh,kw) with ONE ds_read_b128 per
//     4 MFMA k-steps; the LDS pixel pitch is 36 floats (and the patch row pitch odd for 8x4 blocks) so that the 8 lanes
//     of a b128 phase cover the 32 banks exactly once;
//   * weights stream from L2 in pack_conv_weight() order (one coalesced 1-KiB load per (tap, 8 channels, n-tile)),
//     prefetched two stages ahead; with fp32 MFMA (64 cycles per 32x32x2) both operand streams are tiny: per MFMA a
//     wave needs 256 B from LDS and 256 B from L1, so the matrix pipe is the only busy resource in the K loop;
//   * input channels beyond 32 are walked in chunks with the next chunk's global loads in flight during the MFMAs of
//     the current one (double-buffered LDS, one barrier per chunk).
// Statistics slot = (tile, wave): one writer per (sample, slot, channel), fixed summation order in gn_finalize.
#include <cstdlib>
#include <cstring>
#include <map>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr unsigned OOB = 0x80000000u;
constexpr int PP = 36;   // floats per pixel in LDS

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__host__ __device__ constexpr int patch_pitch(int blk, int pwr) { return (blk && (pwr % 2 == 0)) ? pwr + 1 : pwr; }
}  // namespace

// BLK 0: wave block = 4 rows x 8 cols (lane i -> x = i&7, y = i>>3); BLK 1: 8 rows x 4 cols (y = i&7, x = i>>3).
// WY x WX wave blocks per workgroup.  MODE 0: x holds final activations; MODE 1: relu(x*scale+shift) while staging.
// Workgroups are PERSISTENT: each walks work items (tile, n-tile group, channel chunk) with a grid stride; the global
// loads of item q+1 are in flight during the MFMAs of item q and land in the other LDS buffer (one barrier per item), so
// staging never idles the matrix pipe and there is no per-tile launch / drain.
template <int BLK, int WY, int WX, int NT, int MODE>
__global__ __launch_bounds__(64 * WY * WX) void conv3_lds_kernel(const ConvArgs p, int tiles_x, int tiles_y, int ngroups,
                                                                 int nwork) {
  constexpr int BH = BLK ? 8 : 4, BW = BLK ? 4 : 8;
  constexpr int TH = BH * WY, TW = BW * WX, PH = TH + 2, PWR = TW + 2, PWP = patch_pitch(BLK, PWR);
  constexpr int BUF = PH * PWP * PP;                 // floats per LDS buffer
  constexpr int NTHR = 64 * WY * WX;
  constexpr int ITEMS = PH * PWR * 8;                // (pixel, 16-byte slot) staging items per 32-channel chunk
  constexpr int NIT = (ITEMS + NTHR - 1) / NTHR;
  static_assert(NTHR % 8 == 0, "a thread keeps its 16-byte slot across staging passes");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int CIN = p.CIN, J = CIN >> 3, NCH = CIN >> 5;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const int g = tid & 7;

  // work item w (a tile of one sample for one group of NT n-tiles) -> coordinates; consecutive w = adjacent tiles
  struct Item {
    int n, y0, x0, ntg0, tile;
  };
  auto decode = [&](int w) -> Item {
    Item it;
    it.ntg0 = (w % ngroups) * NT;
    int t = w / ngroups;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    it.n = t / tiles_y;
    it.y0 = ty * TH;
    it.x0 = tx * TW;
    it.tile = ty * tiles_x + tx;
    return it;
  };

  // ---- staging: patch pixel/slot of this thread per pass is fixed; the image offset depends on the tile
  int ppr[NIT], ppc[NIT], loff[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int itx = tid + k * NTHR;
    const int pix = itx >> 3;
    ppr[k] = pix / PWR;
    ppc[k] = pix - ppr[k] * PWR;
    loff[k] = itx < ITEMS ? (ppr[k] * PWP + ppc[k]) * PP + 4 * g : -1;
  }
  auto gload = [&](const Item &it, int c, f32x4 (&v)[NIT], unsigned &inmask) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.x + (long)it.n * H * W * CIN), 0, (unsigned)((long)H * W * CIN * 4), 0x00020000);
    inmask = 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int yy = it.y0 - 1 + ppr[k], xx = it.x0 - 1 + ppc[k];
      const bool in = loff[k] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      inmask |= (in ? 1u : 0u) << k;
      v[k] = bload4(rx, in ? (unsigned)(((yy * W + xx) * CIN + 4 * g) * 4) : OOB, (unsigned)c * 128u);
    }
  };
  auto lstore = [&](const Item &it, int c, float *buf, f32x4 (&v)[NIT], unsigned inmask) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      sc = *reinterpret_cast<const f32x4 *>(p.in_scale + (long)it.n * CIN + 32 * c + 4 * g);
      sh = *reinterpret_cast<const f32x4 *>(p.in_shift + (long)it.n * CIN + 32 * c + 4 * g);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (loff[k] >= 0) {
        f32x4 t = v[k];
        if (MODE == 1) {
          const bool in = (inmask >> k) & 1u;        // zero padding is applied AFTER the producer's GroupNorm + ReLU
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = in ? fmaxf(__builtin_fmaf(t[e], sc[e], sh[e]), 0.f) : 0.f;
        }
        *reinterpret_cast<f32x4 *>(buf + loff[k]) = t;
      }
    }
  };

  // ---- this lane's pixel and operand addresses
  const int wy = wave / WX, wx = wave - wy * WX;
  const int ly = BLK ? (i & 7) : (i >> 3), lx = BLK ? (i >> 3) : (i & 7);
  const int lane_base = ((wy * BH + ly) * PWP + (wx * BW + lx)) * PP + 4 * h;
  const int T = 9, SJ = T * J;
  const unsigned w_nt_bytes = (unsigned)SJ * 1024u;
  const unsigned wlane = (unsigned)lane * 16u;

  // NT == 1: two accumulators (even / odd k-steps) so that consecutive MFMAs never share one — an instruction issued
  // between two MFMAs on the SAME accumulator costs ~43 cycles (MI355X_MICROARCH.md), on different ones ~6
  constexpr int NA = NT == 1 ? 2 : NT;
  f32x16 acc[NA];

  // one chunk = 36 stages (tap, 8-channel group): A one stage ahead (LDS), B two stages ahead (L2)
  auto compute = [&](int ntg0, int c, const float *L) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.wpk + (long)ntg0 * SJ * 256), 0, (unsigned)NT * w_nt_bytes, 0x00020000);
    const float *la = L + lane_base;
    auto lda = [&](int s) -> f32x4 {
      const int tap = s >> 2, j = s & 3;
      return *reinterpret_cast<const f32x4 *>(la + ((tap / 3) * PWP + tap % 3) * PP + 8 * j);
    };
    auto ldb = [&](int s, f32x4 (&b)[NT]) {
      const int tap = s >> 2, j = s & 3;
      const unsigned soff = (unsigned)((tap * J + 4 * c + j) * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = bload4(rw, wlane + (unsigned)nt * w_nt_bytes, soff);
    };
    f32x4 a[2], b[3][NT];
    ldb(0, b[0]);
    ldb(1, b[1]);
    a[0] = lda(0);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      if (s + 2 < 36) ldb(s + 2, b[(s + 2) % 3]);
      if (s + 1 < 36) a[(s + 1) & 1] = lda(s + 1);
      // pin the prefetch distance: without the fence the compiler sinks these loads to just before their first use (one
      // L2 round trip exposed per stage: 58 % instead of 75+ % MFMA-busy on the 64/128-channel stages)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int ai = NT == 1 ? (t & 1) : nt;
          acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][t], b[s % 3][nt][t], acc[ai], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of one work item: raw output + GroupNorm partials.
  //      C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  auto epilogue = [&](const Item &it) {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.y + (long)it.n * Ho * Wo * p.y_cstride), 0, (unsigned)((long)Ho * Wo * p.y_cstride * 4), 0x00020000);
    unsigned roff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ry_ = BLK ? (row & 7) : (row >> 3), rx_ = BLK ? (row >> 3) : (row & 7);
      const int oy = it.y0 + wy * BH + ry_, ox = it.x0 + wx * BW + rx_;
      roff[r] = (oy < Ho && ox < Wo) ? (unsigned)((oy * Wo + ox) * p.y_cstride) * 4u : OOB;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (it.ntg0 + nt) * 32 + i;
      const bool cvalid = co < p.y_cstride;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = roff[r] != OOB;
        const float v = ok ? (NT == 1 ? acc[0][r] + acc[1][r] : acc[nt][r]) : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry,
                                              (ok && cvalid) ? roff[r] + (unsigned)co * 4u : OOB, 0, 0);
        s1 += v;
        s2 = __builtin_fmaf(v, v, s2);
      }
      if (p.stats != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0) {
          const int slot = it.tile * (WY * WX) + wave;
          float *dst = p.stats + (((long)it.n * p.slots + slot) * p.COUTP + co) * 2;
          dst[0] = s1;
          dst[1] = s2;
        }
      }
    }
  };

  // ---- the pipeline over (work item, channel chunk)
  int w = blockIdx.x;
  if (w >= nwork) return;
  Item cur = decode(w);
  int c = 0, buf = 0;
  f32x4 sv[NIT];
  unsigned inm;
  gload(cur, 0, sv, inm);
  lstore(cur, 0, lds, sv, inm);
  __syncthreads();
  for (;;) {
    // what comes after (cur, c)?
    int nc = c + 1, nw = w;
    if (nc == NCH) {
      nc = 0;
      nw = w + gridDim.x;
    }
    const bool more = nw < nwork;
    Item nxt = cur;
    if (more && nw != w) nxt = decode(nw);
    if (more) gload(nxt, nc, sv, inm);               // in flight while the matrix cores work on (cur, c)
    if (c == 0) {
#pragma unroll
      for (int nt = 0; nt < NA; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    }
    compute(cur.ntg0, c, lds + buf * BUF);
    if (c == NCH - 1) epilogue(cur);
    if (!more) break;
    lstore(nxt, nc, lds + (buf ^ 1) * BUF, sv, inm);
    __syncthreads();
    buf ^= 1;
    cur = nxt;
    w = nw;
    c = nc;
  }
}

// ----------------------------------------------------------------------------------------------------------------------
// Wave-private variant: every wave stages the (BH+2) x (BW+2) patch of its OWN pixel block into its own slice of LDS and
// walks work items (pixel block(s), n-tile group) with a grid stride.  No workgroup barrier anywhere: the tile kernel above
// loses 25-30 % of the 64/128-channel stages to barrier waits (3-wave workgroups land unevenly on the 4 SIMDs and every
// item ends in a barrier; measured with s_memtime brackets).  LDS accesses of one wave execute in order, so a single
// patch buffer per wave is enough: the next chunk's global loads are parked in registers during the MFMAs and written
// after them.  Halo over-fetch is 60/32 pixels per block (from L2), nothing else changes.
template <int BLK, int NT, int MODE, int MT>
__global__ __launch_bounds__(256) void conv3_wave_kernel(const ConvArgs p, int blocks_x, int blocks_y, int ngroups,
                                                         int nwork) {
  constexpr int BH = BLK ? 8 : 4, BW = BLK ? 4 : 8;
  constexpr int PH = BH * MT + 2, PWR = BW + 2, PWP = patch_pitch(BLK, PWR);   // MT pixel blocks stacked vertically
  constexpr int BUFW = PH * PWP * PP;                // floats per wave
  constexpr int ITEMS = PH * PWR * 8;
  constexpr int NIT = (ITEMS + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *L = lds + wave * BUFW;
  const int i = lane & 31, h = lane >> 5;
  const int CIN = p.CIN, J = CIN >> 3, NCH = CIN >> 5;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
  const int g = lane & 7;

  struct Item {
    int n, y0, x0, ntg0, blk;
  };
  auto decode = [&](int w) -> Item {
    Item it;
    it.ntg0 = (w % ngroups) * NT;
    int t = w / ngroups;
    const int bx = t % blocks_x;
    t /= blocks_x;
    const int by = t % blocks_y;
    it.n = t / blocks_y;
    it.y0 = by * BH * MT;
    it.x0 = bx * BW;
    it.blk = by * blocks_x + bx;
    return it;
  };
  int wnext = (int)blockIdx.x * 4 + wave;            // static grid-stride walk (a shared atomic work counter was 2x
  auto next_item = [&]() -> int {                    // slower: its latency sits in the in-order vmcnt queue)
    const int w = wnext;
    wnext += (int)gridDim.x * 4;
    return w;
  };

  int ppr[NIT], ppc[NIT], loff[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int itx = lane + k * 64;
    const int pix = itx >> 3;
    ppr[k] = pix / PWR;
    ppc[k] = pix - ppr[k] * PWR;
    loff[k] = itx < ITEMS ? (ppr[k] * PWP + ppc[k]) * PP + 4 * g : -1;
  }
  auto gload = [&](const Item &it, int c, f32x4 (&v)[NIT], unsigned &inmask) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.x + (long)it.n * H * W * CIN), 0, (unsigned)((long)H * W * CIN * 4), 0x00020000);
    inmask = 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int yy = it.y0 - 1 + ppr[k], xx = it.x0 - 1 + ppc[k];
      const bool in = loff[k] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      inmask |= (in ? 1u : 0u) << k;
      v[k] = bload4(rx, in ? (unsigned)(((yy * W + xx) * CIN + 4 * g) * 4) : OOB, (unsigned)c * 128u);
    }
  };
  auto lstore = [&](const Item &it, int c, f32x4 (&v)[NIT], unsigned inmask) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      sc = *reinterpret_cast<const f32x4 *>(p.in_scale + (long)it.n * CIN + 32 * c + 4 * g);
      sh = *reinterpret_cast<const f32x4 *>(p.in_shift + (long)it.n * CIN + 32 * c + 4 * g);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (loff[k] >= 0) {
        f32x4 t = v[k];
        if (MODE == 1) {
          const bool in = (inmask >> k) & 1u;        // zero padding is applied AFTER the producer's GroupNorm + ReLU
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = in ? fmaxf(__builtin_fmaf(t[e], sc[e], sh[e]), 0.f) : 0.f;
        }
        *reinterpret_cast<f32x4 *>(L + loff[k]) = t;
      }
    }
  };

  const int ly = BLK ? (i & 7) : (i >> 3), lx = BLK ? (i >> 3) : (i & 7);
  const float *la = L + (ly * PWP + lx) * PP + 4 * h;
  const int SJ = 9 * J;
  const unsigned w_nt_bytes = (unsigned)SJ * 1024u;
  const unsigned wlane = (unsigned)lane * 16u;
  constexpr int NA = NT == 1 ? 2 : NT;
  f32x16 acc[MT][NA];

  auto compute = [&](int ntg0, int c) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.wpk + (long)ntg0 * SJ * 256), 0, (unsigned)NT * w_nt_bytes, 0x00020000);
    auto lda = [&](int s, f32x4 (&av)[MT]) {
      const int tap = s >> 2, j = s & 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        av[mt] = *reinterpret_cast<const f32x4 *>(la + ((tap / 3 + mt * BH) * PWP + tap % 3) * PP + 8 * j);
    };
    auto ldb = [&](int s, f32x4 (&b)[NT]) {
      const int tap = s >> 2, j = s & 3;
      const unsigned soff = (unsigned)((tap * J + 4 * c + j) * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = bload4(rw, wlane + (unsigned)nt * w_nt_bytes, soff);
    };
    f32x4 a[2][MT], b[3][NT];
    ldb(0, b[0]);
    ldb(1, b[1]);
    lda(0, a[0]);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      if (s + 2 < 36) ldb(s + 2, b[(s + 2) % 3]);
      if (s + 1 < 36) lda(s + 1, a[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);             // keep the prefetch distance
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int ai = NT == 1 ? (t & 1) : nt;
            acc[mt][ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][mt][t], b[s % 3][nt][t], acc[mt][ai], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto epilogue = [&](const Item &it) {
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.y + (long)it.n * Ho * Wo * p.y_cstride), 0, (unsigned)((long)Ho * Wo * p.y_cstride * 4), 0x00020000);
    unsigned roff[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ry_ = BLK ? (row & 7) : (row >> 3), rx_ = BLK ? (row >> 3) : (row & 7);
        const int oy = it.y0 + mt * BH + ry_, ox = it.x0 + rx_;
        roff[mt][r] = (oy < Ho && ox < Wo) ? (unsigned)((oy * Wo + ox) * p.y_cstride) * 4u : OOB;
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (it.ntg0 + nt) * 32 + i;
      const bool cvalid = co < p.y_cstride;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = roff[mt][r] != OOB;
          const float v = ok ? (NT == 1 ? acc[mt][0][r] + acc[mt][1][r] : acc[mt][nt][r]) : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry,
                                                (ok && cvalid) ? roff[mt][r] + (unsigned)co * 4u : OOB, 0, 0);
          s1 += v;
          s2 = __builtin_fmaf(v, v, s2);
        }
      if (p.stats != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0) {
          float *dst = p.stats + (((long)it.n * p.slots + it.blk) * p.COUTP + co) * 2;
          dst[0] = s1;
          dst[1] = s2;
        }
      }
    }
  };

  int w = next_item();
  if (w >= nwork) return;
  Item cur = decode(w);
  int c = 0;
  f32x4 sv[NIT];
  unsigned inm;
  gload(cur, 0, sv, inm);
  lstore(cur, 0, sv, inm);
  int wn = next_item();                              // the item after `cur`, fetched one item ahead
  for (;;) {
    int nc = c + 1;
    bool more = true;
    Item nxt = cur;
    if (nc == NCH) {
      nc = 0;
      more = wn < nwork;
      if (more) nxt = decode(wn);
    }
    if (more) gload(nxt, nc, sv, inm);               // in flight during the MFMAs below
    if (c == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NA; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    }
    compute(cur.ntg0, c);
    if (c == NCH - 1) epilogue(cur);
    if (!more) break;
    lstore(nxt, nc, sv, inm);                        // this wave's LDS reads above are older: in-order, no barrier
    if (nc == 0) {
      cur = nxt;
      wn = next_item();
    }
    c = nc;
  }
}

namespace {
struct Cfg {
  int blk, wy, wx;
};
constexpr Cfg CFGS[3] = {{0, 4, 1}, {1, 3, 1}, {0, 3, 1}};

inline void cfg_tiles(const Cfg &c, int Ho, int Wo, int *tx, int *ty) {
  const int th = (c.blk ? 8 : 4) * c.wy, tw = (c.blk ? 4 : 8) * c.wx;
  *tx = (Wo + tw - 1) / tw;
  *ty = (Ho + th - 1) / th;
}
inline double cfg_waste(const Cfg &c, int Ho, int Wo) {
  int tx, ty;
  cfg_tiles(c, Ho, Wo, &tx, &ty);
  const int th = (c.blk ? 8 : 4) * c.wy, tw = (c.blk ? 4 : 8) * c.wx;
  return (double)tx * tw * ty * th / ((double)Ho * Wo);
}
inline int pick_cfg(int Ho, int Wo) {
  int best = 0;
  for (int k = 1; k < 3; ++k)
    if (cfg_waste(CFGS[k], Ho, Wo) < cfg_waste(CFGS[best], Ho, Wo) - 1e-9) best = k;
  return best;
}

template <int BLK, int WY, int WX, int NT>
hipError_t launch_cfg(const ConvArgs &a, int tiles_x, int tiles_y, hipStream_t s) {
  constexpr int BH = BLK ? 8 : 4, BW = BLK ? 4 : 8;
  constexpr int PH = BH * WY + 2, PWR = BW * WX + 2, PWP = patch_pitch(BLK, PWR);
  const size_t lds = (size_t)PH * PWP * PP * 4 * 2;
  const int ngroups = a.COUTP / 32 / NT;
  const long nwork = (long)a.B * tiles_x * tiles_y * ngroups;
  // persistent grid: as many workgroups as fit (LDS-limited), rounded so that every workgroup gets the same item count
  long resident = 256L * (long)((160 * 1024) / lds);
  static int cap = -1;
  if (cap < 0) {
    const char *e = std::getenv("PNVO_CONV3_WGS");     // experiment knob: workgroups per CU
    cap = e ? std::atoi(e) : 0;
  }
  if (cap > 0) resident = 256L * cap;
  const long rounds = (nwork + resident - 1) / resident;
  const long grid_x = (nwork + rounds - 1) / rounds;
  dim3 grid((unsigned)grid_x);
  if (a.in_scale != nullptr)
    hipLaunchKernelGGL((conv3_lds_kernel<BLK, WY, WX, NT, 1>), grid, dim3(64 * WY * WX), lds, s, a, tiles_x, tiles_y,
                       ngroups, (int)nwork);
  else
    hipLaunchKernelGGL((conv3_lds_kernel<BLK, WY, WX, NT, 0>), grid, dim3(64 * WY * WX), lds, s, a, tiles_x, tiles_y,
                       ngroups, (int)nwork);
  return hipGetLastError();
}
}  // namespace

namespace {
inline int wave_blk(int Ho, int Wo, double *waste) {   // block shape (0: 4x8, 1: 8x4) with the least padding
  auto w = [&](int bh, int bw) { return (double)((Ho + bh - 1) / bh * bh) * ((Wo + bw - 1) / bw * bw) / ((double)Ho * Wo); };
  const double w0 = w(4, 8), w1 = w(8, 4);
  if (waste) *waste = w0 <= w1 ? w0 : w1;
  return w0 <= w1 ? 0 : 1;
}

template <int BLK, int NT, int MT>
hipError_t launch_wave(const ConvArgs &a, hipStream_t s) {
  constexpr int BH = (BLK ? 8 : 4) * MT, BW = BLK ? 4 : 8;
  constexpr int PH = BH + 2, PWR = BW + 2, PWP = patch_pitch(BLK, PWR);
  const size_t lds = (size_t)4 * PH * PWP * PP * 4;
  const int bx = (a.Wo + BW - 1) / BW, by = (a.Ho + BH - 1) / BH;
  const int ngroups = a.COUTP / 32 / NT;
  const long nwork = (long)a.B * bx * by * ngroups;
  long wgs = 256L * (long)((160 * 1024) / lds);
  // 3 workgroups per CU (12 waves): measured 112 vs 103 TFLOP/s against 4 on the 64-channel stage (8448 items: with 4096
  // waves a quarter of the SIMDs end with one wave running a third item alone), equal on the 128-channel stage.
  static const int cap = std::getenv("PNVO_WAVE_WGS") ? std::atoi(std::getenv("PNVO_WAVE_WGS")) : 3;
  if (cap > 0 && wgs > 256L * cap) wgs = 256L * cap;
  if (wgs * 4 > nwork) wgs = (nwork + 3) / 4;
  if (a.in_scale != nullptr)
    hipLaunchKernelGGL((conv3_wave_kernel<BLK, NT, 1, MT>), dim3((unsigned)wgs), dim3(256), lds, s, a, bx, by, ngroups,
                       (int)nwork);
  else
    hipLaunchKernelGGL((conv3_wave_kernel<BLK, NT, 0, MT>), dim3((unsigned)wgs), dim3(256), lds, s, a, bx, by, ngroups,
                       (int)nwork);
  return hipGetLastError();
}

// 32-channel layers carry few MFMAs per block: stack two 4x8 blocks per wave item when the height allows it.
int wave_mt(const ConvArgs &a, int blk) { return (blk == 0 && a.CIN == 32 && a.COUTP == 32 && a.Ho % 8 == 0) ? 2 : 1; }

// Which LDS-staged kernel?  The wave-private one wins where an item carries enough MFMAs to amortise its private staging
// (>= 64 input channels: 105-112 vs 87-93 TFLOP/s); the 32-channel stage keeps the workgroup-tile kernel (100 vs 85).
// PNVO_CONV3 = tile | wave forces one of them.
bool use_wave_kernel(const ConvArgs &a) {
  static const int v = [] {
    const char *e = std::getenv("PNVO_CONV3");
    return !e ? -1 : (std::strcmp(e, "tile") == 0 ? 0 : (std::strcmp(e, "wave") == 0 ? 1 : -1));
  }();
  if (v >= 0) return v != 0;
  return a.CIN >= 64;
}
}  // namespace

// Can this layer run on an LDS-staged kernel?  (3x3, stride 1, pad 1, 32-channel multiples, per-sample image big
// enough that the pixel blocks waste < 15 % of the MFMA rows, offsets within 32 bits.)
bool conv3_lds_supported(const ConvArgs &a) {
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.up == 2 || a.accum || a.src_mode || a.bias != nullptr ||
      a.relu_out)
    return false;
  if (a.CIN % 32 != 0 || a.COUTP % 32 != 0 || a.H != a.Ho || a.W != a.Wo) return false;
  if ((long)a.H * a.W * a.CIN * 4 >= 0x7FFFF000L || (long)a.Ho * a.Wo * a.y_cstride * 4 >= 0x7FFFF000L) return false;
  if (use_wave_kernel(a)) {
    double waste;
    (void)wave_blk(a.Ho, a.Wo, &waste);
    return waste < 1.15;
  }
  return cfg_waste(CFGS[pick_cfg(a.Ho, a.Wo)], a.Ho, a.Wo) < 1.15;
}

int conv3_lds_slots(const ConvArgs &a) {
  if (use_wave_kernel(a)) {
    const int blk = wave_blk(a.Ho, a.Wo, nullptr);
    const int bh = (blk ? 8 : 4) * wave_mt(a, blk), bw = blk ? 4 : 8;
    return ((a.Ho + bh - 1) / bh) * ((a.Wo + bw - 1) / bw);
  }
  const Cfg &c = CFGS[pick_cfg(a.Ho, a.Wo)];
  int tx, ty;
  cfg_tiles(c, a.Ho, a.Wo, &tx, &ty);
  return tx * ty * c.wy * c.wx;
}

// nt: output n-tiles (32 channels) per wave, 1 or 2 (a.COUTP / 32 must be divisible by it).
hipError_t launch_conv3_lds(const ConvArgs &a, int nt, hipStream_t s) {
  if (!conv3_lds_supported(a) || (nt != 1 && nt != 2) || (a.COUTP / 32) % nt != 0) return hipErrorInvalidValue;
  if (use_wave_kernel(a)) {
    // one output-channel tile per item: twice the items of half the size divide more evenly over the 3072 waves (stage 3:
    // 9216 items = exactly 3 each) — 115-119 vs 109-112 TFLOP/s with two (PNVO_WAVE_NT=2)
    static const int wnt = std::getenv("PNVO_WAVE_NT") ? std::atoi(std::getenv("PNVO_WAVE_NT")) : 1;
    if (wnt == 1) nt = 1;
    const int blk = wave_blk(a.Ho, a.Wo, nullptr);
    const int mt = wave_mt(a, blk);
    switch (blk * 100 + nt * 10 + mt) {
      case 11: return launch_wave<0, 1, 1>(a, s);
      case 12: return launch_wave<0, 1, 2>(a, s);
      case 21: return launch_wave<0, 2, 1>(a, s);
      case 111: return launch_wave<1, 1, 1>(a, s);
      default: return launch_wave<1, 2, 1>(a, s);
    }
  }
  const int k = pick_cfg(a.Ho, a.Wo);
  int tx, ty;
  cfg_tiles(CFGS[k], a.Ho, a.Wo, &tx, &ty);
  switch (k * 10 + nt) {
    case 1: return launch_cfg<0, 4, 1, 1>(a, tx, ty, s);
    case 2: return launch_cfg<0, 4, 1, 2>(a, tx, ty, s);
    case 11: return launch_cfg<1, 3, 1, 1>(a, tx, ty, s);
    case 12: return launch_cfg<1, 3, 1, 2>(a, tx, ty, s);
    case 21: return launch_cfg<0, 3, 1, 1>(a, tx, ty, s);
    case 22: return launch_cfg<0, 3, 1, 2>(a, tx, ty, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pnvo
