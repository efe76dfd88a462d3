// train_kernels.hip — kernels of the VO training step (BASELINE config 4), gfx950 only.
//   weight gradient of every conv / linear on the fp32 matrix cores (wgrad_kernel + fixed-order reduction),
//   GroupNorm backward (reduce / finalize / apply), ReLU masks, max-pool with argmax and its backward, bias gradients,
//   squared-error loss, Adam, device-side re-packing of the kernel operands after an optimiser step, and the per-channel
//   moments behind RunningMeanAndVar's train-mode update.
// Backward-DATA of the convs reuses conv_mfma_kernel (flipped/transposed packed weights, `up` = forward stride).
// Reductions have one writer per partial and a fixed summation order: gradients are bit-reproducible.
#include <cstdlib>
#include <cstring>

#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PNVO_OOB 0x80000000u

__device__ __forceinline__ unsigned clampb(long bytes) {
  return (unsigned)(bytes > 0x7FFFF000L ? 0x7FFFF000L : (bytes < 0 ? 0 : bytes));
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// ------------------------------------------------------------------------------------------------------------------
// dW[co][ci][kh][kw] = sum over pixels m=(n,ho,wo) of dY[m][co] * Xin[n, ho*s-p+kh, wo*s-p+kw, ci]
// as D[i = ci][j = co] += sum_k A[i][k = pixel] * B[k][j] with v_mfma_f32_32x32x2_f32 (2 pixels per instruction):
// lane (i, h) feeds ONE dword per operand: A = Xin[pixel(2s+h) shifted by the tap][ci0+i], B = dY[pixel(2s+h)][co0+i]
// (32 consecutive channels of one pixel = one coalesced 128-B segment per half wave).  A wave owns one
// (ci-tile, co-tile, tap-group, pixel-chunk) unit and keeps TG accumulators (one per tap of its group); partial results
// go to `partial[unit]` and are summed over the pixel chunks in a fixed order by wgrad_reduce_kernel.
// Xin modes: 0 plain tensor, 1 relu(x*scale[n,c]+shift[n,c]) (the producer's GroupNorm+ReLU, recomputed),
//            2 gathered from the observation tensors and whitened (the stem's input).
template <int TG, int MODE>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long unit = (long)blockIdx.x * 4 + wave;
  const long nunits = (long)p.chunks * p.pairs * p.groups;
  if (unit >= nunits) return;
  const int chunk = (int)(unit % p.chunks);
  long rest = unit / p.chunks;
  const int pair = (int)(rest % p.pairs);
  const int grp = (int)(rest / p.pairs);
  const int cit = pair % p.ci_tiles, cot = pair / p.ci_tiles;
  const int ci = cit * 32 + i, co = cot * 32 + i;
  const long P = (long)p.Ho * p.Wo, M = (long)p.B * P;
  const long m0 = (long)chunk * p.pix_per_chunk;
  long m1 = m0 + p.pix_per_chunk;
  if (m1 > M) m1 = M;
  const int T = p.KH * p.KW;
  const int t0 = grp * TG;

  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void *)p.dy, 0, clampb(M * p.DYC * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void *)p.x, 0, clampb((long)p.B * p.H * p.W * p.CIN * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(
      (void *)p.in_scale, 0, clampb((long)p.B * p.CIN * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(
      (void *)p.in_shift, 0, clampb((long)p.B * p.CIN * 4), 0x00020000);

  // MODE 2: this lane's input channel lives in one of the observation tensors
  const float *sbase = nullptr;
  int snch = 0, schoff = 0;
  float ssc = 0.f, ssh = 0.f;
  if (MODE == 2) {
    sbase = p.src[i].base;
    snch = p.src[i].nch;
    schoff = p.src[i].choff;
    ssc = p.in_scale ? p.in_scale[i] : p.src[i].sc;     // whitening x*sc+sh of this lane's channel (device table)
    ssh = p.in_shift ? p.in_shift[i] : p.src[i].sh;
  }

  f32x16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // this lane's pixel m = m0 + 2s + h, tracked as (n, ho, wo)
  long m = m0 + h;
  int n = (int)(m / P);
  int rem = (int)(m - (long)n * P);
  int ho = rem / p.Wo, wo = rem - ho * p.Wo;

  struct Stage {
    float b;
    float a[TG];
    bool ok[TG];
  };
  auto fetch = [&](Stage &st) {
    const bool vm = m < m1;
    st.b = bload(rdy, (vm && co < p.DYC) ? (unsigned)((m * p.DYC + co) * 4) : PNVO_OOB);
    float sc = 1.f, sh = 0.f;
    if (MODE == 1) {
      sc = bload(rsc, (vm && ci < p.CIN) ? (unsigned)(((long)n * p.CIN + ci) * 4) : PNVO_OOB);
      sh = bload(rsh, (vm && ci < p.CIN) ? (unsigned)(((long)n * p.CIN + ci) * 4) : PNVO_OOB);
    }
    // this pixel's window origin; the tap only adds wave-uniform constants
    const int hb = ho * p.stride - p.pad, wb = wo * p.stride - p.pad;
    const long pix0 = ((long)n * p.H + hb) * p.W + wb;
    const bool cok = vm && ci < p.CIN;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const int tap = t0 + t;                       // wave-uniform
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      const bool ok = vm && tap < T && (unsigned)(hb + kh) < (unsigned)p.H && (unsigned)(wb + kw) < (unsigned)p.W;
      st.ok[t] = ok;
      const long pix = pix0 + (long)kh * p.W + kw;
      float v;
      if (MODE == 2) {
        const float *addr = (ok && sbase != nullptr) ? sbase + pix * snch + schoff : p.zero_page;
        v = __builtin_fmaf(*addr, ssc, ssh);
      } else {
        v = bload(rx, (ok && cok) ? (unsigned)((pix * p.CIN + ci) * 4) : PNVO_OOB);
        if (MODE == 1) v = fmaxf(__builtin_fmaf(v, sc, sh), 0.f);
      }
      st.a[t] = v;
    }
    // advance this lane by 2 pixels
    m += 2;
    wo += 2;
    while (wo >= p.Wo) {
      wo -= p.Wo;
      ++ho;
    }
    while (ho >= p.Ho) {
      ho -= p.Ho;
      ++n;
    }
  };
  auto compute = [&](Stage &st) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const float a = st.ok[t] ? st.a[t] : 0.f;   // zero padding AFTER the input transform
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, st.b, acc[t], 0, 0, 0);
    }
  };

  const long steps = (m1 - m0 + 1) / 2;
  Stage s0, s1;
  if (steps > 0) fetch(s0);
  long s = 0;
  for (; s + 2 <= steps - 1; s += 2) {
    fetch(s1);
    compute(s0);
    fetch(s0);
    compute(s1);
  }
  if (s + 1 <= steps - 1) {
    fetch(s1);
    compute(s0);
    compute(s1);
  } else if (steps > 0) {
    compute(s0);
  }

  // C/D layout: col j (= co) = lane&31, row i (= ci) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float *dst = p.partial + ((unit * TG) * 32) * 32;
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[((long)t * 32 + row) * 32 + i] = acc[t][r];
    }
}


// ------------------------------------------------------------------------------------------------------------------
// LDS-staged weight gradient of the 3x3 stride-1 convs (every residual-stage conv but the three strided ones):
//   dW[tap][ci][co] = sum_pix X[pix + tap][ci] * dY[pix][co]
// A persistent workgroup owns one (ci-tile, co-tile) pair and a chunk of spatial tiles (TH x TW output pixels of one
// sample).  Per tile it stages the (TH+2) x (TW+2) input patch (32 channels; the producer's GroupNorm+ReLU and the zero
// padding applied while staging) and the TH x TW slab of dY (32 channels) into LDS with coalesced 16-byte loads; its 9
// waves are the 9 taps: wave (kh,kw) runs D[ci][co] += X[pix+(kh,kw)][ci] * dY[pix][co] over the tile's pixels, two
// pixels per v_mfma_f32_32x32x2_f32, both operands one conflict-free ds_read_b32 (32 consecutive channels per half
// wave).  The accumulators live in registers across all tiles of the chunk; the per-chunk partials are summed in a
// fixed order by wgrad_reduce_kernel (same layout as wgrad_kernel with TG = 9, one tap group).
template <int MODE, int S>                // S: stride of the forward conv (1 or 2); patch = (T - 1) * S + 3 input pixels per axis
__global__ __launch_bounds__(576) void wgrad3_lds_kernel(const WgradArgs p) {
  constexpr int PP = 36;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int TH = p.TH, TW = p.TW, PH = (TH - 1) * S + 3, PWR = (TW - 1) * S + 3;
  float *xs = lds;                         // [PH*PWR][PP]
  float *ds = lds + PH * PWR * PP;         // [TH*TW][PP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = tap
  const int i = lane & 31, h = lane >> 5;
  const int kh = wave / 3, kw = wave - 3 * kh;
  const int unit = blockIdx.x;
  const int pair = unit / p.chunks, chunk = unit - pair * p.chunks;
  const int cit = pair % p.ci_tiles, cot = pair / p.ci_tiles;
  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  const int g = tid & 7;                   // 576 % 8 == 0: a thread keeps its 16-byte slot

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;

  // staging through registers: the global loads of tile t+1 are in flight during the MFMAs of tile t
  constexpr int NX = 4, ND = 3;            // 16-byte items per thread: patch <= 10x26 pixels, slab <= 8x24 (wgrad_plan)
  f32x4 vx[NX], vd[ND];
  auto gload = [&](int t) {
    int q = t;
    const int tx = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty = q % p.tiles_y;
    const int n = q / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;      // output-tile origin; the input patch starts at (y0 * S - 1, x0 * S - 1)
    const float *xb = p.x + ((long)n * p.H * p.W) * p.CIN + cit * 32 + 4 * g;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      sc = *reinterpret_cast<const f32x4 *>(p.in_scale + (long)n * p.CIN + cit * 32 + 4 * g);
      sh = *reinterpret_cast<const f32x4 *>(p.in_shift + (long)n * p.CIN + cit * 32 + 4 * g);
    }
    bool inx[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int pix = (tid + k * 576) >> 3;
      const int pr = pix / PWR, pc = pix - pr * PWR;
      const int yy = y0 * S - 1 + pr, xx = x0 * S - 1 + pc;
      inx[k] = pix < PH * PWR && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      vx[k] = inx[k] ? *reinterpret_cast<const f32x4 *>(xb + ((long)yy * p.W + xx) * p.CIN) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float *db = p.dy + ((long)n * p.Ho * p.Wo) * p.DYC + cot * 32 + 4 * g;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int pix = (tid + k * 576) >> 3;
      const int qy = pix / TW, qx = pix - qy * TW;
      const int oy = y0 + qy, ox = x0 + qx;
      const bool in = pix < TH * TW && oy < p.Ho && ox < p.Wo;       // pixels outside the image contribute nothing
      vd[k] = in ? *reinterpret_cast<const f32x4 *>(db + ((long)oy * p.Wo + ox) * p.DYC) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (MODE == 1) {                        // zero padding AFTER the input transform
#pragma unroll
      for (int k = 0; k < NX; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) vx[k][e] = inx[k] ? fmaxf(__builtin_fmaf(vx[k][e], sc[e], sh[e]), 0.f) : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int pix = (tid + k * 576) >> 3;
      if (pix < PH * PWR) *reinterpret_cast<f32x4 *>(xs + pix * PP + 4 * g) = vx[k];
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int pix = (tid + k * 576) >> 3;
      if (pix < TH * TW) *reinterpret_cast<f32x4 *>(ds + pix * PP + 4 * g) = vd[k];
    }
  };

  int t0 = chunk * p.tiles_per_chunk, t1 = t0 + p.tiles_per_chunk;
  if (t1 > ntiles) t1 = ntiles;
  const float *xa = xs + (kh * PWR + kw + h * S) * PP + i;
  const float *da = ds + h * PP + i;
  const int half_w = TW >> 1;
  if (t0 < t1) gload(t0);
  for (int t = t0; t < t1; ++t) {
    __syncthreads();                       // the previous tile's readers are done
    lstore();
    __syncthreads();
    if (t + 1 < t1) gload(t + 1);
    // K loop of this tap: pixel pairs (2s, 2s+1) of each tile row; lane half h takes pixel 2s+h
    for (int qy = 0; qy < TH; ++qy) {
      const float *xr = xa + qy * S * PWR * PP, *dr = da + qy * TW * PP;
      int sx = 0;
      for (; sx + 2 <= half_w; sx += 2) {
        const float a0 = xr[(2 * sx) * S * PP], b0 = dr[(2 * sx) * PP];
        const float a1 = xr[(2 * sx + 2) * S * PP], b1 = dr[(2 * sx + 2) * PP];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1, 0, 0, 0);
      }
      if (sx < half_w) {
        const float a0 = xr[(2 * sx) * S * PP], b0 = dr[(2 * sx) * PP];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
      }
    }
  }
  // C/D layout: col j (= co) = lane&31, row i (= ci) = (r&3) + 8*(r>>2) + 4*(lane>>5); fixed order acc0 + acc1
  float *dst = p.partial + (((long)unit * 9 + wave) * 32) * 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    dst[(long)row * 32 + i] = acc0[r] + acc1[r];
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Twelve-wave version of the kernel above (the default for the 3x3 convs): three waves per SIMD instead of 3/2/2/2.
// A workgroup owns FOUR units of (ci-tile, co-tile, row group of the tile) — cs x os x rs = 4, chosen by wgrad_plan from
// the layer's tile counts — and wave (kh, u) runs the three taps (kh, 0..2) of unit u: one dY fragment feeds three MFMAs,
// and with lane half h on pixel s + h*TW/2 the input fragments slide along the row (stride 1: one new ds_read_b32 per
// three MFMAs).  The patch / slab of tile t+1 is written to the second LDS buffer while tile t is multiplied (one
// barrier per tile; single buffer + two barriers when 2 buffers exceed 160 KB).  Row groups are summed inside the
// workgroup; partials: unit = pair * chunks + chunk, summed by wgrad_reduce_kernel in the same fixed order as before.
template <int MODE, int S>
__global__ __launch_bounds__(768) void wgrad3_mw_kernel(const WgradArgs p) {
  constexpr int PP = 36;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int TH = p.TH, TW = p.TW, PH = (TH - 1) * S + 3, PWR = (TW - 1) * S + 3;
  const int CS = p.cs, OS = p.os, RS = p.rs;
  const int npx = PH * PWR, npd = TH * TW;
  const int xplane = npx * PP, dplane = npd * PP;
  const int bufsz = CS * xplane + OS * dplane;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int kh = wave % 3, u = wave / 3;
  const int ci_sub = u % CS, co_sub = (u / CS) % OS, row_sub = u / (CS * OS);
  const int pg = blockIdx.x / p.wg_chunks, chunk = blockIdx.x - pg * p.wg_chunks;
  const int cigs = p.ci_tiles / CS;
  const int cig = pg % cigs, cog = pg / cigs;
  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  const int g = tid & 7;                   // 768 % 8 == 0: a thread keeps its 16-byte channel slot

  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  // staging items (16 bytes): X patch planes first, then dY slab planes; per item: LDS float offset | row | col | plane
  constexpr int NX = 5, ND = 2;            // wgrad_plan keeps the item counts within these
  int xl[NX], xrc[NX], dl[ND], drc[ND];
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int q = (tid + k * 768) >> 3;
    const int plane = q / npx, pix = q - plane * npx;
    const int pr = pix / PWR, pc = pix - pr * PWR;
    xl[k] = plane < CS ? plane * xplane + pix * PP + 4 * g : -1;
    xrc[k] = pr | (pc << 8) | (plane << 16);
  }
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    const int q = (tid + k * 768) >> 3;
    const int plane = q / npd, pix = q - plane * npd;
    const int qy = pix / TW, qx = pix - qy * TW;
    dl[k] = plane < OS ? CS * xplane + plane * dplane + pix * PP + 4 * g : -1;
    drc[k] = qy | (qx << 8) | (plane << 16);
  }
  f32x4 vx[NX], vd[ND];
  f32x4 scp[2], shp[2];                    // MODE 1: scale / shift of this thread's four channels in each ci plane
  unsigned inmask = 0;                     // MODE 1: in-image flags of the patch items (zero padding AFTER the transform)
  auto gload = [&](int t) {
    int q = t;
    const int tx = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty = q % p.tiles_y;
    const int n = q / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;      // output-tile origin; the input patch starts at (y0 * S - 1, x0 * S - 1)
    const float *xb = p.x + ((long)n * p.H * p.W) * p.CIN + cig * CS * 32 + 4 * g;
    inmask = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int pr = xrc[k] & 255, pc = (xrc[k] >> 8) & 255, plane = xrc[k] >> 16;
      const int yy = y0 * S - 1 + pr, xx = x0 * S - 1 + pc;
      const bool in = xl[k] >= 0 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      vx[k] = in ? *reinterpret_cast<const f32x4 *>(xb + ((long)yy * p.W + xx) * p.CIN + plane * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
      inmask |= (in ? 1u : 0u) << k;
    }
    if (MODE == 1) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const long so = (long)n * p.CIN + (cig * CS + (pl < CS ? pl : 0)) * 32 + 4 * g;
        scp[pl] = *reinterpret_cast<const f32x4 *>(p.in_scale + so);
        shp[pl] = *reinterpret_cast<const f32x4 *>(p.in_shift + so);
      }
    }
    const float *db = p.dy + ((long)n * p.Ho * p.Wo) * p.DYC + cog * OS * 32 + 4 * g;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const int qy = drc[k] & 255, qx = (drc[k] >> 8) & 255, plane = drc[k] >> 16;
      const int oy = y0 + qy, ox = x0 + qx;
      const bool in = dl[k] >= 0 && oy < p.Ho && ox < p.Wo;          // pixels outside the image contribute nothing
      vd[k] = in ? *reinterpret_cast<const f32x4 *>(db + ((long)oy * p.Wo + ox) * p.DYC + plane * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](float *buf) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
      if (xl[k] >= 0) {
        f32x4 v = vx[k];
        if (MODE == 1) {
          const bool pl1 = (xrc[k] >> 16) != 0, in = (inmask >> k) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = in ? fmaxf(__builtin_fmaf(v[e], pl1 ? scp[1][e] : scp[0][e], pl1 ? shp[1][e] : shp[0][e]), 0.f) : 0.f;
        }
        *reinterpret_cast<f32x4 *>(buf + xl[k]) = v;
      }
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if (dl[k] >= 0) *reinterpret_cast<f32x4 *>(buf + dl[k]) = vd[k];
  };

  int t0 = chunk * p.tiles_per_chunk, t1 = t0 + p.tiles_per_chunk;
  if (t1 > ntiles) t1 = ntiles;
  const int half_w = TW >> 1;
  const int rpw = (TH + RS - 1) / RS;
  const int r0 = row_sub * rpw, r1 = min(TH, r0 + rpw);
  const int xo = ci_sub * xplane + (kh * PWR + h * half_w * S) * PP + i;
  const int dofs = CS * xplane + co_sub * dplane + h * half_w * PP + i;
  const bool two = p.nbuf == 2;
  int cur = 0;
  if (t0 < t1) {
    gload(t0);
    lstore(lds);
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const float *buf = lds + cur * bufsz;
    if (t + 1 < t1) gload(t + 1);
    for (int qy = r0; qy < r1; ++qy) {
      const float *xr = buf + xo + qy * S * PWR * PP, *dr = buf + dofs + qy * TW * PP;
      if (S == 1) {
        float a0 = xr[0], a1 = xr[PP];
        for (int sx = 0; sx < half_w; ++sx) {
          const float a2 = xr[(sx + 2) * PP], b = dr[sx * PP];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc[2], 0, 0, 0);
          a0 = a1;
          a1 = a2;
        }
      } else {
        float a0 = xr[0];
        for (int sx = 0; sx < half_w; ++sx) {
          const float a1 = xr[(2 * sx + 1) * PP], a2 = xr[(2 * sx + 2) * PP], b = dr[sx * PP];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc[2], 0, 0, 0);
          a0 = a2;
        }
      }
    }
    if (t + 1 < t1) {
      if (!two) __syncthreads();           // single buffer: every wave is done reading it
      lstore(lds + (two ? (cur ^ 1) * bufsz : 0));
      if (two) cur ^= 1;
    }
    __syncthreads();
  }
  // row groups of one (ci, co) pair are summed here, in a fixed order, through the (now idle) staging buffers
  if (RS > 1) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      if (kw > 0) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) lds[(wave * 16 + r) * 64 + lane] = acc[kw][r];
      __syncthreads();
      if (row_sub == 0)
        for (int k = 1; k < RS; ++k) {
          const int w2 = wave + 3 * CS * OS * k;          // same kh / ci / co, row group k
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[kw][r] += lds[(w2 * 16 + r) * 64 + lane];
        }
    }
    if (row_sub != 0) return;
  }
  // C/D layout: col j (= co) = lane&31, row i (= ci) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int pair = (cog * OS + co_sub) * p.ci_tiles + cig * CS + ci_sub;
  const long unit = (long)pair * p.chunks + chunk;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    float *dst = p.partial + ((unit * 9 + kh * 3 + kw) * 32) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[(long)row * 32 + i] = acc[kw][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-staged weight gradient of the 7x7 stride-2 stem (mode 2: input gathered from the observation tensors and whitened):
//   dW[kh][kw][c][co] = sum_pix X[2*pix + (kh,kw) - 3][c] * dY[pix][co],   49 taps x (32 x 32) = 49 MFMA accumulators.
// One persistent workgroup per CU, 7 waves = the 7 kernel rows; wave kh keeps the 7 accumulators of its row (independent
// MFMA chains).  Per 4x16-pixel output tile the 13x37 input patch (whitened, zero padded AFTER whitening) and the dY
// slab go to LDS; the global loads of tile t+1 are issued before the K loop of tile t and parked in registers (the K loop
// itself touches only LDS, so nothing waits on them).  K loop: two pixels per v_mfma_f32_32x32x2_f32, operands are
// conflict-free ds_read_b32 (32 consecutive channels per half wave), 8 reads per 7 MFMAs.
__global__ __launch_bounds__(448) void wgrad_stem_lds_kernel(const WgradArgs p) {
  constexpr int TH = 4, TW = 16, PH = 2 * TH + 5, PWR = 2 * TW + 5, NPIX = PH * PWR, PP = 36, NTHR = 448;
  constexpr int PXP = NTHR / 16;                      // pixels per staging pass (16 channel pairs per pixel)
  constexpr int NXP = (NPIX + PXP - 1) / PXP;         // 18 passes
  constexpr int NDP = (TH * TW * 8 + NTHR - 1) / NTHR;   // 2 passes of 16-byte dY items
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xs = lds;                                    // [NPIX][PP]
  float *ds = lds + NPIX * PP;                        // [TH*TW][PP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int kh = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int chunk = blockIdx.x;
  const int ntiles = p.B * p.tiles_x * p.tiles_y;

  // staging role: channel pair pq of pixel pp0 + 28*k
  const int pq = tid & 15, pp0 = tid >> 4;
  const SrcLane s0 = p.src[2 * pq], s1 = p.src[2 * pq + 1];
  const float sc0 = p.in_scale ? p.in_scale[2 * pq] : s0.sc, sh0 = p.in_shift ? p.in_shift[2 * pq] : s0.sh;
  const float sc1 = p.in_scale ? p.in_scale[2 * pq + 1] : s1.sc, sh1 = p.in_shift ? p.in_shift[2 * pq + 1] : s1.sh;
  const int g8 = tid & 7;

  f32x2 vx[NXP];
  f32x4 vd[NDP];
  unsigned okm = 0;
  auto gload = [&](int t) {
    int q = t;
    const int tx = q % p.tiles_x;
    q /= p.tiles_x;
    const int ty = q % p.tiles_y;
    const int n = q / p.tiles_y;
    const int hb = 2 * ty * TH - 3, wb = 2 * tx * TW - 3;
    okm = 0;
#pragma unroll
    for (int k = 0; k < NXP; ++k) {
      const int pix = pp0 + k * PXP;
      const int pr = pix / PWR, pc = pix - pr * PWR;
      const int hi = hb + pr, wi = wb + pc;
      const bool ok = pix < NPIX && s0.base != nullptr && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      okm |= (ok ? 1u : 0u) << k;
      const float *addr = ok ? s0.base + (((long)n * p.H + hi) * p.W + wi) * s0.nch + s0.choff : p.zero_page;
      vx[k] = *reinterpret_cast<const f32x2 *>(addr);
    }
    const float *db = p.dy + ((long)n * p.Ho * p.Wo) * p.DYC + 4 * g8;
#pragma unroll
    for (int k = 0; k < NDP; ++k) {
      const int pix = (tid + k * NTHR) >> 3;
      const int qy = pix / TW, qx = pix - qy * TW;
      const int oy = ty * TH + qy, ox = tx * TW + qx;
      const bool in = pix < TH * TW && oy < p.Ho && ox < p.Wo;
      vd[k] = in ? *reinterpret_cast<const f32x4 *>(db + ((long)oy * p.Wo + ox) * p.DYC) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int k = 0; k < NXP; ++k) {
      const int pix = pp0 + k * PXP;
      if (pix < NPIX) {
        const bool ok = (okm >> k) & 1u;              // zero padding (and pad channels) AFTER the whitening
        f32x2 v;
        v[0] = ok ? __builtin_fmaf(vx[k][0], sc0, sh0) : 0.f;
        v[1] = ok ? __builtin_fmaf(vx[k][1], sc1, sh1) : 0.f;
        *reinterpret_cast<f32x2 *>(xs + pix * PP + 2 * pq) = v;
      }
    }
#pragma unroll
    for (int k = 0; k < NDP; ++k) {
      const int pix = (tid + k * NTHR) >> 3;
      if (pix < TH * TW) *reinterpret_cast<f32x4 *>(ds + pix * PP + 4 * g8) = vd[k];
    }
  };

  f32x16 acc[7];
#pragma unroll
  for (int kw = 0; kw < 7; ++kw)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kw][r] = 0.f;

  int t0 = chunk * p.tiles_per_chunk, t1 = t0 + p.tiles_per_chunk;
  if (t1 > ntiles) t1 = ntiles;
  const float *xa = xs + (kh * PWR + 2 * h) * PP + i;     // pixel (2*qy + kh, 2*(2*sx + h) + kw)
  const float *da = ds + h * PP + i;
  if (t0 < t1) gload(t0);
  for (int t = t0; t < t1; ++t) {
    __syncthreads();                                  // the previous tile's readers are done
    lstore();
    __syncthreads();
    if (t + 1 < t1) gload(t + 1);                     // lands during the K loop (which only touches LDS)
    for (int qy = 0; qy < TH; ++qy) {
      const float *xr = xa + (2 * qy) * PWR * PP, *dr = da + qy * TW * PP;
#pragma unroll 2
      for (int sx = 0; sx < TW / 2; ++sx) {
        const float b = dr[(2 * sx) * PP];
        float a[7];
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) a[kw] = xr[(4 * sx + kw) * PP];
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kw], b, acc[kw], 0, 0, 0);
      }
    }
  }
  // C/D layout: col j (= co) = lane&31, row i (= channel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int kw = 0; kw < 7; ++kw) {
    float *dst = p.partial + ((((long)kh * p.chunks + chunk) * 7 + kw) * 32) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[(long)row * 32 + i] = acc[kw][r];
    }
  }
}

// grad[map(co, ci, tap)] = sum over chunks (fixed order) of partial[chunk, pair, grp][t][ci_row][co_col].
// One block per (pair, tap, ci row): 32 output channels x 8 chunk lanes; lane k sums chunks k, k+8, ... in fp64
// (coalesced 128-B reads), the 8 lane sums are combined in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs p, int TG, float *grad, const int *ci_perm,
                                                         int cin_out) {
  __shared__ double red[8][32];
  const int T = p.KH * p.KW;
  int b = blockIdx.x;
  const int row = b & 31;
  b >>= 5;
  const int tap = b % T;
  const int pair = b / T;
  const int cit = pair % p.ci_tiles, cot = pair / p.ci_tiles;
  const int col = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int grp = tap / TG, t = tap - grp * TG;
  double acc = 0.0;
  for (int c = cl; c < p.chunks; c += 8) {
    const long unit = ((long)grp * p.pairs + pair) * p.chunks + c;
    acc += (double)p.partial[(((unit * TG) + t) * 32 + row) * 32 + col];
  }
  red[cl][col] = acc;
  __syncthreads();
  if (cl == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][col];
    const int ci = cit * 32 + row, co = cot * 32 + col;
    if (ci < p.CIN && co < p.COUT) {
      const int cio = ci_perm ? ci_perm[ci] : ci;     // stem: kernel channel order -> reference channel order
      if (cio >= 0 && cio < cin_out)
        grad[(long)co * (p.grad_pitch ? p.grad_pitch : (long)cin_out * T) + (long)cio * T + tap] = (float)s;
    }
  }
}

template <int TG>
static hipError_t launch_wgrad_t(const WgradArgs &a, hipStream_t s) {
  const long nunits = (long)a.chunks * a.pairs * a.groups;
  dim3 grid((unsigned)((nunits + 3) / 4));
  if (a.mode == 2)
    hipLaunchKernelGGL((wgrad_kernel<TG, 2>), grid, dim3(256), 0, s, a);
  else if (a.mode == 1)
    hipLaunchKernelGGL((wgrad_kernel<TG, 1>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((wgrad_kernel<TG, 0>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

// tile + unit plan of wgrad3_mw_kernel; false: shape outside it (the nine-wave kernel or the generic one takes over)
static bool wgrad_plan_mw(WgradArgs &a) {
  const int S = a.stride;
  const int cit = a.CIN / 32, cot = (a.COUT + 31) / 32;    // (a padded last co-tile reads dY's zero pad channels)
  if (cit % 2 == 0 && cot % 2 == 0) {
    a.cs = 2; a.os = 2; a.rs = 1;
  } else if (cit == 1 && cot % 2 == 0) {
    a.cs = 1; a.os = 2; a.rs = 2;
  } else if (cit % 2 == 0 && cot == 1) {
    a.cs = 2; a.os = 1; a.rs = 2;
  } else if (cit == 1 && cot == 1) {
    a.cs = 1; a.os = 1; a.rs = 4;
  } else {
    return false;
  }
  const int tw_max = S == 1 ? 24 : 12;
  a.tiles_x = (a.Wo + tw_max - 1) / tw_max;
  a.TW = ((a.Wo + a.tiles_x - 1) / a.tiles_x + 1) / 2 * 2;
  const size_t lds_max = 160 * 1024;
  auto bytes = [&](int th) {
    return (size_t)(a.cs * ((th - 1) * S + 3) * ((a.TW - 1) * S + 3) + a.os * th * a.TW) * 36 * 4;
  };
  auto items_ok = [&](int th) {
    return (long)a.cs * ((th - 1) * S + 3) * ((a.TW - 1) * S + 3) * 8 <= 5 * 768 && (long)a.os * th * a.TW * 8 <= 2 * 768;
  };
  int th = a.Ho < 8 ? a.Ho : 8;
  while (th > 1 && (!items_ok(th) || 2 * bytes(th) > lds_max)) --th;
  if (!items_ok(th) || bytes(th) > lds_max) return false;
  a.tiles_y = (a.Ho + th - 1) / th;
  a.TH = (a.Ho + a.tiles_y - 1) / a.tiles_y;
  a.nbuf = 2 * bytes(a.TH) <= lds_max ? 2 : 1;
  a.TG = 9;
  a.groups = 1;
  a.ci_tiles = cit;
  a.pairs = cit * cot;
  const int pgs = (cit / a.cs) * (cot / a.os);
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  long chunks = 256 / pgs;                              // one 12-wave workgroup per CU
  if (chunks < 1) chunks = 1;
  if (chunks > ntiles) chunks = ntiles;
  a.tiles_per_chunk = (int)((ntiles + chunks - 1) / chunks);
  a.wg_chunks = (int)((ntiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk);
  a.chunks = a.wg_chunks;                               // row groups are summed inside the workgroup
  a.pix_per_chunk = 0;
  a.lds3 = S == 1 ? 4 : 5;
  return true;
}

void wgrad_plan(WgradArgs &a) {
  a.lds3 = 0;
  if (a.use_x3 && wgrad_x3_plan(a)) return;              // 3x3 stride-1 convs: three-piece operands on the bf16 matrix cores
  static const bool no_lds = std::getenv("PNVO_WGRAD") && std::strcmp(std::getenv("PNVO_WGRAD"), "generic") == 0;
  static const bool nine = std::getenv("PNVO_WGRAD") && std::strcmp(std::getenv("PNVO_WGRAD"), "lds9") == 0;
  if (!no_lds && !nine && a.mode != 2 && a.KH == 3 && a.KW == 3 && a.pad == 1 && a.CIN % 32 == 0 && a.DYC >= (a.COUT + 31) / 32 * 32 &&
      a.DYC % 4 == 0 && ((a.stride == 1 && a.H == a.Ho && a.W == a.Wo) || a.stride == 2) && wgrad_plan_mw(a))
    return;
  if (!no_lds && a.mode != 2 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.CIN % 32 == 0 &&
      a.COUT % 32 == 0 && a.DYC % 4 == 0 && a.H == a.Ho && a.W == a.Wo) {
    a.lds3 = 1;
    a.TG = 9;
    a.groups = 1;
    a.ci_tiles = a.CIN / 32;
    a.pairs = a.ci_tiles * (a.COUT / 32);
    a.tiles_x = (a.W + 23) / 24;
    a.TW = ((a.W + a.tiles_x - 1) / a.tiles_x + 1) / 2 * 2;
    static const int th_max = std::getenv("PNVO_WGRAD_TH") ? std::atoi(std::getenv("PNVO_WGRAD_TH")) : 8;
    static const int wg_cu = std::getenv("PNVO_WGRAD_WGS") ? std::atoi(std::getenv("PNVO_WGRAD_WGS")) : 2;
    a.tiles_y = (a.H + th_max - 1) / th_max;
    a.TH = (a.H + a.tiles_y - 1) / a.tiles_y;
    const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
    long chunks = 256L * wg_cu / a.pairs;             // persistent workgroups per CU
    if (chunks < 1) chunks = 1;
    if (chunks > ntiles) chunks = ntiles;
    a.tiles_per_chunk = (int)((ntiles + chunks - 1) / chunks);
    a.chunks = (int)((ntiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk);
    a.pix_per_chunk = 0;
    return;
  }
  if (!no_lds && a.mode != 2 && a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.CIN % 32 == 0 &&
      a.COUT % 32 == 0 && a.DYC % 4 == 0) {
    // strided 3x3: same kernel with a (2T+1)-pixel patch per axis; tiles of <= 4 x 12 outputs keep the patch at 9 x 25
    // pixels (<= 4 sixteen-byte staging items per thread)
    a.lds3 = 3;
    a.TG = 9;
    a.groups = 1;
    a.ci_tiles = a.CIN / 32;
    a.pairs = a.ci_tiles * (a.COUT / 32);
    a.tiles_x = (a.Wo + 11) / 12;
    a.TW = ((a.Wo + a.tiles_x - 1) / a.tiles_x + 1) / 2 * 2;
    a.tiles_y = (a.Ho + 3) / 4;
    a.TH = (a.Ho + a.tiles_y - 1) / a.tiles_y;
    const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
    long chunks = 256L * 2 / a.pairs;
    if (chunks < 1) chunks = 1;
    if (chunks > ntiles) chunks = ntiles;
    a.tiles_per_chunk = (int)((ntiles + chunks - 1) / chunks);
    a.chunks = (int)((ntiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk);
    a.pix_per_chunk = 0;
    return;
  }
  if (!no_lds && a.mode == 2 && a.KH == 7 && a.KW == 7 && a.stride == 2 && a.pad == 3 && a.CIN <= 32 && a.COUT == 32 &&
      a.DYC % 4 == 0) {                                  // the stem: wgrad_stem_lds_kernel, one workgroup per CU
    a.lds3 = 2;
    a.TG = 7;
    a.groups = 7;
    a.ci_tiles = 1;
    a.pairs = 1;
    a.TH = 4;
    a.TW = 16;
    a.tiles_x = (a.Wo + 15) / 16;
    a.tiles_y = (a.Ho + 3) / 4;
    const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
    long chunks = ntiles < 256 ? ntiles : 256;
    a.tiles_per_chunk = (int)((ntiles + chunks - 1) / chunks);
    a.chunks = (int)((ntiles + a.tiles_per_chunk - 1) / a.tiles_per_chunk);
    a.pix_per_chunk = 0;
    return;
  }
  const int T = a.KH * a.KW;
  a.TG = T >= 9 ? (T % 9 == 0 ? 9 : (T % 7 == 0 ? 7 : (T % 6 == 0 ? 6 : 9))) : (T >= 3 ? 3 : 1);
  if (T == 1) a.TG = 1;
  a.groups = (T + a.TG - 1) / a.TG;
  a.ci_tiles = (a.CIN + 31) / 32;
  a.pairs = a.ci_tiles * ((a.COUT + 31) / 32);
  const long M = (long)a.B * a.Ho * a.Wo;
  long chunks = 2048 / ((long)a.pairs * a.groups);
  if (chunks < 1) chunks = 1;
  long ppc = (M + chunks - 1) / chunks;
  if (ppc < 64) ppc = 64;
  ppc = (ppc + 1) / 2 * 2;
  a.pix_per_chunk = ppc;
  a.chunks = (int)((M + ppc - 1) / ppc);
}

size_t wgrad_partial_floats(const WgradArgs &a) { return (size_t)a.chunks * a.pairs * a.groups * a.TG * 1024; }

hipError_t launch_wgrad(const WgradArgs &a, float *grad, const int *ci_perm, int cin_out, hipStream_t s) {
  hipError_t e;
  if (a.lds3 == 6) {
    e = launch_wgrad_x3(a, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(a.pairs * 9 * 32)), dim3(256), 0, s, a, a.TG, grad, ci_perm, cin_out);
    return hipGetLastError();
  }
  if (a.lds3 == 2) {
    const size_t lds = (size_t)(13 * 37 + 64) * 36 * 4;
    hipLaunchKernelGGL(wgrad_stem_lds_kernel, dim3((unsigned)a.chunks), dim3(448), lds, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(49 * 32)), dim3(256), 0, s, a, a.TG, grad, ci_perm, cin_out);
    return hipGetLastError();
  }
  if (a.lds3 == 4 || a.lds3 == 5) {
    const int S = a.lds3 == 5 ? 2 : 1;
    size_t lds = (size_t)(a.cs * ((a.TH - 1) * S + 3) * ((a.TW - 1) * S + 3) + a.os * a.TH * a.TW) * 36 * 4 * a.nbuf;
    if (a.rs > 1 && lds < 12 * 16 * 64 * 4) lds = 12 * 16 * 64 * 4;      // the row-group sum at the end goes through LDS
    dim3 grid((unsigned)((a.ci_tiles / a.cs) * ((a.COUT + 31) / 32 / a.os) * a.wg_chunks));
    if (S == 2 && a.mode == 1)
      hipLaunchKernelGGL((wgrad3_mw_kernel<1, 2>), grid, dim3(768), lds, s, a);
    else if (S == 2)
      hipLaunchKernelGGL((wgrad3_mw_kernel<0, 2>), grid, dim3(768), lds, s, a);
    else if (a.mode == 1)
      hipLaunchKernelGGL((wgrad3_mw_kernel<1, 1>), grid, dim3(768), lds, s, a);
    else
      hipLaunchKernelGGL((wgrad3_mw_kernel<0, 1>), grid, dim3(768), lds, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(a.pairs * 9 * 32)), dim3(256), 0, s, a, a.TG, grad, ci_perm,
                       cin_out);
    return hipGetLastError();
  }
  if (a.lds3) {
    const int S = a.lds3 == 3 ? 2 : 1;
    const size_t lds = (size_t)(((a.TH - 1) * S + 3) * ((a.TW - 1) * S + 3) + a.TH * a.TW) * 36 * 4;
    dim3 grid((unsigned)(a.pairs * a.chunks));
    if (S == 2 && a.mode == 1)
      hipLaunchKernelGGL((wgrad3_lds_kernel<1, 2>), grid, dim3(576), lds, s, a);
    else if (S == 2)
      hipLaunchKernelGGL((wgrad3_lds_kernel<0, 2>), grid, dim3(576), lds, s, a);
    else if (a.mode == 1)
      hipLaunchKernelGGL((wgrad3_lds_kernel<1, 1>), grid, dim3(576), lds, s, a);
    else
      hipLaunchKernelGGL((wgrad3_lds_kernel<0, 1>), grid, dim3(576), lds, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(a.pairs * 9 * 32)), dim3(256), 0, s, a, a.TG, grad, ci_perm,
                       cin_out);
    return hipGetLastError();
  }
  switch (a.TG) {
    case 9: e = launch_wgrad_t<9>(a, s); break;
    case 7: e = launch_wgrad_t<7>(a, s); break;
    case 6: e = launch_wgrad_t<6>(a, s); break;
    case 3: e = launch_wgrad_t<3>(a, s); break;
    case 1: e = launch_wgrad_t<1>(a, s); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(a.pairs * a.KH * a.KW * 32)), dim3(256), 0, s, a, a.TG, grad,
                     ci_perm, cin_out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm backward.  y = gamma*xh + beta, xh = (x - mu)*rstd;  g = dOut [* (mask_src > 0)]:
//   dgamma_c = sum_{n,p} g*xh,  dbeta_c = sum_{n,p} g
//   dx = rstd * (gamma*g - S1/N - xh*S2/N),  S1 = sum_{group} gamma*g,  S2 = sum_{group} gamma*g*xh,  N = cpg*P
// mask: 0 none, 1 recompute a = x*scale+shift (this layer's own GN+ReLU output) and use a > 0.
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float *x, const float *dout, const float *scale,
                                                          const float *shift, const float *mu, const float *rstd, int C,
                                                          int Creal, int G, long P, int chunks, int mask, float *part) {
  // grid = B * chunks; thread t owns channel c = t % C for pixels t / C, + 256 / C, ...   (C divides 256 or C >= 256)
  __shared__ float red[512];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const long ppc = (P + chunks - 1) / chunks;
  const long p0 = (long)chunk * ppc;
  long p1 = p0 + ppc;
  if (p1 > P) p1 = P;
  const int cpg = Creal / G;
  for (int cb = 0; cb < C; cb += 256) {
    const int lanes_c = C - cb < 256 ? C - cb : 256;      // channels handled in this pass
    const int c = cb + (int)(threadIdx.x % lanes_c);
    const int pl = threadIdx.x / lanes_c, pstep = 256 / lanes_c;
    float s1 = 0.f, s2 = 0.f;
    if (c < Creal) {
      const float m_ = mu[n * G + c / cpg], r_ = rstd[n * G + c / cpg];
      const float sc = mask ? scale[(long)n * C + c] : 0.f, sh = mask ? shift[(long)n * C + c] : 0.f;
      for (long p = p0 + pl; p < p1; p += pstep) {
        const long idx = ((long)n * P + p) * C + c;
        const float xv = x[idx];
        float g = dout[idx];
        if (mask && !(__builtin_fmaf(xv, sc, sh) > 0.f)) g = 0.f;
        s1 += g;
        s2 = __builtin_fmaf(g, (xv - m_) * r_, s2);
      }
    }
    red[threadIdx.x * 2] = s1;
    red[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    if ((int)threadIdx.x < lanes_c) {
      float a1 = 0.f, a2 = 0.f;
      for (int k = 0; k < pstep; ++k) {
        a1 += red[(k * lanes_c + threadIdx.x) * 2];
        a2 += red[(k * lanes_c + threadIdx.x) * 2 + 1];
      }
      float *dst = part + (((long)n * chunks + chunk) * C + cb + threadIdx.x) * 2;
      dst[0] = a1;
      dst[1] = a2;
    }
    __syncthreads();
  }
}

// The same reduction with FOUR consecutive channels per thread (16-byte loads; C % 4 == 0 and C / 4 divides 256 or is a multiple
// of it — every GroupNorm of these models): a quarter of the load instructions of the kernel above.
__global__ __launch_bounds__(256) void gn_bwd_reduce4_kernel(const float *x, const float *dout, const float *scale,
                                                           const float *shift, const float *mu, const float *rstd, int C,
                                                           int Creal, int G, long P, int chunks, int mask, float *part) {
  __shared__ float red[256 * 8];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const long ppc = (P + chunks - 1) / chunks;
  const long p0 = (long)chunk * ppc;
  long p1 = p0 + ppc;
  if (p1 > P) p1 = P;
  const int cpg = Creal / G;
  const int Q = C / 4;                                     // channel quads
  for (int qb = 0; qb < Q; qb += 256) {
    const int lanes_q = Q - qb < 256 ? Q - qb : 256;       // quads handled in this pass
    const int c0 = 4 * (qb + (int)(threadIdx.x % lanes_q));
    const int pl = threadIdx.x / lanes_q, pstep = 256 / lanes_q;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float m_[4], r_[4];
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = c0 + t < Creal ? c0 + t : Creal - 1;   // (pad channels: any valid statistics; their sums are never read)
      m_[t] = mu[n * G + c / cpg];
      r_[t] = rstd[n * G + c / cpg];
    }
    if (mask) {
      sc = *reinterpret_cast<const f32x4 *>(scale + (long)n * C + c0);
      sh = *reinterpret_cast<const f32x4 *>(shift + (long)n * C + c0);
    }
    for (long p = p0 + pl; p < p1; p += pstep) {
      const long idx = ((long)n * P + p) * C + c0;
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + idx), gv = *reinterpret_cast<const f32x4 *>(dout + idx);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float g = gv[t];
        if (mask && !(__builtin_fmaf(xv[t], sc[t], sh[t]) > 0.f)) g = 0.f;
        s1[t] += g;
        s2[t] = __builtin_fmaf(g, (xv[t] - m_[t]) * r_[t], s2[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      red[threadIdx.x * 8 + 2 * t] = s1[t];
      red[threadIdx.x * 8 + 2 * t + 1] = s2[t];
    }
    __syncthreads();
    if ((int)threadIdx.x < lanes_q) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < pstep; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += red[(k * lanes_q + threadIdx.x) * 8 + e];
      float *dst = part + (((long)n * chunks + chunk) * C + c0) * 2;
      *reinterpret_cast<f32x4 *>(dst) = f32x4{a[0], a[1], a[2], a[3]};
      *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{a[4], a[5], a[6], a[7]};
    }
    __syncthreads();
  }
}

// stage A: per (n, channel) sum of the chunk partials (fixed order) -> nc[n][c][2]
__global__ __launch_bounds__(256) void gn_bwd_sum_chunks_kernel(const float *part, int B, int C, int chunks, float *nc) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * C) return;
  const int n = (int)(e / C), c = (int)(e % C);
  double a1 = 0.0, a2 = 0.0;
  for (int ch = 0; ch < chunks; ++ch) {
    const float *src = part + (((long)n * chunks + ch) * C + c) * 2;
    a1 += (double)src[0];
    a2 += (double)src[1];
  }
  nc[e * 2] = (float)a1;
  nc[e * 2 + 1] = (float)a2;
}

// stage B: per (n, group): c1 = S1/N, c2 = S2/N;  per channel: dgamma, dbeta (sum over n in a fixed order, fp64).
// Blocks [0, nb_coef) do the coefficients (one thread per (n, group)); the following blocks take one channel per WAVE:
// lane l sums samples l, l+64, ... and a fixed xor-shuffle tree combines the lanes.
// (chunks > 0: `nc` holds the CHUNK partials [B][chunks][C][2] and the sum over chunks — stage A — happens here, in fp64, in
// chunk order: one launch fewer per GroupNorm; chunks == 0: nc is stage A's output.)
__device__ __forceinline__ void gn_nc(const float *nc, int chunks, int C, int n, int c, double &a1, double &a2) {
  if (chunks <= 0) {
    a1 = (double)nc[((long)n * C + c) * 2];
    a2 = (double)nc[((long)n * C + c) * 2 + 1];
    return;
  }
  a1 = a2 = 0.0;
  const float2 *src = reinterpret_cast<const float2 *>(nc) + ((long)n * chunks) * C + c;
  int ch = 0;
  for (; ch + 8 <= chunks; ch += 8) {             // eight loads in flight, added in chunk order
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(long)(ch + k) * C];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a1 += (double)v[k].x;
      a2 += (double)v[k].y;
    }
  }
  for (; ch < chunks; ++ch) {
    const float2 v = src[(long)ch * C];
    a1 += (double)v.x;
    a2 += (double)v.y;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float *nc, int B, int C, int Creal, int G, long P,
                                                            const float *gamma, float *coef, float *dgamma, float *dbeta,
                                                            int nb_coef, int chunks = 0) {
  const int cpg = Creal / G;
  if ((int)blockIdx.x >= nb_coef) {               // per-channel parameter gradients
    const int c = ((int)blockIdx.x - nb_coef) * 4 + (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= Creal) return;
    double dg = 0.0, db = 0.0;
    for (int n = lane; n < B; n += 64) {
      double a1, a2;
      gn_nc(nc, chunks, C, n, c, a1, a2);
      db += (double)(float)a1;                    // (rounded to float as stage A stores it: same bits either way)
      dg += (double)(float)a2;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      db += __shfl_xor(db, o);
      dg += __shfl_xor(dg, o);
    }
    if (lane == 0) {
      dgamma[c] = (float)dg;
      dbeta[c] = (float)db;
    }
    return;
  }
  // per-(sample, group) coefficients: one WAVE per (n, g); lane k sums the chunk partials of channel g*cpg + k (chunk order,
  // fp64), lane 0 then adds the channels in channel order — the arithmetic of the one-thread-per-group form, cpg x fewer
  // dependent loads per thread and 64 x more workgroups (that form: 13-49 us per GroupNorm on 8-24 workgroups)
  const double N = (double)cpg * (double)P;
  const int e = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e < B * G) {
    const int n = e / G, g = e % G;
    double t1 = 0.0, t2 = 0.0;
    for (int k0 = 0; k0 < cpg; k0 += 64) {
      const int k = k0 + lane;
      double p1 = 0.0, p2 = 0.0;
      if (k < cpg) {
        const int c = g * cpg + k;
        double a1, a2;
        gn_nc(nc, chunks, C, n, c, a1, a2);
        p1 = (double)gamma[c] * (double)(float)a1;
        p2 = (double)gamma[c] * (double)(float)a2;
      }
      const int lim = cpg - k0 < 64 ? cpg - k0 : 64;
      for (int q = 0; q < lim; ++q) {            // channel order, as a single thread would add them
        t1 += __shfl(p1, q);
        t2 += __shfl(p2, q);
      }
    }
    if (lane == 0) {
      coef[((long)n * G + g) * 2] = (float)(t1 / N);
      coef[((long)n * G + g) * 2 + 1] = (float)(t2 / N);
    }
  }
}

// dx for 4 consecutive channels per thread (C % 4 == 0; the 4 channels may straddle groups when C/G < 4)
// absmax (optional, device uint[PNVO_ABSMAX_SLOTS * 16]): the float bits of max |dx| over the tensor, spread over 64 slots by
// atomicMax (non-negative floats order like their bit patterns; readers take the maximum of the slots) — the power-of-two scale the float16 forms of the backward-data conv and of the weight gradient
// put on this gradient before they split it (conv_x3.hip in_absmax, wgrad_x3.hip dy_absmax).
__device__ __forceinline__ void block_absmax(float v, unsigned *absmax) {
  __shared__ float wmax[4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    // 64 slots, one cache line apart (the workgroups of a launch would serialise on one address: measured +1.3 ms per step), and
    // no atomic at all when the slot already holds a value at least as large (most workgroups after the first few)
    const unsigned m = __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
    unsigned *slot = absmax + (blockIdx.x & 63u) * 16u;
    if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *x, const float *dout, const float *scale,
                                                         const float *shift, const float *mu, const float *rstd,
                                                         const float *gamma, const float *coef, int C, int Creal, int G,
                                                         long P, long total4, int mask, float *dx, unsigned *absmax) {
  const long e4 = (long)blockIdx.x * 256 + threadIdx.x;
  float amax = 0.f;
  if (e4 < total4) {
    const long e = e4 * 4;
    const int c0 = (int)(e % C);
    const int n = (int)(e / (P * C));
    const int cpg = Creal / G;
    const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + e), gv = *reinterpret_cast<const f32x4 *>(dout + e);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (mask) {
      sc = *reinterpret_cast<const f32x4 *>(scale + (long)n * C + c0);
      sh = *reinterpret_cast<const f32x4 *>(shift + (long)n * C + c0);
    }
    f32x4 out;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = c0 + t;
      float o = 0.f;
      if (c < Creal) {
        const int g = c / cpg;
        float gd = gv[t];
        if (mask && !(__builtin_fmaf(xv[t], sc[t], sh[t]) > 0.f)) gd = 0.f;
        const float r_ = rstd[n * G + g];
        const float xh = (xv[t] - mu[n * G + g]) * r_;
        o = r_ * (gamma[c] * gd - coef[((long)n * G + g) * 2] - xh * coef[((long)n * G + g) * 2 + 1]);
      }
      out[t] = o;                                   // channel-pad lanes (compression 31 -> 32) get 0
      amax = fmaxf(amax, __builtin_fabsf(o));
    }
    *reinterpret_cast<f32x4 *>(dx + e) = out;
  }
  if (absmax != nullptr) block_absmax(amax, absmax);   // (wave-uniform condition: the whole workgroup takes part)
}

// ---- the stem's GroupNorm backward with the max-pool backward folded in: dOut of the stem's GN+ReLU is the gather of the
// pooled gradient (pixel (hi, wi) collects dPool of every 3x3 stride-2 window that elected it), computed on the fly in both
// passes instead of materialising the 4 x larger tensor (one write + two reads of 270 MB at 128 pairs saved).
struct PoolGrad {
  const float *dpool;          // [B, Hp, Wp, C] gradient of the pooled activations
  const unsigned char *idx;    // [B, Hp, Wp, C] elected tap (kh * 3 + kw) of each window
  int H, W, Hp, Wp;            // H x W: the stem's output, Hp x Wp: pooled
};
__device__ __forceinline__ f32x4 pool_grad4(const PoolGrad &q, int n, int hi, int wi, int C, int c) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int t = hi + 1 - kh;          // 2*ho = hi + 1 - kh
    if (t < 0 || (t & 1)) continue;
    const int ho = t >> 1;
    if (ho >= q.Hp) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int u = wi + 1 - kw;
      if (u < 0 || (u & 1)) continue;
      const int wo = u >> 1;
      if (wo >= q.Wp) continue;
      const long o = (((long)n * q.Hp + ho) * q.Wp + wo) * C + c;
      const unsigned ix = *reinterpret_cast<const unsigned *>(q.idx + o);
      const f32x4 dp = *reinterpret_cast<const f32x4 *>(q.dpool + o);
      const unsigned k = (unsigned)(kh * 3 + kw);
      if ((ix & 0xffu) == k) acc[0] += dp[0];
      if (((ix >> 8) & 0xffu) == k) acc[1] += dp[1];
      if (((ix >> 16) & 0xffu) == k) acc[2] += dp[2];
      if ((ix >> 24) == k) acc[3] += dp[3];
    }
  }
  return acc;
}

// reduce pass, four channels per thread: block = (sample, pixel chunk); thread = (channel quad t % (C/4), pixel lane t / (C/4))
__global__ __launch_bounds__(256) void gn_bwd_reduce_pool_kernel(const float *x, const PoolGrad q, const float *scale, const float *shift,
                                                               const float *mu, const float *rstd, int C, int G, long P, int chunks,
                                                               float *part) {
  __shared__ float red[256 * 8];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const long ppc = (P + chunks - 1) / chunks;
  const long p0 = (long)chunk * ppc;
  long p1 = p0 + ppc;
  if (p1 > P) p1 = P;
  const int Q = C >> 2, cq = threadIdx.x % Q, pl = threadIdx.x / Q, pstep = 256 / Q, c = 4 * cq;
  const int cpg = C / G;
  f32x4 m_, r_;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m_[e] = mu[n * G + (c + e) / cpg];
    r_[e] = rstd[n * G + (c + e) / cpg];
  }
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + (long)n * C + c), sh = *reinterpret_cast<const f32x4 *>(shift + (long)n * C + c);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (long p = p0 + pl; p < p1; p += pstep) {
    const int hi = (int)(p / q.W), wi = (int)(p - (long)hi * q.W);
    const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + ((long)n * P + p) * C + c);
    const f32x4 gv = pool_grad4(q, n, hi, wi, C, c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = __builtin_fmaf(xv[e], sc[e], sh[e]) > 0.f ? gv[e] : 0.f;
      s1[e] += g;
      s2[e] = __builtin_fmaf(g, (xv[e] - m_[e]) * r_[e], s2[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[threadIdx.x * 8 + 2 * e] = s1[e];
    red[threadIdx.x * 8 + 2 * e + 1] = s2[e];
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {                              // thread = channel: sums its quad's pixel lanes in a fixed order
    const int qd = threadIdx.x >> 2, e = threadIdx.x & 3;
    float a1 = 0.f, a2 = 0.f;
    for (int k = 0; k < pstep; ++k) {
      a1 += red[(k * Q + qd) * 8 + 2 * e];
      a2 += red[(k * Q + qd) * 8 + 2 * e + 1];
    }
    float *dst = part + (((long)n * chunks + chunk) * C + threadIdx.x) * 2;
    dst[0] = a1;
    dst[1] = a2;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_pool_kernel(const float *x, const PoolGrad q, const float *scale, const float *shift,
                                                              const float *mu, const float *rstd, const float *gamma, const float *coef,
                                                              int C, int G, long P, long total4, float *dx) {
  const long e4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (e4 >= total4) return;
  const long e = e4 * 4;
  const int c0 = (int)(e % C);
  const long pix = e / C;
  const int n = (int)(pix / P);
  const long p = pix - (long)n * P;
  const int hi = (int)(p / q.W), wi = (int)(p - (long)hi * q.W);
  const int cpg = C / G;
  const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + e), gv = pool_grad4(q, n, hi, wi, C, c0);
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + (long)n * C + c0), sh = *reinterpret_cast<const f32x4 *>(shift + (long)n * C + c0);
  f32x4 out;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = c0 + t, g = c / cpg;
    const float gd = __builtin_fmaf(xv[t], sc[t], sh[t]) > 0.f ? gv[t] : 0.f;
    const float r_ = rstd[n * G + g];
    const float xh = (xv[t] - mu[n * G + g]) * r_;
    out[t] = r_ * (gamma[c] * gd - coef[((long)n * G + g) * 2] - xh * coef[((long)n * G + g) * 2 + 1]);
  }
  *reinterpret_cast<f32x4 *>(dx + e) = out;
}

// GroupNorm (+ ReLU mask) backward of the stem with dOut = max-pool backward of dpool (C % 4 == 0, C <= 256, no pad channels)
hipError_t launch_gn_bwd_pool(const float *x, const float *dpool, const unsigned char *idx, int Hs, int Ws, int Hp, int Wp,
                              const float *scale, const float *shift, const float *mu, const float *rstd, const float *gamma, int B, int C,
                              int G, float *part, float *coef, float *dgamma, float *dbeta, float *dx, hipStream_t s) {
  const long P = (long)Hs * Ws;
  int chunks = (int)(P / 256);
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  PoolGrad q;
  q.dpool = dpool;
  q.idx = idx;
  q.H = Hs;
  q.W = Ws;
  q.Hp = Hp;
  q.Wp = Wp;
  hipLaunchKernelGGL(gn_bwd_reduce_pool_kernel, dim3((unsigned)(B * chunks)), dim3(256), 0, s, x, q, scale, shift, mu, rstd, C, G, P,
                     chunks, part);
  const int nb_coef = (B * G + 3) / 4;               // one wave per (sample, group)
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)(nb_coef + (C + 3) / 4)), dim3(256), 0, s, part, B, C, C, G, P, gamma, coef,
                     dgamma, dbeta, nb_coef, chunks);
  const long total4 = (long)B * P * C / 4;
  hipLaunchKernelGGL(gn_bwd_apply_pool_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, x, q, scale, shift, mu, rstd,
                     gamma, coef, C, G, P, total4, dx);
  return hipGetLastError();
}

hipError_t launch_gn_bwd(const float *x, const float *dout, const float *scale, const float *shift, const float *mu,
                         const float *rstd, const float *gamma, int B, long P, int C, int Creal, int G, int mask,
                         float *part, float *coef, float *dgamma, float *dbeta, float *dx, hipStream_t s, unsigned *absmax) {
  // pixel chunks per sample: >= 64 pixels each, and enough workgroups to fill the chip on the small late stages too
  // (one chunk per sample left the 12x22 / 6x11 stages on 128 workgroups: 33 us for 17 MB)
  int chunks = (int)(P / 256);
  if ((long)B * chunks < 1024) chunks = (int)((P + 63) / 64);
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  // the reduce kernel walks channels 0..C-1 of the padded tensor; pad channels are never read by finalize
  const int Q = C / 4;
  if (C % 4 == 0 && (256 % Q == 0 || Q % 256 == 0))
    hipLaunchKernelGGL(gn_bwd_reduce4_kernel, dim3((unsigned)(B * chunks)), dim3(256), 0, s, x, dout, scale, shift, mu, rstd,
                       C, Creal, G, P, chunks, mask, part);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((unsigned)(B * chunks)), dim3(256), 0, s, x, dout, scale, shift, mu, rstd,
                       C, Creal, G, P, chunks, mask, part);
  const int nb_coef = (B * G + 3) / 4;               // one wave per (sample, group); stage A — the sum over chunks — happens inside
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)(nb_coef + (Creal + 3) / 4)), dim3(256), 0, s, part, B, C, Creal,
                     G, P, gamma, coef, dgamma, dbeta, nb_coef, chunks);
  const long total4 = (long)B * P * C / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, x, dout, scale, shift, mu,
                     rstd, gamma, coef, C, Creal, G, P, total4, mask, dx, absmax);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Dropout with a counter-based mask (splitmix64 finaliser of (seed, step, layer, element)): reproducible, stateless, the
// backward pass recomputes the same mask.
__device__ __forceinline__ unsigned dropout_bits(uint64_t seed, uint64_t step, int layer, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1) + 0xD1B54A32D192ED03ull * (step * 2 + (uint64_t)layer + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32);
}

__global__ __launch_bounds__(256) void dropout_kernel(const float *x, const float *scale, const float *shift, long total,
                                                    long row, int C, unsigned thresh, float inv_keep, uint64_t seed,
                                                    uint64_t step, int layer, float *y) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const float keep = dropout_bits(seed, step, layer, (uint64_t)e) >= thresh ? inv_keep : 0.f;
  float v = 1.f;
  if (x != nullptr) {
    v = x[e];
    if (scale != nullptr) {
      const long n = e / row;
      const int c = (int)(e % C);
      v = fmaxf(__builtin_fmaf(v, scale[n * C + c], shift[n * C + c]), 0.f);
    }
  }
  y[e] = v * keep;
}

hipError_t launch_dropout(const float *x, const float *scale, const float *shift, int B, long P, int C, float p,
                          uint64_t seed, uint64_t step, int layer, float *y, hipStream_t s) {
  const long total = (long)B * P * C;
  double th = (double)p * 4294967296.0;
  if (th > 4294967295.0) th = 4294967295.0;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, scale, shift, total, P * C, C,
                     (unsigned)th, 1.0f / (1.0f - p), seed, step, layer, y);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Action-embedding variants (vo_cnn_act_embed.py:56-72): hidden = relu(Linear([visual | emb[action]])).  The visual columns
// run as the fh x fw "conv"; the 32 embedding columns are a per-sample bias row
//   bias[n][o] = b1[o] + sum_e W1[o][flat + e] * efeat[n][e],   efeat = dropout(emb[action_n])
// and their backward is three tiny kernels (dW1 columns, d efeat, scatter into the embedding rows; fixed summation order).
__global__ __launch_bounds__(256) void embed_gather_kernel(const float *emb, const long long *actions, int B, int rows,
                                                         float *out, int *err) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * 32) return;
  const int n = e >> 5;
  const long long a = actions ? actions[n] : (long long)n;        // actions == nullptr: the table itself (rows = B)
  if (a < 0 || a >= rows) {
    if (err) *err = 1;
    out[e] = 0.f;
    return;
  }
  out[e] = emb[a * 32 + (e & 31)];
}

__global__ __launch_bounds__(256) void embed_bias_kernel(const float *efeat, const float *w1, long pitch, int flat,
                                                       const float *b1, int B, int hidden, float *bias) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * hidden) return;
  const int n = e / hidden, o = e - n * hidden;
  const float *w = w1 + (long)o * pitch + flat, *f = efeat + (long)n * 32;
  double acc = 0.0;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) acc += (double)w[k] * (double)f[k];
  bias[e] = (float)(acc + (double)b1[o]);
}

// dW1[o][flat + e] = sum_n gh[n][o] * efeat[n][e]
__global__ __launch_bounds__(256) void embed_wgrad_kernel(const float *gh, const float *efeat, int B, int hidden, long pitch,
                                                        int flat, float *dw1) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= hidden * 32) return;
  const int o = t >> 5, e = t & 31;
  double acc = 0.0;
  for (int n = 0; n < B; ++n) acc += (double)gh[(long)n * hidden + o] * (double)efeat[(long)n * 32 + e];
  dw1[(long)o * pitch + flat + e] = (float)acc;
}

// d efeat[n][e] = sum_o gh[n][o] * W1[o][flat + e]
__global__ __launch_bounds__(256) void embed_dfeat_kernel(const float *gh, const float *w1, long pitch, int flat, int B,
                                                        int hidden, float *dfeat) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * 32) return;
  const int n = t >> 5, e = t & 31;
  double acc = 0.0;
  for (int o = 0; o < hidden; ++o) acc += (double)gh[(long)n * hidden + o] * (double)w1[(long)o * pitch + flat + e];
  dfeat[t] = (float)acc;
}

// demb[a][e] = sum over the samples with action a, in sample order (torch's embedding backward sums duplicates)
__global__ __launch_bounds__(32) void embed_scatter_kernel(const float *dfeat, const long long *actions, int B, float *demb) {
  const int a = blockIdx.x, e = threadIdx.x;
  double acc = 0.0;
  for (int n = 0; n < B; ++n)
    if (actions[n] == a) acc += (double)dfeat[(long)n * 32 + e];
  demb[a * 32 + e] = (float)acc;
}

hipError_t launch_embed_gather(const float *emb, const long long *actions, int B, int rows, float *out, int *err, hipStream_t s) {
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((B * 32 + 255) / 256)), dim3(256), 0, s, emb, actions, B, rows, out, err);
  return hipGetLastError();
}
hipError_t launch_embed_bias(const float *efeat, const float *w1, long pitch, int flat, const float *b1, int B, int hidden,
                             float *bias, hipStream_t s) {
  hipLaunchKernelGGL(embed_bias_kernel, dim3((unsigned)(((long)B * hidden + 255) / 256)), dim3(256), 0, s, efeat, w1, pitch, flat,
                     b1, B, hidden, bias);
  return hipGetLastError();
}
hipError_t launch_embed_backward(const float *gh, const float *efeat, const float *w1, long pitch, int flat, int B, int hidden,
                                 float *dw1, float *dfeat, hipStream_t s) {
  hipLaunchKernelGGL(embed_wgrad_kernel, dim3((unsigned)((hidden * 32 + 255) / 256)), dim3(256), 0, s, gh, efeat, B, hidden, pitch,
                     flat, dw1);
  hipLaunchKernelGGL(embed_dfeat_kernel, dim3((unsigned)((B * 32 + 255) / 256)), dim3(256), 0, s, gh, w1, pitch, flat, B, hidden,
                     dfeat);
  return hipGetLastError();
}
hipError_t launch_embed_scatter(const float *dfeat, const long long *actions, int B, int rows, float *demb, hipStream_t s) {
  hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)rows), dim3(32), 0, s, dfeat, actions, B, demb);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// g = dy * (y > 0)   (ReLU backward on a materialised activation);  optionally  g += add
__global__ __launch_bounds__(256) void relu_mask_kernel(const float *dy, const float *y, const float *add, long n4,
                                                      float *g) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const f32x4 d = reinterpret_cast<const f32x4 *>(dy)[e], v = reinterpret_cast<const f32x4 *>(y)[e];
  f32x4 o;
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = v[t] > 0.f ? d[t] : 0.f;
  if (add != nullptr) {
    const f32x4 a = reinterpret_cast<const f32x4 *>(add)[e];
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] += a[t];
  }
  reinterpret_cast<f32x4 *>(g)[e] = o;
}

hipError_t launch_relu_mask(const float *dy, const float *y, const float *add, long n, float *g, hipStream_t s) {
  const long n4 = n / 4;
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dy, y, add, n4, g);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_kernel(const float *a, const float *b, long n4, float *o) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  const f32x4 x = reinterpret_cast<const f32x4 *>(a)[e], y = reinterpret_cast<const f32x4 *>(b)[e];
  reinterpret_cast<f32x4 *>(o)[e] = x + y;
}

hipError_t launch_add(const float *a, const float *b, long n, float *o, hipStream_t s) {
  const long n4 = n / 4;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a, b, n4, o);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Stem tail in training: relu(gn(x)) + MaxPool(3, 2, 1) that also records WHICH window element won (first maximum in
// (kh, kw) scan order, as torch's max_pool2d backward does), and the backward that routes dPool to those elements.
__global__ __launch_bounds__(256) void gn_relu_maxpool_idx_kernel(const float *x, const float *scale, const float *shift,
                                                                int B, int H, int W, int C, int Ho, int Wo, float *out,
                                                                unsigned char *idx) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Ho * Wo * C;
  if (g >= total) return;
  const int c = (int)(g % C);
  long r = g / C;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int n = (int)(r / Ho);
  const float sc = scale[(long)n * C + c], sh = shift[(long)n * C + c];
  float m = -INFINITY;
  int best = 0;
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const float v = fmaxf(__builtin_fmaf(x[(((long)n * H + hi) * W + wi) * C + c], sc, sh), 0.f);
      if (v > m) {
        m = v;
        best = kh * 3 + kw;
      }
    }
  }
  out[g] = m;
  idx[g] = (unsigned char)best;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *dpool, const unsigned char *idx, int B, int H,
                                                        int W, int C, int Ho, int Wo, float *dact) {
  // gather form (deterministic): input pixel (hi, wi) collects dPool of every window that elected it; four channels per
  // thread (16-byte gradient loads / stores, 4-byte index loads)
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const int Q = C >> 2;
  const long total = (long)B * H * W * Q;
  if (g >= total) return;
  const int c = 4 * (int)(g % Q);
  long r = g / Q;
  const int wi = (int)(r % W);
  r /= W;
  const int hi = (int)(r % H);
  const int n = (int)(r / H);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int t = hi + 1 - kh;          // 2*ho = hi + 1 - kh
    if (t < 0 || (t & 1)) continue;
    const int ho = t >> 1;
    if (ho >= Ho) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int u = wi + 1 - kw;
      if (u < 0 || (u & 1)) continue;
      const int wo = u >> 1;
      if (wo >= Wo) continue;
      const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
      const unsigned ix = *reinterpret_cast<const unsigned *>(idx + o);
      const float4 dp = *reinterpret_cast<const float4 *>(dpool + o);
      const unsigned k = (unsigned)(kh * 3 + kw);
      if ((ix & 0xffu) == k) acc[0] += dp.x;
      if (((ix >> 8) & 0xffu) == k) acc[1] += dp.y;
      if (((ix >> 16) & 0xffu) == k) acc[2] += dp.z;
      if ((ix >> 24) == k) acc[3] += dp.w;
    }
  }
  reinterpret_cast<float4 *>(dact)[g] = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

hipError_t launch_maxpool_train(const float *x, const float *scale, const float *shift, int B, int H, int W, int C,
                                float *out, unsigned char *idx, hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * Ho * Wo * C;
  hipLaunchKernelGGL(gn_relu_maxpool_idx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, scale, shift, B,
                     H, W, C, Ho, Wo, out, idx);
  return hipGetLastError();
}

hipError_t launch_maxpool_bwd(const float *dpool, const unsigned char *idx, int B, int H, int W, int C, float *dact,
                              hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * H * W * (C / 4);          // C is a multiple of 32
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpool, idx, B, H, W, C, Ho,
                     Wo, dact);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// out[c] = sum_r x[r][c]  (bias gradients), fixed order, fp64
// one wave per column: lane l sums rows l, l + 64, ... in fp64, fixed xor tree (bias gradients of the Linear layers)
__global__ __launch_bounds__(256) void colsum_kernel(const float *x, int rows, int cols, int ld, float *out) {
  const int c = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= cols) return;
  double a = 0.0;
  for (int r = lane; r < rows; r += 64) a += (double)x[(long)r * ld + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) out[c] = (float)a;
}
hipError_t launch_colsum(const float *x, int rows, int cols, int ld, float *out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, s, x, rows, cols, ld, out);
  return hipGetLastError();
}

// dst[r][0..ldd) = src[r][0..cols) zero padded (grad_out [B,3] -> [B,8] so it can feed the conv kernel)
__global__ __launch_bounds__(256) void padcopy_kernel(const float *src, int rows, int cols, int ldd, float *dst) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)rows * ldd) return;
  const int c = (int)(e % ldd);
  const long r = e / ldd;
  dst[e] = c < cols ? src[r * cols + c] : 0.f;
}

hipError_t launch_padcopy(const float *src, int rows, int cols, int ldd, float *dst, hipStream_t s) {
  const long total = (long)rows * ldd;
  hipLaunchKernelGGL(padcopy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, rows, cols, ldd, dst);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// loss = sum_d mean_i (target - pred)^2  (vo_cnn_engine.py:146-194, unit weights);  grad = dloss/dpred
__global__ __launch_bounds__(256) void mse_loss_kernel(const float *pred, const float *target, int B, int D, float *loss,
                                                     float *grad) {
  __shared__ double red[256];
  double a = 0.0;
  for (int e = threadIdx.x; e < B * D; e += 256) {
    const float d = target[e] - pred[e];
    a += (double)d * (double)d;
    if (grad) grad[e] = -2.0f * d / (float)B;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) *loss = (float)(red[0] / (double)B);
}

hipError_t launch_mse_loss(const float *pred, const float *target, int B, int D, float *loss, float *grad, hipStream_t s) {
  hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(256), 0, s, pred, target, B, D, loss, grad);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Regression loss with per-element coefficients: loss = sum_e coef[e] (target[e] - pred[e])^2, grad = -2 coef (t - p).
// coef folds everything _compute_loss multiplies or divides by (vo_cnn_engine.py:146-194: loss_weights, dz_regress_masks,
// 1/len of the data-type subset the mean is taken over, ...geo_invariance_engine.py:690-740).
__global__ __launch_bounds__(256) void mse_loss_coef_kernel(const float *pred, const float *target, const float *coef, int n,
                                                          float *loss, float *grad) {
  __shared__ double red[256];
  double a = 0.0;
  for (int e = threadIdx.x; e < n; e += 256) {
    const float d = target[e] - pred[e];
    const float c = coef[e];
    a += (double)c * (double)d * (double)d;
    if (grad) grad[e] = -2.0f * c * d;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) *loss = (float)red[0];
}

hipError_t launch_mse_loss_coef(const float *pred, const float *target, const float *coef, int n, float *loss, float *grad,
                                hipStream_t s) {
  hipLaunchKernelGGL(mse_loss_coef_kernel, dim3(1), dim3(256), 0, s, pred, target, coef, n, loss, grad);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// _compute_geo_invariance_inverse_loss (vo_cnn_regression_geo_invariance_engine.py:367-449).  deltas [2P,3] alternate
// a = cur_rel_to_prev_i, b = prev_rel_to_cur_i; with R(yaw_b) = [[cos, sin], [-sin, cos]]:
//   L = mean_i (yaw_a + yaw_b)^2 + mean_{i,k} m_ik (pos_b + R pos_a)_k^2,   m_i1 = 0 when action_i == MOVE_FORWARD.
// out[0] = weight * L; out[1] = mean |yaw_a + yaw_b|; out[2..3] = mean_i sqrt(m d^2) per column (the three logged values);
// grad [2P,3] = weight * dL/ddeltas.  One block, fixed-order fp64 reduction.
__global__ __launch_bounds__(256) void geo_inverse_loss_kernel(const float *deltas, const int *actions, int P, int move_forward,
                                                             float weight, float *out, float *grad) {
  __shared__ double red[4][256];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const float ip = 1.f / (float)P;
  for (int i = threadIdx.x; i < P; i += 256) {
    const float *a = deltas + (size_t)(2 * i) * 3, *b = a + 3;
    const float xa = a[0], za = a[1], ya = a[2], xb = b[0], zb = b[1], yb = b[2];
    const float m = actions[2 * i] == move_forward ? 0.f : 1.f;
    const float c = cosf(yb), s = sinf(yb);
    const float r = ya + yb;
    const float d0 = xb + (c * xa + s * za);
    const float d1 = zb + (-s * xa + c * za);
    acc[0] += (double)r * r + 0.5 * ((double)d0 * d0 + (double)m * d1 * d1);
    acc[1] += fabs((double)r);
    acc[2] += fabs((double)d0);
    acc[3] += (double)m * fabs((double)d1);
    if (grad) {
      const float g0 = weight * d0 * ip, g1 = weight * m * d1 * ip, gr = weight * 2.f * r * ip;
      float *ga = grad + (size_t)(2 * i) * 3, *gb = ga + 3;
      ga[0] = g0 * c - g1 * s;
      ga[1] = g0 * s + g1 * c;
      ga[2] = gr;
      gb[0] = g0;
      gb[1] = g1;
      gb[2] = gr + g0 * (-s * xa + c * za) + g1 * (-c * xa - s * za);
    }
  }
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4 && out)
    out[threadIdx.x] = (float)((threadIdx.x == 0 ? (double)weight : 1.0) * red[threadIdx.x][0] / (double)P);
}

hipError_t launch_geo_inverse_loss(const float *deltas, const int *actions, int P, int move_forward, float weight, float *out,
                                   float *grad, hipStream_t s) {
  hipLaunchKernelGGL(geo_inverse_loss_kernel, dim3(1), dim3(256), 0, s, deltas, actions, P, move_forward, weight, out, grad);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// torch.optim.Adam (weight_decay 0, amsgrad off) on flat buffers; step counts from 1
__global__ __launch_bounds__(256) void adam_kernel(float *p, const float *g, float *m, float *v, long n, float lr, float b1,
                                                 float b2, float eps, float bc1, float sqrt_bc2) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float gi = g[e];
  const float mi = b1 * m[e] + (1.f - b1) * gi;
  const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
  m[e] = mi;
  v[e] = vi;
  p[e] -= (lr / bc1) * mi / (sqrtf(vi) / sqrt_bc2 + eps);
}

hipError_t launch_adam(float *p, const float *g, float *m, float *v, long n, float lr, float b1, float b2, float eps,
                       int step, hipStream_t s) {
  const float bc1 = 1.f - (float)pow((double)b1, (double)step);
  const float sq2 = (float)sqrt(1.0 - pow((double)b2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, bc1,
                     sq2);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// dst[i] = map[i] > 0 ? src[map[i] - 1] : 0   (re-pack kernel operands from the flat parameter buffer on device)
__global__ __launch_bounds__(256) void gather_kernel(const float *src, const int *map, long n, float *dst) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int k = map[e];
  dst[e] = k > 0 ? src[k - 1] : 0.f;
}

// All re-pack maps of a model in ONE launch (they were ~50 launches of a few microseconds each after every Adam step):
// element e of the concatenation belongs to the segment with the largest start <= e.
__global__ __launch_bounds__(256) void gather_all_kernel(const float *src, const GatherSeg *segs, int nseg, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].start <= e) lo = mid; else hi = mid - 1;
  }
  const GatherSeg sg = segs[lo];
  const long i = e - sg.start;
  const int k = sg.map[i];
  sg.dst[i] = k > 0 ? src[k - 1] : 0.f;
}

hipError_t launch_gather_all(const float *src, const GatherSeg *segs, int nseg, long total, hipStream_t s) {
  hipLaunchKernelGGL(gather_all_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, segs, nseg, total);
  return hipGetLastError();
}

hipError_t launch_gather(const float *src, const int *map, long n, float *dst, hipStream_t s) {
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, map, n, dst);
  return hipGetLastError();
}

// whitening table of the fused stem from device-resident running statistics (reference channel order -> stem order)
__global__ void whiten_table_kernel(const float *mean, const float *var, const int *ref_of_new, const int *tensor_of_new,
                                    int CPL, float *sc, float *sh) {
  const int c = threadIdx.x;
  if (c >= CPL) return;
  const int r = ref_of_new[c];
  float a = 0.f, b = 0.f;
  if (r >= 0) {
    const double sd = sqrt(fmax((double)var[r], 1e-2));
    const double div = tensor_of_new[c] == 0 ? 255.0 : 1.0;
    a = (float)(1.0 / (div * sd));
    b = (float)(-(double)mean[r] / sd);
  }
  sc[c] = a;
  sh[c] = b;
}

hipError_t launch_whiten_table(const float *mean, const float *var, const int *ref_of_new, const int *tensor_of_new, int CPL,
                               float *sc, float *sh, hipStream_t s) {
  hipLaunchKernelGGL(whiten_table_kernel, dim3(1), dim3(64), 0, s, mean, var, ref_of_new, tensor_of_new, CPL, sc, sh);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Per-channel moments of the assembled input for RunningMeanAndVar's train-mode update (running_mean_and_var.py:24-38):
//   out[c] = mean over (n, pixel) of (x_c - center_c)^pw,  x_c in the reference channel order, rgb / 255.
// Two stages (per-block partial, then fixed-order fp64 sum).
// Stage 1: blockIdx.y = observation tensor; a thread owns one 2-channel piece of the tensor's pixels and walks the
// tensor with fully coalesced 8-byte loads (the first version read one channel per block: 30 strided passes).
__global__ __launch_bounds__(256) void moments_partial_kernel(const MomentsArgs a, int C, double *part) {
  __shared__ double red[2][256];
  const int t = blockIdx.y;
  const float *base = a.src[t];
  const int nch = a.nch[t];
  if (base == nullptr || nch <= 0 || (int)blockIdx.x >= a.nblk[t]) return;   // blocks per tensor ~ its channel count
  const int np = nch >> 1;                                 // 2-channel pieces per pixel (nch is even: 6, 2, 20, 2)
  const long T = (long)a.nblk[t] * 256;
  const long stride = T - T % np;                          // a multiple of np: the piece of a thread never changes
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int piece = (int)(gid % np);
  const bool rgb = t == 0;
  // reference channels of this thread's two tensor channels
  int c0 = -1, c1 = -1;
  for (int c = 0; c < C; ++c) {
    if (a.tensor[c] == t && a.ch[c] == 2 * piece) c0 = c;
    if (a.tensor[c] == t && a.ch[c] == 2 * piece + 1) c1 = c;
  }
  const float ctr0 = (a.center && c0 >= 0) ? a.center[c0] : 0.f, ctr1 = (a.center && c1 >= 0) ? a.center[c1] : 0.f;
  double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;           // pw 3: first AND second moment about `center` in one pass
  if (gid < stride) {
    const long total = a.npix * np;
    constexpr int U = 4;                                   // independent loads in flight per thread
    for (long e = gid; e < total; e += U * stride) {
      f32x2 v[U];
#pragma unroll
      for (int k = 0; k < U; ++k)
        v[k] = e + k * stride < total ? *reinterpret_cast<const f32x2 *>(base + 2 * (e + k * stride)) : f32x2{0.f, 0.f};
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (e + k * stride >= total) break;
        const float d0 = (rgb ? v[k][0] / 255.0f : v[k][0]) - ctr0, d1 = (rgb ? v[k][1] / 255.0f : v[k][1]) - ctr1;
        const double e0 = (double)d0, e1 = (double)d1;
        if (a.pw == 2) {
          s0 += e0 * e0;
          s1 += e1 * e1;
        } else {
          s0 += e0;
          s1 += e1;
        }
        if (a.pw == 3) {
          q0 += e0 * e0;
          q1 += e1 * e1;
        }
      }
    }
  }
  for (int pass = 0; pass < (a.pw == 3 ? 2 : 1); ++pass) {
    if (pass) __syncthreads();
    red[0][threadIdx.x] = pass ? q0 : s0;
    red[1][threadIdx.x] = pass ? q1 : s1;
    __syncthreads();
    // thread j < nch sums, in a fixed order, the lanes that own channel j's piece
    if ((int)threadIdx.x < nch) {
      const int q = threadIdx.x >> 1, e = threadIdx.x & 1;
      const int first = (int)(((long)q - (long)blockIdx.x * 256 % np + np) % np);   // first tid of this block with piece q
      double s = 0.0;
      for (int k = first; k < 256; k += np) s += red[e][k];
      int c = -1;
      for (int cc = 0; cc < C; ++cc)
        if (a.tensor[cc] == t && a.ch[cc] == (int)threadIdx.x) c = cc;
      if (c >= 0) part[((long)pass * C + c) * gridDim.x + blockIdx.x] = s;
    }
  }
}

// one wave per output row (channel, or channel + C for the second moments): fixed-order fp64 sum of its tensor's blocks
__global__ __launch_bounds__(64) void moments_final_kernel(const MomentsArgs a, const double *part, int nbx, int C, float *out) {
  const int row = blockIdx.x, c = row % C;
  const int nb = a.nblk[a.tensor[c]];
  double s = 0.0;
  for (int k = threadIdx.x; k < nb; k += 64) s += part[(long)row * nbx + k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) out[row] = (float)(s / (double)a.npix);
}

// RunningMeanAndVar's train-mode update in one launch (single process; running_mean_and_var.py:41-60): batch mean and the
// variance about it from the one-pass moments e1 = E[x - c], e2 = E[(x - c)^2] about the old running mean c, then Chan's merge.
__global__ void rmv_merge_kernel(const float *m12, int C, float nb, float *mean, float *var, float *count) {
  const int c = threadIdx.x;
  const float cnt = *count;
  __syncthreads();                                        // (everybody has read the old count)
  if (c < C) {
    const double ctr = (double)mean[c], e1 = (double)m12[c], e2 = (double)m12[C + c];
    const float new_mean = (float)(ctr + e1);
    const double delta = (double)new_mean - ctr;
    const float new_var = (float)(e2 - 2.0 * delta * e1 + delta * delta);
    const float m_a = var[c] * cnt, m_b = new_var * nb;
    const float dm = new_mean - mean[c];
    const float M2 = m_a + m_b + dm * dm * cnt * nb / (cnt + nb);
    var[c] = M2 / (cnt + nb);
    mean[c] = (cnt * mean[c] + nb * new_mean) / (cnt + nb);
  }
  if (c == 0) *count = cnt + nb;
}

hipError_t launch_rmv_merge(const float *m12, int C, int B, float *mean, float *var, float *count, hipStream_t s) {
  hipLaunchKernelGGL(rmv_merge_kernel, dim3(1), dim3(((C + 63) / 64) * 64), 0, s, m12, C, (float)B, mean, var, count);
  return hipGetLastError();
}

hipError_t launch_moments(const MomentsArgs &a0, int C, double *part, float *out, hipStream_t s) {
  MomentsArgs a = a0;
  int maxch = 1;
  for (int t = 0; t < 4; ++t)
    if (a.src[t] != nullptr && a.nch[t] > maxch) maxch = a.nch[t];
  for (int t = 0; t < 4; ++t) {
    a.nblk[t] = a.src[t] != nullptr && a.nch[t] > 0 ? (MOMENTS_BLOCKS * a.nch[t] + maxch - 1) / maxch : 0;
    if (a.nblk[t] > MOMENTS_BLOCKS) a.nblk[t] = MOMENTS_BLOCKS;
  }
  hipLaunchKernelGGL(moments_partial_kernel, dim3(MOMENTS_BLOCKS, 4), dim3(256), 0, s, a, C, part);
  const int rows = a.pw == 3 ? 2 * C : C;               // pw 3: out[0..C) first moments, out[C..2C) second moments
  hipLaunchKernelGGL(moments_final_kernel, dim3(rows), dim3(64), 0, s, a, part, MOMENTS_BLOCKS, C, out);
  return hipGetLastError();
}

}  // namespace pnvo
