// wgrad_x3.hip — weight gradient of the 3x3 stride-1 convs of the residual stages on the bf16 matrix cores, gfx950 only.
//
//   dW[kh][kw][ci][co] = sum over (n, y, x) of  X[n, y + kh - 1, x + kw - 1, ci] * dY[n, y, x, co]
// (backward of resnet.py:29-55's convs in the training step, vo_cnn_regression_geo_invariance_engine.py:855-901; 13 of the 16
// 3x3 convs — the three strided ones keep the fp32 kernels of train_kernels.hip).  As a GEMM the contraction index is the PIXEL:
// D[ci][co] += A[ci][k = pixel] * B[k = pixel][co] on v_mfma_f32_32x32x16_bf16, so every lane wants 8 consecutive pixels of ONE
// channel in its registers.  NHWC makes exactly that a coalesced access when the lane index is the channel: a dword load per
// pixel reads 32 consecutive channels per half wave (one 128-byte segment), and the 8 loads of a lane ARE its fragment — no LDS,
// no transposition.  Lane half h takes the 8 output columns x0 + 8h .. x0 + 8h + 7 of a 16-column strip; the wave walks down the
// rows of the strip with a rolling three-row window of X (each row is loaded once: 10 pixels = 8 + the two halo columns), and the
// three kw taps are register views of that row: kw = 0 / 2 are the packed pairs as loaded, kw = 1 is four v_alignbit of them.
//
// Arithmetic = conv_x3.hip's: both operands split into three bf16 pieces (hi + mid + lo == the float32 value), six product terms
// per MAC, each exact in the float32 accumulator; the dropped three are below 2^-23 of the product.  (Gradients span too many
// binades for the two-piece float16 form of the forward.)  54 MFMAs per row step (9 taps x 6 terms) against ~220 vector
// instructions (loads, GroupNorm + ReLU of the producer recomputed, the splits): matrix-pipe bound, one wave per SIMD with its
// 9 x 16 accumulators.
//
// Work split: workgroup = (ci-tile, co-tile, chunk of units), unit = (image, 16-column strip, row segment); the four waves of a
// workgroup take consecutive units of its chunk and meet in LDS in wave order (one partial per workgroup); wgrad_reduce_kernel
// (train_kernels.hip) sums the workgroups' partials in fp64 in a fixed order: bit-reproducible.
#include <type_traits>

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
__device__ __forceinline__ unsigned clampb(long bytes) {
  return (unsigned)(bytes > 0x7FFFF000L ? 0x7FFFF000L : (bytes < 0 ? 0 : bytes));
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 r = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// three bf16 pieces of a pair of float32 values
__device__ __forceinline__ void split3(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2) {
  p0 = pack2(a, b);
  const float ra = a - lo_f(p0), rb = b - hi_f(p0);
  p1 = pack2(ra, rb);
  p2 = pack2(ra - lo_f(p1), rb - hi_f(p1));
}

// two float16 pieces of a pair of float32 values (22 significant bits; NP = 2)
__device__ __forceinline__ void split2(float a, float b, unsigned &p0, unsigned &p1) {
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  p0 = __builtin_bit_cast(unsigned, h);
  const f16x2 m = __builtin_convertvector(f32x2{a - (float)h[0], b - (float)h[1]}, f16x2);
  p1 = __builtin_bit_cast(unsigned, m);
}
template <int NP>
__device__ __forceinline__ void splitn(float a, float b, unsigned (&q)[NP]) {
  if constexpr (NP == 2)
    split2(a, b, q[0], q[1]);
  else
    split3(a, b, q[0], q[1], q[2]);
}

template <int NP>
struct XRow {          // one X row of the window: 10 pixels (columns cb - 1 .. cb + 8) as 5 packed pairs per piece
  unsigned p[NP][5];
};
}  // namespace

// NP = 3: three bf16 pieces per operand, six exact product terms.  NP = 2 (option train_pieces = 2): two float16 pieces, three
// terms — X is a forward activation (inside float16's range by the forward's own bound), dY is multiplied by 2^(14 - e) from its
// tensor's absolute maximum before the split (gn_bwd_apply tracks it) and the accumulators are divided by it at the end (exact).
template <int MODE, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad3_x3_kernel(const WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float red[3 * 64 * 16];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int pair = (int)blockIdx.x / p.chunks, chunk = (int)blockIdx.x - pair * p.chunks;
  const int cit = pair % p.ci_tiles, cot = pair / p.ci_tiles;
  const int ci = cit * 32 + i, co = cot * 32 + i;
  const int H = p.H, W = p.W;
  const int strips = (W + 15) >> 4, rsegs = p.xr_rsegs, rows_seg = (H + rsegs - 1) / rsegs;
  const int nunits = p.B * strips * rsegs;
  const int wid = chunk * 4 + wave, nw = p.chunks * 4;
  const int u0 = (int)((long)nunits * wid / nw), u1 = (int)((long)nunits * (wid + 1) / nw);

  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void *)p.dy, 0, clampb((long)p.B * H * W * p.DYC * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, clampb((long)p.B * H * W * p.CIN * 4), 0x00020000);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dmul = 1.f, ddiv = 1.f;
  if (NP == 2 && p.dy_absmax != nullptr) {
    unsigned mb = p.dy_absmax[lane * 16];                               // maximum of the 64 slots: one per lane, butterfly
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, d));
    const int e = mb != 0u ? (int)((mb >> 23) & 0xffu) - 126 : 14;
    dmul = __builtin_bit_cast(float, (unsigned)(14 - e + 127) << 23);
    ddiv = __builtin_bit_cast(float, (unsigned)(e - 14 + 127) << 23);
  }

  for (int u = u0; u < u1; ++u) {
    int q = u;
    const int rseg = q % rsegs;
    q /= rsegs;
    const int sidx = q % strips;
    const int n = q / strips;
    const int ya = rseg * rows_seg, yb = min(H, ya + rows_seg);
    if (ya >= yb) continue;
    const int cb = sidx * 16 + 8 * h;                       // first output column of this lane half
    float sc = 1.f, sh = 0.f;
    if (MODE == 1) {
      sc = p.in_scale[(long)n * p.CIN + ci];
      sh = p.in_shift[(long)n * p.CIN + ci];
    }
    const unsigned xbase = (unsigned)(((long)n * H * W) * p.CIN + ci) * 4u;     // (byte offsets fit 32 bits: clampb)
    const unsigned dbase = (unsigned)(((long)n * H * W) * p.DYC + co) * 4u;

    // raw float32 loads of one X row (10 pixels) / one dY row (8 pixels).  Branch-free: an invalid pixel gets bit 31 of its byte
    // offset set — beyond the buffer's num_records, so the load returns 0 — by sign-bit arithmetic (a select on a comparison here
    // makes the compiler wrap every address computation in its own exec-mask branch: ~80 basic blocks per step, nothing overlaps)
    auto inv31 = [](int v, int n) -> unsigned { return (unsigned)((v | (n - 1 - v)) >> 31) << 31; };   // 0x80000000 unless 0 <= v < n
    auto loadX = [&](int yy, float (&raw)[10]) {
      const unsigned rinv = inv31(yy, H);
      const unsigned rowoff = xbase + (unsigned)(yy * W) * (unsigned)(p.CIN * 4);
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int xx = cb - 1 + k;
        raw[k] = bload(rx, (rowoff + (unsigned)xx * (unsigned)(p.CIN * 4)) | rinv | inv31(xx, W));
      }
    };
    auto loadD = [&](int y, float (&raw)[8]) {
      const unsigned rinv = inv31(y, yb);
      const unsigned rowoff = dbase + (unsigned)(y * W) * (unsigned)(p.DYC * 4);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int xx = cb + k;
        raw[k] = bload(rdy, (rowoff + (unsigned)xx * (unsigned)(p.DYC * 4)) | rinv | inv31(xx, W));
      }
    };
    // the producer's GroupNorm + ReLU (MODE 1), zero padding AFTER it, then the three-piece split
    auto convX = [&](int yy, const float (&raw)[10], XRow<NP> &row) {
      const unsigned rinv = inv31(yy, H);
      float v[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        float t = raw[k];
        if (MODE == 1) {
          const unsigned bad = rinv | inv31(cb - 1 + k, W);
          t = fmaxf(__builtin_fmaf(t, sc, sh), 0.f);
          t = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) & ~(unsigned)((int)bad >> 31));   // 0 outside the image
        }
        v[k] = t;
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        unsigned q[NP];
        splitn<NP>(v[2 * j], v[2 * j + 1], q);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) row.p[pc][j] = q[pc];
      }
    };
    // A fragment (8 bf16 = 4 dwords) of a window row for tap column kw: pixels kw .. kw + 7 of the 10 held
    auto afrag = [&](const XRow<NP> &row, int pc, int kw) -> u32x4 {
      const unsigned *r = row.p[pc];
      if (kw == 0) return u32x4{r[0], r[1], r[2], r[3]};
      if (kw == 2) return u32x4{r[1], r[2], r[3], r[4]};
      return u32x4{__builtin_amdgcn_alignbit(r[1], r[0], 16), __builtin_amdgcn_alignbit(r[2], r[1], 16),
                   __builtin_amdgcn_alignbit(r[3], r[2], 16), __builtin_amdgcn_alignbit(r[4], r[3], 16)};
    };
    // the 18 MFMAs of one kernel row kh: three kw taps x six product terms, against dY pieces d[3][4]
    auto mfma_row = [&](const XRow<NP> &row, int kh, const unsigned (&d)[NP][4]) {
      const u32x4 b0 = u32x4{d[0][0], d[0][1], d[0][2], d[0][3]}, b1 = u32x4{d[1][0], d[1][1], d[1][2], d[1][3]};
      if constexpr (NP == 2) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const u32x4 a0 = afrag(row, 0, kw), a1 = afrag(row, 1, kw);
          f32x16 &c = acc[kh * 3 + kw];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b0), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b1), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), c, 0, 0, 0);
        }
        return;
      }
      const u32x4 b2 = u32x4{d[NP - 1][0], d[NP - 1][1], d[NP - 1][2], d[NP - 1][3]};
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const u32x4 a0 = afrag(row, 0, kw), a1 = afrag(row, 1, kw), a2 = afrag(row, NP - 1, kw);
        f32x16 &c = acc[kh * 3 + kw];
        // smallest terms first: a1 b1, a2 b0, a0 b2, a1 b0, a0 b1, a0 b0
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, b0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b2), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), c, 0, 0, 0);
      }
    };

    XRow<NP> win[3];
    unsigned dpk[NP][4];
    float xr[10], dr[8];
    auto convD = [&](const float (&raw)[8], unsigned (&d)[NP][4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned q[NP];
        if constexpr (NP == 2)
          splitn<NP>(raw[2 * j] * dmul, raw[2 * j + 1] * dmul, q);
        else
          splitn<NP>(raw[2 * j], raw[2 * j + 1], q);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) d[pc][j] = q[pc];
      }
    };
    // prologue of the unit: rows ya - 1, ya, ya + 1 converted; row ya + 2 and dY row ya in flight
    {                                                       // (all three rows in flight at once: one memory latency, not three)
      float x0[10], x1[10];
      loadX(ya - 1, x0);
      loadX(ya, x1);
      loadX(ya + 1, xr);
      loadD(ya, dr);
      convX(ya - 1, x0, win[0]);
      convX(ya, x1, win[1]);
      convX(ya + 1, xr, win[2]);
      loadX(ya + 2, xr);
    }
    // One output row per step.  Kernel row 0 (the window's oldest row) is multiplied first; its slot then takes row y + 2,
    // converted while the 36 MFMAs of kernel rows 1 and 2 run, and the loads of row y + 3 / dY row y + 1 fly behind them.  The
    // window's roles rotate through three unrolled copies (no register moves); a segment runs whole groups of three steps — rows
    // past its end read dY = 0 and add nothing (wgrad_x3_plan makes segments multiples of three rows where the map allows).
    auto step = [&](int y, auto R0, auto R1, auto R2) {
      convD(dr, dpk);                                       // dY row y (loaded during the previous step)
      loadD(y + 1, dr);
      mfma_row(win[decltype(R0)::value], 0, dpk);
      convX(y + 2, xr, win[decltype(R0)::value]);           // row y + 2 replaces row y - 1
      loadX(y + 3, xr);                                     // (rows past the image: offsets out of range, zeros)
      mfma_row(win[decltype(R1)::value], 1, dpk);
      mfma_row(win[decltype(R2)::value], 2, dpk);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
    for (int y = ya; y < yb; y += 3) {
      step(y, I0{}, I1{}, I2{});
      step(y + 1, I1{}, I2{}, I0{});
      step(y + 2, I2{}, I0{}, I1{});
    }
  }

  // the four waves of the workgroup meet in LDS, tap by tap, in wave order; wave 0 writes the workgroup's partial
  // (C/D layout: col j (= co) = lane & 31, row i (= ci) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
  if (NP == 2) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= ddiv;
  }
  const long unit = (long)pair * p.chunks + chunk;
  float *dst = p.partial + unit * 9 * 1024;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4 *>(red + (((wave - 1) * 4 + rq) * 64 + lane) * 4) =
            f32x4{acc[t][4 * rq], acc[t][4 * rq + 1], acc[t][4 * rq + 2], acc[t][4 * rq + 3]};
    }
    __syncthreads();
    if (wave == 0) {
      f32x16 s = acc[t];
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x4 o = *reinterpret_cast<const f32x4 *>(red + ((w * 4 + rq) * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[4 * rq + e] += o[e];
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        dst[((long)t * 32 + row) * 32 + i] = s[r];
      }
    }
  }
}

// plan: false = shape outside this kernel
bool wgrad_x3_plan(WgradArgs &a) {
  if (a.mode == 2 || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.H != a.Ho || a.W != a.Wo) return false;
  if (a.CIN % 32 || a.COUT % 32 || a.DYC < a.COUT) return false;
  if ((long)a.B * a.H * a.W * (long)(a.CIN > a.DYC ? a.CIN : a.DYC) * 4 >= 0x7FFFF000L) return false;   // 32-bit buffer offsets
  a.TG = 9;
  a.groups = 1;
  a.ci_tiles = a.CIN / 32;
  a.pairs = a.ci_tiles * (a.COUT / 32);
  // one wave per SIMD: ~1024 waves per launch, four per workgroup
  int wgs = 256 / a.pairs;
  if (wgs < 1) wgs = 1;
  a.chunks = wgs;
  const int strips = (a.W + 15) / 16;
  // units per wave >= ~4 keeps the split even; a unit's prologue costs two extra row loads, so row segments stay >= 4 rows
  long want = 4L * wgs * 4;
  int rsegs = (int)((want + (long)a.B * strips - 1) / ((long)a.B * strips));
  const int max_rsegs = (a.H + 3) / 4;
  if (rsegs > max_rsegs) rsegs = max_rsegs;
  if (rsegs < 1) rsegs = 1;
  while (rsegs > 1 && ((a.H + rsegs - 1) / rsegs) % 3 != 0) --rsegs;     // whole groups of three rows per segment where the map allows
  a.xr_rsegs = rsegs;
  if ((long)a.B * strips * rsegs < (long)wgs * 4) {        // fewer units than waves (tiny batches): fewer workgroups
    a.chunks = (int)(((long)a.B * strips * rsegs + 3) / 4);
    if (a.chunks < 1) a.chunks = 1;
  }
  a.pix_per_chunk = 0;
  a.lds3 = 6;
  return true;
}

hipError_t launch_wgrad_x3(const WgradArgs &a, hipStream_t s) {
  dim3 grid((unsigned)(a.pairs * a.chunks));
  if (a.np == 2 && a.mode == 1)
    hipLaunchKernelGGL((wgrad3_x3_kernel<1, 2>), grid, dim3(256), 0, s, a);
  else if (a.np == 2)
    hipLaunchKernelGGL((wgrad3_x3_kernel<0, 2>), grid, dim3(256), 0, s, a);
  else if (a.mode == 1)
    hipLaunchKernelGGL((wgrad3_x3_kernel<1, 3>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((wgrad3_x3_kernel<0, 3>), grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace pnvo
