// stem_lds.hip — the 7x7 stride-2 stem (resnet.py:156-163) with input assembly + /255 + whitening (vo_cnn.py:110-176)
// fused in, on the CDNA4 fp32 matrix cores (gfx950 only).
//
// Why a dedicated kernel: the stem is 57.5 % of the path's FLOPs and every input pixel feeds 49/4 output pixels.
// Served from L2/HBM that reuse costs ~17x the algorithmic input in fabric traffic (measured with rocprofv3:
// FETCH_SIZE 34 GB per dispatch for 2 GB of input, L2 hit rate 37 % — profiles/r1_stem_gather_pmc.md), so the
// kernel ran at the fabric's speed, not the matrix cores'.  Here each workgroup stages its input patch ONCE:
//
//   * persistent workgroups of 8 waves; a workgroup walks 4-row x 16-column tiles of output pixels.  The
//     (2*4+5) x (2*16+5) input patch of a tile is gathered from the four observation tensors (rgb | depth |
//     discretized_depth | top_down_view) in 2-channel pieces, whitened (x*sc[c]+sh[c]; zero outside the image = the
//     conv's zero padding, applied after whitening) and written to LDS as [13][37][CPL] fp32 with a 16-byte-slot XOR
//     swizzle (ds_read_b128 of 16 consecutive output pixels at stride 2 would otherwise be a 16-way bank conflict);
//   * 61.6 KB of LDS per workgroup -> 2 workgroups = 16 waves per CU = 4 waves per SIMD: measured, the K loop
//     reaches 68 % of the MFMA pipe with 1 wave/SIMD, 83 % with 2 (profiles/r1_stem_ablation.md), so the patch is
//     shared by TWO wave groups that split K (stage parity) and are summed through LDS in a fixed order;
//   * wave (row r, half h) owns output row r x 16 pixels x all Cout for its half of the K stages:
//     v_mfma_f32_16x16x4_f32, A from LDS (one ds_read_b128 per lane and stage, immediate offsets: the tap loop is
//     fully unrolled), B = pre-packed weights streamed from L2 through a 4-stage register ring; no barrier in the loop;
//   * epilogue: raw conv output + deterministic per-tile partial (sum, sumsq) per channel for the GroupNorm that
//     follows (fixed order, no atomics).
#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 16;                   // output tile
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;  // input patch 13 x 37
constexpr int NTHREADS = 512;

__device__ __forceinline__ f32x4 wload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// PAIRED (the device-side input-contract repair and the handle's fallback stem, pnvo_api.hip pnvo_run_stem): the launch stands in
// for a stem on the 16-bit matrix cores (stem_mx.hip / stem_rs.hip / stem_dd.hip) — same raw output, GroupNorm partials in THEIR
// slot layout (one slot per 8 x 16 tile = two vertically adjacent tiles of this kernel, summed upper + lower), and PREDICATED:
// with `only_if` set the whole launch returns at once unless *only_if != 0 (the flag those stems raise when a value breaks the
// observation contract), so the decision to redo the stem on float32 operands is taken on the device, not by a waiting host.
template <int NT16, int CPL, bool PAIRED = false>   // Cout / 16; channels per pixel in LDS (16 or 32)
__global__ __launch_bounds__(NTHREADS, 4) void stem_lds_kernel(const StemArgs p) {
  if (PAIRED && p.only_if != nullptr) {
    if (*reinterpret_cast<const volatile int *>(p.only_if) == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.publish != nullptr) *reinterpret_cast<volatile int *>(p.publish) = 1;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int G = CPL >> 2;         // 16-byte slots per pixel (4 or 8)
  constexpr int J16 = CPL >> 4;       // 16-channel K groups per tap (1 or 2)
  constexpr int S = 49 * J16;         // K stages; wave half h takes stages s = 2u + h
  constexpr int COUT = NT16 * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = wave & 3, half = wave >> 2;
  const int i = lane & 15, kq = lane >> 4;

  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (unsigned)(S * NT16 * 1024), 0x00020000);
  const unsigned wlane = (unsigned)lane * 16u;
  constexpr unsigned SB = NT16 * 1024u;          // bytes of packed weights per stage

  // per-thread staging constants: 512 % G == 0, so a thread serves ONE 4-channel slot g for all its patch pixels
  const int sg = threadIdx.x & (G - 1);
  const SrcPiece e0 = p.pieces[sg >> 1][sg & 1][0], e1 = p.pieces[sg >> 1][sg & 1][1];
  const f32x4 wsc = *reinterpret_cast<const f32x4 *>(p.sc + 4 * sg);
  const f32x4 wsh = *reinterpret_cast<const f32x4 *>(p.sh + 4 * sg);

  constexpr int NSUB = PAIRED ? 2 : 1;
  const int ntiles = p.B * p.tiles_x * p.tiles_y;     // PAIRED: tiles_y counts 8-row tile pairs
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    float pair_s1 = 0.f, pair_s2 = 0.f;               // PAIRED: the upper tile's partials (threads < COUT)
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
    int bid = tile;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty_slot = bid % p.tiles_y;
    const int ty = NSUB * ty_slot + sub;
    const int n = bid / p.tiles_y;
    const int ho0 = ty * TH, wo0 = tx * TW;
    const int hi_base = 2 * ho0 - 3, wi_base = 2 * wo0 - 3;

    // ---- stage the whitened input patch: all loads of a thread are issued before any is consumed
    if (!(p.dbg & 1)) {
      constexpr int PSTEP = NTHREADS / G;
      constexpr int NB = (PH * PW + PSTEP - 1) / PSTEP;   // 8 for G = 8, 4 for G = 4
      f32x2 x0[NB], x1[NB];
      bool ok[NB];
      const int pp0 = threadIdx.x / G;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = pp0 + b * PSTEP;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int hi = hi_base + pr, wi = wi_base + pc;
        ok[b] = pp < PH * PW && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const long pix = ((long)n * p.H + hi) * p.W + wi;
        const float *a0 = (ok[b] && e0.base != nullptr) ? e0.base + pix * e0.nch + e0.choff : p.zero_page;
        const float *a1 = (ok[b] && e1.base != nullptr) ? e1.base + pix * e1.nch + e1.choff : p.zero_page;
        x0[b] = *reinterpret_cast<const f32x2 *>(a0);
        x1[b] = *reinterpret_cast<const f32x2 *>(a1);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = pp0 + b * PSTEP;
        if (pp < PH * PW) {
          const int pc = pp % PW;
          f32x4 v = {x0[b][0], x0[b][1], x1[b][0], x1[b][1]};
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = ok[b] ? __builtin_fmaf(v[t], wsc[t], wsh[t]) : 0.f;   // pad channels: sc = sh = 0
          *reinterpret_cast<f32x4 *>(lds + pp * CPL + 4 * (sg ^ ((pc >> 1) & (G - 1)))) = v;
        }
      }
    }
    __syncthreads();

    // ---- K loop over this wave's stages s = 2u + half
    f32x4 acc[NT16];
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const f32x4 &a, const f32x4 (&b)[NT16]) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[nt][t], acc[nt], 0, 0, 0);
    };
    auto ld_b = [&](f32x4 (&b)[NT16], int u) {     // weights of stage s = 2u + half
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) b[nt] = wload4(rw, wlane + (unsigned)nt * 1024u, (unsigned)(2 * u + half) * SB);
    };
    // patch pixel of (output row, column i) at tap (0,0), in floats; the tap adds (kh*PW + kw)*CPL
    const float *abase = lds + ((2 * row) * PW + 2 * i) * CPL;
    auto ld_a = [&](int u) -> f32x4 {
      if constexpr (J16 == 2) {
        // stage s = 2u + half  <=>  tap u, 16-channel group `half`: everything but the swizzle is an immediate
        const int kh = u / 7, kw = u % 7;
        const int g = 4 * half + kq;
        return *reinterpret_cast<const f32x4 *>(abase + (kh * PW + kw) * CPL + 4 * (g ^ ((i + (kw >> 1)) & (G - 1))));
      } else {
        const int tap = 2 * u + half;             // wave-uniform
        const int kh = tap / 7, kw = tap - 7 * kh;
        return *reinterpret_cast<const f32x4 *>(abase + (kh * PW + kw) * CPL + 4 * (kq ^ ((i + (kw >> 1)) & (G - 1))));
      }
    };
    const int NU = (J16 == 2) ? 49 : (S - half + 1) / 2;   // stages of this wave (49 | 25/24)
    {
      // weights: 4-deep register ring (L2 latency >> one 256-cycle stage); A: one stage ahead (LDS latency)
      f32x4 b0[NT16], b1[NT16], b2[NT16], b3[NT16];
      ld_b(b0, 0);
      ld_b(b1, 1);
      ld_b(b2, 2);
      f32x4 a_cur = ld_a(0), a_nxt;
      constexpr int NUMAX = (S + 1) / 2;
#pragma unroll
      for (int u = 0; u < NUMAX; u += 4) {        // fully unrolled; guards on u are compile-time for J16 == 2
        if (u + 3 < NU) ld_b(b3, u + 3);
        if (u + 1 < NU) a_nxt = ld_a(u + 1);
        if (u < NU) mma(a_cur, b0);
        if (u + 4 < NU) ld_b(b0, u + 4);
        if (u + 2 < NU) a_cur = ld_a(u + 2);
        if (u + 1 < NU) mma(a_nxt, b1);
        if (u + 5 < NU) ld_b(b1, u + 5);
        if (u + 3 < NU) a_nxt = ld_a(u + 3);
        if (u + 2 < NU) mma(a_cur, b2);
        if (u + 6 < NU) ld_b(b2, u + 6);
        if (u + 4 < NU) a_cur = ld_a(u + 4);
        if (u + 3 < NU) mma(a_nxt, b3);
      }
    }

    // ---- sum the two K halves in a fixed order (half 1 -> LDS -> half 0); the patch is dead after the barrier
    __syncthreads();
    float *xch = lds;                              // [4 rows][NT16][64 lanes] float4
    if (half == 1) {
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) *reinterpret_cast<f32x4 *>(xch + ((row * NT16 + nt) * 64 + lane) * 4) = acc[nt];
    }
    __syncthreads();
    float *red = lds + 4 * NT16 * 64 * 4;          // [4 rows][COUT][2]
    if (half == 0 && !(p.dbg & 2)) {
      // C/D of 16x16x4: col (channel within the n-tile) = lane&15, row (pixel) = 4*(lane>>4) + reg
      const int ho = ho0 + row;
      const bool rvalid = ho < p.Ho;
      float *yrow = p.y + (((long)n * p.Ho + ho) * p.Wo) * COUT;
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) {
        const f32x4 o = *reinterpret_cast<const f32x4 *>(xch + ((row * NT16 + nt) * 64 + lane) * 4);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wo = wo0 + 4 * kq + r;
          const bool ok = rvalid && wo < p.Wo;
          const float v = ok ? acc[nt][r] + o[r] : 0.f;
          if (ok) yrow[(long)wo * COUT + nt * 16 + i] = v;
          s1 += v;
          s2 = __builtin_fmaf(v, v, s2);
        }
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kq == 0) {
          red[(row * COUT + nt * 16 + i) * 2] = s1;
          red[(row * COUT + nt * 16 + i) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < COUT && !(p.dbg & 2)) {   // per-tile GroupNorm partials, rows combined in a fixed order
      const int c = threadIdx.x;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s1 += red[(w * COUT + c) * 2];
        s2 += red[(w * COUT + c) * 2 + 1];
      }
      if (PAIRED && sub == 0) {
        pair_s1 = s1;
        pair_s2 = s2;
      } else {
        const int slot = ty_slot * p.tiles_x + tx;
        float *dst = p.stats + (((long)n * p.slots + slot) * COUT + c) * 2;
        dst[0] = PAIRED ? pair_s1 + s1 : s1;
        dst[1] = PAIRED ? pair_s2 + s2 : s2;
      }
    }
    __syncthreads();   // xch/red alias the patch: the next tile's staging must not start before they are consumed
    }  // sub-tile
  }  // tile loop
}

int stem_tiles_x(int Wo) { return (Wo + TW - 1) / TW; }
int stem_tiles_y(int Ho) { return (Ho + TH - 1) / TH; }
size_t stem_lds_bytes(int CPL) { return (size_t)PH * PW * CPL * 4; }

hipError_t launch_stem_lds(const StemArgs &a, int cout, hipStream_t s) {
  StemArgs p = a;
  p.tiles_x = stem_tiles_x(a.Wo);
  p.tiles_y = a.paired ? (a.Ho + 2 * TH - 1) / (2 * TH) : stem_tiles_y(a.Ho);
  const size_t lds = stem_lds_bytes(a.CPL);
  if ((a.CPL != 16 && a.CPL != 32) || lds < (size_t)(4 * (cout / 16) * 64 * 4 + 4 * cout * 2) * 4) return hipErrorInvalidValue;
  const long ntiles = (long)a.B * p.tiles_x * p.tiles_y;
  const long resident = (long)(160 * 1024 / (lds + (size_t)a.lds_pad)) * 256;   // workgroups the chip holds at once
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  const size_t dyn = lds + (size_t)a.lds_pad;
  if (a.paired) {                              // stand-in for the 8 x 16-tile stems (their slot layout), optionally predicated
    if (p.slots != p.tiles_x * p.tiles_y) return hipErrorInvalidValue;
    if (cout == 32 && a.CPL == 32)
      hipLaunchKernelGGL((stem_lds_kernel<2, 32, true>), grid, dim3(NTHREADS), dyn, s, p);
    else if (cout == 32 && a.CPL == 16)
      hipLaunchKernelGGL((stem_lds_kernel<2, 16, true>), grid, dim3(NTHREADS), dyn, s, p);
    else if (cout == 64 && a.CPL == 32)
      hipLaunchKernelGGL((stem_lds_kernel<4, 32, true>), grid, dim3(NTHREADS), dyn, s, p);
    else if (cout == 64 && a.CPL == 16)
      hipLaunchKernelGGL((stem_lds_kernel<4, 16, true>), grid, dim3(NTHREADS), dyn, s, p);
    else
      return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (cout == 32 && a.CPL == 32)
    hipLaunchKernelGGL((stem_lds_kernel<2, 32>), grid, dim3(NTHREADS), dyn, s, p);
  else if (cout == 32 && a.CPL == 16)
    hipLaunchKernelGGL((stem_lds_kernel<2, 16>), grid, dim3(NTHREADS), dyn, s, p);
  else if (cout == 64 && a.CPL == 32)
    hipLaunchKernelGGL((stem_lds_kernel<4, 32>), grid, dim3(NTHREADS), dyn, s, p);
  else if (cout == 64 && a.CPL == 16)
    hipLaunchKernelGGL((stem_lds_kernel<4, 16>), grid, dim3(NTHREADS), dyn, s, p);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// packed weights for the 16x16x4 stem: float4 index ((s*NT16 + nt)*64 + lane), s = tap*J16 + j16, lane = kq*16 + n,
// component t  <->  W[cout = nt*16 + n][cin = 16*j16 + 4*kq + t][tap]   (cin in the stem's own channel order)
void pack_stem_weight(const float *oihw_new, int cout, int cinp, int cpl, float *out) {
  const int NT16 = cout / 16, J16 = cpl / 16, T = 49;
  for (int tap = 0; tap < T; ++tap)
    for (int j = 0; j < J16; ++j)
      for (int nt = 0; nt < NT16; ++nt)
        for (int kq = 0; kq < 4; ++kq)
          for (int n = 0; n < 16; ++n)
            for (int t = 0; t < 4; ++t) {
              const int co = nt * 16 + n, ci = 16 * j + 4 * kq + t;
              const float v = ci < cinp ? oihw_new[((size_t)co * cinp + ci) * T + tap] : 0.f;
              out[((((size_t)(tap * J16 + j) * NT16 + nt) * 64) + kq * 16 + n) * 4 + t] = v;
            }
}

}  // namespace pnvo
