// stem_lds.hip — the 7x7 stride-2 stem (resnet.py:156-163) with input assembly + /255 + whitening (vo_cnn.py:110-176)
// fused in, on the CDNA4 fp32 matrix cores (gfx950 only).
//
// Why a dedicated kernel: the stem is 57.5 % of the path's FLOPs and every input pixel feeds 49/4 output pixels.
// Served from L2/HBM that reuse costs ~17x the algorithmic input in fabric traffic (measured with rocprofv3:
// FETCH_SIZE 34 GB per dispatch for 2 GB of input, L2 hit rate 37 % — profiles/r1_stem_gather_pmc.md), so the
// kernel ran at the fabric's speed, not the matrix cores'.  Here each workgroup stages its input patch ONCE:
//
//   * workgroup = 4 waves = a 4-row x 16-column tile of output pixels of one frame pair; its (2*4+5) x (2*16+5)
//     input patch is gathered from the four observation tensors (rgb | depth | discretized_depth | top_down_view)
//     in 2-channel pieces, whitened (x*sc[c]+sh[c], zero outside the image = the conv's zero padding applied after
//     whitening) and written to LDS as [13][37][CPL] fp32 with a 16-byte-slot XOR swizzle;
//   * 61.6 KB of LDS per workgroup -> 2 workgroups per CU: one stages while the other feeds the matrix cores;
//   * each wave owns one output row (16 pixels) x all Cout: v_mfma_f32_16x16x4_f32, A from LDS (ds_read_b128 per
//     lane = 4 consecutive channels of its pixel at the current tap), B = pre-packed weights streamed from L2 with a
//     4-stage register ring; no barrier inside the K loop;
//   * epilogue: raw conv output + deterministic per-tile partial (sum, sumsq) per channel for the GroupNorm that
//     follows (fixed order, no atomics).
#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 16;                 // output tile
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;  // input patch 13 x 37

__device__ __forceinline__ f32x4 wload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int NT16>   // Cout / 16
__global__ __launch_bounds__(256) void stem_lds_kernel(const StemArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int CPL = p.CPL;              // channels per pixel in LDS (multiple of 16)
  const int G = CPL >> 2;             // 16-byte slots per pixel (4 or 8)
  const int J16 = CPL >> 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;

  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y;
  const int n = bid / p.tiles_y;
  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hi_base = 2 * ho0 - 3, wi_base = 2 * wo0 - 3;

  // ---- stage the whitened input patch.  256 % G == 0, so a thread serves ONE 4-channel slot g for all its pixels:
  // its two source pieces and whitening constants are loaded once; pixels are processed in batches of NB whose
  // loads are all issued before any is consumed (memory-level parallelism instead of a dependent-load chain).
  {
    const int g = threadIdx.x & (G - 1);
    const int pstep = 256 / G;
    const SrcPiece e0 = p.pieces[g >> 1][g & 1][0], e1 = p.pieces[g >> 1][g & 1][1];
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.sc + 4 * g);
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(p.sh + 4 * g);
    constexpr int NB = 8;
    for (int pp0 = threadIdx.x / G; pp0 < PH * PW; pp0 += pstep * NB) {
      f32x2 x0[NB], x1[NB];
      bool ok[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int pp = pp0 + b * pstep;
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
        const int pp = pp0 + b * pstep;
        if (pp < PH * PW) {
          const int pc = pp % PW;
          f32x4 v = {x0[b][0], x0[b][1], x1[b][0], x1[b][1]};
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = ok[b] ? __builtin_fmaf(v[t], sc[t], sh[t]) : 0.f;   // pad channels: sc = sh = 0
          *reinterpret_cast<f32x4 *>(lds + (long)pp * CPL + 4 * (g ^ ((pc >> 1) & (G - 1)))) = v;
        }
      }
    }
  }
  __syncthreads();

  // ---- K loop: stage s = (tap, j16); A from LDS, B streamed from the packed weights
  const int S = 49 * J16;
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (unsigned)(S * NT16 * 1024), 0x00020000);
  const unsigned wlane = (unsigned)lane * 16u;
  f32x4 acc[NT16];
#pragma unroll
  for (int nt = 0; nt < NT16; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int rowbase = (2 * wave) * PW + 2 * i;   // patch pixel of (output row `wave`, column i) at tap (0,0)
  int a_kh = 0, a_kw = 0, a_j = 0;               // loader state for A (next stage to read)
  auto lds_a = [&]() -> f32x4 {
    const int pc = 2 * i + a_kw;
    const int pp = rowbase + a_kh * PW + a_kw;
    const int g = 4 * a_j + kq;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(lds + pp * CPL + 4 * (g ^ ((pc >> 1) & (G - 1))));
    if (++a_j == J16) {
      a_j = 0;
      if (++a_kw == 7) {
        a_kw = 0;
        ++a_kh;
      }
    }
    return v;
  };
  auto ld_b = [&](f32x4 (&b)[NT16], int s) {
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) b[nt] = wload4(rw, wlane + (unsigned)nt * 1024u, (unsigned)s * (NT16 * 1024u));
  };
  auto mma = [&](const f32x4 &a, const f32x4 (&b)[NT16]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT16; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[nt][t], acc[nt], 0, 0, 0);
  };

  {
    // weights: 4-deep register ring (L2 latency >> one 256-cycle stage); A: one stage ahead (LDS latency)
    f32x4 b0[NT16], b1[NT16], b2[NT16], b3[NT16];
    ld_b(b0, 0);
    ld_b(b1, 1);
    ld_b(b2, 2);
    f32x4 a_cur = lds_a(), a_nxt;
    int s = 0;
    for (; s + 4 <= S - 3; s += 4) {        // stages s..s+3 computed, s+3..s+6 fetched (all < S)
      ld_b(b3, s + 3);
      a_nxt = lds_a();
      mma(a_cur, b0);
      ld_b(b0, s + 4);
      a_cur = lds_a();
      mma(a_nxt, b1);
      ld_b(b1, s + 5);
      a_nxt = lds_a();
      mma(a_cur, b2);
      ld_b(b2, s + 6);
      a_cur = lds_a();
      mma(a_nxt, b3);
    }
    // tail: stages s .. S-1 (between 3 and 6 left); b0,b1,b2 hold s, s+1, s+2
    for (; s < S; s += 4) {
      if (s + 3 < S) ld_b(b3, s + 3);
      if (s + 1 < S) a_nxt = lds_a();
      mma(a_cur, b0);
      if (s + 4 < S) ld_b(b0, s + 4);
      if (s + 1 < S) {
        if (s + 2 < S) a_cur = lds_a();
        mma(a_nxt, b1);
      }
      if (s + 5 < S) ld_b(b1, s + 5);
      if (s + 2 < S) {
        if (s + 3 < S) a_nxt = lds_a();
        mma(a_cur, b2);
      }
      if (s + 6 < S) ld_b(b2, s + 6);
      if (s + 3 < S) {
        if (s + 4 < S) a_cur = lds_a();
        mma(a_nxt, b3);
      }
    }
  }

  // ---- epilogue.  C/D of 16x16x4: col (channel within the n-tile) = lane&15, row (pixel) = 4*(lane>>4) + reg
  const int ho = ho0 + wave;
  const int COUT = NT16 * 16;
  const bool rvalid = ho < p.Ho;
  float *yrow = p.y + (((long)n * p.Ho + ho) * p.Wo) * COUT;
  float ssum[NT16], sq[NT16];
#pragma unroll
  for (int nt = 0; nt < NT16; ++nt) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int wo = wo0 + 4 * kq + r;
      const bool ok = rvalid && wo < p.Wo;
      const float v = ok ? acc[nt][r] : 0.f;
      if (ok) yrow[(long)wo * COUT + nt * 16 + i] = v;
      s1 += v;
      s2 = __builtin_fmaf(v, v, s2);
    }
    s1 += __shfl_xor(s1, 16);
    s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    ssum[nt] = s1;
    sq[nt] = s2;
  }
  // per-tile GroupNorm partials: combine the 4 waves in a fixed order through LDS (patch is dead by now)
  __syncthreads();
  float *red = lds;   // [4 waves][COUT][2]
  if (kq == 0) {
#pragma unroll
    for (int nt = 0; nt < NT16; ++nt) {
      red[(wave * COUT + nt * 16 + i) * 2] = ssum[nt];
      red[(wave * COUT + nt * 16 + i) * 2 + 1] = sq[nt];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < COUT) {
    const int c = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s1 += red[(w * COUT + c) * 2];
      s2 += red[(w * COUT + c) * 2 + 1];
    }
    const int slot = ty * p.tiles_x + tx;
    float *dst = p.stats + (((long)n * p.slots + slot) * COUT + c) * 2;
    dst[0] = s1;
    dst[1] = s2;
  }
}

int stem_tiles_x(int Wo) { return (Wo + TW - 1) / TW; }
int stem_tiles_y(int Ho) { return (Ho + TH - 1) / TH; }
size_t stem_lds_bytes(int CPL) { return (size_t)PH * PW * CPL * 4; }

hipError_t launch_stem_lds(const StemArgs &a, int cout, hipStream_t s) {
  StemArgs p = a;
  p.tiles_x = stem_tiles_x(a.Wo);
  p.tiles_y = stem_tiles_y(a.Ho);
  const size_t lds = stem_lds_bytes(a.CPL);
  if (a.CPL % 16 || a.CPL > 32 || lds < (size_t)4 * cout * 2 * 4) return hipErrorInvalidValue;
  dim3 grid((unsigned)((long)a.B * p.tiles_x * p.tiles_y));
  if (cout == 32)
    hipLaunchKernelGGL((stem_lds_kernel<2>), grid, dim3(256), lds, s, p);
  else if (cout == 64)
    hipLaunchKernelGGL((stem_lds_kernel<4>), grid, dim3(256), lds, s, p);
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
