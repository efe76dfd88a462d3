// conv_mfma.hip — implicit-GEMM convolution / linear layer on the CDNA4 fp32 matrix cores (gfx950 only).
//
// Computes every conv of the reference backbone (resnet.py:11-26,156-163; vo_cnn.py:85-92) and both Linear
// layers (vo_cnn.py:219,225) as   D[M = B*Ho*Wo pixels][N = Cout] = A[M][K = KH*KW*Cin] * W[K][N]
// with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
//
// Design (MI355X-first, not a tiling of a warp-shaped kernel):
//  * one wave owns MT x 32 output pixels x NT x 32 output channels; its A rows are PRIVATE to it, so the A operand
//    never goes through LDS: lane (i = lane&31, h = lane>>5) loads 16 B = 4 consecutive input channels
//    [8j+4h, 8j+4h+4) of its own pixel straight from HBM/L2 (NHWC => the 4 j-steps of a tap consume exactly the
//    128-B line of a 32-channel pixel).  The MFMA k index is the lane half h, so K is walked in the order
//    (tap, j, t, h) and the weights are pre-packed in that order (pack_conv_weight) => one 1-KiB fully coalesced
//    float4 load per wave per (tap, j, n-tile);
//  * GroupNorm(+ReLU) of the PRODUCER layer is applied on the fly to A (x*scale[n,c]+shift[n,c], max 0) with the
//    per-(sample,channel) tables staged in LDS — GroupNorm statistics are per sample at inference, so they cannot
//    be folded into weights;
//  * the epilogue writes the raw conv output and deterministic per-wave partial (sum, sumsq) per (sample, channel)
//    for THIS layer's GroupNorm — no atomics, fixed summation order => bit-reproducible across runs and GPUs;
//  * no barriers in the K loop: waves are independent, 2-3 waves/SIMD hide the load latency under the 64-cycle MFMAs.
#include "pnvo_internal.h"
#include <cmath>

namespace pnvo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PNVO_OOB 0x80000000u   // byte offset >= num_records of every descriptor => buffer load returns 0, store dropped

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ unsigned clamp_records(long bytes) {
  return (unsigned)(bytes > 0x7FFFF000L ? 0x7FFFF000L : (bytes < 0 ? 0 : bytes));
}

// MODE 0: A = x as stored (already final activations)
// MODE 1: A = relu(x * scale[n,c] + shift[n,c])   — GroupNorm+ReLU of the producer layer, per-sample tables
// MODE 2: A gathered from the observation tensors (rgb | depth | discretized_depth | top_down_view) in 2-channel
//         pieces and whitened on the fly, x * scale[c] + shift[c] — the reference's input assembly + /255 +
//         RunningMeanAndVar (vo_cnn.py:110-176) fused into the stem's operand fetch; no [B,H,W,30] tensor exists.
// JC = 8-channel groups fetched per pipeline stage.  JC = 4 makes a stage one (tap, 32-channel chunk): the four 16-B
// loads of a lane hit ONE 128-B line back to back, so the line comes from L2 once per tap instead of once per 16-B
// slice (measured 4x L2 over-fetch with JC = 1: profiles/r1_kernel_mfma_busy.md).
// FEAT (compile-time so that the common kernels carry none of it): 0 plain; 1 the 4 waves of a workgroup share one tile and
// split its reduction (ConvArgs::wsplit); 2 strided output map (ConvArgs::y_sh, parity phases of a stride-2 backward-data conv).
template <int MT, int NT, int MODE, int JC, int FEAT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs p) {
  constexpr bool XF = (MODE != 0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WM = MT * 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long P = (long)p.Ho * p.Wo;
  const long M = (long)p.B * P;
  constexpr bool WS = FEAT == 1;                          // the 4 waves split the reduction of one tile (see ConvArgs)
  const long wg_m0 = (long)blockIdx.x * (WS ? WM : 4 * WM);
  const long m_base = WS ? wg_m0 : wg_m0 + (long)wave * WM;
  const int CIN = p.CIN;
  const int J = CIN >> 3;
  const long HWC = (long)p.H * p.W * CIN;

  // ---- stage the input-transform tables of the samples this workgroup touches into LDS
  int n_lo = 0, tab = 0;
  bool use_lds = false;
  if (XF) {
    long last = wg_m0 + (WS ? WM : 4 * WM) - 1;
    if (last > M - 1) last = M - 1;
    n_lo = (MODE == 2) ? 0 : (int)(wg_m0 / P);
    const int cnt = (MODE == 2) ? 1 : (int)(last / P) - n_lo + 1;
    tab = cnt * CIN;
    use_lds = (2 * tab <= p.lds_floats);
    if (use_lds) {
      const float *gs = p.in_scale + (long)n_lo * CIN, *gt = p.in_shift + (long)n_lo * CIN;
      for (int k = threadIdx.x; k < tab; k += 256) {
        lds[k] = gs[k];
        lds[tab + k] = gt[k];
      }
    }
    __syncthreads();
  }
  if (m_base >= M) return;

  // ---- descriptors: A rows relative to the first sample of this wave, packed weights of this n-tile group
  const int n0 = (int)(m_base / P);                      // wave-uniform
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(p.x + (long)n0 * HWC), 0, clamp_records(((long)p.B - n0) * HWC * 4), 0x00020000);
  const int T = p.KH * p.KW;
  const int SJ = T * J;                                  // (tap, 8-channel group) steps; JC of them per pipeline stage
  const int ntg0 = blockIdx.y * NT;
  const long w_nt_bytes = (long)SJ * 1024;               // one n-tile of packed weights
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(p.wpk + (long)ntg0 * SJ * 256), 0, clamp_records((long)NT * w_nt_bytes), 0x00020000);

  // ---- per-lane pixel coordinates of the MT pixel tiles
  int hi0[MT], wi0[MT], nrel[MT];
  unsigned pbase[MT];
  bool vm[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    long m = m_base + mt * 32 + i;
    vm[mt] = m < M;
    if (!vm[mt]) m = M - 1;
    const int n = (int)(m / P);
    const int rem = (int)(m - (long)n * P);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    nrel[mt] = (MODE == 2) ? n : n - n0;     // MODE 2 keeps the absolute sample index (pixel index into the sources)
    hi0[mt] = ho * p.stride - p.pad;
    wi0[mt] = wo * p.stride - p.pad;
    pbase[mt] = (unsigned)(((long)(n - n0) * HWC + 4 * h) * 4);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- loader state: the (kh, kw, j) of the next stage to fetch and the byte offsets of its tap
  const int S_all = SJ / JC;             // pipeline stages of the whole reduction
  int s_begin = 0, s_end = S_all;
  if (p.ksplit > 1) {                    // split-K (linear layers with few output tiles): this z-slice's stages
    const int per = (S_all + p.ksplit - 1) / p.ksplit;
    s_begin = (int)blockIdx.z * per;
    s_end = s_begin + per < S_all ? s_begin + per : S_all;
  }
  if (WS) {                              // this wave's quarter of the stages
    const int per = (s_end - s_begin + 3) / 4;
    s_begin += wave * per;
    if (s_begin > s_end) s_begin = s_end;
    if (s_begin + per < s_end) s_end = s_begin + per;
  }
  int l_s = s_begin * JC, l_j = l_s % J;
  int l_kh = (l_s / J) / p.KW, l_kw = (l_s / J) % p.KW;
  unsigned toff[MT];
  auto set_tap = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int hi = hi0[mt] + l_kh, wi = wi0[mt] + l_kw;
      bool ok = vm[mt];
      if (p.up == 2) {          // backward-data of a stride-2 conv: the input grid is the output gradient upsampled by 2
        ok = ok && (((hi | wi) & 1) == 0);
        hi >>= 1;
        wi >>= 1;
      }
      ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      if (MODE == 2)
        toff[mt] = ok ? (unsigned)((nrel[mt] * p.H + hi) * p.W + wi) : PNVO_OOB;
      else
        toff[mt] = ok ? pbase[mt] + (unsigned)((hi * p.W + wi) * CIN) * 4u : PNVO_OOB;
    }
  };
  set_tap();
  const unsigned wlane = (unsigned)lane * 16u;

  // fetch stage l_s into (a, b, okm, jc) and advance the loader state
  auto fetch = [&](f32x4 (&a)[MT][JC], f32x4 (&b)[NT][JC], unsigned &okm, int &jc) {
    okm = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        if (MODE == 2) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const SrcPiece e0 = p.pieces[l_j + jj][0][q], e1 = p.pieces[l_j + jj][1][q];   // wave-uniform (scalar loads)
            const float *base = h ? e1.base : e0.base;
            const int nch = h ? e1.nch : e0.nch, co = h ? e1.choff : e0.choff;
            const float *addr =
                (toff[mt] != PNVO_OOB && base != nullptr) ? base + ((long)toff[mt] * nch + co) : p.zero_page;
            const f32x2 v = *reinterpret_cast<const f32x2 *>(addr);
            a[mt][jj][2 * q] = v[0];
            a[mt][jj][2 * q + 1] = v[1];
          }
        } else {
          a[mt][jj] = buf_load4(rx, toff[mt], (unsigned)(l_j + jj) * 32u);
        }
      }
      okm |= (toff[mt] != PNVO_OOB ? 1u : 0u) << mt;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int jj = 0; jj < JC; ++jj)
        b[nt][jj] = buf_load4(rw, wlane + (unsigned)(nt * w_nt_bytes), (unsigned)(l_s + jj) * 1024u);
    jc = l_j;
    l_s += JC;
    l_j += JC;
    if (l_j == J) {
      l_j = 0;
      if (++l_kw == p.KW) {
        l_kw = 0;
        ++l_kh;
      }
      set_tap();
    }
  };

  // consume one fetched stage: optional producer GroupNorm+ReLU on A, then 4 k-steps of MFMA
  auto compute = [&](f32x4 (&a)[MT][JC], f32x4 (&b)[NT][JC], unsigned okm, int jc0) {
#pragma unroll
    for (int jj = 0; jj < JC; ++jj) {
      if (XF) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 sc, sh;
          const int c = 8 * (jc0 + jj) + 4 * h;
          if (use_lds) {
            const int k = (MODE == 2) ? c : (n0 + nrel[mt] - n_lo) * CIN + c;
            sc = *reinterpret_cast<const f32x4 *>(lds + k);
            sh = *reinterpret_cast<const f32x4 *>(lds + tab + k);
          } else {
            const long row = (MODE == 2) ? 0 : (long)(n0 + nrel[mt]);
            sc = *reinterpret_cast<const f32x4 *>(p.in_scale + row * CIN + c);
            sh = *reinterpret_cast<const f32x4 *>(p.in_shift + row * CIN + c);
          }
          const bool ok = (okm >> mt) & 1u;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float v = __builtin_fmaf(a[mt][jj][t], sc[t], sh[t]);
            if (MODE == 1) v = fmaxf(v, 0.f);
            a[mt][jj][t] = ok ? v : 0.f;   // zero padding is applied AFTER the producer's GN+ReLU / the whitening
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][jj][t], b[nt][jj][t], acc[mt][nt], 0, 0, 0);
    }
  };

  // ---- main loop, register double-buffered: stage s+1 is in flight while stage s feeds the matrix cores
  {
    f32x4 a0[MT][JC], b0[NT][JC], a1[MT][JC], b1[NT][JC];
    unsigned ok0 = 0, ok1 = 0;
    int j0 = 0, j1 = 0;
    const int S = s_end - s_begin;       // pipeline stages (>= 1; 0 only for a trailing wave of a split tile)
    if (!WS || S > 0) {
      fetch(a0, b0, ok0, j0);
      int s = 0;
      for (; s + 2 <= S - 1; s += 2) {   // invariant: stage s is in buffer 0, stages s+1, s+2 exist
        fetch(a1, b1, ok1, j1);
        compute(a0, b0, ok0, j0);
        fetch(a0, b0, ok0, j0);
        compute(a1, b1, ok1, j1);
      }
      if (s + 1 <= S - 1) {              // two stages left: s (buffer 0) and s+1
        fetch(a1, b1, ok1, j1);
        compute(a0, b0, ok0, j0);
        compute(a1, b1, ok1, j1);
      } else {                           // one stage left
        compute(a0, b0, ok0, j0);
      }
    }
  }

  // ---- split tile: waves 1..3 hand their partial accumulators to wave 0 through LDS; fixed summation order
  if (WS) {
    float *red = lds + p.lds_floats;     // [3][MT*NT*16][64], behind the input-transform tables
    if (wave > 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[(((wave - 1) * MT * NT + mt * NT + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] += red[((w * MT * NT + mt * NT + nt) * 16 + r) * 64 + lane];
  }

  // ---- epilogue 1: store.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  {
    long rows = M - m_base;
    if (rows > WM) rows = WM;
    float *ybase = p.ksplit > 1 ? p.kpart + (long)blockIdx.z * M * p.y_cstride : p.y;
    constexpr bool strided = FEAT == 2;                    // parity phase of a stride-2 backward-data conv
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(strided ? ybase : ybase + m_base * p.y_cstride), 0,
        clamp_records(strided ? (long)p.B * p.y_H * p.y_W * p.y_cstride * 4 : rows * p.y_cstride * 4), 0x00020000);
    const unsigned rstride = (unsigned)p.y_cstride * 4u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (ntg0 + nt) * 32 + i;
      const bool cvalid = co < p.y_cstride;   // y_cstride is COUT (exact) or COUTP (padded: zeros are stored)
      float bv[MT][4];
      if (p.bias != nullptr) {                // linear layers: bias may depend on the sample (act-embed variants)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) bv[mt][rr] = 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned vbase = cvalid ? (unsigned)(mt * 32 + 4 * h) * rstride + (unsigned)co * 4u : PNVO_OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          float v = acc[mt][nt][r];
          if (p.bias != nullptr) {
            const long m = m_base + mt * 32 + row + 4 * h;
            const long brow = (p.bias_row != nullptr && m < M) ? p.bias_row[m / P] : 0;
            v += (co < p.COUT) ? p.bias[brow * p.COUT + co] : 0.f;
          }
          if (p.relu_out) v = fmaxf(v, 0.f);
          unsigned yo = cvalid ? vbase + (unsigned)row * rstride : PNVO_OOB;
          if (strided) {                     // rare path (backward-data phases): the address math stays out of the common one
            const long m = m_base + mt * 32 + row + 4 * h;
            const int n = (int)(m / P);
            const int rem = (int)(m - (long)n * P);
            const int oi = rem / p.Wo, oj = rem - oi * p.Wo;
            const unsigned so = (unsigned)((((long)n * p.y_H + oi * p.y_sh + p.y_oh) * p.y_W + oj * p.y_sw + p.y_ow) * p.y_cstride) * 4u;
            yo = (cvalid && m < M) ? so + (unsigned)co * 4u : PNVO_OOB;
          }
          if (p.accum) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, yo, 0, 0));   // y += conv
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, yo, 0, 0);
        }
      }
      (void)bv;
    }
  }

  // ---- epilogue 2: per-(sample, channel) partial sums for this layer's GroupNorm
  if (p.stats != nullptr) {
    long last = m_base + WM - 1;
    if (last > M - 1) last = M - 1;
    const int n_first = n0, n_last = (int)(last / P);
    const long wt_idx = m_base / WM;
    for (int n = n_first; n <= n_last; ++n) {
      const long lo = (long)n * P, hi = lo + P;
      const int slot = (int)(wt_idx - lo / WM);
      const int rlo = (int)(lo - m_base), rhi = (int)((hi < M ? hi : M) - m_base);   // valid local rows [rlo, rhi)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = (row >= rlo && row < rhi) ? acc[mt][nt][r] : 0.f;
            s += v;
            q = __builtin_fmaf(v, v, q);
          }
        }
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (h == 0) {
          const int co = (ntg0 + nt) * 32 + i;
          float *dst = p.stats + (((long)n * p.slots + slot) * p.COUTP + co) * 2;
          dst[0] = s;
          dst[1] = q;
        }
      }
    }
  }
}

int conv_slots(int P, int MT) {
  const int WM = MT * 32;
  return (P + WM - 1) / WM + 1;
}

void choose_tile(long M, int COUTP, int *MT, int *NT) {
  // One 32-pixel tile per wave, 1 / 2 / 4 output-channel tiles.  Measured on every generic conv of the default model at 256
  // pairs (PNVO_CONV_TILE sweep): what decides is how evenly the workgroups divide over the 256 CUs —
  // 528 workgroups (2.06 per CU) lose to 1056 of half the size by 8 % despite the lower operand reuse — so: score =
  // (workgroups per CU) / ceil(workgroups per CU) x a mild preference for more n-tiles per wave; larger pixel tiles
  // (MT = 2, 4) lost everywhere.
  const int ntg = COUTP / 32;
  const long rows = (M + 127) / 128;                       // workgroups along M (4 waves x 32 pixels)
  static const double reuse[5] = {0.0, 1.0, 1.08, 0.0, 1.12};
  int best_nt = 1;
  double best = -1.0;
  for (int nt = 1; nt <= 4; nt *= 2) {
    if (ntg % nt) continue;
    const double per_cu = (double)(rows * (ntg / nt)) / 256.0;
    const double balance = per_cu >= 1.0 ? per_cu / std::ceil(per_cu) : per_cu;
    const double score = balance * reuse[nt];
    if (score > best) {
      best = score;
      best_nt = nt;
    }
  }
  int mt = 1, nt = best_nt;
  static const int force = std::getenv("PNVO_CONV_TILE") ? std::atoi(std::getenv("PNVO_CONV_TILE")) : 0;   // experiment knob: 10*MT+NT
  if (force > 0 && ntg % (force % 10) == 0 && M >= 8192) {
    mt = force / 10;
    nt = force % 10;
  }
  *MT = mt;
  *NT = nt;
}

// Split the reduction of every tile over the 4 waves of its workgroup?  Pays when whole-tile items divide badly over the
// SIMDs (stage 4 at 256 pairs: 4.125 wave items per SIMD run as 5; split, 16.5 quarter items run as 17) or do not fill the
// chip at all (batch 1), and the reduction is long enough to quarter.
int conv_wsplit(const ConvArgs &a) {
  static const int force = std::getenv("PNVO_CONV_WSPLIT") ? std::atoi(std::getenv("PNVO_CONV_WSPLIT")) : -1;   // experiment knob
  if (a.src_mode || a.ksplit > 1 || a.MT != 1 || a.NT > 2 || a.y_sh != 0) return 0;
  const int jc = (a.MT != 4 && a.CIN % 32 == 0) ? ((a.in_scale != nullptr && a.MT * a.NT >= 4) ? 2 : 4) : 1;
  const int S = a.KH * a.KW * (a.CIN / 8) / jc;
  if (S < 16) return 0;
  if (force >= 0) return force;
  const long P = (long)a.Ho * a.Wo, M = (long)a.B * P;
  const long WM = a.MT * 32, groups = a.COUTP / 32 / a.NT;
  const double whole = (double)(((M + 4 * WM - 1) / (4 * WM)) * groups) / 256.0;      // 4-wave workgroups per CU = waves per SIMD
  const double split = (double)(((M + WM - 1) / WM) * groups) / 256.0;                // quarter-size waves per SIMD
  const double t_whole = std::ceil(whole), t_split = std::ceil(split) / 4.0;
  return t_split < 0.88 * t_whole ? 1 : 0;      // the LDS hand-over and three idle epilogues cost ~8 % (measured: 33/4 vs 9 lost)
}

template <int MT, int NT>
static hipError_t launch_t(const ConvArgs &a, hipStream_t s) {
  const long P = (long)a.Ho * a.Wo, M = (long)a.B * P;
  const int WM = MT * 32;
  ConvArgs p = a;
  p.wsplit = conv_wsplit(a);
  const long wg_rows = p.wsplit ? WM : 4L * WM;
  const size_t red_bytes = p.wsplit ? (size_t)3 * MT * NT * 16 * 64 * 4 : 0;
  dim3 grid((unsigned)((M + wg_rows - 1) / wg_rows), (unsigned)(a.COUTP / 32 / NT), (unsigned)(a.ksplit > 1 ? a.ksplit : 1));
  if (a.ksplit > 1) {            // raw partials: bias and ReLU belong to the reduce
    p.bias = nullptr;
    p.bias_row = nullptr;
    p.relu_out = 0;
  }
  size_t lds_bytes = 0;
  // 32-channel stages where the register file allows 2 waves/SIMD; the GN-prologue variant of the 4-accumulator
  // tiles (2x2, 1x4) only fits 16-channel stages
  constexpr int JCF = (MT == 4) ? 1 : 4;
  constexpr int JCX = (MT == 4) ? 1 : ((MT * NT >= 4) ? 2 : 4);
  const bool wide = (JCF > 1) && (a.CIN % 32 == 0);
  const int feat = p.wsplit ? 1 : (a.y_sh != 0 ? 2 : 0);
  if (feat != 0 && (MT != 1 || NT > 2 + 2 * (feat == 2) || a.src_mode || (feat == 2 && a.in_scale != nullptr))) return hipErrorInvalidValue;
#define PNVO_LAUNCH(MODE_, JC_, LDS_)                                                                                   \
  do {                                                                                                                  \
    if (feat == 0)                                                                                                      \
      hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, MODE_, JC_, 0>), grid, dim3(256), (LDS_), s, p);                     \
    else if constexpr (MT == 1 && NT <= 2 && MODE_ != 2) {                                                              \
      if (feat == 1)                                                                                                    \
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, MODE_, JC_, 1>), grid, dim3(256), (LDS_) + red_bytes, s, p);       \
      else if constexpr (MODE_ == 0)                                                                                    \
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, MODE_, JC_, 2>), grid, dim3(256), (LDS_), s, p);                   \
    } else if constexpr (MT == 1 && MODE_ == 0) {                                                                       \
      if (feat == 2) hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, MODE_, JC_, 2>), grid, dim3(256), (LDS_), s, p);     \
    }                                                                                                                   \
  } while (0)
  if (a.src_mode) {
    p.lds_floats = 2 * a.CIN;
    PNVO_LAUNCH(2, 1, (size_t)p.lds_floats * 4);
  } else if (a.in_scale != nullptr) {
    const long cnt_max = (wg_rows + P - 2) / P + 1;
    const long need = cnt_max * a.CIN * 2;
    if (need * 4 <= 48 * 1024) {
      p.lds_floats = (int)need;
      lds_bytes = (size_t)need * 4;
    } else {
      p.lds_floats = 0;
    }
    if (wide)
      PNVO_LAUNCH(1, JCX, lds_bytes);
    else
      PNVO_LAUNCH(1, 1, lds_bytes);
  } else {
    p.lds_floats = 0;
    if (wide)
      PNVO_LAUNCH(0, JCF, (size_t)0);
    else
      PNVO_LAUNCH(0, 1, (size_t)0);
  }
#undef PNVO_LAUNCH
  return hipGetLastError();
}

// Split-K pays when a linear layer has too few output tiles to fill the chip and a long reduction (the 6x11 "conv" of the
// FC: 32 workgroups x 66 stages at 256 pairs).  Only for plain linear epilogues (no statistics, no accumulate).
int conv_ksplit(const ConvArgs &a) {
  if (a.kpart == nullptr || a.stats != nullptr || a.accum || a.src_mode) return 1;
  const long P = (long)a.Ho * a.Wo, M = (long)a.B * P;
  const long wgs = ((M + 4L * a.MT * 32 - 1) / (4L * a.MT * 32)) * (a.COUTP / 32 / a.NT);
  int jc = 1;                          // channel groups per pipeline stage, as launch_t picks them
  if (a.MT != 4 && a.CIN % 32 == 0) jc = (a.in_scale != nullptr && a.MT * a.NT >= 4) ? 2 : 4;
  const int S = a.KH * a.KW * (a.CIN / 8) / jc;
  if (wgs >= 128 || S < 16) return 1;
  int k = (int)(512 / wgs);
  if (k > S / 4) k = S / 4;
  if (k > 32) k = 32;
  if (k < 2) return 1;
  const int per = (S + k - 1) / k;
  return (S + per - 1) / per;        // every slice non-empty
}

__global__ __launch_bounds__(256) void ksplit_reduce_kernel(const float *part, int ks, long M, int C, const float *bias,
                                                          const int64_t *bias_row, int COUT, long P, int relu, float *y) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * C) return;
  const long m = e / C;
  const int c = (int)(e - m * C);
  float v = 0.f;
  for (int z = 0; z < ks; ++z) v += part[(long)z * M * C + e];      // fixed order
  if (bias != nullptr && c < COUT) v += bias[(bias_row ? bias_row[m / P] : 0) * COUT + c];
  if (relu) v = fmaxf(v, 0.f);
  y[e] = v;
}

// The same reduction for a linear layer (P == 1: one row per sample) with the OUTPUT HEAD riding on it (vo_cnn.py:216-227: Linear
// hidden -> out_dim behind the hidden layer's ReLU): one workgroup per sample sums the K slices of its row (fixed order), adds the bias,
// applies the ReLU, writes the hidden vector — and multiplies it with the head's [OD][C] weight on the way: per-lane partial dot
// products, a butterfly over the wave, the four waves in order (a fixed tree: deterministic; float32-grade equal to the fp32-MFMA head
// launch it replaces, whose chain runs in K order).
__global__ __launch_bounds__(256) void ksplit_reduce_head_kernel(const float *part, int ks, long M, int C, const float *bias,
                                                               const int64_t *bias_row, int relu, float *y, const float *w2,
                                                               const float *b2, int OD, float *out) {
  __shared__ float red[4][4];
  const long m = blockIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = (int)threadIdx.x; c < C; c += 256) {
    const long e = m * C + c;
    float v = 0.f;
    for (int z = 0; z < ks; ++z) v += part[(long)z * M * C + e];      // fixed order (ksplit_reduce_kernel's)
    if (bias != nullptr) v += bias[(bias_row ? bias_row[m] : 0) * C + c];
    if (relu) v = fmaxf(v, 0.f);
    y[e] = v;
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < OD) acc[o] = __builtin_fmaf(v, w2[(long)o * C + c], acc[o]);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc[o] += __shfl_xor(acc[o], d);
  const int wave = (int)(threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int o = 0; o < 4; ++o) red[wave][o] = acc[o];
  __syncthreads();
  if ((int)threadIdx.x < OD) {
    const int o = (int)threadIdx.x;
    out[m * OD + o] = (((red[0][o] + red[1][o]) + red[2][o]) + red[3][o]) + b2[o];
  }
}

hipError_t launch_ksplit_reduce_head(const ConvArgs &a, const float *w2, const float *b2, int out_dim, float *out, hipStream_t s) {
  if (a.Ho * a.Wo != 1 || out_dim < 1 || out_dim > 4 || a.y_cstride != a.COUT) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ksplit_reduce_head_kernel, dim3((unsigned)a.B), dim3(256), 0, s, a.kpart, a.ksplit, (long)a.B, a.y_cstride, a.bias,
                     a.bias_row, a.relu_out, a.y, w2, b2, out_dim, out);
  return hipGetLastError();
}

hipError_t launch_ksplit_reduce(const ConvArgs &a, hipStream_t s) {
  const long P = (long)a.Ho * a.Wo, M = (long)a.B * P;
  hipLaunchKernelGGL(ksplit_reduce_kernel, dim3((unsigned)((M * a.y_cstride + 255) / 256)), dim3(256), 0, s, a.kpart, a.ksplit,
                     M, a.y_cstride, a.bias, a.bias_row, a.COUT, P, a.relu_out, a.y);
  return hipGetLastError();
}

hipError_t launch_conv(const ConvArgs &a, hipStream_t s) {
  if (a.CIN % 8 != 0 || a.COUTP % 32 != 0 || (a.COUTP / 32) % a.NT != 0) return hipErrorInvalidValue;
  switch (a.MT * 10 + a.NT) {
    case 41: return launch_t<4, 1>(a, s);
    case 21: return launch_t<2, 1>(a, s);
    case 11: return launch_t<1, 1>(a, s);
    case 22: return launch_t<2, 2>(a, s);
    case 12: return launch_t<1, 2>(a, s);
    case 14: return launch_t<1, 4>(a, s);
    default: return hipErrorInvalidValue;
  }
}

// Packed layout consumed by the kernel: float4 index ((ntg*T + tap)*J + j)*64 + lane, lane = h*32 + n,
// component t  <->  W[cout = ntg*32 + n][cin = 8j + 4h + t][kh][kw]   (zero outside the real Cout/Cin).
size_t packed_conv_floats(int cout, int cin, int kh, int kw) {
  const size_t coutp = (size_t)(cout + 31) / 32 * 32, cinp = (size_t)(cin + 7) / 8 * 8;
  return coutp * cinp * kh * kw;
}

void pack_conv_weight(const float *oihw, int cout, int cin, int kh, int kw, float *out) {
  const int ntg_n = (cout + 31) / 32, J = (cin + 7) / 8, T = kh * kw;
  for (int ntg = 0; ntg < ntg_n; ++ntg)
    for (int tap = 0; tap < T; ++tap)
      for (int j = 0; j < J; ++j)
        for (int h = 0; h < 2; ++h)
          for (int n = 0; n < 32; ++n)
            for (int t = 0; t < 4; ++t) {
              const int co = ntg * 32 + n, ci = 8 * j + 4 * h + t;
              float v = 0.f;
              if (co < cout && ci < cin) v = oihw[((size_t)co * cin + ci) * T + tap];
              out[((((size_t)ntg * T + tap) * J + j) * 64 + h * 32 + n) * 4 + t] = v;
            }
}

}  // namespace pnvo
