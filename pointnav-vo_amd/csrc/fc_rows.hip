// fc_rows.hip — the two Linear layers behind the compression conv (vo_cnn.py:216-227: Flatten -> Linear(flat, hidden) + ReLU -> Linear(hidden,
// out_dim)) for the batches of the navigation loop (<= 32 samples), gfx950.
//
// At 8-32 samples the hidden layer is a skinny GEMM (512 x 2046 weights = 4.2 MB read for a few rows): the split-K MFMA kernel plus its
// reduction took 26 us at 8 pairs, and a grouped forward (three action models) ran them once per model.  Here ONE launch serves every
// model of the group: a wave owns one hidden unit and up to four samples of one model (blockIdx.y = the chunk of samples); the GroupNorm + ReLU of the compression conv (scale / shift per sample and channel) is applied to the
// activations as they are read.  A second small launch is the output head.  float32 FMA chains, fixed lane tree: deterministic.
#include "pnvo_internal.h"

namespace pnvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fc_rows_kernel(const FcRowsArgs p) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (n >= p.hidden) return;
  const int K4 = p.Kp >> 2;
  const int c4 = (4 * lane) % p.cp;                     // the lane's channels (256 % cp == 0: the same for all of its k)
  // blockIdx.y = a chunk of up to four samples of ONE model (chunks are counted model by model)
  int g = 0, b = 0, c = (int)blockIdx.y;
  for (;; ++g) {
    if (g >= p.ngroups) return;
    const int nc = (p.end[g] - b + 3) >> 2;
    if (c < nc) break;
    c -= nc;
    b = p.end[g];
  }
  b += 4 * c;
  const int nb = min(4, p.end[g] - b);
  const f32x4 *wr = reinterpret_cast<const f32x4 *>(p.w[g] + (long)n * p.Kp);
  // four samples at a time: their loads are in flight together and the four lane reductions interleave (a sample alone is one
  // dependent chain of a load round trip, 36 FMAs and six shuffles: measured 2.3 us per sample and wave)
  const f32x4 *xr[4];
  f32x4 sc[4], sh[4];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int bu = b + min(u, nb - 1);                   // (past the chunk's end: the last sample again, result unused)
    xr[u] = reinterpret_cast<const f32x4 *>(p.x + (long)bu * p.Kp);
    sc[u] = *reinterpret_cast<const f32x4 *>(p.sc + (long)bu * p.cp + c4);
    sh[u] = *reinterpret_cast<const f32x4 *>(p.sh + (long)bu * p.cp + c4);
  }
#pragma unroll
  for (int i = 0; i < FC_ROWS_MAXV; ++i) {
    const int k = lane + 64 * i;
    if (k < K4) {
      const f32x4 w = wr[k];
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = xr[u][k];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[u] = __builtin_fmaf(w[e], fmaxf(__builtin_fmaf(v[u][e], sc[u][e], sh[u][e]), 0.f), s[u]);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += __shfl_xor(s[u], o);
  if (lane < nb) {
    const int bu = b + lane;
    const float sv = lane == 0 ? s[0] : (lane == 1 ? s[1] : (lane == 2 ? s[2] : s[3]));
    const long row = p.actions != nullptr ? (long)p.actions[bu] : 0;
    p.hid[(long)bu * p.hidden + n] = fmaxf(sv + p.bias[g][row * p.hidden + n], 0.f);
  }
}

// out[b][o] = bias[o] + hid[b] . W[o]   (one wave per (sample, output) pair in turn)
__global__ __launch_bounds__(256) void head_rows_kernel(const FcRowsArgs p) {
  const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6);
  const int K4 = p.hidden >> 2;
  for (int it = blockIdx.x * 4 + wave; it < p.B * p.out_dim; it += gridDim.x * 4) {
    const int b = it / p.out_dim, o = it - b * p.out_dim;
    int g = 0;
    while (g + 1 < p.ngroups && b >= p.end[g]) ++g;
    const f32x4 *wr = reinterpret_cast<const f32x4 *>(p.head_w[g] + (long)o * p.hidden);
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(p.hid + (long)b * p.hidden);
    float s = 0.f;
    for (int k = lane; k < K4; k += 64) {
      const f32x4 w = wr[k], v = xr[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) s = __builtin_fmaf(w[e], v[e], s);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) p.out[(long)b * p.out_dim + o] = s + p.head_b[g][o];
  }
}

hipError_t launch_fc_rows(const FcRowsArgs &a, bool with_head, hipStream_t s) {
  int chunks = 0, b0 = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    chunks += (a.end[g] - b0 + 3) / 4;
    b0 = a.end[g];
  }
  hipLaunchKernelGGL(fc_rows_kernel, dim3((unsigned)((a.hidden + 3) / 4), (unsigned)std::max(1, chunks)), dim3(256), 0, s, a);
  if (with_head) {
    const int items = a.B * a.out_dim;
    hipLaunchKernelGGL(head_rows_kernel, dim3((unsigned)std::max(1, std::min(64, (items + 3) / 4))), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace pnvo
