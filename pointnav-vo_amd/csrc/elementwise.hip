// elementwise.hip — HBM-bound kernels of the VO hot path (gfx950): input assembly + whitening, GroupNorm statistics
// finalisation, fused GN+ReLU+maxpool, residual tail, one-hot depth and the ego top-down view.
// All are pure streaming kernels: 16-byte accesses per lane, consecutive lanes on consecutive addresses.
#include "pnvo_internal.h"

// One rounding per source-level operation in this file: the top-down view reproduces the reference's float32 op
// sequence bit for bit (floor/ceil of the results pick histogram bins), so the compiler must not fuse a*b+c.
// Kernels that WANT an fma say so with __builtin_fmaf.
#pragma clang fp contract(off)

namespace pnvo {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// single-rounding float32 primitives (HIP's __fmul_rn/__fadd_rn are header inlines that still carry the `contract`
// fast-math flag and get fused into v_fma by the backend; these, defined under contract(off), do not)
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }   // correctly rounded (hipcc default)

// ------------------------------------------------------------------------------------------------------------------
// Input assembly + RunningMeanAndVar (vo_cnn.py:110-176, running_mean_and_var.py:62-63):
//   out[pix][c] = (src_c(pix) [/255 for rgb] - mean[c]) / stdev[c],   channel order
//   [prev_rgb, prev_d, prev_dd, prev_tdv, cur_rgb, cur_d, cur_dd, cur_tdv], zero-filled up to CP (multiple of 8).
struct AssembleDev {
  const float *src[4];
  int nsrc[4];
  long npix;
  float *out;
  int C, CP, normalize;
  signed char sid[64];   // source tensor of output channel c (-1: pad)
  unsigned char sch[64]; // channel inside that tensor
  float mean[64], stdev[64];
};

__global__ __launch_bounds__(256) void assemble_kernel(const AssembleDev a) {
  const int Q = a.CP >> 2;
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  if (g >= a.npix * Q) return;
  const int q = (int)(g % Q);
  const long pix = g / Q;
  f32x4 o;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = 4 * q + t;
    const int s = a.sid[c];
    float v = 0.f;
    if (s >= 0) {
      v = a.src[s][pix * a.nsrc[s] + a.sch[c]];
      if (s == 0) v = div_rn(v, 255.0f);
      if (a.normalize) v = div_rn(sub_rn(v, a.mean[c]), a.stdev[c]);
    }
    o[t] = v;
  }
  reinterpret_cast<f32x4 *>(a.out)[g] = o;
}

hipError_t launch_assemble(const AssembleArgs &h, hipStream_t s) {
  if (h.CP > 64 || h.CP % 4) return hipErrorInvalidValue;
  AssembleDev d;
  for (int k = 0; k < 4; ++k) {
    d.src[k] = h.src[k];
    d.nsrc[k] = h.nsrc[k];
  }
  d.npix = h.npix;
  d.out = h.out;
  d.C = h.C;
  d.CP = h.CP;
  d.normalize = h.mean != nullptr;
  int c = 0;
  for (int half = 0; half < 2; ++half)
    for (int k = 0; k < 4; ++k) {
      const int n = h.nsrc[k] / 2;
      for (int j = 0; j < n; ++j, ++c) {
        d.sid[c] = (signed char)k;
        d.sch[c] = (unsigned char)(half * n + j);
      }
    }
  for (; c < 64; ++c) {
    d.sid[c] = -1;
    d.sch[c] = 0;
  }
  for (int k = 0; k < 64; ++k) {
    d.mean[k] = (h.mean && k < h.C) ? h.mean[k] : 0.f;
    d.stdev[k] = (h.mean && k < h.C) ? h.stdev[k] : 1.f;
  }
  const long total = h.npix * (h.CP / 4);
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm finalisation: per-(sample, group) mean / biased variance from the conv epilogue's per-wave partial
// (sum, sumsq), reduced in a FIXED order in fp64, folded with the affine parameters into
//   scale[n][c] = rstd * gamma[c],  shift[n][c] = beta[c] - mean * scale[n][c]
// (torch.nn.GroupNorm, eps inside the sqrt; used at resnet.py:39,42,165,194 and vo_cnn.py:93).
__device__ __forceinline__ void gn_finalize_block(const float *stats, int slots, int CP, int C, int G, long P,
                                                  int WM, const float *gamma, const float *beta, float eps,
                                                  float *scale, float *shift, int fixed_ns, float *mu_out,
                                                  float *rstd_out, const GnGroup &grp) {
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int mdl = (n >= grp.end0) + (n >= grp.end1);           // grouped forward: the sample's action model
  if (mdl > 0) {
    gamma = grp.gamma[mdl - 1];
    beta = grp.beta[mdl - 1];
  }
  const int cpg = C / G;
  const long t0 = ((long)n * P) / WM, t1 = ((long)(n + 1) * P - 1) / WM;
  const int ns = fixed_ns > 0 ? fixed_ns : (int)(t1 - t0 + 1);
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < ns * cpg; k += 64) {
    const int slot = k / cpg, c = g * cpg + k % cpg;
    const float *src = stats + (((long)n * slots + slot) * CP + c) * 2;
    s1 += (double)src[0];
    s2 += (double)src[1];
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
  if (threadIdx.x == 0 && mu_out != nullptr) {
    mu_out[n * G + g] = (float)mu;
    rstd_out[n * G + g] = (float)rstd;
  }
  for (int k = threadIdx.x; k < cpg; k += 64) {
    const int c = g * cpg + k;
    const double sc = rstd * (double)gamma[c];
    scale[(long)n * CP + c] = (float)sc;
    shift[(long)n * CP + c] = (float)((double)beta[c] - mu * sc);
  }
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(const float *stats, int slots, int CP, int C, int G, long P,
                                                        int WM, const float *gamma, const float *beta, float eps,
                                                        float *scale, float *shift, int fixed_ns, float *mu_out,
                                                        float *rstd_out, const GnGroup grp) {
  gn_finalize_block(stats, slots, CP, C, G, P, WM, gamma, beta, eps, scale, shift, fixed_ns, mu_out, rstd_out, grp);
}

// Two GroupNorms of the same geometry in one launch (a stride-2 block's first conv and the downsample conv that rode on it):
// blockIdx.y selects the set; the arithmetic of each is gn_finalize_kernel's.
struct GnFinPair {
  const float *stats[2], *gamma[2], *beta[2];
  float *scale[2], *shift[2], *mu[2], *rstd[2];
  GnGroup grp[2];
};
__global__ __launch_bounds__(64) void gn_finalize_pair_kernel(const GnFinPair q, int slots, int CP, int C, int G, long P, float eps) {
  const int z = blockIdx.y;
  gn_finalize_block(q.stats[z], slots, CP, C, G, P, 1, q.gamma[z], q.beta[z], eps, q.scale[z], q.shift[z], slots, q.mu[z], q.rstd[z], q.grp[z]);
}

hipError_t launch_gn_finalize_pair(const float *const *stats, int B, int slots, int CP, int C, int G, long P, const float *const *gamma,
                                   const float *const *beta, float eps, float *const *scale, float *const *shift, float *const *mu_out,
                                   float *const *rstd_out, hipStream_t s, const GnGroup *grp0, const GnGroup *grp1) {
  GnFinPair q;
  const GnGroup none{0x7fffffff, 0x7fffffff, {nullptr, nullptr}, {nullptr, nullptr}};
  q.grp[0] = grp0 ? *grp0 : none;
  q.grp[1] = grp1 ? *grp1 : none;
  for (int z = 0; z < 2; ++z) {
    q.stats[z] = stats[z];
    q.gamma[z] = gamma[z];
    q.beta[z] = beta[z];
    q.scale[z] = scale[z];
    q.shift[z] = shift[z];
    q.mu[z] = mu_out[z];
    q.rstd[z] = rstd_out[z];
  }
  hipLaunchKernelGGL(gn_finalize_pair_kernel, dim3((unsigned)(B * G), 2u), dim3(64), 0, s, q, slots, CP, C, G, P, eps);
  return hipGetLastError();
}

// Two models in one launch (the bf16 dual forward): blockIdx.y = model.
struct GnFin2 {
  const float *stats[2], *gamma[2], *beta[2];
  float *scale[2], *shift[2];
};
// One 256-thread block per (sample, model): 256 / G threads per group read the per-slot partials (coalesced over the
// channel pairs of a slot), fixed-order fp64 reduction through LDS.
__global__ __launch_bounds__(256) void gn_finalize2_kernel(const GnFin2 q, int slots, int CP, int C, int G, long P, float eps) {
  __shared__ double red[2][256];
  const int z = blockIdx.y, n = blockIdx.x;
  const int cpg = C / G, lpg = 256 / G;
  const int g = threadIdx.x / lpg, l = threadIdx.x - g * lpg;
  const float *stats = q.stats[z] + (long)n * slots * CP * 2;
  double s1 = 0.0, s2 = 0.0;
  for (int k = l; k < slots * cpg; k += lpg) {
    const int slot = k / cpg, c = g * cpg + k - slot * cpg;
    const float2 v = *reinterpret_cast<const float2 *>(stats + ((long)slot * CP + c) * 2);
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (l < cpg || l == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < lpg; ++k) {                     // every writer of the group sums in the same order
      t1 += red[0][g * lpg + k];
      t2 += red[1][g * lpg + k];
    }
    const double cnt = (double)P * cpg;
    const double mu = t1 / cnt;
    double var = t2 / cnt - mu * mu;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int k = l; k < cpg; k += lpg) {
      const int c = g * cpg + k;
      const double sc = rstd * (double)q.gamma[z][c];
      q.scale[z][(long)n * CP + c] = (float)sc;
      q.shift[z][(long)n * CP + c] = (float)((double)q.beta[z][c] - mu * sc);
    }
  }
}

hipError_t launch_gn_finalize2(const float *const *stats, int B, int slots, int CP, int C, int G, long P, const float *const *gamma,
                               const float *const *beta, float eps, float *const *scale, float *const *shift, int nmodels,
                               hipStream_t s) {
  GnFin2 q;
  for (int z = 0; z < 2; ++z) {
    const int k = z < nmodels ? z : 0;
    q.stats[z] = stats[k];
    q.gamma[z] = gamma[k];
    q.beta[z] = beta[k];
    q.scale[z] = scale[k];
    q.shift[z] = shift[k];
  }
  hipLaunchKernelGGL(gn_finalize2_kernel, dim3((unsigned)B, (unsigned)nmodels), dim3(256), 0, s, q, slots, CP, C, G, P, eps);
  return hipGetLastError();
}

hipError_t launch_gn_finalize(const float *stats, int B, int slots, int CP, int C, int G, long P, int WM,
                              const float *gamma, const float *beta, float eps, float *scale, float *shift,
                              hipStream_t s, int fixed_ns, float *mu_out, float *rstd_out, const GnGroup *grp) {
  const GnGroup none{0x7fffffff, 0x7fffffff, {nullptr, nullptr}, {nullptr, nullptr}};
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)(B * G)), dim3(64), 0, s, stats, slots, CP, C, G, P, WM, gamma,
                     beta, eps, scale, shift, fixed_ns, mu_out, rstd_out, grp ? *grp : none);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Stem tail: GroupNorm+ReLU (as scale/shift) fused into MaxPool2d(3, stride 2, padding 1) (resnet.py:165-168).
// The affine transform is applied BEFORE the max (gamma may be negative).
__global__ __launch_bounds__(256) void gn_relu_maxpool_kernel(const float *x, const float *scale, const float *shift,
                                                            int B, int H, int W, int C, int Ho, int Wo, float *out) {
  const int Q = C >> 2;
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * Ho * Wo * Q;
  if (g >= total) return;
  const int q = (int)(g % Q);
  long r = g / Q;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int n = (int)(r / Ho);
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + (long)n * C + 4 * q);
  const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + (long)n * C + 4 * q);
  f32x4 m = {0.f, 0.f, 0.f, 0.f};   // every window holds >= 1 real pixel and relu(.) >= 0, so 0 == -inf padding
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (((long)n * H + hi) * W + wi) * C + 4 * q);
#pragma unroll
      for (int t = 0; t < 4; ++t) m[t] = fmaxf(m[t], __builtin_fmaf(v[t], sc[t], sh[t]));
    }
  }
  reinterpret_cast<f32x4 *>(out)[g] = m;
}

hipError_t launch_gn_relu_maxpool(const float *x, const float *scale, const float *shift, int B, int H, int W, int C,
                                  float *out, hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(gn_relu_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, scale, shift, B,
                     H, W, C, Ho, Wo, out);
  return hipGetLastError();
}

// The pooled stem output in the form the fused stems write it (stem_mx.hip POOL: order-preserving integer keys of the 3x3/2 window
// maximum of sgn(gamma) * x, decoded by conv_x3 MODE 3), rebuilt from a RAW stem output: the second half of the device-side
// input-contract repair (pnvo_api.hip).  Every key is a plain store, so whatever the contract-breaking launch left is replaced.
// Predicated like stem_lds_kernel<.., PAIRED>: a no-op while *only_if == 0 (only_if == nullptr: always).
__global__ __launch_bounds__(256) void pool_keys_from_raw_kernel(const float *x, const float *gamma, int B, int H, int W, int C, int Ho,
                                                               int Wo, int *keys, const int *only_if) {
  if (only_if != nullptr && *reinterpret_cast<const volatile int *>(only_if) == 0) return;
  const int Q = C >> 2;
  const long total = (long)B * Ho * Wo * Q;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
    const int q = (int)(g % Q);
    long r = g / Q;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + 4 * q);
    float m0 = -__builtin_inff(), m1 = m0, m2 = m0, m3 = m0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = 2 * ho - 1 + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = 2 * wo - 1 + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (((long)n * H + hi) * W + wi) * C + 4 * q);
        m0 = fmaxf(m0, gm.x < 0.f ? -v.x : v.x);
        m1 = fmaxf(m1, gm.y < 0.f ? -v.y : v.y);
        m2 = fmaxf(m2, gm.z < 0.f ? -v.z : v.z);
        m3 = fmaxf(m3, gm.w < 0.f ? -v.w : v.w);
      }
    }
    auto key_of = [](float f) {
      const int b = __builtin_bit_cast(int, f);
      return b >= 0 ? b : b ^ 0x7fffffff;
    };
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 kv;
    kv.x = key_of(m0);
    kv.y = key_of(m1);
    kv.z = key_of(m2);
    kv.w = key_of(m3);
    *reinterpret_cast<i32x4 *>(keys + 4 * g) = kv;
  }
}

__global__ void flag_publish_kernel(const int *dev_flag, int *host_flag) {
  if (threadIdx.x == 0 && *reinterpret_cast<const volatile int *>(dev_flag) != 0) *reinterpret_cast<volatile int *>(host_flag) = 1;
}

hipError_t launch_flag_publish(const int *dev_flag, int *host_flag, hipStream_t s) {
  hipLaunchKernelGGL(flag_publish_kernel, dim3(1), dim3(64), 0, s, dev_flag, host_flag);
  return hipGetLastError();
}

hipError_t launch_pool_keys_from_raw(const float *x, const float *gamma, int B, int H, int W, int C, int *keys, const int *only_if,
                                     hipStream_t s) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / 4);
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(pool_keys_from_raw_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, x, gamma, B, H, W, C,
                     Ho, Wo, keys, only_if);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// BasicBlock tail (resnet.py:47-55): y = relu(GN2(conv2) + residual), residual = x or GN_d(conv1x1(x)).
__global__ __launch_bounds__(256) void residual_kernel(const float *a, const float *sa, const float *ta, const float *b,
                                                     const float *sb, const float *tb, long PC, int C, long total4,
                                                     float *y) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  if (g >= total4) return;
  const long e = g * 4;
  const int n = (int)(e / PC);
  const int c = (int)(e % C);
  const f32x4 va = reinterpret_cast<const f32x4 *>(a)[g];
  const f32x4 vb = reinterpret_cast<const f32x4 *>(b)[g];
  const f32x4 s1 = *reinterpret_cast<const f32x4 *>(sa + (long)n * C + c);
  const f32x4 t1 = *reinterpret_cast<const f32x4 *>(ta + (long)n * C + c);
  f32x4 r = vb;
  if (sb != nullptr) {
    const f32x4 s2 = *reinterpret_cast<const f32x4 *>(sb + (long)n * C + c);
    const f32x4 t2 = *reinterpret_cast<const f32x4 *>(tb + (long)n * C + c);
#pragma unroll
    for (int t = 0; t < 4; ++t) r[t] = __builtin_fmaf(vb[t], s2[t], t2[t]);
  }
  f32x4 o;
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = fmaxf(__builtin_fmaf(va[t], s1[t], t1[t]) + r[t], 0.f);
  reinterpret_cast<f32x4 *>(y)[g] = o;
}

hipError_t launch_residual(const float *a, const float *sa, const float *ta, const float *b, const float *sb,
                           const float *tb, int B, long P, int C, float *y, hipStream_t s) {
  const long total4 = (long)B * P * C / 4;
  hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, a, sa, ta, b, sb, tb,
                     P * C, C, total4, y);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void apply_ss_relu_kernel(const float *x, const float *sc, const float *sh, long PC,
                                                          int C, long total4, float *y) {
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  if (g >= total4) return;
  const long e = g * 4;
  const int n = (int)(e / PC);
  const int c = (int)(e % C);
  const f32x4 v = reinterpret_cast<const f32x4 *>(x)[g];
  const f32x4 s1 = *reinterpret_cast<const f32x4 *>(sc + (long)n * C + c);
  const f32x4 t1 = *reinterpret_cast<const f32x4 *>(sh + (long)n * C + c);
  f32x4 o;
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = fmaxf(__builtin_fmaf(v[t], s1[t], t1[t]), 0.f);
  reinterpret_cast<f32x4 *>(y)[g] = o;
}

hipError_t launch_apply_ss_relu(const float *x, const float *sc, const float *sh, int B, long P, int C, float *y,
                                hipStream_t st) {
  const long total4 = (long)B * P * C / 4;
  hipLaunchKernelGGL(apply_ss_relu_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, x, sc, sh, P * C, C,
                     total4, y);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// One-hot depth (base_trainer_with_vo.py:105-115,135-167): bin i fires iff e_i <= d < e_{i+1} (last bin closed),
// edges e_i = float32(i / bins) (python-float edge rounded to the tensor dtype by torch's scalar promotion).
struct DepthEdges {
  float e[65];
};

__global__ __launch_bounds__(256) void discretize_depth_kernel(const float *depth, long n, long in_stride, int bins,
                                                             float *onehot, long out_stride, int *err_flag,
                                                             const DepthEdges ed) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float v = depth[p * in_stride];
  if (!(v >= 0.f && v <= 1.f) && err_flag != nullptr) *err_flag = 1;
  float *o = onehot + p * out_stride;
  for (int i = 0; i < bins; ++i) {
    const float lo = ed.e[i], hi = ed.e[i + 1];
    const bool hit = (i == bins - 1) ? (v >= lo && v <= hi) : (v >= lo && v < hi);
    o[i] = hit ? 1.0f : 0.0f;
  }
}

hipError_t launch_discretize_depth(const float *depth, int64_t n, int64_t in_stride, int bins, float *onehot,
                                   int64_t out_stride, int32_t *err_flag, hipStream_t s, const float *edges) {
  if (bins < 1 || bins > 64) return hipErrorInvalidValue;
  DepthEdges ed;
  for (int i = 0; i < bins; ++i) ed.e[i] = (float)((double)i * 1.0 / (double)bins);   // :105-115, rounded to fp32
  ed.e[bins] = 1.0f;
  if (edges != nullptr)                                                               // caller-defined (HOST) edges
    for (int i = 0; i <= bins; ++i) ed.e[i] = edges[i];
  hipLaunchKernelGGL(discretize_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, depth, (long)n,
                     (long)in_stride, bins, onehot, (long)out_stride, err_flag, ed);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Ego top-down view (geometry_utils.py:516-721), batched over frames, no host round trip.
// Every float32 operation of the reference is reproduced with one rounding each (__f*_rn: no FMA contraction),
// because floor()/ceil() of the results select integer histogram bins.
struct TopdownWork {       // per frame, in the caller-provided scratch
  int bbox[4];             // min_row, max_row, min_col, max_col of the non-zero crop (:582-606)
  int maxcnt;
  int pad[3];
};

size_t topdown_workspace_bytes(int N, int H, int W) {
  return (size_t)N * (sizeof(TopdownWork) + sizeof(int) * (size_t)H * W);
}

// bbox is kept in a zero-initialised, max-only encoding so that any number of workgroups per frame can contribute with
// atomicMax and the whole workspace is prepared by ONE memset:  e[0] = max(H - r), e[1] = max(r + 1), e[2] = max(W - c),
// e[3] = max(c + 1) over the non-zero pixels (0 = none)  ->  min_row = H - e[0], max_row = e[1] - 1, ... (empty: H, -1, W, -1).
__device__ __forceinline__ void topdown_decode_bbox(const TopdownWork &w, int H, int W, int &r0, int &r1, int &c0, int &c1) {
  r0 = H - w.bbox[0];
  r1 = w.bbox[1] - 1;
  c0 = W - w.bbox[2];
  c1 = w.bbox[3] - 1;
}

__global__ __launch_bounds__(256) void topdown_bbox_kernel(const float *depth, long fstride, long pstride, int H, int W,
                                                         TopdownWork *work) {
  const int n = blockIdx.y;
  int e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  const float *d = depth + (long)n * fstride;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
    if (d[(long)p * pstride] > 0.f) {   // depth >= 0, so "row/col sum > 0" == "any element > 0"
      const int r = p / W, c = p - r * W;
      e0 = max(e0, H - r);
      e1 = max(e1, r + 1);
      e2 = max(e2, W - c);
      e3 = max(e3, c + 1);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    e0 = max(e0, __shfl_xor(e0, o));
    e1 = max(e1, __shfl_xor(e1, o));
    e2 = max(e2, __shfl_xor(e2, o));
    e3 = max(e3, __shfl_xor(e3, o));
  }
  if ((threadIdx.x & 63) == 0) {
    if (e0) atomicMax(&work[n].bbox[0], e0);
    if (e1) atomicMax(&work[n].bbox[1], e1);
    if (e2) atomicMax(&work[n].bbox[2], e2);
    if (e3) atomicMax(&work[n].bbox[3], e3);
  }
}

__device__ __forceinline__ float blur_row(const float *d, long pstride, int W, int r, int c, int c0, int c1) {
  // row pass of the separable {1/4,1/2,1/4} blur on the crop [.., c0..c1], zero outside the crop
  const float m = d[((long)r * W + c) * pstride];
  const float l = c > c0 ? d[((long)r * W + c - 1) * pstride] : 0.f;
  const float rr = c < c1 ? d[((long)r * W + c + 1) * pstride] : 0.f;
  return add_rn(mul_rn(m, 0.5f), mul_rn(add_rn(l, rr), 0.25f));
}

struct TopdownConsts {
  float c[8];
};

__global__ __launch_bounds__(256) void topdown_project_kernel(const float *depth, long fstride, long pstride, int H,
                                                            int W, const TopdownConsts tc, int rows_around_center,
                                                            TopdownWork *work, int *cnt) {
  const int n = blockIdx.y;
  int r0, r1, c0, c1;
  topdown_decode_bbox(work[n], H, W, r0, r1, c0, c1);
  if (r1 < r0 || c1 < c0) return;              // all-zero frame (:522-525)
  const int hc = r1 - r0 + 1, wc = c1 - c0 + 1;
  const int half = (hc + 1) / 2;               // int(np.ceil(hc / 2)) (:609-617)
  int b0 = half - rows_around_center;
  if (b0 < 0) b0 = 0;
  int b1 = half + rows_around_center;
  if (b1 > hc) b1 = hc;
  const int t0_ = blockIdx.x * 256 + threadIdx.x;
  const bool active = t0_ < (b1 - b0) * wc;
  const int t = active ? t0_ : 0;              // (tail lanes compute pixel 0 and contribute nothing)
  const int j = b0 + t / wc, i = t % wc;       // crop coordinates
  const int r = r0 + j, c = c0 + i;            // image coordinates
  const float *d = depth + (long)n * fstride;
  // column pass over the three row-pass values (zero border outside the crop rows)
  const float m = blur_row(d, pstride, W, r, c, c0, c1);
  const float u = r > r0 ? blur_row(d, pstride, W, r - 1, c, c0, c1) : 0.f;
  const float dn = r < r1 ? blur_row(d, pstride, W, r + 1, c, c0, c1) : 0.f;
  const float db = add_rn(mul_rn(m, 0.5f), mul_rn(add_rn(u, dn), 0.25f));
  const float kinv00 = tc.c[0], kinv02 = tc.c[1], min_x = tc.c[2], x_den = tc.c[3], dscale = tc.c[4],
              z_den = tc.c[5], min_depth = tc.c[6];
  const float uu = add_rn(add_rn((float)i, (float)c0), 0.5f);        // :626-638
  const float xc = add_rn(mul_rn(kinv00, uu), kinv02);               // :648-650 (row 0 of Kinv @ [u,v,1])
  const float z = add_rn(mul_rn(db, dscale), min_depth);            // :558-560
  const float X = mul_rn(xc, z);                                        // :655
  const float xn = div_rn(sub_rn(X, min_x), x_den);                  // :676-678
  const float zn = div_rn(sub_rn(z, min_depth), z_den);              // :679-681
  const float rf = sub_rn((float)H, ceilf(mul_rn((float)H, zn)));    // :686-688
  const float cf = floorf(mul_rn((float)W, xn));                        // :689
  const long row = (long)rf, col = (long)cf;                               // .long() (:692)
  int now = 0;
  if (active && row >= 0 && row < H && col >= 0 && col < W) now = atomicAdd(&cnt[((long)n * H + row) * W + col], 1) + 1;
  // running maximum of the frame's counts (the normalisation divides by it): one atomicMax per wave, and only when the
  // wave's maximum exceeds a (possibly stale, but monotone) read of the current one
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) now = max(now, __shfl_xor(now, o));
  if ((threadIdx.x & 63) == 0 && now > __hip_atomic_load(&work[n].maxcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&work[n].maxcnt, now);
}

// out_pair > 0: frames come as (prev, cur) pairs and frame n goes to channel n & 1 of pair n >> 1:
//   out[(n >> 1) * ofstride + (n & 1) * out_pair + p * opstride]   (both top-down views of a pair tensor in one pass)
__global__ __launch_bounds__(256) void topdown_normalize_kernel(int H, int W, const TopdownWork *work, const int *cnt,
                                                              float *out, long ofstride, long opstride, long out_pair) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int mx = work[n].maxcnt;
  float v = 0.f;
  if (mx > 0) v = fminf(div_rn((float)cnt[(long)n * H * W + p], (float)mx), 1.0f);   // :543-554
  const long base = out_pair > 0 ? (long)(n >> 1) * ofstride + (long)(n & 1) * out_pair : (long)n * ofstride;
  out[base + (long)p * opstride] = v;
}

// The dataset-side twin NormalizedDepth2TopDownViewHabitat (numpy, geometry_utils.py:275-470) does the same projection
// in float64: pixel centres (float16, exact for W < 1024) through np.linalg.inv(K) in double, true depth still in
// float32 (:336-338: float32 array * python float), everything after it in double (:425-451).
struct TopdownConsts64 {
  double c[8];
};

__global__ __launch_bounds__(256) void topdown_project_f64_kernel(const float *depth, long fstride, long pstride, int H,
                                                                int W, const TopdownConsts64 tc, int rows_around_center,
                                                                TopdownWork *work, int *cnt) {
  const int n = blockIdx.y;
  int r0, r1, c0, c1;
  topdown_decode_bbox(work[n], H, W, r0, r1, c0, c1);
  if (r1 < r0 || c1 < c0) return;              // all-zero frame (:296-297)
  const int hc = r1 - r0 + 1, wc = c1 - c0 + 1;
  const int half = (hc + 1) / 2;               // int(np.ceil(hc / 2)) (:371-378)
  int b0 = half - rows_around_center;
  if (b0 < 0) b0 = 0;
  int b1 = half + rows_around_center;
  if (b1 > hc) b1 = hc;
  const int t0_ = blockIdx.x * 256 + threadIdx.x;
  const bool active = t0_ < (b1 - b0) * wc;
  const int t = active ? t0_ : 0;              // (tail lanes compute pixel 0 and contribute nothing)
  const int j = b0 + t / wc, i = t % wc;
  const int r = r0 + j, c = c0 + i;
  const float *d = depth + (long)n * fstride;
  const float m = blur_row(d, pstride, W, r, c, c0, c1);
  const float u = r > r0 ? blur_row(d, pstride, W, r - 1, c, c0, c1) : 0.f;
  const float dn = r < r1 ? blur_row(d, pstride, W, r + 1, c, c0, c1) : 0.f;
  const float db = add_rn(mul_rn(m, 0.5f), mul_rn(add_rn(u, dn), 0.25f));
  const double kinv00 = tc.c[0], kinv02 = tc.c[1], min_x = tc.c[2], x_den = tc.c[3], z_den = tc.c[5], min_depth = tc.c[6];
  const float dscale = (float)tc.c[4], min_depth_f = (float)tc.c[6];
  const double uu = (double)(i + c0) + 0.5;                              // :392-397
  const double xc = kinv00 * uu + kinv02;                                // :405 (row 0 of inv(K) @ [u, v, 1])
  const double z = (double)add_rn(mul_rn(db, dscale), min_depth_f);      // :336-338, :408
  const double X = xc * z;                                               // :409
  const double xn = (X - min_x) / x_den;                                 // :432
  const double zn = (z - min_depth) / z_den;                             // :433-435
  const double rf = (double)H - ceil((double)H * zn);                    // :443-445
  const double cf = floor((double)W * xn);                               // :446
  const long row = (long)rf, col = (long)cf;                             // .astype(np.int) (:448)
  int now = 0;
  if (active && row >= 0 && row < H && col >= 0 && col < W) now = atomicAdd(&cnt[((long)n * H + row) * W + col], 1) + 1;
  // running maximum of the frame's counts (the normalisation divides by it): one atomicMax per wave, and only when the
  // wave's maximum exceeds a (possibly stale, but monotone) read of the current one
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) now = max(now, __shfl_xor(now, o));
  if ((threadIdx.x & 63) == 0 && now > __hip_atomic_load(&work[n].maxcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&work[n].maxcnt, now);
}

hipError_t launch_topdown_f64(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                              const double *consts, int rows_around_center, float *out, int64_t out_fstride,
                              int64_t out_pstride, void *work, hipStream_t s) {
  TopdownWork *tw = reinterpret_cast<TopdownWork *>(work);
  int *cnt = reinterpret_cast<int *>(tw + N);
  hipError_t e = hipMemsetAsync(work, 0, topdown_workspace_bytes(N, H, W), s);   // counts, bbox encodings, maxima
  if (e != hipSuccess) return e;
  const int gb = N >= 64 ? 4 : 32;               // workgroups per frame of the bbox scan
  hipLaunchKernelGGL(topdown_bbox_kernel, dim3((unsigned)gb, (unsigned)N), dim3(256), 0, s, depth, (long)in_fstride,
                     (long)in_pstride, H, W, tw);
  const int band = 2 * rows_around_center < H ? 2 * rows_around_center : H;
  TopdownConsts64 tc;
  for (int k = 0; k < 7; ++k) tc.c[k] = consts[k];   // HOST array
  tc.c[7] = 0.0;
  hipLaunchKernelGGL(topdown_project_f64_kernel, dim3((unsigned)((band * W + 255) / 256), (unsigned)N), dim3(256), 0, s,
                     depth, (long)in_fstride, (long)in_pstride, H, W, tc, rows_around_center, tw, cnt);
  hipLaunchKernelGGL(topdown_normalize_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)N), dim3(256), 0, s, H, W, tw,
                     cnt, out, (long)out_fstride, (long)out_pstride, 0L);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Training-batch assembly from the arrays of one dataset chunk (StatePairRegressionDataset._process_data,
// vo/dataset/regression_geo_invariance_iter_dataset.py:205-454): uint8 rgb and float16 depth frames as stored in the
// HDF5 file -> the float32 NHWC observation pairs the model takes, for M entries; entry m reads sample src[m] and is
// (prev, cur) or, when swap[m], (cur, prev) — the geometric-inversion entries (:342-386).  The one-hot depth uses the
// caller's edges (the dataset compares float16 depth with float16-rounded edges, regression_iter_dataset.py:32-69).
__device__ __forceinline__ float half_bits_to_float(unsigned short h) {
  _Float16 v;
  __builtin_memcpy(&v, &h, 2);
  return (float)v;
}

struct DatasetPairsArgs {
  const unsigned char *rgb[2];     // prev, cur: [N][H*W*3]
  const unsigned short *depth[2];  // prev, cur: [N][H*W] float16 bits
  const float *tdv;                // [2][N][H*W] (prev frames, then cur frames) or nullptr
  const int *src, *swap;           // [M]
  long npix;                       // H*W
  long nsamples;                   // N
  int bins;
  float *o_rgb, *o_depth, *o_dd, *o_tdv;   // [M][npix][6 | 2 | 2*bins | 2], any may be nullptr
  int *err_flag;
  DepthEdges ed;
};

__global__ __launch_bounds__(256) void dataset_pairs_kernel(const DatasetPairsArgs a) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  if (p >= a.npix) return;
  const int i = a.src[m];
  const int sw = a.swap[m] != 0;
#pragma unroll
  for (int f = 0; f < 2; ++f) {                 // f: slot in the pair; which: 0 = prev frame, 1 = cur frame
    const int which = f ^ sw;
    const long e = ((long)m * a.npix + p);
    if (a.o_rgb != nullptr) {
      const unsigned char *r = a.rgb[which] + ((long)i * a.npix + p) * 3;
      float *o = a.o_rgb + e * 6 + 3 * f;
      o[0] = (float)r[0];
      o[1] = (float)r[1];
      o[2] = (float)r[2];
    }
    const float d = half_bits_to_float(a.depth[which][(long)i * a.npix + p]);
    if (a.o_depth != nullptr) a.o_depth[e * 2 + f] = d;
    if (a.o_dd != nullptr) {
      if (!(d >= 0.f && d <= 1.f) && a.err_flag != nullptr) *a.err_flag = 1;
      float *o = a.o_dd + e * (2 * a.bins) + (long)f * a.bins;
      for (int k = 0; k < a.bins; ++k) {
        const float lo = a.ed.e[k], hi = a.ed.e[k + 1];
        o[k] = ((k == a.bins - 1) ? (d >= lo && d <= hi) : (d >= lo && d < hi)) ? 1.0f : 0.0f;
      }
    }
    if (a.o_tdv != nullptr) a.o_tdv[e * 2 + f] = a.tdv != nullptr ? a.tdv[((long)which * a.nsamples + i) * a.npix + p] : 0.f;
  }
}

__global__ __launch_bounds__(256) void half_to_float_kernel(const unsigned short *src, long n, float *dst) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) dst[e] = half_bits_to_float(src[e]);
}

hipError_t launch_half_to_float(const unsigned short *src, long n, float *dst, hipStream_t s) {
  hipLaunchKernelGGL(half_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, n, dst);
  return hipGetLastError();
}

hipError_t launch_dataset_pairs(const unsigned char *prev_rgb, const unsigned char *cur_rgb, const unsigned short *prev_depth,
                                const unsigned short *cur_depth, const float *tdv_frames, const int *src, const int *swap,
                                int N, int M, int H, int W, int bins, const float *edges, float *o_rgb, float *o_depth,
                                float *o_dd, float *o_tdv, int *err_flag, hipStream_t s) {
  if (bins < 0 || bins > 64) return hipErrorInvalidValue;
  DatasetPairsArgs a;
  a.rgb[0] = prev_rgb;
  a.rgb[1] = cur_rgb;
  a.depth[0] = prev_depth;
  a.depth[1] = cur_depth;
  a.tdv = tdv_frames;
  a.src = src;
  a.swap = swap;
  a.npix = (long)H * W;
  a.nsamples = N;
  a.bins = bins;
  a.o_rgb = o_rgb;
  a.o_depth = o_depth;
  a.o_dd = bins > 0 ? o_dd : nullptr;
  a.o_tdv = o_tdv;
  a.err_flag = err_flag;
  for (int k = 0; k <= bins; ++k) a.ed.e[k] = edges ? edges[k] : (k == bins ? 1.0f : (float)((double)k / (double)bins));
  hipLaunchKernelGGL(dataset_pairs_kernel, dim3((unsigned)((a.npix + 255) / 256), (unsigned)M), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Boundary: raw simulator frames -> the observation-pair tensors of _compute_local_delta_states_from_vo
// (base_trainer_with_vo.py:172-229): rgb uint8 [n][2][H][W][3] and depth float32 [n][2][H][W] (prev, cur) ->
// rgb_pairs [n,H,W,6] (0..255 as float), depth_pairs [n,H,W,2], dd_pairs [n,H,W,2*bins] one-hot (float32 edges i/bins,
// last bin closed — the arithmetic of discretize_depth_kernel).  One thread per pixel of a pair.
struct FramePairsArgs {
  const unsigned char *rgb;
  const float *depth;
  float *o_rgb, *o_depth, *o_dd;
  int *err_flag;
  long npix;
  int bins;
  DepthEdges ed;
};

__global__ __launch_bounds__(256) void frame_pairs_kernel(const FramePairsArgs a) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  if (p >= a.npix) return;
  const long e = (long)m * a.npix + p;
  const long f0 = ((long)m * 2) * a.npix + p, f1 = f0 + a.npix;           // the pair's prev / cur frame pixels
  if (a.o_rgb != nullptr) {                                               // 24 B per pair pixel: three 8-byte stores
    const unsigned char *r0 = a.rgb + f0 * 3, *r1 = a.rgb + f1 * 3;
    float2 *o = reinterpret_cast<float2 *>(a.o_rgb + e * 6);
    o[0] = make_float2((float)r0[0], (float)r0[1]);
    o[1] = make_float2((float)r0[2], (float)r1[0]);
    o[2] = make_float2((float)r1[1], (float)r1[2]);
  }
  const float d[2] = {a.depth[f0], a.depth[f1]};
  if (a.o_depth != nullptr) *reinterpret_cast<float2 *>(a.o_depth + e * 2) = make_float2(d[0], d[1]);
  if (a.o_dd != nullptr) {
    if ((!(d[0] >= 0.f && d[0] <= 1.f) || !(d[1] >= 0.f && d[1] <= 1.f)) && a.err_flag != nullptr) *a.err_flag = 1;
    float *o = a.o_dd + e * (2 * a.bins);
    if (a.bins == 10) {                                                   // 80 B per pair pixel: five 16-byte stores
      float v[20];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const float lo = a.ed.e[k], hi = a.ed.e[k + 1];
          v[10 * f + k] = ((k == 9) ? (d[f] >= lo && d[f] <= hi) : (d[f] >= lo && d[f] < hi)) ? 1.0f : 0.0f;
        }
#pragma unroll
      for (int q = 0; q < 5; ++q) reinterpret_cast<float4 *>(o)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
      for (int f = 0; f < 2; ++f)
        for (int k = 0; k < a.bins; ++k) {
          const float lo = a.ed.e[k], hi = a.ed.e[k + 1];
          o[f * a.bins + k] = ((k == a.bins - 1) ? (d[f] >= lo && d[f] <= hi) : (d[f] >= lo && d[f] < hi)) ? 1.0f : 0.0f;
        }
    }
  }
}

hipError_t launch_frame_pairs(const unsigned char *rgb, const float *depth, int n, int H, int W, int bins, float *o_rgb,
                              float *o_depth, float *o_dd, int *err_flag, hipStream_t s) {
  if (bins < 0 || bins > 64) return hipErrorInvalidValue;
  FramePairsArgs a;
  a.rgb = rgb;
  a.depth = depth;
  a.o_rgb = rgb ? o_rgb : nullptr;
  a.o_depth = o_depth;
  a.o_dd = bins > 0 ? o_dd : nullptr;
  a.err_flag = err_flag;
  a.npix = (long)H * W;
  a.bins = bins;
  for (int k = 0; k <= bins; ++k) a.ed.e[k] = (k == bins ? 1.0f : (float)((double)k / (double)(bins > 0 ? bins : 1)));
  hipLaunchKernelGGL(frame_pairs_kernel, dim3((unsigned)((a.npix + 255) / 256), (unsigned)n), dim3(256), 0, s, a);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------------------------
// Frame ring of the batched boundary call (pnvo_ring_assemble): consecutive steps of an environment share a frame — this step's
// prev_obs is the last step's cur_obs (rl/ppo/ppo_trainer.py:724-841) — so a step uploads ONE frame per environment; the other
// half of the pair (and its top-down view) comes from a per-environment device slot, which then takes the new frame.
// Thread = one dword of the rgb frame (H*W*3/4) and one depth / top-down pixel; blockIdx.y = pair.
struct RingArgs {
  const unsigned char *up_rgb;   // [m][H][W][3] uploaded frames (nullptr: model without rgb)
  const float *up_dep, *up_tdv;  // [m][H][W]; up_tdv nullptr: model without the top-down view
  unsigned char *ring_rgb;       // [slots][H][W][3]
  float *ring_dep, *ring_tdv;    // [slots][H][W]
  const int *idx;                // [3][n]: up index of the cur frame | up index of the prev frame or -1 (take it from the ring) | ring slot or -1
  unsigned char *pair_rgb;       // [n][2][H][W][3]
  float *pair_dep, *pair_tdv;    // [n][2][H][W], [n][H][W][2]
  int n;
  long npix;
};
__global__ __launch_bounds__(256) void ring_assemble_kernel(const RingArgs a) {
  const int i = blockIdx.y;
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const int cur = a.idx[i], prev = a.idx[a.n + i], slot = a.idx[2 * a.n + i];
  const long fb = a.npix * 3;                                   // bytes of an rgb frame
  if (a.up_rgb != nullptr) {
    if ((fb & 3) == 0) {                                        // dword copies
      if (g < (fb >> 2)) {
        const unsigned *uc = reinterpret_cast<const unsigned *>(a.up_rgb + (long)cur * fb);
        unsigned *rg = slot >= 0 ? reinterpret_cast<unsigned *>(a.ring_rgb + (long)slot * fb) : nullptr;
        const unsigned pv = prev >= 0 ? reinterpret_cast<const unsigned *>(a.up_rgb + (long)prev * fb)[g] : rg[g];
        const unsigned cv = uc[g];
        unsigned *pr = reinterpret_cast<unsigned *>(a.pair_rgb + (long)i * 2 * fb);
        pr[g] = pv;
        pr[(fb >> 2) + g] = cv;
        if (rg) rg[g] = cv;
      }
    } else {                                                    // odd frame sizes: three bytes of one pixel per thread
      if (g < a.npix) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const long o = 3 * g + c;
          unsigned char *rg = slot >= 0 ? a.ring_rgb + (long)slot * fb : nullptr;
          const unsigned char pv = prev >= 0 ? a.up_rgb[(long)prev * fb + o] : rg[o];
          const unsigned char cv = a.up_rgb[(long)cur * fb + o];
          a.pair_rgb[(long)i * 2 * fb + o] = pv;
          a.pair_rgb[(long)i * 2 * fb + fb + o] = cv;
          if (rg) rg[o] = cv;
        }
      }
    }
  }
  if (g < a.npix) {
    float *rd = slot >= 0 ? a.ring_dep + (long)slot * a.npix : nullptr;
    const float pd = prev >= 0 ? a.up_dep[(long)prev * a.npix + g] : rd[g];
    const float cd = a.up_dep[(long)cur * a.npix + g];
    a.pair_dep[(long)i * 2 * a.npix + g] = pd;
    a.pair_dep[(long)i * 2 * a.npix + a.npix + g] = cd;
    if (rd) rd[g] = cd;
    if (a.up_tdv != nullptr) {
      float *rt = slot >= 0 ? a.ring_tdv + (long)slot * a.npix : nullptr;
      const float pt = prev >= 0 ? a.up_tdv[(long)prev * a.npix + g] : rt[g];
      const float ct = a.up_tdv[(long)cur * a.npix + g];
      *reinterpret_cast<f32x2 *>(a.pair_tdv + ((long)i * a.npix + g) * 2) = f32x2{pt, ct};
      if (rt) rt[g] = ct;
    }
  }
}

hipError_t launch_ring_assemble(const unsigned char *up_rgb, const float *up_dep, const float *up_tdv, unsigned char *ring_rgb,
                                float *ring_dep, float *ring_tdv, const int *idx, int n, int H, int W, unsigned char *pair_rgb,
                                float *pair_dep, float *pair_tdv, hipStream_t s) {
  RingArgs a;
  a.up_rgb = up_rgb;
  a.up_dep = up_dep;
  a.up_tdv = up_tdv;
  a.ring_rgb = ring_rgb;
  a.ring_dep = ring_dep;
  a.ring_tdv = ring_tdv;
  a.idx = idx;
  a.pair_rgb = pair_rgb;
  a.pair_dep = pair_dep;
  a.pair_tdv = pair_tdv;
  a.n = n;
  a.npix = (long)H * W;
  const long work = a.npix;                                     // (>= the dwords of an rgb frame: 3/4 npix)
  hipLaunchKernelGGL(ring_assemble_kernel, dim3((unsigned)((work + 255) / 256), (unsigned)n), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_topdown(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                          const float *consts, int rows_around_center, float *out, int64_t out_fstride,
                          int64_t out_pstride, void *work, hipStream_t s, int64_t out_pair) {
  TopdownWork *tw = reinterpret_cast<TopdownWork *>(work);
  int *cnt = reinterpret_cast<int *>(tw + N);
  hipError_t e = hipMemsetAsync(work, 0, topdown_workspace_bytes(N, H, W), s);   // counts, bbox encodings, maxima
  if (e != hipSuccess) return e;
  const int gb = N >= 64 ? 4 : 32;               // workgroups per frame of the bbox scan
  hipLaunchKernelGGL(topdown_bbox_kernel, dim3((unsigned)gb, (unsigned)N), dim3(256), 0, s, depth, (long)in_fstride,
                     (long)in_pstride, H, W, tw);
  const int band = 2 * rows_around_center < H ? 2 * rows_around_center : H;
  TopdownConsts tc;
  for (int k = 0; k < 7; ++k) tc.c[k] = consts[k];   // HOST array
  tc.c[7] = 0.f;
  hipLaunchKernelGGL(topdown_project_kernel, dim3((unsigned)((band * W + 255) / 256), (unsigned)N), dim3(256), 0, s,
                     depth, (long)in_fstride, (long)in_pstride, H, W, tc, rows_around_center, tw, cnt);
  hipLaunchKernelGGL(topdown_normalize_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)N), dim3(256), 0, s, H, W, tw,
                     cnt, out, (long)out_fstride, (long)out_pstride, (long)out_pair);
  return hipGetLastError();
}

}  // namespace pnvo
