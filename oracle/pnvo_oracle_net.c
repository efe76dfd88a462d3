/*
 * oracle/pnvo_oracle_net.c — CPU restatement of the PointNav-VO network forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pointnav-vo_amd/, include/, the
 * C-ABI library) links, imports or calls this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker / baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
 * golden vectors captured from the *imported reference modules* (tests/golden/gen_golden.py,
 * run in the build container where /root/reference exists).
 *
 * Built twice by oracle/Makefile: REAL=float -> liboracle_f32.so, REAL=double -> liboracle_f64.so.
 * Layout is NHWC throughout (the reference is NCHW; values are identical, only strides differ).
 * Weights are taken in the reference's own state_dict layout (OIHW for conv, [N][K] for linear),
 * so the checker consumes exactly what `model.state_dict()` holds.
 *
 * Reference citations (relative to /root/reference):
 *   input assembly            pointnav_vo/vo/models/vo_cnn.py:110-174
 *   RunningMeanAndVar (eval)  pointnav_vo/model_utils/running_mean_and_var.py:62-63
 *   conv3x3 / conv1x1 / stem  pointnav_vo/model_utils/visual_encoders/resnet.py:11-26,156-167
 *   GroupNorm                 torch.nn.GroupNorm as used at resnet.py:39,42,165,194; vo_cnn.py:93
 *   MaxPool 3x3 s2 p1         resnet.py:168
 *   BasicBlock add + ReLU     resnet.py:47-55
 *   Flatten / Linear head     pointnav_vo/utils/misc_utils.py:45-47; vo_cnn.py:216-227
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifndef REAL
#define REAL float
#endif

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_sizeof_real(void) { return (int)sizeof(REAL); }

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/*
 * Input assembly + whitening.  vo_cnn.py:114-176:
 *   rgb is divided by 255 (:118); every modality is split in channel halves (prev | cur) and the
 *   halves are interleaved as [prev_rgb, prev_d, prev_dd, prev_tdv, cur_rgb, cur_d, cur_dd, cur_tdv]
 *   (:169-174); then RunningMeanAndVar eval: (x - mean) / sqrt(max(var, 1e-2))
 *   (running_mean_and_var.py:62-63).  Any of the four inputs may be NULL (modality absent).
 *   n_rgb / n_depth / n_dd / n_tdv are the PAIR channel counts (6, 2, 2*bins, 2) or 0.
 */
void orc_assemble_whiten(const REAL *rgb, const REAL *depth, const REAL *dd, const REAL *tdv,
                         int B, int H, int W, int n_rgb, int n_depth, int n_dd, int n_tdv,
                         const REAL *mean, const REAL *var, int normalize, REAL *out) {
  const int C = n_rgb + n_depth + n_dd + n_tdv;
  const int hr = n_rgb / 2, hd = n_depth / 2, hdd = n_dd / 2, ht = n_tdv / 2;
  const int half = hr + hd + hdd + ht;
  const long P = (long)B * H * W;
  REAL *stdev = (REAL *)malloc(sizeof(REAL) * (size_t)C);
  for (int c = 0; c < C; ++c) {
    REAL v = normalize ? var[c] : (REAL)1;
    if (normalize && v < (REAL)1e-2) v = (REAL)1e-2;
    stdev[c] = normalize ? (REAL)sqrt((double)v) : (REAL)1;
    if (sizeof(REAL) == 4 && normalize) stdev[c] = (REAL)sqrtf((float)v);
  }
#pragma omp parallel for schedule(static)
  for (long p = 0; p < P; ++p) {
    REAL *o = out + p * C;
    for (int s = 0; s < 2; ++s) { /* s = 0 prev, 1 cur */
      int c = s * half;
      for (int k = 0; k < hr; ++k) o[c++] = rgb[p * n_rgb + s * hr + k] / (REAL)255.0;
      for (int k = 0; k < hd; ++k) o[c++] = depth[p * n_depth + s * hd + k];
      for (int k = 0; k < hdd; ++k) o[c++] = dd[p * n_dd + s * hdd + k];
      for (int k = 0; k < ht; ++k) o[c++] = tdv[p * n_tdv + s * ht + k];
    }
    if (normalize)
      for (int c = 0; c < C; ++c) o[c] = (o[c] - mean[c]) / stdev[c];
  }
  free(stdev);
}

/*
 * Bias-free cross-correlation, zero padding, PyTorch floor output size (resnet.py:11-26,156-163;
 * vo_cnn.py:85-92).  x [B,H,W,Cin] NHWC, w OIHW [Cout,Cin,KH,KW], out [B,Ho,Wo,Cout] NHWC.
 * Accumulation order per output element: kh, kw, ci (ascending), in REAL.
 */
void orc_conv2d(const REAL *x, int B, int H, int W, int Cin, const REAL *w_oihw, int Cout, int KH,
                int KW, int stride, int pad, REAL *out) {
  const int Ho = (H + 2 * pad - KH) / stride + 1;
  const int Wo = (W + 2 * pad - KW) / stride + 1;
  /* re-lay weights to [KH][KW][Cin][Cout] so the innermost loop is a contiguous axpy over Cout */
  REAL *wk = (REAL *)malloc(sizeof(REAL) * (size_t)KH * KW * Cin * Cout);
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i)
      for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw)
          wk[(((size_t)kh * KW + kw) * Cin + i) * Cout + o] =
              w_oihw[(((size_t)o * Cin + i) * KH + kh) * KW + kw];
  const long rows = (long)B * Ho;
#pragma omp parallel for schedule(dynamic, 4)
  for (long r = 0; r < rows; ++r) {
    const int n = (int)(r / Ho), ho = (int)(r % Ho);
    REAL *acc = (REAL *)malloc(sizeof(REAL) * (size_t)Cout);
    for (int wo = 0; wo < Wo; ++wo) {
      for (int o = 0; o < Cout; ++o) acc[o] = (REAL)0;
      for (int kh = 0; kh < KH; ++kh) {
        const int hi = ho * stride - pad + kh;
        if (hi < 0 || hi >= H) continue;
        for (int kw = 0; kw < KW; ++kw) {
          const int wi = wo * stride - pad + kw;
          if (wi < 0 || wi >= W) continue;
          const REAL *xp = x + (((size_t)n * H + hi) * W + wi) * Cin;
          const REAL *wp = wk + ((size_t)kh * KW + kw) * Cin * Cout;
          for (int i = 0; i < Cin; ++i) {
            const REAL xv = xp[i];
            const REAL *wr = wp + (size_t)i * Cout;
            for (int o = 0; o < Cout; ++o) acc[o] += xv * wr[o];
          }
        }
      }
      REAL *op = out + (((size_t)n * Ho + ho) * Wo + wo) * Cout;
      for (int o = 0; o < Cout; ++o) op[o] = acc[o];
    }
    free(acc);
  }
  free(wk);
}

/*
 * torch.nn.GroupNorm(G, C, eps, affine=True) on NHWC data, in place; optional fused ReLU
 * (the reference always follows GN by ReLU except before the residual add, resnet.py:37-43).
 * Statistics: mean and BIASED variance over the (C/G)*P elements of one (sample, group);
 * accumulated in double regardless of REAL (the checker should not add its own noise).
 */
void orc_groupnorm(REAL *x, int B, long P, int C, int G, const REAL *gamma, const REAL *beta,
                   double eps, int relu) {
  const int cpg = C / G;
#pragma omp parallel for schedule(static) collapse(2)
  for (int n = 0; n < B; ++n) {
    for (int g = 0; g < G; ++g) {
      REAL *xs = x + (size_t)n * P * C;
      double s = 0.0;
      for (long p = 0; p < P; ++p)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) s += (double)xs[p * C + c];
      const double cnt = (double)P * cpg;
      const double mu = s / cnt;
      double v = 0.0;
      for (long p = 0; p < P; ++p)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
          const double d = (double)xs[p * C + c] - mu;
          v += d * d;
        }
      const double rstd = 1.0 / sqrt(v / cnt + eps);
      for (long p = 0; p < P; ++p)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
          REAL y = (REAL)(((double)xs[p * C + c] - mu) * rstd) * gamma[c] + beta[c];
          if (relu && y < (REAL)0) y = (REAL)0;
          xs[p * C + c] = y;
        }
    }
  }
}

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (resnet.py:168); padding acts as -inf. */
void orc_maxpool3x3s2p1(const REAL *x, int B, int H, int W, int C, REAL *out) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
#pragma omp parallel for schedule(static) collapse(2)
  for (int n = 0; n < B; ++n)
    for (int ho = 0; ho < Ho; ++ho)
      for (int wo = 0; wo < Wo; ++wo) {
        REAL *op = out + (((size_t)n * Ho + ho) * Wo + wo) * C;
        for (int c = 0; c < C; ++c) op[c] = (REAL)-INFINITY;
        for (int kh = 0; kh < 3; ++kh) {
          const int hi = ho * 2 - 1 + kh;
          if (hi < 0 || hi >= H) continue;
          for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if (wi < 0 || wi >= W) continue;
            const REAL *xp = x + (((size_t)n * H + hi) * W + wi) * C;
            for (int c = 0; c < C; ++c)
              if (xp[c] > op[c]) op[c] = xp[c];
          }
        }
      }
}

/* BasicBlock tail: relu(out + residual) (resnet.py:55), in place on a. */
void orc_add_relu(REAL *a, const REAL *b, long n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) {
    REAL y = a[i] + b[i];
    a[i] = y < (REAL)0 ? (REAL)0 : y;
  }
}

/*
 * Flatten in NCHW order (misc_utils.py:45-47 applied to an NCHW tensor, vo_cnn.py:217):
 * in NHWC [B,H,W,C] -> out [B, C*H*W] with index c*H*W + h*W + w.
 */
void orc_flatten_nchw(const REAL *x, int B, int H, int W, int C, REAL *out) {
  for (int n = 0; n < B; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        for (int c = 0; c < C; ++c)
          out[(size_t)n * C * H * W + ((size_t)c * H + h) * W + w] =
              x[(((size_t)n * H + h) * W + w) * C + c];
}

/* nn.Linear(K, N): y = x W^T + b, optional ReLU (vo_cnn.py:219-220,225). w is [N][K]. */
void orc_linear(const REAL *x, int B, int K, const REAL *w, const REAL *bias, int N, int relu,
                REAL *out) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int n = 0; n < B; ++n)
    for (int j = 0; j < N; ++j) {
      REAL acc = (REAL)0;
      const REAL *xr = x + (size_t)n * K, *wr = w + (size_t)j * K;
      for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
      acc += bias ? bias[j] : (REAL)0;
      if (relu && acc < (REAL)0) acc = (REAL)0;
      out[(size_t)n * N + j] = acc;
    }
}
