/*
 * oracle/pnvo_oracle_pre.c — CPU restatement of the VO pre-processing (one-hot depth, ego top-down view).
 *
 * TEST INFRASTRUCTURE ONLY (see pnvo_oracle_net.c header).  Always fp32: these functions restate the
 * reference's float32 torch op sequence one rounding at a time, because floor()/ceil() of the results
 * decide integer bins.  Compile with -ffp-contract=off (oracle/Makefile does).
 *
 * Parity status:
 *   one-hot depth ............ PINNED against the reference's _discretize_depth_func
 *                              (pointnav_vo/rl/common/base_trainer_with_vo.py:135-167; edges :105-115).
 *   top-down view ............ PINNED *except the blur*: steps before/after the 3x3 Gaussian blur are checked
 *                              bit-exactly against NormalizedDepth2TopDownViewHabitatTorch
 *                              (pointnav_vo/utils/geometry_utils.py:491-721) by feeding both sides the same blur.
 *   3x3 Gaussian blur ........ PARITY UNPINNED.  The arithmetic lives in OpenCV (opencv-python, unpinned in
 *                              environment.yml:21; call site geometry_utils.py:529-535), which is absent from
 *                              /root/reference and from this image.  Restated from OpenCV's published
 *                              algorithm: ksize=3, sigma<=0 -> fixed kernel {1/4, 1/2, 1/4}
 *                              (getGaussianKernel small table), separable row pass then column pass in
 *                              float32, symmetric taps added first, BORDER_ISOLATED(=BORDER_CONSTANT 0 on the
 *                              cropped ROI).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/*
 * base_trainer_with_vo.py:105-115: edges e_i = i * 1.0 / bins (python float), e_bins = 1.0.
 * :135-167: out[p][i] = 1 iff e_i <= d < e_{i+1} (i < bins-1) or e_i <= d <= e_{i+1} (i = bins-1).
 * The comparison is float32 tensor vs python scalar => the scalar is rounded to float32.
 * Returns the number of pixels that fired exactly one bin (the reference asserts this == n, :163).
 */
long orc_discretize_depth(const float *d, long n, int bins, float *out) {
  float edges[65];
  if (bins > 64) return -1;
  for (int i = 0; i < bins; ++i) edges[i] = (float)((double)i * 1.0 / (double)bins);
  edges[bins] = 1.0f;
  long fired = 0;
  memset(out, 0, sizeof(float) * (size_t)n * bins);
  for (long p = 0; p < n; ++p) {
    const float v = d[p];
    int cnt = 0;
    for (int i = 0; i < bins; ++i) {
      int hit = (i == bins - 1) ? (v >= edges[i] && v <= edges[i + 1]) : (v >= edges[i] && v < edges[i + 1]);
      if (hit) {
        out[p * bins + i] = 1.0f;
        ++cnt;
      }
    }
    fired += (cnt == 1);
  }
  return fired;
}

/*
 * Constants of NormalizedDepth2TopDownViewHabitatTorch, restated in closed form:
 *   geometry_utils.py:562-568  f = (W/2)/tan(hfov/2) in float64, stored as float32 in K
 *   :570-580                    max_x = (Kinv @ [W-0.5, 0, 1])[0] * max_depth ; min_x = -max_x
 *   :676-681                    x_den = (max_x - min_x) * (1 + eps) ; z_den = (max_d - min_d) * (1 + eps)
 * c[0]=kinv00 c[1]=kinv02 c[2]=min_x c[3]=x_den c[4]=depth_scale c[5]=z_den c[6]=min_depth
 * The inverse of the upper-triangular K is taken in closed form (1/f, -u0/f); the golden generator compares
 * these against torch.inverse and the host-side product code uses torch.inverse like the reference does.
 */
void orc_topdown_consts(int H, int W, double hfov_rad, double min_depth, double max_depth, double eps,
                        float *c) {
  (void)H;
  const float f = (float)(((double)W / 2.0) / tan(hfov_rad / 2.0));
  const float u0 = (float)((double)W / 2.0);
  const float kinv00 = 1.0f / f;
  const float kinv02 = -(u0 * kinv00);
  volatile float t = kinv00 * ((float)W - 0.5f);
  const float xc = t + kinv02;
  const float max_x = xc * (float)max_depth;
  const float min_x = -max_x;
  const float x_range = max_x - min_x;
  c[0] = kinv00;
  c[1] = kinv02;
  c[2] = min_x;
  c[3] = x_range * (float)(1.0 + eps);
  c[4] = (float)(max_depth - min_depth);
  c[5] = (float)((max_depth - min_depth) * (1.0 + eps));
  c[6] = (float)min_depth;
}

/* 3x3 {1/4,1/2,1/4} separable blur, zero border, on a [h][w] float image with row stride `ld`. */
static void blur3x3_zero_border(const float *src, int ld, int h, int w, float *dst) {
  float *tmp = (float *)malloc(sizeof(float) * (size_t)h * w);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const float l = c > 0 ? src[(size_t)r * ld + c - 1] : 0.0f;
      const float rr = c + 1 < w ? src[(size_t)r * ld + c + 1] : 0.0f;
      const float m = src[(size_t)r * ld + c];
      const float pair = l + rr;
      tmp[(size_t)r * w + c] = m * 0.5f + pair * 0.25f;
    }
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const float u = r > 0 ? tmp[(size_t)(r - 1) * w + c] : 0.0f;
      const float d = r + 1 < h ? tmp[(size_t)(r + 1) * w + c] : 0.0f;
      const float m = tmp[(size_t)r * w + c];
      const float pair = u + d;
      dst[(size_t)r * w + c] = m * 0.5f + pair * 0.25f;
    }
  free(tmp);
}

/*
 * gen_top_down_view (geometry_utils.py:516-556) for one [H][W] normalized depth frame.
 *   bbox[4] (out, may be NULL): min_row, max_row, min_col, max_col of the non-zero border crop (:582-606)
 *   blur_in (may be NULL): if given, a [h'][w'] already-blurred crop to use INSTEAD of the internal blur
 *       (lets the tests pin everything around the unpinned OpenCV step);
 *   blur_out (may be NULL): receives the [h'][w'] blurred crop actually used (caller sizes it H*W).
 *   cnt_out (may be NULL): the raw integer counts [H][W] (as int).
 * Returns 0 on success, 1 if the depth frame was all zero (output all zero, :522-525).
 */
int orc_topdown(const float *depth, int H, int W, const float *c, int rows_around_center,
                const float *blur_in, float *out, int *bbox, float *blur_out, int *cnt_out) {
  const float kinv00 = c[0], kinv02 = c[1], min_x = c[2], x_den = c[3], dscale = c[4], z_den = c[5],
              min_depth = c[6];
  memset(out, 0, sizeof(float) * (size_t)H * W);
  if (cnt_out) memset(cnt_out, 0, sizeof(int) * (size_t)H * W);
  /* :582-606 — first/last row and column whose sum is > 0 (depth >= 0, so "any element > 0") */
  int min_row = H - 1, max_row = 0, min_col = W - 1, max_col = 0;
  for (int i = 0; i < H; ++i) {
    float s = 0.0f;
    for (int j = 0; j < W; ++j) s += depth[(size_t)i * W + j];
    if (s > 0.0f) { min_row = i; break; }
  }
  for (int i = H - 1; i >= 0; --i) {
    float s = 0.0f;
    for (int j = 0; j < W; ++j) s += depth[(size_t)i * W + j];
    if (s > 0.0f) { max_row = i; break; }
  }
  for (int j = 0; j < W; ++j) {
    float s = 0.0f;
    for (int i = 0; i < H; ++i) s += depth[(size_t)i * W + j];
    if (s > 0.0f) { min_col = j; break; }
  }
  for (int j = W - 1; j >= 0; --j) {
    float s = 0.0f;
    for (int i = 0; i < H; ++i) s += depth[(size_t)i * W + j];
    if (s > 0.0f) { max_col = j; break; }
  }
  if (bbox) { bbox[0] = min_row; bbox[1] = max_row; bbox[2] = min_col; bbox[3] = max_col; }
  const int hc = max_row - min_row + 1, wc = max_col - min_col + 1;
  if (hc <= 0 || wc <= 0) return 1; /* torch.numel(crop) == 0 */

  float *db = (float *)malloc(sizeof(float) * (size_t)hc * wc);
  if (blur_in)
    memcpy(db, blur_in, sizeof(float) * (size_t)hc * wc);
  else
    blur3x3_zero_border(depth + (size_t)min_row * W + min_col, W, hc, wc, db);
  if (blur_out) memcpy(blur_out, db, sizeof(float) * (size_t)hc * wc);

  /* :609-617 — rows around the centre of the CROP */
  const int half = (hc + 1) / 2; /* int(np.ceil(hc / 2)) */
  int r0 = half - rows_around_center; if (r0 < 0) r0 = 0;
  int r1 = half + rows_around_center; if (r1 > hc) r1 = hc;

  int *cnt = (int *)calloc((size_t)H * W, sizeof(int));
  int maxcnt = 0;
  for (int j = r0; j < r1; ++j)
    for (int i = 0; i < wc; ++i) {
      /* :626-655 */
      volatile float u = (float)i + (float)min_col;
      u = u + 0.5f;
      volatile float t = kinv00 * u;      /* row 0 of Kinv @ [u, v, 1]: a*u (+ 0*v) + c, k ascending */
      volatile float xc = t + kinv02;
      volatile float z = db[(size_t)j * wc + i] * dscale;
      z = z + min_depth;
      volatile float X = xc * z;
      /* :676-681 */
      volatile float xn = X - min_x;
      xn = xn / x_den;
      volatile float zn = z - min_depth;
      zn = zn / z_den;
      /* :686-692 */
      volatile float rf = (float)H * zn;
      rf = (float)H - ceilf(rf);
      volatile float cf = (float)W * xn;
      cf = floorf(cf);
      const long row = (long)rf, col = (long)cf;
      if (row >= 0 && row < H && col >= 0 && col < W) {
        const int v = ++cnt[(size_t)row * W + col];
        if (v > maxcnt) maxcnt = v;
      }
    }
  if (cnt_out) memcpy(cnt_out, cnt, sizeof(int) * (size_t)H * W);
  /* :543-554 */
  if (maxcnt > 0) {
    const float bound = (float)maxcnt;
    for (size_t p = 0; p < (size_t)H * W; ++p) {
      float v = (float)cnt[p] / bound;
      out[p] = v > 1.0f ? 1.0f : v;
    }
  }
  free(cnt);
  free(db);
  return 0;
}
