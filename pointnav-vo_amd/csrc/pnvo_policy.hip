// pnvo_policy.hip — the navigation policy's per-step forward, gfx950 only.
//
// PointNavResNetPolicy.act (reference: pointnav_vo/rl/policies/policy.py:29-46, resnet_policy.py:26-58,177-282):
//   depth [B,H,W,1] -> avg_pool2d(2) -> GroupNorm-ResNet18 (baseplanes 32) -> compression conv + GN(1) + ReLU
//   -> Flatten + Linear + ReLU (visual_fc)                              | the VO path's kernels, via a pnvo handle
//   x = [visual (hidden) | tgt_embeding([rho, cos(-phi), sin(-phi)]) (32) | prev_action_embedding (32)]
//   -> 2-layer LSTM with the hidden state masked at episode starts (model_utils/rnns/rnn_state_encoder.py:63-79)
//   -> action logits (CategoricalNet, utils/misc_utils.py:67-78) and value (CriticHead, policy.py:66-74).
// The visual encoder is the SAME kernel set as the VO model: a pnvo handle configured with (W/2, H/2), one depth
// modality of 2 channels [pooled depth | 0] and no whitening (normalize_visual_inputs is False for the depth-only policy,
// ddppo_trainer.py:118-121).  The recurrent part is tiny and weight-bandwidth-bound at B = number of environments
// (9.4 MB of LSTM weights per step), so its Linears are wave-per-output-row dot products on the vector ALU, not MFMA.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pnvo.h"
#include "pnvo_internal.h"
#include "pnvo_model.h"

namespace pnvo {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// F.avg_pool2d(x, 2) of a 1-channel NHWC frame (floor: odd trailing row / column dropped) -> [N,H/2,W/2,2] with
// channel 1 = 0 (the encoder's stem consumes 2-channel pieces).
__global__ __launch_bounds__(256) void avgpool2_kernel(const float *d, int N, int H, int W, float *out) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = (long)N * Ho * Wo;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % Wo);
  const int y = (int)((e / Wo) % Ho);
  const long n = e / ((long)Wo * Ho);
  const float *p = d + (n * H + 2 * y) * W + 2 * x;
  const float s = ((p[0] + p[1]) + p[W]) + p[W + 1];
  out[2 * e] = s * 0.25f;
  out[2 * e + 1] = 0.f;
}

// LSTM input x [B, hidden + 64]: visual | Linear(3 -> 32)(rho, cos(-phi), sin(-phi)) | Embedding((a + 1) * mask)
__global__ __launch_bounds__(256) void policy_inputs_kernel(const float *visual, const float *goal, const int64_t *prev,
                                                          const float *masks, const float *tgt_w, const float *tgt_b,
                                                          const float *emb, int n_emb, int B, int hidden, float *x) {
  const int K = hidden + 64;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * K) return;
  const int b = (int)(e / K), k = (int)(e % K);
  float v;
  if (k < hidden) {
    v = visual[(long)b * hidden + k];
  } else if (k < hidden + 32) {
    const int j = k - hidden;
    const float rho = goal[2 * b], phi = goal[2 * b + 1];
    const float g0 = rho, g1 = cosf(-phi), g2 = sinf(-phi);
    v = __builtin_fmaf(tgt_w[3 * j + 2], g2, __builtin_fmaf(tgt_w[3 * j + 1], g1, tgt_w[3 * j] * g0)) + tgt_b[j];
  } else {
    const int j = k - hidden - 32;
    long row = (long)(((float)prev[b] + 1.0f) * masks[b]);     // ((prev_actions.float() + 1) * masks).long()
    if (row < 0) row = 0;
    if (row >= n_emb) row = n_emb - 1;
    v = emb[row * 32 + j];
  }
  x[e] = v;
}

// One LSTM layer in one launch (round 6; it was two linear_rows launches + lstm_cell): workgroup = hidden unit j, wave = gate (i, f, g, o),
// gate[b] = (h_prev[b] . W_hh[n] * mask[b] + b_hh[n]) + (x[b] . W_ih[n] + b_ih[n]) with n = gate * Hd + j — one wave per gate row, all rows b, W in torch's
// [N][K] layout, K % 4 == 0; torch.nn.LSTM gate order, c_prev masked like h_prev — then the cell for (b, j) by the first lanes.
__global__ __launch_bounds__(256) void lstm_layer_kernel(const float *x, int K, const float *w_ih, const float *b_ih, const float *h_prev,
                                                       const float *w_hh, const float *b_hh, const float *c_prev, const float *masks,
                                                       int B, int Hd, float *h_out, float *c_out) {
  __shared__ float sg[4][64];
  const int lane = threadIdx.x & 63, gate = (int)(threadIdx.x >> 6), j = blockIdx.x;
  const int n = gate * Hd + j;
  const f32x4 *wi = reinterpret_cast<const f32x4 *>(w_ih + (long)n * K), *wh = reinterpret_cast<const f32x4 *>(w_hh + (long)n * Hd);
  const int K4 = K >> 2, H4 = Hd >> 2;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = min(64, B - b0);
    for (int bb = 0; bb < nb; ++bb) {
      const int b = b0 + bb;
      const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (long)b * K), *hr = reinterpret_cast<const f32x4 *>(h_prev + (long)b * Hd);
      float s = 0.f, u = 0.f;
      for (int k = lane; k < K4; k += 64) {
        const f32x4 w = wi[k], v = xr[k];
        s = __builtin_fmaf(w[0], v[0], s);
        s = __builtin_fmaf(w[1], v[1], s);
        s = __builtin_fmaf(w[2], v[2], s);
        s = __builtin_fmaf(w[3], v[3], s);
      }
      for (int k = lane; k < H4; k += 64) {
        const f32x4 w = wh[k], v = hr[k];
        u = __builtin_fmaf(w[0], v[0], u);
        u = __builtin_fmaf(w[1], v[1], u);
        u = __builtin_fmaf(w[2], v[2], u);
        u = __builtin_fmaf(w[3], v[3], u);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o);
        u += __shfl_xor(u, o);
      }
      if (lane == 0) sg[gate][bb] = (u * masks[b] + b_hh[n]) + (s + b_ih[n]);
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      const int b = b0 + (int)threadIdx.x;
      const long e = (long)b * Hd + j;
      const float i_ = 1.f / (1.f + expf(-sg[0][threadIdx.x]));
      const float f_ = 1.f / (1.f + expf(-sg[1][threadIdx.x]));
      const float g_ = tanhf(sg[2][threadIdx.x]);
      const float o_ = 1.f / (1.f + expf(-sg[3][threadIdx.x]));
      const float c = f_ * (c_prev[e] * masks[b]) + i_ * g_;
      c_out[e] = c;
      h_out[e] = o_ * tanhf(c);
    }
    __syncthreads();
  }
}

// action logits and value in one launch: rows 0 .. n_actions - 1 of the actor, then the critic's single row (one wave per output row)
__global__ __launch_bounds__(256) void policy_heads_kernel(const float *x, const float *act_w, const float *act_b, const float *cr_w,
                                                         const float *cr_b, int B, int K, int n_actions, float *logits, float *value) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (n > n_actions) return;
  const bool critic = n == n_actions;
  if ((critic ? value : logits) == nullptr) return;
  const f32x4 *wr = reinterpret_cast<const f32x4 *>(critic ? cr_w : act_w + (long)n * K);
  const float bias = critic ? cr_b[0] : act_b[n];
  const int K4 = K >> 2;
  for (int b = 0; b < B; ++b) {
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (long)b * K);
    float s = 0.f;
    for (int k = lane; k < K4; k += 64) {
      const f32x4 w = wr[k], v = xr[k];
      s = __builtin_fmaf(w[0], v[0], s);
      s = __builtin_fmaf(w[1], v[1], s);
      s = __builtin_fmaf(w[2], v[2], s);
      s = __builtin_fmaf(w[3], v[3], s);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      if (critic) value[b] = s + bias;
      else logits[(long)b * n_actions + n] = s + bias;
    }
  }
}

struct Policy {
  pnvo_policy_config cfg;
  int device = 0;
  pnvo_handle enc = nullptr;
  bool loaded = false;
  // device weights (torch layouts)
  float *emb = nullptr, *tgt_w = nullptr, *tgt_b = nullptr;
  std::vector<float *> w_ih, w_hh, b_ih, b_hh;
  float *act_w = nullptr, *act_b = nullptr, *cr_w = nullptr, *cr_b = nullptr;
  // workspace
  int cap = 0;
  float *pooled = nullptr, *visual = nullptr, *x = nullptr;
};

int pfail(int code, const std::string &msg) { return pnvo_fail(nullptr, code, msg); }

#define PCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) return pfail(PNVO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

void dfree(float *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

const float *find(const std::map<std::string, const pnvo_tensor_desc *> &by, const float *blob, size_t n,
                  const std::string &name, std::vector<int64_t> shape, int *rc) {
  auto it = by.find(name);
  if (it == by.end()) {
    *rc = pfail(PNVO_ERR_WEIGHTS, "policy state_dict is missing tensor '" + name + "'");
    return nullptr;
  }
  const pnvo_tensor_desc *d = it->second;
  size_t cnt = 1;
  bool ok = d->ndim == (int)shape.size();
  for (int k = 0; ok && k < d->ndim; ++k) {
    ok = d->shape[k] == shape[k];
    cnt *= (size_t)d->shape[k];
  }
  if (!ok || d->offset + cnt > n) {
    *rc = pfail(PNVO_ERR_WEIGHTS, "policy tensor '" + name + "' has the wrong shape");
    return nullptr;
  }
  return blob + d->offset;
}

int upload(float *&dst, const float *src, size_t n) {
  dfree(dst);
  PCHK(hipMalloc((void **)&dst, n * sizeof(float)));
  PCHK(hipMemcpy(dst, src, n * sizeof(float), hipMemcpyHostToDevice));
  return PNVO_OK;
}

}  // namespace
}  // namespace pnvo

using namespace pnvo;

extern "C" {

struct pnvo_policy_s {
  Policy p;
};

int pnvo_avgpool2(const float *depth, int N, int H, int W, float *out, void *stream) {
  if (!depth || !out || N < 0 || H < 2 || W < 2) return pfail(PNVO_ERR_ARG, "bad argument");
  if (N == 0) return PNVO_OK;
  const long total = (long)N * (H / 2) * (W / 2);
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth, N, H, W,
                     out);
  PCHK(hipGetLastError());
  return PNVO_OK;
}

int pnvo_policy_create(const pnvo_policy_config *cfg, int device, pnvo_policy_handle *out) {
  if (!cfg || !out) return pfail(PNVO_ERR_ARG, "null argument");
  if (cfg->width < 64 || cfg->height < 64 || cfg->hidden % 4 != 0 || cfg->rnn_layers < 1 || cfg->rnn_layers > 4 ||
      cfg->n_actions < 1 || cfg->n_actions > 32)
    return pfail(PNVO_ERR_ARG, "unsupported policy configuration");
  pnvo_policy_s *h = new pnvo_policy_s();
  h->p.cfg = *cfg;
  h->p.device = device;
  pnvo_config ec;
  std::memset(&ec, 0, sizeof(ec));
  ec.width = cfg->width / 2;             // after F.avg_pool2d(x, 2)  (resnet_policy.py:168)
  ec.height = cfg->height / 2;
  ec.n_depth = 2;                        // [pooled depth | 0]
  ec.baseplanes = cfg->baseplanes;
  ec.hidden = cfg->hidden;
  ec.out_dim = 1;                        // unused head (the policy stops at the hidden vector)
  ec.normalize = 0;
  ec.n_acts = 4;
  ec.flat_size = cfg->flat_size;
  ec.max_batch = 16;
  const int rc = pnvo_create(&ec, device, &h->p.enc);
  if (rc != PNVO_OK) {
    delete h;
    return rc;
  }
  h->p.w_ih.assign(cfg->rnn_layers, nullptr);
  h->p.w_hh.assign(cfg->rnn_layers, nullptr);
  h->p.b_ih.assign(cfg->rnn_layers, nullptr);
  h->p.b_hh.assign(cfg->rnn_layers, nullptr);
  *out = h;
  return PNVO_OK;
}

int pnvo_policy_load_weights(pnvo_policy_handle h, const float *blob, size_t n_floats, const pnvo_tensor_desc *toc,
                             int ntoc) {
  if (!h || !blob || !toc) return pfail(PNVO_ERR_ARG, "null argument");
  Policy &p = h->p;
  PCHK(hipSetDevice(p.device));
  std::map<std::string, const pnvo_tensor_desc *> by;
  for (int k = 0; k < ntoc; ++k) by[toc[k].name] = &toc[k];
  const pnvo_policy_config &c = p.cfg;
  const int Hd = c.hidden, K0 = Hd + 64;
  int rc = PNVO_OK;
  // ---- visual encoder + visual_fc: re-key into the VO model's naming, pad the stem to 2 input channels
  {
    const std::string pre = "net.visual_encoder.";
    std::vector<float> eblob;
    std::vector<std::string> names;
    std::vector<pnvo_tensor_desc> etoc;
    auto push = [&](const std::string &name, const float *src, std::vector<int64_t> shape, size_t cnt) {
      pnvo_tensor_desc d;
      std::memset(&d, 0, sizeof(d));
      d.offset = eblob.size();
      d.ndim = (int)shape.size();
      for (size_t k = 0; k < shape.size(); ++k) d.shape[k] = shape[k];
      names.push_back(name);
      etoc.push_back(d);
      if (src)
        eblob.insert(eblob.end(), src, src + cnt);
      else
        eblob.insert(eblob.end(), cnt, 0.f);
    };
    for (int k = 0; k < ntoc; ++k) {
      const std::string nm = toc[k].name;
      size_t cnt = 1;
      std::vector<int64_t> shape;
      for (int d = 0; d < toc[k].ndim; ++d) {
        shape.push_back(toc[k].shape[d]);
        cnt *= (size_t)toc[k].shape[d];
      }
      if (toc[k].offset + cnt > n_floats) return pfail(PNVO_ERR_WEIGHTS, "tensor '" + nm + "' exceeds the blob");
      const float *src = blob + toc[k].offset;
      if (nm == pre + "backbone.conv1.0.weight") {          // [C0,1,7,7] -> [C0,2,7,7], second input channel = 0
        if (shape.size() != 4 || shape[1] != 1) return pfail(PNVO_ERR_WEIGHTS, "policy stem must take 1 depth channel");
        const int64_t co = shape[0], kk = shape[2] * shape[3];
        std::vector<float> w((size_t)co * 2 * kk, 0.f);
        for (int64_t o = 0; o < co; ++o) std::memcpy(&w[(size_t)o * 2 * kk], src + o * kk, sizeof(float) * kk);
        push("visual_encoder.backbone.conv1.0.weight", w.data(), {co, 2, shape[2], shape[3]}, w.size());
      } else if (nm.compare(0, pre.size(), pre) == 0) {
        push("visual_encoder." + nm.substr(pre.size()), src, shape, cnt);
      } else if (nm == "net.visual_fc.1.weight") {
        push("visual_fc.2.weight", src, shape, cnt);
      } else if (nm == "net.visual_fc.1.bias") {
        push("visual_fc.2.bias", src, shape, cnt);
      }
    }
    push("output_head.1.weight", nullptr, {1, Hd}, (size_t)Hd);
    push("output_head.1.bias", nullptr, {1}, 1);
    for (size_t k = 0; k < etoc.size(); ++k) etoc[k].name = names[k].c_str();
    rc = pnvo_load_weights(p.enc, eblob.data(), eblob.size(), etoc.data(), (int)etoc.size());
    if (rc != PNVO_OK) return pfail(rc, std::string("policy visual encoder: ") + pnvo_last_error(p.enc));
  }
  // ---- recurrent part and heads (kept in torch's layouts)
  const float *s;
  if (!(s = find(by, blob, n_floats, "net.prev_action_embedding.weight", {c.n_actions + 1, 32}, &rc))) return rc;
  if ((rc = upload(p.emb, s, (size_t)(c.n_actions + 1) * 32)) != PNVO_OK) return rc;
  if (!(s = find(by, blob, n_floats, "net.tgt_embeding.weight", {32, 3}, &rc))) return rc;
  if ((rc = upload(p.tgt_w, s, 96)) != PNVO_OK) return rc;
  if (!(s = find(by, blob, n_floats, "net.tgt_embeding.bias", {32}, &rc))) return rc;
  if ((rc = upload(p.tgt_b, s, 32)) != PNVO_OK) return rc;
  for (int l = 0; l < c.rnn_layers; ++l) {
    const std::string r = "net.state_encoder.rnn.", sl = "_l" + std::to_string(l);
    const int K = l == 0 ? K0 : Hd;
    if (!(s = find(by, blob, n_floats, r + "weight_ih" + sl, {4 * Hd, K}, &rc))) return rc;
    if ((rc = upload(p.w_ih[l], s, (size_t)4 * Hd * K)) != PNVO_OK) return rc;
    if (!(s = find(by, blob, n_floats, r + "weight_hh" + sl, {4 * Hd, Hd}, &rc))) return rc;
    if ((rc = upload(p.w_hh[l], s, (size_t)4 * Hd * Hd)) != PNVO_OK) return rc;
    if (!(s = find(by, blob, n_floats, r + "bias_ih" + sl, {4 * Hd}, &rc))) return rc;
    if ((rc = upload(p.b_ih[l], s, (size_t)4 * Hd)) != PNVO_OK) return rc;
    if (!(s = find(by, blob, n_floats, r + "bias_hh" + sl, {4 * Hd}, &rc))) return rc;
    if ((rc = upload(p.b_hh[l], s, (size_t)4 * Hd)) != PNVO_OK) return rc;
  }
  if (!(s = find(by, blob, n_floats, "action_distribution.linear.weight", {c.n_actions, Hd}, &rc))) return rc;
  if ((rc = upload(p.act_w, s, (size_t)c.n_actions * Hd)) != PNVO_OK) return rc;
  if (!(s = find(by, blob, n_floats, "action_distribution.linear.bias", {c.n_actions}, &rc))) return rc;
  if ((rc = upload(p.act_b, s, (size_t)c.n_actions)) != PNVO_OK) return rc;
  if (!(s = find(by, blob, n_floats, "critic.fc.weight", {1, Hd}, &rc))) return rc;
  if ((rc = upload(p.cr_w, s, (size_t)Hd)) != PNVO_OK) return rc;
  if (!(s = find(by, blob, n_floats, "critic.fc.bias", {1}, &rc))) return rc;
  if ((rc = upload(p.cr_b, s, 1)) != PNVO_OK) return rc;
  p.loaded = true;
  return PNVO_OK;
}

int pnvo_policy_act(pnvo_policy_handle h, const float *depth, const float *goal, const int64_t *prev_actions,
                    const float *masks, const float *hidden_in, int B, float *hidden_out, float *features, float *logits,
                    float *value, void *stream) {
  if (!h) return pfail(PNVO_ERR_ARG, "null handle");
  Policy &p = h->p;
  if (!p.loaded) return pfail(PNVO_ERR_STATE, "pnvo_policy_act before pnvo_policy_load_weights");
  if (B <= 0 || !depth || !goal || !prev_actions || !masks || !hidden_in || !hidden_out)
    return pfail(PNVO_ERR_ARG, "null argument / bad batch");
  PCHK(hipSetDevice(p.device));
  hipStream_t s = (hipStream_t)stream;
  const pnvo_policy_config &c = p.cfg;
  const int Hd = c.hidden, L = c.rnn_layers, K0 = Hd + 64;
  if (B > p.cap) {
    dfree(p.pooled);
    dfree(p.visual);
    dfree(p.x);
    PCHK(hipMalloc((void **)&p.pooled, (size_t)B * (c.height / 2) * (c.width / 2) * 2 * sizeof(float)));
    PCHK(hipMalloc((void **)&p.visual, (size_t)B * Hd * sizeof(float)));
    PCHK(hipMalloc((void **)&p.x, (size_t)B * K0 * sizeof(float)));
    p.cap = B;
  }
  int rc = pnvo_avgpool2(depth, B, c.height, c.width, p.pooled, stream);
  if (rc != PNVO_OK) return rc;
  rc = pnvo_forward_features(p.enc, nullptr, p.pooled, nullptr, nullptr, nullptr, B, p.visual, stream);
  if (rc != PNVO_OK) return pfail(rc, std::string("policy visual encoder: ") + pnvo_last_error(p.enc));
  hipLaunchKernelGGL(policy_inputs_kernel, dim3((unsigned)(((long)B * K0 + 255) / 256)), dim3(256), 0, s, p.visual, goal,
                     prev_actions, masks, p.tgt_w, p.tgt_b, p.emb, c.n_actions + 1, B, Hd, p.x);
  // hidden_in / hidden_out: [2L, B, Hd] = (h_0 .. h_{L-1}, c_0 .. c_{L-1})  (rnn_state_encoder.py:47-61)
  const float *xin = p.x;
  int K = K0;
  for (int l = 0; l < L; ++l) {
    const float *h_prev = hidden_in + (size_t)l * B * Hd, *c_prev = hidden_in + (size_t)(L + l) * B * Hd;
    float *h_new = hidden_out + (size_t)l * B * Hd, *c_new = hidden_out + (size_t)(L + l) * B * Hd;
    hipLaunchKernelGGL(lstm_layer_kernel, dim3((unsigned)Hd), dim3(256), 0, s, xin, K, p.w_ih[l], p.b_ih[l], h_prev, p.w_hh[l], p.b_hh[l],
                       c_prev, masks, B, Hd, h_new, c_new);
    xin = h_new;
    K = Hd;
  }
  const float *feat = hidden_out + (size_t)(L - 1) * B * Hd;
  if (features) PCHK(hipMemcpyAsync(features, feat, (size_t)B * Hd * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (logits || value)
    hipLaunchKernelGGL(policy_heads_kernel, dim3((unsigned)((c.n_actions + 1 + 3) / 4)), dim3(256), 0, s, feat, p.act_w, p.act_b, p.cr_w,
                       p.cr_b, B, Hd, c.n_actions, logits, value);
  PCHK(hipGetLastError());
  return PNVO_OK;
}

int pnvo_policy_destroy(pnvo_policy_handle h) {
  if (!h) return PNVO_OK;
  Policy &p = h->p;
  (void)hipSetDevice(p.device);
  if (p.enc) pnvo_destroy(p.enc);
  dfree(p.emb);
  dfree(p.tgt_w);
  dfree(p.tgt_b);
  for (auto &v : p.w_ih) dfree(v);
  for (auto &v : p.w_hh) dfree(v);
  for (auto &v : p.b_ih) dfree(v);
  for (auto &v : p.b_hh) dfree(v);
  dfree(p.act_w);
  dfree(p.act_b);
  dfree(p.cr_w);
  dfree(p.cr_b);
  dfree(p.pooled);
  dfree(p.visual);
  dfree(p.x);
  delete h;
  return PNVO_OK;
}

}  // extern "C"
