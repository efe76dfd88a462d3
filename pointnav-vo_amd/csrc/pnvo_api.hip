// pnvo_api.hip — the C ABI of include/pnvo.h: model construction, state_dict import, forward orchestration.
// Host-side only; every kernel lives in conv_mfma.hip / elementwise.hip.  gfx950 only, no fallbacks.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pnvo_model.h"

using namespace pnvo;

thread_local std::string g_err;

int pnvo_fail(pnvo_handle h, int code, const std::string &msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}

// Sizes of the operand buffers upload() owns: a reload of same-sized data (every pnvo_load_weights after the first)
// rewrites them in place, so device addresses cached elsewhere (the training step's re-pack maps, captured graphs)
// stay valid across train -> eval -> train switches.
static std::map<const void *, size_t> g_upload_floats;
static std::mutex g_upload_mutex;      // handles of different host threads (one per device) load and destroy concurrently

void pnvo_free_dev(float *&p) {
  if (p) {
    {
      std::lock_guard<std::mutex> lk(g_upload_mutex);
      g_upload_floats.erase(p);
    }
    (void)hipFree(p);
  }
  p = nullptr;
}

namespace {

inline int fail(pnvo_handle h, int code, const std::string &msg) { return pnvo_fail(h, code, msg); }
inline void free_dev(float *&p) { pnvo_free_dev(p); }


Layer make_layer(const std::string &name, const std::string &gn, int cin, int cout, int k, int stride, int pad,
                 int hin, int win, int groups) {
  Layer l;
  l.name = name;
  l.gn = gn;
  l.cin = cin;
  l.cinp = rup(cin, 8);
  l.cout = cout;
  l.coutp = rup(cout, 32);
  l.k = l.kw = k;
  l.stride = stride;
  l.pad = pad;
  l.hin = hin;
  l.win = win;
  l.hout = (hin + 2 * pad - k) / stride + 1;
  l.wout = (win + 2 * pad - k) / stride + 1;
  l.groups = groups;
  return l;
}

// Mirrors ResNet.__init__/_make_layer (resnet.py:153-212) and ResNetEncoder.__init__ (vo_cnn.py:70-101).
void build_plan(pnvo_model_s *m) {
  const pnvo_config &c = m->cfg;
  m->C = c.n_rgb + c.n_depth + c.n_dd + c.n_tdv;
  m->CP = rup(m->C, 8);
  m->CPL = rup(m->C, 16);
  m->Hs = halve(c.height);
  m->Ws = halve(c.width);
  m->Hp = halve(m->Hs);
  m->Wp = halve(m->Ws);
  {
    // reference order: [prev_rgb, prev_d, prev_dd, prev_tdv, cur_rgb, cur_d, cur_dd, cur_tdv]; tensor t holds
    // [prev half | cur half] on its channel axis (vo_cnn.py:114-174)
    const int n[4] = {c.n_rgb, c.n_depth, c.n_dd, c.n_tdv};
    int prev_off[4], half = 0;
    for (int t = 0; t < 4; ++t) {
      prev_off[t] = half;
      half += n[t] / 2;
    }
    m->stem_ref_of_new.clear();
    m->stem_tensor_of_new.clear();
    m->stem_ch_of_new.clear();
    for (int t = 0; t < 4; ++t)
      for (int ch = 0; ch < n[t]; ++ch) {
        const int hn = n[t] / 2;
        m->stem_ref_of_new.push_back(ch < hn ? prev_off[t] + ch : half + prev_off[t] + (ch - hn));
        m->stem_tensor_of_new.push_back(t);
        m->stem_ch_of_new.push_back(ch);
      }
    while ((int)m->stem_ref_of_new.size() < m->CP) {
      m->stem_ref_of_new.push_back(-1);
      m->stem_tensor_of_new.push_back(-1);
      m->stem_ch_of_new.push_back(0);
    }
  }
  const int g = c.baseplanes / 2;
  const std::string bb = "visual_encoder.backbone.";
  m->convs.clear();
  m->convs.push_back(make_layer(bb + "conv1.0", bb + "conv1.1", m->C, c.baseplanes, 7, 2, 3, c.height, c.width, g));
  int h = m->Hp, w = m->Wp, cin = c.baseplanes;
  const bool bott = c.backbone_depth == 50 || c.backbone_depth == 101;
  const int nblk[4] = {bott ? 3 : 2, bott ? 4 : 2, bott ? (c.backbone_depth == 101 ? 23 : 6) : 2, bott ? 3 : 2};
  m->bottleneck = bott;
  m->nblocks.assign(nblk, nblk + 4);
  for (int li = 1; li <= 4; ++li) {
    const int planes = c.baseplanes << (li - 1);
    for (int bi = 0; bi < nblk[li - 1]; ++bi) {
      const std::string p = bb + "layer" + std::to_string(li) + "." + std::to_string(bi) + ".";
      const int stride = (li > 1 && bi == 0) ? 2 : 1;
      if (bott) {                                    // Bottleneck: 1x1 -> 3x3 (stride) -> 1x1 (x4), resnet.py:58-69
        Layer b1 = make_layer(p + "convs.0", p + "convs.1", cin, planes, 1, 1, 0, h, w, g);
        Layer b2 = make_layer(p + "convs.3", p + "convs.4", planes, planes, 3, stride, 1, h, w, g);
        Layer b3 = make_layer(p + "convs.6", p + "convs.7", planes, planes * 4, 1, 1, 0, b2.hout, b2.wout, g);
        m->convs.push_back(b1);
        m->convs.push_back(b2);
        m->convs.push_back(b3);
        if (stride != 1 || cin != planes * 4)
          m->convs.push_back(make_layer(p + "downsample.0", p + "downsample.1", cin, planes * 4, 1, stride, 0, h, w, g));
        h = b2.hout;
        w = b2.wout;
        cin = planes * 4;
        continue;
      }
      Layer c1 = make_layer(p + "convs.0", p + "convs.1", cin, planes, 3, stride, 1, h, w, g);
      m->convs.push_back(c1);
      m->convs.push_back(make_layer(p + "convs.3", p + "convs.4", planes, planes, 3, 1, 1, c1.hout, c1.wout, g));
      if (stride != 1 || cin != planes)
        m->convs.push_back(make_layer(p + "downsample.0", p + "downsample.1", cin, planes, 1, stride, 0, h, w, g));
      h = c1.hout;
      w = c1.wout;
      cin = planes;
    }
  }
  m->fh = h;
  m->fw = w;
  m->comp_c = (int)std::lround((double)c.flat_size / (double)(h * w));   // vo_cnn.py:82-84 (python round)
  {
    // python round() is banker's rounding; replicate for exact .5 cases
    const double q = (double)c.flat_size / (double)(h * w);
    const double fl = std::floor(q);
    if (q - fl == 0.5) m->comp_c = ((long)fl % 2 == 0) ? (int)fl : (int)fl + 1;
  }
  m->comp_cp = rup(m->comp_c, 32);
  m->convs.push_back(make_layer("visual_encoder.compression.0", "visual_encoder.compression.1", cin, m->comp_c, 3, 1, 1,
                                h, w, 1));
  // Linear(flat -> hidden) as a valid (pad 0) fh x fw "conv" over the channel-padded compression map
  m->fc = make_layer(c.act_embed ? "hidden_generator.1" : "visual_fc.2", "", m->comp_c, c.hidden, 1, 1, 0, h, w, 1);
  m->fc.cinp = m->comp_cp;
  m->fc.k = h;
  m->fc.kw = w;
  m->fc.hout = m->fc.wout = 1;
  m->head = make_layer("output_head.1", "", c.hidden, c.out_dim, 1, 1, 0, 1, 1, 1);
}

struct Toc {
  std::map<std::string, const pnvo_tensor_desc *> by_name;
  const float *blob;
  size_t n;
};

const float *find_tensor(pnvo_handle h, const Toc &t, const std::string &name, std::vector<int64_t> shape, int *rc) {
  auto it = t.by_name.find(name);
  if (it == t.by_name.end()) {
    *rc = fail(h, PNVO_ERR_WEIGHTS, "state_dict is missing tensor '" + name + "'");
    return nullptr;
  }
  const pnvo_tensor_desc *d = it->second;
  size_t cnt = 1;
  bool ok = d->ndim == (int)shape.size();
  for (int k = 0; ok && k < d->ndim; ++k) {
    ok = d->shape[k] == shape[k];
    cnt *= (size_t)d->shape[k];
  }
  if (!ok || d->offset + cnt > t.n) {
    std::string s = "tensor '" + name + "' has the wrong shape (want [";
    for (auto v : shape) s += std::to_string(v) + ",";
    s += "])";
    *rc = fail(h, PNVO_ERR_WEIGHTS, s);
    return nullptr;
  }
  return t.blob + d->offset;
}

int upload(pnvo_handle h, float *&dst, const float *src, size_t n) {
  bool same = false;
  if (dst) {
    std::lock_guard<std::mutex> lk(g_upload_mutex);
    auto it = g_upload_floats.find(dst);
    same = it != g_upload_floats.end() && it->second == n;
  }
  if (!same) {
    free_dev(dst);
    HIPCHK(h, hipMalloc((void **)&dst, n * sizeof(float)));
    std::lock_guard<std::mutex> lk(g_upload_mutex);
    g_upload_floats[dst] = n;
  }
  HIPCHK(h, hipMemcpy(dst, src, n * sizeof(float), hipMemcpyHostToDevice));
  return PNVO_OK;
}

}  // namespace

// pack with an explicit padded input-channel count (the FC reads a channel-padded activation)
void pnvo_pack_conv_weight_cinp(const float *oihw, int cout, int cin, int cinp, int kh, int kw, std::vector<float> &out) {
  const int coutp = rup(cout, 32), J = cinp / 8, T = kh * kw, ntg_n = coutp / 32;
  out.assign((size_t)coutp * cinp * T, 0.f);
  for (int ntg = 0; ntg < ntg_n; ++ntg)
    for (int tap = 0; tap < T; ++tap)
      for (int j = 0; j < J; ++j)
        for (int hh = 0; hh < 2; ++hh)
          for (int n = 0; n < 32; ++n)
            for (int t = 0; t < 4; ++t) {
              const int co = ntg * 32 + n, ci = 8 * j + 4 * hh + t;
              if (co < cout && ci < cin)
                out[((((size_t)ntg * T + tap) * J + j) * 64 + hh * 32 + n) * 4 + t] =
                    oihw[((size_t)co * cin + ci) * T + tap];
            }
}

namespace {
inline void pack_conv_weight_cinp(const float *oihw, int cout, int cin, int cinp, int kh, int kw, std::vector<float> &out) {
  pnvo_pack_conv_weight_cinp(oihw, cout, cin, cinp, kh, kw, out);
}

int load_conv(pnvo_handle h, const Toc &t, Layer &l, bool has_gn) {
  int rc = PNVO_OK;
  const float *w = find_tensor(h, t, l.name + ".weight", {l.cout, l.cin, l.k, l.kw}, &rc);
  if (!w) return rc;
  std::vector<float> pk;
  pack_conv_weight_cinp(w, l.cout, l.cin, l.cinp, l.k, l.kw, pk);
  if ((rc = upload(h, l.wpk, pk.data(), pk.size())) != PNVO_OK) return rc;
  l.host_w.assign(w, w + (size_t)l.cout * l.cin * l.k * l.kw);
  if (has_gn) {
    const float *g = find_tensor(h, t, l.gn + ".weight", {l.cout}, &rc);
    if (!g) return rc;
    const float *b = find_tensor(h, t, l.gn + ".bias", {l.cout}, &rc);
    if (!b) return rc;
    if ((rc = upload(h, l.gamma, g, l.cout)) != PNVO_OK) return rc;
    if ((rc = upload(h, l.beta, b, l.cout)) != PNVO_OK) return rc;
  }
  return PNVO_OK;
}

void free_workspace(pnvo_model_s *m) {
  pnvo_drop_graphs(m);            // captured kernel arguments point into the workspace
  free_dev(m->xin);
  free_dev(m->stem_raw);
  free_dev(m->out_ws);
  free_dev(m->bufY[0]);
  free_dev(m->bufY[1]);
  free_dev(m->rawA);
  free_dev(m->rawB);
  free_dev(m->rawD);
  free_dev(m->rawC);
  free_dev(m->comp_raw);
  free_dev(m->hid);
  free_dev(m->stats);
  free_dev(m->gn_ctr);
  free_dev(m->stats_ds);
  free_dev(m->statsB);
  if (m->keys_stream) (void)hipStreamSynchronize(m->keys_stream);     // a fill of the key buffer may still be in flight
  free_dev(m->pool_keys);
  m->keys_primed = false;
  for (int k = 0; k < 2; ++k) {
    free_dev(m->ssA[k]);
    free_dev(m->ssB[k]);
    free_dev(m->ssD[k]);
    free_dev(m->ssC[k]);
  }
  free_dev(m->tapbuf);
  m->cap = 0;
}

// Would this (GroupNorm-ed, bias-free) conv layer run on the LDS-staged 3x3 kernel?  *slots: its statistics slots.
bool layer_on_lds(pnvo_handle m, const Layer &l, int *slots) {
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.KH = l.k;
  a.KW = l.kw;
  a.stride = l.stride;
  a.pad = l.pad;
  a.up = 1;
  a.CIN = l.cinp;
  a.COUTP = l.coutp;
  a.H = l.hin;
  a.W = l.win;
  a.Ho = l.hout;
  a.Wo = l.wout;
  a.y_cstride = l.coutp;
  if (!conv3_lds_supported(a) || m->opt.conv == 3) return false;
  if (slots) *slots = conv3_lds_slots(a);
  return true;
}

size_t stats_floats(pnvo_handle m, const Layer &l, int B) {
  const long P = (long)l.hout * l.wout, M = (long)B * P;
  int MT, NT;
  choose_tile(M, l.coutp, &MT, &NT);
  int slots = conv_slots((int)P, MT), s2 = 0;
  if (layer_on_lds(m, l, &s2) && s2 > slots) slots = s2;            // conv3_lds: slots = tiles x waves
  if ((l.k == 3 && l.kw == 3 && l.pad == 1 && (l.stride == 1 || l.stride == 2)) || (l.k == 1 && l.stride == 2)) {   // conv_x3: slots = tiles
    ConvX3Args xa;
    std::memset(&xa, 0, sizeof(xa));
    xa.B = B;
    xa.H = l.hin;
    xa.W = l.win;
    xa.CIN = l.cinp;
    xa.Ho = l.hout;
    xa.Wo = l.wout;
    xa.COUTP = l.coutp;
    xa.force = 1;                       // (sized for the forced plan: options may change between forwards)
    int mw, nw;
    size_t ldsb;
    if (conv_x3_plan(xa, l.k, l.stride, &mw, &nw, &ldsb) && xa.slots > slots) slots = xa.slots;
  }
  return (size_t)B * (size_t)slots * l.coutp * 2;
}

int ensure_workspace(pnvo_handle m, int B) {   // (also exported as pnvo_ensure_workspace)
  if (B <= m->cap) return PNVO_OK;
  free_workspace(m);
  const pnvo_config &c = m->cfg;
  const size_t npix = (size_t)B * c.height * c.width;
  size_t act = (size_t)B * m->Hp * m->Wp * c.baseplanes;          // largest residual-stage tensor
  for (size_t k = 1; k < m->convs.size(); ++k) {
    const size_t need = (size_t)B * m->convs[k].hout * m->convs[k].wout * m->convs[k].coutp;
    if (need > act) act = need;
  }
  int maxc = m->comp_cp;
  size_t st = (size_t)B * stem_tiles_x(m->Ws) * stem_tiles_y(m->Hs) * m->convs[0].coutp * 2;   // LDS-staged stem
  {
    const size_t st_mx = (size_t)B * stem_mx_slots(m->Hs, m->Ws) * m->convs[0].coutp * 2;
    if (st_mx > st) st = st_mx;
  }
  for (const Layer &l : m->convs) {
    if (l.coutp > maxc) maxc = l.coutp;
    const size_t s = stats_floats(m, l, B);
    if (s > st) st = s;
  }
  auto alloc = [&](float *&p, size_t n) -> hipError_t { return hipMalloc((void **)&p, n * sizeof(float)); };
  (void)npix;   // the assembled [B,H,W,CP] input is only materialised for the "input" tap (allocated lazily)
  HIPCHK(m, alloc(m->stem_raw, (size_t)B * m->Hs * m->Ws * c.baseplanes));
  HIPCHK(m, alloc(m->bufY[0], act));
  HIPCHK(m, alloc(m->bufY[1], act));
  HIPCHK(m, alloc(m->rawA, act));
  HIPCHK(m, alloc(m->rawB, act));
  HIPCHK(m, alloc(m->rawD, act));
  if (m->bottleneck) HIPCHK(m, alloc(m->rawC, act));
  HIPCHK(m, alloc(m->comp_raw, (size_t)B * m->fh * m->fw * m->comp_cp));
  HIPCHK(m, alloc(m->hid, (size_t)B * c.hidden));
  HIPCHK(m, alloc(m->out_ws, (size_t)B * c.out_dim));
  HIPCHK(m, alloc(m->stats, st));
  m->stats_floats = st;
  HIPCHK(m, alloc(m->stats_ds, st));
  HIPCHK(m, alloc(m->statsB, st));
  HIPCHK(m, alloc(m->pool_keys, (size_t)B * m->Hp * m->Wp * m->convs[0].coutp));
  HIPCHK(m, alloc(m->gn_ctr, (size_t)B * 16));
  HIPCHK(m, hipMemset(m->gn_ctr, 0, (size_t)B * 16 * sizeof(float)));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(m, alloc(m->ssA[k], (size_t)B * maxc));
    HIPCHK(m, alloc(m->ssB[k], (size_t)B * maxc));
    HIPCHK(m, alloc(m->ssD[k], (size_t)B * maxc));
    HIPCHK(m, alloc(m->ssC[k], (size_t)B * m->comp_cp));
    HIPCHK(m, hipMemset(m->ssC[k], 0, (size_t)B * m->comp_cp * sizeof(float)));   // pad channels stay (0, 0)
  }
  m->tapbuf_floats = (size_t)B * m->fh * m->fw * m->comp_cp;
  HIPCHK(m, alloc(m->tapbuf, m->tapbuf_floats));
  {                                     // split-K partials of the linear layers: <= 32 slices of [B, hidden]
    const size_t need = (size_t)32 * B * (size_t)std::max(c.hidden, 32);
    if (need > m->kpart_floats) {
      free_dev(m->kpart);
      m->kpart_floats = 0;
      HIPCHK(m, alloc(m->kpart, need));
      m->kpart_floats = need;
    }
  }
  m->cap = B;
  return PNVO_OK;
}

}  // namespace

PnvoTimed::PnvoTimed(pnvo_model_s *m_, hipStream_t s_, const std::string &name, double flops, double bytes) : m(m_), s(s_) {
  if (!m->timing) return;
  auto it = m->tindex.find(name);
  int idx;
  if (it == m->tindex.end()) {
    pnvo_kernel_time e;
    std::memset(&e, 0, sizeof(e));
    std::snprintf(e.name, sizeof(e.name), "%s", name.c_str());
    idx = (int)m->tentries.size();
    m->tentries.push_back(e);
    m->tindex[name] = idx;
  } else {
    idx = it->second;
  }
  m->tentries[idx].launches += 1;
  m->tentries[idx].flops += flops;
  m->tentries[idx].bytes += bytes;
  TimingRec r;
  auto get = [&]() {
    hipEvent_t e;
    if (!m->evpool.empty()) {
      e = m->evpool.back();
      m->evpool.pop_back();
    } else {
      (void)hipEventCreate(&e);
    }
    return e;
  };
  r.a = get();
  r.b = get();
  r.entry = idx;
  (void)hipEventRecord(r.a, s);
  m->trecs.push_back(r);
  rec = (int)m->trecs.size() - 1;
}

PnvoTimed::~PnvoTimed() {
  if (rec >= 0) (void)hipEventRecord(m->trecs[rec].b, s);
}

namespace {
typedef PnvoTimed Timed;

int maybe_tap(pnvo_handle m, const char *name, const float *src, size_t n, hipStream_t s) {
  if (m->tap_dst == nullptr || m->tap_name != name) return PNVO_OK;
  if (n > m->tap_cap) return fail(m, PNVO_ERR_ARG, std::string("tap buffer too small for '") + name + "'");
  HIPCHK(m, hipMemcpyAsync(m->tap_dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return PNVO_OK;
}

}  // namespace

void pnvo_drop_graphs(pnvo_handle m) {
  for (auto &g : m->graphs) {
    (void)hipGraphExecDestroy(g.exec);
    (void)hipGraphDestroy(g.graph);
  }
  m->graphs.clear();
  m->seen.clear();
}

// Upper bound of |activation| entering each conv of the residual stages from per-layer GroupNorm bounds gn_bound(l) =
// max_c (|gamma_c| sqrt(N) + |beta_c|): a block output adds its skip branch (resnet.py:47-55).  Used at load (host copy of the
// parameters) and by the training step (bounds tracked on the device after every optimiser step, pnvo_train_api.hip).
void pnvo_chain_in_bounds(pnvo_handle h, const std::function<float(const Layer &)> &gn_bound) {
  if (h->bottleneck || h->convs.empty()) return;
  float bin = gn_bound(h->convs[0]);                         // pooled stem output
  size_t li = 1;
  for (int stage = 1; stage <= 4; ++stage)
    for (int bi = 0; bi < h->nblocks[stage - 1]; ++bi) {
      if (li + 1 >= h->convs.size()) return;
      Layer &c1 = h->convs[li++];
      Layer &c2 = h->convs[li++];
      const bool ds = li < h->convs.size() && h->convs[li].name.find("downsample") != std::string::npos;
      c1.in_bound = bin;
      c2.in_bound = gn_bound(c1);
      float skip = bin;
      if (ds) {
        Layer &cd = h->convs[li++];
        cd.in_bound = bin;
        skip = gn_bound(cd);
      }
      bin = gn_bound(c2) + skip;
    }
  if (li < h->convs.size()) h->convs[li].in_bound = bin;      // the compression conv reads the last block's output
}

namespace {
// 3x3 (stride 1 or 2, pad 1) GroupNorm-ed convs run on conv_x3.hip unless option `conv` selects another kernel family
bool x3_layer(pnvo_handle m, const Layer &l) {
  const bool s2 = l.stride == 2 && m->opt.x3_s2;
  const bool k3 = l.k == 3 && l.kw == 3 && l.pad == 1 && (l.stride == 1 || s2);
  const bool k1 = l.k == 1 && l.kw == 1 && l.pad == 0 && s2;      // the 1x1 stride-2 downsample convs (resnet.py:192-195)
  return (k3 || k1) && !l.host_w.empty() && m->opt.conv <= 1;
}
bool x3_two_pieces(pnvo_handle m, const Layer &l) {
  if (m->bottleneck || !(l.in_bound < 6.0e4f)) return false;
  if (m->train != nullptr)             // a training step is attached: operands are rebuilt on the device from the flat parameters
    return m->opt.train_pieces == 2 && pnvo_train_x2_scale(m, l.name + ".weight") != nullptr;
  return m->opt.pieces == 2;
}
bool x3_args(pnvo_handle m, const Layer &l, int B, ConvX3Args &xa, int *mw, int *nw, size_t *ldsb) {
  std::memset(&xa, 0, sizeof(xa));
  xa.force = m->opt.conv == 1;
    xa.strip = m->opt.x3_strip;
    xa.fine = m->opt.x3_fine;
    xa.w8_ok = m->opt.x3_w8;
    xa.ksw_ok = m->opt.x3_ksplit;
  xa.persist_wgs = m->opt.x3_persist ? 3 * m->num_cus : 0;
  xa.np = x3_two_pieces(m, l) ? 2 : 3;
  xa.B = B;
  xa.H = l.hin;
  xa.W = l.win;
  xa.CIN = l.cinp;
  xa.Ho = l.hout;
  xa.Wo = l.wout;
  xa.COUTP = l.coutp;
  return conv_x3_plan(xa, l.k, l.stride, mw, nw, ldsb);
}
}  // namespace

bool pnvo_conv_on_x3(pnvo_handle m, const Layer &l, int B) {
  if (!x3_layer(m, l) || l.groups <= 0) return false;
  ConvX3Args xa;
  int mw, nw;
  size_t ldsb;
  return x3_args(m, l, B, xa, &mw, &nw, &ldsb);
}

// The block's downsample conv rides on its first 3x3 conv when both run on the float16-piece form of conv_x3_kernel and share the
// output geometry (they always do: resnet.py:189-212), at inference (a training forward keeps the block input for its backward pass
// and its own launch schedule), and not while a tap wants the plain schedule.
bool pnvo_conv_takes_ds(pnvo_handle m, const Layer &c1, const Layer &cd, int B) {
  if (!m->opt.ds_fuse || m->tap_dst != nullptr || m->in_train_forward || m->bottleneck) return false;
  if (c1.k != 3 || c1.kw != 3 || c1.stride != 2 || c1.pad != 1 || cd.k != 1 || cd.kw != 1 || cd.stride != 2 || cd.pad != 0) return false;
  if (c1.cinp != cd.cinp || c1.coutp != cd.coutp || c1.cout != cd.cout || c1.hout != cd.hout || c1.wout != cd.wout || c1.hin != cd.hin ||
      c1.win != cd.win || c1.cout != c1.coutp || c1.groups != cd.groups || c1.groups <= 0 || cd.host_w.empty())
    return false;
  if (!x3_layer(m, c1) || !x3_layer(m, cd) || !x3_two_pieces(m, c1) || !x3_two_pieces(m, cd)) return false;
  return pnvo_conv_on_x3(m, c1, B);
}

bool pnvo_conv_takes_tail(pnvo_handle m, const Layer &l, int B) {
  if (m->tap_dst != nullptr || !m->opt.tail) return false;   // taps want the block outputs of the plain schedule; option tail=separate: the pass of its own
  return pnvo_conv_on_x3(m, l, B);
}

// One conv + (optionally) the GroupNorm statistics finalisation that follows it.
int pnvo_run_conv(pnvo_handle m, const Layer &l, int B, const float *x, const float *in_scale, const float *in_shift,
                  float *y, int y_cstride, float *ss[2], const float *bias, const int64_t *bias_row, int relu_out,
                  hipStream_t s, const float *const *src, float *mu_out, float *rstd_out, const BlockTail *tail, const DsRide *ride) {
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  if (src != nullptr) {          // fused stem: gather A from the observation tensors
    const int nsrc[4] = {m->cfg.n_rgb, m->cfg.n_depth, m->cfg.n_dd, m->cfg.n_tdv};
    a.src_mode = 1;
    a.zero_page = m->zero_page;
    for (int j = 0; j < m->CP / 8; ++j)
      for (int hh = 0; hh < 2; ++hh)
        for (int q = 0; q < 2; ++q) {
          const int nc = 8 * j + 4 * hh + 2 * q;
          const int tn = m->stem_tensor_of_new[nc];
          SrcPiece &pc = a.pieces[j][hh][q];
          pc.base = tn >= 0 ? src[tn] : nullptr;
          pc.nch = tn >= 0 ? nsrc[tn] : 0;
          pc.choff = m->stem_ch_of_new[nc];
        }
  }
  a.x = x;
  a.wpk = l.wpk;
  a.y = y;
  a.in_scale = in_scale;
  a.in_shift = in_shift;
  a.stats = ss ? m->stats : nullptr;
  a.bias = bias;
  a.bias_row = bias_row;
  a.B = B;
  a.H = l.hin;
  a.W = l.win;
  a.CIN = l.cinp;
  a.Ho = l.hout;
  a.Wo = l.wout;
  a.COUT = l.cout;
  a.COUTP = l.coutp;
  a.KH = l.k;
  a.KW = l.kw;
  a.stride = l.stride;
  a.pad = l.pad;
  a.y_cstride = y_cstride;
  a.relu_out = relu_out;
  a.up = 1;
  const long P = (long)l.hout * l.wout, M = (long)B * P;
  choose_tile(M, l.coutp, &a.MT, &a.NT);
  a.slots = conv_slots((int)P, a.MT);
  const double macs = (double)M * l.cout * l.cin * l.k * l.kw;
  const double bytes = 4.0 * ((double)B * l.hin * l.win * l.cin + (double)M * l.cout + (double)l.cout * l.cin * l.k * l.kw);
  // 3x3 stride-1 convs with GroupNorm: float32 results from the bf16 matrix cores (three-piece operands, conv_x3.hip);
  // option conv=fp32 keeps the fp32-MFMA kernels.  Also in the training forward (the three-piece operand is rebuilt on the
  // device after every optimiser step); not with a fused stem source, bias or output ReLU.
  if (ss && src == nullptr && bias == nullptr && !relu_out && y_cstride == l.coutp && x3_layer(m, l)) {
    ConvX3Args xa;
    std::memset(&xa, 0, sizeof(xa));
    xa.force = m->opt.conv == 1;
    xa.strip = m->opt.x3_strip;
    xa.fine = m->opt.x3_fine;
    xa.w8_ok = m->opt.x3_w8;
    xa.ksw_ok = m->opt.x3_ksplit;
    xa.persist_wgs = m->opt.x3_persist ? 3 * m->num_cus : 0;
    xa.B = B;
    xa.H = l.hin;
    xa.W = l.win;
    xa.CIN = l.cinp;
    xa.Ho = l.hout;
    xa.Wo = l.wout;
    xa.COUTP = l.coutp;
    int mw = 0, nw = 0;
    size_t ldsb = 0;
    // operand pieces: two float16 pieces (three product terms) at inference when the layer's input is provably inside float16's
    // range; three bf16 pieces (six exact terms) otherwise, on request (option pieces=3) and whenever a training step is attached
    // (its device-side re-pack builds the three-piece operand from the flat parameters)
    const bool two = x3_two_pieces(m, l);
    xa.np = two ? 2 : 3;
    const int x3_mode = tail ? (tail->res ? 2 : 3) : (in_scale ? 1 : 0);
    // 32 -> 32 channels (layer1): the row-streaming kernel (conv_rows.hip) where it takes the launch — fewer statistics slots than
    // the tile plan the buffer is sized for (stats_floats)
    // (the plan refuses a block tail whose skip branch carries its own GroupNorm — a downsample skip, reachable with baseplanes 16:
    //  the rows kernel adds the raw skip tensor — so the tail's fields are in place BEFORE the plan looks at them)
    if (tail != nullptr) {
      xa.res = tail->res;
      xa.res_scale = tail->res_scale;
      xa.res_shift = tail->res_shift;
    }
    const bool rows = two && m->opt.x3_rows && conv_rows32_plan(xa, l.k, l.stride, x3_mode, m->num_cus);
    if (rows || conv_x3_plan(xa, l.k, l.stride, &mw, &nw, &ldsb)) {     // (the statistics buffer is sized for it: stats_floats)
      Layer &lm = const_cast<Layer &>(l);
      auto ensure_x2 = [&](Layer &q, pnvo_handle hq = nullptr) -> int {   // (re)build the two-piece float16 operand of a layer (of handle hq)
        if (hq == nullptr) hq = m;
        if (q.wpk_x2 && q.x2_gen == hq->weights_gen) return PNVO_OK;
        const size_t nel = (size_t)q.k * q.kw * q.cinp * q.coutp * 2;
        if (!q.wpk_x2) HIPCHK(m, hipMalloc((void **)&q.wpk_x2, nel * 2));
        const float *dev_w = hq->train ? pnvo_train_weight_ptr(hq, q.name + ".weight") : nullptr;
        if (dev_w != nullptr) {          // training attached: weight and scale live on the device (pnvo_train_refresh)
          HIPCHK(m, launch_conv_x2_repack(dev_w, q.cout, q.cin, q.cinp, q.coutp, q.k, q.kw, pnvo_train_x2_scale(hq, q.name + ".weight"),
                                          q.wpk_x2, s));
        } else {
          std::vector<unsigned short> pk(nel);
          q.x2_oscale = pack_conv_x2_weight(q.host_w.data(), q.cout, q.cin, q.cinp, q.coutp, q.k, q.kw, pk.data());
          HIPCHK(m, hipMemcpyAsync(q.wpk_x2, pk.data(), nel * 2, hipMemcpyHostToDevice, s));
          HIPCHK(m, hipStreamSynchronize(s));
        }
        q.x2_gen = hq->weights_gen;
        return PNVO_OK;
      };
      if (two) {
        if (int rc2 = ensure_x2(lm)) return rc2;
      }
      if (ride != nullptr) {             // the block's downsample conv on this launch (pnvo_conv_takes_ds said yes)
        if (rows || !two || l.stride != 2 || l.k != 3) return fail(m, PNVO_ERR_STATE, "downsample ride on a conv that cannot carry it (" + l.name + ")");
        Layer &dm = const_cast<Layer &>(*ride->cd);
        if (int rc2 = ensure_x2(dm)) return rc2;
        xa.ds_wpk = dm.wpk_x2;
        xa.ds_oscale = dm.x2_oscale;
        if (m->train != nullptr) {
          const float *sp = pnvo_train_x2_scale(m, dm.name + ".weight");
          xa.ds_oscale_ptr = sp ? sp + 1 : nullptr;
        }
        xa.ds_y = ride->y;
        xa.ds_stats = m->stats_ds;
      }
      if (!two && (!lm.wpk_x3 || lm.x3_gen != m->weights_gen)) {   // (re)build the three-piece operand of this layer
        const size_t nel = (size_t)l.k * l.kw * l.cinp * l.coutp * 3;
        if (!lm.wpk_x3) HIPCHK(m, hipMalloc((void **)&lm.wpk_x3, nel * 2));
        const float *dev_w = m->train ? pnvo_train_weight_ptr(m, l.name + ".weight") : nullptr;
        if (dev_w != nullptr) {          // training attached: the current weight lives in the flat parameter buffer
          HIPCHK(m, launch_conv_x3_repack(dev_w, l.cout, l.cin, l.cinp, l.coutp, l.k, l.kw, 0, lm.wpk_x3, s));
        } else {
          std::vector<unsigned short> pk(nel);
          pack_conv_x3_weight(l.host_w.data(), l.cout, l.cin, l.cinp, l.coutp, l.k, l.kw, pk.data());
          HIPCHK(m, hipMemcpyAsync(lm.wpk_x3, pk.data(), nel * 2, hipMemcpyHostToDevice, s));
          HIPCHK(m, hipStreamSynchronize(s));
        }
        lm.x3_gen = m->weights_gen;
      }
      GnGroup gg{0x7fffffff, 0x7fffffff, {nullptr, nullptr}, {nullptr, nullptr}}, ggd = gg;   // grouped forward: models 1 / 2 of this layer
      if (m->grp_n > 1) {
        if (!two || rows) return fail(m, PNVO_ERR_STATE, "grouped forward: layer " + l.name + " is not on the float16-piece tile kernel");
        const size_t idx = (size_t)(&l - m->convs.data());
        if (idx >= m->convs.size()) return fail(m, PNVO_ERR_STATE, "grouped forward: layer outside the conv list");
        xa.grp_end0 = m->grp_end[0];
        xa.grp_end1 = m->grp_n > 2 ? m->grp_end[1] : 0;
        gg.end0 = ggd.end0 = m->grp_end[0];
        if (m->grp_n > 2) gg.end1 = ggd.end1 = m->grp_end[1];
        for (int k = 1; k < m->grp_n; ++k) {
          pnvo_handle hk = m->grp[k];
          Layer &lk = hk->convs[idx];
          if (!x3_two_pieces(hk, lk)) return fail(m, PNVO_ERR_STATE, "grouped forward: a model's " + l.name + " left the float16-piece form");
          if (int rc2 = ensure_x2(lk, hk)) return rc2;
          xa.wpk_g[k - 1] = lk.wpk_x2;
          xa.oscale_g[k - 1] = lk.x2_oscale;
          xa.gn_gamma_g[k - 1] = gg.gamma[k - 1] = lk.gamma;
          xa.gn_beta_g[k - 1] = gg.beta[k - 1] = lk.beta;
          if (ride != nullptr) {
            const size_t idd = (size_t)(ride->cd - m->convs.data());
            Layer &dk = hk->convs[idd];
            if (!x3_two_pieces(hk, dk)) return fail(m, PNVO_ERR_STATE, "grouped forward: a model's downsample conv left the float16-piece form");
            if (int rc2 = ensure_x2(dk, hk)) return rc2;
            xa.ds_wpk_g[k - 1] = dk.wpk_x2;
            xa.ds_oscale_g[k - 1] = dk.x2_oscale;
            xa.ds_gamma_g[k - 1] = ggd.gamma[k - 1] = dk.gamma;
            xa.ds_beta_g[k - 1] = ggd.beta[k - 1] = dk.beta;
          }
        }
      }
      xa.x = x;
      xa.wpk = two ? lm.wpk_x2 : lm.wpk_x3;
      xa.oscale = lm.x2_oscale;
      if (two && m->train != nullptr) {
        const float *sp = pnvo_train_x2_scale(m, l.name + ".weight");
        xa.oscale_ptr = sp ? sp + 1 : nullptr;
      }
      // partial sums: the convs that write ssB keep a buffer of their own (a deferred finalisation reads the producer's sums while
      // the consumer writes its own)
      float *stats_buf = (ss[0] == m->ssB[0] && m->statsB != nullptr) ? m->statsB : m->stats;
      auto pend_key = [&](const float *sc) { return sc == nullptr ? -1 : sc == m->ssA[0] ? 0 : sc == m->ssB[0] ? 1 : sc == m->ssD[0] ? 2 : -1; };
      auto take_pend = [&](const float *sc, ConvX3Args::Fin &f) -> int {      // a pending finalisation behind `sc`: this launch does it
        const int k = pend_key(sc);
        if (k < 0 || !m->gn_pend[k].valid) return PNVO_OK;
        if (rows || !(x3_mode == 1 || x3_mode == 2))
          return fail(m, PNVO_ERR_STATE, "a deferred GroupNorm finalisation reached a launch that cannot do it (" + l.name + ")");
        const auto &pd = m->gn_pend[k];
        const Layer &pl = m->convs[pd.layer];
        f.stats = pd.stats;
        f.slots = pd.slots;
        f.cpg = pd.cpg;
        f.gamma = pl.gamma;
        f.beta = pl.beta;
        for (int g = 1; g < m->grp_n; ++g) {
          f.gamma_g[g - 1] = m->grp[g]->convs[pd.layer].gamma;
          f.beta_g[g - 1] = m->grp[g]->convs[pd.layer].beta;
        }
        m->gn_pend[k].valid = false;
        return PNVO_OK;
      };
      if (int rcp = take_pend(in_scale, xa.fin_in)) return rcp;
      if (tail != nullptr)
        if (int rcp = take_pend(tail->res_scale, xa.fin_res)) return rcp;
      if (xa.fin_in.stats != nullptr || xa.fin_res.stats != nullptr) ldsb += (size_t)xa.CIN * 16;   // the scale / shift tables built in the prologue
      xa.y = y;
      xa.in_scale = in_scale;
      xa.in_shift = in_shift;
      xa.stats = stats_buf;
      if (tail != nullptr) {
        if (in_scale == nullptr) return fail(m, PNVO_ERR_STATE, "block tail without the conv's GroupNorm scale/shift");
        xa.res = tail->res;
        xa.res_scale = tail->res_scale;
        xa.res_shift = tail->res_shift;
        xa.xout = tail->out;
      }
      // one tile per sample (the 12 x 22 and 6 x 11 maps): the workgroup that sums a sample's channels also turns the sums into the
      // GroupNorm scale / shift — the same fp64 arithmetic as gn_finalize_kernel, bit for bit, one launch less (option gn_fuse)
      const int cpg = l.groups > 0 ? l.cout / l.groups : 0;
      const bool fuse = m->opt.gn_fuse && xa.slots == 1 && l.cout == l.coutp && cpg >= 1 && cpg <= 32 &&
                        32 % cpg == 0 && l.cout % cpg == 0 && (rows || !(xa.persist_wgs > 0 && l.cin == 32 && l.coutp == 32));
      // several tiles per sample: the sample's LAST workgroup to arrive finalises (gn_last_arrival; conv_x3_kernel only — the rows
      // kernel with several bands and the persistent form keep the launch).  Not next to a side stream (one counter array).
      const long x3_gy = rows ? 1 : ((l.coutp / 32) + xa.wn * nw - 1) / (xa.wn * nw);
      const bool x3p = !rows && conv_x3_persistent(xa, l.k, l.stride, x3_mode, mw, nw);
      const bool fuse_last = m->opt.gn_fuse == 1 && !rows && !x3p && xa.slots > 1 && l.cout == l.coutp && cpg >= 1 && cpg <= 32 && 32 % cpg == 0 &&
                             l.cout % cpg == 0 && x3_gy <= 16 && (m->side_stream == nullptr || s != m->side_stream) && m->gn_ctr != nullptr;
      if (fuse_last) xa.gn_ctr = reinterpret_cast<unsigned *>(m->gn_ctr);
      if ((fuse || fuse_last) && ride != nullptr) {
        xa.ds_gamma = ride->cd->gamma;
        xa.ds_beta = ride->cd->beta;
        xa.ds_scale = ride->ss[0];
        xa.ds_shift = ride->ss[1];
        xa.ds_mu = ride->mu;
        xa.ds_rstd = ride->rstd;
      }
      if (fuse || fuse_last) {
        xa.gn_gamma = l.gamma;
        xa.gn_beta = l.beta;
        xa.gn_scale = ss[0];
        xa.gn_shift = ss[1];
        xa.gn_mu = mu_out;
        xa.gn_rstd = rstd_out;
        xa.gn_cpg = cpg;
        xa.gn_eps = 1e-5f;
        xa.gn_P = P;
      }
      {
        Timed t(m, s, "conv:" + l.name, 2.0 * macs, bytes + (tail ? 8.0 * B * l.hin * l.win * l.cin : 0.0));
        if (rows)
          HIPCHK(m, launch_conv_rows32(xa, x3_mode, m->num_cus, s));
        else
          HIPCHK(m, launch_conv_x3(xa, l.k, l.stride, x3_mode, mw, nw, ldsb, s));
      }
      if (fuse || fuse_last) return PNVO_OK;
      // deferred: the consumer launch finalises (the forward set defer_main / defer_ride for this call: it knows the consumer)
      const bool can_defer = l.cout == l.coutp && cpg >= 1 && l.cout % cpg == 0 && mu_out == nullptr;
      const bool dm = can_defer && m->defer_main && pend_key(ss[0]) >= 0, dr = can_defer && ride != nullptr && m->defer_ride && ride->mu == nullptr && pend_key(ride->ss[0]) >= 0;
      if (dm) {
        auto &pd = m->gn_pend[pend_key(ss[0])];
        pd.stats = stats_buf;
        pd.slots = xa.slots;
        pd.cpg = cpg;
        pd.layer = (size_t)(&l - m->convs.data());
        pd.valid = true;
      }
      if (dr) {
        auto &pd = m->gn_pend[pend_key(ride->ss[0])];
        pd.stats = m->stats_ds;
        pd.slots = xa.slots;
        pd.cpg = cpg;
        pd.layer = (size_t)(ride->cd - m->convs.data());
        pd.valid = true;
      }
      if (dm && (ride == nullptr || dr)) return PNVO_OK;
      Timed t(m, s, "gn_finalize", 0.0, 0.0);
      if (ride != nullptr && (dm || dr)) {      // one of the two stays a launch
        if (!dm)
          HIPCHK(m, launch_gn_finalize(stats_buf, B, xa.slots, l.coutp, l.cout, l.groups, P, 1, l.gamma, l.beta, 1e-5f, ss[0], ss[1], s, xa.slots,
                                       mu_out, rstd_out, m->grp_n > 1 ? &gg : nullptr));
        if (!dr)
          HIPCHK(m, launch_gn_finalize(m->stats_ds, B, xa.slots, l.coutp, l.cout, l.groups, P, 1, ride->cd->gamma, ride->cd->beta, 1e-5f,
                                       ride->ss[0], ride->ss[1], s, xa.slots, ride->mu, ride->rstd, m->grp_n > 1 ? &ggd : nullptr));
        return PNVO_OK;
      }
      if (ride != nullptr) {             // the conv's GroupNorm and the riding downsample conv's in one launch
        const float *st2[2] = {stats_buf, m->stats_ds}, *ga2[2] = {l.gamma, ride->cd->gamma}, *be2[2] = {l.beta, ride->cd->beta};
        float *sc2[2] = {ss[0], ride->ss[0]}, *sh2[2] = {ss[1], ride->ss[1]};
        float *mu2[2] = {mu_out, ride->mu}, *rs2[2] = {rstd_out, ride->rstd};
        HIPCHK(m, launch_gn_finalize_pair(st2, B, xa.slots, l.coutp, l.cout, l.groups, P, ga2, be2, 1e-5f, sc2, sh2, mu2, rs2, s,
                                          m->grp_n > 1 ? &gg : nullptr, m->grp_n > 1 ? &ggd : nullptr));
        return PNVO_OK;
      }
      HIPCHK(m, launch_gn_finalize(stats_buf, B, xa.slots, l.coutp, l.cout, l.groups, P, 1, l.gamma, l.beta, 1e-5f, ss[0], ss[1], s,
                                   xa.slots, mu_out, rstd_out, m->grp_n > 1 ? &gg : nullptr));
      return PNVO_OK;
    }
  }
  if (tail != nullptr) return fail(m, PNVO_ERR_STATE, "block tail handed to a conv that cannot take it (" + l.name + ")");
  if (m->grp_n > 1 && ss != nullptr) return fail(m, PNVO_ERR_STATE, "grouped forward: layer " + l.name + " fell off the float16-piece tile kernel");
  const bool lds3 = conv3_lds_supported(a) && m->opt.conv != 3;
  if (lds3) {                    // 3x3 stride-1 residual-stage conv: input patch staged in LDS
    int nt = (l.coutp / 32) % 2 == 0 ? 2 : 1;
    if (m->opt.conv3_nt == 1) nt = 1;
    a.slots = conv3_lds_slots(a);
    {
      Timed t(m, s, "conv:" + l.name, 2.0 * macs, bytes);
      HIPCHK(m, launch_conv3_lds(a, nt, s));
    }
    if (ss) {
      Timed t(m, s, "gn_finalize", 0.0, 0.0);
      HIPCHK(m, launch_gn_finalize(m->stats, B, a.slots, l.coutp, l.cout, l.groups, P, 1, l.gamma, l.beta, 1e-5f, ss[0],
                                   ss[1], s, a.slots, mu_out, rstd_out));
    }
    return PNVO_OK;
  }
  if (bias != nullptr && !ss) {  // linear layer: split the reduction when the output tiles alone cannot fill the chip
    a.kpart = reinterpret_cast<float *>(8);          // non-null: "scratch available" for the query
    const int ks = conv_ksplit(a);
    a.kpart = nullptr;
    if (ks > 1) {
      const size_t need = (size_t)ks * M * y_cstride;
      if (need > m->kpart_floats) {
        pnvo_drop_graphs(m);                         // captured launches point into the old scratch
        if (m->kpart) (void)hipFree(m->kpart);
        m->kpart = nullptr;
        m->kpart_floats = 0;
        HIPCHK(m, hipMalloc((void **)&m->kpart, need * sizeof(float)));
        m->kpart_floats = need;
      }
      a.kpart = m->kpart;
      a.ksplit = ks;
    }
  }
  {
    Timed t(m, s, "conv:" + l.name, 2.0 * macs, bytes);
    HIPCHK(m, launch_conv(a, s));
    if (a.ksplit > 1 && m->head_ride_out != nullptr && P == 1 && y_cstride == l.cout) {   // the output head on the reduction launch
      HIPCHK(m, launch_ksplit_reduce_head(a, m->head_ride_w, m->head_bias, m->cfg.out_dim, m->head_ride_out, s));
      m->head_rode = true;
    } else if (a.ksplit > 1) {
      HIPCHK(m, launch_ksplit_reduce(a, s));
    }
  }
  if (ss) {
    Timed t(m, s, "gn_finalize", 0.0, 0.0);
    HIPCHK(m, launch_gn_finalize(m->stats, B, a.slots, l.coutp, l.cout, l.groups, P, a.MT * 32, l.gamma, l.beta, 1e-5f,
                                 ss[0], ss[1], s, 0, mu_out, rstd_out));
  }
  return PNVO_OK;
}

namespace {
inline int run_conv(pnvo_handle m, const Layer &l, int B, const float *x, const float *in_scale, const float *in_shift,
                    float *y, int y_cstride, float *ss[2], const float *bias, const int64_t *bias_row, int relu_out,
                    hipStream_t s, const float *const *src = nullptr) {
  return pnvo_run_conv(m, l, B, x, in_scale, in_shift, y, y_cstride, ss, bias, bias_row, relu_out, s, src, nullptr, nullptr);
}
}  // namespace

// The fused stem: input assembly + /255 + whitening gathered in the operand fetch (LDS-staged kernel when the channel
// count allows, else MODE 2 of the generic kernel), raw output + GroupNorm scale/shift (+ optional mean/rstd).
void pnvo_stem_raw_args(pnvo_handle m, StemMXArgs &a) {
  if (m->raw_depth == nullptr) return;
  a.raw_rgb = m->raw_rgb;
  a.raw_depth = m->raw_depth;
  a.raw_flags = (m->cfg.n_depth > 0 ? 1 : 0) | (m->cfg.n_dd > 0 ? 2 : 0);
  a.raw_err = m->raw_err;
  const int bins = 10;                                             // the mx stem's K-slot layout (n_dd == 20)
  for (int i = 0; i < bins; ++i) a.edges[i] = (float)((double)i * 1.0 / (double)bins);   // base_trainer_with_vo.py:105-115
  a.edges[bins] = 1.0f;
  a.edges[bins + 1] = 1.0f;
  a.src[0] = a.src[1] = a.src[2] = nullptr;
}

// The float32-MFMA stem (stem_lds.hip) serves this model: the kernel the input-contract repair runs.
static bool stem_lds_serves(pnvo_handle m) {
  const Layer &stem = m->convs[0];
  return (m->CPL <= 32) && (stem.coutp == 32 || stem.coutp == 64) && stem.cout == stem.coutp;
}

// Inference forwards decide on the DEVICE whether the stem is redone on float32 operands (pnvo_stem_repair); the training forward
// keeps the host-side event (its backward has to know, too).
static bool stem_repairs_on_device(pnvo_handle m) {
  return m->opt.input_fallback && m->dd_flag != nullptr && !m->in_train_forward && stem_lds_serves(m);
}

// The stem's place in the forward is the 16-bit-matrix-core stems' (8 x 16-tile slots, pooled keys, raw entry).  Once the input
// fallback engaged (dense_sticky) that place is kept and the float32 stem stands in (stem_lds_kernel<.., PAIRED>) wherever it
// serves the model; elsewhere, and in the training forward, the handle leaves the mx path as before.
bool pnvo_stem_on_mx(pnvo_handle m) {
  if (m->dense_sticky && (m->in_train_forward || !stem_lds_serves(m))) return false;
  return m->mx_ok && (!m->in_train_forward || m->train_mx) && m->opt.stem <= 1;
}

// The float32 stem in the place of an 8 x 16-tile stem: raw output + that stem's GroupNorm slot layout (+ pooled keys).  `only_if`
// (device-readable flag) makes both launches no-ops while it is zero.
static int pnvo_stem_standin(pnvo_handle m, int B, const float *const *src, float *y, int slots, int *pool_keys, const int *only_if,
                             hipStream_t s) {
  const pnvo_config &c = m->cfg;
  const Layer &stem = m->convs[0];
  const int nsrc[4] = {c.n_rgb, c.n_depth, c.n_dd, c.n_tdv};
  StemArgs a;
  std::memset(&a, 0, sizeof(a));
  for (int j = 0; j < m->CPL / 8; ++j)
    for (int hh = 0; hh < 2; ++hh)
      for (int q = 0; q < 2; ++q) {
        const int nc = 8 * j + 4 * hh + 2 * q;
        const int tn = nc < m->CP ? m->stem_tensor_of_new[nc] : -1;
        a.pieces[j][hh][q].base = tn >= 0 ? src[tn] : nullptr;
        a.pieces[j][hh][q].nch = tn >= 0 ? nsrc[tn] : 0;
        a.pieces[j][hh][q].choff = tn >= 0 ? m->stem_ch_of_new[nc] : 0;
      }
  a.sc = m->stem_sc;
  a.sh = m->stem_sh;
  a.wpk = m->stem_wpk16;
  a.zero_page = m->zero_page;
  a.y = y;
  a.stats = m->stats;
  a.B = B;
  a.H = c.height;
  a.W = c.width;
  a.Ho = m->Hs;
  a.Wo = m->Ws;
  a.CPL = m->CPL;
  a.slots = slots;
  a.paired = 1;
  a.only_if = only_if;
  a.publish = only_if ? m->dd_flag : nullptr;     // a raised flag reaches the host-mapped copy from this launch
  {
    const double M = (double)B * m->Hs * m->Ws;
    Timed t(m, s, only_if ? "stem_repair" : "conv:" + stem.name, only_if ? 0.0 : 2.0 * M * stem.cout * stem.cin * 49,
            only_if ? 0.0 : 4.0 * ((double)B * c.height * c.width * stem.cin + M * stem.cout + (double)stem.cout * stem.cin * 49));
    HIPCHK(m, launch_stem_lds(a, stem.coutp, s));
  }
  if (pool_keys != nullptr) {
    Timed t(m, s, only_if ? "stem_repair" : "pool_keys", 0.0, only_if ? 0.0 : 4.0 * B * ((double)m->Hs * m->Ws + (double)m->Hp * m->Wp) * stem.coutp);
    HIPCHK(m, launch_pool_keys_from_raw(y, stem.gamma, B, m->Hs, m->Ws, stem.coutp, pool_keys, only_if, s));
  }
  return PNVO_OK;
}

// An event behind a contract-checking stem launch (see pnvo_input_fallback); nothing while a stream capture is under way
// (an event recorded into a graph cannot be waited for: such forwards keep the deferred check of pnvo_check_inputs).
// Does the stem kernel pnvo_run_stem would launch leave per-tile statistics ([B][slots][CP][2]) in m->stats?
bool stem_writes_slots(pnvo_handle m) {
  return pnvo_stem_on_mx(m) || (m->dd_ok && m->opt.stem != 3 && !m->dense_sticky) || stem_lds_serves(m);
}

int pnvo_mark_stem(pnvo_handle m, hipStream_t s) {
  if (!m->dd_flag || m->dense_sticky || m->raw_depth != nullptr) return PNVO_OK;   // sensor frames: inside the contract by construction
  if (stem_repairs_on_device(m)) return PNVO_OK;    // decided on the device: pnvo_stem_standin(.., only_if = the flag) follows the stem
  HIPCHK(m, launch_flag_publish(m->dd_flag_dev, m->dd_flag, s));   // the host-side decisions below / pnvo_check_inputs read the host copy
  if (!m->opt.input_fallback) return PNVO_OK;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return PNVO_OK;
  if (!m->stem_ev) HIPCHK(m, hipEventCreateWithFlags(&m->stem_ev, hipEventDisableTiming));
  HIPCHK(m, hipEventRecord(m->stem_ev, s));
  m->stem_ev_pending = true;
  return PNVO_OK;
}

int pnvo_run_stem(pnvo_handle m, int B, const float *const *src, float *y, float *ss[2], float *mu_out, float *rstd_out,
                  hipStream_t s, int *pool_keys) {
  const pnvo_config &c = m->cfg;
  const Layer &stem = m->convs[0];
  int rc = PNVO_OK;
  const bool lds_stem = (m->CPL <= 32) && (stem.coutp == 32 || stem.coutp == 64) && stem.cout == stem.coutp;
  if (pool_keys != nullptr && !pnvo_stem_on_mx(m)) return fail(m, PNVO_ERR_STATE, "pooled stem output asked of a stem kernel without it");
  if (pnvo_stem_on_mx(m)) {
    // bf16 matrix cores, three exact weight pieces: float32 results (stem_mx.hip).  The training step keeps the kernels
    // below, whose operands it rebuilds on the device after every Adam step.
    StemMXArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int k = 0; k < 4; ++k) a.src[k] = src[k];
    pnvo_stem_raw_args(m, a);
    // two float16 weight pieces at inference (5 MFMAs per tap); three bf16 pieces (7, every product exact) on request and
    // whenever a training step is attached (its device-side re-pack builds the three-piece operand)
    // (an eval forward of a handle with a training step attached takes the float16 operand too when it is current: packed by the
    //  pnvo_load_weights that followed the last optimiser step)
    //  pnvo_load_weights that followed the last optimiser step — or kept current by the training step's device-side re-pack)
    const bool h2_current = m->train == nullptr || m->mx_wpk2_dev || (!m->in_train_forward && m->weights_gen == m->weights_gen_at_load);
    const bool want2 = m->in_train_forward ? m->opt.train_pieces == 2 : m->opt.pieces == 2;
    const int pieces = (want2 && h2_current && m->mx_wpk2 != nullptr) ? 2 : 3;
    a.zero_page = m->mx_pages;
    a.wpk = pieces == 2 ? m->mx_wpk2 : m->mx_wpk3;
    a.oscale = m->mx_oscale;
    // (after a device-side re-pack the scale lives on the device; a later pnvo_load_weights re-packs on the host with its own)
    a.oscale_ptr = m->mx_wpk2_dev ? m->mx_scale2_dev + 1 : nullptr;
    a.bad_input = m->dd_flag_dev;
    const int ntn = stem.cout / 32;
    for (int g = 0; g < ntn; ++g) {
      a.y[g] = y;
      a.stats[g] = m->stats;
      a.y_coff[g] = 32 * g;
    }
    a.y_cstride = stem.coutp;
    a.stats_cstride = stem.coutp;
    a.B = B;
    a.H = c.height;
    a.W = c.width;
    a.Ho = m->Hs;
    a.Wo = m->Ws;
    a.slots = stem_mx_slots(m->Hs, m->Ws);
    a.pool = pool_keys;
    a.pool_gamma = stem.gamma;
    a.Hp = m->Hp;
    a.Wp = m->Wp;
    GnGroup sgg{0x7fffffff, 0x7fffffff, {nullptr, nullptr}, {nullptr, nullptr}};
    if (m->grp_n > 1) {                 // grouped forward: the other models' stem operands (tile kernel only)
      if (pieces != 2 || pool_keys == nullptr) return fail(m, PNVO_ERR_STATE, "grouped forward needs the float16-piece stem with pooled keys");
      a.grp_end0 = sgg.end0 = m->grp_end[0];
      if (m->grp_n > 2) a.grp_end1 = sgg.end1 = m->grp_end[1];
      for (int k = 1; k < m->grp_n; ++k) {
        pnvo_handle hk = m->grp[k];
        if (hk->mx_wpk2 == nullptr || hk->mx_wpk2_dev) return fail(m, PNVO_ERR_STATE, "grouped forward: a model has no host-packed float16 stem operand");
        a.wpk_g[k - 1] = hk->mx_wpk2;
        a.oscale_g[k - 1] = hk->mx_oscale;
        a.pool_gamma_g[k - 1] = sgg.gamma[k - 1] = hk->convs[0].gamma;
        sgg.beta[k - 1] = hk->convs[0].beta;
      }
    }
    a.dbg = m->opt.stem_dbg >= 16 ? m->opt.stem_dbg - 16 : 0;
    if (m->opt.stem_dbg == 9 || m->opt.stem_dbg >= 16) {
      if (!m->mx_prof) {
        HIPCHK(m, hipMalloc((void **)&m->mx_prof, 2048));
        HIPCHK(m, hipMemset(m->mx_prof, 0, 2048));
      }
      a.prof = m->mx_prof;
    }
    const double M = (double)B * m->Hs * m->Ws;
    if (m->dense_sticky) {
      // the input fallback engaged: the float32 stem stands in (same place in the forward, same slot layout, pooled keys)
      if ((rc = pnvo_stem_standin(m, B, src, y, a.slots, pool_keys, nullptr, s)) != PNVO_OK) return rc;
    } else {
      // algorithmic bytes: the observation tensors (or, RAW: 6 B of rgb + 8 B of depth + 8 B of top-down view per pixel) once
      const double in_bytes = (double)B * c.height * c.width * (m->raw_depth ? (c.n_rgb ? 6.0 : 0.0) + 8.0 + (c.n_tdv ? 8.0 : 0.0) : 4.0 * stem.cin);
      Timed t(m, s, "conv:" + stem.name, 2.0 * M * stem.cout * stem.cin * 49,
              in_bytes + 4.0 * (M * stem.cout + (double)stem.cout * stem.cin * 49));
      // Forms of the float16-piece stem (option stem_form: auto | fast | resident | tiles):
      //   fast (auto when every workgroup gets >= 4 tiles): one 4-wave workgroup per CU for the whole launch, the weights in its
      //     registers, staging of the next tile and epilogue of the previous one between the MFMAs (stem_rs.hip), the remainder MFMAs
      //     of four taps in one K chunk and tap 48 split over the waves — 0.78 ms at 256 pairs, float32-grade equal to the others;
      //   resident: the same kernel in the tile kernel's summation order — 0.81 ms, bit-identical to tiles;
      //   tiles (auto otherwise): one tile per workgroup, two workgroups per CU — 0.99 ms, bound by the CU's vector-memory pipe
      //     (245 KB of weight fragments + 93 KB of patch per 128-pixel tile);
      //   (round 4's role-specialised persistent form — stem_ps_kernel, as fast as tiles — was retired in round 5: HISTORY.md.)
      const bool rs = m->grp_n <= 1 && (m->opt.stem_form == 3 || m->opt.stem_form == 4 || m->opt.stem_form == 0) && stem_rs_takes(a, pieces, ntn, false, m->num_cus);
      m->mx_prof_rs = rs;
      if (rs)
        HIPCHK(m, launch_stem_rs(a, pieces, m->opt.stem_form == 4 || m->opt.stem_form == 0, m->num_cus, s));
      else
        HIPCHK(m, launch_stem_mx(a, pieces, ntn, false, s));
    }
    // a value outside the observation contract (the flag the stager raised): redone on float32 operands, decided on the DEVICE —
    // two launches that return at once while the flag is down; the host never waits (it reads the flag at its next entry and
    // moves the handle to the stand-in for good: pnvo_check_inputs)
    if (!m->dense_sticky && m->raw_depth == nullptr && stem_repairs_on_device(m) &&
        (rc = pnvo_stem_standin(m, B, src, y, a.slots, pool_keys, m->dd_flag_dev, s)) != PNVO_OK)
      return rc;
    if ((rc = pnvo_mark_stem(m, s)) != PNVO_OK) return rc;
    m->stem_slots_out = a.slots;
    if (!m->stem_skip_finalize) {
      Timed t(m, s, "gn_finalize", 0.0, 0.0);
      HIPCHK(m, launch_gn_finalize(m->stats, B, a.slots, stem.coutp, stem.cout, stem.groups, (long)m->Hs * m->Ws, 1,
                                   stem.gamma, stem.beta, 1e-5f, ss[0], ss[1], s, a.slots, mu_out, rstd_out, m->grp_n > 1 ? &sgg : nullptr));
    }
  } else if (m->dd_ok && m->opt.stem != 3 && m->dense_sticky && !m->in_train_forward && stem_lds_serves(m)) {
    // the input fallback engaged on the one-hot-aware stem: the float32 stem stands in, in that stem's slot layout
    const int slots = stem_dd_slots(m->Hs, m->Ws);
    if ((rc = pnvo_stem_standin(m, B, src, y, slots, nullptr, nullptr, s)) != PNVO_OK) return rc;
    m->stem_slots_out = slots;
    if (!m->stem_skip_finalize) {
      Timed t(m, s, "gn_finalize", 0.0, 0.0);
      HIPCHK(m, launch_gn_finalize(m->stats, B, slots, stem.coutp, stem.cout, stem.groups, (long)m->Hs * m->Ws, 1,
                                   stem.gamma, stem.beta, 1e-5f, ss[0], ss[1], s, slots, mu_out, rstd_out));
    }
  } else if (m->dd_ok && m->opt.stem != 3 && !m->dense_sticky) {
    // one-hot-aware stem (in training its operands are rebuilt on the device every step: refresh_stem_dd)
    const int nsrc[4] = {c.n_rgb, c.n_depth, c.n_dd, c.n_tdv};
    StemDDArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int j = 0; j < 3; ++j)
      for (int q = 0; q < 2; ++q) {
        const int d = 4 * j + 2 * q;
        const int tn = m->dd_dense_tensor[d];
        a.pieces[j][0][q].base = tn >= 0 ? src[tn] : nullptr;
        a.pieces[j][0][q].nch = tn >= 0 ? nsrc[tn] : 0;
        a.pieces[j][0][q].choff = tn >= 0 ? m->dd_dense_ch[d] : 0;
      }
    a.sc = m->dd_sc;
    a.sh = m->dd_sh;
    a.wpk = m->dd_wpk;
    a.table = m->dd_table;
    a.dd = src[2];
    a.zero_page = m->zero_page;
    a.bad_onehot = m->dd_flag_dev;
    a.y = y;
    a.stats = m->stats;
    a.B = B;
    a.H = c.height;
    a.W = c.width;
    a.Ho = m->Hs;
    a.Wo = m->Ws;
    a.bins = m->dd_bins;
    a.slots = stem_dd_slots(m->Hs, m->Ws);
    a.dbg = m->opt.stem_dbg;
    if (a.dbg == 9) {
      if (!m->dd_prof) {
        HIPCHK(m, hipMalloc((void **)&m->dd_prof, 64));
        HIPCHK(m, hipMemset(m->dd_prof, 0, 64));
      }
      a.prof = m->dd_prof;
    }
    const double M = (double)B * m->Hs * m->Ws;
    {
      Timed t(m, s, "conv:" + stem.name, 2.0 * M * stem.cout * stem.cin * 49,
              4.0 * ((double)B * c.height * c.width * stem.cin + M * stem.cout + (double)stem.cout * stem.cin * 49));
      HIPCHK(m, launch_stem_dd(a, s));
    }
    if (stem_repairs_on_device(m) && (rc = pnvo_stem_standin(m, B, src, y, a.slots, nullptr, m->dd_flag_dev, s)) != PNVO_OK) return rc;
    if ((rc = pnvo_mark_stem(m, s)) != PNVO_OK) return rc;
    m->stem_slots_out = a.slots;
    if (!m->stem_skip_finalize) {
      Timed t(m, s, "gn_finalize", 0.0, 0.0);
      HIPCHK(m, launch_gn_finalize(m->stats, B, a.slots, stem.coutp, stem.cout, stem.groups, (long)m->Hs * m->Ws, 1,
                                   stem.gamma, stem.beta, 1e-5f, ss[0], ss[1], s, a.slots, mu_out, rstd_out));
    }
  } else if (lds_stem) {
    const int nsrc[4] = {c.n_rgb, c.n_depth, c.n_dd, c.n_tdv};
    StemArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int j = 0; j < m->CPL / 8; ++j)
      for (int hh = 0; hh < 2; ++hh)
        for (int q = 0; q < 2; ++q) {
          const int nc = 8 * j + 4 * hh + 2 * q;
          const int tn = nc < m->CP ? m->stem_tensor_of_new[nc] : -1;
          a.pieces[j][hh][q].base = tn >= 0 ? src[tn] : nullptr;
          a.pieces[j][hh][q].nch = tn >= 0 ? nsrc[tn] : 0;
          a.pieces[j][hh][q].choff = tn >= 0 ? m->stem_ch_of_new[nc] : 0;
        }
    a.sc = m->stem_sc;
    a.sh = m->stem_sh;
    a.wpk = m->stem_wpk16;
    a.zero_page = m->zero_page;
    a.y = y;
    a.stats = m->stats;
    a.B = B;
    a.H = c.height;
    a.W = c.width;
    a.Ho = m->Hs;
    a.Wo = m->Ws;
    a.CPL = m->CPL;
    a.slots = stem_tiles_x(m->Ws) * stem_tiles_y(m->Hs);
    a.dbg = m->opt.stem_dbg;
    a.lds_pad = m->opt.stem_dbg_pad;
    const double M = (double)B * m->Hs * m->Ws;
    {
      Timed t(m, s, "conv:" + stem.name, 2.0 * M * stem.cout * stem.cin * 49,
              4.0 * ((double)B * c.height * c.width * stem.cin + M * stem.cout + (double)stem.cout * stem.cin * 49));
      HIPCHK(m, launch_stem_lds(a, stem.coutp, s));
    }
    m->stem_slots_out = a.slots;
    if (!m->stem_skip_finalize) {
      Timed t(m, s, "gn_finalize", 0.0, 0.0);
      HIPCHK(m, launch_gn_finalize(m->stats, B, a.slots, stem.coutp, stem.cout, stem.groups, (long)m->Hs * m->Ws, 1,
                                   stem.gamma, stem.beta, 1e-5f, ss[0], ss[1], s, a.slots, mu_out, rstd_out));
    }
  } else {
    if ((rc = pnvo_run_conv(m, stem, B, nullptr, m->stem_sc, m->stem_sh, y, stem.coutp, ss, nullptr, nullptr, 0, s, src,
                            mu_out, rstd_out)) != PNVO_OK)
      return rc;
  }
  return rc;
}

int pnvo_ensure_workspace(pnvo_handle m, int B) { return ensure_workspace(m, B); }

namespace {
// option table: key -> (member, accepted spellings).  The PNVO_* environment variables of earlier rounds are the DEFAULTS of
// these options, read once per handle in pnvo_create; nothing on the forward path reads the environment.
struct OptChoice {
  const char *word;
  int value;
};
struct OptDef {
  const char *key, *env;
  int PnvoOptions::*field;
  bool numeric;                 // value = atoi(word) instead of a choice
  OptChoice choices[7];
};
const OptDef kOptions[] = {
    {"stem", "PNVO_STEM", &PnvoOptions::stem, false, {{"auto", 0}, {"mx", 1}, {"dd", 2}, {"dense", 3}, {nullptr, 0}}},
    {"conv", "PNVO_CONV", &PnvoOptions::conv, false, {{"auto", 0}, {"x3", 1}, {"fp32", 2}, {"generic", 3}, {nullptr, 0}}},
    {"pieces", "PNVO_PIECES", &PnvoOptions::pieces, false, {{"2", 2}, {"3", 3}, {nullptr, 0}}},
    {"stem_form", "PNVO_STEM_FORM", &PnvoOptions::stem_form, false, {{"auto", 0}, {"tiles", 2}, {"resident", 3}, {"fast", 4}, {nullptr, 0}}},
    {"train_pieces", "PNVO_TRAIN_PIECES", &PnvoOptions::train_pieces, false, {{"2", 2}, {"3", 3}, {nullptr, 0}}},
    {"x3_persist", "PNVO_X3_PERSIST", &PnvoOptions::x3_persist, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"x3_rows", "PNVO_X3_ROWS", &PnvoOptions::x3_rows, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"ds_side", "PNVO_DS_SIDE", &PnvoOptions::ds_side, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"x3_strip", "PNVO_X3_STRIP", &PnvoOptions::x3_strip, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"pool_async", "PNVO_POOL_ASYNC", &PnvoOptions::pool_async, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"gn_defer", "PNVO_GN_DEFER", &PnvoOptions::gn_defer, true, {{nullptr, 0}}},
    {"x3_fine", "PNVO_X3_FINE", &PnvoOptions::x3_fine, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"x3_w8", "PNVO_X3_W8", &PnvoOptions::x3_w8, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"x3_ksplit", "PNVO_X3_KSPLIT", &PnvoOptions::x3_ksplit, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"fc_rows", "PNVO_FC_ROWS", &PnvoOptions::fc_rows, true, {{nullptr, 0}}},
    {"head_fuse", "PNVO_HEAD_FUSE", &PnvoOptions::head_fuse, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"ds_fuse", "PNVO_DS_FUSE", &PnvoOptions::ds_fuse, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"gn_fuse", "PNVO_GN_FUSE", &PnvoOptions::gn_fuse, false, {{"on", 2}, {"single", 2}, {"last", 1}, {"off", 0}, {nullptr, 0}}},
    {"x3_s2", nullptr, &PnvoOptions::x3_s2, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"tail", "PNVO_TAIL", &PnvoOptions::tail, false, {{"fused", 1}, {"separate", 0}, {nullptr, 0}}},
    {"pool", "PNVO_POOL", &PnvoOptions::pool, false, {{"fused", 1}, {"separate", 0}, {nullptr, 0}}},
    {"conv3_nt", "PNVO_CONV3_NT", &PnvoOptions::conv3_nt, true, {{nullptr, 0}}},
    {"graph", "PNVO_GRAPH", &PnvoOptions::graph, true, {{nullptr, 0}}},
    {"stem_dbg", nullptr, &PnvoOptions::stem_dbg, true, {{nullptr, 0}}},
    {"stem_dbg_pad", nullptr, &PnvoOptions::stem_dbg_pad, true, {{nullptr, 0}}},
    {"wgrad_stem", "PNVO_WGRAD_STEM", &PnvoOptions::wgrad_stem, false, {{"mx", 0}, {"fp32", 1}, {nullptr, 0}}},
    {"wgrad3", "PNVO_WGRAD3", &PnvoOptions::wgrad3, false, {{"x3", 1}, {"fp32", 0}, {nullptr, 0}}},
    {"pool_bwd", "PNVO_POOL_BWD", &PnvoOptions::pool_bwd, false, {{"fused", 1}, {"separate", 0}, {nullptr, 0}}},
    {"dgrad", "PNVO_DGRAD", &PnvoOptions::dgrad, false, {{"phase", 1}, {"masked", 0}, {nullptr, 0}}},
    {"bf16_fuse", nullptr, &PnvoOptions::bf16_fuse, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"bf16_stem3", "PNVO_BF16_STEM3", &PnvoOptions::bf16_stem3, true, {{nullptr, 0}}},
    {"input_fallback", "PNVO_INPUT_FALLBACK", &PnvoOptions::input_fallback, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"small_net", "PNVO_SMALL_NET", &PnvoOptions::small_net, false, {{"on", 1}, {"off", 0}, {nullptr, 0}}},
    {"small_max", "PNVO_SMALL_MAX", &PnvoOptions::small_max, true, {{nullptr, 0}}},
    {"small_coop", "PNVO_SMALL_COOP", &PnvoOptions::small_coop, true, {{nullptr, 0}}},
    {"small_prof", "PNVO_SMALL_PROF", &PnvoOptions::small_prof, true, {{nullptr, 0}}},
};

const OptDef *find_option(const char *key) {
  for (const OptDef &d : kOptions)
    if (std::strcmp(d.key, key) == 0) return &d;
  return nullptr;
}

bool parse_option(const OptDef &d, const char *word, int *out) {
  if (d.numeric) {
    char *end = nullptr;
    const long v = std::strtol(word, &end, 10);
    if (end == word || *end != 0) return false;
    *out = (int)v;
    return true;
  }
  for (const OptChoice *c = d.choices; c->word != nullptr; ++c)
    if (std::strcmp(c->word, word) == 0) {
      *out = c->value;
      return true;
    }
  return false;
}

void env_defaults(pnvo_model_s *m) {
  for (const OptDef &d : kOptions) {
    const char *e = d.env ? std::getenv(d.env) : nullptr;
    int v;
    if (e && parse_option(d, e, &v)) m->opt.*(d.field) = v;
  }
  // spellings of earlier rounds that do not fit the table
  if (std::getenv("PNVO_X3_S2_OFF")) m->opt.x3_s2 = 0;
  if (std::getenv("PNVO_BF16_NOFUSE")) m->opt.bf16_fuse = 0;
  if (const char *e = std::getenv("PNVO_STEM_DBG")) std::sscanf(e, "%d,%d", &m->opt.stem_dbg, &m->opt.stem_dbg_pad);
}
}  // namespace


// The stems that exploit the observation contract (uint8-valued rgb, one-hot depth: stem_mx.hip, stem_dd.hip) raise a
// host-mapped flag when a value breaks it.  pnvo_run_stem records an event behind such a stem; once the rest of the forward
// is enqueued the caller waits for THAT event (the stem is the first kernel of ~55: it has normally finished by then, and the
// GPU keeps the rest of the queue), and a raised flag re-runs the forward on the dense fp32 stem — which takes any float
// input, as the reference model does (vo_cnn.py:110-176) — and keeps the handle on it.
int pnvo_input_fallback(pnvo_handle m, hipStream_t s, bool *rerun) {
  *rerun = false;
  if (!m->stem_ev_pending) return PNVO_OK;
  m->stem_ev_pending = false;
  HIPCHK(m, hipEventSynchronize(m->stem_ev));
  if (!m->dd_flag || *(volatile int *)m->dd_flag == 0) return PNVO_OK;
  *(volatile int *)m->dd_flag = 0;
  HIPCHK(m, hipMemsetAsync(m->dd_flag_dev, 0, sizeof(int), s));
  m->dense_sticky = true;
  m->fallback_count += 1;
  pnvo_drop_graphs(m);
  m->note = "note: observation values outside the fused stems' contract (fractional rgb or depth codes that are not one-hot) — "
           "the forward was re-run on the dense fp32 stem and this handle stays on it";
  *rerun = true;
  return PNVO_OK;
}

namespace {
int forward_dispatch(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv,
                     const int64_t *actions, int B, float *out, hipStream_t s);
}


// ==================================================================================================================
extern "C" {

// "pnvo 0.3 (gfx950) src:<sha256[:16] of the tracked sources this library was built from>": the Makefile writes src_hash.h from
// csrc/*.hip, csrc/*.h and include/pnvo.h (sorted by name), tests/test_abi.py recomputes it — a loaded .so that does not match the
// tree it sits in is caught by the round-end GPU test, not by a reader of the numbers.
#include "src_hash.h"
const char *pnvo_version(void) { return "pnvo 0.3 (gfx950: fp32 + f16/bf16 MFMA) src:" PNVO_SRC_HASH; }

const char *pnvo_last_error(pnvo_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

const char *pnvo_last_note(pnvo_handle h) {
  if (h && h->loaded && h->opt.input_fallback && stem_lds_serves(h)) (void)pnvo_check_inputs(h);     // a raised input-contract flag (host-mapped, no wait) is noted here at the latest
  return h ? h->note.c_str() : "";
}

int pnvo_create(const pnvo_config *cfg, int device, pnvo_handle *out) {
  if (!cfg || !out) return fail(nullptr, PNVO_ERR_ARG, "null argument");
  const int C = cfg->n_rgb + cfg->n_depth + cfg->n_dd + cfg->n_tdv;
  if (C <= 0) return fail(nullptr, PNVO_ERR_ARG, "visual odometry must not be blind (no input modality)");   // vo_cnn.py:68
  if (C > 64 || (cfg->n_rgb % 2) || (cfg->n_depth % 2) || (cfg->n_dd % 2) || (cfg->n_tdv % 2))
    return fail(nullptr, PNVO_ERR_ARG, "unsupported input channel configuration");
  if (cfg->width < 32 || cfg->height < 32) return fail(nullptr, PNVO_ERR_ARG, "observation_size must be >= 32x32");
  if (cfg->baseplanes < 32 || cfg->baseplanes % 32 || cfg->hidden % 8 || cfg->hidden <= 0 || cfg->out_dim <= 0)
    return fail(nullptr, PNVO_ERR_ARG, "unsupported baseplanes / hidden_size / output_dim");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, PNVO_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(nullptr, PNVO_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, PNVO_ERR_ARG, std::string("libpnvo is built for gfx950 only, device is ") + prop.gcnArchName);
  pnvo_model_s *m = new pnvo_model_s();
  static std::atomic<unsigned long long> next_uid{1};
  m->uid = next_uid.fetch_add(1);
  m->cfg = *cfg;
  m->device = device;
  m->num_cus = prop.multiProcessorCount;
  if (m->cfg.flat_size <= 0) m->cfg.flat_size = 2048;
  if (m->cfg.n_acts <= 0) m->cfg.n_acts = 4;
  env_defaults(m);                 // the ONLY place the PNVO_* environment is read for this handle
  build_plan(m);
  *out = m;
  return PNVO_OK;
}

int pnvo_load_weights(pnvo_handle h, const float *blob, size_t n_floats, const pnvo_tensor_desc *toc, int ntoc) {
  if (!h || !blob || !toc) return fail(h, PNVO_ERR_ARG, "null argument");
  HIPCHK(h, hipSetDevice(h->device));
  pnvo_drop_graphs(h);            // weight buffers may be re-allocated
  Toc t;
  t.blob = blob;
  t.n = n_floats;
  for (int k = 0; k < ntoc; ++k) t.by_name[toc[k].name] = &toc[k];
  int rc = PNVO_OK;
  const pnvo_config &c = h->cfg;
  h->mean.assign(h->C, 0.f);
  h->stdev.assign(h->C, 1.f);
  if (c.normalize) {
    const std::string pre = "visual_encoder.running_mean_and_var.";
    const float *mu = find_tensor(h, t, pre + "_mean", {1, h->C, 1, 1}, &rc);
    if (!mu) return rc;
    const float *var = find_tensor(h, t, pre + "_var", {1, h->C, 1, 1}, &rc);
    if (!var) return rc;
    for (int k = 0; k < h->C; ++k) {
      h->mean[k] = mu[k];
      h->stdev[k] = std::sqrt(std::fmax(var[k], 1e-2f));   // running_mean_and_var.py:62, float32
    }
  }
  for (Layer &l : h->convs)
    if ((rc = load_conv(h, t, l, true)) != PNVO_OK) return rc;
  if (!h->bottleneck) {
    // Upper bounds of |activation| entering each conv of the residual stages, from the GroupNorm parameters alone: a group of N
    // elements normalised to unit variance has |x^| <= sqrt(N), so |relu(GN(x))| <= max_c (|gamma_c| sqrt(N) + |beta_c|); a block
    // output adds its skip branch.  The two-piece float16 operands of conv_x3 need the bound below 65504 (pnvo_run_conv).
    auto gn_bound = [&](const Layer &l) {
      const float *g = find_tensor(h, t, l.gn + ".weight", {l.cout}, &rc), *b = find_tensor(h, t, l.gn + ".bias", {l.cout}, &rc);
      if (!g || !b) return 3.0e38f;
      const double rootn = std::sqrt((double)(l.cout / l.groups) * l.hout * l.wout);
      double mx = 0.0;
      for (int c2 = 0; c2 < l.cout; ++c2) mx = std::fmax(mx, std::fabs((double)g[c2]) * rootn + std::fabs((double)b[c2]));
      return (float)std::fmin(mx, 3.0e38);
    };
    pnvo_chain_in_bounds(h, gn_bound);
  }
  {
    // fused stem: re-pack conv1 with its input channels in observation-tensor order, and fold /255 and the
    // RunningMeanAndVar whitening into x*sc+sh:  (x/255 - mean)/std = x * 1/(255 std) - mean/std
    Layer &st = h->convs[0];
    const float *w = find_tensor(h, t, st.name + ".weight", {st.cout, st.cin, st.k, st.kw}, &rc);
    if (!w) return rc;
    const int T = st.k * st.kw;
    std::vector<float> wp((size_t)st.cout * h->CP * T, 0.f), sc(h->CPL, 0.f), sh(h->CPL, 0.f);
    for (int nc = 0; nc < h->CP; ++nc) {
      const int rc_ = h->stem_ref_of_new[nc];
      if (rc_ < 0) continue;
      for (int o = 0; o < st.cout; ++o)
        std::memcpy(&wp[((size_t)o * h->CP + nc) * T], &w[((size_t)o * st.cin + rc_) * T], sizeof(float) * T);
      const double sd = (double)h->stdev[rc_], mu = (double)h->mean[rc_];
      const double div = (h->stem_tensor_of_new[nc] == 0) ? 255.0 : 1.0;
      sc[nc] = (float)(1.0 / (div * sd));
      sh[nc] = (float)(-mu / sd);
    }
    std::vector<float> pk;
    pack_conv_weight_cinp(wp.data(), st.cout, h->CP, h->CP, st.k, st.kw, pk);
    if ((rc = upload(h, st.wpk, pk.data(), pk.size())) != PNVO_OK) return rc;
    if ((rc = upload(h, h->stem_sc, sc.data(), sc.size())) != PNVO_OK) return rc;
    if ((rc = upload(h, h->stem_sh, sh.data(), sh.size())) != PNVO_OK) return rc;
    std::vector<float> z(64, 0.f);
    if ((rc = upload(h, h->zero_page, z.data(), z.size())) != PNVO_OK) return rc;
    std::vector<float> pk16((size_t)49 * h->CPL * st.cout);
    pack_stem_weight(wp.data(), st.cout, h->CP, h->CPL, pk16.data());
    if ((rc = upload(h, h->stem_wpk16, pk16.data(), pk16.size())) != PNVO_OK) return rc;
    // ---- one-hot-aware stem: split the input channels into dense (matrix cores) and one-hot depth bins (table gather)
    h->dd_ok = false;
    const int bins = c.n_dd / 2;
    if (c.n_dd > 0 && stem_dd_supported(bins) && st.cout == 32 && c.normalize) {
      h->dd_dense_tensor.clear();
      h->dd_dense_ch.clear();
      std::vector<int> dense_ref;
      for (int nc = 0; nc < h->CP; ++nc)
        if (h->stem_ref_of_new[nc] >= 0 && h->stem_tensor_of_new[nc] != 2) {
          h->dd_dense_tensor.push_back(h->stem_tensor_of_new[nc]);
          h->dd_dense_ch.push_back(h->stem_ch_of_new[nc]);
          dense_ref.push_back(h->stem_ref_of_new[nc]);
        }
      const int nd = (int)dense_ref.size();
      if (nd % 2 == 0 && nd + 1 <= 12) {
        std::vector<float> wd((size_t)st.cout * 12 * T, 0.f), sc16(12, 0.f), sh16(12, 0.f);
        for (int d = 0; d < nd; ++d) {
          const int r = dense_ref[d];
          for (int o = 0; o < st.cout; ++o)
            std::memcpy(&wd[((size_t)o * 12 + d) * T], &w[((size_t)o * st.cin + r) * T], sizeof(float) * T);
          const double sd = (double)h->stdev[r], mu = (double)h->mean[r];
          const double div = (h->dd_dense_tensor[d] == 0) ? 255.0 : 1.0;
          sc16[d] = (float)(1.0 / (div * sd));
          sh16[d] = (float)(-mu / sd);
        }
        sh16[nd] = 1.0f;                                   // indicator channel: 1 inside the image (x = 0, sc = 0)
        while ((int)h->dd_dense_tensor.size() < 12) {
          h->dd_dense_tensor.push_back(-1);
          h->dd_dense_ch.push_back(0);
        }
        // reference channel of one-hot entry (frame f, bin b)
        std::vector<int> ddref(2 * bins, -1);
        for (int nc = 0; nc < h->CP; ++nc)
          if (h->stem_ref_of_new[nc] >= 0 && h->stem_tensor_of_new[nc] == 2) ddref[h->stem_ch_of_new[nc]] = h->stem_ref_of_new[nc];
        const int slice = stem_dd_slice_floats(bins), brows = bins + 1;
        std::vector<float> tab((size_t)7 * slice, 0.f);
        for (int kh = 0; kh < 7; ++kh)
          for (int kw = 0; kw < 7; ++kw) {
            const int tap = kh * 7 + kw;
            for (int o = 0; o < st.cout; ++o) {
              double ind = 0.0;
              for (int f = 0; f < 2; ++f)
                for (int b = 0; b < bins; ++b) {
                  const int r = ddref[f * bins + b];
                  const double wv = (double)w[((size_t)o * st.cin + r) * T + tap];
                  const double sd = (double)h->stdev[r], mu = (double)h->mean[r];
                  tab[(size_t)kh * slice + (((size_t)kw * brows + b) * 2 + f) * 32 + o] = (float)(wv / sd);
                  ind -= wv * mu / sd;
                }
              wd[((size_t)o * 12 + nd) * T + tap] = (float)ind;   // weight of the "inside the image" indicator
            }
          }
        std::vector<float> pkd((size_t)49 * 12 * st.cout);
        pack_stem_dd_weight(wd.data(), st.cout, pkd.data());
        if ((rc = upload(h, h->dd_wpk, pkd.data(), pkd.size())) != PNVO_OK) return rc;
        if ((rc = upload(h, h->dd_table, tab.data(), tab.size())) != PNVO_OK) return rc;
        if ((rc = upload(h, h->dd_sc, sc16.data(), 12)) != PNVO_OK) return rc;
        if ((rc = upload(h, h->dd_sh, sh16.data(), 12)) != PNVO_OK) return rc;
        if (!h->dd_flag) HIPCHK(h, hipHostMalloc((void **)&h->dd_flag, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
        if (!h->dd_flag_dev) HIPCHK(h, hipMalloc((void **)&h->dd_flag_dev, sizeof(int)));
        *(volatile int *)h->dd_flag = 0;
        HIPCHK(h, hipMemset(h->dd_flag_dev, 0, sizeof(int)));
        h->dd_bins = bins;
        h->dd_ok = true;
      }
    }
  }
  {
    // ---- stem on the bf16 matrix cores (stem_mx.hip)
    Layer &st = h->convs[0];
    const float *w = find_tensor(h, t, st.name + ".weight", {st.cout, st.cin, st.k, st.kw}, &rc);
    if (!w) return rc;
    h->mx_ok = false;
    const int nsrc[4] = {c.n_rgb, c.n_depth, c.n_dd, c.n_tdv};
    // fixed K-slot layout of stem_mx.hip: 0-19 discretised depth | 20-25 rgb | 26-27 depth | 28-29 top-down view | 30 indicator
    const bool shape_ok = (c.n_rgb == 0 || c.n_rgb == 6) && (c.n_depth == 0 || c.n_depth == 2) && (c.n_dd == 0 || c.n_dd == 20) &&
                          (c.n_tdv == 0 || c.n_tdv == 2);
    if (shape_ok && (st.cout == 32 || st.cout == 64) && st.k == 7) {
      std::vector<int> slot_ref(32, -1), slot_ref_sw(32, -1), slot_tensor(32, -1), slot_new(32, -1);
      const int first_slot[4] = {20, 26, 0, 28};           // rgb, depth, dd, tdv
      for (int x = 0; x < 4; ++x) h->mx_xslot[x] = -1;
      if (c.n_depth) h->mx_xslot[0] = 26, h->mx_xslot[1] = 27;
      if (c.n_tdv) h->mx_xslot[2] = 28, h->mx_xslot[3] = 29;
      for (int tn = 0; tn < 4; ++tn)
        for (int ch = 0; ch < nsrc[tn]; ++ch) {
          int nc = -1, ncs = -1;                            // position in the stem's "new" (tensor-major) channel order
          for (int k = 0; k < h->CP; ++k) {
            if (h->stem_tensor_of_new[k] != tn) continue;
            if (h->stem_ch_of_new[k] == ch) nc = k;
            if (h->stem_ch_of_new[k] == (ch + nsrc[tn] / 2) % nsrc[tn]) ncs = k;   // the frame-swapped twin
          }
          slot_ref[first_slot[tn] + ch] = h->stem_ref_of_new[nc];
          slot_new[first_slot[tn] + ch] = nc;
          slot_ref_sw[first_slot[tn] + ch] = h->stem_ref_of_new[ncs];
          slot_tensor[first_slot[tn] + ch] = tn;
        }
      const int T = 49;
      auto fold = [&](const std::vector<int> &ref, std::vector<float> &wk) {
        wk.assign((size_t)st.cout * 32 * T, 0.f);
        for (int o = 0; o < st.cout; ++o)
          for (int tap = 0; tap < T; ++tap) {
            double ind = 0.0;
            for (int k = 0; k < 30; ++k) {
              const int r = ref[k];
              if (r < 0) continue;
              const double wv = (double)w[((size_t)o * st.cin + r) * T + tap];
              const double sd = (double)h->stdev[r], mu = (double)h->mean[r];
              const double div = slot_tensor[k] == 0 ? 255.0 : 1.0;
              wk[((size_t)o * 32 + k) * T + tap] = (float)(wv / (div * sd));
              ind -= wv * mu / sd;
            }
            wk[((size_t)o * 32 + 30) * T + tap] = (float)ind;      // "inside the image" indicator
          }
      };
      fold(slot_ref, h->mx_wk);
      fold(slot_ref_sw, h->mx_wk_swapped);
      h->mx_slot_ref = slot_ref;
      h->mx_slot_new = slot_new;
      std::vector<unsigned short> pk(stem_mx_packed_u16(3, st.cout / 32));
      pack_stem_mx_weight(h->mx_wk.data(), st.cout, 3, h->mx_xslot, pk.data());
      if ((rc = upload(h, reinterpret_cast<float *&>(h->mx_wpk3), reinterpret_cast<const float *>(pk.data()), pk.size() / 2)) !=
          PNVO_OK)
        return rc;
      {
        std::vector<unsigned short> pk2(stem_mx_packed_u16(2, st.cout / 32));
        h->mx_oscale = pack_stem_mx_weight_h(h->mx_wk.data(), st.cout, h->mx_xslot, pk2.data());
        h->mx_wpk2_dev = false;
        if ((rc = upload(h, reinterpret_cast<float *&>(h->mx_wpk2), reinterpret_cast<const float *>(pk2.data()), pk2.size() / 2)) !=
            PNVO_OK)
          return rc;
      }
      {
        std::vector<float> pages(64, 0.f);
        if ((rc = upload(h, h->mx_pages, pages.data(), pages.size())) != PNVO_OK) return rc;
      }
      if (!h->dd_flag) HIPCHK(h, hipHostMalloc((void **)&h->dd_flag, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
      if (!h->dd_flag_dev) HIPCHK(h, hipMalloc((void **)&h->dd_flag_dev, sizeof(int)));
      *(volatile int *)h->dd_flag = 0;
      HIPCHK(h, hipMemset(h->dd_flag_dev, 0, sizeof(int)));
      h->mx_ok = true;
    }
  }
  // Linear(flat[+embed] -> hidden): visual columns become the fh x fw "conv"; the embedding columns fold into a
  // per-action bias row:  bias[a][o] = b[o] + sum_e W[o][flat+e] * emb[a][e]   (vo_cnn_act_embed.py:63-72)
  const int flat = h->comp_c * h->fh * h->fw;
  const int fc_in = flat + (c.act_embed ? 32 : 0);
  const float *w1 = find_tensor(h, t, h->fc.name + ".weight", {c.hidden, fc_in}, &rc);
  if (!w1) return rc;
  const float *b1 = find_tensor(h, t, h->fc.name + ".bias", {c.hidden}, &rc);
  if (!b1) return rc;
  {
    std::vector<float> vis((size_t)c.hidden * flat);
    for (int o = 0; o < c.hidden; ++o) std::memcpy(&vis[(size_t)o * flat], w1 + (size_t)o * fc_in, sizeof(float) * flat);
    std::vector<float> pk;
    h->fc.host_w = vis;                                    // (smallnet.hip packs its own layout from these)
    {                                                      // fc_rows.hip: plain rows in the activation's (h, w, padded channel) order
      const int hw = h->fh * h->fw, kp = hw * h->comp_cp;
      std::vector<float> rows((size_t)c.hidden * kp, 0.f);
      for (int o = 0; o < c.hidden; ++o)
        for (int ch = 0; ch < h->comp_c; ++ch)
          for (int q = 0; q < hw; ++q) rows[(size_t)o * kp + (size_t)q * h->comp_cp + ch] = vis[(size_t)o * flat + (size_t)ch * hw + q];
      if ((rc = upload(h, h->fc_rows_w, rows.data(), rows.size())) != PNVO_OK) return rc;
    }
    pack_conv_weight_cinp(vis.data(), c.hidden, h->comp_c, h->comp_cp, h->fh, h->fw, pk);
    if ((rc = upload(h, h->fc.wpk, pk.data(), pk.size())) != PNVO_OK) return rc;
    const int rows = c.act_embed ? c.n_acts + 1 : 1;
    std::vector<float> bias((size_t)rows * c.hidden);
    const float *emb = nullptr;
    if (c.act_embed) {
      emb = find_tensor(h, t, "action_embedding.weight", {c.n_acts + 1, 32}, &rc);
      if (!emb) return rc;
    }
    for (int a = 0; a < rows; ++a)
      for (int o = 0; o < c.hidden; ++o) {
        double acc = 0.0;
        if (emb)
          for (int e = 0; e < 32; ++e) acc += (double)w1[(size_t)o * fc_in + flat + e] * (double)emb[a * 32 + e];
        bias[(size_t)a * c.hidden + o] = (float)(acc + (double)b1[o]);
      }
    if ((rc = upload(h, h->fc_bias, bias.data(), bias.size())) != PNVO_OK) return rc;
  }
  const float *w2 = find_tensor(h, t, "output_head.1.weight", {c.out_dim, c.hidden}, &rc);
  if (!w2) return rc;
  const float *b2 = find_tensor(h, t, "output_head.1.bias", {c.out_dim}, &rc);
  if (!b2) return rc;
  {
    std::vector<float> pk;
    h->head.host_w.assign(w2, w2 + (size_t)c.out_dim * c.hidden);
    pack_conv_weight_cinp(w2, c.out_dim, c.hidden, c.hidden, 1, 1, pk);
    if ((rc = upload(h, h->head.wpk, pk.data(), pk.size())) != PNVO_OK) return rc;
    if ((rc = upload(h, h->head_bias, b2, c.out_dim)) != PNVO_OK) return rc;
    if ((rc = upload(h, h->head_w_plain, w2, (size_t)c.out_dim * c.hidden)) != PNVO_OK) return rc;
  }
  h->loaded = true;
  h->load_gen += 1;
  h->weights_gen += 1;
  h->weights_gen_at_load = h->weights_gen;
  return PNVO_OK;
}

int pnvo_set_option(pnvo_handle h, const char *key, const char *value) {
  if (!h || !key || !value) return fail(h, PNVO_ERR_ARG, "null argument");
  const OptDef *d = find_option(key);
  if (!d) return fail(h, PNVO_ERR_ARG, std::string("unknown option '") + key + "'");
  int v;
  if (!parse_option(*d, value, &v)) {
    std::string msg = std::string("option '") + key + "' does not take '" + value + "' (";
    if (d->numeric) msg += "an integer";
    for (const OptChoice *c = d->choices; !d->numeric && c->word != nullptr; ++c) msg += std::string(c == d->choices ? "" : " | ") + c->word;
    return fail(h, PNVO_ERR_ARG, msg + ")");
  }
  const bool lift = d->field == &PnvoOptions::stem && h->dense_sticky;   // an explicit stem choice lifts the fallback
  if (h->opt.*(d->field) == v && !lift) return PNVO_OK;
  h->opt.*(d->field) = v;
  pnvo_drop_graphs(h);                       // captured launches encode the kernel selection
  if (lift) {
    // repairs of forwards still in flight read the flag on the device: wait for them before it is lowered (a rare, explicit call)
    if (h->dd_flag && *(volatile int *)h->dd_flag != 0) {
      HIPCHK(h, hipSetDevice(h->device));
      HIPCHK(h, hipDeviceSynchronize());
      *(volatile int *)h->dd_flag = 0;
      HIPCHK(h, hipMemset(h->dd_flag_dev, 0, sizeof(int)));
    }
    h->dense_sticky = false;
  }
  return PNVO_OK;
}

int pnvo_get_option(pnvo_handle h, const char *key, char *buf, size_t cap) {
  if (!h || !key || !buf || cap == 0) return fail(h, PNVO_ERR_ARG, "null argument");
  const OptDef *d = find_option(key);
  if (!d) return fail(h, PNVO_ERR_ARG, std::string("unknown option '") + key + "'");
  const int v = h->opt.*(d->field);
  std::string word = std::to_string(v);
  for (const OptChoice *c = d->choices; !d->numeric && c->word != nullptr; ++c)
    if (c->value == v) {
      word = c->word;
      break;
    }
  if (d->field == &PnvoOptions::stem && h->loaded && h->opt.input_fallback && stem_lds_serves(h)) (void)pnvo_check_inputs(h);   // see pnvo_last_note
  if (d->field == &PnvoOptions::stem && h->dense_sticky) word = "dense (fallback)";
  std::snprintf(buf, cap, "%s", word.c_str());
  return PNVO_OK;
}

int pnvo_set_precision(pnvo_handle h, int precision) {
  if (!h) return fail(h, PNVO_ERR_ARG, "null handle");
  if (precision != 0 && precision != 1) return fail(h, PNVO_ERR_ARG, "precision must be 0 (float32) or 1 (bfloat16)");
  h->precision = precision;
  return PNVO_OK;
}

int pnvo_forward_dual(pnvo_handle ha, pnvo_handle hb, const float *rgb, const float *depth, const float *dd, const float *tdv,
                      int B, float *out_a, float *out_b, void *stream) {
  if (!ha || !hb) return fail(ha, PNVO_ERR_ARG, "null handle");
  if (!ha->loaded || !hb->loaded) return fail(ha, PNVO_ERR_STATE, "pnvo_forward_dual before pnvo_load_weights");
  if (B <= 0 || !out_a || !out_b) return fail(ha, PNVO_ERR_ARG, "bad batch / null output");
  if (ha->precision != 1 || hb->precision != 1)
    return fail(ha, PNVO_ERR_STATE, "pnvo_forward_dual runs the bfloat16 path: call pnvo_set_precision(h, 1) on both models");
  if (std::memcmp(&ha->cfg, &hb->cfg, sizeof(pnvo_config)) != 0 || ha->device != hb->device)
    return fail(ha, PNVO_ERR_ARG, "the two models of a dual forward must share architecture and device");
  if (ha->cfg.act_embed)
    return fail(ha, PNVO_ERR_ARG, "pnvo_forward_dual covers the separate-action models (act_left_right_inv_joint); act-embed "
                                  "models take their actions through pnvo_forward");
  const pnvo_config &c = ha->cfg;
  if ((c.n_rgb > 0) != (rgb != nullptr) || (c.n_depth > 0) != (depth != nullptr) || (c.n_dd > 0) != (dd != nullptr) ||
      (c.n_tdv > 0) != (tdv != nullptr))
    return fail(ha, PNVO_ERR_ARG, "observation tensors do not match the model's observation_space");
  if (int rc0 = pnvo_check_inputs(ha)) return rc0;
  HIPCHK(ha, hipSetDevice(ha->device));
  pnvo_handle hs[2] = {ha, hb};
  float *outs[2] = {out_a, out_b};
  return pnvo_forward_bf16(hs, 2, rgb, depth, dd, tdv, nullptr, B, outs, (hipStream_t)stream);
}

int pnvo_check_inputs(pnvo_handle m) {
  if (!m) return fail(m, PNVO_ERR_ARG, "null handle");
  if (m->dd_flag && *(volatile int *)m->dd_flag != 0) {
    // already repaired: the handle runs the float32 stand-in since an earlier check; the flag only stays up for forwards still in
    // flight (turning input_fallback off afterwards must not turn every later forward into an error)
    if (m->dense_sticky) return PNVO_OK;
    if (m->opt.input_fallback && !m->in_train_forward && stem_lds_serves(m)) {
      // The flag is host-mapped: read without waiting for anything.  The forward that raised it repaired itself on the device
      // (pnvo_stem_standin behind its stem); from here on this handle launches the float32 stand-in directly.  The flag stays up
      // — repairs of forwards still in flight read it — until pnvo_set_option(h, "stem", ..) lifts the fallback.
      if (!m->dense_sticky) {
        m->dense_sticky = true;
        m->fallback_count += 1;
        pnvo_drop_graphs(m);
        m->note = "note: observation values outside the fused stems' contract (fractional rgb or depth codes that are not one-hot) — "
                  "the forward redid its stem on float32 operands (decided on the device) and this handle stays on the dense float32 stem";
      }
      return PNVO_OK;
    }
    return fail(m, PNVO_ERR_INPUT,
                "an earlier forward met observation values outside the reference's contract — a discretised-depth pixel "
                "that is not one-hot (base_trainer_with_vo.py:163) or an rgb value that is not an integer 0..255 — so its "
                "outputs are invalid (this handle runs with input_fallback=off, is inside a training forward, or its model is one the "
                "float32 stand-in stem does not serve — e.g. a forward that was stream-captured on such a model).  Feed "
                "contract inputs, keep option input_fallback on, or select the dense stem: pnvo_set_option(h, \"stem\", \"dense\")");
  }
  return PNVO_OK;
}

namespace {
int forward_body(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv,
                 const int64_t *actions, int B, float *out, hipStream_t s);

}  // namespace

int pnvo_forward(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv,
                 const int64_t *actions, int B, float *out, void *stream) {
  if (!m) return fail(m, PNVO_ERR_ARG, "null handle");
  if (!m->loaded) return fail(m, PNVO_ERR_STATE, "pnvo_forward before pnvo_load_weights");
  if (B <= 0 || !out) return fail(m, PNVO_ERR_ARG, "bad batch / null output");
  const pnvo_config &c = m->cfg;
  if ((c.n_rgb > 0) != (rgb != nullptr) || (c.n_depth > 0) != (depth != nullptr) || (c.n_dd > 0) != (dd != nullptr) ||
      (c.n_tdv > 0) != (tdv != nullptr))
    return fail(m, PNVO_ERR_ARG, "observation tensors do not match the model's observation_space");
  if (c.act_embed && !actions) return fail(m, PNVO_ERR_ARG, "act_embed model needs actions");
  if (int rc0 = pnvo_check_inputs(m)) return rc0;
  HIPCHK(m, hipSetDevice(m->device));
  if (m->precision == 1) {
    pnvo_handle hs[1] = {m};
    float *outs[1] = {out};
    return pnvo_forward_bf16(hs, 1, rgb, depth, dd, tdv, actions, B, outs, (hipStream_t)stream);
  }
  int rc = ensure_workspace(m, B);
  if (rc != PNVO_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = forward_dispatch(m, rgb, depth, dd, tdv, actions, B, out, s)) != PNVO_OK) return rc;
  bool rerun = false;
  if ((rc = pnvo_input_fallback(m, s, &rerun)) != PNVO_OK) return rc;
  return rerun ? forward_dispatch(m, rgb, depth, dd, tdv, actions, B, out, s) : PNVO_OK;
}

namespace {
int forward_dispatch(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv,
                     const int64_t *actions, int B, float *out, hipStream_t s) {
  const pnvo_config &c = m->cfg;
  int rc = PNVO_OK;
  {
    // Opt-in (option graph=1).  Measured on ROCm 7.2 / MI355X: replaying the ~60-node graph is no faster than the plain
    // asynchronous launches — 6.32 vs 6.22 ms at B = 256 (the GPU is never starved) and 0.77 vs 0.80 ms at B = 1 (the
    // dependent-kernel latency of the chain, not the host launches, is what a batch-1 call pays).
    m->graph_mode = m->opt.graph;
  }
  const bool plain = !m->graph_mode || m->timing || m->tap_dst != nullptr || m->train != nullptr || m->opt.stem_dbg != 0;
  if (plain) return forward_body(m, rgb, depth, dd, tdv, actions, B, out, s);

  // ---- graph replay: key = everything the captured kernel arguments depend on
  const void *key[8] = {rgb, depth, dd, tdv, actions, m->raw_rgb, m->raw_depth, m->raw_err};   // (pnvo_set_option drops the captured graphs)
  const size_t out_bytes = (size_t)B * c.out_dim * sizeof(float);
  for (auto &g : m->graphs)
    if (g.B == B && std::memcmp(g.key, key, sizeof(key)) == 0) {
      g.stamp = ++m->graph_clock;
      HIPCHK(m, hipGraphLaunch(g.exec, s));
      HIPCHK(m, hipMemcpyAsync(out, m->out_ws, out_bytes, hipMemcpyDeviceToDevice, s));
      return pnvo_mark_stem(m, s);
    }
  bool again = false;                              // capture only call shapes that come back
  for (auto &g : m->seen) again = again || (g.B == B && std::memcmp(g.key, key, sizeof(key)) == 0);
  if (!again) {
    pnvo_model_s::GraphEntry sn;
    std::memset(&sn, 0, sizeof(sn));
    std::memcpy(sn.key, key, sizeof(key));
    sn.B = B;
    if (m->seen.size() >= 16) m->seen.erase(m->seen.begin());
    m->seen.push_back(sn);
    return forward_body(m, rgb, depth, dd, tdv, actions, B, out, s);
  }
  if (!m->cap_stream) HIPCHK(m, hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
  pnvo_model_s::GraphEntry g;
  std::memcpy(g.key, key, sizeof(key));
  g.B = B;
  g.stamp = ++m->graph_clock;
  HIPCHK(m, hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
  rc = forward_body(m, rgb, depth, dd, tdv, actions, B, m->out_ws, m->cap_stream);
  const hipError_t ce = hipStreamEndCapture(m->cap_stream, &g.graph);
  if (rc != PNVO_OK) {
    if (ce == hipSuccess && g.graph) (void)hipGraphDestroy(g.graph);
    return rc;
  }
  HIPCHK(m, ce);
  HIPCHK(m, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
  if (m->graphs.size() >= 8) {                     // keep the 8 most recently used call shapes
    size_t old = 0;
    for (size_t k = 1; k < m->graphs.size(); ++k)
      if (m->graphs[k].stamp < m->graphs[old].stamp) old = k;
    (void)hipGraphExecDestroy(m->graphs[old].exec);
    (void)hipGraphDestroy(m->graphs[old].graph);
    m->graphs.erase(m->graphs.begin() + (long)old);
  }
  m->graphs.push_back(g);
  HIPCHK(m, hipGraphLaunch(g.exec, s));
  HIPCHK(m, hipMemcpyAsync(out, m->out_ws, out_bytes, hipMemcpyDeviceToDevice, s));
  return pnvo_mark_stem(m, s);
}
}  // namespace

namespace {
// Side stream, its two events and a statistics buffer of the main one's size (lazily; nothing while `s` is being captured: a graph
// keeps the in-stream order).
bool side_stream_ready(pnvo_handle m, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  if (!m->side_stream && hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking) != hipSuccess) return false;
  if (!m->side_fork && hipEventCreateWithFlags(&m->side_fork, hipEventDisableTiming) != hipSuccess) return false;
  if (!m->side_join && hipEventCreateWithFlags(&m->side_join, hipEventDisableTiming) != hipSuccess) return false;
  if (m->stats_side_floats < m->stats_floats) {
    if (m->stats_side) (void)hipFree(m->stats_side);
    m->stats_side = nullptr;
    m->stats_side_floats = 0;
    if (hipMalloc((void **)&m->stats_side, m->stats_floats * sizeof(float)) != hipSuccess) return false;
    m->stats_side_floats = m->stats_floats;
  }
  return true;
}

int run_fc_head(pnvo_handle m, int B, const float *comp_raw, const float *sc, const float *sh, const int64_t *actions, float *out, hipStream_t s);
bool fc_rows_usable(pnvo_handle m, int B);
int run_fc_rows(pnvo_handle m, pnvo_handle const *grp, const int *end, int ng, int B, const float *comp_raw, const float *sc, const float *sh,
                const int64_t *actions, float *out, hipStream_t s);

// Stream + events of option pool_async (lazily; not while `s` is being captured: a graph keeps the fill in-stream).
bool keys_stream_ready(pnvo_handle m, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  if (!m->keys_stream && hipStreamCreateWithFlags(&m->keys_stream, hipStreamNonBlocking) != hipSuccess) return false;
  if (!m->keys_free_ev && hipEventCreateWithFlags(&m->keys_free_ev, hipEventDisableTiming) != hipSuccess) return false;
  if (!m->keys_ready_ev && hipEventCreateWithFlags(&m->keys_ready_ev, hipEventDisableTiming) != hipSuccess) return false;
  return true;
}

int forward_body(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv,
                 const int64_t *actions, int B, float *out, hipStream_t s) {
  const pnvo_config &c = m->cfg;
  int rc = PNVO_OK;
  // (a4+a5+a6) input assembly + /255 + whitening are fused into the stem kernel's operand fetch (pnvo_run_stem): the
  // [B,H,W,30] tensor of the reference (vo_cnn.py:174-176) is never materialised.
  if (m->tap_dst != nullptr && m->tap_name == "input") {   // introspection only: materialise it for the tap
    AssembleArgs a;
    a.src[0] = rgb;
    a.src[1] = depth;
    a.src[2] = dd;
    a.src[3] = tdv;
    a.nsrc[0] = c.n_rgb;
    a.nsrc[1] = c.n_depth;
    a.nsrc[2] = c.n_dd;
    a.nsrc[3] = c.n_tdv;
    a.mean = c.normalize ? m->mean.data() : nullptr;
    a.stdev = m->stdev.data();
    a.C = m->C;
    a.CP = m->CP;
    a.npix = (long)B * c.height * c.width;
    free_dev(m->xin);
    HIPCHK(m, hipMalloc((void **)&m->xin, (size_t)a.npix * m->CP * sizeof(float)));
    a.out = m->xin;
    HIPCHK(m, launch_assemble(a, s));
    if ((rc = maybe_tap(m, "input", m->xin, (size_t)B * c.height * c.width * m->CP, s)) != PNVO_OK) return rc;
  }
  size_t li = 0;
  for (auto &pd : m->gn_pend) pd.valid = false;
  m->defer_main = m->defer_ride = false;
  const Layer &stem = m->convs[li++];
  // (a7) GN + ReLU + maxpool.  Default: no pass at all — the stem writes pooled order-preserving keys (stem_mx.hip POOL), the
  // first block's first conv decodes / normalises them while staging and writes the pooled activations the skip branch needs
  // (conv_x3 MODE 3).  PNVO_POOL=separate, taps, Bottleneck models and the other stem kernels keep the pass.
  float *cur = m->bufY[0], *nxt = m->bufY[1];
  const bool pool_fused = !m->bottleneck && pnvo_stem_on_mx(m) && m->opt.pool && m->convs.size() > 1 &&
                          stem.coutp == stem.cout && pnvo_conv_takes_tail(m, m->convs[1], B);
  // Batches of the navigation loop (one or two pairs): everything behind the stem conv is ONE persistent launch (smallnet.hip),
  // which also reduces the stem's GroupNorm statistics itself.
  float *keys = nxt;             // where the pooled keys live: the ping-pong buffer, or (pool_async) a buffer of their own
  bool keys_async = false;
  const bool small = !pool_fused && stem_writes_slots(m) && pnvo_small_usable(m, B);
  if (small) {
    const float *src[4] = {rgb, depth, dd, tdv};
    m->stem_skip_finalize = true;
    rc = pnvo_run_stem(m, B, src, m->stem_raw, m->ssA, nullptr, nullptr, s, nullptr);
    m->stem_skip_finalize = false;
    if (rc != PNVO_OK) return rc;
    return pnvo_small_forward(m, B, c.act_embed ? actions : nullptr, out, s);
  }
  {
    const float *src[4] = {rgb, depth, dd, tdv};
    if (pool_fused) {
      const size_t nkeys = (size_t)B * m->Hp * m->Wp * stem.coutp;
      keys_async = m->opt.pool_async && !m->timing && m->pool_keys != nullptr && keys_stream_ready(m, s);
      keys = keys_async ? m->pool_keys : nxt;
      if (keys_async && m->keys_primed) HIPCHK(m, hipStreamWaitEvent(s, m->keys_ready_ev, 0));   // the fill enqueued behind the last forward
      if (!(keys_async && m->keys_primed && m->keys_primed_B >= B)) {
        Timed t(m, s, "pool_init", 0.0, 4.0 * B * m->Hp * m->Wp * stem.coutp);
        HIPCHK(m, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(keys), STEM_POOL_INIT, nkeys, s));
      }
      if (keys_async) m->keys_primed = false;
    }
    if ((rc = pnvo_run_stem(m, B, src, m->stem_raw, m->ssA, nullptr, nullptr, s, pool_fused ? reinterpret_cast<int *>(keys) : nullptr)) !=
        PNVO_OK)
      return rc;
  }
  if ((rc = maybe_tap(m, "stem_conv", m->stem_raw, (size_t)B * m->Hs * m->Ws * stem.coutp, s)) != PNVO_OK) return rc;
  if (!pool_fused) {
    Timed t(m, s, "gn_relu_maxpool", 0.0, 4.0 * B * ((double)m->Hs * m->Ws + (double)m->Hp * m->Wp) * stem.coutp);
    HIPCHK(m, launch_gn_relu_maxpool(m->stem_raw, m->ssA[0], m->ssA[1], B, m->Hs, m->Ws, stem.coutp, cur, s));
  }
  if ((rc = maybe_tap(m, "maxpool", cur, (size_t)B * m->Hp * m->Wp * stem.coutp, s)) != PNVO_OK) return rc;

  // (a8) residual stages
  BlockTail tail{};
  bool have_tail = false, have_keys = pool_fused;
  for (int stage = 1; stage <= 4; ++stage) {
    for (int bi = 0; bi < m->nblocks[stage - 1]; ++bi) {
      if (m->bottleneck) {                           // conv1x1 -> GN -> ReLU -> conv3x3(s) -> GN -> ReLU -> conv1x1 -> GN
        const Layer &b1 = m->convs[li++];
        const Layer &b2 = m->convs[li++];
        const Layer &b3 = m->convs[li++];
        const bool dsb = (li < m->convs.size() && m->convs[li].name.find("downsample") != std::string::npos);
        if ((rc = run_conv(m, b1, B, cur, nullptr, nullptr, m->rawA, b1.coutp, m->ssA, nullptr, nullptr, 0, s)) != PNVO_OK)
          return rc;
        if ((rc = run_conv(m, b2, B, m->rawA, m->ssA[0], m->ssA[1], m->rawC, b2.coutp, m->ssB, nullptr, nullptr, 0, s)) !=
            PNVO_OK)
          return rc;
        if ((rc = run_conv(m, b3, B, m->rawC, m->ssB[0], m->ssB[1], m->rawB, b3.coutp, m->ssA, nullptr, nullptr, 0, s)) !=
            PNVO_OK)
          return rc;
        const long Pb = (long)b3.hout * b3.wout;
        if (dsb) {
          const Layer &cd = m->convs[li++];
          if ((rc = run_conv(m, cd, B, cur, nullptr, nullptr, m->rawD, cd.coutp, m->ssD, nullptr, nullptr, 0, s)) != PNVO_OK)
            return rc;
          Timed t(m, s, "residual", 0.0, 12.0 * B * Pb * b3.coutp);
          HIPCHK(m, launch_residual(m->rawB, m->ssA[0], m->ssA[1], m->rawD, m->ssD[0], m->ssD[1], B, Pb, b3.coutp, nxt, s));
        } else {
          Timed t(m, s, "residual", 0.0, 12.0 * B * Pb * b3.coutp);
          HIPCHK(m, launch_residual(m->rawB, m->ssA[0], m->ssA[1], cur, nullptr, nullptr, B, Pb, b3.coutp, nxt, s));
        }
        std::swap(cur, nxt);
        const std::string tnb = "layer" + std::to_string(stage) + "." + std::to_string(bi);
        if ((rc = maybe_tap(m, tnb.c_str(), cur, (size_t)B * Pb * b3.coutp, s)) != PNVO_OK) return rc;
        continue;
      }
      if (keys_async && !m->keys_primed && (stage >= 3 || (stage == 2 && m->nblocks[2] + m->nblocks[3] == 0))) {
        // the keys were consumed two stages ago: their fill for the NEXT forward goes to the key stream now, next to the MFMA-bound
        // deep stages of this one (the first stage's convs are HBM-bound themselves)
        HIPCHK(m, hipEventRecord(m->keys_free_ev, s));
        HIPCHK(m, hipStreamWaitEvent(m->keys_stream, m->keys_free_ev, 0));
        HIPCHK(m, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m->pool_keys), STEM_POOL_INIT, (size_t)B * m->Hp * m->Wp * stem.coutp,
                                    m->keys_stream));
        HIPCHK(m, hipEventRecord(m->keys_ready_ev, m->keys_stream));
        m->keys_primed = true;
        m->keys_primed_B = B;
      }
      const Layer &c1 = m->convs[li++];
      const Layer &c2 = m->convs[li++];
      const bool ds = (li < m->convs.size() && m->convs[li].name.find("downsample") != std::string::npos);
      // Deferred GroupNorm finalisation (option gn_defer, launches of up to that many pairs): c1's GroupNorm is finalised by c2's
      // launch when that is a conv_x3_kernel launch; c2's and the downsample conv's by the launch that takes this block's tail
      const bool defer_on = m->opt.gn_defer > 0 && B <= m->opt.gn_defer && m->tap_dst == nullptr && m->train == nullptr;
      const size_t li_next = li + (ds ? 1 : 0);
      const bool next_takes = li_next < m->convs.size() && pnvo_conv_takes_tail(m, m->convs[li_next], B);
      const bool c2_x3 = pnvo_conv_on_x3(m, c2, B);
      const bool defer_c1 = defer_on && c2_x3, defer_tail = defer_on && next_takes;
      // the block's downsample conv rides on c1's launch (conv_x3_kernel DSF): no launch, no finalisation of its own, and in the
      // block-tail mode the block input is not written to HBM at all — c1 and the downsample conv are its only readers
      const bool ds_ride = ds && !have_keys && pnvo_conv_takes_ds(m, c1, m->convs[li], B);
      const DsRide ride{ds_ride ? &m->convs[li] : nullptr, m->rawD, m->ssD, nullptr, nullptr};
      m->defer_main = defer_c1;
      m->defer_ride = defer_tail;
      if (have_keys) {           // pooled stem keys in `nxt`: decoded + normalised by this conv's stager, activations -> `cur`
        BlockTail ktail{nullptr, nullptr, nullptr, cur};
        if ((rc = pnvo_run_conv(m, c1, B, keys, m->ssA[0], m->ssA[1], m->rawA, c1.coutp, m->ssA, nullptr, nullptr, 0, s, nullptr, nullptr,
                                nullptr, &ktail)) != PNVO_OK)
          return rc;
        have_keys = false;
      } else if (have_tail) {    // the previous block's tail rides on this conv's stager, which also writes the block output
        if (ds_ride) tail.out = nullptr;                       // (nobody else reads this block's input)
        if ((rc = pnvo_run_conv(m, c1, B, m->rawB, m->ssB[0], m->ssB[1], m->rawA, c1.coutp, m->ssA, nullptr, nullptr, 0, s, nullptr,
                                nullptr, nullptr, &tail, ds_ride ? &ride : nullptr)) != PNVO_OK)
          return rc;
        std::swap(cur, nxt);
        have_tail = false;
      } else if ((rc = pnvo_run_conv(m, c1, B, cur, nullptr, nullptr, m->rawA, c1.coutp, m->ssA, nullptr, nullptr, 0, s, nullptr, nullptr,
                                     nullptr, nullptr, ds_ride ? &ride : nullptr)) != PNVO_OK) {
        return rc;
      }
      m->defer_main = m->defer_ride = false;
      const long P = (long)c2.hout * c2.wout;
      const bool c2_small_generic = !layer_on_lds(m, c2, nullptr) && !pnvo_conv_on_x3(m, c2, B) && (size_t)B * P * c2.cinp * 4 <= ((size_t)48 << 20);
      // The block's 1x1 stride-2 downsample conv reads only the block input (written by the first conv's stager) and owns rawD / ssD:
      // it and its GroupNorm finalisation run on a side stream NEXT TO the second 3x3 conv instead of behind it (option ds_side) —
      // forked here, joined where the downsample used to be launched.  Its GroupNorm partials go to a buffer of their own.
      bool ds_forked = false;
      // (measured: 256 pairs 2.331 -> 2.305 ms; 64 pairs no change; 16 pairs +0.05 ms — the fork / join events cost what the small
      //  launches save, so only from 128 pairs on)
      if (ds && !ds_ride && m->opt.ds_side && B >= 128 && !c2_small_generic && !m->timing && m->tap_dst == nullptr && m->train == nullptr && side_stream_ready(m, s)) {
        const Layer &cd = m->convs[li + 0];
        HIPCHK(m, hipEventRecord(m->side_fork, s));
        HIPCHK(m, hipStreamWaitEvent(m->side_stream, m->side_fork, 0));
        std::swap(m->stats, m->stats_side);
        rc = run_conv(m, cd, B, cur, nullptr, nullptr, m->rawD, cd.coutp, m->ssD, nullptr, nullptr, 0, m->side_stream);
        std::swap(m->stats, m->stats_side);
        if (rc != PNVO_OK) return rc;
        HIPCHK(m, hipEventRecord(m->side_join, m->side_stream));
        ds_forked = true;
      }
      if (c2_small_generic) {
        // small deep stage on the generic kernel: its per-tap GroupNorm+ReLU prologue costs more than one streaming
        // pass over the (L2-sized) tensor, so normalise once and run the conv on final activations
        // (scratch: rawD — unless the downsample conv rode on c1 and its output already sits there; `nxt` is free until the block tail)
        float *napp = ds_ride ? nxt : m->rawD;
        {
          Timed t(m, s, "gn_relu_apply", 0.0, 8.0 * B * P * c2.cinp);
          HIPCHK(m, launch_apply_ss_relu(m->rawA, m->ssA[0], m->ssA[1], B, P, c2.cinp, napp, s));
        }
        if ((rc = run_conv(m, c2, B, napp, nullptr, nullptr, m->rawB, c2.coutp, m->ssB, nullptr, nullptr, 0, s)) != PNVO_OK)
          return rc;
      } else {
        m->defer_main = defer_tail;
        rc = run_conv(m, c2, B, m->rawA, m->ssA[0], m->ssA[1], m->rawB, c2.coutp, m->ssB, nullptr, nullptr, 0, s);
        m->defer_main = false;
        if (rc != PNVO_OK) return rc;
      }
      if (ds) {
        const Layer &cd = m->convs[li++];
        m->defer_main = defer_tail && !ds_ride && !ds_forked;          // (a downsample conv launched on its own: the same consumer)
        if (ds_ride) {                                                 // rawD / ssD came out of c1's launch
        } else if (ds_forked) {
          HIPCHK(m, hipStreamWaitEvent(s, m->side_join, 0));           // rawD / ssD are complete before the block tail's consumer
        } else if ((rc = run_conv(m, cd, B, cur, nullptr, nullptr, m->rawD, cd.coutp, m->ssD, nullptr, nullptr, 0, s)) != PNVO_OK) {
          m->defer_main = false;
          return rc;
        }
        m->defer_main = false;
      }
      // (the last block's tail rides on the compression conv when that runs on conv_x3_kernel: nobody else reads that block output)
      if (li < m->convs.size() && pnvo_conv_takes_tail(m, m->convs[li], B)) {   // relu(GN2(conv2) + skip): computed by the next conv's stager
        tail.res = ds ? m->rawD : cur;
        tail.res_scale = ds ? m->ssD[0] : nullptr;
        tail.res_shift = ds ? m->ssD[1] : nullptr;
        tail.out = nxt;
        have_tail = true;
        continue;
      }
      {
        Timed t(m, s, "residual", 0.0, 12.0 * B * P * c2.coutp);
        HIPCHK(m, launch_residual(m->rawB, m->ssB[0], m->ssB[1], ds ? m->rawD : cur, ds ? m->ssD[0] : nullptr, ds ? m->ssD[1] : nullptr, B,
                                  P, c2.coutp, nxt, s));
      }
      std::swap(cur, nxt);
      const std::string tn = "layer" + std::to_string(stage) + "." + std::to_string(bi);
      if ((rc = maybe_tap(m, tn.c_str(), cur, (size_t)B * P * c2.coutp, s)) != PNVO_OK) return rc;
    }
  }

  // (a10) compression conv + GroupNorm(1, C)
  const Layer &comp = m->convs[li++];
  if (have_tail) {               // the last block's tail in the compression conv's stager; its output is not materialised
    tail.out = nullptr;
    if ((rc = pnvo_run_conv(m, comp, B, m->rawB, m->ssB[0], m->ssB[1], m->comp_raw, comp.coutp, m->ssC, nullptr, nullptr, 0, s, nullptr, nullptr,
                            nullptr, &tail)) != PNVO_OK)
      return rc;
    have_tail = false;
  } else if ((rc = run_conv(m, comp, B, cur, nullptr, nullptr, m->comp_raw, comp.coutp, m->ssC, nullptr, nullptr, 0, s)) != PNVO_OK) {
    return rc;
  }
  if (m->tap_dst != nullptr && m->tap_name == "compression") {
    HIPCHK(m, launch_apply_ss_relu(m->comp_raw, m->ssC[0], m->ssC[1], B, (long)m->fh * m->fw, m->comp_cp, m->tapbuf, s));
    if ((rc = maybe_tap(m, "compression", m->tapbuf, (size_t)B * m->fh * m->fw * m->comp_cp, s)) != PNVO_OK) return rc;
  }
  // (a11) Flatten + Linear + ReLU, then the output head — per action model in a grouped forward (each on its own handle: its
  // weights, bias rows and split-K scratch; the sample ranges are contiguous)
  if (m->grp_n > 1) {
    bool rows_ok = !c.act_embed;
    for (int k = 0; k < m->grp_n; ++k) rows_ok = rows_ok && fc_rows_usable(m->grp[k], B);
    if (rows_ok) return run_fc_rows(m, m->grp, m->grp_end, m->grp_n, B, m->comp_raw, m->ssC[0], m->ssC[1], nullptr, out, s);
    int start = 0;
    for (int k = 0; k < m->grp_n; ++k) {
      const int Bk = m->grp_end[k] - start;
      pnvo_handle hk = m->grp[k];
      if (k > 0 && (rc = ensure_workspace(hk, Bk)) != PNVO_OK) return fail(m, rc, std::string("grouped forward: ") + pnvo_last_error(hk));
      const size_t crow = (size_t)m->fh * m->fw * m->comp_cp;
      rc = run_fc_head(hk, Bk, m->comp_raw + start * crow, m->ssC[0] + (size_t)start * m->comp_cp, m->ssC[1] + (size_t)start * m->comp_cp,
                       nullptr, out + (size_t)start * c.out_dim, s);
      if (rc != PNVO_OK) return k > 0 ? fail(m, rc, std::string("grouped forward: ") + pnvo_last_error(hk)) : rc;
      start = m->grp_end[k];
    }
    return PNVO_OK;
  }
  return run_fc_head(m, B, m->comp_raw, m->ssC[0], m->ssC[1], actions, out, s);
}

// fc_rows.hip takes the two Linear layers of this handle at B samples (inference handles on float32, default kernels)
bool fc_rows_usable(pnvo_handle m, int B) {
  const long kp = (long)m->fh * m->fw * m->comp_cp;
  return m->opt.fc_rows >= B && B >= 1 && m->fc_rows_w != nullptr && m->train == nullptr && m->precision == 0 && m->opt.conv == 0 &&
         m->comp_cp > 0 && 256 % m->comp_cp == 0 && kp % 4 == 0 && kp / 4 <= 64L * FC_ROWS_MAXV && m->cfg.hidden % 4 == 0 &&
         (m->features_only || m->head_w_plain != nullptr);
}

// ... of up to three handles (a grouped forward: grp[k] serves the samples up to end[k]) in one launch each
int run_fc_rows(pnvo_handle m, pnvo_handle const *grp, const int *end, int ng, int B, const float *comp_raw, const float *sc, const float *sh,
                const int64_t *actions, float *out, hipStream_t s) {
  const pnvo_config &c = m->cfg;
  FcRowsArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = comp_raw;
  a.sc = sc;
  a.sh = sh;
  a.hid = m->features_only ? out : m->hid;
  a.out = out;
  a.actions = c.act_embed ? actions : nullptr;
  a.B = B;
  a.Kp = m->fh * m->fw * m->comp_cp;
  a.cp = m->comp_cp;
  a.hidden = c.hidden;
  a.out_dim = c.out_dim;
  a.ngroups = ng;
  for (int k = 0; k < ng; ++k) {
    a.end[k] = end[k];
    a.w[k] = grp[k]->fc_rows_w;
    a.bias[k] = grp[k]->fc_bias;
    a.head_w[k] = grp[k]->head_w_plain;
    a.head_b[k] = grp[k]->head_bias;
  }
  Timed t(m, s, "conv:" + m->fc.name, 2.0 * B * (double)a.Kp * c.hidden, 4.0 * ng * (double)a.Kp * c.hidden);
  HIPCHK(m, launch_fc_rows(a, !m->features_only, s));
  return PNVO_OK;
}

// The hidden layer and the output head of handle m on B rows of the compression output.
int run_fc_head(pnvo_handle m, int B, const float *comp_raw, const float *sc, const float *sh, const int64_t *actions, float *out, hipStream_t s) {
  const pnvo_config &c = m->cfg;
  int rc = PNVO_OK;
  if (fc_rows_usable(m, B)) {
    pnvo_handle one[1] = {m};
    const int end[1] = {B};
    if ((rc = run_fc_rows(m, one, end, 1, B, comp_raw, sc, sh, actions, out, s)) != PNVO_OK) return rc;
    return m->features_only ? PNVO_OK : maybe_tap(m, "hidden", m->hid, (size_t)B * c.hidden, s);
  }
  // the output head rides on the hidden layer's split-K reduction when there is one (option head_fuse); with a training step attached
  // the head's weight is read where the optimiser keeps it (the flat parameter buffer), the bias from its re-packed copy
  m->head_rode = false;
  m->head_ride_w = m->train != nullptr ? pnvo_train_weight_ptr(m, "output_head.1.weight") : m->head_w_plain;   // (OIHW of a 1x1 conv = [out_dim][hidden])
  m->head_ride_out = (m->opt.head_fuse && !m->features_only && c.out_dim <= 4 && m->head_ride_w != nullptr) ? out : nullptr;
  rc = run_conv(m, m->fc, B, comp_raw, sc, sh, m->hid, c.hidden, nullptr, m->fc_bias, c.act_embed ? actions : nullptr, 1, s);
  m->head_ride_out = nullptr;
  if (rc != PNVO_OK) return rc;
  if ((rc = maybe_tap(m, "hidden", m->hid, (size_t)B * c.hidden, s)) != PNVO_OK) return rc;
  if (m->head_rode) return PNVO_OK;
  if (m->features_only) {                          // pnvo_forward_features: `out` receives the hidden vector
    HIPCHK(m, hipMemcpyAsync(out, m->hid, (size_t)B * c.hidden * sizeof(float), hipMemcpyDeviceToDevice, s));
    return PNVO_OK;
  }
  if ((rc = run_conv(m, m->head, B, m->hid, nullptr, nullptr, out, c.out_dim, nullptr, m->head_bias, nullptr, 0, s)) !=
      PNVO_OK)
    return rc;
  return PNVO_OK;
}
}  // namespace

namespace {
// Can this call run on the RAW stager (frames straight into the stem)?  Else the raw entry materialises the observation pairs
// (pnvo_build_obs_pairs into a workspace of the handle) and takes the ordinary path: same results, the old cost.
bool raw_direct(pnvo_handle m, const float *depth_frames) {
  if (depth_frames == nullptr || !m->mx_ok || m->dense_sticky || m->opt.stem > 1 || m->tap_dst != nullptr) return false;
  if (m->precision == 1) return true;                                // bf16 path: its stem is the mx kernel (PIECES = 1)
  return m->opt.pieces == 2 && m->mx_wpk2 != nullptr && (m->train == nullptr || m->weights_gen == m->weights_gen_at_load);
}

int raw_materialise(pnvo_handle m, const uint8_t *rgb_frames, const float *depth_frames, int B, int32_t *err_flag, hipStream_t s) {
  const pnvo_config &c = m->cfg;
  if (B > m->rawws_cap) {
    for (float *&q : m->rawws) pnvo_free_dev(q);
    const size_t px = (size_t)B * c.height * c.width;
    if (c.n_rgb) HIPCHK(m, hipMalloc((void **)&m->rawws[0], px * 6 * sizeof(float)));
    HIPCHK(m, hipMalloc((void **)&m->rawws[1], px * 2 * sizeof(float)));
    if (c.n_dd) HIPCHK(m, hipMalloc((void **)&m->rawws[2], px * (size_t)c.n_dd * sizeof(float)));
    m->rawws_cap = B;
  }
  HIPCHK(m, launch_frame_pairs(c.n_rgb ? rgb_frames : nullptr, depth_frames, B, c.height, c.width, c.n_dd / 2, m->rawws[0], m->rawws[1],
                               m->rawws[2], err_flag, s));
  return PNVO_OK;
}
}  // namespace

int pnvo_forward_raw(pnvo_handle m, const uint8_t *rgb_frames, const float *depth_frames, const float *tdv, const int64_t *actions, int B,
                     float *out, int32_t *err_flag, void *stream) {
  if (!m) return fail(m, PNVO_ERR_ARG, "null handle");
  if (!m->loaded) return fail(m, PNVO_ERR_STATE, "pnvo_forward_raw before pnvo_load_weights");
  if (B <= 0 || !out) return fail(m, PNVO_ERR_ARG, "bad batch / null output");
  const pnvo_config &c = m->cfg;
  if ((c.n_rgb > 0) != (rgb_frames != nullptr) || ((c.n_depth > 0 || c.n_dd > 0) && depth_frames == nullptr) ||
      (c.n_tdv > 0) != (tdv != nullptr))
    return fail(m, PNVO_ERR_ARG, "sensor frames do not match the model's observation_space");
  if (c.act_embed && !actions) return fail(m, PNVO_ERR_ARG, "act_embed model needs actions");
  if (int rc0 = pnvo_check_inputs(m)) return rc0;
  HIPCHK(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  if (!raw_direct(m, depth_frames)) {
    int rc = raw_materialise(m, rgb_frames, depth_frames, B, err_flag, s);
    if (rc != PNVO_OK) return rc;
    return pnvo_forward(m, c.n_rgb ? m->rawws[0] : nullptr, c.n_depth ? m->rawws[1] : nullptr, c.n_dd ? m->rawws[2] : nullptr, tdv,
                        actions, B, out, stream);
  }
  m->raw_rgb = rgb_frames;
  m->raw_depth = depth_frames;
  m->raw_err = err_flag;
  int rc;
  if (m->precision == 1) {
    pnvo_handle hs[1] = {m};
    float *outs[1] = {out};
    rc = pnvo_forward_bf16(hs, 1, nullptr, nullptr, nullptr, tdv, actions, B, outs, s);
  } else {
    rc = ensure_workspace(m, B);
    // (frames are uint8 / the one-hot is derived in the stager: nothing for the input-contract check to find — no stem event)
    if (rc == PNVO_OK) rc = forward_dispatch(m, nullptr, nullptr, nullptr, tdv, actions, B, out, s);
    m->stem_ev_pending = false;
  }
  m->raw_rgb = nullptr;
  m->raw_depth = nullptr;
  m->raw_err = nullptr;
  return rc;
}

// Can these handles share a grouped forward?  PNVO_OK, or the error pnvo_forward_grouped_raw would return (reason in pnvo_last_error
// of the first handle) — the caller's dispatch between one grouped and several per-model forwards asks here, no exception-driven retry.
int pnvo_grouped_supported(const pnvo_handle *hs, int n) {
  if (!hs || n < 1 || n > 3 || !hs[0]) return fail(nullptr, PNVO_ERR_ARG, "bad handle list");
  pnvo_handle m = hs[0];
  const pnvo_config &c = m->cfg;
  for (int k = 0; k < n; ++k) {
    pnvo_handle h = hs[k];
    if (!h) return fail(m, PNVO_ERR_ARG, "grouped forward: null handle");
    for (int j = 0; j < k; ++j)
      if (hs[j] == h) return fail(m, PNVO_ERR_ARG, "grouped forward: a handle appears twice");
    if (!h->loaded) return fail(m, PNVO_ERR_STATE, "grouped forward before pnvo_load_weights");
    if (std::memcmp(&h->cfg, &c, sizeof(pnvo_config)) != 0 || h->device != m->device)
      return fail(m, PNVO_ERR_ARG, "the models of a grouped forward must share architecture and device");
    if (h->precision != 0 || h->train != nullptr || h->bottleneck || c.act_embed || h->dense_sticky || h->opt.pieces != 2 || !h->mx_ok ||
        h->mx_wpk2 == nullptr || h->tap_dst != nullptr || h->opt.stem > 1 || !h->opt.pool || !h->opt.tail || h->opt.conv > 1)
      return fail(m, PNVO_ERR_STATE, "grouped forward needs float32 inference handles on the default float16-piece kernels (no training step, "
                                     "no tap, no act-embed, options pieces=2 / stem=auto / conv=auto / pool / tail at their defaults)");
    for (size_t li = 1; li < h->convs.size(); ++li)
      if (h->convs[li].groups > 0 && !x3_two_pieces(h, h->convs[li]))
        return fail(m, PNVO_ERR_STATE, "grouped forward: layer " + h->convs[li].name + " of a model is range-guarded off the float16-piece form");
  }
  return PNVO_OK;
}

// One launch chain for the pairs of up to three action models (round 6; the navigation loop's call shape: 8-32 pairs per simulator
// step split over the forward / left / right models, base_trainer_with_vo.py:277-294 — three small forwards of ~58 dependent launches
// each are bound by launch latency, not work).  Pairs are sorted by model: handles[k] serves counts[k] consecutive pairs.  Every
// conv runs on conv_x3_kernel (forced; fine plan for small launches) and the tile stem, which pick a sample's weights, weight scale
// and GroupNorm affine by its model; the hidden layer and head run per model on its rows.  Same arithmetic per pair as a forward
// of its model that selects these kernels (options conv=x3, x3_rows=off, stem_form=tiles): float32-grade equal to the default
// forward of the same pairs, whose kernel choice depends on the batch size.
int pnvo_forward_grouped_raw(const pnvo_handle *handles, const int32_t *counts, int n_models, const uint8_t *rgb_frames,
                             const float *depth_frames, const float *tdv, int B, float *out, int32_t *err_flag, void *stream) {
  if (!handles || !counts || n_models < 1 || n_models > 3 || !handles[0]) return fail(nullptr, PNVO_ERR_ARG, "bad handle list");
  pnvo_handle hs[3] = {nullptr, nullptr, nullptr};
  int cnt[3] = {0, 0, 0}, ng = 0, total = 0;
  for (int k = 0; k < n_models; ++k) {
    if (counts[k] < 0 || (counts[k] > 0 && !handles[k])) return fail(handles[0], PNVO_ERR_ARG, "bad pair count / null handle");
    total += counts[k];
    if (counts[k] > 0) {
      hs[ng] = handles[k];
      cnt[ng++] = counts[k];
    }
  }
  pnvo_handle m = hs[0];
  if (ng == 0 || total != B || B <= 0 || !out) return fail(handles[0], PNVO_ERR_ARG, "pair counts do not add up to the batch / null output");
  if (ng == 1) return pnvo_forward_raw(m, rgb_frames, depth_frames, tdv, nullptr, B, out, err_flag, stream);
  const pnvo_config &c = m->cfg;
  if (int rcs = pnvo_grouped_supported(hs, ng)) return rcs;
  for (int k = 0; k < ng; ++k)
    if (int rc0 = pnvo_check_inputs(hs[k])) return rc0;
  if ((c.n_rgb > 0) != (rgb_frames != nullptr) || ((c.n_depth > 0 || c.n_dd > 0) && depth_frames == nullptr) || (c.n_tdv > 0) != (tdv != nullptr))
    return fail(m, PNVO_ERR_ARG, "sensor frames do not match the models' observation_space");
  if (!raw_direct(m, depth_frames)) return fail(m, PNVO_ERR_STATE, "grouped forward needs the sensor-frame stager of the float16-piece stem");
  HIPCHK(m, hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_workspace(m, B);
  if (rc != PNVO_OK) return rc;
  const PnvoOptions saved = m->opt;
  m->opt.conv = 1;                 // every GroupNorm-ed conv on conv_x3_kernel (its fine plan for small launches)
  m->opt.x3_rows = 0;              // (the row-streaming and resident-weight kernels hold ONE model's weights per workgroup)
  m->opt.x3_persist = 0;
  if (m->opt.gn_fuse == 1) m->opt.gn_fuse = 2;
  m->grp_n = ng;
  int acc = 0;
  for (int k = 0; k < ng; ++k) {
    m->grp[k] = hs[k];
    acc += cnt[k];
    m->grp_end[k] = acc;
  }
  m->raw_rgb = rgb_frames;
  m->raw_depth = depth_frames;
  m->raw_err = err_flag;
  rc = forward_body(m, nullptr, nullptr, nullptr, tdv, nullptr, B, out, s);
  m->stem_ev_pending = false;
  m->raw_rgb = nullptr;
  m->raw_depth = nullptr;
  m->raw_err = nullptr;
  m->grp_n = 0;
  for (int k = 0; k < 3; ++k) m->grp[k] = nullptr;
  m->opt = saved;
  return rc;
}

int pnvo_forward_dual_raw(pnvo_handle ha, pnvo_handle hb, const uint8_t *rgb_frames, const float *depth_frames, const float *tdv, int B,
                          float *out_a, float *out_b, int32_t *err_flag, void *stream) {
  if (!ha || !hb) return fail(ha, PNVO_ERR_ARG, "null handle");
  if (!ha->loaded || !hb->loaded) return fail(ha, PNVO_ERR_STATE, "pnvo_forward_dual_raw before pnvo_load_weights");
  if (B <= 0 || !out_a || !out_b) return fail(ha, PNVO_ERR_ARG, "bad batch / null output");
  if (ha->precision != 1 || hb->precision != 1)
    return fail(ha, PNVO_ERR_STATE, "pnvo_forward_dual_raw runs the bfloat16 path: call pnvo_set_precision(h, 1) on both models");
  if (std::memcmp(&ha->cfg, &hb->cfg, sizeof(pnvo_config)) != 0 || ha->device != hb->device)
    return fail(ha, PNVO_ERR_ARG, "the two models of a dual forward must share architecture and device");
  const pnvo_config &c = ha->cfg;
  if (c.act_embed) return fail(ha, PNVO_ERR_ARG, "pnvo_forward_dual_raw covers the separate-action models");
  if ((c.n_rgb > 0) != (rgb_frames != nullptr) || ((c.n_depth > 0 || c.n_dd > 0) && depth_frames == nullptr) ||
      (c.n_tdv > 0) != (tdv != nullptr))
    return fail(ha, PNVO_ERR_ARG, "sensor frames do not match the model's observation_space");
  HIPCHK(ha, hipSetDevice(ha->device));
  hipStream_t s = (hipStream_t)stream;
  if (!raw_direct(ha, depth_frames)) {
    int rc = raw_materialise(ha, rgb_frames, depth_frames, B, err_flag, s);
    if (rc != PNVO_OK) return rc;
    return pnvo_forward_dual(ha, hb, c.n_rgb ? ha->rawws[0] : nullptr, c.n_depth ? ha->rawws[1] : nullptr,
                             c.n_dd ? ha->rawws[2] : nullptr, tdv, B, out_a, out_b, stream);
  }
  ha->raw_rgb = rgb_frames;
  ha->raw_depth = depth_frames;
  ha->raw_err = err_flag;
  pnvo_handle hs[2] = {ha, hb};
  float *outs[2] = {out_a, out_b};
  const int rc = pnvo_forward_bf16(hs, 2, nullptr, nullptr, nullptr, tdv, nullptr, B, outs, s);
  ha->raw_rgb = nullptr;
  ha->raw_depth = nullptr;
  ha->raw_err = nullptr;
  return rc;
}

int pnvo_forward_features(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv,
                          const int64_t *actions, int B, float *hidden_out, void *stream) {
  if (!m) return fail(m, PNVO_ERR_ARG, "null handle");
  if (!m->loaded) return fail(m, PNVO_ERR_STATE, "pnvo_forward_features before pnvo_load_weights");
  if (B <= 0 || !hidden_out) return fail(m, PNVO_ERR_ARG, "bad batch / null output");
  const pnvo_config &c = m->cfg;
  if ((c.n_rgb > 0) != (rgb != nullptr) || (c.n_depth > 0) != (depth != nullptr) || (c.n_dd > 0) != (dd != nullptr) ||
      (c.n_tdv > 0) != (tdv != nullptr))
    return fail(m, PNVO_ERR_ARG, "observation tensors do not match the model's observation_space");
  if (int rc0 = pnvo_check_inputs(m)) return rc0;
  HIPCHK(m, hipSetDevice(m->device));
  int rc = ensure_workspace(m, B);
  if (rc != PNVO_OK) return rc;
  m->features_only = true;
  rc = forward_body(m, rgb, depth, dd, tdv, actions, B, hidden_out, (hipStream_t)stream);
  bool rerun = false;
  if (rc == PNVO_OK) rc = pnvo_input_fallback(m, (hipStream_t)stream, &rerun);
  if (rc == PNVO_OK && rerun) rc = forward_body(m, rgb, depth, dd, tdv, actions, B, hidden_out, (hipStream_t)stream);
  m->features_only = false;
  return rc;
}

int pnvo_discretize_depth(const float *depth, int64_t n, int64_t in_stride, int bins, float *onehot,
                          int64_t out_stride, int32_t *err_flag, void *stream) {
  if (!depth || !onehot || n < 0 || bins < 1 || bins > 64) return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (n == 0) return PNVO_OK;
  HIPCHK(nullptr, launch_discretize_depth(depth, n, in_stride, bins, onehot, out_stride, err_flag, (hipStream_t)stream));
  return PNVO_OK;
}

size_t pnvo_topdown_workspace_bytes(int N, int H, int W) { return topdown_workspace_bytes(N, H, W); }

int pnvo_topdown_view(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                      const float *consts, int rows_around_center, float *out, int64_t out_fstride,
                      int64_t out_pstride, void *work, void *stream) {
  if (!depth || !out || !consts || !work || N < 0 || H <= 0 || W <= 0)
    return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (N == 0) return PNVO_OK;
  HIPCHK(nullptr, launch_topdown(depth, N, H, W, in_fstride, in_pstride, consts, rows_around_center, out, out_fstride,
                                 out_pstride, work, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_topdown_view_pairs(const float *depth_frames, int n_pairs, int H, int W, const float *consts, int rows_around_center,
                            float *tdv_pairs, void *work, void *stream) {
  if (!depth_frames || !tdv_pairs || !consts || !work || n_pairs < 0 || H <= 0 || W <= 0)
    return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (n_pairs == 0) return PNVO_OK;
  const int64_t hw = (int64_t)H * W;
  HIPCHK(nullptr, launch_topdown(depth_frames, 2 * n_pairs, H, W, hw, 1, consts, rows_around_center, tdv_pairs, 2 * hw, 2, work,
                                 (hipStream_t)stream, 1));
  return PNVO_OK;
}

int pnvo_topdown_view_f64(const float *depth, int N, int H, int W, int64_t in_fstride, int64_t in_pstride,
                          const double *consts, int rows_around_center, float *out, int64_t out_fstride,
                          int64_t out_pstride, void *work, void *stream) {
  if (!depth || !out || !consts || !work || N < 0 || H <= 0 || W <= 0)
    return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (N == 0) return PNVO_OK;
  HIPCHK(nullptr, launch_topdown_f64(depth, N, H, W, in_fstride, in_pstride, consts, rows_around_center, out, out_fstride,
                                     out_pstride, work, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_half_to_float(const uint16_t *src, int64_t n, float *dst, void *stream) {
  if (!src || !dst || n < 0) return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (n == 0) return PNVO_OK;
  HIPCHK(nullptr, launch_half_to_float(src, (long)n, dst, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_dataset_pairs(const uint8_t *prev_rgb, const uint8_t *cur_rgb, const uint16_t *prev_depth, const uint16_t *cur_depth,
                       const float *tdv_frames, const int32_t *src, const int32_t *swap, int N, int M, int H, int W, int bins,
                       const float *edges, float *rgb_pairs, float *depth_pairs, float *dd_pairs, float *tdv_pairs,
                       int32_t *err_flag, void *stream) {
  if (!prev_depth || !cur_depth || !src || !swap || N <= 0 || M < 0 || H <= 0 || W <= 0 || bins < 0 || bins > 64)
    return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (rgb_pairs && (!prev_rgb || !cur_rgb)) return fail(nullptr, PNVO_ERR_ARG, "rgb pairs requested without rgb frames");
  if (M == 0) return PNVO_OK;
  HIPCHK(nullptr, launch_dataset_pairs(prev_rgb, cur_rgb, prev_depth, cur_depth, tdv_frames, src, swap, N, M, H, W, bins, edges,
                                       rgb_pairs, depth_pairs, dd_pairs, tdv_pairs, err_flag, (hipStream_t)stream));
  return PNVO_OK;
}

namespace {
// Persistent copy workers of pnvo_stage_frames: a per-call std::thread spawn costs more than the copy of a small chunk, and the
// gather of 2N simulator frames into pinned staging is the host-side critical path of the batched boundary call (59 MB at 64
// pairs).  Heap-allocated and never destroyed: workers may still be parked on the condition variable at process exit.
struct StagePool {
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  // current job: items [0, n) of array 0 then items [0, n) of array 1 (rgb frames and depth frames of a chunk in ONE wake-up)
  const void *const *src[2] = {nullptr, nullptr};
  char *dst[2] = {nullptr, nullptr};
  size_t bytes[2] = {0, 0};
  int n = 0, narr = 1;
  std::atomic<int> next{0};
  int active = 0;            // workers still inside the current job
  int want = 0;              // workers the current job admits
  unsigned long long gen = 0;
  void run(int nworkers, int narr_, const void *const *s0, size_t b0, void *d0, const void *const *s1, size_t b1, void *d1, int n_) {
    std::unique_lock<std::mutex> lk(mu);
    while ((int)workers.size() < nworkers) {
      const int id = (int)workers.size();
      workers.emplace_back([this, id] { loop(id); });
      workers.back().detach();
    }
    src[0] = s0;
    dst[0] = static_cast<char *>(d0);
    bytes[0] = b0;
    src[1] = s1;
    dst[1] = static_cast<char *>(d1);
    bytes[1] = b1;
    narr = narr_;
    n = n_;
    next.store(0);
    want = nworkers;
    active = nworkers;
    ++gen;
    cv_work.notify_all();
    lk.unlock();
    copy_items();                                   // the caller works too
    lk.lock();
    cv_done.wait(lk, [this] { return active == 0; });
  }
  void copy_items() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n * narr) return;
      const int a = i / n, k = i - a * n;
      std::memcpy(dst[a] + (size_t)k * bytes[a], src[a][k], bytes[a]);
    }
  }
  void loop(int id) {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&] { return gen != seen; });
      seen = gen;
      if (id >= want) continue;                     // this job runs on fewer workers
      lk.unlock();
      copy_items();
      lk.lock();
      if (--active == 0) cv_done.notify_all();
    }
  }
};
StagePool *stage_pool() {
  static StagePool *p = new StagePool();
  return p;
}
std::mutex g_stage_call;                            // one gather at a time (the pool holds one job)
}  // namespace

int pnvo_stage_frames(const void *const *src, int n, size_t bytes_each, void *dst, int threads) {
  if (!src || !dst || n < 0) return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (threads < 1) threads = 1;
  if (threads > 32) threads = 32;
  if (threads > n) threads = n > 0 ? n : 1;
  if (threads == 1 || (size_t)n * bytes_each < ((size_t)1 << 20)) {      // small jobs: a wake-up would dominate
    for (int i = 0; i < n; ++i) std::memcpy(static_cast<char *>(dst) + (size_t)i * bytes_each, src[i], bytes_each);
    return PNVO_OK;
  }
  std::lock_guard<std::mutex> lk(g_stage_call);
  stage_pool()->run(threads - 1, 1, src, bytes_each, dst, nullptr, 0, nullptr, n);   // threads - 1 workers + the caller
  return PNVO_OK;
}

int pnvo_stage_frames2(const void *const *src_a, size_t bytes_a, void *dst_a, const void *const *src_b, size_t bytes_b, void *dst_b, int n,
                       int threads) {
  if (!src_a || !dst_a || !src_b || !dst_b || n < 0) return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if (threads < 1) threads = 1;
  if (threads > 32) threads = 32;
  if (threads > 2 * n) threads = n > 0 ? 2 * n : 1;
  if (threads == 1 || (size_t)n * (bytes_a + bytes_b) < ((size_t)1 << 20)) {
    for (int i = 0; i < n; ++i) std::memcpy(static_cast<char *>(dst_a) + (size_t)i * bytes_a, src_a[i], bytes_a);
    for (int i = 0; i < n; ++i) std::memcpy(static_cast<char *>(dst_b) + (size_t)i * bytes_b, src_b[i], bytes_b);
    return PNVO_OK;
  }
  std::lock_guard<std::mutex> lk(g_stage_call);
  stage_pool()->run(threads - 1, 2, src_a, bytes_a, dst_a, src_b, bytes_b, dst_b, n);
  return PNVO_OK;
}

int pnvo_build_obs_pairs(const uint8_t *rgb_frames, const float *depth_frames, int n, int H, int W, int bins,
                         const float *tdv_consts, int rows_around_center, void *tdv_work, float *rgb_pairs, float *depth_pairs,
                         float *dd_pairs, float *tdv_pairs, int32_t *err_flag, void *stream) {
  if (!depth_frames || !depth_pairs || n < 0 || H <= 0 || W <= 0 || bins < 0 || bins > 64)
    return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if ((bins > 0) != (dd_pairs != nullptr)) return fail(nullptr, PNVO_ERR_ARG, "dd_pairs must be given exactly when bins > 0");
  if (tdv_pairs && (!tdv_consts || !tdv_work)) return fail(nullptr, PNVO_ERR_ARG, "top-down view requested without constants / workspace");
  if (n == 0) return PNVO_OK;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(nullptr, launch_frame_pairs(rgb_frames, depth_frames, n, H, W, bins, rgb_pairs, depth_pairs, dd_pairs, err_flag, s));
  if (tdv_pairs) {
    const int64_t hw = (int64_t)H * W;
    for (int k = 0; k < 2; ++k)        // prev frames then cur frames (base_trainer_with_vo.py:239-249), written into channel k
      HIPCHK(nullptr, launch_topdown(depth_frames + k * hw, n, H, W, 2 * hw, 1, tdv_consts, rows_around_center, tdv_pairs + k, 2 * hw,
                                     2, tdv_work, s));
  }
  return PNVO_OK;
}

int pnvo_ring_assemble(const uint8_t *up_rgb, const float *up_depth, const float *up_tdv, uint8_t *ring_rgb, float *ring_depth,
                       float *ring_tdv, const int32_t *idx, int n, int H, int W, uint8_t *rgb_frames, float *depth_frames, float *tdv_pairs,
                       void *stream) {
  if (!up_depth || !ring_depth || !idx || !depth_frames || n < 0 || H <= 0 || W <= 0) return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  if ((up_rgb != nullptr) != (rgb_frames != nullptr) || (up_rgb != nullptr) != (ring_rgb != nullptr))
    return fail(nullptr, PNVO_ERR_ARG, "rgb buffers must be given together");
  if ((up_tdv != nullptr) != (tdv_pairs != nullptr) || (up_tdv != nullptr) != (ring_tdv != nullptr))
    return fail(nullptr, PNVO_ERR_ARG, "top-down buffers must be given together");
  if (n == 0) return PNVO_OK;
  HIPCHK(nullptr, launch_ring_assemble(up_rgb, up_depth, up_tdv, ring_rgb, ring_depth, ring_tdv, idx, n, H, W, rgb_frames, depth_frames,
                                       tdv_pairs, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_destroy(pnvo_handle m) {
  if (!m) return PNVO_OK;
  (void)hipSetDevice(m->device);
  pnvo_train_free(m);
  pnvo_bf16_free(m);
  pnvo_small_free(m);
  free_workspace(m);
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  if (m->stem_ev) (void)hipEventDestroy(m->stem_ev);
  if (m->keys_stream) {
    (void)hipStreamSynchronize(m->keys_stream);
    (void)hipStreamDestroy(m->keys_stream);
  }
  if (m->keys_free_ev) (void)hipEventDestroy(m->keys_free_ev);
  if (m->keys_ready_ev) (void)hipEventDestroy(m->keys_ready_ev);
  if (m->side_fork) (void)hipEventDestroy(m->side_fork);
  if (m->side_join) (void)hipEventDestroy(m->side_join);
  if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
  if (m->stats_side) (void)hipFree(m->stats_side);
  for (Layer &l : m->convs) {
    free_dev(l.wpk);
    free_dev(reinterpret_cast<float *&>(l.wpk_x3));
    free_dev(reinterpret_cast<float *&>(l.wpk_x2));
    free_dev(l.gamma);
    free_dev(l.beta);
  }
  free_dev(m->fc.wpk);
  free_dev(m->head.wpk);
  free_dev(m->fc_bias);
  free_dev(m->fc_rows_w);
  free_dev(m->head_bias);
  free_dev(m->head_w_plain);
  free_dev(m->stem_sc);
  free_dev(m->stem_sh);
  free_dev(m->stem_wpk16);
  free_dev(reinterpret_cast<float *&>(m->mx_wpk3));
  free_dev(reinterpret_cast<float *&>(m->mx_wpk2));
  if (m->mx_scale2_dev) (void)hipFree(m->mx_scale2_dev);
  free_dev(m->mx_pages);
  free_dev(m->dd_wpk);
  free_dev(m->dd_table);
  free_dev(m->dd_sc);
  free_dev(m->dd_sh);
  if (m->dd_flag) (void)hipHostFree(m->dd_flag);
  if (m->dd_flag_dev) (void)hipFree(m->dd_flag_dev);
  if (m->mx_prof && m->mx_prof_rs) {
    unsigned long long pr[256];
    (void)hipMemcpy(pr, m->mx_prof, 2048, hipMemcpyDeviceToHost);
    for (int w = 0; w < 12; ++w) {
      const unsigned long long *q = pr + 16 * w;
      if (q[5] == 0) continue;
      const double nt_ = (double)q[5];
      std::fprintf(stderr, "[pnvo] stem_rs wave %d (cycles per tile): k-loop with the next patch's staging %.0f  wait others %.0f  "
                   "exchange + epilogue %.0f  (%llu tiles)\n", w, q[0] / nt_, q[1] / nt_, q[2] / nt_, q[5]);
    }
    (void)hipFree(m->mx_prof);
  } else if (m->mx_prof) {
    unsigned long long pr[32];
    (void)hipMemcpy(pr, m->mx_prof, 256, hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; ++w) {
      const double nt_ = (double)(pr[8 * w + 4] ? pr[8 * w + 4] : 1);
      std::fprintf(stderr, "[pnvo] stem_mx wave %d (cycles per tile): staging %.0f  barrier %.0f  k-loop %.0f  reduce+epilogue %.0f  "
                   "(%llu tiles)\n", w, pr[8 * w] / nt_, pr[8 * w + 1] / nt_, pr[8 * w + 2] / nt_, pr[8 * w + 3] / nt_, pr[8 * w + 4]);
    }
    (void)hipFree(m->mx_prof);
  }
  if (m->dd_prof) {
    unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpy(pr, m->dd_prof, 64, hipMemcpyDeviceToHost);
    const double nt_ = (double)(pr[3] ? pr[3] : 1);
    std::fprintf(stderr, "[pnvo] stem_dd phases (cycles per tile, workgroup thread 0): staging %.0f  k-loop %.0f (of which "
                 "row barriers %.0f)  epilogue %.0f  (%llu tiles)\n", (double)pr[0] / nt_, (double)pr[1] / nt_,
                 (double)pr[4] / nt_, (double)pr[2] / nt_, pr[3]);
    (void)hipFree(m->dd_prof);
  }
  free_dev(m->zero_page);
  free_dev(m->kpart);
  for (float *&q : m->rawws) free_dev(q);
  for (auto &r : m->trecs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : m->evpool) (void)hipEventDestroy(e);
  delete m;
  return PNVO_OK;
}

int pnvo_set_tap(pnvo_handle h, const char *name, float *dst, size_t capacity) {
  if (!h) return fail(h, PNVO_ERR_ARG, "null handle");
  h->tap_name = name ? name : "";
  h->tap_dst = name ? dst : nullptr;
  h->tap_cap = capacity;
  return PNVO_OK;
}

int pnvo_tap_shape(pnvo_handle h, const char *name, int B, int64_t shape[4]) {
  if (!h || !name || !shape) return fail(h, PNVO_ERR_ARG, "null argument");
  const pnvo_config &c = h->cfg;
  const std::string n = name;
  auto set = [&](int64_t a, int64_t b, int64_t cc, int64_t d) {
    shape[0] = a;
    shape[1] = b;
    shape[2] = cc;
    shape[3] = d;
    return PNVO_OK;
  };
  if (n == "input") return set(B, c.height, c.width, h->CP);
  if (n == "stem_conv") return set(B, h->Hs, h->Ws, c.baseplanes);
  if (n == "maxpool") return set(B, h->Hp, h->Wp, c.baseplanes);
  if (n == "compression") return set(B, h->fh, h->fw, h->comp_cp);
  if (n == "hidden") return set(B, 1, 1, c.hidden);
  if (n.rfind("layer", 0) == 0 && n.size() == 8) {
    const int li = n[5] - '0';
    int hh = h->Hp, ww = h->Wp;
    for (int k = 1; k < li; ++k) {
      hh = halve(hh);
      ww = halve(ww);
    }
    return set(B, hh, ww, c.baseplanes << (li - 1));
  }
  return fail(h, PNVO_ERR_ARG, "unknown tap '" + n + "'");
}

int pnvo_layer_kernel(pnvo_handle h, const char *name, int B, char *family, size_t cap, double *executed_flops) {
  if (!h || !name || !family || cap == 0 || B <= 0) return fail(h, PNVO_ERR_ARG, "bad argument");
  for (size_t li = 1; li < h->convs.size(); ++li) {
    const Layer &l = h->convs[li];
    if (l.name != name) continue;
    const double alg = 2.0 * B * l.hout * l.wout * (double)l.cout * l.cin * l.k * l.kw;
    if (stem_writes_slots(h) && pnvo_small_usable(h, B)) {      // a phase of the persistent small-batch kernel (fp32 MFMA)
      std::snprintf(family, cap, "smallnet");
      if (executed_flops) {
        const int mb = l.cinp >= 128 ? 1 : l.cinp >= 64 ? 2 : 4, th = mb == 4 ? 8 : 4, tw = mb == 1 ? 4 : 8;
        *executed_flops = 2.0 * B * ((l.hout + th - 1) / th) * ((l.wout + tw - 1) / tw) * (th * tw) * (double)rup(l.cout, 16) * l.cinp * l.k * l.kw;
      }
      return PNVO_OK;
    }
    ConvX3Args xa;
    int mw, nw;
    size_t ldsb;
    if (l.k == 1 && l.stride == 2 && li >= 3 && h->convs[li - 2].stride == 2 && pnvo_conv_takes_ds(h, h->convs[li - 2], l, B) &&
        x3_args(h, h->convs[li - 2], B, xa, &mw, &nw, &ldsb)) {           // a downsample conv riding on its block's first conv
      std::snprintf(family, cap, "x2-rides");
      if (executed_flops) *executed_flops = 3.0 * 2.0 * (double)B * xa.tiles_r * xa.tiles_c * xa.MT * 32.0 * l.coutp * (double)l.cinp;
      return PNVO_OK;
    }
    if (x3_layer(h, l) && l.groups > 0 && x3_args(h, l, B, xa, &mw, &nw, &ldsb)) {
      std::snprintf(family, cap, xa.np == 2 ? "x2" : "x3");
      // tiles x M-tiles x 32 pixels x padded outputs x K, six bf16 (x3) or three float16 (x2) MFMA terms per float32 product
      if (executed_flops)
        *executed_flops = (xa.np == 2 ? 3.0 : 6.0) * 2.0 * (double)B * xa.tiles_r * xa.tiles_c * xa.MT * 32.0 * l.coutp * (double)l.cinp * l.k * l.kw;
      return PNVO_OK;
    }
    std::snprintf(family, cap, "%s", layer_on_lds(h, l, nullptr) ? "fp32-lds" : "fp32-generic");
    if (executed_flops) *executed_flops = alg;
    return PNVO_OK;
  }
  return fail(h, PNVO_ERR_ARG, std::string("no residual-stage conv named '") + name + "'");
}

int pnvo_timing_mode(pnvo_handle h, int mode) {
  if (!h) return fail(h, PNVO_ERR_ARG, "null handle");
  h->timing = mode ? 1 : 0;
  return PNVO_OK;
}

int pnvo_timing_read(pnvo_handle h, pnvo_kernel_time *entries, int cap, int *n_out) {
  if (!h || !n_out) return fail(h, PNVO_ERR_ARG, "null argument");
  for (auto &r : h->trecs) {
    HIPCHK(h, hipEventSynchronize(r.b));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, r.a, r.b));
    h->tentries[r.entry].total_ms += ms;
    h->evpool.push_back(r.a);
    h->evpool.push_back(r.b);
  }
  h->trecs.clear();
  const int n = (int)h->tentries.size();
  *n_out = n;
  for (int k = 0; k < n && k < cap && entries; ++k) entries[k] = h->tentries[k];
  h->tentries.clear();
  h->tindex.clear();
  return PNVO_OK;
}

size_t pnvo_packed_conv_floats(int cout, int cin, int kh, int kw) { return packed_conv_floats(cout, cin, kh, kw); }

int pnvo_pack_conv_weight(const float *oihw, int cout, int cin, int kh, int kw, float *out) {
  if (!oihw || !out || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) return fail(nullptr, PNVO_ERR_ARG, "bad argument");
  pack_conv_weight(oihw, cout, cin, kh, kw, out);
  return PNVO_OK;
}

}  // extern "C"
