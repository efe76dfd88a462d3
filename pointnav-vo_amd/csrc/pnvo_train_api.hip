// pnvo_train_api.hip — C ABI of the VO training step (BASELINE config 4): forward in train mode with
// saved activations, backward through the whole network, and the optimiser/loss helpers.  Replaces, for one action
// model, the body of the reference's training iteration
//   optimizer.zero_grad(); out = vo_model(batch_pairs); loss = sum_d mse_d; loss.backward(); optimizer.step()
// (/root/reference/pointnav_vo/vo/engine/vo_cnn_regression_geo_invariance_engine.py:855-901, :586; loss
//  vo_cnn_engine.py:135-198; Adam :122-133).  The flat parameter / gradient buffers are caller-owned device memory in
// the reference's state_dict parameter order, so the data-parallel all-reduce is ONE collective on one buffer.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>

#include "pnvo_model.h"

using namespace pnvo;

namespace {

struct PackMap {
  float *dst;
  int *map;
  long n;
};

struct ConvSave {
  float *raw = nullptr, *ss[2] = {nullptr, nullptr}, *mu = nullptr, *rstd = nullptr;
};

struct TocEnt {
  size_t off;
  std::vector<int64_t> shape;
  size_t numel;
};

struct TrainState {
  float *params = nullptr, *grads = nullptr;
  size_t n = 0;
  std::map<std::string, TocEnt> toc;
  std::vector<PackMap> maps;
  std::vector<float *> dgrad_w;     // per conv of m->convs (nullptr for the stem: no input gradient)
  std::vector<std::array<float *, 4>> dgrad_wph;   // stride-2 3x3 convs: one sub-kernel per output parity phase (ph*2 + pw)
  std::vector<unsigned short *> dgrad_x3;          // 3x3 stride-1 convs: three-piece operand of the backward-data conv (conv_x3.hip)
  std::vector<unsigned long long> dgrad_x3_gen;    //   ... built from the flat parameters when m->weights_gen moved
  std::vector<unsigned short *> dgrad_x2;          //   ... and its two-piece float16 form (option train_pieces = 2)
  std::vector<unsigned long long> dgrad_x2_gen;
  unsigned *dmax = nullptr;                        // per conv: float bits of max |dRaw| of the backward in flight (gn_bwd_apply)
  float *fc_t = nullptr, *head_t = nullptr;
  int *d_ref_of_new = nullptr, *d_tensor_of_new = nullptr, *d_ciperm = nullptr;
  GatherSeg *d_segs = nullptr;      // all re-pack maps as one segment table (pnvo_train_refresh)
  long seg_total = 0;
  int nseg = 0;
  int *d_mxmaps = nullptr;          // mx stem: [slot_ref 32 | slot_new 32 | xslot 4]
  int *d_ddmaps = nullptr;          // one-hot stem: [dense_ref 12 | dense_new 12 | dd_ref 2*bins | dd_new 2*bins]
  int dd_nd = 0;
  size_t stem_w_off = 0;            // offset of the OIHW stem weight in the flat parameter buffer
  // saved activations (sized for capB)
  int capB = 0;
  int lastB = 0;
  std::vector<ConvSave> cs;
  std::vector<float *> y;           // y[0] = pooled stem output, y[k] = output of residual block k
  unsigned char *pool_idx = nullptr;
  float *hid = nullptr;
  const float *src[4] = {nullptr, nullptr, nullptr, nullptr};   // observation tensors of the last forward
  // scratch
  float *dYa = nullptr, *dYb = nullptr, *G = nullptr, *dRaw = nullptr, *dA = nullptr, *dStem = nullptr;
  float *dout8 = nullptr, *dh = nullptr, *gh = nullptr, *dz = nullptr;
  float *gn_part = nullptr, *gn_coef = nullptr, *wg_partial = nullptr;
  size_t wg_partial_floats = 0;
  double *mom_part = nullptr;
  // dropout (pnvo_train_set_dropout): p, seed, forward counter; dropped activations of the last forward
  float drop_p = 0.f;
  uint64_t drop_seed = 0, drop_step = 0;
  float *zdrop = nullptr, *hdrop = nullptr, *dmask = nullptr;
  // action-embedding variants: the Linear's 32 embedding columns as per-sample bias rows (train_kernels.hip embed_*)
  const long long *actions = nullptr;   // device [B], set by pnvo_train_set_actions for the next forward/backward
  float *egath = nullptr, *efeat = nullptr, *biasB = nullptr, *dfeat = nullptr;
  int64_t *iota = nullptr;
  int *embed_err = nullptr;             // host-mapped flag: action outside the embedding table
  size_t w1_off = 0, b1_off = 0, emb_off = 0;
  long w1_pitch = 0;
  // gradient-ready hook (pnvo_train_set_grad_hook): the backward reports flat ranges [first, first + count) whose gradients
  // are final while the rest of it is still being enqueued; bucket_first = lower bounds of the ranges, latest layers first
  pnvo_grad_ready_fn hook = nullptr;
  void *hook_user = nullptr;
  std::vector<size_t> bucket_first;     // e.g. {offset of layer4's first parameter, offset of layer2's, 0}
  std::vector<char> bucket_done;        // per backward: which ranges have been reported (the end of the backward reports the rest)
  // two-piece float16 operands of the training forward's convs: per conv weight {scale, 1/scale}, recomputed from the flat
  // parameters by ONE launch per pnvo_train_refresh (conv_x3.hip conv_x2_scale_kernel)
  std::map<std::string, int> x2_index;  // conv weight name -> row of x2_scale
  long *x2_seg = nullptr;               // device [rows][2]: flat offset, element count
  float *x2_scale = nullptr;            // device [rows][2]
  // GroupNorm bounds behind Layer::in_bound (the range guard of the float16 pieces), tracked while gamma / beta move: one small
  // launch per pnvo_train_refresh writes max_c (|gamma_c| sqrt(N) + |beta_c|) per conv into host-mapped memory; the next training
  // forward chains them into in_bound without a synchronisation (a value may lag one optimiser step: a step moves gamma by <= lr)
  struct GnbSeg { long goff, boff; int c; float rootn; };
  GnbSeg *gnb_seg = nullptr;            // device [convs]
  float *gnb_host = nullptr;            // host-mapped [3][convs]: the bounds of the last three refreshes (ring); < 0: not computed yet
  hipEvent_t gnb_ev[3] = {nullptr, nullptr, nullptr};   // behind the gn_bound_kernel of each ring entry
  long gnb_refreshes = 0;               // refreshes issued so far: entry (r % 3) holds refresh r
  int gnb_n = 0;
};

TrainState *TS(pnvo_handle m) { return reinterpret_cast<TrainState *>(m->train); }

int dmalloc(pnvo_handle m, void **p, size_t bytes) {
  HIPCHK(m, hipMalloc(p, bytes ? bytes : 16));
  return PNVO_OK;
}
template <class T>
void dfree(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// float "index tensor" of a parameter: value = flat offset + 1 (exact in fp32 below 2^24 elements)
std::vector<float> index_tensor(const TocEnt &e) {
  std::vector<float> v(e.numel);
  for (size_t i = 0; i < e.numel; ++i) v[i] = (float)(e.off + i + 1);
  return v;
}

int add_map(pnvo_handle m, TrainState *t, float *dst, const std::vector<float> &packed_idx) {
  std::vector<int> mp(packed_idx.size());
  for (size_t i = 0; i < mp.size(); ++i) mp[i] = (int)packed_idx[i];
  PackMap pm;
  pm.dst = dst;
  pm.n = (long)mp.size();
  pm.map = nullptr;
  int rc = dmalloc(m, (void **)&pm.map, mp.size() * sizeof(int));
  if (rc != PNVO_OK) return rc;
  HIPCHK(m, hipMemcpy(pm.map, mp.data(), mp.size() * sizeof(int), hipMemcpyHostToDevice));
  t->maps.push_back(pm);
  return PNVO_OK;
}

const TocEnt *need(pnvo_handle m, TrainState *t, const std::string &name, int *rc) {
  auto it = t->toc.find(name);
  if (it == t->toc.end()) {
    *rc = pnvo_fail(m, PNVO_ERR_WEIGHTS, "parameter table is missing '" + name + "'");
    return nullptr;
  }
  return &it->second;
}

// OIHW [cout][cin][K][K] -> backward-data weights [cin][cout][K][K], taps flipped
std::vector<float> transpose_flip(const std::vector<float> &w, int cout, int cin, int kh, int kw) {
  std::vector<float> o((size_t)cin * cout * kh * kw);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int a = 0; a < kh; ++a)
        for (int b = 0; b < kw; ++b)
          o[(((size_t)ci * cout + co) * kh + a) * kw + b] = w[(((size_t)co * cin + ci) * kh + (kh - 1 - a)) * kw + (kw - 1 - b)];
  return o;
}

int build_maps(pnvo_handle m, TrainState *t) {
  int rc = PNVO_OK;
  const pnvo_config &c = m->cfg;
  t->dgrad_w.assign(m->convs.size(), nullptr);
  t->dgrad_wph.assign(m->convs.size(), std::array<float *, 4>{nullptr, nullptr, nullptr, nullptr});
  t->dgrad_x3.assign(m->convs.size(), nullptr);
  t->dgrad_x3_gen.assign(m->convs.size(), 0);
  t->dgrad_x2.assign(m->convs.size(), nullptr);
  t->dgrad_x2_gen.assign(m->convs.size(), 0);
  for (size_t li = 0; li < m->convs.size(); ++li) {
    Layer &l = m->convs[li];
    const TocEnt *w = need(m, t, l.name + ".weight", &rc);
    if (!w) return rc;
    const std::vector<float> idx = index_tensor(*w);
    const int T = l.k * l.kw;
    std::vector<float> pk;
    if (li == 0) {   // stem: channel order of the fused gather (pnvo_load_weights does the same permutation)
      std::vector<float> wp((size_t)l.cout * m->CP * T, 0.f);
      for (int nc = 0; nc < m->CP; ++nc) {
        const int r = m->stem_ref_of_new[nc];
        if (r < 0) continue;
        for (int o = 0; o < l.cout; ++o)
          std::memcpy(&wp[((size_t)o * m->CP + nc) * T], &idx[((size_t)o * l.cin + r) * T], sizeof(float) * T);
      }
      pnvo_pack_conv_weight_cinp(wp.data(), l.cout, m->CP, m->CP, l.k, l.kw, pk);
      if ((rc = add_map(m, t, l.wpk, pk)) != PNVO_OK) return rc;
      std::vector<float> pk16((size_t)49 * m->CPL * l.cout);
      pack_stem_weight(wp.data(), l.cout, m->CP, m->CPL, pk16.data());
      if ((rc = add_map(m, t, m->stem_wpk16, pk16)) != PNVO_OK) return rc;
    } else {
      pnvo_pack_conv_weight_cinp(idx.data(), l.cout, l.cin, l.cinp, l.k, l.kw, pk);
      if ((rc = add_map(m, t, l.wpk, pk)) != PNVO_OK) return rc;
      // backward-data operand: "conv" with CIN' = coutp (gradient channels), COUT' = cin
      const std::vector<float> wt = transpose_flip(idx, l.cout, l.cin, l.k, l.kw);
      std::vector<float> pkt;
      pnvo_pack_conv_weight_cinp(wt.data(), l.cin, l.cout, l.coutp, l.k, l.kw, pkt);
      if ((rc = dmalloc(m, (void **)&t->dgrad_w[li], pkt.size() * sizeof(float))) != PNVO_OK) return rc;
      if ((rc = add_map(m, t, t->dgrad_w[li], pkt)) != PNVO_OK) return rc;
      if (l.stride == 2 && l.k == 3 && l.kw == 3 && l.pad == 1) {
        // dX[2i+ph] = sum over the taps whose source row (2i+ph-1+kh)/2 is an integer: kh = 1 (ph = 0) or kh = 0, 2
        // (ph = 1), reading dY rows i, or i and i+1 — a 1- or 2-tap stride-1 conv per parity phase, no masked taps
        for (int ph = 0; ph < 2; ++ph)
          for (int pw = 0; pw < 2; ++pw) {
            const int th[2] = {ph ? 0 : 1, 2}, tw[2] = {pw ? 0 : 1, 2};
            const int nh = ph ? 2 : 1, nw = pw ? 2 : 1;
            std::vector<float> sub((size_t)l.cin * l.cout * nh * nw);
            for (int ci = 0; ci < l.cin; ++ci)
              for (int co = 0; co < l.cout; ++co)
                for (int a = 0; a < nh; ++a)
                  for (int b = 0; b < nw; ++b)
                    sub[(((size_t)ci * l.cout + co) * nh + a) * nw + b] = wt[(((size_t)ci * l.cout + co) * 3 + th[a]) * 3 + tw[b]];
            std::vector<float> pks;
            pnvo_pack_conv_weight_cinp(sub.data(), l.cin, l.cout, l.coutp, nh, nw, pks);
            float *&dst = t->dgrad_wph[li][ph * 2 + pw];
            if ((rc = dmalloc(m, (void **)&dst, pks.size() * sizeof(float))) != PNVO_OK) return rc;
            if ((rc = add_map(m, t, dst, pks)) != PNVO_OK) return rc;
          }
      }
    }
    const TocEnt *g = need(m, t, l.gn + ".weight", &rc);
    if (!g) return rc;
    const TocEnt *b = need(m, t, l.gn + ".bias", &rc);
    if (!b) return rc;
    if ((rc = add_map(m, t, l.gamma, index_tensor(*g))) != PNVO_OK) return rc;
    if ((rc = add_map(m, t, l.beta, index_tensor(*b))) != PNVO_OK) return rc;
  }
  // linear layers
  const int T = m->fh * m->fw, flat = m->comp_c * T;
  const TocEnt *w1 = need(m, t, m->fc.name + ".weight", &rc);
  if (!w1) return rc;
  const TocEnt *b1 = need(m, t, m->fc.name + ".bias", &rc);
  if (!b1) return rc;
  const TocEnt *w2 = need(m, t, "output_head.1.weight", &rc);
  if (!w2) return rc;
  const TocEnt *b2 = need(m, t, "output_head.1.bias", &rc);
  if (!b2) return rc;
  const int fc_in = flat + (c.act_embed ? 32 : 0);
  if ((int)w1->shape[1] != fc_in) return pnvo_fail(m, PNVO_ERR_ARG, m->fc.name + ".weight has the wrong number of columns");
  t->w1_off = w1->off;
  t->w1_pitch = fc_in;
  t->b1_off = b1->off;
  if (c.act_embed) {
    const TocEnt *emb = need(m, t, "action_embedding.weight", &rc);
    if (!emb) return rc;
    if (emb->shape.size() != 2 || emb->shape[0] != c.n_acts + 1 || emb->shape[1] != 32)
      return pnvo_fail(m, PNVO_ERR_ARG, "action_embedding.weight must be [n_acts + 1, 32]");
    t->emb_off = emb->off;
    if (!t->embed_err) HIPCHK(m, hipHostMalloc((void **)&t->embed_err, sizeof(int), hipHostMallocMapped));
    *t->embed_err = 0;
  }
  {
    const std::vector<float> i1full = index_tensor(*w1), i2 = index_tensor(*w2);
    std::vector<float> i1((size_t)c.hidden * flat);          // the visual columns
    for (int o = 0; o < c.hidden; ++o)
      std::memcpy(&i1[(size_t)o * flat], &i1full[(size_t)o * fc_in], sizeof(float) * (size_t)flat);
    std::vector<float> pk;
    pnvo_pack_conv_weight_cinp(i1.data(), c.hidden, m->comp_c, m->comp_cp, m->fh, m->fw, pk);
    if ((rc = add_map(m, t, m->fc.wpk, pk)) != PNVO_OK) return rc;
    if (!c.act_embed)        // act-embed: the per-action bias rows are rebuilt by refresh_embed_bias
      if ((rc = add_map(m, t, m->fc_bias, index_tensor(*b1))) != PNVO_OK) return rc;
    pnvo_pack_conv_weight_cinp(i2.data(), c.out_dim, c.hidden, c.hidden, 1, 1, pk);
    if ((rc = add_map(m, t, m->head.wpk, pk)) != PNVO_OK) return rc;
    if ((rc = add_map(m, t, m->head_bias, index_tensor(*b2))) != PNVO_OK) return rc;
    // dz = g_h . W1 as a 1x1 conv: CIN' = hidden, COUT' = T*comp_cp, W'[tap*comp_cp + ch][o] = W1[o][ch*T + tap]
    std::vector<float> wt((size_t)T * m->comp_cp * c.hidden, 0.f);
    for (int o = 0; o < c.hidden; ++o)
      for (int ch = 0; ch < m->comp_c; ++ch)
        for (int tap = 0; tap < T; ++tap)
          wt[((size_t)tap * m->comp_cp + ch) * c.hidden + o] = i1[(size_t)o * flat + (size_t)ch * T + tap];
    pnvo_pack_conv_weight_cinp(wt.data(), T * m->comp_cp, c.hidden, c.hidden, 1, 1, pk);
    if ((rc = dmalloc(m, (void **)&t->fc_t, pk.size() * sizeof(float))) != PNVO_OK) return rc;
    if ((rc = add_map(m, t, t->fc_t, pk)) != PNVO_OK) return rc;
    // dh = dOut . W2: CIN' = 8 (out_dim padded), COUT' = hidden, W'[o][d] = W2[d][o]
    std::vector<float> ht((size_t)c.hidden * c.out_dim);
    for (int o = 0; o < c.hidden; ++o)
      for (int d = 0; d < c.out_dim; ++d) ht[(size_t)o * c.out_dim + d] = i2[(size_t)d * c.hidden + o];
    pnvo_pack_conv_weight_cinp(ht.data(), c.hidden, c.out_dim, rup(c.out_dim, 8), 1, 1, pk);
    if ((rc = dmalloc(m, (void **)&t->head_t, pk.size() * sizeof(float))) != PNVO_OK) return rc;
    if ((rc = add_map(m, t, t->head_t, pk)) != PNVO_OK) return rc;
  }
  // stem channel tables on device
  std::vector<int> ron(m->CPL, -1), ton(m->CPL, -1);
  for (int k = 0; k < m->CP; ++k) {
    ron[k] = m->stem_ref_of_new[k];
    ton[k] = m->stem_tensor_of_new[k];
  }
  if ((rc = dmalloc(m, (void **)&t->d_ref_of_new, m->CPL * sizeof(int))) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->d_tensor_of_new, m->CPL * sizeof(int))) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->d_ciperm, 32 * sizeof(int))) != PNVO_OK) return rc;
  std::vector<int> perm(32, -1);
  for (int k = 0; k < m->CP && k < 32; ++k) perm[k] = m->stem_ref_of_new[k];
  HIPCHK(m, hipMemcpy(t->d_ref_of_new, ron.data(), m->CPL * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(m, hipMemcpy(t->d_tensor_of_new, ton.data(), m->CPL * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(m, hipMemcpy(t->d_ciperm, perm.data(), 32 * sizeof(int), hipMemcpyHostToDevice));
  m->train_mx = false;
  if (m->mx_ok && m->convs[0].cout == 32 && m->mx_slot_ref.size() == 32) {   // stem on the bf16 matrix cores: device-side repack
    std::vector<int> mp(68, -1);
    for (int k = 0; k < 32; ++k) {
      mp[k] = m->mx_slot_ref[k];
      mp[32 + k] = m->mx_slot_new[k];
    }
    for (int x = 0; x < 4; ++x) mp[64 + x] = m->mx_xslot[x];
    if ((rc = dmalloc(m, (void **)&t->d_mxmaps, mp.size() * sizeof(int))) != PNVO_OK) return rc;
    HIPCHK(m, hipMemcpy(t->d_mxmaps, mp.data(), mp.size() * sizeof(int), hipMemcpyHostToDevice));
    auto it = t->toc.find(m->convs[0].name + ".weight");
    if (it == t->toc.end()) return pnvo_fail(m, PNVO_ERR_WEIGHTS, "stem weight missing from the parameter table");
    t->stem_w_off = it->second.off;
    m->train_mx = true;
  }
  if (m->dd_ok) {                  // the one-hot-aware stem's operands are rebuilt on the device after every step
    const int bins = m->dd_bins;
    std::vector<int> mp(24 + 4 * bins, -1);
    auto find_new = [&](int tensor, int ch) {
      for (int nc = 0; nc < m->CP; ++nc)
        if (m->stem_tensor_of_new[nc] == tensor && m->stem_ch_of_new[nc] == ch && m->stem_ref_of_new[nc] >= 0) return nc;
      return -1;
    };
    int nd = 0;
    for (int d = 0; d < 12; ++d) {
      if (m->dd_dense_tensor[d] < 0) continue;
      const int nc = find_new(m->dd_dense_tensor[d], m->dd_dense_ch[d]);
      mp[d] = m->stem_ref_of_new[nc];
      mp[12 + d] = nc;
      nd = d + 1;
    }
    for (int k = 0; k < 2 * bins; ++k) {
      const int nc = find_new(2, k);
      mp[24 + k] = m->stem_ref_of_new[nc];
      mp[24 + 2 * bins + k] = nc;
    }
    t->dd_nd = nd;
    if ((rc = dmalloc(m, (void **)&t->d_ddmaps, mp.size() * sizeof(int))) != PNVO_OK) return rc;
    HIPCHK(m, hipMemcpy(t->d_ddmaps, mp.data(), mp.size() * sizeof(int), hipMemcpyHostToDevice));
    auto it = t->toc.find(m->convs[0].name + ".weight");
    if (it == t->toc.end()) return pnvo_fail(m, PNVO_ERR_WEIGHTS, "stem weight missing from the parameter table");
    t->stem_w_off = it->second.off;
  }
  return PNVO_OK;
}

// Rebuild the stem's operands from the flat parameters and the current whitening tables: the three-piece bf16 B operand of
// stem_mx.hip when the model runs it (32 stem outputs), else the one-hot-aware stem's table / dense weights.
int refresh_stem_dd(pnvo_handle m, TrainState *t, hipStream_t s) {
  if (m->train_mx) {
    HIPCHK(m, launch_stem_mx_repack(t->params + t->stem_w_off, m->convs[0].cin, m->stem_sc, m->stem_sh, t->d_mxmaps, t->d_mxmaps + 32,
                                    t->d_mxmaps + 64, m->mx_wpk3, s));
    if (m->opt.train_pieces == 2 && m->mx_wpk2 != nullptr) {      // the float16 operand of the training forward, with its own scale
      if (!m->mx_scale2_dev) {                    // {scale, 1/scale, integer maximum of the folded weights (zero between calls)}
        HIPCHK(m, hipMalloc((void **)&m->mx_scale2_dev, 4 * sizeof(float)));
        HIPCHK(m, hipMemset(m->mx_scale2_dev, 0, 4 * sizeof(float)));
      }
      HIPCHK(m, launch_stem_mx_repack_h(t->params + t->stem_w_off, m->convs[0].cin, m->stem_sc, m->stem_sh, t->d_mxmaps, t->d_mxmaps + 32,
                                        t->d_mxmaps + 64, m->mx_scale2_dev, m->mx_wpk2, s));
      m->mx_wpk2_dev = true;
    }
    if (!(m->dd_ok && m->opt.stem == 2)) return PNVO_OK;       // (option stem=dd on a model that trains on the mx stem: keep the
  }                                                             //  one-hot stem's operands current as well)
  if (!m->dd_ok) return PNVO_OK;
  const int bins = m->dd_bins;
  HIPCHK(m, launch_stem_dd_repack(t->params + t->stem_w_off, m->convs[0].cin, m->stem_sc, m->stem_sh, t->d_ddmaps,
                                  t->d_ddmaps + 12, t->dd_nd, t->d_ddmaps + 24, t->d_ddmaps + 24 + 2 * bins, bins,
                                  m->dd_table, m->dd_wpk, m->dd_sc, m->dd_sh, s));
  return PNVO_OK;
}

void free_train_ws(TrainState *t) {
  for (auto &c : t->cs) {
    dfree(c.raw);
    dfree(c.ss[0]);
    dfree(c.ss[1]);
    dfree(c.mu);
    dfree(c.rstd);
  }
  t->cs.clear();
  for (auto &p : t->y) dfree(p);
  t->y.clear();
  dfree(t->pool_idx);
  dfree(t->hid);
  dfree(t->dYa);
  dfree(t->dYb);
  dfree(t->G);
  dfree(t->dRaw);
  dfree(t->dA);
  dfree(t->dStem);
  dfree(t->dout8);
  dfree(t->dh);
  dfree(t->gh);
  dfree(t->dz);
  dfree(t->gn_part);
  dfree(t->gn_coef);
  dfree(t->wg_partial);
  dfree(t->mom_part);
  dfree(t->zdrop);
  dfree(t->hdrop);
  dfree(t->dmask);
  dfree(t->egath);
  dfree(t->efeat);
  dfree(t->biasB);
  dfree(t->dfeat);
  dfree(t->iota);
  t->capB = 0;
}

// weight-gradient launch descriptor of conv `l` for batch B (input geometry = the forward conv's)
thread_local int g_wgrad_x3 = 1;          // option wgrad3 of the handle whose call is in flight (set at the entry points)

WgradArgs wgrad_args(const Layer &l, int B, int cin_kernel, int dyc, int mode = 0) {
  WgradArgs a;
  std::memset(&a, 0, sizeof(a));
  a.use_x3 = g_wgrad_x3;
  a.mode = mode;                 // wgrad_plan picks the kernel by mode
  a.B = B;
  a.H = l.hin;
  a.W = l.win;
  a.CIN = cin_kernel;
  a.Ho = l.hout;
  a.Wo = l.wout;
  a.COUT = l.cout;
  a.DYC = dyc;
  a.KH = l.k;
  a.KW = l.kw;
  a.stride = l.stride;
  a.pad = l.pad;
  wgrad_plan(a);
  return a;
}

int ensure_train_ws(pnvo_handle m, TrainState *t, int B) {
  g_wgrad_x3 = m->opt.wgrad3;
  if (B <= t->capB) return PNVO_OK;
  free_train_ws(t);
  int rc = pnvo_ensure_workspace(m, B);    // stats buffer etc. of the inference path are reused
  if (rc != PNVO_OK) return rc;
  const pnvo_config &c = m->cfg;
  t->cs.resize(m->convs.size());
  size_t wgmax = 0;
  for (size_t li = 0; li < m->convs.size(); ++li) {
    const Layer &l = m->convs[li];
    ConvSave &s = t->cs[li];
    const size_t n = (size_t)B * l.hout * l.wout * l.coutp;
    if ((rc = dmalloc(m, (void **)&s.raw, n * 4)) != PNVO_OK) return rc;
    for (int k = 0; k < 2; ++k) {
      if ((rc = dmalloc(m, (void **)&s.ss[k], (size_t)B * l.coutp * 4)) != PNVO_OK) return rc;
      HIPCHK(m, hipMemset(s.ss[k], 0, (size_t)B * l.coutp * 4));
    }
    if ((rc = dmalloc(m, (void **)&s.mu, (size_t)B * l.groups * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&s.rstd, (size_t)B * l.groups * 4)) != PNVO_OK) return rc;
    WgradArgs a = wgrad_args(l, B, li == 0 ? 32 : l.cin, l.coutp, li == 0 ? 2 : 0);
    wgmax = std::max(wgmax, wgrad_partial_floats(a));
  }
  {
    WgradArgs a = wgrad_args(m->fc, B, m->comp_c, c.hidden);
    a.CIN = m->comp_c;
    wgmax = std::max(wgmax, wgrad_partial_floats(a));
    WgradArgs h = wgrad_args(m->head, B, c.hidden, 8);
    wgmax = std::max(wgmax, wgrad_partial_floats(h));
  }
  if (m->train_mx) {                 // stem weight gradient on the bf16 matrix cores
    WgradStemMXArgs a;
    std::memset(&a, 0, sizeof(a));
    a.B = B;
    a.Ho = m->Hs;
    a.Wo = m->Ws;
    wgrad_stem_mx_plan(a);
    wgmax = std::max(wgmax, wgrad_stem_mx_scratch_floats(a));
  }
  t->wg_partial_floats = wgmax;
  if ((rc = dmalloc(m, (void **)&t->wg_partial, wgmax * 4)) != PNVO_OK) return rc;
  // block outputs: y[0] = pooled stem output, y[k] = output of residual block k (plan order)
  size_t act = (size_t)B * m->Hp * m->Wp * c.baseplanes;
  const size_t stem = (size_t)B * m->Hs * m->Ws * c.baseplanes;
  {
    const int K = m->bottleneck ? 3 : 2;
    const int nb = m->nblocks[0] + m->nblocks[1] + m->nblocks[2] + m->nblocks[3];
    t->y.assign((size_t)nb + 1, nullptr);
    if ((rc = dmalloc(m, (void **)&t->y[0], act * 4)) != PNVO_OK) return rc;
    size_t li = 1;
    for (int k = 1; k <= nb; ++k) {
      const Layer &last = m->convs[li + K - 1];
      const size_t n = (size_t)B * last.hout * last.wout * last.coutp;
      if ((rc = dmalloc(m, (void **)&t->y[k], n * 4)) != PNVO_OK) return rc;
      li += K;
      if (li < m->convs.size() && m->convs[li].name.find("downsample") != std::string::npos) ++li;
    }
    for (size_t k = 1; k < m->convs.size(); ++k) {      // scratch gradients are as large as the largest activation
      const Layer &l = m->convs[k];
      act = std::max(act, (size_t)B * l.hout * l.wout * l.coutp);
      act = std::max(act, (size_t)B * l.hin * l.win * l.cinp);
    }
  }
  if ((rc = dmalloc(m, (void **)&t->pool_idx, act)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->hid, (size_t)B * c.hidden * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dYa, act * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dYb, act * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->G, act * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dRaw, act * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dA, act * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dStem, stem * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dout8, (size_t)B * 8 * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dh, (size_t)B * c.hidden * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->gh, (size_t)B * c.hidden * 4)) != PNVO_OK) return rc;
  if ((rc = dmalloc(m, (void **)&t->dz, (size_t)B * m->fh * m->fw * m->comp_cp * 4)) != PNVO_OK) return rc;
  int maxc = m->comp_cp, maxg = 1;
  for (const Layer &l : m->convs) {
    maxc = std::max(maxc, l.coutp);
    maxg = std::max(maxg, l.groups);
  }
  if ((rc = dmalloc(m, (void **)&t->gn_part, (size_t)B * 65 * maxc * 2 * 4)) != PNVO_OK) return rc;   // 64 chunks + [B][C][2]
  if ((rc = dmalloc(m, (void **)&t->gn_coef, (size_t)B * maxg * 2 * 4)) != PNVO_OK) return rc;
  {
    const size_t zf = (size_t)B * m->fh * m->fw * m->comp_cp, hf = (size_t)B * c.hidden;
    if ((rc = dmalloc(m, (void **)&t->zdrop, zf * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&t->hdrop, hf * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&t->dmask, (zf > hf ? zf : hf) * 4)) != PNVO_OK) return rc;
  }
  if ((rc = dmalloc(m, (void **)&t->mom_part, (size_t)128 * MOMENTS_BLOCKS * 8)) != PNVO_OK) return rc;
  if (c.act_embed) {
    const int rows = std::max(B, c.n_acts + 1);
    if ((rc = dmalloc(m, (void **)&t->egath, (size_t)rows * 32 * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&t->efeat, (size_t)rows * 32 * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&t->dfeat, (size_t)rows * 32 * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&t->biasB, (size_t)rows * c.hidden * 4)) != PNVO_OK) return rc;
    if ((rc = dmalloc(m, (void **)&t->iota, (size_t)rows * 8)) != PNVO_OK) return rc;
    std::vector<int64_t> io(rows);
    for (int k = 0; k < rows; ++k) io[k] = k;
    HIPCHK(m, hipMemcpy(t->iota, io.data(), (size_t)rows * 8, hipMemcpyHostToDevice));
  }
  t->capB = B;
  return PNVO_OK;
}

float *gradp(pnvo_handle m, TrainState *t, const std::string &name, int *rc) {
  const TocEnt *e = need(m, t, name, rc);
  return e ? t->grads + e->off : nullptr;
}

// backward-data of conv `l`: dX[B, hin, win, cin] (+)= conv(dRaw[B, hout, wout, coutp], flipped/transposed weights)
int run_dgrad(pnvo_handle m, TrainState *t, size_t li, int B, const float *draw, float *dx, bool accum, hipStream_t s) {
  const Layer &l = m->convs[li];
  if (t->dgrad_wph[li][0] != nullptr && m->opt.dgrad) {   // stride-2 3x3: four dense parity-phase convs instead of 9 taps, 3/4 masked
    PnvoTimed tm(m, s, "dgrad:" + l.name, 2.0 * (double)B * l.hout * l.wout * l.cout * l.cin * l.k * l.kw, 0.0);
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.x = draw;
        a.wpk = t->dgrad_wph[li][ph * 2 + pw];
        a.y = dx;
        a.B = B;
        a.H = l.hout;
        a.W = l.wout;
        a.CIN = l.coutp;
        a.Ho = (l.hin - ph + 1) / 2;
        a.Wo = (l.win - pw + 1) / 2;
        if (a.Ho <= 0 || a.Wo <= 0) continue;
        a.COUT = l.cin;
        a.COUTP = rup(l.cin, 32);
        a.KH = ph ? 2 : 1;
        a.KW = pw ? 2 : 1;
        a.stride = 1;
        a.pad = 0;
        a.up = 1;
        a.accum = accum ? 1 : 0;
        a.y_cstride = l.cin;
        a.y_sh = a.y_sw = 2;
        a.y_oh = ph;
        a.y_ow = pw;
        a.y_H = l.hin;
        a.y_W = l.win;
        const long M = (long)B * a.Ho * a.Wo;
        choose_tile(M, a.COUTP, &a.MT, &a.NT);
        a.slots = conv_slots(a.Ho * a.Wo, a.MT);
        HIPCHK(m, launch_conv(a, s));
      }
    return PNVO_OK;
  }
  // 3x3 stride-1: float32 results from the bf16 matrix cores (three-piece operands), as in the forward (PNVO_CONV=fp32: off)
  if (l.k == 3 && l.kw == 3 && l.stride == 1 && l.pad == 1 && !accum && l.cin % 32 == 0 && m->opt.conv <= 1) {
    ConvX3Args xa;
    std::memset(&xa, 0, sizeof(xa));
    xa.force = m->opt.conv == 1;
    xa.strip = m->opt.x3_strip;
    xa.persist_wgs = m->opt.x3_persist ? 3 * m->num_cus : 0;   // (32 -> 32 backward-data convs: resident weights, conv_x3p_kernel)
    xa.B = B;
    xa.H = l.hout;
    xa.W = l.wout;
    xa.CIN = l.coutp;
    xa.Ho = l.hin;
    xa.Wo = l.win;
    xa.COUTP = l.cin;
    int mw = 0, nw = 0;
    size_t ldsb = 0;
    auto it = t->toc.find(l.name + ".weight");
    // two float16 pieces (three terms) when the training forward uses them: the gradient is scaled by a power of two from its
    // absolute maximum (tracked by the GroupNorm backward that produced it), the weights carry the forward's per-tensor scale
    const float *wsc = m->opt.train_pieces == 2 && t->dmax != nullptr ? pnvo_train_x2_scale(m, l.name + ".weight") : nullptr;
    xa.np = wsc != nullptr ? 2 : 3;
    if (it != t->toc.end() && conv_x3_plan(xa, 3, 1, &mw, &nw, &ldsb)) {
      if (wsc != nullptr) {
        if (!t->dgrad_x2[li] || t->dgrad_x2_gen[li] != m->weights_gen) {
          const size_t nel = (size_t)9 * l.coutp * l.cin * 2;
          if (!t->dgrad_x2[li]) {
            int rc = dmalloc(m, (void **)&t->dgrad_x2[li], nel * 2);
            if (rc != PNVO_OK) return rc;
          }
          HIPCHK(m, launch_conv_x2_repack(t->params + it->second.off, l.cin, l.cout, l.coutp, l.cin, 3, 3, wsc, t->dgrad_x2[li], s, 1));
          t->dgrad_x2_gen[li] = m->weights_gen;
        }
        xa.x = draw;
        xa.wpk = t->dgrad_x2[li];
        xa.y = dx;
        xa.oscale_ptr = wsc + 1;
        xa.in_absmax = t->dmax + li * PNVO_ABSMAX_UINTS;
        PnvoTimed tm(m, s, "dgrad:" + l.name, 2.0 * (double)B * l.hout * l.wout * l.cout * l.cin * l.k * l.kw, 0.0);
        HIPCHK(m, launch_conv_x3(xa, 3, 1, 0, mw, nw, ldsb, s));
        return PNVO_OK;
      }
      if (!t->dgrad_x3[li] || t->dgrad_x3_gen[li] != m->weights_gen) {
        const size_t nel = (size_t)9 * l.coutp * l.cin * 3;
        if (!t->dgrad_x3[li]) {
          int rc = dmalloc(m, (void **)&t->dgrad_x3[li], nel * 2);
          if (rc != PNVO_OK) return rc;
        }
        HIPCHK(m, launch_conv_x3_repack(t->params + it->second.off, l.cin, l.cout, l.coutp, l.cin, 3, 3, 1, t->dgrad_x3[li], s));
        t->dgrad_x3_gen[li] = m->weights_gen;
      }
      xa.x = draw;
      xa.wpk = t->dgrad_x3[li];
      xa.y = dx;
      PnvoTimed tm(m, s, "dgrad:" + l.name, 2.0 * (double)B * l.hout * l.wout * l.cout * l.cin * l.k * l.kw, 0.0);
      HIPCHK(m, launch_conv_x3(xa, 3, 1, 0, mw, nw, ldsb, s));
      return PNVO_OK;
    }
  }
  ConvArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = draw;
  a.wpk = t->dgrad_w[li];
  a.y = dx;
  a.B = B;
  a.H = l.hout;
  a.W = l.wout;
  a.CIN = l.coutp;
  a.Ho = l.hin;
  a.Wo = l.win;
  a.COUT = l.cin;
  a.COUTP = rup(l.cin, 32);
  a.KH = l.k;
  a.KW = l.kw;
  a.stride = 1;
  a.pad = l.k - 1 - l.pad;
  a.up = l.stride;
  a.accum = accum ? 1 : 0;
  a.y_cstride = l.cin;
  const long M = (long)B * l.hin * l.win;
  choose_tile(M, a.COUTP, &a.MT, &a.NT);
  a.slots = conv_slots(l.hin * l.win, a.MT);
  PnvoTimed tm(m, s, "dgrad:" + l.name, 2.0 * (double)B * l.hout * l.wout * l.cout * l.cin * l.k * l.kw, 0.0);
  if (conv3_lds_supported(a))         // backward-data of a 3x3 stride-1 conv is a 3x3 stride-1 conv: LDS-staged kernel
    HIPCHK(m, launch_conv3_lds(a, (a.COUTP / 32) % 2 == 0 ? 2 : 1, s));
  else
    HIPCHK(m, launch_conv(a, s));
  return PNVO_OK;
}

int run_wgrad(pnvo_handle m, TrainState *t, WgradArgs &a, const std::string &pname, const int *perm, int cin_out,
              hipStream_t s) {
  int rc = PNVO_OK;
  float *g = gradp(m, t, pname, &rc);
  if (!g) return rc;
  if (wgrad_partial_floats(a) > t->wg_partial_floats) return pnvo_fail(m, PNVO_ERR_STATE, "wgrad scratch too small");
  a.partial = t->wg_partial;
  a.zero_page = m->zero_page;
  PnvoTimed tm(m, s, "wgrad:" + pname, 2.0 * (double)a.B * a.Ho * a.Wo * a.COUT * a.CIN * a.KH * a.KW, 0.0);
  HIPCHK(m, launch_wgrad(a, g, perm, cin_out, s));
  return PNVO_OK;
}

int run_gn_bwd(pnvo_handle m, TrainState *t, size_t li, int B, const float *dout, int mask, float *dx, hipStream_t s) {
  const Layer &l = m->convs[li];
  const ConvSave &c = t->cs[li];
  int rc = PNVO_OK;
  float *dg = gradp(m, t, l.gn + ".weight", &rc);
  if (!dg) return rc;
  float *db = gradp(m, t, l.gn + ".bias", &rc);
  if (!db) return rc;
  PnvoTimed tm(m, s, "gn_bwd", 0.0, 0.0);
  HIPCHK(m, launch_gn_bwd(c.raw, dout, c.ss[0], c.ss[1], c.mu, c.rstd, l.gamma, B, (long)l.hout * l.wout, l.coutp, l.cout,
                          l.groups, mask, t->gn_part, t->gn_coef, dg, db, dx, s, t->dmax ? t->dmax + li * PNVO_ABSMAX_UINTS : nullptr));
  return PNVO_OK;
}

}  // namespace

void pnvo_train_free(pnvo_handle m) {
  if (!m || !m->train) return;
  TrainState *t = TS(m);
  free_train_ws(t);
  for (auto &pm : t->maps) dfree(pm.map);
  for (auto &p : t->dgrad_w) dfree(p);
  for (auto &p : t->dgrad_x2) dfree(p);
  dfree(t->dmax);
  for (auto &q : t->dgrad_wph)
    for (auto &p : q) dfree(p);
  dfree(t->fc_t);
  dfree(t->head_t);
  dfree(t->d_ref_of_new);
  dfree(t->d_ddmaps);
  dfree(t->d_mxmaps);
  for (unsigned short *&q : t->dgrad_x3) dfree(q);
  dfree(t->d_segs);
  dfree(t->x2_seg);
  dfree(t->x2_scale);
  dfree(t->d_tensor_of_new);
  dfree(t->d_ciperm);
  if (t->embed_err) (void)hipHostFree(t->embed_err);
  dfree(t->gnb_seg);
  if (t->gnb_host) (void)hipHostFree(t->gnb_host);
  for (hipEvent_t &e : t->gnb_ev)
    if (e) (void)hipEventDestroy(e);
  m->train_mx = false;
  delete t;
  m->train = nullptr;
}

extern "C" {

int pnvo_train_attach(pnvo_handle m, float *params, float *grads, size_t n_floats, const pnvo_tensor_desc *toc, int ntoc) {
  if (!m || !params || !grads || !toc) return pnvo_fail(m, PNVO_ERR_ARG, "null argument");
  if (!m->loaded) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach before pnvo_load_weights");
  if (n_floats >= (1u << 24)) return pnvo_fail(m, PNVO_ERR_ARG, "flat parameter buffer too large for the index maps");
  HIPCHK(m, hipSetDevice(m->device));
  pnvo_train_free(m);
  TrainState *t = new TrainState();
  m->train = t;
  t->params = params;
  t->grads = grads;
  t->n = n_floats;
  for (int k = 0; k < ntoc; ++k) {
    TocEnt e;
    e.off = toc[k].offset;
    e.numel = 1;
    for (int d = 0; d < toc[k].ndim; ++d) {
      e.shape.push_back(toc[k].shape[d]);
      e.numel *= (size_t)toc[k].shape[d];
    }
    if (e.off + e.numel > n_floats) return pnvo_fail(m, PNVO_ERR_ARG, std::string("parameter '") + toc[k].name + "' out of range");
    t->toc[toc[k].name] = e;
  }
  int rc = build_maps(m, t);
  if (rc != PNVO_OK) {
    pnvo_train_free(m);
    return rc;
  }
  {
    // Buckets of the gradient-ready hook.  The backward finishes the parameters in the order head, hidden layer, compression,
    // layer4 ... layer1, stem; a range can be reported early when everything at or above its first offset belongs to layers
    // the backward has left (true for the reference's state_dict order; any other order falls back to one final range).
    auto stage_of = [](const std::string &nm) {          // 5: after the residual stages, 1..4: residual stage, 0: stem / other
      const std::string bb = "visual_encoder.backbone.layer";
      if (nm.rfind(bb, 0) == 0 && nm.size() > bb.size()) return nm[bb.size()] - '0';
      if (nm.rfind("visual_encoder.backbone.", 0) == 0 || nm.rfind("action_embedding", 0) == 0) return 0;
      return 5;
    };
    t->bucket_first.clear();
    for (int lo_stage : {4, 2}) {
      size_t first = n_floats, below_end = 0;
      for (auto &kv : t->toc) {
        const int st = stage_of(kv.first);
        if (st >= lo_stage) first = std::min(first, kv.second.off);
        else below_end = std::max(below_end, kv.second.off + kv.second.numel);
      }
      if (first < n_floats && below_end <= first && (t->bucket_first.empty() || first < t->bucket_first.back()) && first > 0)
        t->bucket_first.push_back(first);
    }
    t->bucket_first.push_back(0);
  }
  return pnvo_train_refresh(m, nullptr);
}

int pnvo_train_set_grad_hook(pnvo_handle m, pnvo_grad_ready_fn fn, void *user) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  TS(m)->hook = fn;
  TS(m)->hook_user = user;
  return PNVO_OK;
}

int pnvo_train_grad_buckets(pnvo_handle m, uint64_t *first, uint64_t *count, int cap, int *n_out) {
  if (!m || !m->train || !n_out) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  TrainState *t = TS(m);
  *n_out = (int)t->bucket_first.size();
  size_t end = t->n;
  for (int k = 0; k < *n_out && k < cap && first && count; ++k) {
    first[k] = t->bucket_first[k];
    count[k] = end - t->bucket_first[k];
    end = t->bucket_first[k];
  }
  return PNVO_OK;
}

// device pointer of a parameter inside the caller's flat buffer (nullptr: not attached / unknown name)
extern "C++" const float *pnvo_train_x2_scale(pnvo_handle m, const std::string &name) {
  if (!m || !m->train) return nullptr;
  TrainState *t = TS(m);
  auto it = t->x2_index.find(name);
  return it == t->x2_index.end() || !t->x2_scale ? nullptr : t->x2_scale + 2 * it->second;
}

extern "C++" const float *pnvo_train_weight_ptr(pnvo_handle m, const std::string &name) {
  if (!m || !m->train) return nullptr;
  TrainState *t = TS(m);
  auto it = t->toc.find(name);
  return it == t->toc.end() ? nullptr : t->params + it->second.off;
}

__global__ __launch_bounds__(64) void gn_bound_kernel(const float *params, const TrainState::GnbSeg *seg, float *out) {
  const TrainState::GnbSeg sg = seg[blockIdx.x];
  if (sg.c <= 0) {                       // no GroupNorm (or parameters not found) behind this conv: "no bound", never "not computed yet"
    if (threadIdx.x == 0) __hip_atomic_store(out + blockIdx.x, 3.0e38f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  float mx = 0.f;
  for (int c = threadIdx.x; c < sg.c; c += 64) mx = fmaxf(mx, fabsf(params[sg.goff + c]) * sg.rootn + fabsf(params[sg.boff + c]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if (threadIdx.x == 0) __hip_atomic_store(out + blockIdx.x, fminf(mx, 3.0e38f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Layer::in_bound from the bounds of the parameters TWO refreshes back (a fixed lag, so the float16-piece / three-piece choice
// per layer is the same run to run and rank to rank however far the host runs ahead of the stream): the ring entry is waited for
// through its event — the kernel behind it ran two optimiser steps ago, so the wait is over before it starts unless the host is
// more than two whole steps ahead.  Until two refreshes exist the load-time bounds stay.
static void chain_tracked_bounds(pnvo_handle m, TrainState *t) {
  if (!t->gnb_host || m->bottleneck || t->gnb_refreshes < 2) return;
  const int e = (int)((t->gnb_refreshes - 2) % 3);
  if (t->gnb_ev[e] == nullptr || hipEventSynchronize(t->gnb_ev[e]) != hipSuccess) return;
  const float *tab = t->gnb_host + (size_t)e * t->gnb_n;
  for (int i = 0; i < t->gnb_n; ++i)
    if (!(*(volatile const float *)(tab + i) >= 0.f)) return;
  pnvo_chain_in_bounds(m, [&](const Layer &l) {
    const long i = &l - m->convs.data();
    return i >= 0 && i < t->gnb_n ? *(volatile const float *)(tab + i) : 3.0e38f;
  });
}

int pnvo_train_refresh(pnvo_handle m, void *stream) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  HIPCHK(m, hipSetDevice(m->device));
  TrainState *t = TS(m);
  m->weights_gen += 1;               // operands derived lazily from the weights (conv_x3) are rebuilt at their next use
  if (!t->d_segs) {                 // segment table of all re-pack maps (built once; the maps never change after attach)
    std::vector<GatherSeg> segs;
    long start = 0;
    for (const PackMap &pm : t->maps) {
      segs.push_back(GatherSeg{pm.map, pm.dst, start});
      start += pm.n;
    }
    t->seg_total = start;
    t->nseg = (int)segs.size();
    int rc0 = dmalloc(m, (void **)&t->d_segs, segs.size() * sizeof(GatherSeg));
    if (rc0 != PNVO_OK) return rc0;
    HIPCHK(m, hipMemcpy(t->d_segs, segs.data(), segs.size() * sizeof(GatherSeg), hipMemcpyHostToDevice));
  }
  HIPCHK(m, launch_gather_all(t->params, t->d_segs, t->nseg, t->seg_total, (hipStream_t)stream));
  if (!t->x2_seg && !m->bottleneck) {       // the 3x3 / strided convs conv_x3.hip may take: one scale row each
    std::vector<long> seg;
    for (size_t li = 1; li < m->convs.size(); ++li) {
      const Layer &l = m->convs[li];
      auto it = t->toc.find(l.name + ".weight");
      if (it == t->toc.end()) continue;
      t->x2_index[l.name + ".weight"] = (int)(seg.size() / 2);
      seg.push_back((long)it->second.off);
      seg.push_back((long)it->second.numel);
    }
    if (!seg.empty()) {
      int rc0 = dmalloc(m, (void **)&t->x2_seg, seg.size() * sizeof(long));
      if (rc0 != PNVO_OK) return rc0;
      // {scale, 1/scale} per conv + the integer maxima launch_conv_x2_scales reduces into (zero between calls)
      if ((rc0 = dmalloc(m, (void **)&t->x2_scale, (seg.size() + seg.size() / 2) * sizeof(float))) != PNVO_OK) return rc0;
      HIPCHK(m, hipMemset(t->x2_scale, 0, (seg.size() + seg.size() / 2) * sizeof(float)));
      HIPCHK(m, hipMemcpy(t->x2_seg, seg.data(), seg.size() * sizeof(long), hipMemcpyHostToDevice));
    }
  }
  if (t->x2_seg) HIPCHK(m, launch_conv_x2_scales(t->params, t->x2_seg, (int)t->x2_index.size(), t->x2_scale, (hipStream_t)stream));
  if (!m->bottleneck) {                // GroupNorm bounds of the CURRENT parameters -> host-mapped table (chain_tracked_bounds)
    if (!t->gnb_seg) {
      std::vector<TrainState::GnbSeg> seg(m->convs.size(), TrainState::GnbSeg{0, 0, 0, 0.f});
      for (size_t li = 0; li < m->convs.size(); ++li) {
        const Layer &l = m->convs[li];
        auto g = t->toc.find(l.gn + ".weight"), b = t->toc.find(l.gn + ".bias");
        if (l.gn.empty() || g == t->toc.end() || b == t->toc.end() || l.groups <= 0) continue;
        seg[li] = TrainState::GnbSeg{(long)g->second.off, (long)b->second.off, l.cout,
                                     (float)std::sqrt((double)(l.cout / l.groups) * l.hout * l.wout)};
      }
      t->gnb_n = (int)seg.size();
      int rc0 = dmalloc(m, (void **)&t->gnb_seg, seg.size() * sizeof(TrainState::GnbSeg));
      if (rc0 != PNVO_OK) return rc0;
      HIPCHK(m, hipMemcpy(t->gnb_seg, seg.data(), seg.size() * sizeof(TrainState::GnbSeg), hipMemcpyHostToDevice));
      HIPCHK(m, hipHostMalloc((void **)&t->gnb_host, 3 * seg.size() * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
      for (int i = 0; i < 3 * t->gnb_n; ++i) t->gnb_host[i] = -1.f;
      for (hipEvent_t &e : t->gnb_ev) HIPCHK(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // A refresh that is being CAPTURED would bake one ring entry into the graph: every replay rewrites that entry while the host reads
    // another one behind an event that is never recorded.  So a captured refresh leaves the ring alone and marks every entry unknown
    // (-1) and restarts the two-refresh lag: chain_tracked_bounds then leaves Layer::in_bound as last chained (the load-time bounds
    // when nothing was chained yet) instead of reading an entry that no event orders.
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
    if (capturing) {
      for (int i = 0; i < 3 * t->gnb_n; ++i) t->gnb_host[i] = -1.f;
      t->gnb_refreshes = 0;
    } else {
      const int e = (int)(t->gnb_refreshes % 3);
      hipLaunchKernelGGL(gn_bound_kernel, dim3((unsigned)t->gnb_n), dim3(64), 0, (hipStream_t)stream, t->params, t->gnb_seg,
                         t->gnb_host + (size_t)e * t->gnb_n);
      HIPCHK(m, hipGetLastError());
      HIPCHK(m, hipEventRecord(t->gnb_ev[e], (hipStream_t)stream));
      t->gnb_refreshes += 1;
    }
  }
  if (m->cfg.act_embed) {      // eval-mode bias rows bias[a][o] = b1[o] + W1[o][flat:] . emb[a]  (pnvo_load_weights does this on the host)
    const pnvo_config &c = m->cfg;
    const int rows = c.n_acts + 1, flat = m->comp_c * m->fh * m->fw;
    int rc = ensure_train_ws(m, t, t->capB > 0 ? t->capB : 1);
    if (rc != PNVO_OK) return rc;
    HIPCHK(m, launch_embed_gather(t->params + t->emb_off, nullptr, rows, rows, t->egath, nullptr, (hipStream_t)stream));
    HIPCHK(m, launch_embed_bias(t->egath, t->params + t->w1_off, t->w1_pitch, flat, t->params + t->b1_off, rows, c.hidden,
                                m->fc_bias, (hipStream_t)stream));
  }
  return refresh_stem_dd(m, t, (hipStream_t)stream);
}

static int train_forward_body(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv, int B,
                              const float *run_mean, const float *run_var, float *out, void *stream);

int pnvo_train_forward(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv, int B,
                       const float *run_mean, const float *run_var, float *out, void *stream) {
  int rc = train_forward_body(m, rgb, depth, dd, tdv, B, run_mean, run_var, out, stream);
  if (rc != PNVO_OK) return rc;
  bool rerun = false;                  // an input outside the fused stems' contract: once more on the dense stem (pnvo_api.hip)
  if ((rc = pnvo_input_fallback(m, (hipStream_t)stream, &rerun)) != PNVO_OK) return rc;
  return rerun ? train_forward_body(m, rgb, depth, dd, tdv, B, run_mean, run_var, out, stream) : PNVO_OK;
}

static int train_forward_body(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv, int B,
                              const float *run_mean, const float *run_var, float *out, void *stream) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  if (B <= 0 || !out) return pnvo_fail(m, PNVO_ERR_ARG, "bad batch / null output");
  const pnvo_config &c = m->cfg;
  if ((c.n_rgb > 0) != (rgb != nullptr) || (c.n_depth > 0) != (depth != nullptr) || (c.n_dd > 0) != (dd != nullptr) ||
      (c.n_tdv > 0) != (tdv != nullptr))
    return pnvo_fail(m, PNVO_ERR_ARG, "observation tensors do not match the model's observation_space");
  if (c.normalize && (!run_mean || !run_var)) return pnvo_fail(m, PNVO_ERR_ARG, "running statistics required");
  HIPCHK(m, hipSetDevice(m->device));
  TrainState *t = TS(m);
  chain_tracked_bounds(m, t);            // float16-piece range guard follows the parameters as they train
  int rc = ensure_train_ws(m, t, B);
  if (rc != PNVO_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (c.act_embed && !t->actions) return pnvo_fail(m, PNVO_ERR_ARG, "act_embed model: pnvo_train_set_actions first");
  t->lastB = B;
  t->src[0] = rgb;
  t->src[1] = depth;
  t->src[2] = dd;
  t->src[3] = tdv;
  if (c.normalize)   // RunningMeanAndVar buffers live on the device and change every step (running_mean_and_var.py:54-60)
    HIPCHK(m, launch_whiten_table(run_mean, run_var, t->d_ref_of_new, t->d_tensor_of_new, m->CPL, m->stem_sc, m->stem_sh, s));
  if ((rc = refresh_stem_dd(m, t, s)) != PNVO_OK) return rc;   // W/std table, indicator weights: both move every step

  size_t li = 0;
  {
    ConvSave &cs = t->cs[li];
    m->in_train_forward = true;
    rc = pnvo_run_stem(m, B, t->src, cs.raw, cs.ss, cs.mu, cs.rstd, s);
    m->in_train_forward = false;
    if (rc != PNVO_OK) return rc;
    const Layer &stem = m->convs[li++];
    HIPCHK(m, launch_maxpool_train(cs.raw, cs.ss[0], cs.ss[1], B, m->Hs, m->Ws, stem.coutp, t->y[0], t->pool_idx, s));
  }
  // residual blocks: a chain of K convs (BasicBlock K = 2, resnet.py:29-55; Bottleneck K = 3, :58-117) + the skip branch
  const int K = m->bottleneck ? 3 : 2;
  int yk = 0;
  BlockTail tail{};
  const ConvSave *tail_x = nullptr;
  bool have_tail = false;
  for (int stage = 1; stage <= 4; ++stage)
    for (int bi = 0; bi < m->nblocks[stage - 1]; ++bi) {
      const float *xin = t->y[yk];
      float *yout = t->y[yk + 1];
      size_t ik[3];
      for (int k = 0; k < K; ++k) ik[k] = li++;
      const bool ds = (li < m->convs.size() && m->convs[li].name.find("downsample") != std::string::npos);
      // the block's downsample conv rides on its first conv's launch (conv_x3_kernel DSF); the block input stays in memory here —
      // the backward pass reads it
      const bool ds_ride = ds && K == 2 && pnvo_conv_takes_ds(m, m->convs[ik[0]], m->convs[li], B);
      DsRide ride{nullptr, nullptr, nullptr, nullptr, nullptr};
      if (ds_ride) {
        ConvSave &sd = t->cs[li];
        ride = DsRide{&m->convs[li], sd.raw, sd.ss, sd.mu, sd.rstd};
      }
      for (int k = 0; k < K; ++k) {
        const Layer &ck = m->convs[ik[k]];
        ConvSave &sk = t->cs[ik[k]];
        const ConvSave *sp = k ? &t->cs[ik[k - 1]] : nullptr;
        const DsRide *rd = (k == 0 && ds_ride) ? &ride : nullptr;
        if (k == 0 && have_tail) {       // the previous block's tail rides on this conv's stager, which writes xin (= tail.out)
          rc = pnvo_run_conv(m, ck, B, tail_x->raw, tail_x->ss[0], tail_x->ss[1], sk.raw, ck.coutp, sk.ss, nullptr, nullptr, 0, s, nullptr,
                             sk.mu, sk.rstd, &tail, rd);
          have_tail = false;
        } else {
          rc = pnvo_run_conv(m, ck, B, k ? sp->raw : xin, k ? sp->ss[0] : nullptr, k ? sp->ss[1] : nullptr, sk.raw, ck.coutp, sk.ss, nullptr,
                             nullptr, 0, s, nullptr, sk.mu, sk.rstd, nullptr, rd);
        }
        if (rc != PNVO_OK) return rc;
      }
      const Layer &cl = m->convs[ik[K - 1]];
      ConvSave &sl = t->cs[ik[K - 1]];
      const long P = (long)cl.hout * cl.wout;
      if (ds) {
        const size_t id = li++;
        const Layer &cd = m->convs[id];
        ConvSave &sd = t->cs[id];
        if (!ds_ride && (rc = pnvo_run_conv(m, cd, B, xin, nullptr, nullptr, sd.raw, cd.coutp, sd.ss, nullptr, nullptr, 0, s, nullptr,
                                            sd.mu, sd.rstd)) != PNVO_OK)
          return rc;
        tail = BlockTail{sd.raw, sd.ss[0], sd.ss[1], yout};
      } else {
        tail = BlockTail{xin, nullptr, nullptr, yout};
      }
      const bool last = stage == 4 && bi + 1 == m->nblocks[3];
      if (!last && pnvo_conv_takes_tail(m, m->convs[li], B)) {   // the next block's first conv computes and writes yout
        tail_x = &sl;
        have_tail = true;
      } else {
        HIPCHK(m, launch_residual(sl.raw, sl.ss[0], sl.ss[1], tail.res, tail.res_scale, tail.res_shift, B, P, cl.coutp, yout, s));
      }
      ++yk;
    }
  const int nblk_total = yk;
  {
    const size_t ic = li++;
    const Layer &comp = m->convs[ic];
    ConvSave &sc = t->cs[ic];
    if ((rc = pnvo_run_conv(m, comp, B, t->y[nblk_total], nullptr, nullptr, sc.raw, comp.coutp, sc.ss, nullptr, nullptr, 0, s, nullptr,
                            sc.mu, sc.rstd)) != PNVO_OK)
      return rc;
    ++t->drop_step;
    const float *fcb = m->fc_bias;
    const int64_t *fcrow = nullptr;
    if (c.act_embed) {                // the embedding columns of the Linear as per-sample bias rows (vo_cnn_act_embed.py:63-72)
      const int flat = m->comp_c * m->fh * m->fw;
      HIPCHK(m, launch_embed_gather(t->params + t->emb_off, t->actions, B, c.n_acts + 1, t->egath, t->embed_err, s));
      if (t->drop_p > 0.f)            // nn.Dropout acts on the concatenated [visual | embedding] vector: mask "layer" 2
        HIPCHK(m, launch_dropout(t->egath, nullptr, nullptr, B, 1, 32, t->drop_p, t->drop_seed, t->drop_step, 2, t->efeat, s));
      else
        HIPCHK(m, hipMemcpyAsync(t->efeat, t->egath, (size_t)B * 32 * 4, hipMemcpyDeviceToDevice, s));
      HIPCHK(m, launch_embed_bias(t->efeat, t->params + t->w1_off, t->w1_pitch, flat, t->params + t->b1_off, B, c.hidden,
                                  t->biasB, s));
      fcb = t->biasB;
      fcrow = t->iota;
    }
    if (t->drop_p > 0.f) {            // Dropout -> Linear -> ReLU -> Dropout -> Linear (vo_cnn.py:216-227), masks by hash
      HIPCHK(m, launch_dropout(sc.raw, sc.ss[0], sc.ss[1], B, (long)m->fh * m->fw, m->comp_cp, t->drop_p, t->drop_seed,
                               t->drop_step, 0, t->zdrop, s));
      if ((rc = pnvo_run_conv(m, m->fc, B, t->zdrop, nullptr, nullptr, t->hid, c.hidden, nullptr, fcb, fcrow, 1, s,
                              nullptr, nullptr, nullptr)) != PNVO_OK)
        return rc;
      HIPCHK(m, launch_dropout(t->hid, nullptr, nullptr, B, 1, c.hidden, t->drop_p, t->drop_seed, t->drop_step, 1, t->hdrop,
                               s));
      if ((rc = pnvo_run_conv(m, m->head, B, t->hdrop, nullptr, nullptr, out, c.out_dim, nullptr, m->head_bias, nullptr, 0,
                              s, nullptr, nullptr, nullptr)) != PNVO_OK)
        return rc;
      return PNVO_OK;
    }
    if ((rc = pnvo_run_conv(m, m->fc, B, sc.raw, sc.ss[0], sc.ss[1], t->hid, c.hidden, nullptr, fcb, fcrow, 1, s,
                            nullptr, nullptr, nullptr)) != PNVO_OK)
      return rc;
    if ((rc = pnvo_run_conv(m, m->head, B, t->hid, nullptr, nullptr, out, c.out_dim, nullptr, m->head_bias, nullptr, 0, s,
                            nullptr, nullptr, nullptr)) != PNVO_OK)
      return rc;
  }
  return PNVO_OK;
}

static int train_backward_body(pnvo_handle m, const float *grad_out, void *stream);

// bucket k of the gradient-ready hook is final: report it (host callback; the launches that produce it are enqueued on `s`)
static void report_bucket(TrainState *t, size_t k, hipStream_t s) {
  if (k >= t->bucket_first.size()) return;
  if (t->bucket_done.size() != t->bucket_first.size()) t->bucket_done.assign(t->bucket_first.size(), 0);
  if (t->bucket_done[k]) return;
  t->bucket_done[k] = 1;
  if (!t->hook) return;
  const size_t end = k == 0 ? t->n : t->bucket_first[k - 1];
  t->hook(t->hook_user, (uint64_t)t->bucket_first[k], (uint64_t)(end - t->bucket_first[k]), (void *)s);
}

int pnvo_train_backward(pnvo_handle m, const float *grad_out, void *stream) {
  if (m && m->train) TS(m)->bucket_done.assign(TS(m)->bucket_first.size(), 0);
  const int rc = train_backward_body(m, grad_out, stream);
  // the end of the backward: every range not reported on the way (whatever the number of split points a parameter order allowed —
  // with exactly one of them the in-backward reports are skipped and BOTH ranges are due here), latest layers first
  if (rc == PNVO_OK && m->train)
    for (size_t k = 0; k < TS(m)->bucket_first.size(); ++k) report_bucket(TS(m), k, (hipStream_t)stream);
  return rc;
}

static int train_backward_body(pnvo_handle m, const float *grad_out, void *stream) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  TrainState *t = TS(m);
  g_wgrad_x3 = m->opt.wgrad3;
  if (t->lastB <= 0 || !grad_out) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_forward first");
  HIPCHK(m, hipSetDevice(m->device));
  const pnvo_config &c = m->cfg;
  const int B = t->lastB;
  hipStream_t s = (hipStream_t)stream;
  int rc = PNVO_OK;

  const int nblk_total = m->nblocks[0] + m->nblocks[1] + m->nblocks[2] + m->nblocks[3];
  if (!t->dmax && !m->bottleneck) {
    if ((rc = dmalloc(m, (void **)&t->dmax, m->convs.size() * PNVO_ABSMAX_UINTS * sizeof(unsigned))) != PNVO_OK) return rc;
  }
  if (t->dmax) HIPCHK(m, hipMemsetAsync(t->dmax, 0, m->convs.size() * PNVO_ABSMAX_UINTS * sizeof(unsigned), s));   // maxima of this backward's gradients
  // ---- output head: out = hid . W2^T + b2
  HIPCHK(m, launch_padcopy(grad_out, B, c.out_dim, 8, t->dout8, s));
  {
    float *gb = gradp(m, t, "output_head.1.bias", &rc);
    if (!gb) return rc;
    HIPCHK(m, launch_colsum(grad_out, B, c.out_dim, c.out_dim, gb, s));
    WgradArgs a = wgrad_args(m->head, B, c.hidden, 8);
    a.x = t->drop_p > 0.f ? t->hdrop : t->hid;
    a.dy = t->dout8;
    a.mode = 0;
    if ((rc = run_wgrad(m, t, a, "output_head.1.weight", nullptr, c.hidden, s)) != PNVO_OK) return rc;
    ConvArgs d;
    std::memset(&d, 0, sizeof(d));
    d.x = t->dout8;
    d.wpk = t->head_t;
    d.y = t->dh;
    d.B = B;
    d.H = d.W = d.Ho = d.Wo = 1;
    d.CIN = 8;
    d.COUT = c.hidden;
    d.COUTP = rup(c.hidden, 32);
    d.KH = d.KW = 1;
    d.stride = 1;
    d.up = 1;
    d.y_cstride = c.hidden;
    choose_tile(B, d.COUTP, &d.MT, &d.NT);
    d.slots = conv_slots(1, d.MT);
    HIPCHK(m, launch_conv(d, s));
  }
  // ---- hidden layer: hid = relu(z . W1^T + b1)
  const size_t icomp = m->convs.size() - 1;
  {
    if (t->drop_p > 0.f) {            // dropout backward: the same mask, then the ReLU mask of the un-dropped activation
      HIPCHK(m, launch_dropout(t->dh, nullptr, nullptr, B, 1, c.hidden, t->drop_p, t->drop_seed, t->drop_step, 1, t->dh, s));
    }
    HIPCHK(m, launch_relu_mask(t->dh, t->hid, nullptr, (long)B * c.hidden, t->gh, s));
    float *gb = gradp(m, t, m->fc.name + ".bias", &rc);
    if (!gb) return rc;
    HIPCHK(m, launch_colsum(t->gh, B, c.hidden, c.hidden, gb, s));
    const ConvSave &sc = t->cs[icomp];
    WgradArgs a = wgrad_args(m->fc, B, m->comp_c, c.hidden);
    a.CIN = m->comp_c;
    a.x = sc.raw;
    a.in_scale = sc.ss[0];
    a.in_shift = sc.ss[1];
    a.dy = t->gh;
    a.mode = 1;
    if (t->drop_p > 0.f) {            // the Linear saw the dropped, already activated feature
      a.x = t->zdrop;
      a.in_scale = a.in_shift = nullptr;
      a.mode = 0;
    }
    // the activation tensor is channel-padded: tell the kernel the real row pitch through H/W/CIN = (fh, fw, comp_cp)
    a.CIN = m->comp_cp;
    wgrad_plan(a);
    if (c.act_embed) {                // gradient rows are [flat | 32] wide; the embedding columns and rows come from embed_*
      const int flat = m->comp_c * m->fh * m->fw;
      a.grad_pitch = t->w1_pitch;
      HIPCHK(m, launch_embed_backward(t->gh, t->efeat, t->params + t->w1_off, t->w1_pitch, flat, B, c.hidden,
                                      t->grads + t->w1_off, t->dfeat, s));
      if (t->drop_p > 0.f)
        HIPCHK(m, launch_dropout(t->dfeat, nullptr, nullptr, B, 1, 32, t->drop_p, t->drop_seed, t->drop_step, 2, t->dfeat, s));
      HIPCHK(m, launch_embed_scatter(t->dfeat, t->actions, B, c.n_acts + 1, t->grads + t->emb_off, s));
    }
    if ((rc = run_wgrad(m, t, a, m->fc.name + ".weight", nullptr, m->comp_c, s)) != PNVO_OK) return rc;
    ConvArgs d;
    std::memset(&d, 0, sizeof(d));
    d.x = t->gh;
    d.wpk = t->fc_t;
    d.y = t->dz;
    d.B = B;
    d.H = d.W = d.Ho = d.Wo = 1;
    d.CIN = c.hidden;
    d.COUT = m->fh * m->fw * m->comp_cp;
    d.COUTP = rup(d.COUT, 32);
    d.KH = d.KW = 1;
    d.stride = 1;
    d.up = 1;
    d.y_cstride = d.COUT;
    choose_tile(B, d.COUTP, &d.MT, &d.NT);
    d.slots = conv_slots(1, d.MT);
    HIPCHK(m, launch_conv(d, s));
  }
  if (t->drop_p > 0.f)
    HIPCHK(m, launch_dropout(t->dz, nullptr, nullptr, B, (long)m->fh * m->fw, m->comp_cp, t->drop_p, t->drop_seed,
                             t->drop_step, 0, t->dz, s));
  // ---- compression conv + GroupNorm(1) + ReLU
  float *dY = t->dYa, *dX = t->dYb;
  {
    const Layer &l = m->convs[icomp];
    if ((rc = run_gn_bwd(m, t, icomp, B, t->dz, 1, t->dz, s)) != PNVO_OK) return rc;   // in place: dz -> dCompRaw
    WgradArgs a = wgrad_args(l, B, l.cin, l.coutp);
    a.x = t->y[nblk_total];
    a.dy = t->dz;
    a.mode = 0;
    if ((rc = run_wgrad(m, t, a, l.name + ".weight", nullptr, l.cin, s)) != PNVO_OK) return rc;
    if ((rc = run_dgrad(m, t, icomp, B, t->dz, dY, false, s)) != PNVO_OK) return rc;
  }
  // ---- residual blocks, last to first.  Conv indices: walk m->convs backwards from the compression layer.
  size_t li = icomp;
  const int K = m->bottleneck ? 3 : 2;
  for (int blk = nblk_total; blk >= 1; --blk) {
    const bool ds = m->convs[li - 1].name.find("downsample") != std::string::npos;
    const size_t id = ds ? li - 1 : 0;
    size_t ik[3];
    {
      size_t last = ds ? li - 2 : li - 1;
      for (int k = K - 1; k >= 0; --k) ik[k] = last - (size_t)(K - 1 - k);
    }
    li = ik[0];
    const Layer &c1 = m->convs[ik[0]], &cl = m->convs[ik[K - 1]];
    const float *xin = t->y[blk - 1];
    const long nout = (long)B * cl.hout * cl.wout * cl.coutp;
    const long nin = (long)B * c1.hin * c1.win * c1.cin;
    // G = dY * (y > 0)
    HIPCHK(m, launch_relu_mask(dY, t->y[blk], nullptr, nout, t->G, s));
    // the chain, last conv first: GroupNorm (+ ReLU mask for every conv but the last) <- incoming gradient
    const float *din = t->G;
    for (int k = K - 1; k >= 0; --k) {
      const Layer &ck = m->convs[ik[k]];
      if ((rc = run_gn_bwd(m, t, ik[k], B, din, k == K - 1 ? 0 : 1, t->dRaw, s)) != PNVO_OK) return rc;
      WgradArgs a = wgrad_args(ck, B, ck.cin, ck.coutp, k ? 1 : 0);
      if (k) {                        // the conv saw relu(GN(raw of the previous conv)): recomputed in the fetch
        const ConvSave &sp = t->cs[ik[k - 1]];
        a.x = sp.raw;
        a.in_scale = sp.ss[0];
        a.in_shift = sp.ss[1];
        a.mode = 1;
      } else {
        a.x = xin;
        a.mode = 0;
      }
      a.dy = t->dRaw;
      if (a.lds3 == 6 && m->opt.train_pieces == 2 && t->dmax != nullptr && ck.in_bound < 6.0e4f) {
        a.np = 2;                     // float16 pieces: X inside float16's range (the forward's bound), dY scaled from its maximum
        a.dy_absmax = t->dmax + ik[k] * PNVO_ABSMAX_UINTS;
      }
      if ((rc = run_wgrad(m, t, a, ck.name + ".weight", nullptr, ck.cin, s)) != PNVO_OK) return rc;
      if ((rc = run_dgrad(m, t, ik[k], B, t->dRaw, k ? t->dA : dX, false, s)) != PNVO_OK) return rc;
      din = t->dA;
    }
    // skip connection
    if (ds) {
      const Layer &cd = m->convs[id];
      if ((rc = run_gn_bwd(m, t, id, B, t->G, 0, t->dRaw, s)) != PNVO_OK) return rc;
      WgradArgs a = wgrad_args(cd, B, cd.cin, cd.coutp);
      a.x = xin;
      a.dy = t->dRaw;
      a.mode = 0;
      if ((rc = run_wgrad(m, t, a, cd.name + ".weight", nullptr, cd.cin, s)) != PNVO_OK) return rc;
      if ((rc = run_dgrad(m, t, id, B, t->dRaw, dX, true, s)) != PNVO_OK) return rc;
    } else {
      HIPCHK(m, launch_add(dX, t->G, nin, dX, s));
    }
    std::swap(dY, dX);
    // leaving residual stage 4 / stage 2: every parameter from that stage's first offset upwards has its final gradient
    if (t->bucket_first.size() == 3) {
      if (blk - 1 == m->nblocks[0] + m->nblocks[1] + m->nblocks[2]) report_bucket(t, 0, s);
      if (blk - 1 == m->nblocks[0]) report_bucket(t, 1, s);
    }
  }
  // ---- stem: maxpool <- dY, GroupNorm + ReLU, weight gradient (no input gradient)
  {
    const Layer &l = m->convs[0];
    if (m->opt.pool_bwd && l.coutp == l.cout && l.cout % 4 == 0 && l.cout <= 256 && 256 % (l.cout / 4) == 0) {
      // max-pool backward folded into the stem's GroupNorm backward: the un-pooled gradient is never materialised
      const ConvSave &cs0 = t->cs[0];
      float *dg = gradp(m, t, l.gn + ".weight", &rc);
      if (!dg) return rc;
      float *db = gradp(m, t, l.gn + ".bias", &rc);
      if (!db) return rc;
      PnvoTimed tm(m, s, "gn_bwd", 0.0, 0.0);
      HIPCHK(m, launch_gn_bwd_pool(cs0.raw, dY, t->pool_idx, m->Hs, m->Ws, m->Hp, m->Wp, cs0.ss[0], cs0.ss[1], cs0.mu, cs0.rstd, l.gamma,
                                   B, l.cout, l.groups, t->gn_part, t->gn_coef, dg, db, t->dStem, s));
    } else {
      HIPCHK(m, launch_maxpool_bwd(dY, t->pool_idx, B, m->Hs, m->Ws, l.coutp, t->dStem, s));
      if ((rc = run_gn_bwd(m, t, 0, B, t->dStem, 1, t->dStem, s)) != PNVO_OK) return rc;
    }
    // option wgrad_stem=fp32: the float32-MFMA kernel; also once an input outside the exact-operand contract was met
    const bool stem_fp32 = m->opt.wgrad_stem == 1 || m->dense_sticky;
    if (m->train_mx && l.coutp == 32 && !stem_fp32) {
      WgradStemMXArgs a;
      std::memset(&a, 0, sizeof(a));
      for (int k = 0; k < 4; ++k) a.src[k] = t->src[k];
      a.dy = t->dStem;
      a.zero_page = m->zero_page;
      a.B = B;
      a.H = c.height;
      a.W = c.width;
      a.Ho = m->Hs;
      a.Wo = m->Ws;
      wgrad_stem_mx_plan(a);
      if (wgrad_stem_mx_scratch_floats(a) > t->wg_partial_floats) return pnvo_fail(m, PNVO_ERR_STATE, "wgrad scratch too small");
      int rc2 = PNVO_OK;
      float *g = gradp(m, t, l.name + ".weight", &rc2);
      if (!g) return rc2;
      PnvoTimed tm(m, s, "wgrad:" + l.name + ".weight", 2.0 * (double)B * m->Hs * m->Ws * l.cout * l.cin * 49, 0.0);
      HIPCHK(m, launch_wgrad_stem_mx(a, t->wg_partial, m->stem_sc, m->stem_sh, t->d_mxmaps, t->d_mxmaps + 32, l.cin, g, s));
      return PNVO_OK;
    }
    WgradArgs a = wgrad_args(l, B, 32, l.coutp, 2);
    a.dy = t->dStem;
    const int nsrc[4] = {c.n_rgb, c.n_depth, c.n_dd, c.n_tdv};
    std::vector<float> sc(m->CPL), sh(m->CPL);
    // the whitening constants are on the device (stem_sc/sh); the kernel wants them per lane: read them back once per
    // call is a sync — instead pass pointers: lanes load sc/sh from the device table themselves
    for (int k = 0; k < 32; ++k) {
      const int tn = k < m->CP ? m->stem_tensor_of_new[k] : -1;
      a.src[k].base = tn >= 0 ? t->src[tn] : nullptr;
      a.src[k].nch = tn >= 0 ? nsrc[tn] : 0;
      a.src[k].choff = tn >= 0 ? m->stem_ch_of_new[k] : 0;
      a.src[k].sc = 0.f;
      a.src[k].sh = 0.f;
    }
    a.in_scale = m->stem_sc;     // mode 2 reads its per-channel whitening from these device tables
    a.in_shift = m->stem_sh;
    if ((rc = run_wgrad(m, t, a, l.name + ".weight", t->d_ciperm, l.cin, s)) != PNVO_OK) return rc;
  }
  return PNVO_OK;
}

int pnvo_train_set_dropout(pnvo_handle m, float p, uint64_t seed) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  if (!(p >= 0.f && p < 1.f)) return pnvo_fail(m, PNVO_ERR_ARG, "dropout p must be in [0, 1)");
  TrainState *t = TS(m);
  t->drop_p = p;
  t->drop_seed = seed;
  return PNVO_OK;
}

int pnvo_train_set_actions(pnvo_handle m, const int64_t *actions) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  TrainState *t = TS(m);
  if (t->embed_err && *t->embed_err) {
    *t->embed_err = 0;
    return pnvo_fail(m, PNVO_ERR_INPUT, "an action of the previous step was outside the embedding table");
  }
  t->actions = reinterpret_cast<const long long *>(actions);
  return PNVO_OK;
}

int pnvo_train_dropout_mask(pnvo_handle m, int layer, float *out, void *stream) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  TrainState *t = TS(m);
  if (t->lastB <= 0 || !out || layer < 0 || layer > 2) return pnvo_fail(m, PNVO_ERR_ARG, "bad argument / no forward yet");
  const pnvo_config &c = m->cfg;
  const float p = t->drop_p;
  if (layer == 2)
    HIPCHK(m, launch_dropout(nullptr, nullptr, nullptr, t->lastB, 1, 32, p, t->drop_seed, t->drop_step, 2, out,
                             (hipStream_t)stream));
  else if (layer == 0)
    HIPCHK(m, launch_dropout(nullptr, nullptr, nullptr, t->lastB, (long)m->fh * m->fw, m->comp_cp, p, t->drop_seed,
                             t->drop_step, 0, out, (hipStream_t)stream));
  else
    HIPCHK(m, launch_dropout(nullptr, nullptr, nullptr, t->lastB, 1, c.hidden, p, t->drop_seed, t->drop_step, 1, out,
                             (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_input_moments(pnvo_handle m, const float *rgb, const float *depth, const float *dd, const float *tdv, int B,
                       const float *center, int power, float *out, void *stream) {
  if (!m || !m->train) return pnvo_fail(m, PNVO_ERR_STATE, "pnvo_train_attach first");
  if (!out || power < 1 || power > 3) return pnvo_fail(m, PNVO_ERR_ARG, "bad argument");
  HIPCHK(m, hipSetDevice(m->device));
  TrainState *t = TS(m);
  int rc = ensure_train_ws(m, t, B);
  if (rc != PNVO_OK) return rc;
  const pnvo_config &c = m->cfg;
  MomentsArgs a;
  std::memset(&a, 0, sizeof(a));
  a.src[0] = rgb;
  a.src[1] = depth;
  a.src[2] = dd;
  a.src[3] = tdv;
  a.nch[0] = c.n_rgb;
  a.nch[1] = c.n_depth;
  a.nch[2] = c.n_dd;
  a.nch[3] = c.n_tdv;
  for (int k = 0; k < m->CP; ++k) {
    const int r = m->stem_ref_of_new[k];
    if (r < 0) continue;
    a.tensor[r] = m->stem_tensor_of_new[k];
    a.ch[r] = m->stem_ch_of_new[k];
  }
  a.center = center;
  a.npix = (long)B * c.height * c.width;
  a.pw = power;
  HIPCHK(m, launch_moments(a, m->C, t->mom_part, out, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_rmv_merge(const float *m12, int C, int B, float *mean, float *var, float *count, void *stream) {
  if (!m12 || !mean || !var || !count || C < 1 || C > 1024 || B < 1) return PNVO_ERR_ARG;
  return launch_rmv_merge(m12, C, B, mean, var, count, (hipStream_t)stream) == hipSuccess ? PNVO_OK : PNVO_ERR_HIP;
}

int pnvo_mse_loss(const float *pred, const float *target, int B, int D, float *loss, float *grad, void *stream) {
  if (!pred || !target || B <= 0 || D <= 0) return pnvo_fail(nullptr, PNVO_ERR_ARG, "bad argument");
  HIPCHK(nullptr, launch_mse_loss(pred, target, B, D, loss, grad, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_mse_loss_coef(const float *pred, const float *target, const float *coef, int n, float *loss, float *grad,
                       void *stream) {
  if (!pred || !target || !coef || n <= 0) return pnvo_fail(nullptr, PNVO_ERR_ARG, "bad argument");
  HIPCHK(nullptr, launch_mse_loss_coef(pred, target, coef, n, loss, grad, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_geo_inverse_loss(const float *deltas, const int32_t *actions, int n_entries, int move_forward, float weight,
                          float *out4, float *grad, void *stream) {
  if (!deltas || !actions || n_entries <= 0 || (n_entries & 1))
    return pnvo_fail(nullptr, PNVO_ERR_ARG, "deltas must hold an even number of alternating (cur_rel_to_prev, prev_rel_to_cur) rows");
  HIPCHK(nullptr, launch_geo_inverse_loss(deltas, actions, n_entries / 2, move_forward, weight, out4, grad, (hipStream_t)stream));
  return PNVO_OK;
}

int pnvo_adam_step(float *p, const float *g, float *mom, float *var, size_t n, float lr, float beta1, float beta2, float eps,
                   int step, void *stream) {
  if (!p || !g || !mom || !var || step < 1) return pnvo_fail(nullptr, PNVO_ERR_ARG, "bad argument");
  HIPCHK(nullptr, launch_adam(p, g, mom, var, (long)n, lr, beta1, beta2, eps, step, (hipStream_t)stream));
  return PNVO_OK;
}

}  // extern "C"
