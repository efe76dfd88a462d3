// pnvo_bf16.hip — host side of the native-bf16 forward (BASELINE config 3: the geometric-invariance dual forward in
// bf16).  pnvo_set_precision(h, 1) routes pnvo_forward here; pnvo_forward_dual runs TWO action models in every launch
// (blockIdx.z / the stem's second N-tile), the second one on the channel-swapped (cur, prev) pair that the reference's
// dataset builds for the opposite action (regression_geo_invariance_iter_dataset.py:342-386, engine :569-602) — here a
// permutation of that model's stem weights, so the observation tensors are read from HBM once for both models.
//
// Numerics: bf16 operands on the matrix cores, fp32 accumulation; activations are stored in bf16 (raw conv outputs,
// block outputs); GroupNorm statistics come from the fp32 accumulators and stay fp32 (scale/shift, whitening constants,
// the compression output, both Linear layers).  Kernels: stem_mx.hip (PIECES = 1), conv_bf16.hip, the fp32 split-K
// linear kernels of conv_mfma.hip for visual_fc / output_head.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pnvo_model.h"

using namespace pnvo;

namespace {

struct BLayer {
  ConvBArgs plan;               // shape + tile plan (pointers filled per launch)
  int mw = 1, nw = 1;
  size_t lds = 0;
  unsigned short *wpk = nullptr;
};

struct Bf16State {
  unsigned long long gen = 0;
  std::vector<BLayer> layers;   // parallel to m->convs (index 0 unused: the stem)
  unsigned short *stem_wpk = nullptr;        // PIECES = 1 packing, one N-tile group (this model on (prev, cur))
  std::vector<unsigned short> stem_host, stem_host_sw;   // host copies: as is / for the swapped pair
  unsigned short *dual_stem = nullptr;       // [tap][fragment][2 models][lane]: this model + a partner's swapped packing
  unsigned long long dual_partner = 0;       // process-unique id of the partner handle (an address can be re-used)
  unsigned long long dual_partner_gen = 0, dual_self_gen = 0;
  int cap = 0;
  unsigned short *stem_raw = nullptr, *bufY[2] = {nullptr, nullptr}, *rawA = nullptr, *rawB = nullptr, *rawD = nullptr;
  float *comp_raw = nullptr, *hid = nullptr, *stats = nullptr, *stats_ds = nullptr;   // stats_ds: partial sums of a riding downsample conv
  float *ssA[2] = {nullptr, nullptr}, *ssB[2] = {nullptr, nullptr}, *ssD[2] = {nullptr, nullptr}, *ssC[2] = {nullptr, nullptr};
};

template <typename T>
void dfree(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

void free_ws(Bf16State *b) {
  dfree(b->stem_raw);
  dfree(b->bufY[0]);
  dfree(b->bufY[1]);
  dfree(b->rawA);
  dfree(b->rawB);
  dfree(b->rawD);
  dfree(b->comp_raw);
  dfree(b->hid);
  dfree(b->stats);
  dfree(b->stats_ds);
  for (int k = 0; k < 2; ++k) {
    dfree(b->ssA[k]);
    dfree(b->ssB[k]);
    dfree(b->ssD[k]);
    dfree(b->ssC[k]);
  }
  b->cap = 0;
}

int prepare(pnvo_handle m) {
  Bf16State *b = static_cast<Bf16State *>(m->bf);
  // The bf16 operands are packed from the host copies pnvo_load_weights took.  Parameters that moved on the device since
  // then (pnvo_adam_step + pnvo_train_refresh bump weights_gen) would silently evaluate as the pre-training weights.
  if (m->weights_gen != m->weights_gen_at_load)
    return pnvo_fail(m, PNVO_ERR_STATE, "the bfloat16 operands are built from the weights given to pnvo_load_weights, and the parameters "
                                        "have been updated on the device since (pnvo_train_refresh): call pnvo_load_weights with the "
                                        "current parameters before a bfloat16 forward");
  if (b && b->gen == m->load_gen) return PNVO_OK;
  const pnvo_config &c = m->cfg;
  if (m->bottleneck || c.baseplanes != 32 || !m->mx_ok || m->convs[0].cout != 32)
    return pnvo_fail(m, PNVO_ERR_ARG, "the bfloat16 path covers the resnet18 (BasicBlock, baseplanes 32) models");
  if (!b) {
    b = new Bf16State();
    m->bf = b;
  }
  if (b->layers.size() != m->convs.size()) {
    for (BLayer &l : b->layers) dfree(l.wpk);
    b->layers.assign(m->convs.size(), BLayer());
  }
  for (size_t li = 1; li < m->convs.size(); ++li) {
    const Layer &l = m->convs[li];
    BLayer &bl = b->layers[li];
    std::memset(&bl.plan, 0, sizeof(bl.plan));
    bl.plan.H = l.hin;
    bl.plan.W = l.win;
    bl.plan.CIN = rup(l.cin, 32);
    bl.plan.Ho = l.hout;
    bl.plan.Wo = l.wout;
    bl.plan.COUTP = l.coutp;
    if (!conv_bf16_plan(bl.plan, l.k, l.stride, &bl.mw, &bl.nw, &bl.lds))
      return pnvo_fail(m, PNVO_ERR_ARG, "layer '" + l.name + "' is outside the bfloat16 conv kernel's shapes");
    if (l.host_w.empty()) return pnvo_fail(m, PNVO_ERR_STATE, "weights of '" + l.name + "' were not loaded");
    std::vector<unsigned short> pk((size_t)l.k * l.kw * bl.plan.CIN * l.coutp);
    pack_conv_bf16_weight(l.host_w.data(), l.cout, l.cin, bl.plan.CIN, l.coutp, l.k, l.kw, pk.data());
    if (!bl.wpk) HIPCHK(m, hipMalloc((void **)&bl.wpk, pk.size() * 2));
    HIPCHK(m, hipMemcpy(bl.wpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
  }
  {
    const size_t n = stem_mx_packed_u16(1, 1);
    b->stem_host.resize(n);
    b->stem_host_sw.resize(n);
    pack_stem_mx_weight(m->mx_wk.data(), 32, 1, m->mx_xslot, b->stem_host.data());
    pack_stem_mx_weight(m->mx_wk_swapped.data(), 32, 1, m->mx_xslot, b->stem_host_sw.data());
    if (!b->stem_wpk) HIPCHK(m, hipMalloc((void **)&b->stem_wpk, n * 2));
    HIPCHK(m, hipMemcpy(b->stem_wpk, b->stem_host.data(), n * 2, hipMemcpyHostToDevice));
  }
  b->gen = m->load_gen;
  return PNVO_OK;
}

int ensure_ws(pnvo_handle m, int B) {
  Bf16State *b = static_cast<Bf16State *>(m->bf);
  if (B <= b->cap) return PNVO_OK;
  free_ws(b);
  const pnvo_config &c = m->cfg;
  size_t act = (size_t)B * m->Hp * m->Wp * c.baseplanes, st = (size_t)B * stem_mx_slots(m->Hs, m->Ws) * 32 * 2;
  int maxc = m->comp_cp;
  for (size_t li = 1; li < m->convs.size(); ++li) {
    const Layer &l = m->convs[li];
    act = std::max(act, (size_t)B * l.hout * l.wout * l.coutp);
    st = std::max(st, (size_t)B * b->layers[li].plan.slots * l.coutp * 2);
    maxc = std::max(maxc, l.coutp);
  }
  HIPCHK(m, hipMalloc((void **)&b->stem_raw, (size_t)B * m->Hs * m->Ws * 32 * 2));
  for (unsigned short **pp : {&b->bufY[0], &b->bufY[1], &b->rawA, &b->rawB, &b->rawD}) HIPCHK(m, hipMalloc((void **)pp, act * 2));
  HIPCHK(m, hipMalloc((void **)&b->comp_raw, (size_t)B * m->fh * m->fw * m->comp_cp * 4));
  HIPCHK(m, hipMalloc((void **)&b->hid, (size_t)B * c.hidden * 4));
  HIPCHK(m, hipMalloc((void **)&b->stats, st * 4));
  HIPCHK(m, hipMalloc((void **)&b->stats_ds, st * 4));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(m, hipMalloc((void **)&b->ssA[k], (size_t)B * maxc * 4));
    HIPCHK(m, hipMalloc((void **)&b->ssB[k], (size_t)B * maxc * 4));
    HIPCHK(m, hipMalloc((void **)&b->ssD[k], (size_t)B * maxc * 4));
    HIPCHK(m, hipMalloc((void **)&b->ssC[k], (size_t)B * m->comp_cp * 4));
    HIPCHK(m, hipMemset(b->ssC[k], 0, (size_t)B * m->comp_cp * 4));      // pad channels stay (0, 0)
  }
  b->cap = B;
  return PNVO_OK;
}

// the stem's B operand for a dual launch: [tap][fragment][model 0 | model 1 on the swapped pair][lane]
int dual_stem(pnvo_handle m0, pnvo_handle m1) {
  Bf16State *b0 = static_cast<Bf16State *>(m0->bf), *b1 = static_cast<Bf16State *>(m1->bf);
  if (b0->dual_stem && b0->dual_partner == m1->uid && b0->dual_partner_gen == m1->load_gen && b0->dual_self_gen == m0->load_gen)
    return PNVO_OK;
  const size_t n1 = stem_mx_packed_u16(1, 1);            // 49 taps x 2 fragments x 512 u16
  std::vector<unsigned short> pk(2 * n1);
  for (size_t tf = 0; tf < n1 / 512; ++tf) {
    std::memcpy(&pk[(2 * tf) * 512], &b0->stem_host[tf * 512], 1024);
    std::memcpy(&pk[(2 * tf + 1) * 512], &b1->stem_host_sw[tf * 512], 1024);
  }
  if (!b0->dual_stem) HIPCHK(m0, hipMalloc((void **)&b0->dual_stem, pk.size() * 2));
  HIPCHK(m0, hipMemcpy(b0->dual_stem, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
  b0->dual_partner = m1->uid;
  b0->dual_partner_gen = m1->load_gen;
  b0->dual_self_gen = m0->load_gen;
  return PNVO_OK;
}

}  // namespace

void pnvo_bf16_free(pnvo_handle m) {
  Bf16State *b = static_cast<Bf16State *>(m->bf);
  if (!b) return;
  free_ws(b);
  for (BLayer &l : b->layers) dfree(l.wpk);
  dfree(b->stem_wpk);
  dfree(b->dual_stem);
  delete b;
  m->bf = nullptr;
}

int pnvo_forward_bf16(pnvo_handle *hs, int nm, const float *rgb, const float *depth, const float *dd, const float *tdv,
                      const int64_t *actions, int B, float *const *outs, hipStream_t s) {
  pnvo_handle m = hs[0];
  Bf16State *bs[2] = {nullptr, nullptr};
  int rc;
  for (int z = 0; z < nm; ++z) {
    if ((rc = prepare(hs[z])) != PNVO_OK) return z == 0 ? rc : pnvo_fail(m, rc, hs[z]->err);
    bs[z] = static_cast<Bf16State *>(hs[z]->bf);
    if ((rc = ensure_ws(hs[z], B)) != PNVO_OK) return z == 0 ? rc : pnvo_fail(m, rc, hs[z]->err);
  }
  if (nm == 2 && (rc = dual_stem(hs[0], hs[1])) != PNVO_OK) return rc;
  const pnvo_config &c = m->cfg;
  const Layer &stem = m->convs[0];
  auto each = [&](auto f) {                       // per-model pointer arrays for the batched launches
    using T = decltype(f(0));
    struct R { T v[2]; } r;
    for (int z = 0; z < 2; ++z) r.v[z] = f(z < nm ? z : 0);
    return r;
  };

  // (a4-a6) fused stem, one N-tile per model
  {
    StemMXArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src[0] = rgb;
    a.src[1] = depth;
    a.src[2] = dd;
    a.src[3] = tdv;
    pnvo_stem_raw_args(m, a);                                    // (pnvo_forward_raw: sensor frames instead of src[0..2])
    a.zero_page = m->mx_pages;
    a.wpk = nm == 2 ? bs[0]->dual_stem : bs[0]->stem_wpk;
    for (int z = 0; z < nm; ++z) {
      a.y[z] = bs[z]->stem_raw;
      a.stats[z] = bs[z]->stats;
      a.y_coff[z] = 0;
    }
    a.y_cstride = 32;
    a.stats_cstride = 32;
    a.B = B;
    a.H = c.height;
    a.W = c.width;
    a.Ho = m->Hs;
    a.Wo = m->Ws;
    a.slots = stem_mx_slots(m->Hs, m->Ws);
    const double M = (double)B * m->Hs * m->Ws;
    {
      const double in_bytes = (double)B * c.height * c.width * (m->raw_depth ? (c.n_rgb ? 6.0 : 0.0) + 8.0 + (c.n_tdv ? 8.0 : 0.0) : 4.0 * stem.cin);
      PnvoTimed t(m, s, "bf16:stem", 2.0 * nm * M * stem.cout * stem.cin * 49, in_bytes + 2.0 * nm * M * stem.cout);
      if (m->opt.bf16_stem3 == 1 && nm == 1) {                 // experiment: exact three-piece stem in front of the bf16 stages
        a.wpk = m->mx_wpk3;
        HIPCHK(m, launch_stem_mx(a, 3, 1, true, s));
      } else if ((m->opt.stem_form == 0 || m->opt.stem_form == 3 || m->opt.stem_form == 4) && stem_rs_takes(a, 1, nm, true, m->num_cus)) {
        // two models: the dual stem with its 196 KB of weight fragments resident in registers (stem_rs.hip), bit-identical
        HIPCHK(m, launch_stem_rs(a, 1, false, m->num_cus, s));
      } else {
        HIPCHK(m, launch_stem_mx(a, 1, nm, true, s));
      }
    }
    PnvoTimed t(m, s, "bf16:gn_finalize", 0.0, 0.0);
    auto st = each([&](int z) { return (const float *)bs[z]->stats; });
    auto ga = each([&](int z) { return (const float *)hs[z]->convs[0].gamma; });
    auto be = each([&](int z) { return (const float *)hs[z]->convs[0].beta; });
    auto sc = each([&](int z) { return bs[z]->ssA[0]; });
    auto sh = each([&](int z) { return bs[z]->ssA[1]; });
    HIPCHK(m, launch_gn_finalize2(st.v, B, a.slots, 32, 32, stem.groups, (long)m->Hs * m->Ws, ga.v, be.v, 1e-5f, sc.v, sh.v, nm, s));
  }
  // (a7) GN + ReLU + maxpool
  int cur = 0;
  {
    auto x = each([&](int z) { return (const unsigned short *)bs[z]->stem_raw; });
    auto sc = each([&](int z) { return (const float *)bs[z]->ssA[0]; });
    auto sh = each([&](int z) { return (const float *)bs[z]->ssA[1]; });
    auto o = each([&](int z) { return bs[z]->bufY[0]; });
    PnvoTimed t(m, s, "bf16:gn_relu_maxpool", 0.0, 2.0 * nm * B * ((double)m->Hs * m->Ws + (double)m->Hp * m->Wp) * 32);
    HIPCHK(m, launch_gn_relu_maxpool_bf16(x.v, sc.v, sh.v, B, m->Hs, m->Ws, 32, o.v, nm, s));
  }

  // one conv of the residual stages + its GroupNorm finalisation.  mode 0: plain input; 1: relu(GN(x)) of the producer
  // (ssA); 2: the fused BasicBlock tail relu(GN2(rawB) + skip) — computed while staging and written to `xout`
  struct Skip {
    bool on = false, affine = false;     // affine: the skip branch is the raw downsample conv (GN applied in the fetch)
    int buf = 0;                         // else: the block input bufY[buf]
  };
  const bool fuse = m->opt.bf16_fuse != 0;
  // ride (>= 0): the layer index of the block's 1x1 stride-2 downsample conv, computed by this launch (conv_bf16_kernel DSF): raw output ->
  // rawD, GroupNorm -> ssD
  auto conv = [&](size_t li, int mode, auto xin, auto yout, int ss_sel /*0 A, 1 B, 2 D, 3 C*/, bool f32out, const Skip &sk,
                  int out_buf, long ride = -1) -> int {
    const Layer &l = m->convs[li];
    ConvBArgs a = bs[0]->layers[li].plan;
    a.B = B;
    a.persist_wgs = m->opt.x3_persist ? 3 * m->num_cus : 0;
    for (int z = 0; z < 2; ++z) {
      const int k = z < nm ? z : 0;
      a.x[z] = xin(k);
      a.wpk[z] = bs[k]->layers[li].wpk;
      a.y[z] = yout(k);
      a.stats[z] = bs[k]->stats;
      a.in_scale[z] = mode == 1 ? bs[k]->ssA[0] : mode == 2 ? bs[k]->ssB[0] : nullptr;
      a.in_shift[z] = mode == 1 ? bs[k]->ssA[1] : mode == 2 ? bs[k]->ssB[1] : nullptr;
      a.x2[z] = mode == 2 ? (sk.affine ? bs[k]->rawD : bs[k]->bufY[sk.buf]) : nullptr;
      a.in_scale2[z] = mode == 2 && sk.affine ? bs[k]->ssD[0] : nullptr;
      a.in_shift2[z] = mode == 2 && sk.affine ? bs[k]->ssD[1] : nullptr;
      a.xout[z] = mode == 2 && out_buf >= 0 ? bs[k]->bufY[out_buf] : nullptr;
      if (ride >= 0) {
        a.ds_wpk[z] = bs[k]->layers[ride].wpk;
        a.ds_y[z] = bs[k]->rawD;
        a.ds_stats[z] = bs[k]->stats_ds;
      }
    }
    // one tile per sample (the 12 x 22 and 6 x 11 maps): the workgroup that sums a sample's channels finalises its GroupNorm too
    // (gn_finalize_lane: fp64, the butterfly order of the float32 path's fused finalisation) — one launch less per layer
    const int cpg = l.groups > 0 ? l.cout / l.groups : 0;
    const bool gfuse = m->opt.gn_fuse && a.slots == 1 && l.cout == l.coutp && cpg >= 1 && cpg <= 32 && 32 % cpg == 0 && l.cout % cpg == 0;
    auto ssof = [&](int z) { return ss_sel == 0 ? bs[z]->ssA : ss_sel == 1 ? bs[z]->ssB : ss_sel == 2 ? bs[z]->ssD : bs[z]->ssC; };
    for (int z = 0; z < 2; ++z) {
      const int k = z < nm ? z : 0;
      a.gn_gamma[z] = gfuse ? hs[k]->convs[li].gamma : nullptr;
      a.gn_beta[z] = gfuse ? hs[k]->convs[li].beta : nullptr;
      a.gn_scale[z] = gfuse ? ssof(k)[0] : nullptr;
      a.gn_shift[z] = gfuse ? ssof(k)[1] : nullptr;
      if (ride >= 0) {
        a.ds_gamma[z] = gfuse ? hs[k]->convs[ride].gamma : nullptr;
        a.ds_beta[z] = gfuse ? hs[k]->convs[ride].beta : nullptr;
        a.ds_scale[z] = gfuse ? bs[k]->ssD[0] : nullptr;
        a.ds_shift[z] = gfuse ? bs[k]->ssD[1] : nullptr;
      }
    }
    a.gn_cpg = cpg;
    a.gn_eps = 1e-5f;
    a.gn_P = (long)l.hout * l.wout;
    const double macs = (double)B * l.hout * l.wout * l.cout * l.cin * l.k * l.kw;
    {
      PnvoTimed t(m, s, "bf16:conv:" + l.name, 2.0 * nm * macs,
                  2.0 * nm * ((double)B * l.hin * l.win * l.cin * (mode == 2 ? 3 : 1) + (double)B * l.hout * l.wout * l.cout));
      HIPCHK(m, launch_conv_bf16(a, l.k, l.stride, mode, f32out, bs[0]->layers[li].mw, bs[0]->layers[li].nw, bs[0]->layers[li].lds,
                                 nm, s));
    }
    if (gfuse) return PNVO_OK;
    PnvoTimed t(m, s, "bf16:gn_finalize", 0.0, 0.0);
    auto st = each([&](int z) { return (const float *)bs[z]->stats; });
    auto ga = each([&](int z) { return (const float *)hs[z]->convs[li].gamma; });
    auto be = each([&](int z) { return (const float *)hs[z]->convs[li].beta; });
    auto sc = each([&](int z) { return ssof(z)[0]; });
    auto sh = each([&](int z) { return ssof(z)[1]; });
    HIPCHK(m, launch_gn_finalize2(st.v, B, a.slots, l.coutp, l.cout, l.groups, (long)l.hout * l.wout, ga.v, be.v, 1e-5f, sc.v, sh.v,
                                  nm, s));
    if (ride >= 0) {                     // the riding downsample conv's GroupNorm (same geometry, its own sums and parameters)
      auto std_ = each([&](int z) { return (const float *)bs[z]->stats_ds; });
      auto gad = each([&](int z) { return (const float *)hs[z]->convs[ride].gamma; });
      auto bed = each([&](int z) { return (const float *)hs[z]->convs[ride].beta; });
      auto scd = each([&](int z) { return bs[z]->ssD[0]; });
      auto shd = each([&](int z) { return bs[z]->ssD[1]; });
      HIPCHK(m, launch_gn_finalize2(std_.v, B, a.slots, l.coutp, l.cout, l.groups, (long)l.hout * l.wout, gad.v, bed.v, 1e-5f, scd.v, shd.v,
                                    nm, s));
    }
    return PNVO_OK;
  };

  // (a8) residual stages (BasicBlock: conv3x3(s) -> GN -> ReLU -> conv3x3 -> GN; + identity / conv1x1(s)+GN; ReLU).
  // The tail of block k (GN2 + skip + ReLU) is not a pass of its own: conv1 of block k+1 (and the compression conv after
  // the last block) computes it while staging its input patch and writes the block output for the later readers (the
  // next skip branch / downsample conv).  PNVO_BF16_NOFUSE=1 keeps the separate residual kernel (A/B measurements).
  size_t li = 1;
  Skip pend;                                       // tail of the previous block, still to be applied
  const Skip none;
  for (int stage = 1; stage <= 4; ++stage)
    for (int bi = 0; bi < m->nblocks[stage - 1]; ++bi) {
      const size_t l1 = li++, l2 = li++;
      const bool ds = li < m->convs.size() && m->convs[li].name.find("downsample") != std::string::npos;
      const Layer &c2 = m->convs[l2];
      // the block's downsample conv rides on its first conv's launch (option ds_fuse); in the block-tail mode the block input is then
      // not written at all: both of its readers are that launch
      const Layer &c1 = m->convs[l1];
      const bool ride_ok = ds && m->opt.ds_fuse && !(pend.on && pend.affine) /* the stager would read rawD while the ride writes it */ &&
                           bs[0]->layers[l1].nw == 1 /* two N-tiles per wave: the second accumulator set does not fit (measured 56 -> 103 us) */ &&
                           c1.cinp != 32 /* 32 input channels: the resident-weight persistent form takes the head (140 + 39 us with the separate
                                            downsample conv against 235 us with the ride on the streaming form) */ &&
                           c1.k == 3 && c1.stride == 2 && m->convs[li].k == 1 && m->convs[li].stride == 2 &&
                           c1.cinp == m->convs[li].cinp && c1.coutp == m->convs[li].coutp && c1.hout == m->convs[li].hout &&
                           c1.wout == m->convs[li].wout && c1.groups == m->convs[li].groups;
      const long ride = ride_ok ? (long)li : -1;
      if (pend.on) {                               // input = relu(GN2(rawB) + skip) of the previous block -> bufY[cur ^ 1]
        if ((rc = conv(l1, 2, [&](int z) { return (const unsigned short *)bs[z]->rawB; }, [&](int z) { return (void *)bs[z]->rawA; }, 0,
                       false, pend, ride_ok ? -1 : (cur ^ 1), ride)) != PNVO_OK)
          return rc;
        cur ^= 1;
      } else if ((rc = conv(l1, 0, [&](int z) { return (const unsigned short *)bs[z]->bufY[cur]; },
                            [&](int z) { return (void *)bs[z]->rawA; }, 0, false, none, -1, ride)) != PNVO_OK) {
        return rc;
      }
      if ((rc = conv(l2, 1, [&](int z) { return (const unsigned short *)bs[z]->rawA; }, [&](int z) { return (void *)bs[z]->rawB; }, 1,
                     false, none, -1)) != PNVO_OK)
        return rc;
      const long P = (long)c2.hout * c2.wout;
      if (ds) {
        const size_t ld = li++;
        if (!ride_ok && (rc = conv(ld, 0, [&](int z) { return (const unsigned short *)bs[z]->bufY[cur]; },
                                   [&](int z) { return (void *)bs[z]->rawD; }, 2, false, none, -1)) != PNVO_OK)
          return rc;
      }
      if (fuse) {
        pend.on = true;
        pend.affine = ds;
        pend.buf = cur;
        continue;
      }
      auto a_ = each([&](int z) { return (const unsigned short *)bs[z]->rawB; });
      auto sa = each([&](int z) { return (const float *)bs[z]->ssB[0]; });
      auto ta = each([&](int z) { return (const float *)bs[z]->ssB[1]; });
      auto y_ = each([&](int z) { return bs[z]->bufY[cur ^ 1]; });
      if (ds) {
        auto b_ = each([&](int z) { return (const unsigned short *)bs[z]->rawD; });
        auto sb = each([&](int z) { return (const float *)bs[z]->ssD[0]; });
        auto tb = each([&](int z) { return (const float *)bs[z]->ssD[1]; });
        PnvoTimed t(m, s, "bf16:residual", 0.0, 6.0 * nm * B * P * c2.coutp);
        HIPCHK(m, launch_residual_bf16(a_.v, sa.v, ta.v, b_.v, sb.v, tb.v, B, P, c2.coutp, y_.v, nm, s));
      } else {
        auto b_ = each([&](int z) { return (const unsigned short *)bs[z]->bufY[cur]; });
        PnvoTimed t(m, s, "bf16:residual", 0.0, 6.0 * nm * B * P * c2.coutp);
        HIPCHK(m, launch_residual_bf16(a_.v, sa.v, ta.v, b_.v, nullptr, nullptr, B, P, c2.coutp, y_.v, nm, s));
      }
      cur ^= 1;
    }
  // (a10) compression conv + GroupNorm(1, C): fp32 output for the Linear layers
  if (pend.on) {
    if ((rc = conv(li, 2, [&](int z) { return (const unsigned short *)bs[z]->rawB; }, [&](int z) { return (void *)bs[z]->comp_raw; }, 3,
                   true, pend, -1)) != PNVO_OK)
      return rc;
  } else if ((rc = conv(li, 0, [&](int z) { return (const unsigned short *)bs[z]->bufY[cur]; },
                        [&](int z) { return (void *)bs[z]->comp_raw; }, 3, true, none, -1)) != PNVO_OK) {
    return rc;
  }
  // (a11) Flatten + Linear + ReLU + output head: the fp32 split-K linear kernels on each model
  for (int z = 0; z < nm; ++z) {
    pnvo_handle h = hs[z];
    const int tm = h->timing;
    h->timing = 0;                                  // (their event records belong to the fp32 path's table)
    // (the output head rides on the hidden layer's split-K reduction when there is one: option head_fuse, as in the float32 forward)
    h->head_rode = false;
    h->head_ride_w = h->train != nullptr ? nullptr : h->head_w_plain;
    h->head_ride_out = (h->opt.head_fuse && c.out_dim <= 4 && h->head_ride_w != nullptr) ? outs[z] : nullptr;
    rc = pnvo_run_conv(h, h->fc, B, bs[z]->comp_raw, bs[z]->ssC[0], bs[z]->ssC[1], bs[z]->hid, c.hidden, nullptr, h->fc_bias,
                       c.act_embed ? actions : nullptr, 1, s, nullptr, nullptr, nullptr);
    h->head_ride_out = nullptr;
    if (rc == PNVO_OK && !h->head_rode)
      rc = pnvo_run_conv(h, h->head, B, bs[z]->hid, nullptr, nullptr, outs[z], c.out_dim, nullptr, h->head_bias, nullptr, 0, s,
                         nullptr, nullptr, nullptr);
    h->timing = tm;
    if (rc != PNVO_OK) return z == 0 ? rc : pnvo_fail(m, rc, h->err);
  }
  return PNVO_OK;
}
