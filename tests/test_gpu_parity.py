"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes -> libpnvo.so), against
  (1) the committed golden fixtures captured from the imported reference,
  (2) the oracle on the same seeded inputs,
  (3) size-independent properties at BASELINE sizes (batch-composition invariance, determinism).
Tolerance for the network outputs (fp32): per-pair ||out-ref||_2 / max(||ref||_2, 1e-2) < 1e-4 vs the fp64 reference
(BASELINE.md §5).  Pre-processing (integer/bin work): bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import MODEL_FIXTURES, golden_case, load_golden, pair_rel_err
from oracle import oracle
from pointnav_vo_amd import _lib, synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd import vo_cnn  # noqa: F401
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO, NormalizedDepth2TopDownViewHabitatTorch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def build(rec):
    cfg, sd, obs, actions = golden_case(rec)
    space = str(rec["obs_space"]).split(",")
    kw = dict(observation_space=space, observation_size=(cfg.width, cfg.height), hidden_size=512,
              backbone=str(rec["backbone"]) if "backbone" in rec else "resnet18", normalize_visual_inputs=True,
              output_dim=3, dropout_p=0.2)
    if int(rec["dd_bins"]):
        kw["discretized_depth_channels"] = int(rec["dd_bins"])
    model = baseline_registry.get_vo_model(str(rec["model"]))(**kw)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev()).eval()
    tobs = {k: torch.from_numpy(v).to(dev()) for k, v in obs.items()}
    tact = torch.from_numpy(actions).to(dev()) if actions is not None else None
    return model, cfg, sd, obs, tobs, actions, tact


def test_native_library_is_the_one_running():
    assert "gfx950" in _lib.version()
    with open("/proc/self/maps") as f:
        assert "libpnvo.so" in f.read()


@pytest.mark.parametrize("fname", MODEL_FIXTURES)
def test_forward_matches_reference_golden(fname):
    rec = load_golden(fname)
    model, cfg, sd, obs, tobs, actions, tact = build(rec)
    with torch.no_grad():
        out = (model(tobs, tact) if tact is not None else model(tobs)).cpu().numpy()
    assert out.shape == rec["out64"].shape and np.isfinite(out).all()
    err = pair_rel_err(out, rec["out64"])
    assert err.max() < TOL, (fname, err)
    # and against the oracle evaluated here on the regenerated inputs
    ref = oracle.forward(sd, obs, ngroups=cfg.ngroups, dtype=np.float64, actions=actions)
    assert pair_rel_err(out, ref).max() < TOL


def test_every_intermediate_activation_matches_reference():
    rec = load_golden("model_default_45x37_b3.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    names = [k[4:] for k in rec if k.startswith("tap/")]
    assert len(names) >= 12
    for name in names:
        want = rec[f"tap/{name}"]
        with torch.no_grad():
            _, got = model.tap(name, tobs)
        got = got.cpu().numpy()
        if name == "hidden":
            got = got.reshape(want.shape[0], -1)
        got = got[..., : want.shape[-1]]
        assert got.shape == want.shape, (name, got.shape, want.shape)
        scale = np.abs(want).max() + 1e-6
        assert np.abs(got - want).max() / scale < 2e-5, (name, np.abs(got - want).max(), scale)


def test_deterministic_and_batch_composition_invariant():
    rec = load_golden("model_default_341x192_b2.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    big = synth.make_obs_pairs(5, cfg.height, cfg.width, observation_space=str(rec["obs_space"]).split(","), seed=77)
    tbig = {k: torch.from_numpy(v).to(dev()) for k, v in big.items()}
    with torch.no_grad():
        a = model(tbig).cpu().numpy()
        b = model(tbig).cpu().numpy()
        np.testing.assert_array_equal(a, b)                       # fixed-order reductions: bit-reproducible
        one = model({k: v[3:4] for k, v in tbig.items()}).cpu().numpy()
        perm = [4, 2, 0, 3, 1]
        p = model({k: v[perm] for k, v in tbig.items()}).cpu().numpy()
    assert pair_rel_err(one, a[3:4]).max() < 1e-5                 # pairs are independent (GroupNorm is per sample)
    assert pair_rel_err(p, a[perm]).max() < 1e-5


def test_full_batch_256_properties():
    """BASELINE configs[1] size: B=256 at 341x192.  Checked through size-independent properties: every pair of the
    big batch equals the same pair evaluated in a small batch, and the eight pairs of bench.ORACLE_PAIRS equal the fp64 oracle."""
    import bench
    d = dev()
    model, sd = bench.build_model(d)
    obs = bench.make_inputs(256, d, rank=0)
    with torch.no_grad():
        out = model(obs)
        idx = [0, 1, 100, 255]
        sub = model({k: v[idx] for k, v in obs.items()})
        torch.cuda.synchronize()
    out, sub = out.cpu().numpy(), sub.cpu().numpy()
    assert np.isfinite(out).all()
    assert pair_rel_err(sub, out[idx]).max() < 1e-5
    chk = list(bench.ORACLE_PAIRS)                 # the same 8 pairs bench.py checks: first / last tiles, mid-batch, both halves
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups,
                         dtype=np.float64)
    assert pair_rel_err(out[chk], ref).max() < TOL
    # the synthetic dd/tdv inputs were produced by the HIP pre-processing kernels: check them against the oracle too
    d0 = obs["depth"][0].cpu().numpy()
    dd_ref, _ = oracle.discretize_depth(d0[..., 0], 10)
    np.testing.assert_array_equal(obs["discretized_depth"][0, ..., :10].cpu().numpy(), dd_ref)
    c = oracle.topdown_consts(192, 341, 70, 0.1, 10.0)
    np.testing.assert_array_equal(obs["top_down_view"][0, ..., 1].cpu().numpy(), oracle.topdown_view(d0[..., 1], c)[..., 0])


def test_three_stems_agree():
    """The default model's stem (option stem=auto) runs on the float16 matrix cores: raw inputs exact in float16, the folded
    float32 weight split into TWO float16 pieces (22-bit operands, pieces=2) — stem_mx_kernel at this batch of 2 pairs, the
    resident-weight stem_rs_kernel from 8 pairs on (tests/test_gpu_knobs.py compares those two bit for bit).  Option stem=dd selects
    the one-hot table-gather stem (stem_dd.hip), stem=dense the all-fp32-MFMA stem (stem_lds.hip), both exact float32 products.  The
    three must agree to float32-grade noise (2e-6 of the output range) at every output pixel, borders (zero padding after
    whitening) included, and match the reference."""
    rec = load_golden("model_default_341x192_b2.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    with torch.no_grad():
        out_mx, stem_mx = model.tap("stem_conv", tobs)
        assert model.get_option("stem") == "auto"
        model.set_option("stem", "dd")
        out_dd, stem_dd = model.tap("stem_conv", tobs)
        model.set_option("stem", "dense")
        out_dense, stem_dense = model.tap("stem_conv", tobs)
        model.set_option("stem", "auto")
        model.check_inputs()
    a, d, b = stem_mx.cpu().numpy(), stem_dd.cpu().numpy(), stem_dense.cpu().numpy()
    assert a.shape == b.shape == d.shape and np.abs(b).max() > 0.1
    assert not np.array_equal(a, b) and not np.array_equal(d, b) and not np.array_equal(a, d), "knob did not select another kernel"
    assert np.abs(a - b).max() / np.abs(b).max() < 2e-6
    assert np.abs(d - b).max() / np.abs(b).max() < 2e-6
    want = rec["tap/stem_conv"] if "tap/stem_conv" in rec else None
    if want is not None:
        for x in (a, d, b):
            assert np.abs(x[..., : want.shape[-1]] - want).max() / np.abs(want).max() < 2e-5
    for o in (out_mx, out_dd, out_dense):
        assert pair_rel_err(o.cpu().numpy(), rec["out64"]).max() < TOL


def _contract_breakers(tobs):
    frac = dict(tobs)
    frac["rgb"] = tobs["rgb"].clone()
    frac["rgb"][0, 3, 4, 1] = 17.3
    frac["rgb"][2, 20, 11, 5] = 200.125
    soft = dict(tobs)
    soft["discretized_depth"] = tobs["discretized_depth"].clone()
    soft["discretized_depth"][1, 5, 7, :] = 0.1
    return {"fractional rgb": frac, "soft depth code": soft}


@pytest.mark.parametrize("kind", ["fractional rgb", "soft depth code"])
@pytest.mark.parametrize("stem", ["auto", "dd"])
def test_input_outside_the_stem_contract_is_rerun_on_the_dense_stem(kind, stem):
    """The fused stems rely on uint8-valued rgb and one-hot depth (the reference's own observation construction,
    base_trainer_with_vo.py:163,196-207); the reference MODEL takes any float tensor (vo_cnn.py:110-176).  A forward that
    meets a value outside the contract must return the CORRECT result from that very call (re-run on the dense fp32 stem),
    leave the handle usable — now on the dense stem — and say so once."""
    rec = load_golden("model_default_45x37_b3.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    if stem != "auto":
        model.set_option("stem", stem)
    bad = _contract_breakers(tobs)[kind]
    with torch.no_grad():
        clean0 = model(tobs).cpu().numpy()
        assert model.get_option("stem") == stem
        out = model(bad).cpu().numpy()                    # the offending call itself
        ref = oracle.forward(sd, {k: v.cpu().numpy() for k, v in bad.items()}, ngroups=cfg.ngroups, dtype=np.float64)
        assert pair_rel_err(out, ref).max() < TOL, pair_rel_err(out, ref)
        # (the one-hot stem multiplies rgb on the fp32 pipe: fractional rgb is inside ITS contract, no fallback needed)
        fell_back = not (stem == "dd" and kind == "fractional rgb")
        assert model.get_option("stem") == ("dense (fallback)" if fell_back else "dd")
        assert ("dense" in model.last_note()) == fell_back
        model.check_inputs()                              # nothing pending: the handle is not poisoned
        again = model(bad).cpu().numpy()                  # stays correct, no second fallback needed
        np.testing.assert_array_equal(again, out)
        clean1 = model(tobs).cpu().numpy()                # and contract inputs still evaluate (dense stem now)
    assert pair_rel_err(clean1, rec["out64"]).max() < TOL and pair_rel_err(clean0, rec["out64"]).max() < TOL
    # an explicit stem choice lifts the fallback
    model.set_option("stem", "mx" if stem == "auto" else stem)
    with torch.no_grad():
        np.testing.assert_array_equal(model(tobs).cpu().numpy(), clean0)


def test_input_fallback_off_reports_instead():
    """input_fallback=off keeps the forward free of any host wait: the stem raises a flag, check_inputs() and every later
    forward fail with PNVO_ERR_INPUT (never a silently rounded result that looks fine)."""
    rec = load_golden("model_default_45x37_b3.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    model.set_option("input_fallback", "off")
    bad = _contract_breakers(tobs)["soft depth code"]
    with torch.no_grad():
        model(tobs)
        torch.cuda.synchronize()
        model.check_inputs()                                       # clean input: no complaint
        model(bad)
        torch.cuda.synchronize()
        with pytest.raises(_lib.PnvoError, match="one-hot"):
            model.check_inputs()
        with pytest.raises(_lib.PnvoError, match="one-hot"):
            model(tobs)


def test_unknown_option_is_refused():
    rec = load_golden("model_default_45x37_b3.npz")
    model, *_ = build(rec)
    model.get_option("conv")
    with pytest.raises(_lib.PnvoError, match="unknown option"):
        model.set_option("no_such_knob", "1")
    with pytest.raises(_lib.PnvoError, match="mx"):
        model.set_option("stem", "sparse")


def test_soft_depth_codes_on_the_dense_stem():
    rec = load_golden("model_default_45x37_b3.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    model.set_option("stem", "dense")
    soft = {k: v.copy() for k, v in obs.items()}
    soft["discretized_depth"] = (0.8 * soft["discretized_depth"] + 0.02).astype(np.float32)
    with torch.no_grad():
        out = model({k: torch.from_numpy(v).to(dev()) for k, v in soft.items()}).cpu().numpy()
        model.check_inputs()
    ref = oracle.forward(sd, soft, ngroups=cfg.ngroups, dtype=np.float64)
    assert pair_rel_err(out, ref).max() < TOL


# ----------------------------------------------------------------------------- pre-processing (bit-exact)
def test_discretize_depth_bit_exact():
    rec = load_golden("preproc.npz")
    depth = torch.from_numpy(rec["dd_depth"]).to(dev())
    t = BaseRLTrainerWithVO(AttrDict(VO=dict(REGRESS_MODEL=dict(discretized_depth_channels=int(rec["dd_bins"])))), dev())
    dd = t._discretize_depth_func(depth).cpu().numpy()
    assert dd.shape == rec["dd_depth"].shape + (10,)
    assert dd.sum() == depth.numel()                              # the reference's assert (:163)
    np.testing.assert_array_equal(dd.argmax(-1).astype(np.uint8), rec["dd_index"])
    with pytest.raises(AssertionError):
        t._discretize_depth_func(depth + 1.5)


TDV_CASES = ["full_uniform", "full_border", "full_near", "full_fp32", "full_zero", "full_one_pixel",
             "full_top_band", "small_odd", "small_border"]


@pytest.mark.parametrize("case", TDV_CASES)
def test_topdown_view_bit_exact(case):
    rec = load_golden("preproc.npz")
    d = rec[f"tdv_in/{case}"].astype(np.float32)
    H, W = d.shape[:2]
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    np.testing.assert_array_equal(np.array(gen._consts[:7], dtype=np.float32), rec[f"tdv_consts/{H}x{W}"][:7])
    out = gen.gen_top_down_view(torch.from_numpy(d).to(dev())).cpu().numpy()
    assert out.shape == (H, W, 1)
    want = np.zeros(H * W, np.float32)
    want[rec[f"tdv_nz/{case}"]] = rec[f"tdv_val/{case}"]
    np.testing.assert_array_equal(out.reshape(-1), want)


def test_topdown_view_batched_strided_matches_single():
    H, W = 192, 341
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    pair = torch.from_numpy(np.stack([synth.make_raw_obs(H, W, seed=9, index=i, zero_border=3 * (i % 2))["depth"][..., 0]
                                      for i in range(6)]).reshape(3, 2, H, W).transpose(0, 2, 3, 1).copy()).to(dev())
    tdv = torch.zeros((3, H, W, 2), device=dev())
    for k in range(2):
        gen.gen_top_down_view_batch(pair[..., k], out=tdv, out_channel=k)
    c = oracle.topdown_consts(H, W, 70, 0.1, 10.0)
    for b in range(3):
        for k in range(2):
            np.testing.assert_array_equal(tdv[b, ..., k].cpu().numpy(), oracle.topdown_view(pair[b, ..., k].cpu().numpy(), c)[..., 0])


def test_topdown_views_of_frame_pairs_in_one_pass_match_the_oracle():
    """pnvo_topdown_view_pairs: frames [n,2,H,W] -> pair tensor [n,H,W,2] in one pass of the three kernels (what the boundary
    call uses), bit-exact against the oracle and against the per-channel calls."""
    H, W = 192, 341
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    frames = torch.from_numpy(np.stack([synth.make_raw_obs(H, W, seed=4, index=i, zero_border=2 * (i % 3))["depth"][..., 0]
                                        for i in range(10)]).reshape(5, 2, H, W).copy()).to(dev())
    tdv = torch.full((5, H, W, 2), -1.0, device=dev())
    gen.gen_top_down_view_pairs(frames, tdv)
    ref = torch.zeros((5, H, W, 2), device=dev())
    for k in range(2):
        gen.gen_top_down_view_batch(frames[:, k], out=ref, out_channel=k)
    assert torch.equal(tdv, ref)
    c = oracle.topdown_consts(H, W, 70, 0.1, 10.0)
    for b in (0, 4):
        for k in range(2):
            np.testing.assert_array_equal(tdv[b, ..., k].cpu().numpy(), oracle.topdown_view(frames[b, k].cpu().numpy(), c)[..., 0])


# ----------------------------------------------------------------------------- the drop-in boundary (a1 / a13)
def make_trainer(rec):
    cfg = AttrDict(
        VO=dict(VO_TYPE="REGRESS", OBS_TRANSFORM="none", VIS_SIZE_W=int(rec["width"]), VIS_SIZE_H=int(rec["height"]),
                REGRESS_MODEL=dict(name="vo_cnn_rgb_d_dd_top_down", visual_backbone="resnet18", hidden_size=512,
                                   visual_type=["rgb", "depth", "discretized_depth", "top_down_view"], dropout_p=0.2,
                                   discretize_depth="hard", discretized_depth_channels=int(rec["bins"]),
                                   regress_type="sep_act", mode="det", rnd_mode_n=10, pretrained=False)),
        TASK_CONFIG=dict(SIMULATOR=dict(DEPTH_SENSOR=dict(MIN_DEPTH=0.1, MAX_DEPTH=10.0, HFOV=70))))
    t = BaseRLTrainerWithVO(cfg, dev())
    t._set_up_vo_obs_transformer()
    t._setup_vo_model(cfg)
    from pointnav_vo_amd import model_spec as ms
    for k in ("forward", "left", "right"):
        sd = synth.make_state_dict(ms.state_dict_spec(t.vo_model[k].cfg), seed=int(rec[f"seed_{k}"]))
        t.vo_model[k].load_state_dict({n: torch.from_numpy(np.array(v)) for n, v in sd.items()})
    return t


def test_compute_local_delta_states_from_vo_matches_reference():
    rec = load_golden("boundary.npz")
    t = make_trainer(rec)
    assert list(t.vo_model.keys()) == ["forward", "left", "right"]       # base_trainer_with_vo.py:62
    H, W = int(rec["height"]), int(rec["width"])
    prevs, curs, acts = [], [], []
    for (pi, ci, act, zb), want in zip(rec["steps"], rec["deltas"]):
        prev = synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(pi), zero_border=int(zb))
        cur = synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(ci), zero_border=int(zb))
        deltas, std, extra = t._compute_local_delta_states_from_vo(prev, cur, int(act))
        assert isinstance(deltas, list) and len(deltas) == 3 and std == [0, 0, 0] and extra == {}
        assert pair_rel_err(np.array(deltas)[None], want[None]).max() < TOL, (deltas, want)
        prevs.append(prev), curs.append(cur), acts.append(int(act))
    # batched sibling: same numbers in one call
    batch = t.compute_local_delta_states_batch(prevs, curs, acts)
    assert pair_rel_err(batch, rec["deltas"]).max() < TOL


def test_boundary_with_pretrained_reference_checkpoints(tmp_path):
    """a13, pretrained branch (base_trainer_with_vo.py:83-99): the three action models come from reference-format files —
    `forward` from a {"model_state"} file, `left` and `right` from ONE engine-format {"model_states"} file that also holds
    a config object, optimizer and RNG states — and the boundary call reproduces the reference's deltas."""
    from test_checkpoint import trainer_cfg, write_reference_checkpoints
    from pointnav_vo_amd import model_spec as ms
    rec = load_golden("boundary.npz")
    H, W, bins = int(rec["height"]), int(rec["width"]), int(rec["bins"])
    mcfg = ms.config_from_kwargs(observation_space=["rgb", "depth", "discretized_depth", "top_down_view"], observation_size=(W, H),
                                 hidden_size=512, normalize_visual_inputs=True, output_dim=3, discretized_depth_channels=bins)
    paths = write_reference_checkpoints(str(tmp_path), mcfg, {k: int(rec[f"seed_{k}"]) for k in ("forward", "left", "right")})
    t = BaseRLTrainerWithVO(trainer_cfg(W, H, bins, paths), dev())
    t._set_up_vo_obs_transformer()
    t._setup_vo_model(t.config)
    prevs, curs, acts = [], [], []
    for (pi, ci, act, zb) in rec["steps"]:
        prevs.append(synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(pi), zero_border=int(zb)))
        curs.append(synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(ci), zero_border=int(zb)))
        acts.append(int(act))
    batch = t.compute_local_delta_states_batch(prevs, curs, acts)
    assert pair_rel_err(batch, rec["deltas"]).max() < TOL


def test_rnd_mode_samples_dropout_and_updates_running_stats():
    """VO.REGRESS_MODEL.mode == 'rnd' (base_trainer_with_vo.py:295-308): rnd_mode_n forwards with the model in train()
    mode — dropout active, and (a quirk the reference has and a drop-in must keep) RunningMeanAndVar updated by every
    one of them.  Returns the sample mean and a non-zero per-component std."""
    rec = load_golden("boundary.npz")
    t = make_trainer(rec)
    t.config.VO.REGRESS_MODEL.mode = "rnd"
    t.config.VO.REGRESS_MODEL.rnd_mode_n = 6
    H, W = int(rec["height"]), int(rec["width"])
    pi, ci, act, zb = rec["steps"][0]
    prev = synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(pi), zero_border=int(zb))
    cur = synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(ci), zero_border=int(zb))
    from pointnav_vo_amd.common_vars import ACT_IDX2NAME
    model = t.vo_model[ACT_IDX2NAME[int(act)]]
    cnt0 = float(model.visual_encoder.running_mean_and_var._count)
    deltas, std, extra = t._compute_local_delta_states_from_vo(prev, cur, int(act))
    assert len(deltas) == 3 and len(std) == 3 and np.isfinite(deltas).all()
    assert min(std) > 0.0                                   # six different dropout draws
    assert float(model.visual_encoder.running_mean_and_var._count) == cnt0 + 6   # 6 train-mode forwards of 1 pair
    # and the deterministic mode still works on the same trainer afterwards (eval forward, no dropout)
    t.config.VO.REGRESS_MODEL.mode = "det"
    d2, s2, _ = t._compute_local_delta_states_from_vo(prev, cur, int(act))
    assert s2 == [0, 0, 0] and np.isfinite(d2).all()


# ----------------------------------------------------------------------------- dense float32 depth (the path's real input)
F32_OPTS = [{}, {"conv": "x3"}, {"pieces": "3"}, {"conv": "x3", "pieces": "3"}, {"conv": "fp32", "stem": "dense"}]


@pytest.mark.parametrize("opts", F32_OPTS, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()) or "default")
def test_dense_float32_depth_matches_reference(opts):
    """Every other model fixture has float16-exact depth (the dataset's storage format), for which the low float16 piece of the
    stem's depth operand is identically zero.  The navigation loop hands over float32 simulator depth
    (base_trainer_with_vo.py:177-190): this fixture's depth is uniform float32, so both pieces of every float-valued stem channel
    are live; checked on the default kernels, with the split 3x3 kernels forced at this size, and on the exact three-piece forms."""
    rec = load_golden("model_default_341x192_b2_f32depth.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    d = obs["depth"]
    assert not np.array_equal(d.astype(np.float16).astype(np.float32), d)        # the depth really is not float16-exact
    for k, v in opts.items():
        model.set_option(k, v)
    with torch.no_grad():
        out = model(tobs).cpu().numpy()
        model.set_option("small_net", "off")                                     # (2 pairs: also through the per-layer kernels)
        out_l = model(tobs).cpu().numpy()
    assert pair_rel_err(out, rec["out64"]).max() < TOL, pair_rel_err(out, rec["out64"])
    assert pair_rel_err(out_l, rec["out64"]).max() < TOL, pair_rel_err(out_l, rec["out64"])


@pytest.mark.parametrize("opts", [{}, {"conv": "x3"}, {"pieces": "3"}], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()) or "default")
def test_boundary_with_dense_float32_depth(opts):
    """_compute_local_delta_states_from_vo on simulator-style float32 depth frames (not float16-exact): per-pair method, batched
    sibling (sensor-frame entry) against the reference's deltas."""
    rec = load_golden("boundary_f32depth.npz")
    assert int(rec["depth_fp16"]) == 0
    t = make_trainer(rec)
    for m in t.vo_model.values():
        for k, v in opts.items():
            m.set_option(k, v)
    H, W = int(rec["height"]), int(rec["width"])
    prevs, curs, acts = [], [], []
    for (pi, ci, act, zb), want in zip(rec["steps"], rec["deltas"]):
        prev = synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(pi), zero_border=int(zb), depth_fp16=False)
        cur = synth.make_raw_obs(H, W, seed=int(rec["obs_seed"]), index=int(ci), zero_border=int(zb), depth_fp16=False)
        deltas, std, extra = t._compute_local_delta_states_from_vo(prev, cur, int(act))
        assert pair_rel_err(np.array(deltas)[None], want[None]).max() < TOL, (deltas, want)
        prevs.append(prev), curs.append(cur), acts.append(int(act))
    batch = t.compute_local_delta_states_batch(prevs, curs, acts)
    assert pair_rel_err(batch, rec["deltas"]).max() < TOL
