"""GPU (-m gpu): the forms of the float16-piece stem (option stem_form) against the tile-per-workgroup kernel: `resident` (weights in
registers, the tile kernel's summation order: bit-identical on the observation-tensor entry, the sensor-frame entry and with the max-pool
as its own pass) and `fast` (its own order: float32-grade agreement and run-to-run reproducibility); the input-contract check in the
resident kernel's stager."""
import numpy as np
import pytest
import torch

import bench
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd import vo_cnn  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def run(model, obs, form, _unused=None, pool="fused"):
    model.set_option("stem_form", form)
    model.set_option("pool", pool)
    rgb_f, dep_f = bench.frames_of(obs) if obs["depth"].shape[1:3] == (bench.H, bench.W) else (None, None)
    with torch.no_grad():
        a = model(obs).clone()
        b = model(obs).clone()
        r = model.forward_raw(rgb_f, dep_f, obs["top_down_view"]).clone() if rgb_f is not None else a
    torch.cuda.synchronize()
    return a, b, r


@pytest.mark.parametrize("B", [19, 64])
@pytest.mark.parametrize("pool", ["fused", "separate"])
def test_resident_weight_stem_equals_the_tile_kernel(B, pool):
    """stem_rs_kernel (stem_form=resident, the default from 8 pairs of 341x192 on): one 4-wave workgroup per CU keeps the stem's
    weights in registers for all of its tiles; same tap split, fragment order and K-split summation order as the tile kernel — not
    one bit differs, on the observation-tensor entry, on the sensor-frame entry and with the max-pool as its own pass."""
    model, _ = bench.build_model(DEV)
    obs = bench.make_inputs(B, DEV, 0)
    ref = run(model, obs, "tiles", 4, pool)
    res = run(model, obs, "resident", 4, pool)
    auto = run(model, obs, "auto", 4, pool)
    assert torch.isfinite(ref[0]).all()
    fast = run(model, obs, "fast", 4, pool)
    for k in range(3):
        assert torch.equal(ref[k], res[k]), k
        assert torch.equal(fast[k], auto[k]), k                       # (auto = fast from 8 pairs on: float32-grade, see below)


def test_resident_weight_stem_on_an_odd_resolution():
    """45 x 37 (ragged tiles on both edges, patches that leave the image on every side), 700 pairs so that the kernel takes the launch."""
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(45, 37), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(DEV).eval()
    obs = {k: torch.from_numpy(v).to(DEV) for k, v in
           synth.make_obs_pairs(700, 37, 45, observation_space=bench.SPACE, dd_bins=10, seed=3).items()}
    outs = {}
    for form in ("tiles", "resident"):
        m.set_option("stem_form", form)
        with torch.no_grad():
            outs[form] = m(obs).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(outs["tiles"]).all()
    assert torch.equal(outs["tiles"], outs["resident"])


def test_resident_weight_dual_bf16_stem_equals_the_tile_kernel():
    """configs[2]: the bf16 dual stem (one weight piece, two models' 32 channels) with its 196 KB of fragments resident in registers
    (stem_rs_kernel<1, ...>) against stem_mx_kernel<1, 2, true>: both models' outputs bit-identical, observation-tensor and
    sensor-frame entry."""
    from pointnav_vo_amd import vo_cnn as V
    ma, _ = bench.build_model(DEV, seed=0)
    mb, _ = bench.build_model(DEV, seed=1)
    for m in (ma, mb):
        m.set_precision("bfloat16")
    obs = bench.make_inputs(32, DEV, 0)
    outs = {}
    for form in ("tiles", "resident", "auto"):
        for m in (ma, mb):
            m.set_option("stem_form", form)
        with torch.no_grad():
            oa, ob = V.dual_forward(ma, mb, obs)
        torch.cuda.synchronize()
        outs[form] = (oa.clone(), ob.clone())
    assert torch.isfinite(outs["tiles"][0]).all() and torch.isfinite(outs["tiles"][1]).all()
    for form in ("resident", "auto"):
        assert torch.equal(outs["tiles"][0], outs[form][0]) and torch.equal(outs["tiles"][1], outs[form][1]), form
    assert not torch.equal(outs["tiles"][0], outs["tiles"][1])     # (two different models)


@pytest.mark.parametrize("pool", ["fused", "separate"])
def test_fast_resident_stem_is_float32_grade_equal_and_reproducible(pool):
    """stem_form=fast: the resident-weight stem with the remainder MFMAs of four taps sharing one K chunk and tap 48 split over the
    waves by M-tile — another summation order, so float32-grade agreement with the tile kernel (1e-5 of the output's largest
    magnitude; measured 1e-6 .. 4e-6 over models and sizes, tools/stress_stem_rs.py), run-to-run reproducible, and the sensor-frame entry identical to the observation-tensor entry."""
    model, _ = bench.build_model(DEV)
    obs = bench.make_inputs(64, DEV, 0)
    ref = run(model, obs, "tiles", 4, pool)
    fast = run(model, obs, "fast", 4, pool)
    assert torch.isfinite(fast[0]).all()
    assert torch.equal(fast[0], fast[1]) and torch.equal(fast[0], fast[2])
    rel = (ref[0] - fast[0]).abs().max() / ref[0].abs().max()
    assert 0 < rel < 1e-5, rel


VARIANTS = [("vo_cnn", "rgb,depth", 0), ("vo_cnn_rgb", "rgb", 0), ("vo_cnn_rgb_d_dd", "rgb,depth,discretized_depth", 10),
            ("vo_cnn_d_dd_top_down", "depth,discretized_depth,top_down_view", 10)]


@pytest.mark.parametrize("name,space,bins", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_resident_stems_on_models_without_some_modalities(name, space, bins):
    """Registry variants whose stems lack rgb, depth, the one-hot depth or the top-down view (absent tensors are descriptors of zero
    records in stem_rs_kernel): 600 pairs at 64 x 48 so that the resident kernel takes the launch; `resident` bit-identical to the tile
    kernel, `fast` float32-grade equal, and the tile kernel itself within 2e-5 of the fp64 oracle on the first pairs."""
    from oracle import oracle
    W, H, B = 64, 48, 600
    kw = dict(observation_space=space.split(","), observation_size=(W, H), hidden_size=512, backbone="resnet18",
              normalize_visual_inputs=True, output_dim=3, dropout_p=0.2)
    if bins:
        kw["discretized_depth_channels"] = bins
    model = baseline_registry.get_vo_model(name)(**kw)
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=5)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(DEV).eval()
    obs = synth.make_obs_pairs(B, H, W, observation_space=space.split(","), dd_bins=bins or 10, seed=9)
    tobs = {k: torch.from_numpy(v).to(DEV) for k, v in obs.items()}
    outs = {}
    for form in ("tiles", "resident", "fast"):
        model.set_option("stem_form", form)
        with torch.no_grad():
            outs[form] = model(tobs).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(outs["tiles"]).all()
    assert torch.equal(outs["tiles"], outs["resident"])
    rel = (outs["tiles"] - outs["fast"]).abs().max() / outs["tiles"].abs().max()
    assert rel < 1e-5, rel
    ref = oracle.forward(sd, {k: v[:3] for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    got = outs["fast"][:3].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


@pytest.mark.parametrize("kind", ["fractional rgb", "soft depth code"])
def test_resident_stem_detects_input_outside_its_contract(kind):
    """The contract check (rgb uint8-valued, one-hot depth: exact in float16) lives in the resident kernel's stager too: 32 pairs at
    341 x 192 take stem_rs_kernel; one fractional rgb value / one soft depth code in the LAST pair makes the call re-run on the dense
    stem and return what the dense stem returns (vo_cnn.py:110-176 takes any float tensor)."""
    model, _ = bench.build_model(DEV)
    obs = bench.make_inputs(32, DEV, 0)
    bad = dict(obs)
    if kind == "fractional rgb":
        bad["rgb"] = obs["rgb"].clone()
        bad["rgb"][31, 150, 300, 4] = 17.3
    else:
        bad["discretized_depth"] = obs["discretized_depth"].clone()
        bad["discretized_depth"][31, 100, 7, :] = 0.1
    with torch.no_grad():
        clean = model(obs).clone()
        assert model.get_option("stem") == "auto"
        out = model(bad).clone()                              # the offending call itself: its stem redone on the device
        torch.cuda.synchronize()                              # (the host learns of it at its next entry, once the stem has run)
        assert model.get_option("stem") == "dense (fallback)" and "float32 stem" in model.last_note()
        again = model(bad).clone()                            # the stand-in launched directly: the same kernel, the same bits
        model.set_option("stem", "dense")
        ref = model(bad).clone()                              # the classic dense path (4 x 16-tile GroupNorm slots, separate max-pool)
        torch.cuda.synchronize()
    assert torch.equal(out, again)
    assert float((out - ref).abs().max() / ref.abs().max()) < 5e-6      # float32-grade: only the statistics' slot partition differs
    assert not torch.equal(out[31], clean[31])
    rel = (out[:31] - clean[:31]).abs().max() / clean.abs().max()
    assert rel < 5e-6, rel                                    # the untouched pairs: dense stem vs split stem, float32-grade
