"""GPU (-m gpu): every experiment knob of libpnvo selects a different kernel or grid shape, never a different result.
The knobs are read once per process, so each setting runs the golden check in a fresh interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import golden_case, load_golden
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd import vo_cnn
worst, bits = 0.0, []
for fname in ("model_default_341x192_b2.npz", "model_default_45x37_b3.npz", "model_wider_64x48_b2.npz"):
    rec = load_golden(fname)
    cfg, sd, obs, _ = golden_case(rec)
    kw = dict(observation_space=str(rec["obs_space"]).split(","), observation_size=(cfg.width, cfg.height), hidden_size=512,
              backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2)
    if int(rec["dd_bins"]):
        kw["discretized_depth_channels"] = int(rec["dd_bins"])
    m = baseline_registry.get_vo_model(str(rec["model"]))(**kw)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    with torch.no_grad():
        out = m({k: torch.from_numpy(v).to("cuda:0") for k, v in obs.items()}).cpu().numpy().astype(np.float64)
    ref = rec["out64"]
    err = np.linalg.norm(out - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    worst = max(worst, float(err.max()))
    bits.append(out.tobytes().hex())
print("WORST", worst)
assert worst < 1e-4, worst
""" % (ROOT, os.path.join(ROOT, "tests"))

# default = float32 convs on the bf16 matrix cores (conv_x3.hip) for every launch of >= 192 workgroups — none of the golden sizes,
# so PNVO_CONV=x3 (forced) is what covers those kernels here; PNVO_CONV=fp32 selects the fp32-MFMA kernels, whose own knobs only
# matter then
_FP32 = {"PNVO_CONV": "fp32"}
KNOBS = [{}, {"PNVO_CONV": "x3"}, _FP32, {**_FP32, "PNVO_CONV_WSPLIT": "1"}, {**_FP32, "PNVO_CONV_WSPLIT": "0"}, {**_FP32, "PNVO_CONV_TILE": "12"},
         {**_FP32, "PNVO_CONV_TILE": "22"}, {**_FP32, "PNVO_WAVE_NT": "2", "PNVO_WAVE_WGS": "4"}, {**_FP32, "PNVO_CONV3": "tile"},
         {**_FP32, "PNVO_CONV3": "wave"}, {"PNVO_CONV": "generic"}, {"PNVO_STEM": "dense"}, {"PNVO_STEM": "dd"}, {"PNVO_GRAPH": "1"},
         {"PNVO_CONV": "x3", "PNVO_TAIL": "separate"}, {"PNVO_CONV": "x3", "PNVO_POOL": "separate"},
         {"PNVO_CONV": "x3", "PNVO_X3_S2_OFF": "1"},
         # the exact-product fallbacks the documents advertise: three bf16 pieces everywhere (round 2's arithmetic)
         {"PNVO_PIECES": "3"}, {"PNVO_CONV": "x3", "PNVO_PIECES": "3"}, {"PNVO_CONV": "x3", "PNVO_PIECES": "3", "PNVO_POOL": "separate"}]


@pytest.mark.parametrize("env", KNOBS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_knob_keeps_parity(env):
    r = subprocess.run([sys.executable, "-c", CHECK], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("knob", ["PNVO_TAIL", "PNVO_POOL"])
def test_fused_passes_are_bit_identical_to_the_separate_ones(knob):
    """relu(GN2(conv2) + skip) computed in the next conv's stager (default; PNVO_TAIL=separate: residual_kernel) and the
    max-pool taken on order-preserving keys of sgn(gamma) * x in the stem's epilogue, normalised by the first conv's stager
    (default; PNVO_POOL=separate: gn_relu_maxpool_kernel) use the same float operations as the passes they replace: the
    network output must not change by a single bit."""
    outs = []
    for env in ({"PNVO_CONV": "x3"}, {"PNVO_CONV": "x3", knob: "separate"}):
        r = subprocess.run([sys.executable, "-c", CHECK + "\nprint('BITS', ''.join(bits))\n"], env={**os.environ, **env},
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("BITS")][-1])
    assert outs[0] == outs[1]


def test_parity_and_training_suites_with_conv_x3_forced():
    """The golden sizes launch fewer than 192 workgroups per conv, where the default keeps the fp32-MFMA kernels: run the
    forward parity suite (every intermediate activation) and the training suite again with conv_x3 forced."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_train.py"), "-m", "gpu", "-x", "-q"],
                       env={**os.environ, "PNVO_CONV": "x3"}, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


SIZES = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd import vo_cnn
space = ["rgb", "depth", "discretized_depth", "top_down_view"]
for (w, h, b) in ((97, 55, 3), (120, 67, 2), (200, 113, 3), (64, 33, 5), (341, 192, 12)):
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=space, observation_size=(w, h), hidden_size=512,
        backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=w)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    obs = synth.make_obs_pairs(b, h, w, observation_space=space, dd_bins=10, seed=h)
    with torch.no_grad():
        out = m({k: torch.from_numpy(np.asarray(v)).to("cuda:0") for k, v in obs.items()}).cpu().numpy()
    print("OUT", w, h, " ".join(repr(float(x)) for x in out.reshape(-1)))
""" % ROOT


def test_conv_x3_agrees_with_the_fp32_kernels_at_odd_sizes():
    """Ragged tiles, odd maps under the stride-2 convs, tile ownership of the fused block tails and of the pooled stem keys:
    conv_x3 forced against the fp32-MFMA kernels on sizes no golden covers (and 12 pairs at 341x192, where the default mixes
    both families).  Float32-grade agreement: 2e-5 of the pose norm (measured <= 3e-6)."""
    res = {}
    for name, env in (("x3", {"PNVO_CONV": "x3"}), ("fp32", {"PNVO_CONV": "fp32"}), ("auto", {})):
        r = subprocess.run([sys.executable, "-c", SIZES], env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[name] = [[float(x) for x in ln.split()[3:]] for ln in r.stdout.splitlines() if ln.startswith("OUT")]
        assert len(res[name]) == 5
    import numpy as np
    for other in ("x3", "auto"):
        for a, b in zip(res[other], res["fp32"]):
            a, b = np.array(a).reshape(-1, 3), np.array(b).reshape(-1, 3)
            err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-2)
            assert err.max() < 2e-5, (other, err.max())


TRAIN_KNOBS = [{"PNVO_CONV": "x3"}, {"PNVO_WGRAD_STEM": "fp32"}, {"PNVO_WGRAD": "lds9"}, {"PNVO_WGRAD": "generic"},
               # the exact fallbacks of the training step: three bf16 pieces (forward + backward), 3x3 weight gradients on the fp32 pipe
               {"PNVO_CONV": "x3", "PNVO_TRAIN_PIECES": "3"}, {"PNVO_CONV": "x3", "PNVO_WGRAD3": "fp32"},
               {"PNVO_CONV": "x3", "PNVO_TRAIN_PIECES": "3", "PNVO_WGRAD3": "fp32"}]


@pytest.mark.parametrize("env", TRAIN_KNOBS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_training_knob_keeps_gradient_parity(env):
    """The alternative weight-gradient kernels pass the same golden gradient check as the defaults."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_train.py"), "-m", "gpu", "-x", "-q",
                        "-k", "test_train_step_matches_reference"], env={**os.environ, **env}, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_persistent_resident_weight_convs_are_bit_identical():
    """conv_x3p_kernel (option x3_persist, default on: the 32 -> 32 convs of the first stage walk their tiles with the layer's whole
    B operand resident in registers) uses conv_x3_kernel's MFMA and statistics order: the network output does not change by a bit,
    at 16 and at 64 pairs of 341x192 (the launches it takes need >= 1024 tiles)."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    model.set_option("x3_rows", "off")                 # (from 64 pairs on the row-streaming kernel would take these layers)
    for B in (16, 64):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "on"):
            model.set_option("x3_persist", v)
            with torch.no_grad():
                outs.append(model(obs).clone())
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        fam = model.layer_kernel("visual_encoder.backbone.layer1.0.convs.3", B)[0]
        assert fam in ("x2", "x3"), fam


@pytest.mark.parametrize("B", [64, 100, 128, 256])
def test_row_streaming_convs_agree_with_the_tile_kernels(B):
    """conv_rows32_kernel (option x3_rows, default on: the 32 -> 32 convs of the first stage whose input is a GroupNorm-ed raw tensor,
    from one band per CU on) against conv_x3_kernel / conv_x3p_kernel: the same float16 pieces and products, two accumulators instead
    of one and one statistics slot per band — float32-grade agreement (5e-6 of the output range; measured 3e-6), run-to-run
    reproducible; 4 / 2 / 2 / 1 bands per sample at these batch sizes, the last with the GroupNorm finalised inside the kernel."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 2)
    outs = {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("x3_rows", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    assert torch.isfinite(outs["on"]).all() and torch.equal(outs["on"], outs["on2"])
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel                                   # (0 would mean the option selected nothing)
    chk = [0, B // 2, B - 1]
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    got = outs["on"][chk].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_row_streaming_convs_on_an_odd_resolution():
    """101 x 75 frames (51 x 38 stem output, 26 x 19 maps in the first stage: ragged half-groups, rows shorter than a tile), 300 pairs."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    from pointnav_vo_amd import model_spec as ms, synth
    from pointnav_vo_amd.registry import baseline_registry
    dev = torch.device("cuda", 0)
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(101, 75), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    obs = synth.make_obs_pairs(300, 75, 101, observation_space=bench.SPACE, dd_bins=10, seed=3)
    tobs = {k: torch.from_numpy(v).to(dev) for k, v in obs.items()}
    outs = {}
    with torch.no_grad():
        for v in ("on", "off"):
            m.set_option("x3_rows", v)
            outs[v] = m(tobs).clone()
        torch.cuda.synchronize()
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel
    ref = oracle.forward(sd, {k: v[:3] for k, v in obs.items()}, ngroups=m.cfg.ngroups, dtype=np.float64)
    got = outs["on"][:3].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_strip_tiles_of_the_128_channel_stage_agree_with_square_tiles():
    """Option x3_strip (default on): the 128-channel stage's conv_x3 launches use full-width strip tiles with a 5x1 wave grid and
    the N tiles over blockIdx.y.  Another tile plan changes which pixels a workgroup sums for the GroupNorm partials, so the
    outputs agree to float32 rounding of those sums, not by bit: 2e-5 of the pose norm (measured 3e-6), at 16 and 64 pairs."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    for B in (16, 64):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "on"):
            model.set_option("x3_strip", v)
            with torch.no_grad():
                outs.append(model(obs).double().cpu().numpy().copy())
        assert np.isfinite(outs[0]).all()
        assert np.array_equal(outs[0], outs[2])          # same plan, same bits
        err = np.linalg.norm(outs[0] - outs[1], axis=1) / np.maximum(np.linalg.norm(outs[1], axis=1), 1e-2)
        assert err.max() < 2e-5, err.max()


def test_in_kernel_groupnorm_finalisation_is_bit_identical():
    """Option gn_fuse: conv_x3 launches turn their channel sums into the GroupNorm scale / shift themselves, in gn_finalize_kernel's
    fp64 arithmetic and reduction order.  `on` (default, round 4): launches with one tile per sample (the 12x22 and 6x11 maps) — the
    tile that holds the sums.  `last` (round 6): also launches with several tiles per sample (24x43 maps, 48x86 below 64 pairs, the
    stride-2 block heads) — the sample's LAST workgroup to arrive behind a device-scope counter, partial sums moved with agent-scope
    accesses.  Which workgroup arrives last differs run to run; the network output must not change by a bit against the separate
    launch, at 8 ... 256 pairs, and repeat identically.  (`last` measured slower than the launches it removes — every workgroup
    waits for its stores and an atomic round trip — so it is not the default; the protocol is tested all the same.)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    for B in (8, 16, 33, 64, 256):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "last", "on", "last"):
            model.set_option("gn_fuse", v)
            with torch.no_grad():
                outs.append(model(obs).clone())
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), B
        # ... and the launches are really gone: finalisation launches per forward
        nfin = {}
        for v in ("last", "on", "off"):
            model.set_option("gn_fuse", v)
            model.timing(True)
            with torch.no_grad():
                model(obs)
            torch.cuda.synchronize()
            nfin[v] = sum(k["launches"] for k in model.timing_read() if k["name"] == "gn_finalize")
            model.timing(False)
        model.set_option("gn_fuse", "on")
        assert nfin["last"] < nfin["on"] <= nfin["off"], (B, nfin)
        if B == 256:                                       # what `last` leaves: the stem's and the compression conv's
            assert nfin["last"] <= 2, nfin
        fam = model.layer_kernel("visual_encoder.backbone.layer1.0.convs.3", B)[0]
        assert fam in ("x2", "x3"), fam


@pytest.mark.parametrize("B", [64, 100, 128, 256])
def test_row_streaming_convs_agree_with_the_tile_kernels(B):
    """conv_rows32_kernel (option x3_rows, default on: the 32 -> 32 convs of the first stage whose input is a GroupNorm-ed raw tensor,
    from one band per CU on) against conv_x3_kernel / conv_x3p_kernel: the same float16 pieces and products, two accumulators instead
    of one and one statistics slot per band — float32-grade agreement (5e-6 of the output range; measured 3e-6), run-to-run
    reproducible; 4 / 2 / 2 / 1 bands per sample at these batch sizes, the last with the GroupNorm finalised inside the kernel."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 2)
    outs = {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("x3_rows", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    assert torch.isfinite(outs["on"]).all() and torch.equal(outs["on"], outs["on2"])
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel                                   # (0 would mean the option selected nothing)
    chk = [0, B // 2, B - 1]
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    got = outs["on"][chk].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_row_streaming_convs_on_an_odd_resolution():
    """101 x 75 frames (51 x 38 stem output, 26 x 19 maps in the first stage: ragged half-groups, rows shorter than a tile), 300 pairs."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    from pointnav_vo_amd import model_spec as ms, synth
    from pointnav_vo_amd.registry import baseline_registry
    dev = torch.device("cuda", 0)
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(101, 75), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    obs = synth.make_obs_pairs(300, 75, 101, observation_space=bench.SPACE, dd_bins=10, seed=3)
    tobs = {k: torch.from_numpy(v).to(dev) for k, v in obs.items()}
    outs = {}
    with torch.no_grad():
        for v in ("on", "off"):
            m.set_option("x3_rows", v)
            outs[v] = m(tobs).clone()
        torch.cuda.synchronize()
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel
    ref = oracle.forward(sd, {k: v[:3] for k, v in obs.items()}, ngroups=m.cfg.ngroups, dtype=np.float64)
    got = outs["on"][:3].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_strip_tiles_of_the_128_channel_stage_agree_with_square_tiles():
    """Option x3_strip (default on): the 128-channel stage's conv_x3 launches use full-width strip tiles with a 5x1 wave grid and
    the N tiles over blockIdx.y.  Another tile plan changes which pixels a workgroup sums for the GroupNorm partials, so the
    outputs agree to float32 rounding of those sums, not by bit: 2e-5 of the pose norm (measured 3e-6), at 16 and 64 pairs."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    for B in (16, 64):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "on"):
            model.set_option("x3_strip", v)
            with torch.no_grad():
                outs.append(model(obs).double().cpu().numpy().copy())
        assert np.isfinite(outs[0]).all()
        assert np.array_equal(outs[0], outs[2])          # same plan, same bits
        err = np.linalg.norm(outs[0] - outs[1], axis=1) / np.maximum(np.linalg.norm(outs[1], axis=1), 1e-2)
        assert err.max() < 2e-5, err.max()


def test_in_kernel_groupnorm_finalisation_is_bit_identical():
    """Option gn_fuse (default on): conv_x3 launches turn their channel sums into the GroupNorm scale / shift themselves, in
    gn_finalize_kernel's fp64 arithmetic and reduction order — with one tile per sample (the 12x22 and 6x11 maps) the tile that holds
    the sums (`single`, round 4), with several tiles per sample (the 24x43 maps; 48x86 below 64 pairs; the stride-2 block heads) the
    sample's LAST workgroup to arrive, behind a device-scope counter (round 6).  Which workgroup arrives last differs run to run; the
    network output must not change by a bit against the separate launch, at 8 ... 256 pairs, and repeat identically."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    for B in (8, 16, 33, 64, 256):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "single", "on", "on"):
            model.set_option("gn_fuse", v)
            with torch.no_grad():
                outs.append(model(obs).clone())
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        assert all(torch.equal(outs[0], o) for o in outs[1:]), B
        # ... and the launches are really gone: finalisation launches per forward with the option on / one-tile launches only / off
        nfin = {}
        for v in ("on", "single", "off"):
            model.set_option("gn_fuse", v)
            model.timing(True)
            with torch.no_grad():
                model(obs)
            torch.cuda.synchronize()
            nfin[v] = sum(k["launches"] for k in model.timing_read() if k["name"] == "gn_finalize")
            model.timing(False)
        model.set_option("gn_fuse", "on")
        assert nfin["on"] < nfin["single"] <= nfin["off"], (B, nfin)
        if B == 256:                                       # what is left: the stem's and the compression conv's
            assert nfin["on"] <= 2, nfin
        fam = model.layer_kernel("visual_encoder.backbone.layer1.0.convs.3", B)[0]
        assert fam in ("x2", "x3"), fam


@pytest.mark.parametrize("B", [64, 100, 128, 256])
def test_row_streaming_convs_agree_with_the_tile_kernels(B):
    """conv_rows32_kernel (option x3_rows, default on: the 32 -> 32 convs of the first stage whose input is a GroupNorm-ed raw tensor,
    from one band per CU on) against conv_x3_kernel / conv_x3p_kernel: the same float16 pieces and products, two accumulators instead
    of one and one statistics slot per band — float32-grade agreement (5e-6 of the output range; measured 3e-6), run-to-run
    reproducible; 4 / 2 / 2 / 1 bands per sample at these batch sizes, the last with the GroupNorm finalised inside the kernel."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 2)
    outs = {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("x3_rows", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    assert torch.isfinite(outs["on"]).all() and torch.equal(outs["on"], outs["on2"])
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel                                   # (0 would mean the option selected nothing)
    chk = [0, B // 2, B - 1]
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    got = outs["on"][chk].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_row_streaming_convs_on_an_odd_resolution():
    """101 x 75 frames (51 x 38 stem output, 26 x 19 maps in the first stage: ragged half-groups, rows shorter than a tile), 300 pairs."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    from pointnav_vo_amd import model_spec as ms, synth
    from pointnav_vo_amd.registry import baseline_registry
    dev = torch.device("cuda", 0)
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(101, 75), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    obs = synth.make_obs_pairs(300, 75, 101, observation_space=bench.SPACE, dd_bins=10, seed=3)
    tobs = {k: torch.from_numpy(v).to(dev) for k, v in obs.items()}
    outs = {}
    with torch.no_grad():
        for v in ("on", "off"):
            m.set_option("x3_rows", v)
            outs[v] = m(tobs).clone()
        torch.cuda.synchronize()
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel
    ref = oracle.forward(sd, {k: v[:3] for k, v in obs.items()}, ngroups=m.cfg.ngroups, dtype=np.float64)
    got = outs["on"][:3].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_strip_tiles_of_the_128_channel_stage_agree_with_square_tiles():
    """Option x3_strip (default on): the 128-channel stage's conv_x3 launches use full-width strip tiles with a 5x1 wave grid and
    the N tiles over blockIdx.y.  Another tile plan changes which pixels a workgroup sums for the GroupNorm partials, so the
    outputs agree to float32 rounding of those sums, not by bit: 2e-5 of the pose norm (measured 3e-6), at 16 and 64 pairs."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    for B in (16, 64):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "on"):
            model.set_option("x3_strip", v)
            with torch.no_grad():
                outs.append(model(obs).double().cpu().numpy().copy())
        assert np.isfinite(outs[0]).all()
        assert np.array_equal(outs[0], outs[2])          # same plan, same bits
        err = np.linalg.norm(outs[0] - outs[1], axis=1) / np.maximum(np.linalg.norm(outs[1], axis=1), 1e-2)
        assert err.max() < 2e-5, err.max()


def test_in_kernel_groupnorm_finalisation_is_bit_identical():
    """Option gn_fuse (default on): conv_x3 launches with one tile per sample (the 12x22 and 6x11 maps) turn their channel sums into
    the GroupNorm scale / shift themselves, in gn_finalize_kernel's fp64 arithmetic and butterfly order: the network output does
    not change by a bit, at 16, 64 and 256 pairs."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    for B in (16, 64, 256):
        obs = bench.make_inputs(B, dev, 1)
        outs = []
        for v in ("on", "off", "on"):
            model.set_option("gn_fuse", v)
            with torch.no_grad():
                outs.append(model(obs).clone())
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("B", [16, 64, 100, 256])
def test_downsample_conv_riding_on_its_block_head(B):
    """Option ds_fuse (default on, round 6): the 1x1 stride-2 downsample conv of layer2.0 / layer3.0 / layer4.0 (resnet.py:192-195) is
    computed by its block's first 3x3 stride-2 conv launch (conv_x3_kernel<.., DSF>): same float16 pieces, the same three MFMA terms
    per k-chunk in the same order as the separate 1x1 launch — only the GroupNorm partial sums are grouped by the head's tiles instead
    of the 1x1 plan's, so the network output agrees to float32 summation noise (2e-6 of the output range), reproduces run to run,
    and matches the fp64 oracle like the separate launches do.  In the block-tail mode the block input is not written at all."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 3)
    outs = {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("ds_fuse", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    model.set_option("ds_fuse", "on")
    fams = [model.layer_kernel(f"visual_encoder.backbone.layer{s}.0.downsample.0", B)[0] for s in (2, 3, 4)]
    assert "x2-rides" in fams, fams                              # (deep stages at small batches stay on the fp32 kernels)
    model.set_option("ds_fuse", "off")
    assert all(model.layer_kernel(f"visual_encoder.backbone.layer{s}.0.downsample.0", B)[0] != "x2-rides" for s in (2, 3, 4))
    assert torch.isfinite(outs["on"]).all() and torch.equal(outs["on"], outs["on2"])
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 2e-6, rel                                   # (0 would mean the option selected nothing)
    chk = [0, B // 2, B - 1]
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    for v in ("on", "off"):
        got = outs[v][chk].double().cpu().numpy()
        err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
        assert err.max() < 2e-5, (v, err)


def test_downsample_ride_on_an_odd_resolution_and_with_taps():
    """101 x 75 frames, 300 pairs (ragged stride-2 tiles: 26 x 19 -> 13 x 10 -> 7 x 5 -> 4 x 3 maps); and a tap of a block output forces the
    plain schedule (the block input IS materialised then): same network output as the default schedule to float32 noise."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    from pointnav_vo_amd import model_spec as ms, synth
    from pointnav_vo_amd.registry import baseline_registry
    dev = torch.device("cuda", 0)
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(101, 75), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    obs = synth.make_obs_pairs(300, 75, 101, observation_space=bench.SPACE, dd_bins=10, seed=3)
    tobs = {k: torch.from_numpy(v).to(dev) for k, v in obs.items()}
    outs = {}
    with torch.no_grad():
        for v in ("on", "off"):
            m.set_option("ds_fuse", v)
            outs[v] = m(tobs).clone()
        m.set_option("ds_fuse", "on")
        out_tap, _ = m.tap("layer2.0", tobs)
        torch.cuda.synchronize()
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert rel < 2e-6, rel
    # (the tap's plain schedule also swaps the fused pool / block tails for their separate passes and runs the row-streaming kernel in
    #  its plain-input mode 0 — where round 6 found sample 0's top-right padding corner reading pixel (0, 0): 5e-3 on that sample)
    assert float((out_tap - outs["on"]).abs().max() / outs["on"].abs().max()) < 1e-5
    ref = oracle.forward(sd, {k: v[:3] for k, v in obs.items()}, ngroups=m.cfg.ngroups, dtype=np.float64)
    got = outs["on"][:3].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 2e-5, err


def test_narrow_models_are_refused_not_mis_served():
    """ADVICE r5 asked for a resnet_baseplanes = 16 model: its layer2 would be a 32 -> 32 channel stage whose second block takes a block
    tail with a DOWNSAMPLE skip (res * scale + shift), which conv_rows32_kernel's block-tail mode does not apply.  conv_rows32_plan now
    sees the tail's fields before it decides (csrc/pnvo_api.hip) — and such a model cannot reach it in the first place: pnvo_create
    refuses base widths below 32 (every registered reference variant has 32 or 64, vo_cnn.py:236-561) with an error, never a fallback."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from pointnav_vo_amd import _lib, model_spec as ms, synth
    from pointnav_vo_amd.registry import baseline_registry
    dev = torch.device("cuda", 0)
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(101, 75), hidden_size=512, backbone="resnet18", resnet_baseplanes=16,
        normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=2)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    obs = synth.make_obs_pairs(2, 75, 101, observation_space=bench.SPACE, dd_bins=10, seed=4)
    with pytest.raises(_lib.PnvoError, match="baseplanes"):
        with torch.no_grad():
            m.to(dev).eval()({k: torch.from_numpy(v).to(dev) for k, v in obs.items()})


def test_pooled_keys_refilled_on_a_side_stream_are_bit_identical():
    """Option pool_async (round 6; OFF by default — bit-identical but measured slower: the event hand-overs between the streams cost more
    than the 23 us fill they hide): the pooled stem keys live in a buffer of their own whose re-initialisation for the NEXT
    forward runs on a side stream behind this forward's consumer (next to the deep stages) instead of in front of the next stem.
    Same kernels, same arithmetic: outputs bit-identical to the in-stream fill, over repeated forwards, batches that grow and shrink
    between calls (a fill sized for the smaller batch must not be trusted for the larger one), changing inputs, a timed forward
    in between (which fills in-stream), and back-to-back forwards without any host synchronisation."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    obs_all = bench.make_inputs(96, dev, 5)
    seq = [64, 64, 16, 96, 8, 96, 33, 64]
    def run(mode):
        model.set_option("pool_async", mode)
        outs = []
        with torch.no_grad():
            for i, B in enumerate(seq):
                o = {k: v[(i % 3):(i % 3) + B].contiguous() if (i % 3) + B <= 96 else v[:B].contiguous() for k, v in obs_all.items()}
                outs.append(model(o).clone())
                if i == 4:
                    model.timing(True)
                    outs.append(model(o).clone())
                    model.timing_read()
                    model.timing(False)
            for _ in range(6):                           # a backlog of forwards, no synchronisation in between
                last = model(o)
            outs.append(last.clone())
        torch.cuda.synchronize()
        return outs
    a, b, c = run("on"), run("off"), run("on")
    assert all(torch.isfinite(x).all() for x in a)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and all(torch.equal(x, y) for x, y in zip(a, c))


@pytest.mark.parametrize("B", [6, 8, 16, 32, 48])
def test_fine_plan_keeps_small_batches_on_the_float16_pipe(B):
    """Option x3_fine (default on, round 6): a conv launch that the regular tile plan would give fewer than 224 workgroups takes ONE
    N-tile per workgroup (blockIdx.y = N-tile, four waves along M) instead of falling back to the fp32-pipe kernels: the deep stages
    of 6-48-pair batches then run on the float16 pieces with block tails, riding downsample convs and in-kernel GroupNorm
    finalisation.  Against the fallback (x3_fine=off) the output agrees to float32 noise, both match the fp64 oracle, the option
    really changes the kernel family of the deep layers, and there are fewer launches."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 7)
    outs, fam, nl = {}, {}, {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("x3_fine", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
            fam[v] = [model.layer_kernel(f"visual_encoder.backbone.layer{s}.1.convs.3", B)[0] for s in (3, 4)]
            model.timing(True)
            model(obs)
            torch.cuda.synchronize()
            nl[v] = sum(k["launches"] for k in model.timing_read())
            model.timing(False)
        torch.cuda.synchronize()
    model.set_option("x3_fine", "on")
    assert torch.isfinite(outs["on"]).all() and torch.equal(outs["on"], outs["on2"])
    assert "x2" in fam["on"] and fam["on"] != fam["off"], (fam, B)
    assert nl["on"] < nl["off"], nl
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel
    chk = sorted({0, B // 2, B - 1})
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    for v in ("on", "off"):
        got = outs[v][chk].double().cpu().numpy()
        err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
        assert err.max() < 2e-5, (v, err)


@pytest.mark.parametrize("B", [5, 8, 33, 64, 100])
def test_k_split_over_the_waves_agrees_with_the_m_split_fine_plan(B):
    """Option x3_ksplit (default on, round 6): fine-plan tiles of three / four M-tiles behind >= 128 input channels (the 12 x 22 and 6 x 11
    stages of 5-~110-pair batches) are multiplied by all four waves, each over a quarter of the K steps, and the partial accumulators
    meet in LDS in wave order — instead of one M-tile per wave over the whole K walk.  Another summation order of the same float32-grade
    products: agreement with the M-split plan to float32 noise, both within the oracle tolerance, deterministic."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 13)
    outs = {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("x3_ksplit", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    model.set_option("x3_ksplit", "on")
    assert torch.isfinite(outs["on"]).all() and torch.equal(outs["on"], outs["on2"])
    rel = float((outs["on"] - outs["off"]).abs().max() / outs["off"].abs().max())
    assert 0 < rel < 5e-6, rel
    chk = sorted({0, B // 2, B - 1})
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    for v in ("on", "off"):
        got = outs[v][chk].double().cpu().numpy()
        err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
        assert err.max() < 2e-5, (v, err)


def test_weight_changes_reach_the_handle_through_the_cached_tensor_list():
    """The module's parameter objects are resolved once (round 6); what changes afterwards must still reach the handle: an in-place
    edit (version counter), load_state_dict, a re-assigned buffer (RunningMeanAndVar's, read through getattr every call), .to() / .float()
    (nn.Module._apply drops the cache).  Each against a fresh model built with the same values."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(5, dev, 3)

    def fresh_out(state):
        m, _ = bench.build_model(dev)
        m.load_state_dict(state)
        with torch.no_grad():
            return m(obs).clone()

    with torch.no_grad():
        base = model(obs).clone()
        p = dict(model.named_parameters())["visual_encoder.backbone.layer2.0.convs.0.weight"]
        p.mul_(1.25)                                               # in place: same object, same pointer, version + 1
        a = model(obs).clone()
        assert not torch.equal(a, base)
        assert torch.equal(a, fresh_out({k: v.clone() for k, v in model.state_dict().items()}))
        rm = model.visual_encoder.running_mean_and_var
        rm._mean = (rm._mean + 0.05).clone()                       # a buffer re-assigned (what a train-mode forward does)
        b = model(obs).clone()
        assert not torch.equal(b, a)
        assert torch.equal(b, fresh_out({k: v.clone() for k, v in model.state_dict().items()}))
        model = model.cpu().to(dev)                                # _apply twice: tensors re-resolved, handle re-fed
        assert torch.equal(model(obs), b)
        model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        assert torch.equal(model(obs), base)


def test_eight_wave_forms_in_a_training_step_are_bit_identical():
    """From 200 pairs on the training forward takes the eight-wave deep-stage forms too (the stride-2 head keeps the block input and
    hands mean / rstd to the backward): two training steps at 208 pairs with x3_w8 on / off end in the same losses and parameters."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from pointnav_vo_amd import model_spec as ms, synth
    from pointnav_vo_amd.registry import baseline_registry
    from pointnav_vo_amd.train import VOTrainStep
    dev = torch.device("cuda", 0)
    B, res = 208, {}
    obs = bench.make_inputs(B, dev, 0)
    tgt = (torch.arange(B * 3, device=dev, dtype=torch.float32).reshape(B, 3) % 7 - 3) * 0.05
    for v in ("on", "off"):
        model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
            observation_space=bench.SPACE, observation_size=(bench.W, bench.H), hidden_size=512, backbone="resnet18",
            normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=bench.BINS)
        sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=0)
        model.load_state_dict({k: torch.from_numpy(np.array(x)) for k, x in sd.items()})
        model = model.to(dev)
        model.set_option("x3_w8", v)
        ts = VOTrainStep(model)
        losses = [float(ts.step(obs, tgt)[1]) for _ in range(2)]
        torch.cuda.synchronize()
        res[v] = (losses, torch.cat([q.detach().reshape(-1) for q in model.parameters()]).clone())
        del ts, model
    assert res["on"][0] == res["off"][0], (res["on"][0], res["off"][0])
    assert torch.isfinite(res["on"][1]).all() and torch.equal(res["on"][1], res["off"][1])


@pytest.mark.parametrize("B", [4, 7, 17, 48])
def test_hidden_layer_and_head_as_row_kernels_agree_with_the_mfma_path(B):
    """Option fc_rows (default 48 samples, round 6): up to that batch the Linear(flat, hidden) + ReLU and the output head (vo_cnn.py:216-227)
    run as fc_rows.hip's two launches — a wave per hidden unit and chunk of four samples, the compression conv's GroupNorm + ReLU applied
    while reading — instead of the split-K MFMA kernel and its reduction.  float32 FMA chains against float32 MFMA accumulation:
    agreement to float32 noise, both within the oracle tolerance; fc_rows=0 keeps the MFMA path."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 17)
    outs = {}
    with torch.no_grad():
        for v in ("48", "0", "48b"):
            model.set_option("fc_rows", v[:2].rstrip("b"))
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    model.set_option("fc_rows", "48")
    assert torch.isfinite(outs["48"]).all() and torch.equal(outs["48"], outs["48b"])
    rel = float((outs["48"] - outs["0"]).abs().max() / outs["0"].abs().max())
    assert 0 < rel < 5e-6, rel
    chk = sorted({0, B // 2, B - 1})
    ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    for v in ("48", "0"):
        got = outs[v][chk].double().cpu().numpy()
        err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
        assert err.max() < 2e-5, (v, err)


@pytest.mark.parametrize("B", [200, 256])
def test_eight_wave_deep_stage_is_bit_identical(B):
    """Option x3_w8 (default on, round 6): the 256-channel 3x3 convs on the 6 x 11 maps (resnet.py:29-55, layer4) are one tile per pair;
    from 200 pairs on their workgroup is eight waves of (3,1) tiles — two waves per SIMD — instead of four of (3,2).  Every output is
    the same MFMA chain in the same order and the GroupNorm sums are taken per wave over the same three M-tiles: bit-identical."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 11)
    outs = {}
    with torch.no_grad():
        for v in ("on", "off", "on2"):
            model.set_option("x3_w8", v[:2] if v != "off" else "off")
            outs[v] = model(obs).clone()
        torch.cuda.synchronize()
    model.set_option("x3_w8", "on")
    assert torch.isfinite(outs["on"]).all()
    assert torch.equal(outs["on"], outs["off"]) and torch.equal(outs["on"], outs["on2"])


@pytest.mark.parametrize("B", [5, 8, 16, 33, 48])
def test_groupnorm_finalisation_deferred_to_the_consumer_is_bit_identical(B):
    """Option gn_defer (round 6; OFF by default = 0 pairs — bit-identical but measured slower: a memory round trip and half a microsecond
    of fp64 in every consumer workgroup cost more than the 4 us launches they remove): up to that batch a conv_x3 producer with several tiles per sample leaves its
    GroupNorm partial sums un-finalised and the CONSUMER launch (the block's second conv; the conv that takes the block tail, for
    the second conv's and the downsample conv's GroupNorms) builds its sample's scale / shift table in its prologue — one wave,
    sixteen groups at once, gn_finalize_kernel's fp64 arithmetic and reduction tree reproduced with four lanes per group.  Bit for
    bit the separate launches' result; about ten launches fewer per forward."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    obs = bench.make_inputs(B, dev, 9)
    outs, nfin = {}, {}
    with torch.no_grad():
        for v in ("48", "0", "48"):
            model.set_option("gn_defer", v)
            o = model(obs).clone()
            outs.setdefault(v, []).append(o)
            model.timing(True)
            model(obs)
            torch.cuda.synchronize()
            nfin[v] = sum(k["launches"] for k in model.timing_read() if k["name"] == "gn_finalize")
            model.timing(False)
        torch.cuda.synchronize()
    assert torch.isfinite(outs["48"][0]).all()
    assert torch.equal(outs["48"][0], outs["0"][0]) and torch.equal(outs["48"][0], outs["48"][1])
    assert nfin["48"] < nfin["0"], nfin


def test_deferred_finalisation_in_the_grouped_forward():
    """The grouped forward with and without deferred finalisation (the consumer picks the PRODUCER layer's affine parameters of the
    sample's own model): bit-identical."""
    import torch
    from test_gpu_grouped import _frames, _models
    from pointnav_vo_amd.vo_cnn import grouped_forward_raw
    models = _models(3)
    rgb, dep, tdv = _frames(14, 77)
    outs = []
    with torch.no_grad():
        for v in ("48", "0"):
            for m, _ in models:
                m.set_option("gn_defer", v)
            outs.append(grouped_forward_raw([m for m, _ in models], (5, 4, 5), rgb, dep, tdv).clone())
        torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
