"""GPU (-m gpu): the sensor-frame entry (pnvo_forward_raw / pnvo_forward_dual_raw, SURVEY.md section 8(b) sketch) against the
materialised path it replaces — pnvo_build_obs_pairs (base_trainer_with_vo.py:172-269) followed by the model forward on the
float32 observation-pair tensors.  Same values reach the stem (uint8 rgb is exact, the one-hot depth is derived with the
reference's own comparisons), so the results must be IDENTICAL bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle
from pointnav_vo_amd import _lib, model_spec as ms, synth, vo_cnn
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd.trainer import NormalizedDepth2TopDownViewHabitatTorch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SPACE = ["rgb", "depth", "discretized_depth", "top_down_view"]


def build_model(name, space, W, H, seed, bins=10):
    kw = dict(observation_space=space, observation_size=(W, H), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
              output_dim=3, dropout_p=0.2)
    if bins:
        kw["discretized_depth_channels"] = bins
    m = baseline_registry.get_vo_model(name)(**kw)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=seed)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.to(DEV).eval(), sd


def frames(n, H, W, seed, depth_edit=None):
    """device tensors: rgb uint8 [n,2,H,W,3], depth float32 [n,2,H,W] (pair i = frames i, i + 1 of a synthetic walk)."""
    obs = [synth.make_raw_obs(H, W, seed=seed, index=i, zero_border=3 * (i % 3 == 0)) for i in range(n + 1)]
    rgb = np.stack([np.stack([obs[i]["rgb"], obs[i + 1]["rgb"]]) for i in range(n)])
    dep = np.stack([np.stack([obs[i]["depth"][..., 0], obs[i + 1]["depth"][..., 0]]) for i in range(n)]).astype(np.float32)
    if depth_edit is not None:
        depth_edit(dep)
    return torch.from_numpy(rgb).to(DEV), torch.from_numpy(dep).to(DEV)


def materialise(rgb, dep, H, W, bins, want_tdv=True):
    """pnvo_build_obs_pairs: the observation-pair tensors of the reference boundary, on the device."""
    n = dep.shape[0]
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    o = {"rgb": torch.empty((n, H, W, 6), device=DEV), "depth": torch.empty((n, H, W, 2), device=DEV),
         "discretized_depth": torch.empty((n, H, W, 2 * bins), device=DEV) if bins else None,
         "top_down_view": torch.empty((n, H, W, 2), device=DEV) if want_tdv else None}
    work = torch.empty(int(_lib.lib.pnvo_topdown_workspace_bytes(n, H, W)), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(_lib.lib.pnvo_build_obs_pairs(p(rgb), p(dep), n, H, W, bins, gen._consts, int(gen._rows_around_center), p(work), p(o["rgb"]),
                                             p(o["depth"]), p(o["discretized_depth"]), p(o["top_down_view"]), p(flag),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return {k: v for k, v in o.items() if v is not None}, flag


@pytest.mark.parametrize("n", [1, 5, 48])
def test_forward_raw_equals_the_materialised_path_bit_for_bit(n):
    """341x192, the default 30-channel model: 1 and 5 pairs (fp32-MFMA convs behind the float16 stem) and 48 (conv_x3 family,
    pooled stem keys); edge bins included (depth values sitting exactly on e_i and one float above / below)."""
    H, W = 192, 341
    model, sd = build_model("vo_cnn_rgb_d_dd_top_down", SPACE, W, H, seed=4)

    def edges(dep):                                   # plant every bin edge +-1 ulp and the closed ends 0 and 1
        e = np.array([np.float32(i / 10) for i in range(11)], dtype=np.float32)
        vals = np.concatenate([e, np.nextafter(e, np.float32(2))[:-1], np.nextafter(e, np.float32(-1))[1:]])
        dep[0, 0, 5, : vals.size] = vals
        dep[0, 1, 7, : vals.size] = vals[::-1]

    rgb, dep = frames(n, H, W, seed=11, depth_edit=edges)
    obs, flag = materialise(rgb, dep, H, W, 10)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    with torch.no_grad():
        want = model(obs)
        got = model.forward_raw(rgb, dep, obs["top_down_view"], err_flag=err)
    assert torch.equal(got, want), (got - want).abs().max()
    assert int(err) == 0 and int(flag) == 0
    if n <= 5:                                        # and both equal the reference arithmetic (fp64 oracle)
        ref = oracle.forward(sd, {k: v.cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
        o = got.cpu().numpy().astype(np.float64)
        assert (np.linalg.norm(o - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)).max() < 1e-4


@pytest.mark.parametrize("name,space,bins", [("vo_cnn_d_dd_top_down", ["depth", "discretized_depth", "top_down_view"], 10),
                                             ("vo_cnn_rgb_dd_top_down", ["rgb", "discretized_depth", "top_down_view"], 10),
                                             ("vo_cnn", ["rgb", "depth"], 0),
                                             ("vo_cnn_rgb_d_dd", ["rgb", "depth", "discretized_depth"], 10)])
def test_forward_raw_on_models_with_fewer_modalities(name, space, bins):
    """No rgb frames, no depth modality (the frames still feed the one-hot), no one-hot, no top-down view."""
    H, W = 66, 98
    model, _ = build_model(name, space, W, H, seed=6, bins=bins)
    rgb, dep = frames(7, H, W, seed=13)
    obs, _ = materialise(rgb, dep, H, W, bins, want_tdv="top_down_view" in space)
    with torch.no_grad():
        want = model({k: v for k, v in obs.items() if k in space})
        got = model.forward_raw(rgb if "rgb" in space else None, dep, obs.get("top_down_view"))
    assert torch.equal(got, want)


@pytest.mark.parametrize("option", [("pieces", "3"), ("stem", "dense"), ("stem", "dd")])
def test_forward_raw_outside_the_frame_reading_stem_materialises_internally(option):
    """Handles whose options take them off the frame-reading stem still accept frames (pairs built into a workspace)."""
    H, W = 66, 98
    model, _ = build_model("vo_cnn_rgb_d_dd_top_down", SPACE, W, H, seed=7)
    model.set_option(*option)
    rgb, dep = frames(4, H, W, seed=14)
    obs, _ = materialise(rgb, dep, H, W, 10)
    with torch.no_grad():
        want = model(obs)
        got = model.forward_raw(rgb, dep, obs["top_down_view"])
    assert torch.equal(got, want)


def test_forward_raw_flags_depth_outside_the_unit_interval():
    H, W = 66, 98
    model, _ = build_model("vo_cnn_rgb_d_dd_top_down", SPACE, W, H, seed=7)

    def spoil(dep):
        dep[2, 1, 30, 40] = 1.25

    rgb, dep = frames(4, H, W, seed=15, depth_edit=spoil)
    tdv = torch.zeros((4, H, W, 2), device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    with torch.no_grad():
        model.forward_raw(rgb, dep, tdv, err_flag=err)
    assert int(err) == 1                               # the reference asserts (base_trainer_with_vo.py:136-137)


def test_dual_forward_raw_equals_the_materialised_dual_forward():
    """BASELINE configs[2] (bf16, two action models, the second on the swapped pair) from the sensor frames."""
    H, W = 192, 341
    ma, _ = build_model("vo_cnn_rgb_d_dd_top_down", SPACE, W, H, seed=0)
    mb, _ = build_model("vo_cnn_rgb_d_dd_top_down", SPACE, W, H, seed=1)
    ma.set_precision("bfloat16")
    mb.set_precision("bfloat16")
    rgb, dep = frames(9, H, W, seed=16)
    obs, _ = materialise(rgb, dep, H, W, 10)
    with torch.no_grad():
        wa, wb = vo_cnn.dual_forward(ma, mb, obs)
        ga, gb = vo_cnn.dual_forward_raw(ma, mb, rgb, dep, obs["top_down_view"])
        one = ma.forward_raw(rgb, dep, obs["top_down_view"])            # single bf16 forward from frames
    assert torch.equal(ga, wa) and torch.equal(gb, wb)
    assert torch.equal(one, wa)
