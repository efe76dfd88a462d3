"""GPU (-m gpu): the grouped forward (pnvo_forward_grouped_raw / vo_cnn.grouped_forward_raw): the pairs of the forward / left / right
action models (base_trainer_with_vo.py:56-81,277-294: one model per action, picked per environment) in ONE launch chain whose
kernels select a pair's weights by its model.  Checked against every model's own forward_raw on its pairs (float32 noise: the
default forward picks other kernels at these batch sizes) and against the fp64 oracle with each model's state dict."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from pointnav_vo_amd import _lib, model_spec as ms, synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd.vo_cnn import grouped_forward_raw
from test_gpu_parity import make_trainer

pytestmark = pytest.mark.gpu
SPACE = ["rgb", "depth", "discretized_depth", "top_down_view"]
W, H = 341, 192


def _models(n, seed0=11):
    dev = torch.device("cuda", 0)
    out = []
    for k in range(n):
        m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
            observation_space=SPACE, observation_size=(W, H), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
            output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
        sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=seed0 + k)
        m.load_state_dict({n_: torch.from_numpy(np.array(v)) for n_, v in sd.items()})
        out.append((m.to(dev).eval(), sd))
    return out


def _frames(B, seed):
    """Sensor frames [B,2,H,W,3] uint8 / [B,2,H,W] float32 and the top-down views the build's own kernel makes of them."""
    from pointnav_vo_amd.trainer import NormalizedDepth2TopDownViewHabitatTorch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rgb = torch.randint(0, 256, (B, 2, H, W, 3), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    dep = (torch.rand((B, 2, H, W), device=dev, generator=g) * 0.9 + 0.05).to(torch.float16).to(torch.float32)
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    tdv = torch.empty((B, H, W, 2), device=dev, dtype=torch.float32)
    gen.gen_top_down_view_pairs(dep, tdv)
    return rgb, dep, tdv


def _obs_pairs(rgb, dep, tdv, idx):
    """Observation pairs (numpy, the oracle's input) of the selected frames."""
    from oracle import oracle
    r = rgb[idx].cpu().numpy().astype(np.float32)                       # [n,2,H,W,3] -> [n,H,W,6]
    d = dep[idx].cpu().numpy()
    obs = {"rgb": np.concatenate([r[:, 0], r[:, 1]], axis=-1), "depth": np.stack([d[:, 0], d[:, 1]], axis=-1),
           "top_down_view": tdv[idx].cpu().numpy()}
    dd = [oracle.discretize_depth(d[:, k], 10)[0] for k in range(2)]
    obs["discretized_depth"] = np.concatenate(dd, axis=-1)
    return obs


@pytest.mark.parametrize("counts", [(5, 3, 5), (7, 0, 4), (4, 9), (16, 12, 12), (1, 1, 1)])
def test_grouped_forward_matches_the_models_own_forwards(counts):
    from oracle import oracle
    B = sum(counts)
    models = _models(len(counts))
    rgb, dep, tdv = _frames(B, 100 + B)
    flag = torch.zeros(1, dtype=torch.int32, device=rgb.device)
    with torch.no_grad():
        got = grouped_forward_raw([m for m, _ in models], counts, rgb, dep, tdv, err_flag=flag)
        again = grouped_forward_raw([m for m, _ in models], counts, rgb, dep, tdv, err_flag=flag)
        sep, lo = [], 0
        for (m, _), c in zip(models, counts):
            if c:
                sep.append(m.forward_raw(rgb[lo:lo + c], dep[lo:lo + c], tdv[lo:lo + c]))
            lo += c
        sep = torch.cat(sep)
        torch.cuda.synchronize()
    assert int(flag[0]) == 0 and torch.isfinite(got).all() and torch.equal(got, again)
    rel = float((got - sep).abs().max() / sep.abs().max())
    assert rel < 1e-5, rel
    # fp64 oracle, each pair with ITS model's parameters: the first and the last pair of every model
    lo = 0
    for (m, sd), c in zip(models, counts):
        if c:
            idx = sorted({lo, lo + c - 1})
            ref = oracle.forward(sd, _obs_pairs(rgb, dep, tdv, idx), ngroups=m.cfg.ngroups, dtype=np.float64)
            g = got[idx].double().cpu().numpy()
            err = np.linalg.norm(g - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
            assert err.max() < 2e-5, (counts, lo, err)
        lo += c
    # a model that is NOT the pair's model gives another answer (the selection is real)
    if len([c for c in counts if c]) > 1 and counts[0] and counts[-1]:
        with torch.no_grad():
            wrong = models[0][0].forward_raw(rgb[B - 1:], dep[B - 1:], tdv[B - 1:])
        assert float((wrong - got[B - 1:]).abs().max()) > 1e-3


def test_grouped_forward_refuses_what_it_cannot_serve():
    models = _models(2)
    rgb, dep, tdv = _frames(4, 3)
    with pytest.raises(ValueError):
        grouped_forward_raw([m for m, _ in models], (1, 1), rgb, dep, tdv)            # counts do not add up
    models[1][0].set_option("pieces", "3")
    with pytest.raises(_lib.PnvoError, match="grouped forward"):
        grouped_forward_raw([m for m, _ in models], (2, 2), rgb, dep, tdv)
    models[1][0].set_option("pieces", "2")
    with torch.no_grad():
        out = grouped_forward_raw([m for m, _ in models], (2, 2), rgb, dep, tdv)      # the handles are usable afterwards
    assert torch.isfinite(out).all()
    assert models[0][0].get_option("conv") == "auto" and models[0][0].get_option("x3_rows") == "on"   # the leader's options are restored


def test_boundary_call_uses_the_grouped_forward_and_agrees_with_per_model_forwards():
    rec = load_golden("boundary.npz")
    Hh, Ww = int(rec["height"]), int(rec["width"])
    grouped, plain = make_trainer(rec), make_trainer(rec)
    plain.group_max_pairs = 0
    rng = np.random.default_rng(5)
    for E in (3, 8, 20):
        prevs = [synth.make_raw_obs(Hh, Ww, seed=60 + e, index=0) for e in range(E)]
        curs = [synth.make_raw_obs(Hh, Ww, seed=60 + e, index=1) for e in range(E)]
        acts = [int(a) for a in rng.integers(1, 4, size=E)]
        a = grouped.compute_local_delta_states_batch(prevs, curs, acts)
        b = plain.compute_local_delta_states_batch(prevs, curs, acts)
        assert a.shape == (E, 3) and np.isfinite(a).all()
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-5
        a2 = grouped.compute_local_delta_states_batch(prevs, curs, acts, env_ids=list(range(E)))
        assert np.array_equal(a, a2)                                     # the frame ring feeds the same frames
