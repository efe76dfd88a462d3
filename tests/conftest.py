import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def pair_rel_err(out, ref):
    """Per-pair ||out-ref||_2 / max(||ref||_2, 1e-2) — the metric of BASELINE.md §5 / SURVEY.md §7."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.linalg.norm(out - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-2)


def golden_case(rec):
    """Rebuild (cfg, state_dict, obs, actions) of a model fixture from its seeds."""
    from pointnav_vo_amd import model_spec as ms
    from pointnav_vo_amd import synth

    obs_space = str(rec["obs_space"]).split(",")
    bins = int(rec["dd_bins"])
    cfg = ms.config_from_kwargs(
        observation_space=obs_space, observation_size=(int(rec["width"]), int(rec["height"])),
        hidden_size=512, resnet_baseplanes=int(rec["baseplanes"]), normalize_visual_inputs=True, output_dim=3,
        discretized_depth_channels=bins, act_embed=bool(int(rec["act_embed"])),
        backbone=str(rec["backbone"]) if "backbone" in rec else "resnet18")
    sd = synth.make_state_dict(ms.state_dict_spec(cfg), seed=int(rec["seed"]))
    obs = synth.make_obs_pairs(int(rec["batch"]), cfg.height, cfg.width, observation_space=obs_space,
                               dd_bins=max(bins, 1), seed=int(rec["seed"]),
                               depth_fp16=bool(int(rec["depth_fp16"])) if "depth_fp16" in rec else True)
    actions = rec["actions"] if "actions" in rec else None
    return cfg, sd, obs, actions


MODEL_FIXTURES = [
    "model_default_341x192_b2.npz",
    "model_default_341x192_b2_f32depth.npz",      # dense float32 depth (simulator-style), not float16-exact
    "model_default_45x37_b3.npz",
    "model_vo_cnn_64x48_b2.npz",
    "model_rgb_d_dd_70x40_b2.npz",
    "model_wider_64x48_b2.npz",
    "model_deeper_64x48_b2.npz",
    "model_act_embed_64x48_b3.npz",
    "model_d_dd_tdv_66x34_b2.npz",
]
