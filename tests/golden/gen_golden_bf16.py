#!/usr/bin/env python
"""Golden fixture for the bf16 dual forward (BASELINE configs[2]), from the IMPORTED reference (build container only).

    python tests/golden/gen_golden_bf16.py         # writes tests/golden/dual_bf16_341x192_b6.npz

Two seeded action models of `vo_cnn_rgb_d_dd_top_down` at 341x192 (the act_left_right_inv_joint setting): model a on the
(prev, cur) pair, model b on the channel-SWAPPED pair, built exactly as the reference's dataset builds the inversion
entries (regression_geo_invariance_iter_dataset.py:342-386: every observation tensor's [prev | cur] halves exchanged).
Stored (outputs only; inputs / weights are regenerated from the seeds by pointnav_vo_amd.synth):
    out64_a / out64_b   the reference in float64
    cast_a / cast_b     the reference cast to bfloat16 AS A WHOLE (model.bfloat16(), bf16 inputs) — the baseline the build's
                        bf16 path is compared with ("bf16 vs fp64 reference (whole-model cast)")
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from pointnav_vo_amd import synth  # noqa: E402

SPACE = ["rgb", "depth", "discretized_depth", "top_down_view"]
W, H, B, BINS = 341, 192, 6, 10


def swapped(obs):
    return {k: np.concatenate([v[..., v.shape[-1] // 2:], v[..., : v.shape[-1] // 2]], axis=-1) for k, v in obs.items()}


def main():
    registry, _ = gg.import_reference()
    torch.manual_seed(0)
    obs = synth.make_obs_pairs(B, H, W, observation_space=SPACE, dd_bins=BINS, seed=0)
    rec = dict(model="vo_cnn_rgb_d_dd_top_down", obs_space=",".join(SPACE), width=W, height=H, batch=B, dd_bins=BINS,
               seed_a=0, seed_b=77, obs_seed=0)
    for tag, seed, o in (("a", 0, obs), ("b", 77, swapped(obs))):
        model, cfg, sd = gg.build_ref_model(registry, "vo_cnn_rgb_d_dd_top_down", SPACE, (W, H), BINS, seed)
        t = {k: torch.from_numpy(v) for k, v in o.items()}
        with torch.no_grad():
            out64 = model.double()({k: v.double() for k, v in t.items()}).numpy()
            cast = model.bfloat16()({k: v.bfloat16() for k, v in t.items()}).float().numpy()
        rec[f"out64_{tag}"], rec[f"cast_{tag}"] = out64, cast
        print(tag, "whole-model bf16 cast: max abs err", np.abs(cast - out64).max(), " per-pair l2", np.linalg.norm(cast - out64, axis=1),
              " |ref|", np.linalg.norm(out64, axis=1))
    np.savez(os.path.join(HERE, "dual_bf16_341x192_b6.npz"), **rec)


if __name__ == "__main__":
    main()
