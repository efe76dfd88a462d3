#!/usr/bin/env python
"""BASELINE configs[2] shape: the geometric-invariance DUAL forward of the joint left/right training/eval
(`act_left_right_inv_joint`): every turn pair is evaluated by its own action model on (prev, cur) and by the opposite
action model on the channel-swapped (cur, prev) (regression_geo_invariance_iter_dataset.py:342-386,
vo_cnn_regression_geo_invariance_engine.py:569-602).  Two weight sets, two forwards per pair.

Runs in fp32 (>= the bf16 that config names; a bf16 kernel set does not exist yet) and reports pairs/s where a pair
counts once although it costs two forwards.   python tools/bench_dual.py [--batch 256] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointnav_vo_amd import model_spec as ms, synth  # noqa: E402
from pointnav_vo_amd.registry import baseline_registry  # noqa: E402


def swap_prev_cur(obs):
    """(prev, cur) -> (cur, prev): every observation tensor stores [prev channels | cur channels] on its last axis."""
    out = {}
    for k, v in obs.items():
        h = v.shape[-1] // 2
        out[k] = torch.cat([v[..., h:], v[..., :h]], dim=-1).contiguous()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    models = []
    for seed in (0, 1):                                   # left and right action models: same architecture, own weights
        m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
            observation_space=bench.SPACE, observation_size=(bench.W, bench.H), hidden_size=512, backbone="resnet18",
            normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=bench.BINS)
        sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=seed)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        models.append(m.to(dev).eval())
    obs = bench.make_inputs(a.batch, dev, 0)
    obs_sw = swap_prev_cur(obs)
    with torch.no_grad():
        for _ in range(a.warmup):
            models[0](obs), models[1](obs_sw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            o_l = models[0](obs)
            o_r = models[1](obs_sw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    flops = 2 * 2.0 * ms.macs_per_pair(models[0].cfg) * a.batch
    print(json.dumps({"metric": "dual-forward (left on (p,c) + right on (c,p)) frame-pairs/s @341x192", "value": a.batch / dt,
                      "unit": "frame-pairs/s", "ms_per_step": dt * 1e3, "batch": a.batch, "dtype": "f32",
                      "forwards_per_pair": 2, "model_tflops": flops / dt / 1e12,
                      "finite": bool(torch.isfinite(o_l).all() and torch.isfinite(o_r).all())}))


if __name__ == "__main__":
    main()
