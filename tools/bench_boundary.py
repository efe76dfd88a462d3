#!/usr/bin/env python
"""Boundary-level timings (not the bench.py metric): host numpy observations in, 3 floats per pair out.
  * batch-1 latency of _compute_local_delta_states_from_vo (the reference's per-env call, ppo_trainer.py:836-841)
  * PCIe-inclusive throughput of compute_local_delta_states_batch (uint8 rgb + fp32 depth over PCIe, dd/top-down on device)
  * model-only batch-1 latency (obs_pairs resident)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import model_spec as ms, synth  # noqa: E402
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO  # noqa: E402

W, H = 341, 192
cfg = AttrDict(
    VO=dict(VO_TYPE="REGRESS", OBS_TRANSFORM="none", VIS_SIZE_W=W, VIS_SIZE_H=H,
            REGRESS_MODEL=dict(name="vo_cnn_rgb_d_dd_top_down", visual_backbone="resnet18", hidden_size=512,
                               visual_type=["rgb", "depth", "discretized_depth", "top_down_view"], dropout_p=0.2,
                               discretize_depth="hard", discretized_depth_channels=10, regress_type="sep_act", mode="det",
                               rnd_mode_n=10, pretrained=False)),
    TASK_CONFIG=dict(SIMULATOR=dict(DEPTH_SENSOR=dict(MIN_DEPTH=0.1, MAX_DEPTH=10.0, HFOV=70))))
dev = torch.device("cuda", 0)
t = BaseRLTrainerWithVO(cfg, dev)
t._set_up_vo_obs_transformer()
t._setup_vo_model(cfg)
for k in t.vo_model:
    sd = synth.make_state_dict(ms.state_dict_spec(t.vo_model[k].cfg), seed=1)
    t.vo_model[k].load_state_dict({n: torch.from_numpy(np.array(v)) for n, v in sd.items()})
obs = [synth.make_raw_obs(H, W, seed=3, index=i) for i in range(65)]
res = {}
for i in range(9):                       # warm-up touches all three action models (weight upload, workspaces)
    t._compute_local_delta_states_from_vo(obs[i], obs[i + 1], 1 + i % 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 200
for i in range(n):
    t._compute_local_delta_states_from_vo(obs[i % 64], obs[i % 64 + 1], 1 + i % 3)
res["batch1_boundary_ms"] = (time.perf_counter() - t0) / n * 1e3
for nb in (8, 64):
    prev, cur, acts = obs[:nb], obs[1:nb + 1], [1] * nb
    for _ in range(3):
        t.compute_local_delta_states_batch(prev, cur, acts)
    t0 = time.perf_counter()
    for _ in range(10):
        t.compute_local_delta_states_batch(prev, cur, acts)
    dt = (time.perf_counter() - t0) / 10
    res[f"batch{nb}_boundary_pairs_per_s_pcie_inclusive"] = nb / dt
# the same calls through the frame ring (env_ids): a rolling sequence per environment, so every call's prev frame IS the last call's
# cur frame (the navigation loop's `prev_obs = observations`) and only the new frame crosses PCIe
seq = [synth.make_raw_obs(H, W, seed=4, index=i) for i in range(64 + 40)]
for nb in (1, 8, 64):
    envs = list(range(nb))
    step = lambda k: t.compute_local_delta_states_batch([seq[e + k] for e in envs], [seq[e + k + 1] for e in envs], [1] * nb, env_ids=envs)
    for k in range(5):
        step(k)
    t0 = time.perf_counter()
    for k in range(5, 35):
        step(k)
    dt = (time.perf_counter() - t0) / 30
    assert t._ring_stats["ring_hits"] == nb, t._ring_stats
    if nb == 1:
        res["batch1_boundary_ring_ms"] = dt * 1e3
    else:
        res[f"batch{nb}_boundary_ring_pairs_per_s_pcie_inclusive"] = nb / dt
m = t.vo_model["forward"].eval()
o1 = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_obs_pairs(1, H, W, observation_space=cfg.VO.REGRESS_MODEL.visual_type, seed=2).items()}
with torch.no_grad():
    for _ in range(5):
        m(o1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        m(o1)
    torch.cuda.synchronize()
res["batch1_model_only_ms"] = (time.perf_counter() - t0) / 100 * 1e3
print(json.dumps(res))
