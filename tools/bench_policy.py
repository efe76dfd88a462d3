#!/usr/bin/env python
"""Per-step latency / throughput of the HIP navigation policy (PointNavResNetPolicy.act)
at the batch sizes a nav loop uses (B = environments per process), with the oracle port timed beside it.
    python tools/bench_policy.py [--envs 1 4 16 64]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import synth  # noqa: E402
from pointnav_vo_amd.policy import PointNavResNetPolicy, policy_state_dict_spec  # noqa: E402

H, W = 192, 341


class Box:
    def __init__(self, shape):
        self.shape = shape


class Space:
    def __init__(self, d):
        self.spaces = d


class Act:
    n = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[1, 4, 16, 64])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    space = Space({"depth": Box((H, W, 1)), "pointgoal_with_gps_compass": Box((2,))})
    pol = PointNavResNetPolicy(observation_space=space, action_space=Act(), hidden_size=512, rnn_type="LSTM",
                               num_recurrent_layers=2, backbone="resnet18", vis_types=["depth"])
    sd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=0)
    pol.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    pol = pol.to(dev).eval()
    res = {"metric": "navigation-policy act() steps", "frame": f"{W}x{H} depth", "dtype": "f32", "results": []}
    for B in a.envs:
        depth, goal, prev, mask = synth.make_policy_inputs(H, W, B, 1, 1)[0]
        obs = {"depth": torch.from_numpy(depth).to(dev), "pointgoal_with_gps_compass": torch.from_numpy(goal).to(dev)}
        hid = torch.zeros(pol.num_recurrent_layers, B, 512, device=dev)
        pa, mk = torch.from_numpy(prev).view(B, 1).to(dev), torch.ones(B, 1, device=dev)
        for _ in range(5):
            _, _, _, hid = pol.act(obs, hid, pa, mk, deterministic=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            _, act, _, hid = pol.act(obs, hid, pa, mk, deterministic=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        res["results"].append({"envs": B, "ms_per_step": dt * 1e3, "frames_per_s": B / dt})
    if not a.no_cpu_baseline:
        from oracle import oracle, policy_oracle
        B = 4
        depth, goal, prev, mask = synth.make_policy_inputs(H, W, B, 1, 1)[0]
        hid = np.zeros((4, B, 512), np.float32)
        policy_oracle.policy_step(sd, depth, goal, prev, mask, hid, dtype=np.float32)
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            policy_oracle.policy_step(sd, depth, goal, prev, mask, hid, dtype=np.float32)
        dt = (time.perf_counter() - t0) / n
        res["cpu_baseline"] = {"envs": B, "ms_per_step": dt * 1e3, "frames_per_s": B / dt, "kind": "port",
                               "cores": oracle.usable_cores() if hasattr(oracle, "usable_cores") else None}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
