#!/usr/bin/env python
"""Throughput of the device-side dataset batcher: one HDF5-chunk-shaped dict of numpy arrays
(uint8 rgb, float16 depth) -> training batch on the GPU (float32 NHWC pairs, one-hot depth, top-down views), H2D included,
with the numpy oracle port of the reference's per-sample path timed beside it on a few samples.
    python tools/bench_dataset.py [--samples 128] [--reps 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import synth  # noqa: E402
from pointnav_vo_amd.dataset import StatePairBatcher  # noqa: E402

H, W = 192, 341


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    infos = dict(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=np.deg2rad(70.0), rows_around_center=50)
    ch = synth.make_dataset_chunk(a.samples, H, W, seed=3)
    b = StatePairBatcher(W, H, act_type=-1, discretize_depth="hard", discretized_depth_channels=10, gen_top_down_view=True,
                         top_down_view_infos=infos)
    out = b.process_chunk(ch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = b.process_chunk(ch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    M = out["rgb_pairs"].shape[0]
    res = {"metric": "dataset entries prepared on the device (H2D of the stored arrays included)", "unit": "entries/s",
           "value": M / dt, "ms_per_chunk": dt * 1e3, "entries": M, "frame": f"{W}x{H}", "dtype": "u8/f16 -> f32"}
    if not a.no_cpu_baseline:
        from oracle import dataset_oracle as do
        n = min(4, a.samples)
        t0 = time.perf_counter()
        for i in range(n):
            do.process_sample(ch, i, H=H, W=W, act_type=-1, bins=10, tdv_infos=dict(infos))
        res["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "entries/s", "cores": 1, "kind": "port",
                               "sample": f"{n} samples through oracle/dataset_oracle.py"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
