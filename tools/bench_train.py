#!/usr/bin/env python
"""Throughput of the VO training step (BASELINE config 4 shape: 128 frame pairs per GPU, fwd + bwd + Adam, fp32).
    python tools/bench_train.py [--batch 128] [--steps 5]
    python -m torch.distributed.run --nproc-per-node N tools/bench_train.py   (adds the RCCL gradient all-reduce)"""
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
from pointnav_vo_amd.train import VOTrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--detail", action="store_true", help="also print every timed kernel")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    lr = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    dev = torch.device("cuda", lr)
    torch.cuda.set_device(dev)
    model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(bench.W, bench.H), hidden_size=512, backbone="resnet18",
        normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=bench.BINS)   # reference p
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=0)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev)
    ts = VOTrainStep(model)
    obs = bench.make_inputs(a.batch, dev, rank)
    tgt = (torch.rand((a.batch, 3), device=dev) - 0.5) * 0.5
    for _ in range(a.warmup):
        ts.step(obs, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, loss = ts.step(obs, tgt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    model.timing(True)                    # per-kernel breakdown: a separate pass (event pairs serialise the launches)
    for _ in range(a.steps):
        ts.step(obs, tgt)
    torch.cuda.synchronize()
    kt = model.timing_read()
    model.timing(False)
    if rank == 0:
        flops = 3 * 2.0 * ms.macs_per_pair(model.cfg) * a.batch
        agg = {}
        for k in kt:
            key = k["name"].split(":")[0]
            agg[key] = agg.get(key, 0.0) + k["total_ms"] / a.steps
        if a.detail:
            for k in sorted(kt, key=lambda k: -k["total_ms"]):
                ms_ = k["total_ms"] / a.steps
                tf = k["flops"] / a.steps / (ms_ * 1e-3) / 1e12 if k["flops"] else 0.0
                print(f"  {k['name'][-52:]:52s} {ms_:8.3f} ms  {tf:6.1f} TF  x{k['launches'] // a.steps}", file=sys.stderr)
        print(json.dumps({"metric": "VO training step (fwd+bwd+Adam) frame-pairs/s", "value": world * a.batch / dt,
                          "ms_per_step": dt * 1e3, "pairs_per_gpu": a.batch, "n_gpus": world, "loss": float(loss),
                          "tflops_3x_fwd": flops / dt / 1e12, "ms_by_kernel_class": agg}))


if __name__ == "__main__":
    main()
