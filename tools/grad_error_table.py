#!/usr/bin/env python
"""Per-tensor relative L2 error of the HIP training step's gradients vs the fp64 checker, next to the deviation of the SAME
checker run in float32 (torch CPU) from its float64 self — the yardstick for the gradient tolerance of tests/test_gpu_train.py."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from test_gpu_train import build
from oracle import torch_train_ref as ref
from pointnav_vo_amd.train import VOTrainStep
for fname in ("train_default_96x64_b3_f32.npz", "train_default_45x37_b4_f64.npz"):
    rec = load_golden(fname)
    model, cfg, sd, obs, tobs = build(rec)
    ts = VOTrainStep(model, lr=float(rec["lr"]), eps=float(rec["eps"]))
    ts.forward_backward(tobs, target=torch.from_numpy(rec["target"]).to("cuda:0"))
    torch.cuda.synchronize()
    c64 = ref.train_step(sd, obs, rec["target"], ngroups=cfg.ngroups, dtype=torch.float64)
    c32 = ref.train_step(sd, obs, rec["target"], ngroups=cfg.ngroups, dtype=torch.float32)
    rows = []
    for name, (off, n) in ts.offsets.items():
        g64 = c64["grads"][name].reshape(-1).numpy()
        gh = ts.grad[off:off + n].cpu().double().numpy()
        g32 = c32["grads"][name].reshape(-1).double().numpy()
        nr = max(np.linalg.norm(g64), 1e-30)
        rows.append((name, np.linalg.norm(gh - g64) / nr, np.linalg.norm(g32 - g64) / nr))
    rows.sort(key=lambda r: -r[1])
    print(fname, "worst HIP", rows[0], " worst torch-fp32", max(rows, key=lambda r: r[2]))
    print("  median HIP %.2e  median torch-fp32 %.2e" % (np.median([r[1] for r in rows]), np.median([r[2] for r in rows])))
    for r in rows[:6]:
        print("   %-58s HIP %.2e   torch fp32 %.2e" % r)
