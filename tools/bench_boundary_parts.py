#!/usr/bin/env python
"""Where the batched boundary call spends its time at N=64: staging threads, H2D, device work (diagnostic)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import _lib, synth
H, W, n = 192, 341, 64
obs = [synth.make_raw_obs(H, W, seed=3, index=i) for i in range(n + 1)]
rgb = [o["rgb"] for pc in zip(obs[:n], obs[1:]) for o in pc]
dep = [o["depth"] for pc in zip(obs[:n], obs[1:]) for o in pc]
h_rgb = torch.empty((n, 2, H, W, 3), dtype=torch.uint8).pin_memory(); h_dep = torch.empty((n, 2, H, W), dtype=torch.float32).pin_memory()
d_rgb = torch.empty_like(h_rgb, device="cuda"); d_dep = torch.empty_like(h_dep, device="cuda")
pr = (C.c_void_p * (2 * n))(*[a.ctypes.data for a in rgb]); pd = (C.c_void_p * (2 * n))(*[a.ctypes.data for a in dep])
def t(f, k=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for thr in (1, 4, 8, 16, 32):
    print("stage rgb+depth, threads", thr, round(t(lambda: (_lib.lib.pnvo_stage_frames(pr, 2 * n, H * W * 3, C.c_void_p(h_rgb.data_ptr()), thr),
                                                           _lib.lib.pnvo_stage_frames(pd, 2 * n, H * W * 4, C.c_void_p(h_dep.data_ptr()), thr))), 3), "ms")
print("H2D rgb+depth (59 MB pinned)", round(t(lambda: (d_rgb.copy_(h_rgb, non_blocking=True), d_dep.copy_(h_dep, non_blocking=True))), 3), "ms")
print("np.stack equivalent", round(t(lambda: (np.stack(rgb), np.stack(dep)), 3), 3), "ms")
