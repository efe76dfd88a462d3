"""Grouped forward (three action models, one launch chain) against a single-model forward of the same size, with per-launch times."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pointnav_vo_amd import vo_cnn
dev = torch.device("cuda", 0)
models = [bench.build_model(dev, seed=k)[0] for k in range(3)]
H, W = 192, 341
def run(counts, timing=False):
    B = sum(counts)
    rgb = torch.randint(0, 256, (B, 2, H, W, 3), dtype=torch.uint8, device=dev)
    dep = torch.rand(B, 2, H, W, device=dev)
    tdv = torch.rand(B, H, W, 2, device=dev)
    ms = models[:len(counts)]
    f = (lambda: vo_cnn.grouped_forward_raw(ms, counts, rgb, dep, tdv)) if len(counts) > 1 else (lambda: ms[0].forward_raw(rgb, dep, tdv))
    with torch.no_grad():
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"counts {counts}: {dt*1e3:.3f} ms (host enqueue {th/50*1e3:.3f})")
        if timing:
            ms[0].timing(True)
            for _ in range(10): f()
            torch.cuda.synchronize()
            kt = ms[0].timing_read(); ms[0].timing(False)
            print("   launches", sum(k['launches'] for k in kt)//10, "sum", round(sum(k['total_ms'] for k in kt)/10, 3))
            for k in kt: print(f"   {k['name'][-46:]:46s} {k['total_ms']/10*1e3:7.1f} us x{k['launches']//10}")
run([8]); run([3, 3, 2], True); run([8], True); run([6,5,5]); run([16])
