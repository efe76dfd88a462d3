#!/usr/bin/env python
"""Model-only forward time of the default model vs batch size (observation tensors resident), fp32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model, sd = bench.build_model(dev)
full = bench.make_inputs(256, dev, 0)
with torch.no_grad():
    for B in ([int(x) for x in sys.argv[1:]] or (1, 2, 4, 8, 16, 32, 64, 128, 256)):
        obs = {k: v[:B].contiguous() for k, v in full.items()}
        for _ in range(5): model(obs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): model(obs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"B={B:4d}  {dt*1e3:7.3f} ms  {B/dt:9.0f} pairs/s")
