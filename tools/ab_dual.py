"""A/B of one handle option on the bf16 dual forward (BASELINE configs[2]): python tools/ab_dual.py <option> <a> <b> [B]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pointnav_vo_amd import vo_cnn
opt, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda", 0)
ma, _ = bench.build_model(dev, seed=0)
mb, _ = bench.build_model(dev, seed=1)
for m in (ma, mb):
    m.set_precision("bfloat16")
obs = bench.make_inputs(B, dev, 0)
res = {}
for v in (va, vb, va, vb):
    for m in (ma, mb):
        m.set_option(opt, v)
    with torch.no_grad():
        for _ in range(5):
            oa, ob = vo_cnn.dual_forward(ma, mb, obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            vo_cnn.dual_forward(ma, mb, obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        ma.timing(True)
        for _ in range(10):
            vo_cnn.dual_forward(ma, mb, obs)
        torch.cuda.synchronize()
        kt = ma.timing_read()
        ma.timing(False)
    res.setdefault(v, []).append((dt, {k["name"]: k["total_ms"] / 10 for k in kt}, oa.clone(), ob.clone()))
for v in (va, vb):
    print(f"{opt}={v}: dual forward ms {[round(1e3 * r[0], 4) for r in res[v]]}  dual pairs/s {B / min(r[0] for r in res[v]):.0f}")
ka, kb = res[va][-1][1], res[vb][-1][1]
for n in sorted(ka, key=lambda n: -ka[n]):
    if abs(ka[n] - kb.get(n, 0)) > 0.003:
        print(f"   {n[-52:]:52s} {ka[n]:.4f} -> {kb.get(n, 0):.4f}")
print("max |diff| a, b:", float((res[va][0][2] - res[vb][0][2]).abs().max()), float((res[va][0][3] - res[vb][0][3]).abs().max()))
