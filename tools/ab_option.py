"""A/B of one handle option on the headline forward: python tools/ab_option.py <option> <value_a> <value_b> [B]
prints forward ms and the per-layer event times of both settings (and the max output difference)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
opt, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda", 0)
model, sd = bench.build_model(dev)
obs = bench.make_inputs(B, dev, 0)
res = {}
for v in (va, vb, va, vb):
    model.set_option(opt, v)
    with torch.no_grad():
        for _ in range(5):
            out = model(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        model.timing(True)
        for _ in range(10):
            model(obs)
        torch.cuda.synchronize()
        kt = model.timing_read()
        model.timing(False)
    res.setdefault(v, []).append((dt, {k["name"]: k["total_ms"] / 10 for k in kt}, out.clone()))
for v in (va, vb):
    print(f"{opt}={v}: forward ms {[round(1e3 * r[0], 4) for r in res[v]]}  pairs/s {B / min(r[0] for r in res[v]):.0f}")
ka, kb = res[va][-1][1], res[vb][-1][1]
for n in sorted(ka, key=lambda n: -ka[n]):
    if abs(ka[n] - kb.get(n, 0)) > (0.002 if B >= 128 else 0.0007):
        print(f"   {n[-48:]:48s} {ka[n]:.4f} -> {kb.get(n, 0):.4f}")
print("max |diff| of outputs:", float((res[va][0][2] - res[vb][0][2]).abs().max()))
