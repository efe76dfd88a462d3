"""Per-launch HIP-event times of one forward at the given batch sizes: python tools/layer_breakdown.py 8 16 32 256"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
model, sd = bench.build_model(dev)
for B in [int(x) for x in sys.argv[1:]]:
    obs = bench.make_inputs(B, dev, 0)
    with torch.no_grad():
        for _ in range(5): model(obs)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(50): model(obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        model.timing(True)
        for _ in range(10): model(obs)
        torch.cuda.synchronize()
        kt = model.timing_read()
        model.timing(False)
    print(f"== B={B}: forward {dt*1e3:.3f} ms; sum of event times {sum(k['total_ms'] for k in kt)/10:.3f} ms; launches {sum(k['launches'] for k in kt)//10}")
    for k in kt:
        fam = ""
        if k["name"].startswith("conv:visual_encoder.backbone.layer"):
            fam = model.layer_kernel(k["name"][5:], B)[0]
        print(f"   {k['name'][-46:]:46s} {k['total_ms']/10*1e3:7.1f} us x{k['launches']//10} {fam}")
