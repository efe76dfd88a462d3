"""GPU check: the resident-weight stem (option stem_form=resident, stem_rs_kernel) against the tile-per-workgroup kernel
(stem_form=tiles): outputs must be bit-identical (same fragment, tap and K-split summation order); then the stem's time."""
import sys, os, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pointnav_vo_amd import synth, model_spec as ms
from pointnav_vo_amd.registry import baseline_registry

dev = torch.device("cuda", 0)
model, sd = bench.build_model(dev)
ok = True
for B in (19, 64, 256):
    obs = bench.make_inputs(B, dev, 0)
    rgb_f, dep_f = bench.frames_of(obs)
    for opts in ({}, {"pool": "separate"}):
        outs = {}
        for form in ("tiles", "resident", "fast"):
            model.set_option("stem_form", form)
            model.set_option("pool", opts.get("pool", "fused"))
            with torch.no_grad():
                a = model(obs).clone()
                b = model(obs).clone()
                r = model.forward_raw(rgb_f, dep_f, obs["top_down_view"]).clone()
            torch.cuda.synchronize()
            outs[form] = (a, b, r)
        same = [bool(torch.equal(outs["tiles"][k], outs["resident"][k])) for k in range(3)]
        fin = bool(torch.isfinite(outs["resident"][0]).all())
        d = [float((outs["tiles"][k] - outs["resident"][k]).abs().max()) for k in range(3)]
        print(f"B={B} {opts}: identical obs/obs2/raw = {same} finite = {fin} maxdiff = {d}", flush=True)
        ok = ok and all(same) and fin
if "--odd" in sys.argv:
    m2 = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=bench.SPACE, observation_size=(45, 37), hidden_size=512,
            backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd2 = synth.make_state_dict(ms.state_dict_spec(m2.cfg), seed=1)
    m2.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
    m2 = m2.to(dev).eval()
    o2 = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_obs_pairs(700, 37, 45, observation_space=bench.SPACE, dd_bins=10, seed=3).items()}
    res = {}
    for form in ("tiles", "resident", "fast"):
        m2.set_option("stem_form", form)
        with torch.no_grad():
            res[form] = m2(o2).clone()
    torch.cuda.synchronize()
    print("45x37 B=700 identical:", bool(torch.equal(res["tiles"], res["resident"])), float((res["tiles"] - res["resident"]).abs().max()))
    ok = ok and bool(torch.equal(res["tiles"], res["resident"]))
obs = bench.make_inputs(256, dev, 0)
rgb_f, dep_f = bench.frames_of(obs)
for form in ("tiles", "resident", "fast"):
    model.set_option("stem_form", form)
    model.set_option("pool", "fused")
    for raw in (False, True):
        run = (lambda: model.forward_raw(rgb_f, dep_f, obs["top_down_view"])) if raw else (lambda: model(obs))
        with torch.no_grad():
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            model.timing(True)
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            kt = model.timing_read()
            model.timing(False)
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        stem = [k for k in kt if k["name"].endswith("conv1.0")][0]
        print(f"{form} raw={raw}: stem {stem['total_ms'] / stem['launches']:.4f} ms per launch, forward {1e3 * dt:.3f} ms ({256 / dt:.0f} pairs/s)", flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
