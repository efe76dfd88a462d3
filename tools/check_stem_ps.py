"""GPU check: the persistent role-specialised stem (option stem_form=persistent) against the tile-per-workgroup kernel
(stem_form=tiles): outputs must be bit-identical (same fragment, tap and K-split summation order)."""
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
for B in (8, 19, 64, 256):
    obs = bench.make_inputs(B, dev, 0)
    rgb_f, dep_f = bench.frames_of(obs)
    for opts in ({}, {"pool": "separate"}):
        outs = {}
        for form in ("tiles", "persistent", "persistent8"):
            model.set_option("stem_form", form.rstrip("8"))
            model.set_option("stem_lwaves", 8 if form.endswith("8") else 4)
            model.set_option("pool", opts.get("pool", "fused"))
            with torch.no_grad():
                a = model(obs).clone()
                b = model(obs).clone()
                r = model.forward_raw(rgb_f, dep_f, obs["top_down_view"]).clone()
            torch.cuda.synchronize()
            outs[form] = (a, b, r)
        same = [bool(torch.equal(outs["tiles"][k], outs["persistent"][k])) for k in range(3)]
        rep = bool(torch.equal(outs["persistent"][0], outs["persistent"][1]))
        fin = bool(torch.isfinite(outs["persistent"][0]).all())
        d = float((outs["tiles"][0] - outs["persistent"][0]).abs().max())
        rep8 = bool(torch.equal(outs["persistent8"][0], outs["persistent8"][1])) and bool(torch.equal(outs["persistent8"][0], outs["persistent8"][2]))
        d8 = float((outs["tiles"][0] - outs["persistent8"][0]).abs().max() / outs["tiles"][0].abs().max())
        print(f"B={B} {opts}: identical obs/obs2/raw = {same}  reproducible = {rep} finite = {fin} maxdiff = {d:.3e} | 8 L waves: reproducible+raw-identical {rep8} rel diff {d8:.2e}")
        ok = ok and all(same) and rep and fin and rep8 and d8 < 2e-6
# small odd resolution, many pairs
m2 = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=bench.SPACE, observation_size=(45, 37), hidden_size=512,
        backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
sd2 = synth.make_state_dict(ms.state_dict_spec(m2.cfg), seed=1)
m2.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
m2 = m2.to(dev).eval()
o2 = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_obs_pairs(300, 37, 45, observation_space=bench.SPACE, dd_bins=10, seed=3).items()}
res = {}
for form in ("tiles", "persistent"):
    m2.set_option("stem_form", form)
    m2.set_option("stem_lwaves", 4)
    with torch.no_grad():
        res[form] = m2(o2).clone()
torch.cuda.synchronize()
print("45x37 B=300 identical:", bool(torch.equal(res["tiles"], res["persistent"])), float((res["tiles"] - res["persistent"]).abs().max()))
ok = ok and bool(torch.equal(res["tiles"], res["persistent"]))
# timing of the stem alone at 256 pairs
obs = bench.make_inputs(256, dev, 0)
for form in ("tiles", "persistent", "persistent8"):
    model.set_option("stem_form", form.rstrip("8"))
    model.set_option("stem_lwaves", 8 if form.endswith("8") else 4)
    model.set_option("pool", "fused")
    with torch.no_grad():
        for _ in range(5):
            model(obs)
        torch.cuda.synchronize()
        model.timing(True)
        for _ in range(20):
            model(obs)
        torch.cuda.synchronize()
        kt = model.timing_read()
        model.timing(False)
        t0 = time.perf_counter()
        for _ in range(20):
            model(obs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    stem = [k for k in kt if k["name"].endswith("conv1.0")][0]
    print(f"{form}: stem {stem['total_ms'] / stem['launches']:.4f} ms per launch, forward {1e3 * dt:.3f} ms ({256 / dt:.0f} pairs/s)")
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
