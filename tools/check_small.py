#!/usr/bin/env python
"""Small-batch persistent kernel (smallnet.hip) against the per-layer launches: outputs and model-only time, B = 1..4."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model, sd = bench.build_model(dev)
full = bench.make_inputs(8, dev, 0)
coops = [int(x) for x in os.environ.get("COOPS", "1,0").split(",")]
with torch.no_grad():
    for B in (1, 2, 3, 4):
        obs = {k: v[:B].contiguous() for k, v in full.items()}
        model.set_option("small_net", "off")
        ref = model(obs).clone()
        torch.cuda.synchronize()
        def t(n=50):
            for _ in range(5): model(obs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): model(obs)
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        t_off = t()
        model.set_option("small_net", "on"); model.set_option("small_max", "4")
        for coop in coops:
            model.set_option("small_coop", str(coop))
            out = model(obs).clone()
            torch.cuda.synchronize()
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            t_on = t()
            print(f"B={B} coop={coop} rel_err={err:.3e}  per-layer {t_off:.3f} ms  persistent {t_on:.3f} ms  note={model.last_note()!r}", flush=True)
    if os.environ.get("PROF"):
        obs = {k: v[:1].contiguous() for k, v in full.items()}
        for lvl in os.environ.get("PROF", "1").split(","):
            model.set_option("small_prof", lvl)
            model(obs); model(obs)
        model.set_option("small_prof", "0")
    if os.environ.get("STRESS"):
        # coherence stress: alternate inputs, every result against the per-layer path's
        obs_list = [{k: v[i:i + 1].contiguous() for k, v in full.items()} for i in range(4)]
        model.set_option("small_net", "off")
        refs = [model(o).clone() for o in obs_list]
        model.set_option("small_net", "on")
        worst = 0.0
        for it in range(int(os.environ["STRESS"])):
            i = (it * 7 + it // 3) % 4
            out = model(obs_list[i])
            err = ((out - refs[i]).abs().max() / refs[i].abs().max()).item()
            worst = max(worst, err)
        print(f"stress: worst rel err over {os.environ['STRESS']} alternating forwards = {worst:.3e}", flush=True)
