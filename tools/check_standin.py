"""The float32 stand-in stem (input-contract repair) against the fused stem and the classic dense path, at a few batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

DEV = torch.device("cuda", 0)
for B, pool in ((2, "fused"), (4, "fused"), (16, "separate"), (16, "fused"), (64, "fused")):
    model, sd = bench.build_model(DEV)
    model.set_option("pool", pool)
    obs = bench.make_inputs(B, DEV, 0)
    bad = dict(obs)
    bad["rgb"] = obs["rgb"].clone()
    bad["rgb"][B - 1, 100, 200, 2] = 17.3
    with torch.no_grad():
        o_mx = model(obs).clone()
        o_bad = model(bad).clone()
        torch.cuda.synchronize()
        st = model.get_option("stem")
        o_st = model(obs).clone()
        o_bad2 = model(bad).clone()
        model.set_option("stem", "dense")
        o_dn = model(obs).clone()
        o_bad3 = model(bad).clone()
        torch.cuda.synchronize()
    r = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print(B, pool, st, "standin vs mx", r(o_st, o_mx), "dense vs mx", r(o_dn, o_mx), "| bad: repaired vs dense", r(o_bad, o_bad3),
          "standin vs dense", r(o_bad2, o_bad3), flush=True)
