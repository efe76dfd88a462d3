"""GPU stress of the resident-weight stems against the tile kernel on assorted (pairs, width, height): `resident` must be
bit-identical, `fast` float32-grade equal (1e-5 of the output range; measured 1.5-4e-6) and reproducible; sensor-frame entry included at full size."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pointnav_vo_amd import synth, model_spec as ms
from pointnav_vo_amd.registry import baseline_registry
dev = torch.device("cuda", 0)
ok = True
for (W, H, Bs) in ((341, 192, (16, 17, 33, 100)), (128, 96, (90, 257)), (45, 37, (300, 701)), (64, 48, (350,)), (33, 65, (512,))):
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=bench.SPACE, observation_size=(W, H), hidden_size=512,
            backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=2)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    for B in Bs:
        o = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_obs_pairs(B, H, W, observation_space=bench.SPACE, dd_bins=10, seed=B).items()}
        res = {}
        for form in ("tiles", "resident", "fast", "fast"):
            m.set_option("stem_form", form)
            with torch.no_grad():
                res.setdefault(form, []).append(m(o).clone())
        torch.cuda.synchronize()
        ident = bool(torch.equal(res["tiles"][0], res["resident"][0]))
        rep = bool(torch.equal(res["fast"][0], res["fast"][1]))
        rel = float((res["tiles"][0] - res["fast"][0]).abs().max() / res["tiles"][0].abs().max())
        fin = bool(torch.isfinite(res["fast"][0]).all())
        print(f"{W}x{H} B={B}: resident identical {ident}  fast reproducible {rep} rel {rel:.2e} finite {fin}", flush=True)
        ok = ok and ident and rep and fin and rel < 1e-5
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
