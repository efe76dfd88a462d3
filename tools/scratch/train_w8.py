"""Training steps at 208 pairs (the eight-wave deep-stage forms engage from 200 pairs on) with x3_w8 on / off: losses and parameters must agree bit for bit."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd.train import VOTrainStep
dev = torch.device("cuda", 0)
B = 208
res = {}
for v in ("on", "off"):
    model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=bench.SPACE, observation_size=(bench.W, bench.H), hidden_size=512,
        backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=bench.BINS)
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=0)
    model.load_state_dict({k: torch.from_numpy(np.array(v_)) for k, v_ in sd.items()})
    model = model.to(dev)
    model.set_option("x3_w8", v)
    ts = VOTrainStep(model)
    obs = bench.make_inputs(B, dev, 0)
    tgt = (torch.arange(B * 3, device=dev, dtype=torch.float32).reshape(B, 3) % 7 - 3) * 0.05
    losses = []
    for _ in range(2):
        _, loss = ts.step(obs, tgt)
        losses.append(float(loss))
    torch.cuda.synchronize()
    p = torch.cat([q.detach().reshape(-1) for q in model.parameters()]).clone()
    res[v] = (losses, p)
    print(v, losses, float(p.abs().sum()))
print("losses equal:", res["on"][0] == res["off"][0], " params equal:", bool(torch.equal(res["on"][1], res["off"][1])), " finite:", bool(torch.isfinite(res["on"][1]).all()))
