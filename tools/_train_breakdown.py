import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd.train import VOTrainStep
dev = torch.device("cuda", 0)
model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=bench.SPACE, observation_size=(bench.W, bench.H), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.0, discretized_depth_channels=10)
sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=0)
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}); model = model.to(dev)
ts = VOTrainStep(model); B = 128
obs = bench.make_inputs(B, dev, 0); tgt = (torch.rand((B, 3), device=dev) - 0.5) * 0.5
for _ in range(2): ts.step(obs, tgt)
model.timing(True); torch.cuda.synchronize()
for _ in range(3): ts.step(obs, tgt)
torch.cuda.synchronize()
for k in sorted(model.timing_read(), key=lambda k: -k["total_ms"])[:28]:
    print(f'{k["name"][-52:]:54s} {k["total_ms"]/3:8.3f} ms  {k["flops"]/max(k["total_ms"],1e-9)/1e9*1e-3*3/3:7.1f} TF' )
