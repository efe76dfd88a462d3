import os, sys, cProfile, pstats, io, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_navloop as bn
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO
from pointnav_vo_amd.policy import PointNavResNetPolicy, policy_state_dict_spec
W, H = 341, 192
cfg = AttrDict(VO=dict(VO_TYPE="REGRESS", OBS_TRANSFORM="none", VIS_SIZE_W=W, VIS_SIZE_H=H,
    REGRESS_MODEL=dict(name="vo_cnn_rgb_d_dd_top_down", visual_backbone="resnet18", hidden_size=512, visual_type=["rgb", "depth", "discretized_depth", "top_down_view"], dropout_p=0.2,
                       discretize_depth="hard", discretized_depth_channels=10, regress_type="sep_act", mode="det", rnd_mode_n=10, pretrained=False)),
    TASK_CONFIG=dict(SIMULATOR=dict(DEPTH_SENSOR=dict(MIN_DEPTH=0.1, MAX_DEPTH=10.0, HFOV=70))))
dev = torch.device("cuda", 0)
t = BaseRLTrainerWithVO(cfg, dev); t._set_up_vo_obs_transformer(); t._setup_vo_model(cfg)
for k in t.vo_model:
    sd = synth.make_state_dict(ms.state_dict_spec(t.vo_model[k].cfg), seed=1)
    t.vo_model[k].load_state_dict({n: torch.from_numpy(np.array(v)) for n, v in sd.items()})
frames = [synth.make_raw_obs(H, W, seed=5, index=i) for i in range(96)]
def one(s): return t._compute_local_delta_states_from_vo(frames[s % 96], frames[(s + 1) % 96], s % 3 + 1)
for s in range(20): one(s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(200): one(s)
print("B=1 boundary ms", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for s in range(300): one(s)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(16); print(st.getvalue()[:3600])
space = bn.Space({"depth": bn.Box((H, W, 1)), "pointgoal_with_gps_compass": bn.Box((2,))})
pol = PointNavResNetPolicy(observation_space=space, action_space=bn.Act(), hidden_size=512, rnn_type="LSTM", num_recurrent_layers=2, backbone="resnet18", vis_types=["depth"])
psd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=0)
pol.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in psd.items()}); pol = pol.to(dev).eval()
E = 8
obs = {"depth": torch.rand(E, H, W, 1, device=dev), "pointgoal_with_gps_compass": torch.rand(E, 2, device=dev)}
hid = torch.zeros(pol.num_recurrent_layers, E, 512, device=dev); pa = torch.zeros(E, 1, dtype=torch.long, device=dev); mk = torch.ones(E, 1, device=dev)
for _ in range(20): pol.act(obs, hid, pa, mk)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): pol.act(obs, hid, pa, mk)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); print(st.getvalue()[:3200])
