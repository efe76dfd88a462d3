import os, sys, cProfile, pstats, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_navloop as bn
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO
W, H, E = 341, 192, 8
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
env_ids = list(range(E))
prev = [frames[e] for e in range(E)]
def step(s):
    global prev
    cur = [frames[(e + s + 1) % 96] for e in range(E)]
    acts = [(e + s) % 3 + 1 for e in range(E)]
    d = t.compute_local_delta_states_batch(prev, cur, acts, env_ids=env_ids)
    prev = cur
for s in range(20): step(s)
pr = cProfile.Profile(); pr.enable()
for s in range(20, 320): step(s)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
