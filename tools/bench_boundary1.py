#!/usr/bin/env python
"""Batch-1 boundary call only (for rocprofv3 / cProfile): N calls of _compute_local_delta_states_from_vo."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import model_spec as ms, synth
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO
W, H = 341, 192
cfg = AttrDict(
    VO=dict(VO_TYPE="REGRESS", OBS_TRANSFORM="none", VIS_SIZE_W=W, VIS_SIZE_H=H,
            REGRESS_MODEL=dict(name="vo_cnn_rgb_d_dd_top_down", visual_backbone="resnet18", hidden_size=512,
                               visual_type=["rgb", "depth", "discretized_depth", "top_down_view"], dropout_p=0.2,
                               discretize_depth="hard", discretized_depth_channels=10, regress_type="sep_act", mode="det",
                               rnd_mode_n=10, pretrained=False)),
    TASK_CONFIG=dict(SIMULATOR=dict(DEPTH_SENSOR=dict(MIN_DEPTH=0.1, MAX_DEPTH=10.0, HFOV=70))))
dev = torch.device("cuda", 0)
t = BaseRLTrainerWithVO(cfg, dev)
t._set_up_vo_obs_transformer()
t._setup_vo_model(cfg)
for k in t.vo_model:
    sd = synth.make_state_dict(ms.state_dict_spec(t.vo_model[k].cfg), seed=1)
    t.vo_model[k].load_state_dict({n: torch.from_numpy(np.array(v)) for n, v in sd.items()})
obs = [synth.make_raw_obs(H, W, seed=3, index=i) for i in range(9)]
for i in range(6):
    t._compute_local_delta_states_from_vo(obs[i], obs[i + 1], 1 + i % 3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.cuda.synchronize()
pr = cProfile.Profile() if "--cprofile" in sys.argv else None
if pr: pr.enable()
t0 = time.perf_counter()
for i in range(n):
    t._compute_local_delta_states_from_vo(obs[i % 8], obs[i % 8 + 1], 1 + i % 3)
dt = (time.perf_counter() - t0) / n
if pr:
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
print("batch1_boundary_ms", dt * 1e3)
