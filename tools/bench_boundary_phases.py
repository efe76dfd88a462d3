#!/usr/bin/env python
"""Where compute_local_delta_states_batch spends a call at N pairs (diagnostic): host-side phases by perf_counter with a
device synchronisation after each (so nothing overlaps: the SUM is the un-pipelined cost), then the real pipelined call."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import _lib, model_spec as ms, synth
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO
import ctypes as C
W, H = 341, 192
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
obs = [synth.make_raw_obs(H, W, seed=3, index=i) for i in range(N + 1)]
prev, cur, acts = obs[:N], obs[1:N + 1], [1] * N
for _ in range(3):
    t.compute_local_delta_states_batch(prev, cur, acts)
st = t._bstage
model = t.vo_model["forward"]
gen = t._top_down_view_generator
frames = [o for pc in zip(prev, cur) for o in pc]
res = {}
def timed(name, f, k=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    res[name] = round((time.perf_counter() - t0) / k * 1e3, 3)
p = lambda x: C.c_void_p(x.data_ptr())
def ptrs():
    return t._frame_ptrs([f["rgb"] for f in frames], np.uint8, (H, W, 3)), t._frame_ptrs([f["depth"] for f in frames], np.float32, (H, W))
timed("python: pointer arrays over 2N frames", ptrs)
(pr, _k1), (pd, _k2) = ptrs()
timed("host gather into pinned staging (4 threads)", lambda: (_lib.lib.pnvo_stage_frames(pr, 2 * N, H * W * 3, p(st["h_rgb"]), 4),
                                                              _lib.lib.pnvo_stage_frames(pd, 2 * N, H * W * 4, p(st["h_dep"]), 4)))
for thr in (2, 8, 12, 16, 24):
    timed(f"host gather, {thr} threads", lambda: (_lib.lib.pnvo_stage_frames(pr, 2 * N, H * W * 3, p(st["h_rgb"]), thr),
                                                   _lib.lib.pnvo_stage_frames(pd, 2 * N, H * W * 4, p(st["h_dep"]), thr)))
timed("H2D (pinned -> device)", lambda: (st["d_rgb"][:N].copy_(st["h_rgb"][:N], non_blocking=True), st["d_dep"][:N].copy_(st["h_dep"][:N], non_blocking=True)))
def tdv():
    gen.gen_top_down_view_pairs(st["d_dep"][:N], st["tdv"][:N])
timed("top-down views of 2N frames", tdv)
with torch.no_grad():
    timed("forward_raw over N pairs", lambda: model.forward_raw(st["d_rgb"][:N], st["d_dep"][:N], st["tdv"][:N], err_flag=st["flag"]))
    o = model.forward_raw(st["d_rgb"][:N], st["d_dep"][:N], st["tdv"][:N])
    timed("result .cpu()", lambda: o.cpu())
copy = torch.cuda.Stream(dev)
def gather_h2d(nch):
    main = torch.cuda.current_stream(dev)
    copy.wait_stream(main)
    for c in range(nch):
        lo, hi = N * c // nch, N * (c + 1) // nch
        m = hi - lo
        with torch.cuda.stream(copy):
            pr_, _a = t._frame_ptrs([f["rgb"] for f in frames[2 * lo:2 * hi]], np.uint8, (H, W, 3))
            _lib.lib.pnvo_stage_frames(pr_, 2 * m, H * W * 3, C.c_void_p(st["h_rgb"][lo:].data_ptr()), 12)
            st["d_rgb"][lo:hi].copy_(st["h_rgb"][lo:hi], non_blocking=True)
            pd_, _b = t._frame_ptrs([f["depth"] for f in frames[2 * lo:2 * hi]], np.float32, (H, W))
            _lib.lib.pnvo_stage_frames(pd_, 2 * m, H * W * 4, C.c_void_p(st["h_dep"][lo:].data_ptr()), 12)
            st["d_dep"][lo:hi].copy_(st["h_dep"][lo:hi], non_blocking=True)
        main.wait_stream(copy)
for nch in (1, 2, 4, 8):
    timed(f"gather + H2D pipelined in {nch} chunks (no device compute)", lambda: gather_h2d(nch))
def host_only(nch):
    t0 = time.perf_counter()
    gather_h2d(nch)
    return time.perf_counter() - t0
torch.cuda.synchronize()
res["host time of the 4-chunk gather + H2D enqueue"] = round(sum(host_only(4) for _ in range(10)) / 10 * 1e3, 3)
torch.cuda.synchronize()
for thr in (4, 8, 12, 16):
    t.stage_threads = thr
    timed(f"whole call, {thr} staging threads", lambda: t.compute_local_delta_states_batch(prev, cur, acts))
t.stage_threads = 12
for nch in (1, 2, 3, 4, 6):
    t.boundary_chunks = nch
    timed(f"whole call, {nch} chunks", lambda: t.compute_local_delta_states_batch(prev, cur, acts))
t.boundary_chunks = None
timed("whole call (pipelined)", lambda: t.compute_local_delta_states_batch(prev, cur, acts))
res["pairs_per_s"] = round(N / (res["whole call (pipelined)"] * 1e-3))
res["N"] = N
print(json.dumps(res, indent=1))
