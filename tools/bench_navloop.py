#!/usr/bin/env python
"""GPU side of the navigation eval loop (BASELINE config 5; reference loop: pointnav_vo/rl/ppo/ppo_trainer.py:724-891), per
simulator step and for E parallel environments of one process:

    policy.act(depth, pointgoal, prev_action, mask)            (:760-770)   -> actions
    [simulator step — NOT here: synthetic frames stand in for what the sim workers would return]
    VO on (prev_obs, cur_obs, action) of every env             (:836-841)   -> (dx, dz, dyaw)
    goal update compute_goal_pos per env                        (:843-860)   -> next pointgoal

Everything the reference does on the trainer process between two simulator steps, with host numpy observations as the
simulator delivers them (uint8 rgb + float32 depth, PCIe copies included).  Reports env-steps/s and the eval wall-clock this
implies for the GPU part of 994 episodes (the reference's full run took 4.5 h including simulation).
    python tools/bench_navloop.py [--envs 8 32] [--steps 30]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnav_vo_amd import geometry, model_spec as ms, synth  # noqa: E402
from pointnav_vo_amd.policy import PointNavResNetPolicy, policy_state_dict_spec  # noqa: E402
from pointnav_vo_amd.trainer import AttrDict, BaseRLTrainerWithVO  # noqa: E402

W, H = 341, 192


class Box:
    def __init__(self, shape):
        self.shape = shape


class Space:
    def __init__(self, d):
        self.spaces = d


class Act:
    n = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[8, 32])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--episode-steps", type=float, default=150.0, help="mean steps per episode used for the extrapolation")
    a = ap.parse_args()
    print(json.dumps(run(a.envs, a.steps, a.episode_steps, torch.device("cuda", 0))))


def run(envs, steps, episode_steps=150.0, dev=None):
    """-> the result record (also what bench.py reports under secondary["navloop_gpu_side"])."""
    import types
    a = types.SimpleNamespace(envs=envs, steps=steps, episode_steps=episode_steps)
    dev = dev if dev is not None else torch.device("cuda", 0)
    cfg = AttrDict(
        VO=dict(VO_TYPE="REGRESS", OBS_TRANSFORM="none", VIS_SIZE_W=W, VIS_SIZE_H=H,
                REGRESS_MODEL=dict(name="vo_cnn_rgb_d_dd_top_down", visual_backbone="resnet18", hidden_size=512,
                                   visual_type=["rgb", "depth", "discretized_depth", "top_down_view"], dropout_p=0.2,
                                   discretize_depth="hard", discretized_depth_channels=10, regress_type="sep_act", mode="det",
                                   rnd_mode_n=10, pretrained=False)),
        TASK_CONFIG=dict(SIMULATOR=dict(DEPTH_SENSOR=dict(MIN_DEPTH=0.1, MAX_DEPTH=10.0, HFOV=70))))
    t = BaseRLTrainerWithVO(cfg, dev)
    t._set_up_vo_obs_transformer()
    t._setup_vo_model(cfg)
    for k in t.vo_model:
        sd = synth.make_state_dict(ms.state_dict_spec(t.vo_model[k].cfg), seed=1)
        t.vo_model[k].load_state_dict({n: torch.from_numpy(np.array(v)) for n, v in sd.items()})
    space = Space({"depth": Box((H, W, 1)), "pointgoal_with_gps_compass": Box((2,))})
    pol = PointNavResNetPolicy(observation_space=space, action_space=Act(), hidden_size=512, rnn_type="LSTM",
                               num_recurrent_layers=2, backbone="resnet18", vis_types=["depth"])
    psd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=0)
    pol.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in psd.items()})
    pol = pol.to(dev).eval()
    frames = [synth.make_raw_obs(H, W, seed=5, index=i) for i in range(96)]
    res = {"metric": "nav-loop env-steps (policy act + VO + goal update; simulator excluded)", "unit": "env-steps/s",
           "frame": f"{W}x{H}", "dtype": "f32", "results": []}
    for E in a.envs:
        goals = [np.array([0.0, 0.0, -5.0]) for _ in range(E)]           # goal in the agent frame (x, y, z)
        hid = torch.zeros(pol.num_recurrent_layers, E, 512, device=dev)
        prev_a = torch.zeros(E, 1, dtype=torch.long, device=dev)
        masks = torch.ones(E, 1, device=dev)
        prev_obs = [frames[e % 96] for e in range(E)]
        env_ids = list(range(E)) if os.environ.get("PNVO_NAVLOOP_RING", "1") != "0" else None
        ring_depth = os.environ.get("PNVO_NAVLOOP_RING_DEPTH", "1") != "0"

        phases = os.environ.get("PNVO_NAVLOOP_PHASES") is not None      # developer: host-synchronised time of each part (slower loop)
        ph = {"policy input (stack + H2D)": 0.0, "policy.act": 0.0, "actions to host": 0.0, "VO boundary call": 0.0, "goal update": 0.0}

        def mark(name, t_prev):
            if not phases:
                return t_prev
            torch.cuda.synchronize()
            now = time.perf_counter()
            ph[name] += now - t_prev
            return now

        def step(s):
            nonlocal hid, prev_a, prev_obs
            tp = time.perf_counter()
            if env_ids is not None and ring_depth:       # the policy's depth = the frames the last VO call left in the device ring
                depth = t.ring_depth(prev_obs, env_ids)
            else:
                depth = torch.from_numpy(np.stack([o["depth"] for o in prev_obs])).to(dev, non_blocking=True)
            polar = geometry.compute_goal_pos_batch(np.stack(goals), np.zeros((E, 3)))["polar"]
            obs = {"depth": depth, "pointgoal_with_gps_compass": torch.from_numpy(polar).to(dev)}
            tp = mark("policy input (stack + H2D)", tp)
            _, act, _, hid = pol.act(obs, hid, prev_a, masks, deterministic=False)
            tp = mark("policy.act", tp)
            acts = (act.view(-1).cpu().numpy() % 3 + 1).tolist()          # STOP never ends a synthetic episode here
            tp = mark("actions to host", tp)
            cur_obs = [frames[(e + s + 1) % 96] for e in range(E)]
            deltas = t.compute_local_delta_states_batch(prev_obs, cur_obs, acts, env_ids=env_ids)   # prev_obs IS last step's cur_obs: frame ring
            tp = mark("VO boundary call", tp)
            goals[:] = list(geometry.compute_goal_pos_batch(np.stack(goals), deltas)["cartesian"])
            prev_a = torch.as_tensor(acts, device=dev).view(E, 1)
            prev_obs = cur_obs
            mark("goal update", tp)

        for s in range(12):                 # (every action model, the policy and their small-batch kernels have run before the clock starts)
            step(s)
        torch.cuda.synchronize()
        for k in ph:
            ph[k] = 0.0
        t0 = time.perf_counter()
        for s in range(a.steps):
            step(s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        if phases:
            print(f"[navloop] {E} envs, ms per loop step by phase (host-synchronised):",
                  {k: round(1e3 * v / a.steps, 3) for k, v in ph.items()}, file=sys.stderr)
        res["results"].append({"envs": E, "ms_per_loop_step": dt * 1e3, "env_steps_per_s": E / dt,
                               "gpu_side_minutes_for_994_episodes": 994 * a.episode_steps / (E / dt) / 60.0})
    res["note"] = (f"extrapolation assumes {a.episode_steps:.0f} steps per episode on one GPU; the reference's 4.5 h includes "
                   "the simulator, which is not emulated here")
    return res


if __name__ == "__main__":
    main()
