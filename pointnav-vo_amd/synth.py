"""Deterministic synthetic weights and inputs (numpy only, no torch RNG, stable across library versions).

Used by bench.py, the parity tests and tests/golden/gen_golden.py so that the build container (where the
reference can be imported) and the GPU box (where it cannot) regenerate bit-identical tensors from a seed;
the golden fixtures then only need to store the reference's OUTPUTS.

Shapes/ranges follow SURVEY.md §8(d): rgb uint8 U{0..255}; depth U[0.05,0.95] rounded to fp16 and widened
(the dataset stores fp16, /root/reference/pointnav_vo/vo/dataset/generate_datasets.py:272,290);
discretized_depth = one-hot of the depth bin; running mean ~U[0,0.5], var ~U[0,0.2].
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def bits(seed: int, name: str, n: int) -> np.ndarray:
    """n uint64 pseudo-random words, a pure function of (seed, name, index)."""
    key = np.uint64((_fnv1a(name) ^ (seed * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        return _splitmix64(_splitmix64(key) ^ idx)


def uniform(seed: int, name: str, shape, lo=0.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape))
    u = (bits(seed, name, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).reshape(shape)


def make_state_dict(spec, seed=0):
    """spec: list of (name, shape) in reference state_dict naming (model_spec.state_dict_spec).
    Returns dict name -> float32 ndarray.  Scales mimic a trained net: conv/linear ~ kaiming-uniform-like
    (std = sqrt(2/fan_in)), GroupNorm gamma in +-[0.5,1.5] (a few negative), beta in [-0.3,0.3]."""
    sd = {}
    for name, shape in spec:
        if name.endswith("_count"):
            sd[name] = np.array(1000.0, dtype=np.float32)
        elif name.endswith("_mean"):
            sd[name] = uniform(seed, name, shape, 0.0, 0.5).astype(np.float32)
        elif name.endswith("_var"):
            sd[name] = uniform(seed, name, shape, 0.0, 0.2).astype(np.float32)
        elif name == "action_embedding.weight":
            sd[name] = uniform(seed, name, shape, -1.0, 1.0).astype(np.float32)
        elif len(shape) == 4 or (len(shape) == 2):
            fan_in = int(np.prod(shape[1:]))
            a = np.sqrt(3.0) * np.sqrt(2.0 / fan_in)
            sd[name] = uniform(seed, name, shape, -a, a).astype(np.float32)
        elif name.endswith(".weight"):  # GroupNorm gamma
            gmm = uniform(seed, name, shape, 0.5, 1.5)
            sign = np.where(uniform(seed, name + "#sign", shape) < 0.1, -1.0, 1.0)
            sd[name] = (gmm * sign).astype(np.float32)
        elif name.startswith("visual_encoder"):  # GroupNorm beta
            sd[name] = uniform(seed, name, shape, -0.3, 0.3).astype(np.float32)
        else:  # linear bias
            sd[name] = uniform(seed, name, shape, -0.1, 0.1).astype(np.float32)
    return sd


def onehot_depth(depth, bins):
    """One-hot of floor(depth*bins) clipped to bins-1, float32 {0,1} (same result as the reference's
    _discretize_depth_func for depth in [0,1]; the exact-edge semantics are tested separately)."""
    edges = (np.arange(bins, dtype=np.float64) / bins).astype(np.float32)
    idx = np.clip(np.searchsorted(edges, depth.astype(np.float32), side="right") - 1, 0, bins - 1)
    out = np.zeros(depth.shape + (bins,), dtype=np.float32)
    np.put_along_axis(out, idx[..., None], 1.0, axis=-1)
    return out


def make_obs_pairs(B, H, W, *, observation_space, dd_bins=10, seed=0, tdv_sparsity=0.7, start=0, depth_fp16=True):
    """Synthetic observation pairs in the reference's model-input format (NHWC float32):
    rgb [B,H,W,6] in 0..255, depth [B,H,W,2] in [0,1], discretized_depth [B,H,W,2*bins], top_down_view
    [B,H,W,2] (sparse histogram-like values in [0,1]).  `start` offsets the sample index so shards of one
    global batch can be generated independently (sample i depends only on (seed, start+i)).  `depth_fp16` (default): depth
    rounded through float16 as the dataset stores it (generate_datasets.py:272,290); False: dense float32 depth, what the simulator
    hands the navigation loop (base_trainer_with_vo.py:177-190) — every stem operand then has a non-zero low piece."""
    obs = {}
    rgb = np.empty((B, H, W, 6), dtype=np.float32)
    depth = np.empty((B, H, W, 2), dtype=np.float32)
    tdv = np.empty((B, H, W, 2), dtype=np.float32)
    for i in range(B):
        tag = f"#{start + i}"
        rgb[i] = (bits(seed, "rgb" + tag, H * W * 6) >> np.uint64(56)).astype(np.float32).reshape(H, W, 6)
        d = uniform(seed, "depth" + tag, (H, W, 2), 0.05, 0.95)
        depth[i] = d.astype(np.float16).astype(np.float32) if depth_fp16 else d.astype(np.float32)
        t = uniform(seed, "tdv" + tag, (H, W, 2))
        m = uniform(seed, "tdvmask" + tag, (H, W, 2))
        tdv[i] = np.where(m < tdv_sparsity, 0.0, t).astype(np.float32)
    if "rgb" in observation_space:
        obs["rgb"] = rgb
    if "depth" in observation_space:
        obs["depth"] = depth
    if "discretized_depth" in observation_space:
        obs["discretized_depth"] = np.concatenate(
            [onehot_depth(depth[..., 0], dd_bins), onehot_depth(depth[..., 1], dd_bins)], axis=-1
        )
    if "top_down_view" in observation_space:
        obs["top_down_view"] = tdv
    return obs


def make_raw_obs(H, W, seed=0, index=0, zero_border=0, depth_fp16=True):
    """One simulator-style observation dict: rgb uint8 [H,W,3], depth float32 [H,W,1] in [0,1]
    (what _compute_local_delta_states_from_vo receives,
    /root/reference/pointnav_vo/rl/common/base_trainer_with_vo.py:172-193)."""
    tag = f"#{index}"
    rgb = (bits(seed, "raw_rgb" + tag, H * W * 3) >> np.uint64(56)).astype(np.uint8).reshape(H, W, 3)
    d = uniform(seed, "raw_depth" + tag, (H, W, 1), 0.0, 1.0)
    d = d.astype(np.float16).astype(np.float32) if depth_fp16 else d.astype(np.float32)
    if zero_border:
        d[:zero_border] = 0
        d[-zero_border:] = 0
        d[:, :zero_border] = 0
        d[:, -2 * zero_border:] = 0
    return {"rgb": rgb, "depth": d}


def make_policy_inputs(H, W, B, steps, seed):
    """Per-step inputs of the navigation policy for `steps` consecutive act() calls of B environments:
    list of (depth [B,H,W,1] float32 in [0,1], goal [B,2] = (rho, phi), prev_actions [B] int64, masks [B] float32).
    Step 0 starts every episode (mask 0); at step 2 environment 1 % B resets."""
    out = []
    for t in range(steps):
        depth = np.stack([make_raw_obs(H, W, seed=seed, index=100 * t + b)["depth"] for b in range(B)])
        goal = np.stack([uniform(seed, f"goal{t}", (B,), 0.2, 6.0), uniform(seed, f"phi{t}", (B,), -3.0, 3.0)],
                        axis=-1).astype(np.float32)
        prev = (bits(seed, f"act{t}", B) % np.uint64(4)).astype(np.int64)
        mask = np.ones(B, dtype=np.float32)
        if t == 0:
            mask[:] = 0.0
        if t == 2:
            mask[1 % B] = 0.0
        out.append((depth, goal, prev, mask))
    return out


def make_joint_batch(P, H, W, observation_space, dd_bins=10, seed=0):
    """The batch layout "inverse_joint_train" produces (vo/dataset/regression_geo_invariance_iter_dataset.py:342-386): P turn
    samples, each followed by its channel-swapped (cur, prev) entry for the opposite action.  Returns (obs dict [2P,...],
    actions [2P] in {2 (left), 3 (right)}, data_types [2P] alternating 0 / 1)."""
    base = make_obs_pairs(P, H, W, observation_space=observation_space, dd_bins=dd_bins, seed=seed)
    acts = (bits(seed, "joint_acts", P) % np.uint64(2)).astype(np.int64) + 2

    def swap(a):
        h = a.shape[-1] // 2
        return np.concatenate([a[..., h:], a[..., :h]], axis=-1)

    obs = {k: np.stack([x for i in range(P) for x in (v[i], swap(v[i]))]) for k, v in base.items()}
    actions = np.stack([x for i in range(P) for x in (acts[i], 5 - acts[i])]).astype(np.int64)
    data_types = np.tile(np.array([0, 1], dtype=np.int64), P)
    return obs, actions, data_types


def make_dataset_chunk(N, H, W, seed=0, bins=10):
    """The arrays of one HDF5 chunk as generate_datasets.py stores them (:258-305): uint8 rgb vectors, float16 depth vectors,
    uint8 actions, float16 poses.  Depth frames get a zero border of varying width (exercises the top-down view's crop),
    exact float16 bin edges and their float16 neighbours (exercise the one-hot's float16 comparison), one all-zero frame."""
    ch = {"actions": ((bits(seed, "ds_act", N) % np.uint64(3)) + np.uint64(1)).astype(np.uint8)}
    edge = np.array([np.float16(i / bins) for i in range(bins + 1)], dtype=np.float16)
    near = np.concatenate([edge, np.nextafter(edge[1:], np.float16(0)), np.nextafter(edge[:-1], np.float16(1))])
    for k in ("prev", "cur"):
        ch[f"{k}_rgbs"] = (bits(seed, f"ds_rgb_{k}", N * H * W * 3) & np.uint64(255)).astype(np.uint8).reshape(N, H * W * 3)
        d = uniform(seed, f"ds_depth_{k}", (N, H, W), 0.0, 1.0).astype(np.float16)
        pick = (bits(seed, f"ds_edge_{k}", N * H * W) % np.uint64(16)).reshape(N, H, W)
        which = (bits(seed, f"ds_which_{k}", N * H * W) % np.uint64(near.size)).astype(np.int64).reshape(N, H, W)
        d = np.where(pick == 0, near[which], d)
        for n in range(N):
            b = int(bits(seed, f"ds_border_{k}", N)[n] % np.uint64(4))
            if b:
                d[n, :b] = 0
                d[n, H - 2 * b:] = 0
                d[n, :, :2 * b] = 0
                d[n, :, W - b:] = 0
        if k == "cur" and N > 2:
            d[N - 1] = 0
        ch[f"{k}_depths"] = d.reshape(N, H * W)
        yaw = uniform(seed, f"ds_yaw_{k}", (N,), -np.pi, np.pi)
        ch[f"{k}_global_rotations"] = np.stack([0 * yaw, np.sin(yaw / 2), 0 * yaw, np.cos(yaw / 2)], 1).astype(np.float16)
        ch[f"{k}_global_positions"] = uniform(seed, f"ds_pos_{k}", (N, 3), -3.0, 3.0).astype(np.float16)
    dy = uniform(seed, "ds_dyaw", (N,), -0.3, 0.3)
    ch["delta_rotations"] = np.stack([0 * dy, np.sin(dy / 2), 0 * dy, np.cos(dy / 2)], 1).astype(np.float16)
    ch["delta_positions"] = uniform(seed, "ds_dpos", (N, 3), -0.3, 0.3).astype(np.float16)
    return ch
