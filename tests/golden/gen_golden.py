#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the IMPORTED reference (build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

Runs the reference's *unmodified* modules from /root/reference (SURVEY.md Appendix A recipe: only absent
third-party packages are stubbed) on inputs/weights produced by pointnav_vo_amd.synth, and stores the reference's
OUTPUTS (plus the handful of inputs that cannot be regenerated from a seed).  The fixtures are data: no reference
source text is stored.  /root/reference does not exist on the GPU box; nothing at test time imports this script.

Stubbed step (documented in oracle/pnvo_oracle_pre.c): cv2.GaussianBlur — OpenCV is not installed in this image,
so the stub below implements OpenCV's published ksize=3/sigma<=0 kernel {1/4,1/2,1/4}; the blur is therefore
"parity unpinned", everything around it is pinned.
"""
import ast
import importlib
import logging
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from pointnav_vo_amd import model_spec as ms  # noqa: E402
from pointnav_vo_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- reference import
def blur_stub(src, ksize, sigmaX=0, sigmaY=0, borderType=0):
    """Stand-in for cv2.GaussianBlur(src, (3,3), 0, 0, BORDER_ISOLATED) on a float32 2-D array."""
    assert tuple(ksize) == (3, 3) and sigmaX == 0 and sigmaY == 0
    s = np.ascontiguousarray(src, dtype=np.float32)
    p = np.pad(s, ((0, 0), (1, 1)))
    t = (s * np.float32(0.5) + (p[:, :-2] + p[:, 2:]) * np.float32(0.25)).astype(np.float32)
    p = np.pad(t, ((1, 1), (0, 0)))
    return (t * np.float32(0.5) + (p[:-2, :] + p[2:, :]) * np.float32(0.25)).astype(np.float32)


def import_reference():
    def ns(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [REF + rel]
        sys.modules[name] = m

    for n, p in [("pointnav_vo", "/pointnav_vo"), ("pointnav_vo.utils", "/pointnav_vo/utils"),
                 ("pointnav_vo.vo", "/pointnav_vo/vo"), ("pointnav_vo.vo.models", "/pointnav_vo/vo/models")]:
        ns(n, p)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class Registry:
        mapping = {}

        @classmethod
        def _register_impl(cls, _t, to_register, name, assert_type=None):
            def w(c):
                cls.mapping.setdefault(_t, {})[name or c.__name__] = c
                return c
            return w if to_register is None else w(to_register)

        @classmethod
        def _get_impl(cls, _t, name):
            return cls.mapping.get(_t, {}).get(name)

    stub("habitat", logger=logging.getLogger("habitat"), Config=dict)
    stub("habitat.core")
    stub("habitat.core.registry", Registry=Registry)
    stub("habitat.core.simulator", AgentState=object)
    stub("habitat.tasks")
    stub("habitat.tasks.utils", cartesian_to_polar=None)
    stub("habitat.utils")
    stub("habitat.utils.geometry_utils", quaternion_to_list=None, quaternion_rotate_vector=None)
    stub("habitat.utils.visualizations")
    stub("habitat.utils.visualizations.utils", images_to_video=None)
    stub("gym")
    stub("gym.spaces", Box=object)
    stub("torch.utils.tensorboard", SummaryWriter=object)
    stub("cv2", GaussianBlur=blur_stub, BORDER_ISOLATED=16, setNumThreads=lambda n: None)
    stub("quaternion")
    np.quaternion = object
    importlib.import_module("pointnav_vo.vo.models.vo_cnn")
    importlib.import_module("pointnav_vo.vo.models.vo_cnn_act_embed")
    geo = importlib.import_module("pointnav_vo.utils.geometry_utils")
    from pointnav_vo.utils.baseline_registry import baseline_registry
    return baseline_registry, geo


def extract_methods(path, class_name, names):
    """exec selected methods of a class whose module cannot be imported (habitat trainers)."""
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name in names:
                    out[f.name] = ast.get_source_segment(src, f)
    return out


# ----------------------------------------------------------------------------- helpers
MODEL_KW = dict(hidden_size=512, backbone="resnet18", normalize_visual_inputs=True, output_dim=3, dropout_p=0.2)


def build_ref_model(registry, name, obs_space, size, dd_bins, seed, extra=None):
    kw = dict(MODEL_KW, observation_space=obs_space, observation_size=size, discretized_depth_channels=dd_bins)
    kw.update(extra or {})
    model = registry.get_vo_model(name)(**kw).eval()
    act_embed = "act_embed" in name
    baseplanes = 64 if "wider" in name else 32
    cfg = ms.config_from_kwargs(**dict(kw, resnet_baseplanes=baseplanes, act_embed=act_embed))
    spec = ms.state_dict_spec(cfg)
    ref_sd = model.state_dict()
    # pins SURVEY §8(b): our naming/shape derivation == the reference's state_dict
    assert [n for n, _ in spec] == list(ref_sd.keys()), (name, set(ref_sd) ^ {n for n, _ in spec})
    for n, s in spec:
        assert tuple(ref_sd[n].shape) == tuple(s), (n, ref_sd[n].shape, s)
    sd = synth.make_state_dict(spec, seed=seed)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return model, cfg, sd


def run_ref(model, obs, dtype, actions=None, taps=None):
    m = model.double() if dtype == torch.float64 else model.float()
    t = {k: torch.from_numpy(v).to(dtype) for k, v in obs.items()}
    hooks = []
    if taps is not None:
        enc = m.visual_encoder

        def grab(key):
            def h(_m, _i, o):
                taps[key] = o.detach().permute(0, 2, 3, 1).contiguous().numpy() if o.dim() == 4 else o.detach().numpy()
            return h
        hooks.append(enc.running_mean_and_var.register_forward_hook(grab("input")))
        hooks.append(enc.backbone.conv1[0].register_forward_hook(grab("stem_conv")))
        hooks.append(enc.backbone.maxpool.register_forward_hook(grab("maxpool")))
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(enc.backbone, f"layer{li}")):
                hooks.append(blk.register_forward_hook(grab(f"layer{li}.{bi}")))
        hooks.append(enc.compression.register_forward_hook(grab("compression")))
        fc = m.hidden_generator if hasattr(m, "hidden_generator") else m.visual_fc
        hooks.append(fc.register_forward_hook(grab("hidden")))
    with torch.no_grad():
        out = m(t, torch.from_numpy(actions)) if actions is not None else m(t)
    for h in hooks:
        h.remove()
    return out.numpy()


def sample_idx(name, n, k=24):
    return (synth.bits(1234, "tap:" + name, k) % np.uint64(n)).astype(np.int64)


def model_fixture(registry, fname, name, obs_space, size, B, dd_bins, seed, full_taps, extra=None, depth_fp16=True):
    W, H = size
    model, cfg, sd = build_ref_model(registry, name, obs_space, size, dd_bins, seed, extra)
    obs = synth.make_obs_pairs(B, H, W, observation_space=obs_space, dd_bins=max(dd_bins, 1), seed=seed, depth_fp16=depth_fp16)
    actions = None
    if cfg.act_embed:
        actions = (synth.bits(seed, "actions", B) % np.uint64(4)).astype(np.int64)
    taps64 = {}
    out64 = run_ref(model, obs, torch.float64, actions, taps64)
    out32 = run_ref(model, obs, torch.float32, actions)
    rec = dict(model=name, obs_space=",".join(obs_space), width=W, height=H, batch=B, dd_bins=dd_bins, seed=seed,
               baseplanes=cfg.baseplanes, act_embed=int(cfg.act_embed), out64=out64, out32=out32,
               backbone="resnet%d" % cfg.backbone_depth)
    if not depth_fp16:
        rec["depth_fp16"] = 0
    if actions is not None:
        rec["actions"] = actions
    for k, v in taps64.items():
        flat = v.reshape(-1)
        idx = sample_idx(k, flat.size)
        rec[f"tapidx/{k}"] = idx
        rec[f"tapval/{k}"] = flat[idx]
        rec[f"tapstat/{k}"] = np.array([flat.mean(), np.sqrt((flat ** 2).mean()), np.abs(flat).max()])
        if full_taps:
            rec[f"tap/{k}"] = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, fname), **rec)
    print(f"{fname}: out64[0]={out64[0]}  |out32-out64|max={np.abs(out32 - out64).max():.3e}")


# ----------------------------------------------------------------------------- pre-processing fixtures
def preproc_fixture(geo):
    meths = extract_methods(REF + "/pointnav_vo/rl/common/base_trainer_with_vo.py", "BaseRLTrainerWithVO",
                            ["_discretize_depth_func"])
    nsd = {"torch": torch, "np": np}
    exec(meths["_discretize_depth_func"], nsd)
    bins = 10
    fake = types.SimpleNamespace()
    fake.config = types.SimpleNamespace(VO=types.SimpleNamespace(REGRESS_MODEL=types.SimpleNamespace(
        discretized_depth_channels=bins, discretize_depth="hard")))
    fake._discretized_depth_end_vals = [i * 1.0 / bins for i in np.arange(bins)] + [1.0]  # :105-115
    # edge-value probe: every fp32 edge, its float neighbours, 0 and 1, plus random fp16-rounded depths
    edges32 = np.array([np.float32(i / bins) for i in range(bins + 1)], dtype=np.float32)
    probe = np.concatenate([edges32, np.nextafter(edges32, np.float32(2)), np.nextafter(edges32, np.float32(-1)),
                            (np.arange(bins + 1) / bins).astype(np.float64).astype(np.float32)])
    probe = probe[(probe >= 0) & (probe <= 1)]
    rnd = synth.uniform(7, "dd_probe", (4096 - probe.size,)).astype(np.float16).astype(np.float32)
    depth = np.concatenate([probe, rnd]).reshape(64, 64)
    dd = nsd["_discretize_depth_func"](fake, torch.from_numpy(depth)).numpy()
    assert dd.sum() == depth.size
    rec = {"dd_depth": depth, "dd_bins": bins, "dd_index": dd.argmax(-1).astype(np.uint8)}

    # top-down view: constructor args as at base_trainer_with_vo.py:119-129 with the challenge YAML values
    cases = {}

    def frame(H, W, tag, lo=0.0, hi=1.0, border=0, fp16=True):
        d = synth.uniform(11, "tdv_depth_" + tag, (H, W, 1), lo, hi)
        d = d.astype(np.float16).astype(np.float32) if fp16 else d.astype(np.float32)
        if border:
            d[:border] = 0
            d[-border:] = 0
            d[:, : border + 3] = 0
            d[:, -2 * border:] = 0
        return d

    cases["full_uniform"] = frame(192, 341, "a")
    cases["full_border"] = frame(192, 341, "b", border=7)
    cases["full_near"] = frame(192, 341, "c", 0.0, 0.15)
    cases["full_fp32"] = frame(192, 341, "d", fp16=False)
    cases["full_zero"] = np.zeros((192, 341, 1), np.float32)
    one = np.zeros((192, 341, 1), np.float32)
    one[100, 200, 0] = 0.37
    cases["full_one_pixel"] = one
    top = np.zeros((192, 341, 1), np.float32)
    top[:20] = frame(20, 341, "e")
    cases["full_top_band"] = top          # crop is far from the image centre
    cases["small_odd"] = frame(37, 45, "f")
    cases["small_border"] = frame(48, 64, "g", border=3)
    consts = {}
    for key, d in cases.items():
        H, W = d.shape[:2]
        gen = geo.NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W,
                                                         hfov_rad=70)
        out = gen.gen_top_down_view(torch.from_numpy(d)).numpy()
        assert out.shape == (H, W, 1)
        nz = np.flatnonzero(out)
        rec[f"tdv_in/{key}"] = d.astype(np.float16) if np.array_equal(d.astype(np.float16).astype(np.float32), d) else d
        rec[f"tdv_nz/{key}"] = nz.astype(np.int32)
        rec[f"tdv_val/{key}"] = out.reshape(-1)[nz]
        if (H, W) not in consts:
            kinv = torch.inverse(gen._K)
            min_x, max_x = gen._get_x_range(gen._max_depth, device="cpu")
            x_den = ((max_x - min_x) * (1 + gen._epsilon)).numpy()[0]
            consts[(H, W)] = np.array([kinv[0, 0].item(), kinv[0, 2].item(), min_x.numpy()[0], x_den,
                                       np.float32(gen._max_depth - gen._min_depth),
                                       np.float32((gen._max_depth - gen._min_depth) * (1 + gen._epsilon)),
                                       np.float32(gen._min_depth), kinv[0, 1].item()], dtype=np.float32)
            rec[f"tdv_consts/{H}x{W}"] = consts[(H, W)]
            rec[f"tdv_f/{H}x{W}"] = np.float32(gen._K[0, 0].item())
        print(f"tdv {key}: nnz={nz.size} max={out.max():.3f}")
    np.savez_compressed(os.path.join(HERE, "preproc.npz"), **rec)


# ----------------------------------------------------------------------------- boundary fixture (a1)
def boundary_fixture(registry, geo, fname="boundary.npz", depth_fp16=True, steps=None):
    """_compute_local_delta_states_from_vo (base_trainer_with_vo.py:169-314) end to end, sep_act, det mode."""
    meths = extract_methods(REF + "/pointnav_vo/rl/common/base_trainer_with_vo.py", "BaseRLTrainerWithVO",
                            ["_discretize_depth_func", "_compute_local_delta_states_from_vo"])
    cv = importlib.import_module("pointnav_vo.vo.common.common_vars")
    nsd = {"torch": torch, "np": np, "NormalizedDepth2TopDownViewHabitatTorch": geo.NormalizedDepth2TopDownViewHabitatTorch,
           "NormalizedDepth2TopDownViewHabitat": geo.NormalizedDepth2TopDownViewHabitat,
           "ACT_IDX2NAME": cv.ACT_IDX2NAME}
    for m in meths.values():
        exec(m, nsd)
    W, H, bins = 341, 192, 10
    name = "vo_cnn_rgb_d_dd_top_down"
    obs_space = ["rgb", "depth", "discretized_depth", "top_down_view"]
    fake = types.SimpleNamespace()
    rm = types.SimpleNamespace(name=name, discretized_depth_channels=bins, discretize_depth="hard",
                               regress_type="sep_act", mode="det", rnd_mode_n=10)
    fake.config = types.SimpleNamespace(VO=types.SimpleNamespace(VO_TYPE="REGRESS", REGRESS_MODEL=rm))
    fake.device = torch.device("cpu")
    fake._vo_obs_transformer = None
    fake._discretized_depth_end_vals = [i * 1.0 / bins for i in np.arange(bins)] + [1.0]
    fake._top_down_view_generator = geo.NormalizedDepth2TopDownViewHabitatTorch(
        min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    fake._discretize_depth_func = types.MethodType(nsd["_discretize_depth_func"], fake)
    fake.vo_model = {}
    seeds = {"forward": 21, "left": 22, "right": 23}
    for k, s in seeds.items():
        fake.vo_model[k], _, _ = build_ref_model(registry, name, obs_space, (W, H), bins, s)
    rec = dict(width=W, height=H, bins=bins, seed_forward=21, seed_left=22, seed_right=23, obs_seed=5)
    outs = []
    steps = steps or [(0, 1, 1, 0), (1, 2, 2, 0), (2, 3, 3, 4), (3, 4, 1, 4)]   # (prev idx, cur idx, act, zero_border)
    rec["depth_fp16"] = int(depth_fp16)
    for pi, ci, act, zb in steps:
        prev = synth.make_raw_obs(H, W, seed=5, index=pi, zero_border=zb, depth_fp16=depth_fp16)
        cur = synth.make_raw_obs(H, W, seed=5, index=ci, zero_border=zb, depth_fp16=depth_fp16)
        d, std, _ = nsd["_compute_local_delta_states_from_vo"](fake, prev, cur, act)
        outs.append(np.array(d, dtype=np.float32))
        assert std == [0, 0, 0]
    rec["steps"] = np.array(steps, dtype=np.int32)
    rec["deltas"] = np.stack(outs)
    np.savez_compressed(os.path.join(HERE, fname), **rec)
    print("boundary deltas:\n", rec["deltas"])


# ----------------------------------------------------------------------------- training-step fixtures (a14)
def train_fixture(registry, fname, name, obs_space, size, B, dd_bins, seed, dtype, actions=None, extra=None):
    """model.train() forward (RunningMeanAndVar update), the reference's own _compute_loss for dx/dz/dyaw
    (vo/engine/vo_cnn_engine.py:135-198), backward, torch.optim.Adam(lr=2.5e-4, eps=1e-8) as the engine sets it up
    (vo_cnn_regression_geo_invariance_engine.py:122-133; configs/vo/vo_pointnav.yaml:35-45).  dropout_p = 0 so the step
    is deterministic (torch's dropout RNG cannot be reproduced elsewhere)."""
    W, H = size
    model, cfg, sd = build_ref_model(registry, name, obs_space, size, dd_bins, seed, extra=dict(extra or {}, dropout_p=0.0))
    model = model.double() if dtype == torch.float64 else model.float()
    model.train()
    obs = synth.make_obs_pairs(B, H, W, observation_space=obs_space, dd_bins=max(dd_bins, 1), seed=seed)
    target = synth.uniform(seed, "train_target", (B, 3), -0.3, 0.3).astype(np.float32)
    meths = extract_methods(REF + "/pointnav_vo/vo/engine/vo_cnn_engine.py", "VOCNNBaseEngine", ["_compute_loss"])
    cv = importlib.import_module("pointnav_vo.vo.common.common_vars")
    nsd = {"torch": torch, "np": np, "DEFAULT_LOSS_WEIGHTS": cv.DEFAULT_LOSS_WEIGHTS, "EPSILON": cv.EPSILON}
    exec(meths["_compute_loss"], nsd)
    opt = torch.optim.Adam(model.parameters(), lr=2.5e-4, eps=1e-8, weight_decay=0)
    rec = dict(model=name, obs_space=",".join(obs_space), width=W, height=H, batch=B, dd_bins=dd_bins, seed=seed,
               baseplanes=cfg.baseplanes, act_embed=int(actions is not None), target=target, lr=2.5e-4, eps=1e-8,
               backbone=(extra or {}).get("backbone", "resnet18"))
    if actions is not None:
        rec["actions"] = np.asarray(actions, dtype=np.int64)
    tgt = torch.from_numpy(target).to(dtype)
    tgts = (tgt[:, 0:1], tgt[:, 1:2], tgt[:, 2:3])
    for step in (1, 2):
        opt.zero_grad()
        tobs = {k: torch.from_numpy(v).to(dtype) for k, v in obs.items()}
        out = model(tobs) if actions is None else model(tobs, torch.as_tensor(actions, dtype=torch.long))
        loss = 0
        for d, dt in enumerate(["dx", "dz", "dyaw"]):
            loss = loss + nsd["_compute_loss"](None, out[:, d:d + 1], tgts, d_type=dt)[0]
        loss.backward()
        rec[f"loss{step}"] = loss.item()
        rec[f"out{step}"] = out.detach().numpy().astype(np.float64)
        for k, prm in model.named_parameters():
            gflat = prm.grad.detach().reshape(-1).double().numpy()
            idx = sample_idx("grad:" + k, gflat.size, 16)
            rec[f"gidx/{k}"] = idx
            rec[f"g{step}val/{k}"] = gflat[idx]
            rec[f"g{step}norm/{k}"] = np.linalg.norm(gflat)
        opt.step()
        for k, prm in model.named_parameters():
            pflat = prm.detach().reshape(-1).double().numpy()
            rec[f"p{step}val/{k}"] = pflat[rec[f"gidx/{k}"]]
        for k, b in model.named_buffers():
            rec[f"buf{step}/{k}"] = np.array(b.detach().double().numpy(), copy=True)   # copy: _count is updated in place
    np.savez_compressed(os.path.join(HERE, fname), **rec)
    print(f"{fname}: loss1={rec['loss1']:.6f} loss2={rec['loss2']:.6f}")


def import_geo_engine():
    """The reference's VOCNNRegressionGeometricInvarianceEngine class, unmodified; only absent third-party packages and the
    modules that need them (HDF5 dataset, tensorboard writer, yacs config helpers) are stubbed."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    for n, pth in [("pointnav_vo.vo.engine", "/pointnav_vo/vo/engine"), ("pointnav_vo.vo.common", "/pointnav_vo/vo/common"),
                   ("pointnav_vo.vo.dataset", "/pointnav_vo/vo/dataset")]:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = [REF + pth]
            sys.modules[n] = m
    stub("h5py")
    stub("yacs")
    stub("yacs.config", CfgNode=dict)
    stub("pointnav_vo.utils.tensorboard_utils", TensorboardWriter=object)
    stub("pointnav_vo.utils.config_utils", update_config_log=None)
    stub("pointnav_vo.vo.dataset.regression_geo_invariance_iter_dataset", StatePairRegressionDataset=object,
         normal_collate_func=None, fast_collate_func=None)
    mod = importlib.import_module("pointnav_vo.vo.engine.vo_cnn_regression_geo_invariance_engine")
    return mod.VOCNNRegressionGeometricInvarianceEngine


class _NS(dict):
    """attribute + item access, `in` on keys (what the engine does with its yacs config)."""
    __getattr__ = dict.__getitem__


def joint_train_fixture(registry, fname, size, P, seed, fixed_weights):
    """One training iteration of act_left_right_inv_joint (BASELINE config 3's model pair) through the reference engine's
    own _process_one_batch with invariance_types = ["inverse_joint_train"] (…geo_invariance_engine.py:451-807): a
    TURN_LEFT and a TURN_RIGHT model, P turn samples, each followed by its swapped prev_rel_to_cur entry for the opposite
    action (dataset :342-386), regression losses per data type + the inverse-consistency loss, backward, Adam per action."""
    Engine = import_geo_engine()
    cv = importlib.import_module("pointnav_vo.vo.common.common_vars")
    W, H = size
    space = ["rgb", "depth", "discretized_depth", "top_down_view"]
    models, sds = {}, {}
    for act, sd_seed in ((cv.TURN_LEFT, seed), (cv.TURN_RIGHT, seed + 1)):
        m, cfg, sd = build_ref_model(registry, "vo_cnn_rgb_d_dd_top_down", space, size, 10, sd_seed, extra=dict(dropout_p=0.0))
        models[act], sds[act] = m.double().train(), sd
    base = synth.make_obs_pairs(P, H, W, observation_space=space, dd_bins=10, seed=seed)
    acts = (synth.bits(seed, "joint_acts", P) % np.uint64(2)).astype(np.int64) + cv.TURN_LEFT

    def swap(a):                                     # (prev, cur) -> (cur, prev): halves of the channel axis
        h = a.shape[-1] // 2
        return np.concatenate([a[..., h:], a[..., :h]], axis=-1)

    obs = {k: np.stack([x for i in range(P) for x in (v[i], swap(v[i]))]) for k, v in base.items()}
    actions = np.stack([x for i in range(P) for x in (acts[i], 5 - acts[i])]).astype(np.int64)
    dtypes = np.tile(np.array([cv.CUR_REL_TO_PREV, cv.PREV_REL_TO_CUR], dtype=np.int64), P)
    target = synth.uniform(seed, "joint_target", (2 * P, 3), -0.3, 0.3).astype(np.float32)
    dzmask = np.ones((2 * P, 1), np.float32)
    mult = {"dx": 1.0, "dz": 1.0, "dyaw": 1.0} if fixed_weights else {"dx": 2.0, "dz": 1.5, "dyaw": 3.0}

    eng = object.__new__(Engine)
    eng._config = _NS(VO=_NS(GEOMETRY=_NS(loss_inv_weight=0.7, invariance_types=["inverse_joint_train"]),
                             TRAIN=_NS(loss_weight_fixed=fixed_weights, loss_weight_multiplier=mult),
                             MODEL=_NS(name="vo_cnn_rgb_d_dd_top_down")))
    eng._verbose = False
    eng.device = torch.device("cpu")
    eng._pin_memory_flag = False
    eng._data_collate_mode = "normal"
    eng._observation_space = space
    eng._act_type = [cv.TURN_LEFT, cv.TURN_RIGHT]
    eng._act_list = [cv.TURN_LEFT, cv.TURN_RIGHT]
    eng.vo_model = models
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    tg = t(target)
    batch = (t(dtypes, torch.int64).unsqueeze(1), t(obs["rgb"]), t(obs["depth"]), t(obs["discretized_depth"]),
             t(obs["top_down_view"]), t(actions, torch.int64).unsqueeze(1), tg[:, 0:1], torch.zeros_like(tg[:, 0:1]),
             tg[:, 1:2], tg[:, 2:3], t(dzmask), torch.zeros(2 * P, 1), torch.zeros(2 * P, 1))
    opts = {a: torch.optim.Adam(m.parameters(), lr=2.5e-4, eps=1e-8, weight_decay=0) for a, m in models.items()}
    for o in opts.values():
        o.zero_grad()
    captured = {}
    for a, m in models.items():
        m.register_forward_hook(lambda _m, _i, o, a=a: captured.__setitem__(a, o.detach().clone()))
    loss, bs, _, _, logs = eng._process_one_batch(batch, ["inverse_joint_train"], {}, {}, {}, train_flag=True)
    loss.backward()
    rec = dict(width=W, height=H, pairs=P, seed=seed, fixed_weights=int(fixed_weights), loss_inv_weight=0.7,
               mult=np.array([mult["dx"], mult["dz"], mult["dyaw"]]), actions=actions, data_types=dtypes, target=target,
               loss=loss.item(), abs_diff_geo_inverse_rot=float(logs[3]), abs_diff_geo_inverse_pos=logs[4].detach().numpy())
    for a in models:
        rec[f"pred{a}"] = captured[a].numpy()          # rows in the order of nonzero(actions == a)
    for a, m in models.items():
        for k, prm in m.named_parameters():
            gflat = prm.grad.detach().reshape(-1).double().numpy()
            idx = sample_idx("jgrad:" + k, gflat.size, 16)
            rec[f"gidx/{k}"] = idx
            rec[f"gval{a}/{k}"] = gflat[idx]
            rec[f"gnorm{a}/{k}"] = np.linalg.norm(gflat)
    for a, o in opts.items():
        o.step()
    for a, m in models.items():
        for k, prm in m.named_parameters():
            rec[f"pval{a}/{k}"] = prm.detach().reshape(-1).double().numpy()[rec[f"gidx/{k}"]]
        for k, b in m.named_buffers():
            rec[f"buf{a}/{k}"] = np.array(b.detach().double().numpy(), copy=True)
    np.savez_compressed(os.path.join(HERE, fname), **rec)
    print(f"{fname}: loss={rec['loss']:.6f} inv rot {rec['abs_diff_geo_inverse_rot']:.4f}")


def geo_loss_fixture():
    meths = extract_methods(REF + "/pointnav_vo/vo/engine/vo_cnn_regression_geo_invariance_engine.py",
                            "VOCNNRegressionGeometricInvarianceEngine", ["_compute_geo_invariance_inverse_loss"])
    cv = importlib.import_module("pointnav_vo.vo.common.common_vars")
    nsd = {"torch": torch, "np": np, "CUR_REL_TO_PREV": cv.CUR_REL_TO_PREV, "PREV_REL_TO_CUR": cv.PREV_REL_TO_CUR,
           "MOVE_FORWARD": cv.MOVE_FORWARD}
    exec(meths["_compute_geo_invariance_inverse_loss"], nsd)
    N = 12
    deltas = torch.from_numpy(synth.uniform(3, "geo_deltas", (2 * N, 3), -0.5, 0.5)).double().requires_grad_(True)
    acts = (synth.bits(3, "geo_acts", N) % np.uint64(3)).astype(np.int64) + 1
    actions = torch.from_numpy(np.repeat(acts, 2))
    dtypes = torch.from_numpy(np.tile(np.array([cv.CUR_REL_TO_PREV, cv.PREV_REL_TO_CUR]), N))
    loss = nsd["_compute_geo_invariance_inverse_loss"](None, deltas, actions, dtypes)[0]
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "train_geo_loss.npz"), deltas=deltas.detach().numpy(),
                        actions=actions.numpy(), loss=loss.item(), grad=deltas.grad.numpy())
    print("geo loss", loss.item())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    registry, geo = import_reference()
    full = ["rgb", "depth", "discretized_depth", "top_down_view"]
    if len(sys.argv) > 1 and sys.argv[1] == "f32depth":      # the dense-float32-depth fixtures (simulator-style depth, not float16-exact)
        model_fixture(registry, "model_default_341x192_b2_f32depth.npz", "vo_cnn_rgb_d_dd_top_down", full, (341, 192), 2, 10, 9, False,
                      depth_fp16=False)
        boundary_fixture(registry, geo, "boundary_f32depth.npz", depth_fp16=False, steps=[(5, 6, 2, 0), (6, 7, 1, 3)])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "traindeeper":   # regenerate just the Bottleneck training fixture
        train_fixture(registry, "train_deeper_64x48_b2_f64.npz", "vo_cnn_deeper", ["rgb", "depth"], (64, 48), 2, 0, 34,
                      torch.float64, extra={"backbone": "resnet101"})
        return
    if len(sys.argv) > 1 and sys.argv[1] == "actembed":      # regenerate just the act-embed training fixture
        train_fixture(registry, "train_act_embed_64x48_b5_f64.npz", "vo_cnn_act_embed", ["rgb", "depth"], (64, 48), 5, 0, 33,
                      torch.float64, actions=[1, 3, 2, 3, 1])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "joint":         # regenerate just the joint-training fixtures
        joint_train_fixture(registry, "train_joint_64x48_p4.npz", (64, 48), 4, 41, True)
        joint_train_fixture(registry, "train_joint_45x37_p3_w.npz", (45, 37), 3, 42, False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "deeper":        # regenerate just the resnet101 fixture
        model_fixture(registry, "model_deeper_64x48_b2.npz", "vo_cnn_deeper", ["rgb", "depth"], (64, 48), 2, 0, 8, False,
                      extra={"backbone": "resnet101"})
        return
    model_fixture(registry, "model_default_341x192_b2.npz", "vo_cnn_rgb_d_dd_top_down", full, (341, 192), 2, 10, 0, False)
    model_fixture(registry, "model_default_45x37_b3.npz", "vo_cnn_rgb_d_dd_top_down", full, (45, 37), 3, 10, 1, True)
    model_fixture(registry, "model_vo_cnn_64x48_b2.npz", "vo_cnn", ["rgb", "depth"], (64, 48), 2, 0, 2, False)
    model_fixture(registry, "model_rgb_d_dd_70x40_b2.npz", "vo_cnn_rgb_d_dd", ["rgb", "depth", "discretized_depth"],
                  (70, 40), 2, 10, 3, False)
    model_fixture(registry, "model_wider_64x48_b2.npz", "vo_cnn_wider", ["rgb", "depth"], (64, 48), 2, 0, 4, False)
    model_fixture(registry, "model_deeper_64x48_b2.npz", "vo_cnn_deeper", ["rgb", "depth"], (64, 48), 2, 0, 8, False,
                  extra={"backbone": "resnet101"})
    model_fixture(registry, "model_act_embed_64x48_b3.npz", "vo_cnn_act_embed", ["rgb", "depth"], (64, 48), 3, 0, 5, False)
    model_fixture(registry, "model_d_dd_tdv_66x34_b2.npz", "vo_cnn_d_dd_top_down",
                  ["depth", "discretized_depth", "top_down_view"], (66, 34), 2, 10, 6, False)
    preproc_fixture(geo)
    boundary_fixture(registry, geo)
    train_fixture(registry, "train_default_45x37_b4_f64.npz", "vo_cnn_rgb_d_dd_top_down", full, (45, 37), 4, 10, 31, torch.float64)
    train_fixture(registry, "train_default_96x64_b3_f32.npz", "vo_cnn_rgb_d_dd_top_down", full, (96, 64), 3, 10, 32, torch.float32)
    train_fixture(registry, "train_act_embed_64x48_b5_f64.npz", "vo_cnn_act_embed", ["rgb", "depth"], (64, 48), 5, 0, 33,
                  torch.float64, actions=[1, 3, 2, 3, 1])
    train_fixture(registry, "train_deeper_64x48_b2_f64.npz", "vo_cnn_deeper", ["rgb", "depth"], (64, 48), 2, 0, 34,
                  torch.float64, extra={"backbone": "resnet101"})
    geo_loss_fixture()
    joint_train_fixture(registry, "train_joint_64x48_p4.npz", (64, 48), 4, 41, True)
    joint_train_fixture(registry, "train_joint_45x37_p3_w.npz", (45, 37), 3, 42, False)


if __name__ == "__main__":
    main()
