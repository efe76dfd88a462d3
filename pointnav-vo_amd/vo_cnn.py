"""HIP-backed VO models registered under the reference's names.

Drop-in for the classes of /root/reference/pointnav_vo/vo/models/vo_cnn.py:182-561 and vo_cnn_act_embed.py:17-112:
same registry names, same keyword-only constructor contract (base_trainer_with_vo.py:68-80), same ``state_dict``
keys/shapes, same ``forward(observation_pairs[, actions]) -> Tensor[B, output_dim]``.

The module tree only HOLDS parameters (so ``load_state_dict`` / ``state_dict`` / ``.to`` / ``.parameters`` behave
like the reference's); the forward is one call into libpnvo.so (hand-written gfx950 kernels) on the caller's
current HIP stream.  No torch ops run in the forward, and there is no CPU or eager fallback: a model that is not on
a CUDA(ROCm) device, or whose HIP extension is missing, raises.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import model_spec as ms
from .common_vars import DEFAULT_DELTA_STATE_SIZE, N_ACTS, TOP_DOWN_VIEW_PAIR_CHANNEL
from .registry import baseline_registry


class _Holder(nn.Module):
    """Anonymous container so dotted state_dict names resolve exactly as in the reference module tree."""


def _init_tensor(name, shape, fan_in=None):
    t = torch.empty(shape)
    if len(shape) == 4 or (len(shape) == 2 and name != "action_embedding.weight"):
        if name == "output_head.1.weight":
            nn.init.orthogonal_(t)                      # vo_cnn.py:226
        else:
            nn.init.kaiming_uniform_(t, a=math.sqrt(5))  # torch default of nn.Conv2d / nn.Linear
    elif name == "action_embedding.weight":
        nn.init.normal_(t)                              # torch default of nn.Embedding
    elif len(shape) == 1:
        leaf = name.rsplit(".", 1)[1]
        is_gn = name.startswith("visual_encoder")
        if leaf == "weight":
            t.fill_(1.0)                                # GroupNorm gamma
        elif is_gn or name == "output_head.1.bias":
            t.zero_()                                   # GroupNorm beta; head bias (vo_cnn.py:227)
        else:
            bound = 1.0 / math.sqrt(fan_in) if fan_in else 0.0   # torch default of nn.Linear bias
            t.uniform_(-bound, bound)
    else:
        t.zero_()
    return t


class VisualOdometryCNNBase(nn.Module):
    """Mirror of VisualOdometryCNNBase (vo_cnn.py:182-233)."""

    _ACT_EMBED = False

    def __init__(self, *, observation_space, observation_size, hidden_size=512, resnet_baseplanes=32,
                 backbone="resnet18", normalize_visual_inputs=False, output_dim=DEFAULT_DELTA_STATE_SIZE,
                 dropout_p=0.2, after_compression_flat_size=2048, rgb_pair_channel=ms.RGB_PAIR_CHANNEL,
                 depth_pair_channel=ms.DEPTH_PAIR_CHANNEL, discretized_depth_channels=0,
                 top_down_view_pair_channel=TOP_DOWN_VIEW_PAIR_CHANNEL, n_acts=N_ACTS):
        super().__init__()
        self.cfg = ms.config_from_kwargs(
            observation_space=observation_space, observation_size=observation_size, hidden_size=hidden_size,
            resnet_baseplanes=resnet_baseplanes, backbone=backbone, normalize_visual_inputs=normalize_visual_inputs,
            output_dim=output_dim, dropout_p=dropout_p, discretized_depth_channels=discretized_depth_channels,
            after_compression_flat_size=after_compression_flat_size, rgb_pair_channel=rgb_pair_channel,
            depth_pair_channel=depth_pair_channel, top_down_view_pair_channel=top_down_view_pair_channel,
            act_embed=self._ACT_EMBED, n_acts=n_acts)
        assert self.cfg.in_channels > 0, "visual odometry must not be blind"   # vo_cnn.py:67-68
        self.dropout_p = dropout_p
        self._spec = ms.state_dict_spec(self.cfg)
        for name, shape in self._spec:
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, _Holder())
                mod = getattr(mod, p)
            fan_in = dict(self._spec).get(name[: -len("bias")] + "weight", (0, 0))[-1] if name.endswith(".bias") else None
            t = _init_tensor(name, tuple(shape), fan_in)
            if parts[-1] in ("_mean", "_var", "_count"):     # RunningMeanAndVar buffers (running_mean_and_var.py:16-18)
                mod.register_buffer(parts[-1], torch.zeros(tuple(shape)))
            else:
                mod.register_parameter(parts[-1], nn.Parameter(t))
        self._handle = None
        self._handle_dev = None
        self._loaded_sig = None
        self._precision = "float32"
        self._options = {}

    # ------------------------------------------------------------------ libpnvo plumbing
    def set_precision(self, precision):
        """"float32" (default) or "bfloat16" (BASELINE config 3: bf16 operands / activations, fp32 accumulation and
        normalisation statistics; resnet18 models).  Applies to the eval-mode forward and to dual_forward()."""
        if precision not in ("float32", "bfloat16"):
            raise ValueError(precision)
        self._precision = precision
        if self._handle is not None:
            _lib.check(_lib.lib.pnvo_set_precision(self._handle, int(precision == "bfloat16")), self._handle)
        return self

    def set_option(self, key, value):
        """Per-model kernel-selection / behaviour option (include/pnvo.h pnvo_set_option: "stem", "conv", "tail", "pool",
        "input_fallback", ...).  Remembered and re-applied when the handle is (re)created; the PNVO_* environment only
        provides the defaults, read once at handle creation."""
        self._options[str(key)] = str(value)
        if self._handle is not None:
            _lib.check(_lib.lib.pnvo_set_option(self._handle, str(key).encode(), str(value).encode()), self._handle)
        return self

    def get_option(self, key):
        dev = next(self.parameters()).device
        self._ensure_handle(dev)
        buf = C.create_string_buffer(64)
        _lib.check(_lib.lib.pnvo_get_option(self._handle, str(key).encode(), buf, 64), self._handle)
        return buf.value.decode()

    def last_note(self):
        """pnvo_last_note of this model's handle: the one-line note of a successful call that changed the handle's behaviour
        (the dense-stem fallback); errors travel separately (exceptions raised from pnvo_last_error)."""
        if self._handle is None:
            return ""
        msg = _lib.lib.pnvo_last_note(self._handle)
        return msg.decode() if msg else ""

    def _tensors(self):
        """(name, tensor) in state_dict order.  Walking the module tree costs ~0.1 ms and even one getattr per tensor through
        nn.Module.__getattr__ ~50 us — which a batch-1 boundary call pays on every forward, and a grouped call of three action models
        six times.  PARAMETERS are therefore resolved once and kept (nn.Module._apply — .to() / .cuda() / .float() — swaps their .data
        in place or, with the overwrite-params future flag, replaces them: _apply below drops the cache; load_state_dict copies in
        place: the version counter in _sync_weights' signature sees it); the BUFFERS (RunningMeanAndVar's three, re-assigned by every
        train-mode forward) are read through getattr each time.  A parameter object replaced by hand (module.weight = nn.Parameter(..))
        needs model._tensor_cache = None."""
        cache = getattr(self, "_tensor_cache", None)
        if cache is None:
            cache = []
            for name, _ in self._spec:
                parts = name.split(".")
                mod = self
                for q in parts[:-1]:
                    mod = getattr(mod, q)
                leaf = parts[-1]
                cache.append((name, None, mod, leaf) if leaf in mod._buffers else (name, getattr(mod, leaf), None, None))
            object.__setattr__(self, "_tensor_cache", cache)
        return [(name, t if mod is None else getattr(mod, leaf)) for name, t, mod, leaf in cache]

    def _apply(self, fn, *a, **k):
        object.__setattr__(self, "_tensor_cache", None)
        return super()._apply(fn, *a, **k)

    def _ref_param(self):
        """The model's first parameter (its device decides where the handle lives) without walking the module tree."""
        cache = getattr(self, "_tensor_cache", None)
        if cache is None:
            self._tensors()
            cache = self._tensor_cache
        for _, t, mod, _leaf in cache:
            if mod is None:
                return t
        return next(self.parameters())

    def _ensure_handle(self, device):
        if self._handle is not None and self._handle_dev == device.index:
            return
        self._release()
        c = self.cfg
        cc = _lib.pnvo_config(width=c.width, height=c.height, n_rgb=c.n_rgb, n_depth=c.n_depth, n_dd=c.n_dd,
                              n_tdv=c.n_tdv, baseplanes=c.baseplanes, hidden=c.hidden, out_dim=c.out_dim,
                              normalize=int(c.normalize), act_embed=int(c.act_embed), n_acts=c.n_acts,
                              flat_size=c.after_compression_flat_size, max_batch=0,
                              backbone_depth=c.backbone_depth)
        h = C.c_void_p()
        _lib.check(_lib.lib.pnvo_create(C.byref(cc), int(device.index or 0), C.byref(h)))
        self._handle, self._handle_dev, self._loaded_sig = h, device.index, None
        _lib.check(_lib.lib.pnvo_set_precision(h, int(self._precision == "bfloat16")), h)
        for k, v in self._options.items():
            _lib.check(_lib.lib.pnvo_set_option(h, k.encode(), v.encode()), h)

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib.pnvo_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _sync_weights(self):
        tensors = self._tensors()
        sig = tuple([(t.data_ptr(), t._version) for _, t in tensors])
        if sig == self._loaded_sig:
            return
        blob = np.concatenate([t.detach().to("cpu", torch.float32).reshape(-1).numpy() for _, t in tensors])
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        toc = (_lib.pnvo_tensor_desc * len(tensors))()
        off = 0
        for i, (name, t) in enumerate(tensors):
            toc[i].name = name.encode()
            toc[i].offset = off
            toc[i].ndim = t.dim()
            for k, s in enumerate(t.shape):
                toc[i].shape[k] = int(s)
            off += t.numel()
        _lib.check(_lib.lib.pnvo_load_weights(self._handle, blob.ctypes.data_as(C.c_void_p), blob.size, toc,
                                              len(tensors)), self._handle)
        self._loaded_sig = sig

    # ------------------------------------------------------------------ forward
    def forward(self, observation_pairs, actions=None):
        if self.training:
            # train-mode forward (the reference's 'rnd' mode calls it at inference, base_trainer_with_vo.py:295-308):
            # dropout active AND RunningMeanAndVar updated by the batch, exactly as nn.Module.train() implies there
            if self.cfg.act_embed and actions is None:
                raise TypeError("forward() missing required argument 'actions' (act_embed model)")
            ts = getattr(self, "_train_step", None)
            if ts is None:
                from .train import VOTrainStep
                ts = VOTrainStep(self)
                object.__setattr__(self, "_train_step", ts)
            return ts.forward_train(observation_pairs, actions)
        ref = self._ref_param()
        if ref.device.type != "cuda":
            raise RuntimeError("pointnav_vo_amd VO models run on an MI355X only: move the model with .to('cuda') "
                               "(there is no CPU fallback)")
        dev = ref.device
        self._ensure_handle(dev)
        self._sync_weights()
        c = self.cfg
        ptrs, B, keep = [], None, []
        for key, n in (("rgb", c.n_rgb), ("depth", c.n_depth), ("discretized_depth", c.n_dd), ("top_down_view", c.n_tdv)):
            if n == 0:
                ptrs.append(None)
                continue
            t = observation_pairs[key]
            if t.device != dev:
                raise RuntimeError(f"observation '{key}' is on {t.device}, model on {dev}")
            t = t.to(torch.float32).contiguous()
            if t.dim() != 4 or t.shape[1] != c.height or t.shape[2] != c.width or t.shape[3] != n:
                raise ValueError(f"observation '{key}' has shape {tuple(t.shape)}, expected [B,{c.height},{c.width},{n}]")
            B = t.shape[0] if B is None else B
            assert t.shape[0] == B
            keep.append(t)
            ptrs.append(C.c_void_p(t.data_ptr()))
        act_ptr = None
        if c.act_embed:
            if actions is None:
                raise TypeError("forward() missing required argument 'actions' (act_embed model)")
            a = actions.to(device=dev, dtype=torch.int64).contiguous().reshape(-1)
            keep.append(a)
            act_ptr = C.c_void_p(a.data_ptr())
        out = torch.empty((B, c.out_dim), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.lib.pnvo_forward(self._handle, ptrs[0], ptrs[1], ptrs[2], ptrs[3], act_ptr, int(B),
                                             C.c_void_p(out.data_ptr()), C.c_void_p(stream)), self._handle)
        return out

    def forward_raw(self, rgb_frames, depth_frames, top_down_view=None, actions=None, err_flag=None):
        """The eval forward from the SENSOR frames (pnvo_forward_raw): rgb_frames uint8 [B,2,H,W,3] (prev, cur; None for models
        without rgb), depth_frames float32 [B,2,H,W] in 0..1, top_down_view float32 [B,H,W,2] (None without that modality) —
        CUDA tensors.  Pair concatenation, the uint8 -> float cast and the one-hot depth happen in the stem's operand fetch: no
        observation-pair tensors are built.  err_flag: optional int32 CUDA tensor [1], set when a depth is outside [0, 1]."""
        if self.training:
            raise RuntimeError("forward_raw is the eval-mode forward (train-mode forwards take observation pairs)")
        ref = self._ref_param()
        if ref.device.type != "cuda":
            raise RuntimeError("pointnav_vo_amd VO models run on an MI355X only (there is no CPU fallback)")
        dev = ref.device
        self._ensure_handle(dev)
        self._sync_weights()
        c = self.cfg
        B = depth_frames.shape[0]
        keep = []

        def ptr(t, dtype, shape, what):
            if t is None:
                return None
            if t.device != dev or t.dtype != dtype or tuple(t.shape) != shape:
                raise ValueError(f"{what}: expected {dtype} {shape} on {dev}, got {t.dtype} {tuple(t.shape)} on {t.device}")
            t = t.contiguous()
            keep.append(t)
            return C.c_void_p(t.data_ptr())

        p_rgb = ptr(rgb_frames if c.n_rgb else None, torch.uint8, (B, 2, c.height, c.width, 3), "rgb_frames")
        p_dep = ptr(depth_frames, torch.float32, (B, 2, c.height, c.width), "depth_frames")
        p_tdv = ptr(top_down_view if c.n_tdv else None, torch.float32, (B, c.height, c.width, 2), "top_down_view")
        if c.n_rgb and p_rgb is None:
            raise ValueError("this model has the rgb modality: rgb_frames is required")
        if c.n_tdv and p_tdv is None:
            raise ValueError("this model has the top_down_view modality: top_down_view is required")
        act_ptr = None
        if c.act_embed:
            if actions is None:
                raise TypeError("forward_raw() missing required argument 'actions' (act_embed model)")
            a = actions.to(device=dev, dtype=torch.int64).contiguous().reshape(-1)
            keep.append(a)
            act_ptr = C.c_void_p(a.data_ptr())
        out = torch.empty((B, c.out_dim), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.lib.pnvo_forward_raw(self._handle, p_rgb, p_dep, p_tdv, act_ptr, int(B), C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(err_flag.data_ptr()) if err_flag is not None else None,
                                                 C.c_void_p(stream)), self._handle)
        return out

    # ------------------------------------------------------------------ introspection (tests / bench)
    def tap(self, name, observation_pairs, actions=None):
        """Run a forward and return (output, intermediate activation `name` as an NHWC tensor)."""
        dev = next(self.parameters()).device
        self._ensure_handle(dev)
        first = next(v for v in observation_pairs.values())
        B = first.shape[0]
        shape = (C.c_int64 * 4)()
        _lib.check(_lib.lib.pnvo_tap_shape(self._handle, name.encode(), int(B), shape), self._handle)
        buf = torch.empty(tuple(int(s) for s in shape), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.pnvo_set_tap(self._handle, name.encode(), C.c_void_p(buf.data_ptr()), buf.numel()),
                   self._handle)
        try:
            out = self.forward(observation_pairs, actions)
        finally:
            _lib.lib.pnvo_set_tap(self._handle, None, None, 0)
        return out, buf

    def check_inputs(self):
        """Raise PnvoError if an earlier forward of a handle with input_fallback=off met observation values outside the fused
        stems' contract (fractional rgb, discretised depth that is not one-hot: base_trainer_with_vo.py:163 asserts the
        latter where it builds the observation).  With the default input_fallback=on such a forward is re-run on the dense
        stem inside the call and nothing is raised.  Synchronise first for a definitive answer."""
        if self._handle is not None:
            _lib.check(_lib.lib.pnvo_check_inputs(self._handle), self._handle)

    def layer_kernel(self, name, batch):
        """(kernel family, matrix-core FLOPs one launch executes) of a residual-stage conv at this batch size."""
        dev = next(self.parameters()).device
        self._ensure_handle(dev)
        buf, fl = C.create_string_buffer(32), C.c_double(0.0)
        _lib.check(_lib.lib.pnvo_layer_kernel(self._handle, name.encode(), int(batch), buf, 32, C.byref(fl)), self._handle)
        return buf.value.decode(), float(fl.value)

    def timing(self, enable):
        dev = next(self.parameters()).device
        self._ensure_handle(dev)
        _lib.check(_lib.lib.pnvo_timing_mode(self._handle, int(bool(enable))), self._handle)

    def timing_read(self):
        ent = (_lib.pnvo_kernel_time * 128)()
        n = C.c_int(0)
        _lib.check(_lib.lib.pnvo_timing_read(self._handle, ent, 128, C.byref(n)), self._handle)
        return [dict(name=ent[i].name.decode(), launches=int(ent[i].launches), total_ms=float(ent[i].total_ms),
                     flops=float(ent[i].flops), bytes=float(ent[i].bytes)) for i in range(min(n.value, 128))]


def _obs_ptrs(model, observation_pairs, dev):
    c = model.cfg
    ptrs, B, keep = [], None, []
    for key, n in (("rgb", c.n_rgb), ("depth", c.n_depth), ("discretized_depth", c.n_dd), ("top_down_view", c.n_tdv)):
        if n == 0:
            ptrs.append(None)
            continue
        t = observation_pairs[key]
        if t.device != dev:
            raise RuntimeError(f"observation '{key}' is on {t.device}, model on {dev}")
        t = t.to(torch.float32).contiguous()
        if t.dim() != 4 or t.shape[1] != c.height or t.shape[2] != c.width or t.shape[3] != n:
            raise ValueError(f"observation '{key}' has shape {tuple(t.shape)}, expected [B,{c.height},{c.width},{n}]")
        B = t.shape[0] if B is None else B
        assert t.shape[0] == B
        keep.append(t)
        ptrs.append(C.c_void_p(t.data_ptr()))
    return ptrs, B, keep


def dual_forward(model_a, model_b, observation_pairs):
    """The geometric-invariance dual forward (vo_cnn_regression_geo_invariance_engine.py:569-602): returns
    (model_a(observation_pairs), model_b(swapped observation_pairs)) where the swapped pair exchanges the [prev | cur] halves
    of every observation tensor (regression_geo_invariance_iter_dataset.py:342-386) — computed in ONE pass over the
    observation tensors, both models in every launch, on the bfloat16 path (call set_precision("bfloat16") on both)."""
    ref = next(model_a.parameters())
    if ref.device.type != "cuda" or next(model_b.parameters()).device != ref.device:
        raise RuntimeError("dual_forward: both models must be on the same MI355X (there is no CPU fallback)")
    if model_a.training or model_b.training:
        raise RuntimeError("dual_forward is the eval-mode forward of both models")
    dev = ref.device
    for m in (model_a, model_b):
        m._ensure_handle(dev)
        m._sync_weights()
    ptrs, B, keep = _obs_ptrs(model_a, observation_pairs, dev)
    oa = torch.empty((B, model_a.cfg.out_dim), device=dev, dtype=torch.float32)
    ob = torch.empty_like(oa)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.pnvo_forward_dual(model_a._handle, model_b._handle, ptrs[0], ptrs[1], ptrs[2], ptrs[3], int(B),
                                              C.c_void_p(oa.data_ptr()), C.c_void_p(ob.data_ptr()), C.c_void_p(stream)),
                   model_a._handle)
    return oa, ob


def dual_forward_raw(model_a, model_b, rgb_frames, depth_frames, top_down_view=None, err_flag=None):
    """dual_forward from the sensor frames (pnvo_forward_dual_raw; tensor contract as VisualOdometryCNNBase.forward_raw)."""
    ref = next(model_a.parameters())
    if ref.device.type != "cuda" or next(model_b.parameters()).device != ref.device:
        raise RuntimeError("dual_forward_raw: both models must be on the same MI355X (there is no CPU fallback)")
    if model_a.training or model_b.training:
        raise RuntimeError("dual_forward_raw is the eval-mode forward of both models")
    dev = ref.device
    for m in (model_a, model_b):
        m._ensure_handle(dev)
        m._sync_weights()
    c = model_a.cfg
    B = depth_frames.shape[0]
    rgb = rgb_frames.contiguous() if c.n_rgb else None
    dep = depth_frames.contiguous()
    tdv = top_down_view.contiguous() if c.n_tdv else None
    assert dep.dtype == torch.float32 and tuple(dep.shape) == (B, 2, c.height, c.width) and dep.device == dev
    assert rgb is None or (rgb.dtype == torch.uint8 and tuple(rgb.shape) == (B, 2, c.height, c.width, 3) and rgb.device == dev)
    assert tdv is None or (tdv.dtype == torch.float32 and tuple(tdv.shape) == (B, c.height, c.width, 2) and tdv.device == dev)
    oa = torch.empty((B, c.out_dim), device=dev, dtype=torch.float32)
    ob = torch.empty_like(oa)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.pnvo_forward_dual_raw(model_a._handle, model_b._handle, p(rgb), p(dep), p(tdv), int(B), p(oa), p(ob),
                                                  p(err_flag), C.c_void_p(stream)), model_a._handle)
    return oa, ob


def grouped_supported(models):
    """Can these eval-mode models share a grouped forward (pnvo_grouped_supported)?  -> (bool, reason)."""
    models = list(models)
    if not 1 <= len(models) <= 3 or any(m.training for m in models):
        return False, "one to three eval-mode models"
    dev = models[0]._ref_param().device
    if dev.type != "cuda" or any(m._ref_param().device != dev for m in models):
        return False, "models on one MI355X"
    for m in models:
        m._ensure_handle(dev)
        if m._loaded_sig is None:                  # (the query needs loaded handles, not current values: the forward itself syncs)
            m._sync_weights()
    hs = (C.c_void_p * len(models))(*[m._handle for m in models])
    rc = _lib.lib.pnvo_grouped_supported(hs, len(models))
    return rc == 0, ("" if rc == 0 else _lib.lib.pnvo_last_error(models[0]._handle).decode())


def grouped_forward_raw(models, counts, rgb_frames, depth_frames, top_down_view=None, err_flag=None):
    """The eval forward of up to three SEPARATE-ACTION models over one batch of sensor frames in ONE launch chain
    (pnvo_forward_grouped_raw): models[k] serves counts[k] consecutive pairs (the batch sorted by action model, as
    BaseRLTrainerWithVO.compute_local_delta_states_batch sorts it; tensor contract as VisualOdometryCNNBase.forward_raw).
    What the navigation loop's batched call needs (base_trainer_with_vo.py:277-294 picks vo_model[act] per environment): three
    forwards of a few pairs each are bound by launch latency, one chain over all pairs costs what ONE forward of that size does."""
    models, counts = list(models), [int(x) for x in counts]
    if not 1 <= len(models) <= 3 or len(models) != len(counts):
        raise ValueError("grouped_forward_raw takes one to three models and as many pair counts")
    ref = models[0]._ref_param()
    dev = ref.device
    if dev.type != "cuda" or any(m._ref_param().device != dev for m in models):
        raise RuntimeError("grouped_forward_raw: every model must be on the same MI355X (there is no CPU fallback)")
    if any(m.training for m in models):
        raise RuntimeError("grouped_forward_raw is the eval-mode forward of every model")
    for m in models:
        m._ensure_handle(dev)
        m._sync_weights()
    c = models[0].cfg
    B = depth_frames.shape[0]
    if sum(counts) != B:
        raise ValueError(f"pair counts {counts} do not add up to the batch {B}")
    rgb = rgb_frames.contiguous() if c.n_rgb else None
    dep = depth_frames.contiguous()
    tdv = top_down_view.contiguous() if c.n_tdv else None
    assert dep.dtype == torch.float32 and tuple(dep.shape) == (B, 2, c.height, c.width) and dep.device == dev
    assert rgb is None or (rgb.dtype == torch.uint8 and tuple(rgb.shape) == (B, 2, c.height, c.width, 3) and rgb.device == dev)
    assert tdv is None or (tdv.dtype == torch.float32 and tuple(tdv.shape) == (B, c.height, c.width, 2) and tdv.device == dev)
    out = torch.empty((B, c.out_dim), device=dev, dtype=torch.float32)
    hs = (C.c_void_p * len(models))(*[m._handle for m in models])
    cn = (C.c_int32 * len(models))(*counts)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib.pnvo_forward_grouped_raw(hs, cn, len(models), p(rgb), p(dep), p(tdv), int(B), p(out), p(err_flag),
                                                     C.c_void_p(stream)), models[0]._handle)
    return out


class VisualOdometryCNNActEmbedBase(VisualOdometryCNNBase):
    """Mirror of VisualOdometryCNNActEmbed (vo_cnn_act_embed.py:17-75): forward(observation_pairs, actions)."""
    _ACT_EMBED = True


def _variant(name, *, base=VisualOdometryCNNBase, need=(), forbid=(), dd_zero=False, widen=1, default_dd=0,
             backbone_req="resnet18"):
    """Build and register one reference variant; the asserts are the reference's own (vo_cnn.py:252-255 etc.)."""

    def __init__(self, *, observation_space, observation_size, hidden_size=512, resnet_baseplanes=32,
                 backbone=backbone_req, normalize_visual_inputs=False, output_dim=DEFAULT_DELTA_STATE_SIZE,
                 dropout_p=0.2, discretized_depth_channels=default_dd,
                 top_down_view_pair_channel=TOP_DOWN_VIEW_PAIR_CHANNEL, n_acts=N_ACTS):
        assert backbone == backbone_req
        if dd_zero:
            assert discretized_depth_channels == 0
        for k in need:
            assert k in observation_space
        for k in forbid:
            assert k not in observation_space
        base.__init__(self, observation_space=observation_space, observation_size=observation_size,
                      hidden_size=hidden_size, resnet_baseplanes=widen * resnet_baseplanes, backbone=backbone,
                      normalize_visual_inputs=normalize_visual_inputs, output_dim=output_dim, dropout_p=dropout_p,
                      discretized_depth_channels=discretized_depth_channels, after_compression_flat_size=2048,
                      top_down_view_pair_channel=top_down_view_pair_channel, n_acts=n_acts)

    cls = type("HIP_" + name, (base,), {"__init__": __init__, "__doc__": f"HIP drop-in for registry name '{name}'"})
    return baseline_registry.register_vo_model(cls, name=name)


DD, TDV = "discretized_depth", "top_down_view"
VisualOdometryCNN = _variant("vo_cnn", forbid=(DD, TDV), dd_zero=True)                                   # :236
VisualOdometryCNNRGB = _variant("vo_cnn_rgb", forbid=("depth", DD, TDV), dd_zero=True)                   # :269
VisualOdometryCNNWider = _variant("vo_cnn_wider", forbid=(DD, TDV), dd_zero=True, widen=2)               # :303
VisualOdometryCNNDeeper = _variant("vo_cnn_deeper", forbid=(DD, TDV), dd_zero=True,
                                   backbone_req="resnet101")                                             # :339 (Bottleneck [3,4,23,3])
VisualOdometryCNNDiscretizedDepth = _variant("vo_cnn_rgb_d_dd", need=(DD,), forbid=(TDV,), default_dd=10)  # :373
VisualOdometryCNN_RGB_D_TopDownView = _variant("vo_cnn_rgb_d_top_down", need=("rgb", "depth", TDV), forbid=(DD,))  # :408
VisualOdometryCNN_RGB_DD_TopDownView = _variant("vo_cnn_rgb_dd_top_down", need=("rgb", DD, TDV), forbid=("depth",),
                                                default_dd=10)                                           # :445
VisualOdometryCNN_D_DD_TopDownView = _variant("vo_cnn_d_dd_top_down", need=("depth", DD, TDV), forbid=("rgb",),
                                              default_dd=10)                                             # :483
VisualOdometryCNNDiscretizedDepthTopDownView = _variant("vo_cnn_rgb_d_dd_top_down", need=(DD, TDV), default_dd=10)  # :521
LegacyVisualOdometryCNNDiscretizedDepthTopDownView = _variant("vo_cnn_discretize_depth_top_down", need=(DD, TDV),
                                                              default_dd=10)                             # :557
VisualOdometryCNNActEmbed = _variant("vo_cnn_act_embed", base=VisualOdometryCNNActEmbedBase)             # act_embed.py:17
VisualOdometryCNNWiderActEmbed = _variant("vo_cnn_wider_act_embed", base=VisualOdometryCNNActEmbedBase,
                                          forbid=(DD, TDV), dd_zero=True, widen=2)                       # act_embed.py:78
