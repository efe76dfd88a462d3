"""HIP-backed navigation policy registered under the reference's name ``resnet_rnn_policy``.

Drop-in for PointNavResNetPolicy (/root/reference/pointnav_vo/rl/policies/resnet_policy.py:25-58) in the configuration
the reference's nav loop uses (configs/rl/ddppo_pointnav.yaml:48-54: depth-only resnet18 encoder, 2-layer LSTM, no
observation transform, normalize_visual_inputs False): same constructor keywords (ddppo_trainer.py:122-133), same
``state_dict`` keys/shapes, same ``act`` / ``get_value`` signatures and return values (policy.py:29-50).  The module tree
only HOLDS parameters; ``act`` is one call into libpnvo.so (pnvo_policy_act) on the caller's current HIP stream plus the
categorical sampling / arg-max over the 4 logits, which stays in torch as in the reference (policy.py:38-43).
``evaluate_actions`` (PPO training of the policy) is not built.  No CPU fallback.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .registry import baseline_registry

GOAL_SENSOR = "pointgoal_with_gps_compass"


class _Holder(nn.Module):
    pass


class pnvo_policy_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "baseplanes", "hidden", "n_actions", "rnn_layers", "flat_size")]


def policy_state_dict_spec(*, width, height, baseplanes=32, hidden=512, n_actions=4, rnn_layers=2, flat_size=2048):
    """(name, shape) of every tensor of PointNavResNetPolicy.state_dict() for the depth-only resnet18 configuration."""
    def half(v):
        return (v + 1) // 2
    h, w = height // 2, width // 2                       # F.avg_pool2d(x, 2)
    spec = []
    pre = "net.visual_encoder."
    bb = pre + "backbone."
    spec += [(bb + "conv1.0.weight", (baseplanes, 1, 7, 7)), (bb + "conv1.1.weight", (baseplanes,)),
             (bb + "conv1.1.bias", (baseplanes,))]
    h, w = half(half(h)), half(half(w))                  # stem stride 2 + maxpool
    cin = baseplanes
    for li in range(1, 5):
        planes = baseplanes << (li - 1)
        for bi in range(2):
            p = f"{bb}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            spec += [(p + "convs.0.weight", (planes, cin, 3, 3)), (p + "convs.1.weight", (planes,)),
                     (p + "convs.1.bias", (planes,)), (p + "convs.3.weight", (planes, planes, 3, 3)),
                     (p + "convs.4.weight", (planes,)), (p + "convs.4.bias", (planes,))]
            if stride != 1 or cin != planes:
                spec += [(p + "downsample.0.weight", (planes, cin, 1, 1)), (p + "downsample.1.weight", (planes,)),
                         (p + "downsample.1.bias", (planes,))]
            if stride == 2:
                h, w = half(h), half(w)
            cin = planes
    comp = int(round(flat_size / (h * w)))               # resnet_policy.py:113-117 (python round)
    spec += [(pre + "compression.0.weight", (comp, cin, 3, 3)), (pre + "compression.1.weight", (comp,)),
             (pre + "compression.1.bias", (comp,))]
    spec = [("net.prev_action_embedding.weight", (n_actions + 1, 32)), ("net.tgt_embeding.weight", (32, 3)),
            ("net.tgt_embeding.bias", (32,))] + spec
    spec += [("net.visual_fc.1.weight", (hidden, comp * h * w)), ("net.visual_fc.1.bias", (hidden,))]
    for layer in range(rnn_layers):
        k = hidden + 64 if layer == 0 else hidden
        r = "net.state_encoder.rnn."
        spec += [(f"{r}weight_ih_l{layer}", (4 * hidden, k)), (f"{r}weight_hh_l{layer}", (4 * hidden, hidden)),
                 (f"{r}bias_ih_l{layer}", (4 * hidden,)), (f"{r}bias_hh_l{layer}", (4 * hidden,))]
    spec += [("action_distribution.linear.weight", (n_actions, hidden)), ("action_distribution.linear.bias", (n_actions,)),
             ("critic.fc.weight", (1, hidden)), ("critic.fc.bias", (1,))]
    return spec


def _init(name, shape):
    t = torch.empty(shape)
    leaf = name.rsplit(".", 1)[1]
    if "state_encoder.rnn" in name:                      # rnn_state_encoder.py:36-41
        nn.init.orthogonal_(t) if "weight" in leaf else t.zero_()
    elif name.startswith("action_distribution"):         # misc_utils.py:73-74
        nn.init.orthogonal_(t, gain=0.01) if leaf == "weight" else t.zero_()
    elif name.startswith("critic"):                      # policy.py:70-71
        nn.init.orthogonal_(t) if leaf == "weight" else t.zero_()
    elif name == "net.prev_action_embedding.weight":
        nn.init.normal_(t)
    elif len(shape) in (2, 4):                           # ResNetEncoder.layer_init / torch defaults
        nn.init.kaiming_normal_(t, nn.init.calculate_gain("relu")) if len(shape) == 4 else \
            nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    elif leaf == "weight":
        t.fill_(1.0)                                     # GroupNorm gamma
    else:
        t.zero_()
    return t


class _NetHolder(_Holder):
    """`policy.net` of the reference (PointNavResNetNet, resnet_policy.py:177-282) as far as its callers read it."""

    @property
    def num_recurrent_layers(self):
        return self._layers * 2                            # LSTM: h and c (rnn_state_encoder.py:44-45)

    @property
    def output_size(self):
        return self._hidden

    @property
    def is_blind(self):
        return False                                       # a depth encoder is always present here


@baseline_registry.register_policy(name="resnet_rnn_policy")
class PointNavResNetPolicy(nn.Module):
    def __init__(self, *, observation_space, action_space, goal_sensor_uuid=GOAL_SENSOR, hidden_size=512,
                 num_recurrent_layers=2, rnn_type="LSTM", resnet_baseplanes=32, backbone="resnet18",
                 normalize_visual_inputs=False, obs_transform=None, vis_types=("depth",), **kwargs):
        super().__init__()
        if rnn_type != "LSTM" or backbone != "resnet18":
            raise NotImplementedError("the HIP policy implements the resnet18 + LSTM configuration of ddppo_pointnav.yaml")
        if obs_transform is not None or normalize_visual_inputs or list(vis_types) != ["depth"]:
            raise NotImplementedError("the HIP policy implements the depth-only, untransformed, un-normalised encoder "
                                      "(RL.Policy.visual_types = ['depth'], RL.OBS_TRANSFORM = 'none')")
        if goal_sensor_uuid != GOAL_SENSOR:
            raise NotImplementedError(goal_sensor_uuid)
        shp = observation_space.spaces["depth"].shape     # (H, W, 1)
        self._H, self._W = int(shp[0]), int(shp[1])
        assert int(shp[2]) == 1
        self.dim_actions = int(action_space.n)
        self._hidden, self._layers, self._baseplanes = int(hidden_size), int(num_recurrent_layers), int(resnet_baseplanes)
        self._spec = policy_state_dict_spec(width=self._W, height=self._H, baseplanes=self._baseplanes,
                                            hidden=self._hidden, n_actions=self.dim_actions, rnn_layers=self._layers)
        for name, shape in self._spec:
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, _Holder())
                mod = getattr(mod, p)
            mod.register_parameter(parts[-1], nn.Parameter(_init(name, tuple(shape))))
        # the reference trainers read these through policy.net (ppo_trainer.py:618, ddppo_trainer.py:279)
        net = self.net
        net.__class__ = _NetHolder
        net._layers, net._hidden = self._layers, self._hidden
        self._handle = None
        self._handle_dev = None
        self._loaded_sig = None

    @property
    def num_recurrent_layers(self):
        return self._layers * 2                            # LSTM: h and c (rnn_state_encoder.py:44-45)

    @property
    def output_size(self):
        return self._hidden

    # ------------------------------------------------------------------ libpnvo plumbing
    def _ensure(self, device):
        if self._handle is None or self._handle_dev != device.index:
            self._release()
            cc = pnvo_policy_config(width=self._W, height=self._H, baseplanes=self._baseplanes, hidden=self._hidden,
                                    n_actions=self.dim_actions, rnn_layers=self._layers, flat_size=2048)
            h = C.c_void_p()
            _lib.check(_lib.lib.pnvo_policy_create(C.byref(cc), int(device.index or 0), C.byref(h)))
            self._handle, self._handle_dev, self._loaded_sig = h, device.index, None
        tensors = getattr(self, "_spec_tensors", None)      # (the walk over the module tree costs ~50 us per step: kept until _apply)
        if tensors is None:
            sd = dict(self.named_parameters())
            tensors = self._spec_tensors = [(n, sd[n]) for n, _ in self._spec]
        sig = tuple([(t.data_ptr(), t._version) for _, t in tensors])
        if sig != self._loaded_sig:
            blob = np.ascontiguousarray(np.concatenate(
                [t.detach().to("cpu", torch.float32).reshape(-1).numpy() for _, t in tensors]), dtype=np.float32)
            toc = (_lib.pnvo_tensor_desc * len(tensors))()
            off = 0
            for i, (name, t) in enumerate(tensors):
                toc[i].name = name.encode()
                toc[i].offset = off
                toc[i].ndim = t.dim()
                for k, s in enumerate(t.shape):
                    toc[i].shape[k] = int(s)
                off += t.numel()
            _lib.check(_lib.lib.pnvo_policy_load_weights(self._handle, blob.ctypes.data_as(C.c_void_p), blob.size, toc,
                                                         len(tensors)))
            self._loaded_sig = sig

    def _apply(self, fn, *a, **k):                          # .to() / .cuda() / .float(): parameters may be replaced
        self._spec_tensors = None
        return super()._apply(fn, *a, **k)

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib.pnvo_policy_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ forward
    def _net(self, observations, rnn_hidden_states, prev_actions, masks, want_features=True):
        ref = next(self.parameters())
        if ref.device.type != "cuda":
            raise RuntimeError("pointnav_vo_amd policies run on an MI355X only: move the policy with .to('cuda') "
                               "(there is no CPU fallback)")
        dev = ref.device
        self._ensure(dev)
        depth = observations["depth"].to(device=dev, dtype=torch.float32).contiguous()
        B = depth.shape[0]
        if tuple(depth.shape[1:]) != (self._H, self._W, 1):
            raise ValueError(f"observations['depth'] has shape {tuple(depth.shape)}, expected [B,{self._H},{self._W},1]")
        goal = observations[GOAL_SENSOR].to(device=dev, dtype=torch.float32).contiguous().reshape(B, 2)
        pa = prev_actions.to(device=dev, dtype=torch.int64).contiguous().reshape(B)
        mk = masks.to(device=dev, dtype=torch.float32).contiguous().reshape(B)
        hin = rnn_hidden_states.to(device=dev, dtype=torch.float32).contiguous()
        assert tuple(hin.shape) == (2 * self._layers, B, self._hidden), tuple(hin.shape)
        hout = torch.empty_like(hin)
        feats = torch.empty((B, self._hidden), device=dev, dtype=torch.float32) if want_features else None
        logits = torch.empty((B, self.dim_actions), device=dev, dtype=torch.float32)
        value = torch.empty((B, 1), device=dev, dtype=torch.float32)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(_lib.lib.pnvo_policy_act(self._handle, p(depth), p(goal), p(pa), p(mk), p(hin), int(B), p(hout),
                                                p(feats), p(logits), p(value), stream))
        return feats, hout, logits, value

    def forward(self, *x):
        raise NotImplementedError                          # as the reference (policy.py:26-27)

    def act(self, observations, rnn_hidden_states, prev_actions, masks, deterministic=False):
        """-> (value [B,1], action [B,1] int64, action_log_probs [B,1], rnn_hidden_states)  (policy.py:29-46)."""
        with torch.no_grad():
            _, hout, logits, value = self._net(observations, rnn_hidden_states, prev_actions, masks, want_features=False)
            # CategoricalNet's distribution (policy.py:29-46 -> utils.CategoricalNet): log-probabilities = logits - logsumexp, probabilities
            # = their softmax, sample = multinomial(probabilities, 1), log_prob = gather.  Written out (log_softmax, exp, multinomial,
            # gather: four launches instead of the distribution object's ten; the sampling call and its generator use are unchanged)
            logp_all = torch.log_softmax(logits, dim=-1)
            probs = logp_all.exp()
            action = probs.argmax(dim=-1, keepdim=True) if deterministic else torch.multinomial(probs, 1, True)
            logp = logp_all.gather(-1, action)
        return value, action, logp, hout

    def get_value(self, observations, rnn_hidden_states, prev_actions, masks):
        with torch.no_grad():
            return self._net(observations, rnn_hidden_states, prev_actions, masks)[3]

    def features_and_logits(self, observations, rnn_hidden_states, prev_actions, masks):
        """(features [B,hidden], rnn_hidden_states, logits [B,n], value [B,1]) — for checkers."""
        with torch.no_grad():
            return self._net(observations, rnn_hidden_states, prev_actions, masks)

    def evaluate_actions(self, *a, **k):
        raise NotImplementedError("PPO training of the policy (policy.py:52-63) is outside the built path")
