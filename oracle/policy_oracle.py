"""CPU restatement of the navigation policy's per-step forward (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Follows /root/reference/pointnav_vo/rl/policies/resnet_policy.py:143-174 (ResNetEncoder.forward: avg_pool2d(2), backbone,
compression), :234-282 (PointNavResNetNet.forward: visual_fc, goal embedding of (rho, cos(-phi), sin(-phi)), previous-
action embedding of ((a + 1) * mask), concatenation, RNN), model_utils/rnns/rnn_state_encoder.py:63-79 (masked hidden
state, single step), torch.nn.LSTM's cell equations (gate order i, f, g, o), policy.py:32-36 and utils/misc_utils.py:67-78
(action logits) and policy.py:66-74 (value).  The convolutional part runs on the C oracle (oracle/pnvo_oracle_net.c).
Pinned by tests/golden/policy_*.npz captured from the imported reference (tests/golden/gen_golden_policy.py).
"""
import numpy as np

from . import oracle


def avgpool2(depth):
    """F.avg_pool2d(x, 2) on NHWC [B,H,W,1] (floor)."""
    d = np.asarray(depth)
    B, H, W, _ = d.shape
    Ho, Wo = H // 2, W // 2
    d = d[:, : 2 * Ho, : 2 * Wo, :]
    s = ((d[:, 0::2, 0::2] + d[:, 0::2, 1::2]) + d[:, 1::2, 0::2]) + d[:, 1::2, 1::2]
    return (s * d.dtype.type(0.25)).astype(d.dtype)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def policy_step(sd, depth, goal, prev_actions, masks, hidden, *, baseplanes=32, dtype=np.float64):
    """sd: PointNavResNetPolicy.state_dict() as dict name -> ndarray.  depth [B,H,W,1]; goal [B,2] (rho, phi);
    prev_actions [B] int; masks [B]; hidden [2L,B,Hd] (h layers, then c layers).
    Returns dict(features, hidden, logits, value)."""
    dt = np.dtype(dtype)
    g = lambda k: np.asarray(sd[k]).astype(dt)
    x = avgpool2(np.asarray(depth, dtype=dt))
    comp = oracle.encoder_({k: np.asarray(v) for k, v in sd.items()}, np.ascontiguousarray(x), baseplanes // 2,
                           pre="net.visual_encoder.")
    vis = oracle.linear(oracle.flatten_nchw(comp), g("net.visual_fc.1.weight"), g("net.visual_fc.1.bias"), True)
    goal = np.asarray(goal, dtype=dt)
    gobs = np.stack([goal[:, 0], np.cos(-goal[:, 1]), np.sin(-goal[:, 1])], axis=-1)
    tgt = gobs @ g("net.tgt_embeding.weight").T + g("net.tgt_embeding.bias")
    m = np.asarray(masks, dtype=np.float32).reshape(-1)
    idx = ((np.asarray(prev_actions).astype(np.float32).reshape(-1) + 1.0) * m).astype(np.int64)
    emb = g("net.prev_action_embedding.weight")[idx]
    xin = np.concatenate([vis.astype(dt), tgt, emb], axis=1)
    hidden = np.asarray(hidden, dtype=dt)
    L = hidden.shape[0] // 2
    Hd = hidden.shape[2]
    md = m.astype(dt)[:, None]
    out = np.empty_like(hidden)
    for l in range(L):
        r = "net.state_encoder.rnn."
        h_prev, c_prev = hidden[l] * md, hidden[L + l] * md
        gates = xin @ g(f"{r}weight_ih_l{l}").T + g(f"{r}bias_ih_l{l}") + h_prev @ g(f"{r}weight_hh_l{l}").T + \
            g(f"{r}bias_hh_l{l}")
        i_, f_, g_, o_ = (gates[:, k * Hd:(k + 1) * Hd] for k in range(4))
        c = _sigmoid(f_) * c_prev + _sigmoid(i_) * np.tanh(g_)
        h = _sigmoid(o_) * np.tanh(c)
        out[l], out[L + l] = h, c
        xin = h
    feats = out[L - 1]
    logits = feats @ g("action_distribution.linear.weight").T + g("action_distribution.linear.bias")
    value = feats @ g("critic.fc.weight").T + g("critic.fc.bias")
    return dict(features=feats, hidden=out, logits=logits, value=value)
