"""oracle/oracle.py — ctypes front-end of the CPU checker (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (pointnav-vo_amd/) never does.  It composes the C primitives of pnvo_oracle_net.c / pnvo_oracle_pre.c
into the reference's forward:

    VisualOdometryCNNBase.forward          /root/reference/pointnav_vo/vo/models/vo_cnn.py:229-233
    ResNetEncoder.forward                  vo_cnn.py:110-179
    ResNet.forward / BasicBlock.forward    pointnav_vo/model_utils/visual_encoders/resnet.py:214-223, 47-55
    VisualOdometryCNNActEmbed.forward      pointnav_vo/vo/models/vo_cnn_act_embed.py:61-75

Weights are consumed in the reference's own ``state_dict`` layout (dict name -> ndarray, OIHW convs).
Parity status: pinned against golden vectors captured from the imported reference (tests/golden/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the checker libraries with gcc (make -C oracle)."""
    need = force or any(
        not os.path.exists(os.path.join(_HERE, n))
        or os.path.getmtime(os.path.join(_HERE, n)) < os.path.getmtime(os.path.join(_HERE, src))
        for n, src in (
            ("liboracle_f32.so", "pnvo_oracle_net.c"),
            ("liboracle_f64.so", "pnvo_oracle_net.c"),
            ("liboracle_pre.so", "pnvo_oracle_pre.c"),
        )
    )
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _lib(name):
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _LIBS[name] = C.CDLL(path)
        if name != "liboracle_pre.so":
            # default: a moderate team; a 256-thread OpenMP team on a big host mostly spins (callers may raise it)
            _LIBS[name].orc_set_threads(min(usable_cores(), 16))
    return _LIBS[name]


def _net(dtype):
    dtype = np.dtype(dtype)
    lib = _lib("liboracle_f32.so" if dtype == np.float32 else "liboracle_f64.so")
    assert lib.orc_sizeof_real() == dtype.itemsize
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_threads(n):
    for name in ("liboracle_f32.so", "liboracle_f64.so"):
        _lib(name).orc_set_threads(int(n))


def max_threads():
    return int(_lib("liboracle_f32.so").orc_get_max_threads())


# ----------------------------------------------------------------------------- primitives
def assemble_whiten(obs, mean, var, dtype=np.float32):
    """obs: dict with any of rgb [B,H,W,6], depth [B,H,W,2], discretized_depth [B,H,W,2*bins],
    top_down_view [B,H,W,2] (NHWC, reference value ranges).  mean/var: [C] or None (no whitening)."""
    lib = _net(dtype)
    arrs = {}
    for k in ("rgb", "depth", "discretized_depth", "top_down_view"):
        arrs[k] = np.ascontiguousarray(obs[k], dtype=dtype) if k in obs and obs[k] is not None else None
    first = next(a for a in arrs.values() if a is not None)
    B, H, W = first.shape[:3]
    n = [0 if arrs[k] is None else arrs[k].shape[3] for k in ("rgb", "depth", "discretized_depth", "top_down_view")]
    Cc = sum(n)
    out = np.empty((B, H, W, Cc), dtype=dtype)
    normalize = mean is not None
    m = np.ascontiguousarray(np.asarray(mean).reshape(-1), dtype=dtype) if normalize else None
    v = np.ascontiguousarray(np.asarray(var).reshape(-1), dtype=dtype) if normalize else None
    lib.orc_assemble_whiten(
        _p(arrs["rgb"]), _p(arrs["depth"]), _p(arrs["discretized_depth"]), _p(arrs["top_down_view"]),
        B, H, W, n[0], n[1], n[2], n[3], _p(m), _p(v), int(normalize), _p(out))
    return out


def conv2d(x, w_oihw, stride, pad):
    dtype = x.dtype
    lib = _net(dtype)
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w_oihw, dtype=dtype)
    B, H, W, Cin = x.shape
    Cout, Cin2, KH, KW = w.shape
    assert Cin == Cin2
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    out = np.empty((B, Ho, Wo, Cout), dtype=dtype)
    lib.orc_conv2d(_p(x), B, H, W, Cin, _p(w), Cout, KH, KW, stride, pad, _p(out))
    return out


def groupnorm_(x, G, gamma, beta, relu, eps=1e-5):
    dtype = x.dtype
    lib = _net(dtype)
    assert x.flags.c_contiguous
    B, Cc = x.shape[0], x.shape[-1]
    P = int(np.prod(x.shape[1:-1]))
    g = np.ascontiguousarray(gamma, dtype=dtype)
    b = np.ascontiguousarray(beta, dtype=dtype)
    lib.orc_groupnorm(_p(x), B, C.c_long(P), Cc, G, _p(g), _p(b), C.c_double(eps), int(relu))
    return x


def maxpool3x3s2p1(x):
    dtype = x.dtype
    lib = _net(dtype)
    x = np.ascontiguousarray(x)
    B, H, W, Cc = x.shape
    out = np.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=dtype)
    lib.orc_maxpool3x3s2p1(_p(x), B, H, W, Cc, _p(out))
    return out


def add_relu_(a, b):
    lib = _net(a.dtype)
    assert a.flags.c_contiguous and b.flags.c_contiguous and a.shape == b.shape
    lib.orc_add_relu(_p(a), _p(b), C.c_long(a.size))
    return a


def flatten_nchw(x):
    lib = _net(x.dtype)
    x = np.ascontiguousarray(x)
    B, H, W, Cc = x.shape
    out = np.empty((B, Cc * H * W), dtype=x.dtype)
    lib.orc_flatten_nchw(_p(x), B, H, W, Cc, _p(out))
    return out


def linear(x, w, b, relu):
    dtype = x.dtype
    lib = _net(dtype)
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w, dtype=dtype)
    bb = np.ascontiguousarray(b, dtype=dtype) if b is not None else None
    B, K = x.shape
    N = w.shape[0]
    out = np.empty((B, N), dtype=dtype)
    lib.orc_linear(_p(x), B, K, _p(w), _p(bb), N, int(relu), _p(out))
    return out


# ----------------------------------------------------------------------------- whole forward
def encoder_(sd, x, ngroups, pre="visual_encoder.", taps=None):
    """GroupNorm-ResNet18 backbone + compression block on an NHWC input (resnet.py:153-212; vo_cnn.py:85-95 /
    rl/policies/resnet_policy.py:118-127).  Returns the compressed feature map, NHWC."""
    g = lambda k: np.asarray(sd[k])
    bb = pre + "backbone."
    x = conv2d(x, g(bb + "conv1.0.weight"), 2, 3)                       # resnet.py:156-163
    if taps is not None:
        taps["stem_conv"] = x.copy()
    groupnorm_(x, ngroups, g(bb + "conv1.1.weight"), g(bb + "conv1.1.bias"), True)  # :165-166
    x = maxpool3x3s2p1(x)                                              # :168
    if taps is not None:
        taps["maxpool"] = x.copy()
    for li in range(1, 5):                                             # :175-184
        bi = 0
        while (bb + f"layer{li}.{bi}.convs.0.weight") in sd:
            p = bb + f"layer{li}.{bi}."
            if (p + "convs.6.weight") in sd:                            # Bottleneck, resnet.py:58-69,93-117
                stride = 2 if (li > 1 and bi == 0) else 1
                out = conv2d(x, g(p + "convs.0.weight"), 1, 0)
                groupnorm_(out, ngroups, g(p + "convs.1.weight"), g(p + "convs.1.bias"), True)
                out = conv2d(out, g(p + "convs.3.weight"), stride, 1)
                groupnorm_(out, ngroups, g(p + "convs.4.weight"), g(p + "convs.4.bias"), True)
                out = conv2d(out, g(p + "convs.6.weight"), 1, 0)
                groupnorm_(out, ngroups, g(p + "convs.7.weight"), g(p + "convs.7.bias"), False)
                if (p + "downsample.0.weight") in sd:
                    res = conv2d(x, g(p + "downsample.0.weight"), stride, 0)
                    groupnorm_(res, ngroups, g(p + "downsample.1.weight"), g(p + "downsample.1.bias"), False)
                else:
                    res = x
                x = add_relu_(out, np.ascontiguousarray(res))
                if taps is not None:
                    taps[f"layer{li}.{bi}"] = x.copy()
                bi += 1
                continue
            stride = 2 if (p + "downsample.0.weight") in sd else 1
            out = conv2d(x, g(p + "convs.0.weight"), stride, 1)         # BasicBlock, resnet.py:37-43
            groupnorm_(out, ngroups, g(p + "convs.1.weight"), g(p + "convs.1.bias"), True)
            out = conv2d(out, g(p + "convs.3.weight"), 1, 1)
            groupnorm_(out, ngroups, g(p + "convs.4.weight"), g(p + "convs.4.bias"), False)
            if stride == 2:                                            # :190-195
                res = conv2d(x, g(p + "downsample.0.weight"), 2, 0)
                groupnorm_(res, ngroups, g(p + "downsample.1.weight"), g(p + "downsample.1.bias"), False)
            else:
                res = x
            x = add_relu_(out, np.ascontiguousarray(res))              # :55
            if taps is not None:
                taps[f"layer{li}.{bi}"] = x.copy()
            bi += 1
    x = conv2d(x, g(pre + "compression.0.weight"), 1, 1)               # vo_cnn.py:85-95
    groupnorm_(x, 1, g(pre + "compression.1.weight"), g(pre + "compression.1.bias"), True)
    if taps is not None:
        taps["compression"] = x.copy()
    return x


def forward(sd, obs, *, ngroups, dtype=np.float32, actions=None, taps=None):
    """Reference forward on CPU.  sd: reference state_dict as dict name -> ndarray.
    ngroups = resnet_baseplanes // 2 (vo_cnn.py:206).  actions: int array [B] for the act_embed variants.
    taps: optional dict that receives the intermediate activations (NHWC) by name."""
    dtype = np.dtype(dtype)
    g = lambda k: np.asarray(sd[k])
    pre = "visual_encoder."
    has_norm = (pre + "running_mean_and_var._mean") in sd
    x = assemble_whiten(
        obs,
        g(pre + "running_mean_and_var._mean") if has_norm else None,
        g(pre + "running_mean_and_var._var") if has_norm else None,
        dtype=dtype,
    )
    if taps is not None:
        taps["input"] = x.copy()
    x = encoder_(sd, x, ngroups, pre, taps)
    feats = flatten_nchw(x)                                            # vo_cnn.py:217
    if "action_embedding.weight" in sd:                                # vo_cnn_act_embed.py:63-72
        emb = g("action_embedding.weight").astype(dtype)[np.asarray(actions).astype(np.int64)]
        feats = np.concatenate([feats, emb], axis=1)
        h = linear(feats, g("hidden_generator.1.weight"), g("hidden_generator.1.bias"), True)
    else:
        h = linear(feats, g("visual_fc.2.weight"), g("visual_fc.2.bias"), True)   # vo_cnn.py:219-220
    if taps is not None:
        taps["hidden"] = h.copy()
    return linear(h, g("output_head.1.weight"), g("output_head.1.bias"), False)  # :223-227


# ----------------------------------------------------------------------------- pre-processing
def discretize_depth(depth, bins):
    """base_trainer_with_vo.py:135-167 on an arbitrary-shape float32 array; returns (onehot [...,bins], n_fired)."""
    lib = _lib("liboracle_pre.so")
    lib.orc_discretize_depth.restype = C.c_long
    d = np.ascontiguousarray(depth, dtype=np.float32)
    out = np.empty(d.shape + (bins,), dtype=np.float32)
    fired = lib.orc_discretize_depth(_p(d), C.c_long(d.size), int(bins), _p(out))
    return out, int(fired)


def topdown_consts(H, W, hfov_rad, min_depth, max_depth, eps=0.01):
    lib = _lib("liboracle_pre.so")
    c = np.zeros(8, dtype=np.float32)
    lib.orc_topdown_consts(int(H), int(W), C.c_double(hfov_rad), C.c_double(min_depth), C.c_double(max_depth),
                           C.c_double(eps), _p(c))
    return c


def topdown_view(depth, consts, rows_around_center=50, blur_in=None, return_aux=False):
    """geometry_utils.py:516-556 for one [H,W] (or [H,W,1]) normalized depth frame -> [H,W,1] float32."""
    lib = _lib("liboracle_pre.so")
    d = np.ascontiguousarray(np.asarray(depth, dtype=np.float32).reshape(depth.shape[0], depth.shape[1]))
    H, W = d.shape
    out = np.zeros((H, W), dtype=np.float32)
    bbox = np.zeros(4, dtype=np.int32)
    blur_out = np.zeros(H * W, dtype=np.float32)
    cnt = np.zeros((H, W), dtype=np.int32)
    bi = np.ascontiguousarray(blur_in, dtype=np.float32) if blur_in is not None else None
    c = np.ascontiguousarray(consts, dtype=np.float32)
    rc = lib.orc_topdown(_p(d), H, W, _p(c), int(rows_around_center), _p(bi), _p(out), _p(bbox), _p(blur_out), _p(cnt))
    if return_aux:
        hc, wc = bbox[1] - bbox[0] + 1, bbox[3] - bbox[2] + 1
        blur = blur_out[: max(hc, 0) * max(wc, 0)].reshape(max(hc, 0), max(wc, 0)) if rc == 0 else None
        return out[..., None], dict(bbox=bbox, blur=blur, cnt=cnt, empty=bool(rc))
    return out[..., None]


# ----------------------------------------------------------------------------- baseline form: one pair per host thread
def forward_pairs_parallel(sd, obs, *, ngroups, threads=None, dtype=np.float32, indices=None):
    """The same forward, parallelised over PAIRS instead of inside each layer (frame pairs are independent; per-sample
    GroupNorm): `threads` host threads, each running whole single-pair forwards with an OpenMP team of one, so a pair's
    working set (7.9 MB of input, 2.1 MB of stem output, ...) stays in that core's caches.  ctypes releases the GIL inside
    the C calls.  Results equal `forward` (same per-element summation order).  bench.py's cpu_baseline leg.
    `indices`: the pairs of `obs` to run, repeats allowed (a long timed sample over a small set of distinct pairs)."""
    from concurrent.futures import ThreadPoolExecutor
    threads = int(threads or usable_cores())
    idx = list(range(next(iter(obs.values())).shape[0])) if indices is None else [int(i) for i in indices]
    n = len(idx)
    sdc = {k: np.ascontiguousarray(np.asarray(v)) for k, v in sd.items()}

    def init():
        set_threads(1)                                   # omp nthreads-var is per host thread: a team of one in THIS thread

    def one(i):
        return forward(sdc, {k: v[i:i + 1] for k, v in obs.items()}, ngroups=ngroups, dtype=dtype)

    with ThreadPoolExecutor(max_workers=min(threads, n), initializer=init) as ex:
        outs = list(ex.map(one, idx))
    return np.concatenate(outs, axis=0)
