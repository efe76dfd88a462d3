"""CPU: lane-level numpy model of conv_mfma_kernel (pointnav-vo_amd/csrc/conv_mfma.hip).

The model walks the SAME index math as the HIP kernel — wave tile decode, (tap, j, t, h) K order, the packed
weight layout produced by the library's own pnvo_pack_conv_weight (host code, no GPU needed), the 32x32x2 MFMA
fragment layouts documented in /opt/skills/guides/cdna_hip_programming.md §3, the epilogue row mapping and the
deterministic GroupNorm partial-sum slots — and checks the result against the oracle's conv + GroupNorm.  It catches
layout/indexing mistakes without a GPU; the `-m gpu` tests then check the real kernel.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from pointnav_vo_amd import _lib, synth


def pack(w):
    cout, cin, kh, kw = w.shape
    n = _lib.lib.pnvo_packed_conv_floats(cout, cin, kh, kw)
    out = np.zeros(n, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    _lib.check(_lib.lib.pnvo_pack_conv_weight(w.ctypes.data_as(C.c_void_p), cout, cin, kh, kw,
                                              out.ctypes.data_as(C.c_void_p)))
    return out.reshape(-1, 4)   # float4 view


def emulate_conv(x, wpk, cout, KH, KW, stride, pad, MT, NT, in_ss=None):
    """x: [B,H,W,CINP] float32 (channel padded to 8).  Returns raw y [B,Ho,Wo,COUTP] and stats [B,slots,COUTP,2]."""
    B, H, W, CIN = x.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    P, M = Ho * Wo, B * Ho * Wo
    COUTP = (cout + 31) // 32 * 32
    J, T, WM = CIN // 8, KH * KW, MT * 32
    slots = (P + WM - 1) // WM + 1
    y = np.zeros((M, COUTP), np.float32)
    stats = np.full((B, slots, COUTP, 2), np.nan, np.float32)
    xf = x.reshape(-1)
    lane = np.arange(64)
    i, h = lane & 31, lane >> 5
    for ntg0 in range(0, COUTP // 32, NT):
        for m_base in range(0, M, WM):
            acc = np.zeros((MT, NT, 32, 32), np.float64)   # D[row][col]; fp64 accumulate (checker, not bit model)
            n0 = m_base // P
            m = np.minimum(m_base + np.arange(MT)[:, None] * 32 + i[None, :], M - 1)      # [MT,64]
            vm = (m_base + np.arange(MT)[:, None] * 32 + i[None, :]) < M
            n = m // P
            rem = m - n * P
            ho, wo = rem // Wo, rem % Wo
            hi0, wi0 = ho * stride - pad, wo * stride - pad
            for kh in range(KH):
                for kw in range(KW):
                    tap = kh * KW + kw
                    hi, wi = hi0 + kh, wi0 + kw
                    ok = vm & (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W)
                    base = ((n * H + np.clip(hi, 0, H - 1)) * W + np.clip(wi, 0, W - 1)) * CIN + 4 * h[None, :]
                    for j in range(J):
                        a = np.zeros((MT, 64, 4), np.float32)
                        for t in range(4):
                            a[..., t] = np.where(ok, xf[base + 8 * j + t], 0.0)
                        if in_ss is not None:
                            c = 8 * j + 4 * h
                            for t in range(4):
                                sc, sh = in_ss[0][n, (c + t)[None, :]], in_ss[1][n, (c + t)[None, :]]
                                a[..., t] = np.where(ok, np.maximum(a[..., t] * sc + sh, 0.0), 0.0)
                        for nt in range(NT):
                            b = wpk[(((ntg0 + nt) * T + tap) * J + j) * 64 + lane]       # [64,4]
                            for t in range(4):
                                for mt in range(MT):
                                    A = np.stack([a[mt, :32, t], a[mt, 32:, t]], axis=1)      # A[i][k] = lane i+32k
                                    Bm = np.stack([b[:32, t], b[32:, t]], axis=0)             # B[k][j] = lane j+32k
                                    acc[mt, nt] += A.astype(np.float64) @ Bm.astype(np.float64)
            # epilogue: lane/register view of D, exactly as the kernel indexes it
            last = min(m_base + WM - 1, M - 1)
            for nt in range(NT):
                for mt in range(MT):
                    for r in range(16):
                        for hh in range(2):
                            row = (r & 3) + 8 * (r >> 2) + 4 * hh
                            mm = m_base + mt * 32 + row
                            if mm < M:
                                y[mm, (ntg0 + nt) * 32:(ntg0 + nt) * 32 + 32] = acc[mt, nt, row, :]
                for nn in range(n0, last // P + 1):
                    lo, hi_ = nn * P, min((nn + 1) * P, M)
                    slot = m_base // WM - lo // WM
                    rows = np.arange(m_base, m_base + WM)
                    sel = (rows >= lo) & (rows < hi_)
                    blk = acc[:, nt].reshape(WM, 32)[sel]
                    co = (ntg0 + nt) * 32
                    stats[nn, slot, co:co + 32, 0] = blk.sum(0)
                    stats[nn, slot, co:co + 32, 1] = (blk ** 2).sum(0)
    return y.reshape(B, Ho, Wo, COUTP), stats, slots, WM


def finalize(stats, B, P, WM, C, G):
    """gn_finalize_kernel's slot enumeration: slots (n*P)//WM .. ((n+1)*P-1)//WM of sample n."""
    cpg = C // G
    mu, var = np.zeros((B, G)), np.zeros((B, G))
    for n in range(B):
        ns = ((n + 1) * P - 1) // WM - (n * P) // WM + 1
        s = stats[n, :ns].astype(np.float64)
        assert not np.isnan(s[:, :C]).any(), "a slot the finaliser reads was never written"
        for g in range(G):
            s1 = s[:, g * cpg:(g + 1) * cpg, 0].sum()
            s2 = s[:, g * cpg:(g + 1) * cpg, 1].sum()
            mu[n, g] = s1 / (P * cpg)
            var[n, g] = s2 / (P * cpg) - mu[n, g] ** 2
    return mu, var


CASES = [
    # B, H, W, Cin, Cout, K, stride, pad, MT, NT
    (2, 9, 11, 6, 32, 7, 2, 3, 2, 1),      # stem-like, Cin padded 6->8, tiles straddle samples
    (3, 7, 5, 32, 64, 3, 1, 1, 1, 2),      # 3x3 s1, two n-tiles per wave
    (2, 8, 9, 32, 64, 3, 2, 1, 4, 1),      # 3x3 s2, P=20 < wave tile: several samples per wave
    (2, 6, 7, 64, 31, 1, 2, 0, 1, 1),      # 1x1 s2 downsample-like, Cout padded 31->32
    (5, 2, 3, 16, 128, 3, 1, 1, 1, 4),     # tiny maps (final stages of the small test nets)
]


@pytest.mark.parametrize("case", CASES)
def test_conv_kernel_model_matches_oracle(case):
    B, H, W, Cin, Cout, K, stride, pad, MT, NT = case
    cinp = (Cin + 7) // 8 * 8
    x = synth.uniform(3, f"x{case}", (B, H, W, Cin), -1, 1).astype(np.float32)
    w = synth.uniform(3, f"w{case}", (Cout, Cin, K, K), -1, 1).astype(np.float32)
    xp = np.zeros((B, H, W, cinp), np.float32)
    xp[..., :Cin] = x
    y, stats, slots, WM = emulate_conv(xp, pack(w), Cout, K, K, stride, pad, MT, NT)
    ref = oracle.conv2d(x.astype(np.float64), w.astype(np.float64), stride, pad)
    np.testing.assert_allclose(y[..., :Cout], ref, rtol=1e-6, atol=1e-6)
    assert np.all(y[..., Cout:] == 0)
    # GroupNorm statistics through the slot mechanism == direct statistics of the conv output
    G = 1 if Cout == 31 else 16
    P = ref.shape[1] * ref.shape[2]
    mu, var = finalize(stats, B, P, WM, Cout, G)
    r = ref.reshape(B, P, G, Cout // G)
    np.testing.assert_allclose(mu, r.mean(axis=(1, 3)), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var, r.var(axis=(1, 3)), rtol=1e-4, atol=1e-6)


def test_conv_kernel_model_with_input_transform():
    """Producer GN+ReLU applied on the fly: padding must be zero AFTER the transform."""
    B, H, W, Cin, Cout = 2, 6, 5, 32, 32
    x = synth.uniform(4, "xt", (B, H, W, Cin), -1, 1).astype(np.float32)
    w = synth.uniform(4, "wt", (Cout, Cin, 3, 3), -1, 1).astype(np.float32)
    sc = synth.uniform(4, "sc", (B, Cin), -2, 2).astype(np.float32)
    sh = synth.uniform(4, "sh", (B, Cin), -1, 1).astype(np.float32)
    y, _, _, _ = emulate_conv(x, pack(w), Cout, 3, 3, 1, 1, 2, 1, in_ss=(sc, sh))
    xn = np.maximum(x * sc[:, None, None, :] + sh[:, None, None, :], 0).astype(np.float64)
    ref = oracle.conv2d(xn, w.astype(np.float64), 1, 1)
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)
