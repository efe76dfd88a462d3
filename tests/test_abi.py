"""CPU: the C-ABI library loads and exports exactly the symbols include/pnvo.h declares; host-only entry points
work without a GPU and device entry points fail with an error code (never crash, never fall back)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from conftest import ROOT
from pointnav_vo_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "pnvo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnvo_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    declared = header_symbols()
    assert len(declared) >= 15
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert sorted(_lib._SIGNATURES) == declared          # the ctypes binding covers the whole header


def test_pack_conv_weight_host_only():
    w = np.arange(32 * 8 * 9, dtype=np.float32).reshape(32, 8, 3, 3)
    n = _lib.lib.pnvo_packed_conv_floats(32, 8, 3, 3)
    assert n == 32 * 8 * 9
    out = np.zeros(n, np.float32)
    _lib.check(_lib.lib.pnvo_pack_conv_weight(w.ctypes.data_as(C.c_void_p), 32, 8, 3, 3, out.ctypes.data_as(C.c_void_p)))
    assert sorted(out.tolist()) == sorted(w.reshape(-1).tolist())     # a permutation of the weights
    # float4 #lane of (tap 0, j 0): lane = h*32 + n holds W[n][4h..4h+3][0][0]
    np.testing.assert_array_equal(out[:4], w[0, 0:4, 0, 0])
    np.testing.assert_array_equal(out[(32 + 5) * 4:(32 + 5) * 4 + 4], w[5, 4:8, 0, 0])


def test_bad_arguments_return_error_codes():
    cfg = _lib.pnvo_config(width=341, height=192, n_rgb=0, n_depth=0, n_dd=0, n_tdv=0, baseplanes=32, hidden=512,
                           out_dim=3, normalize=1, act_embed=0, n_acts=4, flat_size=2048, max_batch=0)
    h = C.c_void_p()
    rc = _lib.lib.pnvo_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -1 and b"blind" in _lib.lib.pnvo_last_error(None)      # vo_cnn.py:67-68
    assert _lib.lib.pnvo_forward(None, None, None, None, None, None, 1, None, None) == -1
    assert _lib.lib.pnvo_destroy(None) == 0
    assert _lib.version().startswith("pnvo")


def test_loaded_library_was_built_from_this_tree():
    """pnvo_version() carries sha256[:16] of the sources the library was built from (csrc/Makefile: src_hash.h); recomputed here
    over the same files in the same order — on the GPU box this proves the .so the tests load is the one HEAD's sources build."""
    import glob
    import hashlib
    csrc = os.path.dirname(_lib.LIB_PATH)
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))
                   if os.path.basename(f) != "src_hash.h")
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(csrc, "..", "..", "include", "pnvo.h"), "rb").read())
    assert _lib.version().endswith("src:" + h.hexdigest()[:16]), (_lib.version(), h.hexdigest()[:16])


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpu_box_loads_the_library_of_this_tree():
    """The same check inside the `-m gpu` run: the round-end GPU record then proves which sources the loaded libpnvo.so came from."""
    test_loaded_library_was_built_from_this_tree()
