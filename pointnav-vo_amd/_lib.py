"""ctypes binding of csrc/libpnvo.so — the C ABI declared in include/pnvo.h.

There is NO fallback: if the HIP library is missing or fails to load, importing this module raises.  The product
path never routes through oracle/ or any CPU implementation.
"""
import ctypes as C
import os

# torch bundles its own ROCm runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  It must be the FIRST HIP
# runtime mapped into the process, so that libpnvo.so's NEEDED libamdhip64.so.7 resolves to the copy torch uses:
# two HIP/HSA runtimes in one process do not both get the GPU ("no ROCm-capable device is detected").
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpnvo.so")

PNVO_OK = 0


GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p)


class PnvoError(RuntimeError):
    pass


class pnvo_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "n_rgb", "n_depth", "n_dd", "n_tdv", "baseplanes", "hidden", "out_dim", "normalize",
        "act_embed", "n_acts", "flat_size", "max_batch", "backbone_depth")]


class pnvo_tensor_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class pnvo_kernel_time(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/pnvo.h declares (tests/test_abi.py checks the exported set against the header)
_SIGNATURES = {
    "pnvo_create": (C.c_int, [C.POINTER(pnvo_config), C.c_int, C.POINTER(C.c_void_p)]),
    "pnvo_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(pnvo_tensor_desc), C.c_int]),
    "pnvo_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p]),
    "pnvo_forward_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "pnvo_forward_dual_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "pnvo_forward_grouped_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "pnvo_grouped_supported": (C.c_int, [C.c_void_p, C.c_int]),
    "pnvo_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "pnvo_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "pnvo_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "pnvo_forward_dual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_build_obs_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_stage_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int]),
    "pnvo_stage_frames2": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]),
    "pnvo_discretize_depth": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p]),
    "pnvo_topdown_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pnvo_topdown_view_f64": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "pnvo_half_to_float": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pnvo_dataset_pairs": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p] * 7),
    "pnvo_topdown_view_pairs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "pnvo_topdown_view": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    "pnvo_train_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(pnvo_tensor_desc), C.c_int]),
    "pnvo_train_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pnvo_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_train_set_grad_hook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_train_grad_buckets": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]),
    "pnvo_input_moments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "pnvo_rmv_merge": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_train_set_actions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pnvo_mse_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_mse_loss_coef": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_geo_inverse_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "pnvo_train_set_dropout": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64]),
    "pnvo_train_dropout_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "pnvo_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "pnvo_destroy": (C.c_int, [C.c_void_p]),
    "pnvo_ring_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_last_error": (C.c_char_p, [C.c_void_p]),
    "pnvo_last_note": (C.c_char_p, [C.c_void_p]),
    "pnvo_set_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "pnvo_tap_shape": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    "pnvo_check_inputs": (C.c_int, [C.c_void_p]),
    "pnvo_forward_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "pnvo_policy_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "pnvo_policy_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "pnvo_policy_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pnvo_policy_destroy": (C.c_int, [C.c_void_p]),
    "pnvo_avgpool2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pnvo_layer_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]),
    "pnvo_timing_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pnvo_timing_read": (C.c_int, [C.c_void_p, C.POINTER(pnvo_kernel_time), C.c_int, C.POINTER(C.c_int)]),
    "pnvo_packed_conv_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pnvo_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pnvo_version": (C.c_char_p, []),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise PnvoError(
            f"{LIB_PATH} is missing: the HIP extension has not been built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C pointnav-vo_amd/csrc`).  "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here == the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, handle=None):
    if rc != PNVO_OK:
        msg = lib.pnvo_last_error(handle)
        raise PnvoError(f"libpnvo error {rc}: {msg.decode() if msg else '?'}")


def version():
    return lib.pnvo_version().decode()
