"""Reading the reference's VO checkpoints without executing what is pickled inside them.

The reference writes two layouts with plain ``torch.save``:

  {"model_state": state_dict, ...}                                  single action model
      (loaded at /root/reference/pointnav_vo/rl/common/base_trainer_with_vo.py:92-93)
  {"epoch", "config": <yacs CfgNode>, "model_states": {act: state_dict}, "optim_states": {act: ...},
   "rnd_state": random.getstate(), "np_rnd_state": np.random.get_state(), "torch_rnd_state", "torch_cuda_rnd_state"}
      (written by vo/engine/vo_cnn_regression_geo_invariance_engine.py:1425-1436, loaded at base_trainer_with_vo.py:94-97)

and reads them with ``torch.load(path)`` — a full unpickle.  On torch >= 2.6 the default ``weights_only=True`` refuses those
files (a config object, numpy arrays and RNG tuples are not tensors), and ``weights_only=False`` would run whatever the file
says.  ``load_vo_checkpoint`` unpickles with an allow-list instead: tensors, their storages and plain containers are
rebuilt; every other global the pickle names (yacs / habitat config classes, numpy reconstructors, anything else — whether
or not it is importable here) becomes an inert placeholder that swallows its arguments and state.  Nothing from the file
is ever called except torch's own tensor-rebuild functions.
"""
import collections
import pickle

import torch

_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "tuple"): tuple, ("builtins", "set"): set,
    ("builtins", "frozenset"): frozenset, ("builtins", "int"): int, ("builtins", "float"): float, ("builtins", "bool"): bool,
    ("builtins", "str"): str, ("builtins", "bytes"): bytes, ("builtins", "complex"): complex,
    ("torch._utils", "_rebuild_tensor_v2"): torch._utils._rebuild_tensor_v2,
    ("torch._utils", "_rebuild_tensor"): torch._utils._rebuild_tensor,
    ("torch._utils", "_rebuild_parameter"): torch._utils._rebuild_parameter,
    ("torch", "Size"): torch.Size,
    ("torch", "device"): torch.device,
    ("torch.serialization", "_get_layout"): torch.serialization._get_layout,
}
for _n in ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool"):
    _ALLOWED[("torch", _n)] = getattr(torch, _n)
for _n in ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
           "CharStorage", "ByteStorage", "BoolStorage"):
    _ALLOWED[("torch", _n)] = getattr(torch, _n)


class Dropped:
    """Placeholder for anything in a checkpoint that is not a tensor or a plain container.  Accepts every way the pickle
    machine can build or fill an object (REDUCE / NEWOBJ arguments, BUILD state, SETITEM(S), APPEND(S), ADDITEMS)."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return Dropped()

    def __setstate__(self, state):
        pass

    def __setitem__(self, key, value):
        pass

    def append(self, item):
        pass

    def extend(self, items):
        pass

    def add(self, item):
        pass

    def __repr__(self):
        return "<dropped by pointnav_vo_amd.checkpoint>"


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return _ALLOWED[(module, name)]
        except KeyError:
            return Dropped


class _RestrictedPickle:
    """The `pickle_module` interface torch.load drives (Unpickler for both formats; load for the legacy header fields)."""
    __name__ = "pointnav_vo_amd.checkpoint"
    Unpickler = RestrictedUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kwargs):
        return RestrictedUnpickler(f, **kwargs).load()


def _only_tensors(sd, where):
    if not isinstance(sd, dict):
        raise ValueError(f"{where}: expected a state_dict, found {type(sd).__name__}")
    out = collections.OrderedDict()
    for k, v in sd.items():
        if isinstance(k, str) and isinstance(v, torch.Tensor):
            out[k] = v
        elif not isinstance(v, Dropped):
            raise ValueError(f"{where}: entry {k!r} is a {type(v).__name__}, not a tensor")
    return out


def load_vo_checkpoint(path, map_location="cpu"):
    """-> {"model_state": OrderedDict} or {"model_states": {act: OrderedDict}} (whichever the file holds; "epoch" is kept when
    it is an int).  Tensors only; see the module docstring for what is dropped and why."""
    raw = torch.load(path, map_location=map_location, pickle_module=_RestrictedPickle, weights_only=False)
    if not isinstance(raw, dict):
        raise ValueError(f"{path}: not a checkpoint dictionary")
    out = {}
    if "model_state" in raw:
        out["model_state"] = _only_tensors(raw["model_state"], f"{path}[model_state]")
    if "model_states" in raw:
        ms = raw["model_states"]
        if not isinstance(ms, dict):
            raise ValueError(f"{path}[model_states]: expected a dictionary of state_dicts")
        out["model_states"] = {k: _only_tensors(v, f"{path}[model_states][{k!r}]") for k, v in ms.items()}
    if isinstance(raw.get("epoch"), int):
        out["epoch"] = raw["epoch"]
    return out
