#!/usr/bin/env python
"""Golden fixtures for the dataset input pipeline, build container only.

    python tests/golden/gen_golden_dataset.py        # writes tests/golden/dataset_*.npz

Runs the reference's own StatePairRegressionDataset._process_data + normal_collate_func
(/root/reference/pointnav_vo/vo/dataset/regression_geo_invariance_iter_dataset.py:205-560, unmodified, on an instance
created without __init__ because __init__ opens an HDF5 file) on the seeded chunk of pointnav_vo_amd.synth.make_dataset_chunk
and stores its OUTPUTS (inputs are regenerated from the seed by the tests).

Stand-ins for packages that are not installed here (same policy as gen_golden.py): h5py (unused by _process_data),
cv2.GaussianBlur (gen_golden.blur_stub), `np.int` (removed from numpy >= 1.24; the reference's geometry_utils.py:448 uses
it), and the quaternion helpers of habitat-lab / numpy-quaternion (`quaternion_from_coeff`, `agent_state_target2ref`,
`quaternion_to_list`), written below from their published definitions -> the swapped entries' targets are "parity unpinned".
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
from pointnav_vo_amd import synth  # noqa: E402


class Quat:
    """Minimal stand-in for np.quaternion (w, x, y, z)."""

    def __init__(self, w, x, y, z):
        self.w, self.x, self.y, self.z = float(w), float(x), float(y), float(z)

    def __mul__(self, o):
        return Quat(self.w * o.w - self.x * o.x - self.y * o.y - self.z * o.z,
                    self.w * o.x + self.x * o.w + self.y * o.z - self.z * o.y,
                    self.w * o.y - self.x * o.z + self.y * o.w + self.z * o.x,
                    self.w * o.z + self.x * o.y - self.y * o.x + self.z * o.w)

    def norm2(self):
        return self.w ** 2 + self.x ** 2 + self.y ** 2 + self.z ** 2

    def inverse(self):
        n = self.norm2()
        return Quat(self.w / n, -self.x / n, -self.y / n, -self.z / n)

    def normalized(self):
        n = np.sqrt(self.norm2())
        return Quat(self.w / n, self.x / n, self.y / n, self.z / n)


def quaternion_from_coeff(c):            # habitat: coeffs in [x, y, z, w]
    return Quat(c[3], c[0], c[1], c[2])


def quaternion_to_list(q):
    return [q.x, q.y, q.z, q.w]


def quaternion_rotate_vector(q, v):
    r = q * Quat(0, v[0], v[1], v[2]) * q.inverse()
    return np.array([r.x, r.y, r.z])


def agent_state_target2ref(ref_state, target_state):
    ref_rot, ref_pos = ref_state
    tgt_rot, tgt_pos = target_state
    if not isinstance(ref_rot, Quat):
        ref_rot = quaternion_from_coeff(ref_rot)
    ref_rot = ref_rot.normalized()
    if not isinstance(tgt_rot, Quat):
        tgt_rot = quaternion_from_coeff(tgt_rot)
    tgt_rot = tgt_rot.normalized()
    return ref_rot.inverse() * tgt_rot, quaternion_rotate_vector(ref_rot.inverse(), tgt_pos - ref_pos)


def import_dataset():
    registry, geo = G.import_reference()
    np.int = int                                               # removed alias used at geometry_utils.py:448
    hg = sys.modules["habitat.utils.geometry_utils"]
    hg.quaternion_from_coeff = quaternion_from_coeff
    hg.agent_state_target2ref = agent_state_target2ref
    geo.quaternion_to_list = quaternion_to_list
    sys.modules["h5py"] = types.ModuleType("h5py")
    for n, pth in [("pointnav_vo.vo.common", "/pointnav_vo/vo/common"), ("pointnav_vo.vo.dataset", "/pointnav_vo/vo/dataset")]:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = [G.REF + pth]
            sys.modules[n] = m
    return importlib.import_module("pointnav_vo.vo.dataset.regression_geo_invariance_iter_dataset")


def fixture(mod, fname, N, W, H, seed, act_type, geo_types, bins, tdv):
    ds = object.__new__(mod.StatePairRegressionDataset)
    infos = dict(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=np.deg2rad(70.0), ksize=3,
                 rows_around_center=50 if H > 100 else 9, flag_center_crop=True)
    ds._vis_size_w, ds._vis_size_h = W, H
    ds._act_type, ds._geo_invariance_types = act_type, list(geo_types)
    ds._discretize_depth = "hard" if bins else "none"
    ds._discretized_depth_channels = bins
    ds._discretized_depth_end_vals = [i * 1.0 / bins for i in range(bins)] + [1.0] if bins else []
    ds._gen_top_down_view, ds._top_down_view_infos = bool(tdv), infos
    ch = synth.make_dataset_chunk(N, H, W, seed=seed, bins=max(bins, 1))
    ds._actions = ch["actions"]
    for k in ("prev_rgbs", "cur_rgbs", "prev_depths", "cur_depths", "delta_positions", "delta_rotations",
              "prev_global_positions", "prev_global_rotations", "cur_global_positions", "cur_global_rotations"):
        setattr(ds, "_" + k, ch[k])
    # valid samples as _get_valid_idxes (:170-203) would pick them from the HDF5 group
    a = ch["actions"]
    lr = np.nonzero((a == 2) | (a == 3))[0]
    if isinstance(act_type, int):
        idxs = np.arange(N) if act_type == -1 else (lr if "inverse_data_augment_only" in geo_types else np.nonzero(a == act_type)[0])
    else:
        idxs = lr
    batch = mod.normal_collate_func([ds._process_data(7, int(i)) for i in idxs])
    (dt, rgb, depth, dd, tdvp, acts, dx, dy, dz, dyaw, dzm, cidx, eidx) = batch
    rec = dict(N=N, W=W, H=H, seed=seed, act_type=np.array(act_type), geo=",".join(geo_types), bins=bins, has_tdv=int(bool(tdv)),
               rows_around_center=infos["rows_around_center"], data_types=dt.numpy(), actions=acts.numpy(),
               targets=torch.cat([dx, dy, dz, dyaw], 1).numpy(), dz_masks=dzm.numpy(), entry_idxs=eidx.numpy(),
               chunk_idxs=cidx.numpy(), rgb_sum=rgb.double().sum(dim=(1, 2)).numpy(), depth_sum=depth.double().sum(dim=(1, 2)).numpy(),
               dd_bin=(dd.reshape(*dd.shape[:3], 2, -1).argmax(-1).to(torch.uint8).numpy() if bins else np.zeros(1)),
               dd_sum=dd.double().sum().item(), tdv_pairs=tdvp.numpy().astype(np.float32))
    assert rgb.dtype == torch.uint8 and depth.dtype == torch.float32 and tdvp.dtype == torch.float32
    np.savez_compressed(os.path.join(HERE, fname), **rec)
    print(fname, "entries", dt.shape[0], "tdv nonzero", int((tdvp > 0).sum()))


def main():
    mod = import_dataset()
    fixture(mod, "dataset_64x48_joint.npz", 10, 64, 48, 51, [2, 3], ["inverse_joint_train"], 10, True)
    fixture(mod, "dataset_70x40_all.npz", 6, 70, 40, 52, -1, [], 10, True)
    fixture(mod, "dataset_341x192_left_aug.npz", 4, 341, 192, 53, 2, ["inverse_data_augment_only"], 10, True)
    fixture(mod, "dataset_64x48_rgbd_fwd.npz", 8, 64, 48, 54, 1, [], 0, False)


if __name__ == "__main__":
    main()
