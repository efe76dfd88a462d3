"""VO dataset input pipeline on the MI355X.

The reference prepares every training sample on CPU workers (20 of them, vo_cnn_regression_geo_invariance_engine.py:32):
`StatePairRegressionDataset._process_data` (/root/reference/pointnav_vo/vo/dataset/regression_geo_invariance_iter_dataset.py:
205-454) reshapes the uint8 RGB / float16 depth vectors of one HDF5 entry, builds the one-hot depth with numpy
(regression_iter_dataset.py:32-69), runs the *numpy* top-down-view generator (geometry_utils.py:275-470) on both frames, and
appends the (prev, cur) entry and — for the geometric-inversion modes — the swapped (cur, prev) entry with its own
regression target (:342-420); `normal_collate_func` (:524-560) concatenates the samples.

`StatePairBatcher.process_chunk` does the same for a whole chunk at once on the device: one H2D copy of the stored
uint8 / float16 arrays, the top-down view of all 2N frames in one batched launch (float64 projection = the numpy twin's
arithmetic), one assembly kernel that writes the four float32 NHWC pair tensors for every entry.  Reading the HDF5 file is
left to the caller (h5py is not part of this image); a chunk is the dict of arrays `f[chunk_k][name][()]` yields (:494-509).

The small per-sample arithmetic (regression targets from stored poses) stays on the host in float64 numpy as in the
reference.  The swapped entries' targets go through habitat-lab's `agent_state_target2ref` and the `quaternion` package
in the reference; both are absent from /root/reference and this image, so they are restated from their published
definitions (PARITY UNPINNED for those 4 numbers per swapped entry; see oracle/dataset_oracle.py).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .geometry import _quat_mul, _quat_rotate

CUR_REL_TO_PREV, PREV_REL_TO_CUR = 0, 1            # vo/common/common_vars.py
MOVE_FORWARD, TURN_LEFT, TURN_RIGHT = 1, 2, 3


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class NormalizedDepth2TopDownViewHabitat:
    """The numpy twin's constructor and results (geometry_utils.py:275-470) with the work on the GPU, batched."""

    def __init__(self, min_depth, max_depth, vis_size_h, vis_size_w, hfov_rad, ksize=3, rows_around_center=50,
                 flag_center_crop=True):
        if ksize != 3 or not flag_center_crop:
            raise NotImplementedError("only ksize=3 / flag_center_crop=True (the dataset's call site) is built")
        if vis_size_w >= 1024:
            raise NotImplementedError("pixel centres are float16 in the reference (:392-397): exact only below 1024")
        self._epsilon = 0.01
        self._min_depth, self._max_depth = min_depth, max_depth
        self._vis_size_h, self._vis_size_w = vis_size_h, vis_size_w
        self._rows_around_center = rows_around_center
        f = (vis_size_w / 2) / (np.tan(hfov_rad / 2))                                   # :341-347
        self._K = np.array([[f, 0, vis_size_w / 2], [0, f, vis_size_h / 2], [0, 0, 1.0]])
        kinv = np.linalg.inv(self._K)
        coords = np.matmul(kinv, (vis_size_w - 0.5, 0, 1)) * max_depth                  # _get_x_range :349-353
        min_x, max_x = -coords[0], coords[0]
        x_den = (max_x - min_x) * (1 + self._epsilon)                                   # :432
        z_den = (max_depth - min_depth) * (1 + self._epsilon)                           # :433-435
        assert kinv[0, 1] == 0.0
        self._consts = (C.c_double * 8)(kinv[0, 0], kinv[0, 2], min_x, x_den, max_depth - min_depth, z_den, min_depth, 0.0)
        self._work = None

    def gen_top_down_view_batch(self, depth):
        """depth: CUDA float32 [N,H,W] contiguous -> [N,H,W] float32."""
        H, W = self._vis_size_h, self._vis_size_w
        assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous() and depth.shape[1:] == (H, W)
        n, dev = depth.shape[0], depth.device
        need = _lib.lib.pnvo_topdown_workspace_bytes(int(n), H, W)
        if self._work is None or self._work.numel() < need or self._work.device != dev:
            self._work = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((n, H, W), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.pnvo_topdown_view_f64(_ptr(depth), int(n), H, W, H * W, 1, self._consts,
                                                      int(self._rows_around_center), _ptr(out), H * W, 1, _ptr(self._work),
                                                      _stream(dev)))
        return out

    def gen_top_down_view(self, normalized_depth):
        """[H,W,1] array/tensor -> [H,W,1] (as the reference method; float32 on the device)."""
        d = torch.as_tensor(np.asarray(normalized_depth, dtype=np.float32)[..., 0]) if not torch.is_tensor(normalized_depth) \
            else normalized_depth[..., 0].float()
        dev = d.device if d.is_cuda else torch.device("cuda", torch.cuda.current_device())
        return self.gen_top_down_view_batch(d.to(dev).contiguous().unsqueeze(0))[0].unsqueeze(-1)


def state_target2ref(ref_rot, ref_pos, tgt_rot, tgt_pos):
    """habitat-lab's agent_state_target2ref (habitat/utils/geometry_utils.py) for [x,y,z,w] coefficient arrays: both
    rotations normalised, rotation = ref^-1 * target, position = ref^-1 (target_pos - ref_pos) ref."""
    r = np.asarray(ref_rot, dtype=np.float64)
    t = np.asarray(tgt_rot, dtype=np.float64)
    r = r / np.linalg.norm(r)
    t = t / np.linalg.norm(t)
    rinv = np.array([-r[0], -r[1], -r[2], r[3]])
    diff = np.asarray(tgt_pos) - np.asarray(ref_pos)          # in the stored dtype (float16 in the HDF5 files), as the reference
    return _quat_mul(rinv, t), _quat_rotate(rinv, diff.astype(np.float64))


def entries_of_chunk(actions, act_type=-1, geo_invariance_types=(), idxs=None):
    """(src, swap, action, data_type) of every entry _process_data emits for the samples `idxs` of a chunk, in order
    (:297-420; valid samples as _get_valid_idxes :170-203)."""
    actions = np.asarray(actions).reshape(-1).astype(np.int64)
    geo = tuple(geo_invariance_types)
    lr = (actions == TURN_LEFT) | (actions == TURN_RIGHT)
    if idxs is None:
        if isinstance(act_type, int):
            if act_type == -1:
                idxs = np.arange(actions.size)
            elif "inverse_data_augment_only" in geo:
                idxs = np.nonzero(lr)[0]
            else:
                idxs = np.nonzero(actions == act_type)[0]
        else:
            assert set(act_type) == {TURN_LEFT, TURN_RIGHT}
            idxs = np.nonzero(lr)[0]
    out = []
    for i in np.asarray(idxs).reshape(-1).tolist():
        a = int(actions[i])
        if act_type == -1 or (isinstance(act_type, int) and a == act_type) or "inverse_joint_train" in geo:
            out.append((i, 0, a, CUR_REL_TO_PREV))
        flag1 = act_type != -1 and "inverse_data_augment_only" in geo and a != MOVE_FORWARD and a != act_type
        flag2 = act_type != -1 and a != MOVE_FORWARD and "inverse_joint_train" in geo
        if flag1 or flag2:
            out.append((i, 1, TURN_LEFT if a == TURN_RIGHT else TURN_RIGHT, PREV_REL_TO_CUR))
    return out


class StatePairBatcher:
    """Device-side `_process_data` + collate for one chunk.  Constructor arguments as StatePairRegressionDataset (:37-52)."""

    def __init__(self, vis_size_w, vis_size_h, act_type=-1, discretize_depth="none", discretized_depth_channels=0,
                 gen_top_down_view=False, top_down_view_infos=None, geo_invariance_types=(), device="cuda:0"):
        self.W, self.H = int(vis_size_w), int(vis_size_h)
        self.act_type = act_type
        self.geo = tuple(geo_invariance_types)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("StatePairBatcher runs on an MI355X only")
        self.bins = int(discretized_depth_channels) if discretize_depth == "hard" else 0
        # the dataset compares float16 depth with python-float edges: numpy casts the scalar to float16 (:47-63)
        self.edges = None
        if self.bins:
            e = [float(np.float16(i * 1.0 / self.bins)) for i in range(self.bins)] + [1.0]
            self.edges = (C.c_float * (self.bins + 1))(*e)
        self.tdv = NormalizedDepth2TopDownViewHabitat(**top_down_view_infos) if gen_top_down_view else None
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.dev)

    def process_chunk(self, chunk, idxs=None, chunk_i=0):
        """chunk: dict with the HDF5 datasets of one chunk (numpy): actions [N], prev_rgbs / cur_rgbs [N, H*W*3] uint8,
        prev_depths / cur_depths [N, H*W] float16, delta_positions [N,3], delta_rotations [N,4], and for the inversion modes
        prev/cur_global_positions [N,3], prev/cur_global_rotations [N,4].  Returns the reference's batch tuple as a dict
        of tensors (observations on the device, float32 NHWC; bookkeeping on the host)."""
        H, W, dev = self.H, self.W, self.dev
        N = int(np.asarray(chunk["actions"]).shape[0])
        ent = entries_of_chunk(chunk["actions"], self.act_type, self.geo, idxs)
        M = len(ent)
        src = np.array([e[0] for e in ent], dtype=np.int32)
        swap = np.array([e[1] for e in ent], dtype=np.int32)
        # targets (:264-287 and :388-420), host float64 -> FloatTensor
        dpos = np.asarray(chunk["delta_positions"])                    # stored dtype (float16): arctan2 below runs in it
        drot = np.asarray(chunk["delta_rotations"])
        tgt = np.zeros((M, 4), dtype=np.float32)                       # dx, dy, dz, dyaw
        for m, (i, sw, _a, _t) in enumerate(ent):
            if not sw:
                tgt[m, :3] = dpos[i]
                tgt[m, 3] = 2 * np.arctan2(drot[i, 1], drot[i, 3])
            else:
                q, p = state_target2ref(chunk["cur_global_rotations"][i], chunk["cur_global_positions"][i],
                                        chunk["prev_global_rotations"][i], chunk["prev_global_positions"][i])
                # the reference returns the rotation through quaternion_to_array -> float32 coefficients (:400-403)
                q32 = q.astype(np.float32)
                tgt[m, :3] = p
                tgt[m, 3] = 2 * np.arctan2(q32[1], q32[3])
        with torch.cuda.device(dev), torch.no_grad():
            st = _stream(dev)
            up = lambda k, dt: torch.from_numpy(np.ascontiguousarray(chunk[k]).view(dt)).to(dev, non_blocking=True)
            prgb, crgb = up("prev_rgbs", np.uint8), up("cur_rgbs", np.uint8)
            pd, cd = up("prev_depths", np.int16), up("cur_depths", np.int16)         # float16 bit patterns
            assert prgb.shape == (N, H * W * 3) and pd.shape == (N, H * W), "chunk arrays do not match vis_size"
            dsrc, dswap = torch.from_numpy(src).to(dev), torch.from_numpy(swap).to(dev)
            frames = None
            if self.tdv is not None:
                d32 = torch.empty((2 * N, H, W), device=dev, dtype=torch.float32)
                _lib.check(_lib.lib.pnvo_half_to_float(_ptr(pd), N * H * W, _ptr(d32), st))
                _lib.check(_lib.lib.pnvo_half_to_float(_ptr(cd), N * H * W, C.c_void_p(d32.data_ptr() + 4 * N * H * W), st))
                frames = self.tdv.gen_top_down_view_batch(d32)
            new = lambda c: torch.empty((M, H, W, c), device=dev, dtype=torch.float32)
            rgb, depth = new(6), new(2)
            dd = new(2 * self.bins) if self.bins else torch.zeros((M, H, W, 2), device=dev)   # :238-239 zeros when "none"
            tdv = new(2)
            self._flag.zero_()
            _lib.check(_lib.lib.pnvo_dataset_pairs(_ptr(prgb), _ptr(crgb), _ptr(pd), _ptr(cd), _ptr(frames), _ptr(dsrc),
                                                   _ptr(dswap), N, M, H, W, self.bins, self.edges, _ptr(rgb), _ptr(depth),
                                                   _ptr(dd) if self.bins else None, _ptr(tdv), _ptr(self._flag), st))
            if int(self._flag.item()):
                raise AssertionError("depth outside [0, 1] (regression_iter_dataset.py:33-34)")
        t = torch.from_numpy(tgt)
        return dict(data_types=torch.tensor([e[3] for e in ent], dtype=torch.float32).unsqueeze(1),
                    rgb_pairs=rgb, depth_pairs=depth, discretized_depth_pairs=dd, top_down_view_pairs=tdv,
                    actions=torch.tensor([e[2] for e in ent], dtype=torch.int64).unsqueeze(1),
                    delta_xs=t[:, 0:1], delta_ys=t[:, 1:2], delta_zs=t[:, 2:3], delta_yaws=t[:, 3:4],
                    dz_regress_masks=torch.ones((M, 1)), chunk_idxs=torch.full((M, 1), float(chunk_i)),
                    entry_idxs=torch.from_numpy(src.astype(np.float32)).unsqueeze(1))
