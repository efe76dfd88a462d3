"""Host-side mirror of the reference's VO drop-in boundary.

  BaseRLTrainerWithVO                       /root/reference/pointnav_vo/rl/common/base_trainer_with_vo.py:23-314
  NormalizedDepth2TopDownViewHabitatTorch   /root/reference/pointnav_vo/utils/geometry_utils.py:491-721

Same method names, argument meaning, return values and error behaviour; the compute (one-hot depth, ego top-down
view, network forward) runs in libpnvo.so's HIP kernels, batched, with no host synchronisation except the final
device->host copy of the 3 output floats that the reference also performs (:292).
``compute_local_delta_states_batch`` is the additive batched sibling: one call per simulator
step for all environments instead of the reference's per-env Python loop (rl/ppo/ppo_trainer.py:724-841).
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .checkpoint import load_vo_checkpoint
from .common_vars import ACT_IDX2NAME, ACT_NAME2IDX
from .registry import baseline_registry
from . import vo_cnn  # noqa: F401  (registers the models)


class AttrDict(dict):
    """Minimal stand-in for habitat.Config / yacs CfgNode attribute access (used when habitat is absent)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
            self[k] = v                  # keep the wrapped node: attribute assignment on it must persist
        return v

    __setattr__ = dict.__setitem__


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class NormalizedDepth2TopDownViewHabitatTorch:
    """Ego top-down occupancy view from a normalized depth frame; constructor as geometry_utils.py:492-516."""

    def __init__(self, min_depth, max_depth, vis_size_h, vis_size_w, hfov_rad, ksize=3, rows_around_center=50,
                 flag_center_crop=True):
        if ksize != 3 or not flag_center_crop:
            raise NotImplementedError("only ksize=3 / flag_center_crop=True (the reference's call site, "
                                      "base_trainer_with_vo.py:119-129) is built")
        self._epsilon = 0.01
        self._min_depth, self._max_depth = min_depth, max_depth
        self._vis_size_h, self._vis_size_w = vis_size_h, vis_size_w
        self._hfov_rad = hfov_rad
        self._rows_around_center = rows_around_center
        # geometry_utils.py:562-580 and :676-681, evaluated with the same torch float32 ops as the reference
        f = (vis_size_w / 2) / (np.tan(hfov_rad / 2))
        self._K = torch.FloatTensor([[f, 0, vis_size_w / 2], [0, f, vis_size_h / 2], [0, 0, 1.0]])
        kinv = torch.inverse(self._K)
        coords = torch.matmul(kinv, torch.FloatTensor((vis_size_w - 0.5, 0, 1)).unsqueeze(-1)) * max_depth
        min_x, max_x = -coords[0], coords[0]
        x_den = (max_x - min_x) * (1 + self._epsilon)
        one = torch.ones(1)
        self._consts = (C.c_float * 8)(
            kinv[0, 0].item(), kinv[0, 2].item(), min_x.item(), x_den.item(),
            (one * (max_depth - min_depth)).item(), (one * ((max_depth - min_depth) * (1 + self._epsilon))).item(),
            (one * min_depth).item(), 0.0)
        assert kinv[0, 1].item() == 0.0
        self._work = None

    def _workspace(self, n, dev):
        need = _lib.lib.pnvo_topdown_workspace_bytes(int(n), self._vis_size_h, self._vis_size_w)
        if self._work is None or self._work.numel() < need or self._work.device != dev:
            self._work = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._work

    def gen_top_down_view_batch(self, depth, out=None, out_channel=None):
        """depth: CUDA float32 [N,H,W] (any strides with a regular frame/pixel stride).  Returns [N,H,W] or writes
        channel `out_channel` of `out` [N,H,W,Cc]."""
        H, W = self._vis_size_h, self._vis_size_w
        assert depth.is_cuda and depth.dtype == torch.float32 and depth.dim() == 3 and depth.shape[1:] == (H, W)
        assert depth.stride(1) == W * depth.stride(2)
        n, dev = depth.shape[0], depth.device
        if out is None:
            res = torch.empty((n, H, W), device=dev, dtype=torch.float32)
            o_ptr, ofs, ops = res.data_ptr(), H * W, 1
        else:
            res = out
            o_ptr = out.data_ptr() + 4 * out_channel
            ofs, ops = out.stride(0), out.stride(2)
        work = self._workspace(n, dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.pnvo_topdown_view(
                C.c_void_p(depth.data_ptr()), int(n), H, W, depth.stride(0) if n > 1 else H * W * depth.stride(2),
                depth.stride(2), self._consts, int(self._rows_around_center), C.c_void_p(o_ptr), int(ofs), int(ops),
                C.c_void_p(work.data_ptr()), _stream(dev)))
        return res

    def gen_top_down_view_pairs(self, depth_frames, out):
        """Both views of n (prev, cur) pairs in one pass of the kernels: depth_frames CUDA float32 [n,2,H,W] (contiguous) ->
        out [n,H,W,2] (contiguous), channel 0 = prev frame.  The values of two gen_top_down_view_batch calls."""
        H, W = self._vis_size_h, self._vis_size_w
        n, dev = depth_frames.shape[0], depth_frames.device
        assert depth_frames.is_cuda and depth_frames.dtype == torch.float32 and depth_frames.shape[1:] == (2, H, W)
        assert depth_frames.is_contiguous() and out.is_contiguous() and out.shape == (n, H, W, 2) and out.dtype == torch.float32
        work = self._workspace(2 * n, dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.pnvo_topdown_view_pairs(C.c_void_p(depth_frames.data_ptr()), int(n), H, W, self._consts,
                                                        int(self._rows_around_center), C.c_void_p(out.data_ptr()),
                                                        C.c_void_p(work.data_ptr()), _stream(dev)))
        return out

    def gen_top_down_view(self, normalized_depth):
        """normalized_depth: [H, W, 1] -> [H, W, 1]   (geometry_utils.py:516-556)."""
        d = normalized_depth.to(torch.float32)
        out = self.gen_top_down_view_batch(d[..., 0].contiguous().unsqueeze(0))
        return out[0].unsqueeze(-1)


class BaseRLTrainerWithVO:
    """Mirror of BaseRLTrainerWithVO.  Subclass it (or mix it in) exactly as the reference trainers do; it needs
    ``self.config`` (attribute-style VO / TASK_CONFIG tree) and ``self.device``."""

    boundary_chunks = None    # None: chosen from the pair count (1 below 24 pairs, 2 below 48, else 4); an integer forces it
    stage_threads = 12        # host threads that gather the simulator's frames into pinned staging (batched boundary call)

    def __init__(self, config=None, device=None):
        self.config = config
        self.device = device

    def _set_up_vo_obs_transformer(self) -> None:
        if self.config.VO.OBS_TRANSFORM in ("resize_crop", "resize"):
            raise NotImplementedError("VO.OBS_TRANSFORM other than 'none' is nav-loop plumbing outside the hot path "
                                      "(default is 'none', configs/rl/ddppo_pointnav.yaml:99)")
        self._vo_obs_transformer = None

    def _setup_vo_model(self, all_cfg) -> None:
        # base_trainer_with_vo.py:37-133
        if all_cfg.VO.VO_TYPE == "REGRESS":
            model_cls_name = all_cfg.VO.REGRESS_MODEL.name
            vo_model_cls = baseline_registry.get_vo_model(model_cls_name)
        else:
            raise NotImplementedError
        assert vo_model_cls is not None, f"{model_cls_name} is not supported"
        rm = all_cfg.VO.REGRESS_MODEL
        if rm.regress_type == "unified_act":
            output_dim, model_names = 3, ["all"]
        elif rm.regress_type == "sep_act":
            output_dim, model_names = 3, [_ for _ in list(ACT_IDX2NAME.values()) if _ != "unified"]
        else:
            raise ValueError
        self.vo_model = OrderedDict()
        for k in model_names:
            self.vo_model[k] = vo_model_cls(
                observation_space=rm.visual_type,
                observation_size=(self.config.VO.VIS_SIZE_W, self.config.VO.VIS_SIZE_H),
                hidden_size=rm.hidden_size, backbone=rm.visual_backbone, normalize_visual_inputs=True,
                output_dim=output_dim, dropout_p=rm.dropout_p,
                discretized_depth_channels=self.config.VO.REGRESS_MODEL.discretized_depth_channels)
            self.vo_model[k].to(self.device)
        if rm.pretrained:
            # :83-99.  The reference's files hold a config object and RNG states next to the weights; they are read with an
            # allow-list unpickler that keeps tensors only (checkpoint.py).  left / right usually name the same
            # model_states file (configs/rl/ddppo_pointnav.yaml:124-128): read each file once.
            files = {}
            for k in model_names:
                path = rm.pretrained_ckpt[k]
                if path not in files:
                    files[path] = load_vo_checkpoint(path)
                ckpt = files[path]
                if "model_state" in ckpt:
                    self.vo_model[k].load_state_dict(ckpt["model_state"])
                elif "model_states" in ckpt:
                    self.vo_model[k].load_state_dict(ckpt["model_states"][ACT_NAME2IDX[k]])
                else:
                    raise ValueError
        name = self.config.VO.REGRESS_MODEL.name
        if "discretize_depth" in name or "dd" in name:
            if self.config.VO.REGRESS_MODEL.discretize_depth not in ["hard"]:
                raise NotImplementedError
        if "top_down" in name:
            ds = self.config.TASK_CONFIG.SIMULATOR.DEPTH_SENSOR
            self._top_down_view_generator = NormalizedDepth2TopDownViewHabitatTorch(
                min_depth=ds.MIN_DEPTH, max_depth=ds.MAX_DEPTH, vis_size_h=self.config.VO.VIS_SIZE_H,
                vis_size_w=self.config.VO.VIS_SIZE_W, hfov_rad=ds.HFOV)

    # ------------------------------------------------------------------ pre-processing
    def _discretize_depth_func(self, raw_depth):
        """[..] float32 CUDA depth in [0,1] -> [.., bins] one-hot (base_trainer_with_vo.py:135-167)."""
        bins = self.config.VO.REGRESS_MODEL.discretized_depth_channels
        d = raw_depth.to(torch.float32).contiguous()
        out = torch.empty(d.shape + (bins,), device=d.device, dtype=torch.float32)
        flag = torch.zeros(1, dtype=torch.int32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.lib.pnvo_discretize_depth(C.c_void_p(d.data_ptr()), d.numel(), 1, int(bins),
                                                      C.c_void_p(out.data_ptr()), int(bins),
                                                      C.c_void_p(flag.data_ptr()), _stream(d.device)))
        assert flag.item() == 0, "depth must lie in [0, 1]"      # the reference's asserts (:136-137)
        return out

    def _build_obs_pairs(self, rgb_pair, depth_pair):
        """rgb_pair [B,H,W,6], depth_pair [B,H,W,2] on device -> obs_pairs dict (:209-269), batched on device."""
        obs_pairs = {"rgb": rgb_pair, "depth": depth_pair}
        name = self.config.VO.REGRESS_MODEL.name
        B, H, W, _ = depth_pair.shape
        dev = depth_pair.device
        if "discretize_depth" in name or "dd" in name:
            assert depth_pair.size(-1) == 2
            bins = self.config.VO.REGRESS_MODEL.discretized_depth_channels
            dd = torch.empty((B, H, W, 2 * bins), device=dev, dtype=torch.float32)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                for k in range(2):   # prev | cur halves (:220-229)
                    _lib.check(_lib.lib.pnvo_discretize_depth(
                        C.c_void_p(depth_pair.data_ptr() + 4 * k), B * H * W, 2, int(bins),
                        C.c_void_p(dd.data_ptr() + 4 * k * bins), 2 * int(bins), C.c_void_p(flag.data_ptr()),
                        _stream(dev)))
            self._dd_flag = flag
            obs_pairs["discretized_depth"] = dd
        if "top_down" in name:
            tdv = torch.empty((B, H, W, 2), device=dev, dtype=torch.float32)
            for k in range(2):       # :239-249
                self._top_down_view_generator.gen_top_down_view_batch(depth_pair[..., k], out=tdv, out_channel=k)
            obs_pairs["top_down_view"] = tdv
        return obs_pairs

    def _boundary_buffers(self, n, H, W, bins, want_rgb, want_tdv):
        """Reusable staging for n pairs: pinned host frames, their device twins, the observation-pair tensors, the
        top-down workspace and a pinned error flag — allocated once per (capacity, shape)."""
        pairs = self.config.VO.REGRESS_MODEL.mode != "det"
        st = getattr(self, "_bstage", None)
        if st is None or st["cap"] < n or st["shape"] != (H, W, bins, want_rgb, want_tdv, pairs):
            same = st is not None and st["shape"] == (H, W, bins, want_rgb, want_tdv, pairs)
            cap = max(n, 2 * st["cap"]) if same else n        # grow geometrically for callers with a varying pair count
            dev = self.device
            st = dict(cap=cap, shape=(H, W, bins, want_rgb, want_tdv, pairs),
                      h_rgb=torch.empty((cap, 2, H, W, 3), dtype=torch.uint8).pin_memory() if want_rgb else None,
                      h_dep=torch.empty((cap, 2, H, W), dtype=torch.float32).pin_memory(),
                      d_rgb=torch.empty((cap, 2, H, W, 3), dtype=torch.uint8, device=dev) if want_rgb else None,
                      d_dep=torch.empty((cap, 2, H, W), dtype=torch.float32, device=dev),
                      # (the float32 observation-pair tensors exist for mode 'rnd' only: 'det' feeds the frames to the model)
                      rgb=torch.empty((cap, H, W, 6), dtype=torch.float32, device=dev) if (want_rgb and pairs) else None,
                      depth=torch.empty((cap, H, W, 2), dtype=torch.float32, device=dev) if pairs else None,
                      dd=torch.empty((cap, H, W, 2 * bins), dtype=torch.float32, device=dev) if (bins and pairs) else None,
                      tdv=torch.empty((cap, H, W, 2), dtype=torch.float32, device=dev) if want_tdv else None,
                      work=torch.empty(int(_lib.lib.pnvo_topdown_workspace_bytes(cap, H, W)), dtype=torch.uint8, device=dev)
                      if want_tdv else None,
                      flag=torch.zeros(1, dtype=torch.int32, device=dev), h_flag=torch.zeros(1, dtype=torch.int32).pin_memory())
            self._bstage = st
        return st

    @staticmethod
    def _frame_ptrs(frames, dtype, shape):
        """ctypes pointer array over separately allocated numpy frames (made contiguous / cast only when they are not)."""
        need = 1
        for d in shape:
            need *= int(d)
        dt = np.dtype(dtype)
        keep = [f if (type(f) is np.ndarray and f.dtype == dt and f.size == need and f.flags.c_contiguous)
                else np.ascontiguousarray(np.asarray(f, dtype=dtype).reshape(shape)) for f in frames]
        arr = (C.c_void_p * len(keep))(*[k.__array_interface__["data"][0] for k in keep])   # (.ctypes builds an object per frame)
        return arr, keep

    # ------------------------------------------------------------------------------------------------ frame ring (env_ids)
    def reset_frame_ring(self, env_ids=None):
        """Forget the recorded frames of the given environments (all of them by default) and give their device slots back: a caller
        that mints fresh environment ids (one per episode, say) and resets the finished ones keeps the ring at its working size."""
        src = getattr(self, "_ring_src", None)
        if src is None:
            return
        rg = getattr(self, "_ring", None)
        ids = list(src.keys() if env_ids is None else env_ids)
        if rg is not None and env_ids is None:
            ids = list(set(ids) | set(rg["slot_of"].keys()))
        for e in ids:
            src.pop(e, None)
            if rg is not None and e in rg["slot_of"]:
                rg["free"].append(rg["slot_of"].pop(e))

    @staticmethod
    def _frame_fingerprint(depth, rgb):
        """A strided sample of a frame's bytes (~0.3 KB): what the ring compares on an identity hit, so that an observation buffer
        REFILLED IN PLACE (shared-memory vector environments, preallocated buffers, np.copyto) is recognised as a new frame — a
        new observation differs in nearly every sampled pixel — and uploaded, instead of being served from the ring.
        LIMIT (documented contract): only WHOLE-FRAME refills are detected.  An in-place edit that misses all ~61 / ~67
        sampled positions (a small moving region, a patch painted into the frame) is served stale; callers that edit recorded frames
        partially must pass fresh arrays or call reset_frame_ring(env_ids)."""
        # (~64 samples per tensor through the flat iterator: a view for contiguous frames, and for a non-contiguous one it gathers
        #  only the sampled elements — ravel() would copy the whole frame on every call)
        d = depth if type(depth) is np.ndarray else np.asarray(depth)
        fd = d.reshape(-1)[:: max(1, d.size // 61)] if d.flags.c_contiguous else d.flat[:: max(1, d.size // 61)]
        if rgb is None:
            return fd.tobytes()
        r = rgb if type(rgb) is np.ndarray else np.asarray(rgb)
        fr = r.reshape(-1)[:: max(1, r.size // 67)] if r.flags.c_contiguous else r.flat[:: max(1, r.size // 67)]
        return (fd.tobytes(), fr.tobytes())

    def _ring_buffers(self, slots, m, H, W, want_rgb, want_tdv):
        """Device ring (one slot per environment: its last cur frame + top-down view) and the upload staging of m frames."""
        dev = self.device
        rg = getattr(self, "_ring", None)
        shape = (H, W, want_rgb, want_tdv)
        if rg is None or rg["shape"] != shape:
            rg = dict(shape=shape, slots=0, cap=0, slot_of={}, free=[])
            self._ring = rg
            self._ring_src = {}
        if rg["slots"] < slots:
            ns = max(slots, 2 * rg["slots"], 8)
            new = dict(rgb=torch.empty((ns, H, W, 3), dtype=torch.uint8, device=dev) if want_rgb else None,
                       dep=torch.empty((ns, H, W), dtype=torch.float32, device=dev),
                       tdv=torch.empty((ns, H, W), dtype=torch.float32, device=dev) if want_tdv else None)
            for k, t in new.items():
                if t is not None and rg["slots"] > 0:
                    t[: rg["slots"]].copy_(rg[k])
            rg.update(new)
            rg["slots"] = ns
        if rg["cap"] < m:
            cap = max(m, 2 * rg["cap"])
            rg.update(cap=cap,
                      h_rgb=torch.empty((cap, H, W, 3), dtype=torch.uint8).pin_memory() if want_rgb else None,
                      h_dep=torch.empty((cap, H, W), dtype=torch.float32).pin_memory(),
                      u_rgb=torch.empty((cap, H, W, 3), dtype=torch.uint8, device=dev) if want_rgb else None,
                      u_dep=torch.empty((cap, H, W), dtype=torch.float32, device=dev),
                      u_tdv=torch.empty((cap, H, W), dtype=torch.float32, device=dev) if want_tdv else None,
                      h_idx=torch.empty((3 * cap,), dtype=torch.int32).pin_memory(),
                      d_idx=torch.empty((3 * cap,), dtype=torch.int32, device=dev))
            rg["h_idx_np"] = rg["h_idx"].numpy()              # (filled through numpy: a torch op per field costs more than the copy)
        return rg

    def _stage_pairs_through_ring(self, st, prev_obs_list, cur_obs_list, env_ids, H, W, want_rgb, want_tdv, gen):
        """The frames of n pairs into st[d_rgb / d_dep / tdv] with ONE uploaded frame per pair wherever the pair's prev frame is the
        frame its environment handed over as cur the call before (object identity of the numpy arrays: what `prev_obs = observations`
        in the reference's loop gives; a reset, a new environment or a copied frame simply uploads both).  Bit-identical to staging
        both frames: the ring holds the very bytes that were uploaded, and the top-down view is a function of the frame alone.
        CONTRACT: a recorded frame is not modified between the call that handed it over as cur and the call that hands it over as
        prev.  Guard: identity alone would also match a buffer refilled in place, so a hit additionally needs the frame's strided
        fingerprint (_frame_fingerprint, taken when it was recorded) to be unchanged; otherwise both frames are uploaded."""
        n, dev = len(env_ids), self.device
        assert len(set(env_ids)) == n, "env_ids must be distinct within a call"
        src = getattr(self, "_ring_src", None)
        if src is None or getattr(self, "_ring", None) is None or self._ring["shape"] != (H, W, want_rgb, want_tdv):
            src = {}
        hits = []
        for pv, cv, e in zip(prev_obs_list, cur_obs_list, env_ids):
            rec = src.get(e)
            ok = (rec is not None and rec[0] is pv["depth"] and (not want_rgb or rec[1] is pv["rgb"])
                  and pv["depth"] is not cv["depth"]
                  and rec[2] == self._frame_fingerprint(pv["depth"], pv["rgb"] if want_rgb else None))
            hits.append(ok)
        miss = [i for i in range(n) if not hits[i]]
        m = n + len(miss)
        same = getattr(self, "_ring", None) is not None and self._ring["shape"] == (H, W, want_rgb, want_tdv)
        known = self._ring["slot_of"] if same else {}
        nfree = len(self._ring["free"]) if same else 0
        need_slots = len(known) + nfree + max(0, sum(1 for e in env_ids if e not in known) - nfree)
        rg = self._ring_buffers(need_slots, m, H, W, want_rgb, want_tdv)
        src = self._ring_src
        slot_of = rg["slot_of"]
        for e in env_ids:
            if e not in slot_of:                                      # a slot given back by reset_frame_ring first, a new one otherwise
                slot_of[e] = rg["free"].pop() if rg["free"] else len(slot_of) + len(rg["free"])
        up = list(cur_obs_list) + [prev_obs_list[i] for i in miss]                 # frames to upload: every cur, the missed prevs
        hx = rg["h_idx_np"]                                   # [3][n]: cur index | prev index or -1 (ring) | slot
        hx[0:n] = np.arange(n, dtype=np.int32)
        hx[n:2 * n] = -1
        for k, i in enumerate(miss):
            hx[n + i] = n + k
        hx[2 * n:3 * n] = [slot_of[e] for e in env_ids]
        rg["d_idx"][: 3 * n].copy_(rg["h_idx"][: 3 * n], non_blocking=True)      # (first on the stream: long landed when the assemble kernel runs)
        p = lambda t, off=0: C.c_void_p(t[off:].data_ptr()) if t is not None else None
        pd_all, keep_d = self._frame_ptrs([f["depth"] for f in up], np.float32, (H, W))
        pr_all, keep_r = self._frame_ptrs([f["rgb"] for f in up], np.uint8, (H, W, 3)) if want_rgb else (None, None)
        vp = C.sizeof(C.c_void_p)
        nchunks = self.boundary_chunks or (1 if m < 24 else (2 if m < 64 else 4))
        nchunks = max(1, min(int(nchunks), m))           # (a chunk holds at least one frame)
        bounds = [(m * c // nchunks, m * (c + 1) // nchunks) for c in range(nchunks)]
        main = torch.cuda.current_stream(dev)
        if nchunks > 1 and getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(dev)
        copy = self._copy_stream if nchunks > 1 else main
        if nchunks > 1:
            copy.wait_stream(main)
        for lo, hi in bounds:
            k = hi - lo
            threads = min(self.stage_threads, k)
            pd_c = C.c_void_p(C.addressof(pd_all) + lo * vp)
            with torch.cuda.stream(copy):
                if want_rgb:
                    pr_c = C.c_void_p(C.addressof(pr_all) + lo * vp)
                    _lib.check(_lib.lib.pnvo_stage_frames2(pr_c, H * W * 3, p(rg["h_rgb"], lo), pd_c, H * W * 4, p(rg["h_dep"], lo), k, threads))
                    rg["u_rgb"][lo:hi].copy_(rg["h_rgb"][lo:hi], non_blocking=True)
                else:
                    _lib.check(_lib.lib.pnvo_stage_frames(pd_c, k, H * W * 4, p(rg["h_dep"], lo), threads))
                rg["u_dep"][lo:hi].copy_(rg["h_dep"][lo:hi], non_blocking=True)
            if nchunks > 1:
                main.wait_stream(copy)
            if want_tdv:
                gen.gen_top_down_view_batch(rg["u_dep"][lo:hi], out=rg["u_tdv"][lo:hi].unsqueeze(-1), out_channel=0)
        _lib.check(_lib.lib.pnvo_ring_assemble(p(rg["u_rgb"]), p(rg["u_dep"]), p(rg["u_tdv"]), p(rg["rgb"]), p(rg["dep"]), p(rg["tdv"]),
                                               p(rg["d_idx"]), int(n), int(H), int(W), p(st["d_rgb"]), p(st["d_dep"]), p(st["tdv"]),
                                               _stream(dev)))
        for cv, e in zip(cur_obs_list, env_ids):
            src[e] = (cv["depth"], cv["rgb"] if want_rgb else None, self._frame_fingerprint(cv["depth"], cv["rgb"] if want_rgb else None))
        self._ring_stats = dict(pairs=n, uploaded_frames=m, ring_hits=n - len(miss))
        return keep_d, keep_r

    def ring_depth(self, obs_list, env_ids):
        """Depth observations of the given environments as ONE device tensor [E,H,W,1] float32 — what the navigation policy consumes
        (rl/ppo/ppo_trainer.py:760-770: `batch["depth"]` of the step's observations, uploaded there by batch_obs).  In the reference's
        loop the observation the policy acts on is the frame the VO call of the step before received as cur_obs (`prev_obs = observations`,
        ppo_trainer.py:724-841): with env_ids that frame's bytes are already in the device ring, so an environment whose observation is
        the recorded array (object identity + the strided fingerprint, the ring's own hit rule) is served from there by one gather
        launch and only the others are uploaded.  Bit-identical to uploading obs["depth"] of every environment.  The result is a new
        tensor (not a view of the ring)."""
        E, dev = len(env_ids), self.device
        first = np.asarray(obs_list[0]["depth"])
        H, W = first.shape[0], first.shape[1]
        rg, src = getattr(self, "_ring", None), getattr(self, "_ring_src", None)
        hit = []
        if rg is not None and src is not None and rg["shape"][:2] == (H, W):
            want_rgb = rg["shape"][2]
            for o, e in zip(obs_list, env_ids):
                rec = src.get(e)
                hit.append(rec is not None and e in rg["slot_of"] and rec[0] is o["depth"] and (not want_rgb or rec[1] is o.get("rgb"))
                           and rec[2] == self._frame_fingerprint(o["depth"], o["rgb"] if want_rgb else None))
        else:
            hit = [False] * E
        if not any(hit):
            return torch.from_numpy(np.stack([np.asarray(o["depth"], dtype=np.float32).reshape(H, W, 1) for o in obs_list])).to(dev)
        cache = rg.setdefault("gather_idx", {})
        if all(hit):
            key = tuple(rg["slot_of"][e] for e in env_ids)
            idx = cache.get(key)
            if idx is None:
                if len(cache) > 64:
                    cache.clear()
                idx = cache[key] = torch.tensor(key, dtype=torch.long, device=dev)
            return rg["dep"].index_select(0, idx).unsqueeze(-1)
        out = torch.empty((E, H, W, 1), dtype=torch.float32, device=dev)
        hi = [i for i in range(E) if hit[i]]
        mi = [i for i in range(E) if not hit[i]]
        out[torch.tensor(hi, device=dev)] = rg["dep"].index_select(
            0, torch.tensor([rg["slot_of"][env_ids[i]] for i in hi], dtype=torch.long, device=dev)).unsqueeze(-1)
        out[torch.tensor(mi, device=dev)] = torch.from_numpy(
            np.stack([np.asarray(obs_list[i]["depth"], dtype=np.float32).reshape(H, W, 1) for i in mi])).to(dev)
        return out

    def compute_local_delta_states_batch(self, prev_obs_list, cur_obs_list, acts, env_ids=None):
        """Batched sibling of _compute_local_delta_states_from_vo: lists of observation dicts and actions ->
        float32 array [N,3].  env_ids (optional, one hashable id per pair, distinct within a call; mode 'det'): the call keeps each
        environment's cur frame and its top-down view on the device and, when the next call's prev frame IS that frame (the
        reference's loop: prev_obs = observations), uploads and pre-processes only the new frame — results are bit-identical.  The 2N raw frames are gathered into pinned staging by pnvo_stage_frames (parallel memcpy) and
        cross PCIe in 1-4 chunks on a copy stream; behind each chunk the two top-down views of its frames are built on the
        caller's stream.  Mode 'det': the frames then go STRAIGHT into the model (pnvo_forward_raw: pair concatenation, uint8 ->
        float and the one-hot depth of base_trainer_with_vo.py:172-269 happen in the stem's operand fetch; no observation-pair
        tensors are built), one forward per action model (sep_act) over all of its pairs.  Mode 'rnd' (train-mode forwards with
        dropout, :295-308) builds the observation pairs (pnvo_build_obs_pairs).  One host synchronisation at the end."""
        assert len(prev_obs_list) == len(cur_obs_list) == len(acts)
        n = len(acts)
        H, W = prev_obs_list[0]["depth"].shape[:2]
        rm = self.config.VO.REGRESS_MODEL
        if rm.mode not in ("det", "rnd"):
            raise NotImplementedError(f"VO.REGRESS_MODEL.mode == {rm.mode!r}")
        name = rm.name
        vis = list(rm.visual_type)
        want_rgb = "rgb" in vis
        bins = int(rm.discretized_depth_channels) if ("discretize_depth" in name or "dd" in name) else 0
        want_tdv = "top_down" in name
        st = self._boundary_buffers(n, H, W, bins, want_rgb, want_tdv)
        dev = self.device
        p = lambda t, off=0: C.c_void_p(t[off:].data_ptr()) if t is not None else None
        gen = self._top_down_view_generator if want_tdv else None
        out = np.zeros((n, 3), dtype=np.float32)
        std = np.zeros((n, 3), dtype=np.float32)
        keys = ["all"] * n if rm.regress_type == "unified_act" else [ACT_IDX2NAME[a] for a in acts]
        # pairs travel grouped by action model (stable order): every model then reads ONE contiguous slice of the staged frames —
        # no gather kernels between the copy and the forwards; results go back to the caller's order on the host
        order = sorted(range(n), key=lambda i: keys[i])
        if order != list(range(n)):
            prev_obs_list = [prev_obs_list[i] for i in order]
            cur_obs_list = [cur_obs_list[i] for i in order]
            acts = [acts[i] for i in order]
            keys = [keys[i] for i in order]
            if env_ids is not None:
                env_ids = [env_ids[i] for i in order]
        else:
            order = None
        use_ring = env_ids is not None and rm.mode == "det"
        if env_ids is not None:
            assert len(env_ids) == n
        # Large batches travel as 2-4 chunks: while chunk c is gathered on the host and crosses PCIe on a copy stream, the
        # top-down views of chunk c-1 are built on the caller's stream (host staging, transfer and device work overlap).
        nchunks = self.boundary_chunks or (1 if n < 24 else (2 if n < 48 else 4))
        nchunks = max(1, min(int(nchunks), n))           # (a chunk holds at least one pair)
        bounds = [(n * c // nchunks, n * (c + 1) // nchunks) for c in range(nchunks)]
        main = torch.cuda.current_stream(dev)
        if nchunks > 1 and getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(dev)
        copy = self._copy_stream if nchunks > 1 else main
        pending = []
        if use_ring:
            bounds = []                                   # the ring path stages (and chunks) on its own
            with torch.cuda.device(dev), torch.no_grad():
                keep_d, keep_r = self._stage_pairs_through_ring(st, prev_obs_list, cur_obs_list, env_ids, H, W, want_rgb, want_tdv, gen)
        else:
            frames = [o for pc in zip(prev_obs_list, cur_obs_list) for o in pc]
            # address tables over all 2N frames, once (frames are used in place when they are contiguous arrays of the right type)
            pd_all, keep_d = self._frame_ptrs([f["depth"] for f in frames], np.float32, (H, W))
            pr_all, keep_r = self._frame_ptrs([f["rgb"] for f in frames], np.uint8, (H, W, 3)) if want_rgb else (None, None)
        vp = C.sizeof(C.c_void_p)
        with torch.cuda.device(dev), torch.no_grad():
            st["flag"].zero_()
            if nchunks > 1 and not use_ring:
                copy.wait_stream(main)
            for lo, hi in bounds:
                m = hi - lo
                threads = min(self.stage_threads, 2 * m)  # persistent copy workers of libpnvo (pnvo_stage_frames)
                pd_c = C.c_void_p(C.addressof(pd_all) + 2 * lo * vp)
                with torch.cuda.stream(copy):
                    if want_rgb:
                        pr_c = C.c_void_p(C.addressof(pr_all) + 2 * lo * vp)
                        _lib.check(_lib.lib.pnvo_stage_frames2(pr_c, H * W * 3, p(st["h_rgb"], lo), pd_c, H * W * 4, p(st["h_dep"], lo),
                                                               2 * m, threads))
                        st["d_rgb"][lo:hi].copy_(st["h_rgb"][lo:hi], non_blocking=True)
                    else:
                        _lib.check(_lib.lib.pnvo_stage_frames(pd_c, 2 * m, H * W * 4, p(st["h_dep"], lo), threads))
                    st["d_dep"][lo:hi].copy_(st["h_dep"][lo:hi], non_blocking=True)
                if nchunks > 1:
                    main.wait_stream(copy)
                if rm.mode == "rnd":                      # the train-mode forward takes observation pairs
                    _lib.check(_lib.lib.pnvo_build_obs_pairs(
                        p(st["d_rgb"], lo), p(st["d_dep"], lo), int(m), int(H), int(W), int(bins), gen._consts if gen else None,
                        int(gen._rows_around_center) if gen else 0, p(st["work"]), p(st["rgb"], lo), p(st["depth"], lo),
                        p(st["dd"], lo), p(st["tdv"], lo), p(st["flag"]), _stream(dev)))
                elif want_tdv:                            # :239-249: prev frames -> channel 0, cur frames -> channel 1
                    gen.gen_top_down_view_pairs(st["d_dep"][lo:hi], st["tdv"][lo:hi])
            ukeys = sorted(set(keys))
            # Several action models, few pairs each (the navigation loop: 8-32 environments over forward / left / right): ONE launch
            # chain over all pairs (vo_cnn.grouped_forward_raw) instead of one small forward per model — each of those is bound by its
            # ~50 dependent launches, not by its work.  From `group_max_pairs` on the per-model forwards win (larger batches run on
            # the resident-weight stem and the row-streaming kernels, which hold one model's weights).
            grouped = (rm.mode == "det" and "act_embed" not in name and 2 <= len(ukeys) <= 3 and n <= int(getattr(self, "group_max_pairs", 48))
                       and all(getattr(self.vo_model[k], "_precision", "float32") == "float32" for k in ukeys))
            if grouped:
                models = [self.vo_model[k] for k in ukeys]
                for mdl in models:
                    if mdl.training:
                        mdl.eval()
                from .vo_cnn import grouped_forward_raw, grouped_supported
                grouped, self._grouped_reason = grouped_supported(models)      # (handles off the default kernels: per-model forwards)
            if grouped:
                counts = [sum(1 for k in keys if k == key) for key in ukeys]      # (the pairs are sorted by key)
                pending.append((list(range(n)), grouped_forward_raw(models, counts, st["d_rgb"][:n] if want_rgb else None, st["d_dep"][:n],
                                                                    st["tdv"][:n] if want_tdv else None, err_flag=st["flag"])))
                ukeys = []
            for key in ukeys:
                idx = [i for i, k in enumerate(keys) if k == key]            # contiguous: the pairs are sorted by key
                lo_k, hi_k = idx[0], idx[-1] + 1
                pick = lambda t: None if t is None else t[lo_k:hi_k]
                model = self.vo_model[key]
                a = torch.as_tensor([acts[i] for i in idx], dtype=torch.long, device=dev) if "act_embed" in name else None
                if rm.mode == "det":                      # :285-294
                    if model.training:                    # (eval() walks ~75 sub-modules: only when it changes anything)
                        model.eval()
                    pending.append((idx, model.forward_raw(pick(st["d_rgb"]) if want_rgb else None, pick(st["d_dep"]),
                                                           pick(st["tdv"]) if want_tdv else None, a, err_flag=st["flag"])))
                else:                                     # 'rnd', :295-308: rnd_mode_n train-mode (dropout) forwards
                    sub = {"depth": pick(st["depth"])}
                    if want_rgb:
                        sub["rgb"] = pick(st["rgb"])
                    if bins:
                        sub["discretized_depth"] = pick(st["dd"])
                    if want_tdv:
                        sub["top_down_view"] = pick(st["tdv"])
                    sub = {k: v for k, v in sub.items() if k in vis}
                    model.train()
                    samples = np.stack([(model(sub, a) if a is not None else model(sub)).cpu().numpy()
                                        for _ in range(int(rm.rnd_mode_n))])
                    out[idx] = samples.mean(axis=0)
                    std[idx] = samples.std(axis=0)
            st["h_flag"].copy_(st["flag"], non_blocking=True)
            for idx, res in pending:                      # the first .cpu() is the one synchronisation of the call
                out[idx] = res.cpu().numpy()
        if not pending:
            torch.cuda.current_stream(dev).synchronize()
        assert int(st["h_flag"][0]) == 0, "depth must lie in [0, 1]"      # the reference's asserts (:136-137)
        if order is not None:                             # back to the caller's order
            inv = np.empty(n, dtype=np.int64)
            inv[np.asarray(order)] = np.arange(n)
            out, std = out[inv], std[inv]
        self._last_std = std
        return out

    def _compute_local_delta_states_from_vo(self, prev_obs, cur_obs, act, vis_video=False):
        """(prev_obs, cur_obs, act) -> (list of 3 np.float32, std list, extra_infos)  (:169-314; std is [0,0,0] in mode
        'det' and the per-component standard deviation of the rnd_mode_n dropout samples in mode 'rnd')."""
        if getattr(self, "_vo_obs_transformer", None) is not None:
            raise NotImplementedError
        deltas = self.compute_local_delta_states_batch([prev_obs], [cur_obs], [act])
        extra_infos = {}
        if vis_video and "top_down" in self.config.VO.REGRESS_MODEL.name:
            d = torch.from_numpy(np.ascontiguousarray(cur_obs["depth"], dtype=np.float32)).to(self.device)
            extra_infos["ego_top_down_map"] = self._top_down_view_generator.gen_top_down_view(d)
        stds = [0, 0, 0] if self.config.VO.REGRESS_MODEL.mode == "det" else list(self._last_std[0])
        return list(deltas[0]), stds, extra_infos
