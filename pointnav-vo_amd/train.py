"""One optimisation step of a VO action model on the MI355X (BASELINE config 4).

Mirrors, for one action model, the body of the reference's training iteration
(/root/reference/pointnav_vo/vo/engine/vo_cnn_regression_geo_invariance_engine.py:855-901):

    optimizer.zero_grad(); out = vo_model(batch_pairs)      (model.train(): RunningMeanAndVar updates, dropout)
    loss = sum_d mean((gt_d - pred_d)^2)                     (vo_cnn_engine.py:135-198, loss_weight_fixed)
    loss.backward(); optimizer.step()                        (Adam lr 2.5e-4, eps 1e-8, wd 0: :122-133)

Forward, backward, loss and Adam are HIP kernels behind the C ABI (pnvo_train_*); this class only owns the flat
device buffers and performs the collectives the reference semantics call for when torch.distributed is initialised:
RunningMeanAndVar's all-reduces (running_mean_and_var.py:27-38; batch sum and count travel as one buffer, the variance
needs the global mean and is the second round) and the all-reduce of the flat gradient buffer (RCCL over xGMI on the GPU box;
15.85 MB for the default model, in three buckets that start while the backward is still running: pnvo_train_set_grad_hook).  Dropout (the reference trains with p = 0.2 before both
Linear layers) uses a counter-based hash mask instead of torch's RNG stream — same distribution and arithmetic, a different
but reproducible random draw (pnvo_train_set_dropout; `dropout_masks()` returns the masks of the last step for checkers).
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib, parallel


def ms_feature_hw(cfg):
    """Spatial size of the compressed feature map: 5 stride-2 stages, ceil at each (resnet.py:156-212)."""
    h, w = cfg.height, cfg.width
    for _ in range(5):
        h, w = (h + 1) // 2, (w + 1) // 2
    return h, w


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class VOTrainStep:
    def __init__(self, model, lr=2.5e-4, eps=1e-8, betas=(0.9, 0.999), dropout_seed=0):
        self.model = model
        self.lr, self.eps, self.betas = float(lr), float(eps), betas
        ref = next(model.parameters())
        if ref.device.type != "cuda":
            raise RuntimeError("VOTrainStep runs on an MI355X only: move the model with .to('cuda') first")
        self.dev = ref.device
        model._ensure_handle(self.dev)
        model._sync_weights()                                   # allocates the kernel operand buffers
        named = [(n, p) for n, p in model.named_parameters()]
        total = sum(p.numel() for _, p in named)
        self.flat = torch.empty(total, device=self.dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=self.dev, dtype=torch.float32)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        toc = (_lib.pnvo_tensor_desc * len(named))()
        off = 0
        self.offsets = {}
        with torch.no_grad():
            for i, (n, p) in enumerate(named):
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)    # the module's parameters alias the flat buffer
                p.grad = self.grad[off:off + k].view(p.shape)
                toc[i].name = n.encode()
                toc[i].offset = off
                toc[i].ndim = p.dim()
                for d, sz in enumerate(p.shape):
                    toc[i].shape[d] = int(sz)
                self.offsets[n] = (off, k)
                off += k
        self._toc = toc
        self._named = named
        model._loaded_sig = None          # p.data now aliases the flat buffer: the eval path re-reads it on its next call
        _lib.check(_lib.lib.pnvo_train_attach(model._handle, _ptr(self.flat), _ptr(self.grad), total, toc, len(named)),
                   model._handle)
        self.dropout_p = float(getattr(model, "dropout_p", 0.0) or 0.0)
        _lib.check(_lib.lib.pnvo_train_set_dropout(model._handle, C.c_float(self.dropout_p), C.c_uint64(int(dropout_seed))),
                   model._handle)
        self.step_count = 0
        enc = model.visual_encoder
        self.rmv = getattr(enc, "running_mean_and_var", None) if model.cfg.normalize else None
        Cc = model.cfg.in_channels
        self._m12 = torch.empty(2 * Cc, device=self.dev)
        self._loss = torch.zeros(1, device=self.dev)
        self._psig = self._param_sig()
        # data parallel: the gradient all-reduce travels in buckets that start while the backward is still running
        # (pnvo_train_set_grad_hook; layer4..head first — 80 % of the bytes — then layer2-3, then stem + layer1);
        # bucketed = False: ONE flat all-reduce after the backward (bench.py --no-overlap, the A/B of the two schedules)
        self.bucketed = True
        self._pending = []                     # (work handle, first, count) of the all-reduces in flight
        self._comm = None
        self._hook = _lib.GRAD_READY_FN(self._on_grad_ready)
        _lib.check(_lib.lib.pnvo_train_set_grad_hook(model._handle, C.cast(self._hook, C.c_void_p), None), model._handle)

    # ------------------------------------------------------------------ parameter / optimizer state
    def _param_sig(self):
        return tuple((p.data_ptr(), p._version) for _, p in self._named)

    def _sync_params(self, stream):
        """Parameters edited outside the HIP Adam step (model.load_state_dict on resume, an in-place torch edit) land in
        the flat buffer but not in the packed kernel operands: re-pack before the next train-mode forward."""
        sig = self._param_sig()
        if sig == self._psig:
            return
        off = 0
        with torch.no_grad():
            for _, p in self._named:                      # a tensor re-pointed by the caller: alias it again
                k = p.numel()
                view = self.flat[off:off + k].view(p.shape)
                if p.data_ptr() != view.data_ptr():
                    view.copy_(p.detach())
                    p.data = view
                    p.grad = self.grad[off:off + k].view(p.shape)
                off += k
        # an outside edit can move a GroupNorm weight by any amount: the float16-piece range guard must be current NOW.  The forward
        # reads the bounds of the parameters TWO refreshes back (a fixed lag — fine for Adam's lr-sized steps, and the same on every
        # run and rank): three refreshes fill that ring with the edited parameters' bounds
        for _ in range(3):
            _lib.check(_lib.lib.pnvo_train_refresh(self.model._handle, stream), self.model._handle)
        torch.cuda.current_stream(self.dev).synchronize()
        self.model._loaded_sig = None
        self._psig = self._param_sig()

    def state_dict(self):
        """Optimizer state in torch.optim.Adam's layout (the reference checkpoints `optim_states`,
        vo_cnn_regression_geo_invariance_engine.py:1425-1433): per-parameter step / exp_avg / exp_avg_sq."""
        state, off = {}, 0
        for i, (_, p) in enumerate(self._named):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + k].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape).clone()}
            off += k
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self._named)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        self.lr, self.eps, self.betas = float(group["lr"]), float(group["eps"]), tuple(group["betas"])
        off, steps = 0, set()
        with torch.no_grad():
            for i, (_, p) in enumerate(self._named):
                k = p.numel()
                st = sd["state"].get(i)
                if st is None:
                    self.exp_avg[off:off + k].zero_()
                    self.exp_avg_sq[off:off + k].zero_()
                else:
                    self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                off += k
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ; the HIP Adam keeps one step count for the model")
        self.step_count = steps.pop() if steps else 0

    def dropout_masks(self, batch):
        """Scaled masks (0 or 1/(1-p)) of the last forward: (m0 [B, fh*fw, Cpad] in the kernel's NHWC order with the
        compression channels padded to a multiple of 32, m1 [B, hidden])."""
        cfg = self.model.cfg
        fh, fw = ms_feature_hw(cfg)
        cpad = (cfg.comp_channels + 31) // 32 * 32
        m0 = torch.empty((batch, fh * fw, cpad), device=self.dev, dtype=torch.float32)
        m1 = torch.empty((batch, cfg.hidden), device=self.dev, dtype=torch.float32)
        s = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(_lib.lib.pnvo_train_dropout_mask(self.model._handle, 0, _ptr(m0), C.c_void_p(s)), self.model._handle)
        _lib.check(_lib.lib.pnvo_train_dropout_mask(self.model._handle, 1, _ptr(m1), C.c_void_p(s)), self.model._handle)
        if cfg.act_embed:                     # the embedding columns of the concatenated feature the Dropout acts on
            m2 = torch.empty((batch, 32), device=self.dev, dtype=torch.float32)
            _lib.check(_lib.lib.pnvo_train_dropout_mask(self.model._handle, 2, _ptr(m2), C.c_void_p(s)), self.model._handle)
            return m0, m1, m2
        return m0, m1

    # ------------------------------------------------------------------ pieces
    def _obs_ptrs(self, obs):
        c = self.model.cfg
        ptrs, keep, B = [], [], None
        for key, n in (("rgb", c.n_rgb), ("depth", c.n_depth), ("discretized_depth", c.n_dd), ("top_down_view", c.n_tdv)):
            if n == 0:
                ptrs.append(None)
                continue
            t = obs[key].to(device=self.dev, dtype=torch.float32).contiguous()
            assert t.shape[1:] == (c.height, c.width, n), (key, tuple(t.shape))
            B = t.shape[0] if B is None else B
            keep.append(t)
            ptrs.append(_ptr(t))
        return ptrs, keep, B

    def _input_moments(self, ptrs, B, center, power, out, stream):
        """Per-channel mean over (B, H, W) of (x - center)^power of the assembled input, by HIP kernels (one coalesced
        pass per observation tensor).  The only device work of the RunningMeanAndVar update: tests of the host logic
        substitute it."""
        h = self.model._handle
        _lib.check(_lib.lib.pnvo_input_moments(h, *ptrs, int(B), _ptr(center), int(power), _ptr(out), stream), h)

    def _update_running_stats(self, ptrs, B, stream):
        """RunningMeanAndVar.forward, training branch (running_mean_and_var.py:23-60): batch statistics from ONE pass over
        the observation tensors (_input_moments power 3: first and second moment about the current running mean c), the
        three all-reduces of :27-38 when torch.distributed is initialised, Chan's merge :44-60.  The reference's second
        pass — the variance about the GLOBAL batch mean mu — follows algebraically from the one-pass moments of this
        rank: mean((x - mu)^2) = E[(x-c)^2] - 2 (mu - c) E[x-c] + (mu - c)^2."""
        rmv = self.rmv
        distributed = dist.is_available() and dist.is_initialized()
        C_ = self._m12.numel() // 2
        center = rmv._mean.reshape(-1).to(torch.float32).contiguous()
        if B > 0:
            self._input_moments(ptrs, B, center, 3, self._m12, stream)
        else:                      # a rank without entries for this action model (participate_absent): contributes zeros
            self._m12.zero_()
        if not distributed and self._fused_rmv_ok():
            # one process: batch mean, variance about it and Chan's merge in ONE launch on the module's own buffers
            # (the ~30 tiny torch kernels below cost 0.12 ms of a 9.6 ms step)
            _lib.check(_lib.lib.pnvo_rmv_merge(_ptr(self._m12), C_, int(B), _ptr(rmv._mean), _ptr(rmv._var), _ptr(rmv._count), stream))
            # the buffers changed IN PLACE (same data_ptr, same _version): operands the eval path derives from its host copy
            # of the whitening statistics (bf16 stem, 64-output mx stems) must be rebuilt at the next eval / dual forward
            self.model._loaded_sig = None
            return
        e1, e2 = self._m12[:C_].double(), self._m12[C_:].double()
        # the reference's first two all-reduces (batch sum of the per-sample means, sample count: running_mean_and_var.py:
        # 27-33) travel as ONE buffer of C + 1 floats; the variance needs the global mean and stays a second round (:34-38)
        mc = torch.empty(C_ + 1, device=self._m12.device, dtype=torch.float32)
        mc[:C_] = ((center.double() + e1) * B).to(torch.float32)        # = adaptive_avg_pool2d(x, 1).sum(0)
        mc[C_] = float(B)
        if distributed:
            dist.all_reduce(mc)
        new_count = mc[C_].reshape_as(rmv._count).to(rmv._count.dtype)
        new_mean = (mc[:C_] / mc[C_]).view(1, -1, 1, 1)
        delta = new_mean.reshape(-1).double() - center.double()
        new_var = ((e2 - 2.0 * delta * e1 + delta * delta) * B).to(torch.float32).view(1, -1, 1, 1)
        if distributed:
            dist.all_reduce(new_var)
        new_var = new_var / new_count
        m_a = rmv._var * rmv._count
        m_b = new_var * new_count
        M2 = m_a + m_b + (new_mean - rmv._mean).pow(2) * rmv._count * new_count / (rmv._count + new_count)
        rmv._var = M2 / (rmv._count + new_count)
        rmv._mean = (rmv._count * rmv._mean + new_count * new_mean) / (rmv._count + new_count)
        rmv._count += new_count

    def _fused_rmv_ok(self):
        """The in-place device merge needs the reference's float32 buffers, contiguous, on this device."""
        rmv = self.rmv
        return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (rmv._mean, rmv._var, rmv._count, self._m12))

    def _set_actions(self, actions, B):
        """act_embed variants: hand the [B] int64 actions of this step to the library (kept alive until the backward)."""
        if not self.model.cfg.act_embed:
            return
        if actions is None:
            raise TypeError("act_embed model: actions are required")
        a = torch.as_tensor(actions).to(device=self.dev, dtype=torch.int64).contiguous().reshape(-1)
        if a.numel() != B:
            raise ValueError(f"actions has {a.numel()} entries for a batch of {B}")
        self._actions = a
        _lib.check(_lib.lib.pnvo_train_set_actions(self.model._handle, _ptr(a)), self.model._handle)

    def forward_train(self, obs_pairs, actions=None):
        """model.train(); out = model(obs_pairs[, actions]): RunningMeanAndVar update + dropout, activations kept for a
        backward."""
        h = self.model._handle
        ptrs, keep, B = self._obs_ptrs(obs_pairs)
        self._set_actions(actions, B)
        out = torch.empty((B, self.model.cfg.out_dim), device=self.dev, dtype=torch.float32)
        with torch.cuda.device(self.dev), torch.no_grad():
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            self._sync_params(stream)
            mean = var = None
            if self.rmv is not None:
                self._update_running_stats(ptrs, B, stream)
                mean = self.rmv._mean.reshape(-1).contiguous()
                var = self.rmv._var.reshape(-1).contiguous()
            _lib.check(_lib.lib.pnvo_train_forward(h, *ptrs, int(B), _ptr(mean), _ptr(var), _ptr(out), stream), h)
        return out

    def forward_backward(self, obs_pairs, target=None, grad_out=None, actions=None):
        """Train-mode forward + backward.  Either `target` [B,3] (the reference's regression loss) or an explicit
        `grad_out` = dLoss/dOut [B,3] (e.g. from the geometric-invariance loss computed on the [B,3] outputs).
        Returns (out, loss or None); gradients are left in self.grad (not yet all-reduced)."""
        h = self.model._handle
        ptrs, keep, B = self._obs_ptrs(obs_pairs)
        self._set_actions(actions, B)
        out = torch.empty((B, self.model.cfg.out_dim), device=self.dev, dtype=torch.float32)
        with torch.cuda.device(self.dev), torch.no_grad():
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            self._sync_params(stream)
            mean = var = None
            if self.rmv is not None:
                self._update_running_stats(ptrs, B, stream)
                mean = self.rmv._mean.reshape(-1).contiguous()
                var = self.rmv._var.reshape(-1).contiguous()
            _lib.check(_lib.lib.pnvo_train_forward(h, *ptrs, int(B), _ptr(mean), _ptr(var), _ptr(out), stream), h)
            loss = None
            if grad_out is None:
                tgt = target.to(device=self.dev, dtype=torch.float32).contiguous()
                grad_out = torch.empty_like(out)
                _lib.check(_lib.lib.pnvo_mse_loss(_ptr(out), _ptr(tgt), int(B), out.shape[1], _ptr(self._loss),
                                                  _ptr(grad_out), stream))
                loss = self._loss.clone()
            else:
                grad_out = grad_out.to(device=self.dev, dtype=torch.float32).contiguous()
            _lib.check(_lib.lib.pnvo_train_backward(h, _ptr(grad_out), stream), h)
        return out, loss

    def backward(self, grad_out):
        """loss.backward() for the LAST forward_train of this model given dLoss/dOut [B,3]."""
        h = self.model._handle
        grad_out = grad_out.to(device=self.dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.dev), torch.no_grad():
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            _lib.check(_lib.lib.pnvo_train_backward(h, _ptr(grad_out), stream), h)

    # ------------------------------------------------------------------ gradient all-reduce
    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _allreduce_bucket(self, first, count):
        """Start the all-reduce of grad[first : first + count] behind everything enqueued on the current stream so far, on the
        communication stream (RCCL runs next to the rest of the backward); optimizer_step waits for it."""
        main = torch.cuda.current_stream(self.dev)
        if self._comm is None:
            self._comm = torch.cuda.Stream(self.dev)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ev)
            work = dist.all_reduce(self.grad[first:first + count], async_op=True)
        self._pending.append((work, first, count))

    def _on_grad_ready(self, user, first, count, stream):
        """pnvo_grad_ready_fn: called by pnvo_train_backward on the host when a flat gradient range is final."""
        if self.bucketed and self._distributed():
            self._allreduce_bucket(int(first), int(count))

    def _bucket_ranges(self):
        first, count, n = (C.c_uint64 * 8)(), (C.c_uint64 * 8)(), C.c_int(0)
        _lib.check(_lib.lib.pnvo_train_grad_buckets(self.model._handle, first, count, 8, C.byref(n)), self.model._handle)
        return [(int(first[k]), int(count[k])) for k in range(n.value)]

    def _finish_gradient_allreduce(self):
        """Mean of the gradient over the ranks: wait for the buckets the backward started (or issue the same sequence now — a
        rank that had no backward this step — or the one flat all-reduce when bucketing is off), then divide."""
        if not self._distributed():
            return
        world = dist.get_world_size()
        if self.bucketed:
            if not self._pending:
                for first, count in self._bucket_ranges():
                    self._allreduce_bucket(first, count)
            for work, _, _ in self._pending:
                work.wait()                    # the current stream waits for the collective (no host block with RCCL)
            if self._comm is not None:
                torch.cuda.current_stream(self.dev).wait_stream(self._comm)
            self._pending = []
            self.grad /= world
        else:
            parallel.allreduce_mean_(self.grad)

    def participate_absent(self):
        """Data-parallel training, a rank whose batch holds NO entry of this action model while another rank's does: issue
        the same collectives a forward_train would (RunningMeanAndVar's two rounds, with zero contributions — every rank
        ends with the same running statistics, as in the reference where all ranks merge the all-reduced moments) and leave
        an all-zero gradient for optimizer_step's all-reduce.  Without it the ranks' collective sequences diverge: a hang,
        or buffers of different models reduced into one another."""
        with torch.cuda.device(self.dev), torch.no_grad():
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            if self.rmv is not None:
                self._update_running_stats(None, 0, stream)
                self.model._loaded_sig = None
            self.grad.zero_()
            self._pending = []                  # (no backward ran: optimizer_step issues the bucket sequence itself)

    def absent_backward(self):
        """The point of the collective sequence where this model's backward would start its bucket all-reduces, on a rank that
        ran participate_absent() instead of a forward (zero gradient)."""
        if self._distributed():
            if self.bucketed:
                for first, count in self._bucket_ranges():
                    self._allreduce_bucket(first, count)

    def optimizer_step(self, allreduce=True):
        """All-reduce (mean) of the flat gradient buffer across ranks, Adam, re-pack of the kernel operands."""
        h = self.model._handle
        if allreduce:
            self._finish_gradient_allreduce()                             # RCCL over xGMI: 15.85 MB in three buckets (or one)
        self.step_count += 1
        with torch.cuda.device(self.dev):
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            _lib.check(_lib.lib.pnvo_adam_step(_ptr(self.flat), _ptr(self.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                               self.flat.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                               self.step_count, stream))
            _lib.check(_lib.lib.pnvo_train_refresh(h, stream), h)
        self.model._loaded_sig = None     # operands only the eval path owns are rebuilt from the flat buffer on its next call

    def step(self, obs_pairs, target, actions=None):
        """zero_grad / forward / loss / backward / all-reduce / Adam — returns (out [B,3], loss tensor)."""
        out, loss = self.forward_backward(obs_pairs, target=target, actions=actions)
        self.optimizer_step()
        return out, loss


# ---------------------------------------------------------------------------------------------------------------------
CUR_REL_TO_PREV, PREV_REL_TO_CUR = 0, 1            # pointnav_vo/vo/common/common_vars.py
MOVE_FORWARD, TURN_LEFT, TURN_RIGHT = 1, 2, 3      # pointnav_vo/utils/misc_utils.py ACT_NAME2IDX values


NO_NOISE_DELTAS = {MOVE_FORWARD: (0.0, -0.25, 0.0), TURN_LEFT: (0.0, 0.0, math.radians(10.0)),
                   TURN_RIGHT: (0.0, 0.0, -math.radians(10.0))}     # common_vars.py (x, z, yaw)


def compute_loss_weights(actions, targets, multiplier=None, fixed=True):
    """VOCNNBaseEngine._compute_loss_weights (vo_cnn_engine.py:200-228): {"dx","dz","dyaw"} -> [M] float32 (host).  fixed
    (configs/vo/vo_pointnav.yaml:41, the default): the constant multipliers.  Otherwise exp(mult_d * |noise-free delta_d of
    the action - dx target|) — the reference passes the dx targets into all three exponents; kept as is."""
    multiplier = multiplier or {"dx": 1.0, "dz": 1.0, "dyaw": 1.0}
    t = torch.as_tensor(targets, dtype=torch.float32).reshape(-1, 3).cpu()
    if fixed:
        return {k: torch.full((t.shape[0],), float(multiplier[k])) for k in ("dx", "dz", "dyaw")}
    nn = torch.tensor([NO_NOISE_DELTAS[int(a)] for a in torch.as_tensor(actions).reshape(-1).tolist()], dtype=torch.float32)
    return {k: torch.exp(float(multiplier[k]) * torch.abs(nn[:, d] - t[:, 0])) for d, k in enumerate(("dx", "dz", "dyaw"))}


def regression_coef(n, data_types=None, loss_weights=None, dz_regress_masks=None, device=None, allowed_types=None):
    """coef [n,3] such that sum(coef * (gt - pred)^2) is what the reference adds up for one action model
    (vo_cnn_regression_geo_invariance_engine.py:618-740 calling vo_cnn_engine.py:135-198): for every data type the engine
    regresses (`tmp_data_types`, :679-684: CUR_REL_TO_PREV always, PREV_REL_TO_CUR only with inverse_data_augment_only /
    inverse_joint_train) and every d in (dx, dz, dyaw): mean over that subset of (diff^2 [* dz_regress_mask for dz]) *
    loss_weights[d].  Entries of any other data type carry no regression loss (coef 0)."""
    coef = torch.ones((n, 3), dtype=torch.float32)
    if data_types is None:
        coef /= float(n)
    else:
        dt = torch.as_tensor(data_types).reshape(-1).cpu()
        for t in torch.unique(dt).tolist():
            sel = dt == t
            if allowed_types is not None and t not in allowed_types:
                coef[sel] = 0.0
            else:
                coef[sel] /= float(int(sel.sum()))
    if loss_weights is not None:
        for d, k in enumerate(("dx", "dz", "dyaw")):
            coef[:, d] *= torch.as_tensor(loss_weights[k], dtype=torch.float32).reshape(-1).cpu()
    if dz_regress_masks is not None:
        coef[:, 1] *= torch.as_tensor(dz_regress_masks, dtype=torch.float32).reshape(-1).cpu()
    return coef.to(device) if device is not None else coef


class GeoInvarianceTrainStep:
    """One training iteration of VOCNNRegressionGeometricInvarianceEngine for a dict of action models
    (vo_cnn_regression_geo_invariance_engine.py:855-901 around _process_one_batch :451-807): every model sees the
    entries of its action, the regression losses are per data type, and with "inverse_joint_train" the predictions of all
    models, put back in batch order, are tied by the inverse-consistency loss (:367-449, weight
    VO.GEOMETRY.loss_inv_weight).  Forward/backward/Adam of each model are the pnvo_train_* kernels; both losses and
    their gradients are HIP kernels (pnvo_mse_loss_coef, pnvo_geo_inverse_loss); index bookkeeping stays on the host as
    in the reference.

    steps: {act: VOTrainStep} with act in {-1 (all actions), MOVE_FORWARD, TURN_LEFT, TURN_RIGHT}."""

    def __init__(self, steps, invariance_types=("inverse_joint_train",), loss_inv_weight=1.0):
        self.steps = dict(steps)
        self.invariance_types = tuple(invariance_types)
        self.loss_inv_weight = float(loss_inv_weight)
        self.dev = next(iter(self.steps.values())).dev
        self.last_logs = {}

    def step(self, batch, actions, data_types, targets, loss_weights=None, dz_regress_masks=None):
        """batch: dict of NHWC device tensors [M,...] (the model's observation_pairs); actions [M] int; data_types [M]
        (CUR_REL_TO_PREV / PREV_REL_TO_CUR); targets [M,3] (dx, dz, dyaw).  Returns (total loss tensor [1], preds [M,3]
        in batch order)."""
        dev = self.dev
        actions = torch.as_tensor(actions).reshape(-1).to("cpu", torch.int64)
        data_types = torch.as_tensor(data_types).reshape(-1).to("cpu", torch.int64)
        targets = torch.as_tensor(targets, dtype=torch.float32).reshape(-1, 3).to(dev)
        M = actions.numel()
        joint = "inverse_joint_train" in self.invariance_types
        use_types = len(self.invariance_types) > 0
        allowed = [CUR_REL_TO_PREV] + ([PREV_REL_TO_CUR] if (joint or "inverse_data_augment_only" in self.invariance_types) else [])
        preds = torch.zeros((M, 3), device=dev, dtype=torch.float32)
        idx_of, grads, total = {}, {}, torch.zeros(1, device=dev)
        # Data parallel: which action models have entries on ANY rank this iteration.  Every rank must walk the same
        # sequence of collectives (forward_train: RunningMeanAndVar rounds; optimizer_step: the gradient all-reduce), so
        # the set of models that run is decided globally, not from this rank's batch.
        order = list(self.steps.keys())
        here = torch.tensor([1.0 if (act == -1 and M > 0) or bool((actions == act).any()) else 0.0 for act in order])
        anywhere = here.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            anywhere = anywhere.to(dev) if dist.get_backend() == "nccl" else anywhere
            dist.all_reduce(anywhere, op=dist.ReduceOp.MAX)
            anywhere = anywhere.cpu()
        present_anywhere = {act: bool(anywhere[i] > 0) for i, act in enumerate(order)}
        with torch.cuda.device(dev), torch.no_grad():
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for act, st in self.steps.items():
                idx = torch.arange(M) if act == -1 else torch.nonzero(actions == act, as_tuple=True)[0]
                if idx.numel() == 0:
                    if present_anywhere[act]:
                        st.participate_absent()          # same collectives as the ranks that hold entries of this action
                    continue
                di = idx.to(dev)
                sub = {k: v.index_select(0, di).contiguous() for k, v in batch.items()}
                out = st.forward_train(sub, actions[idx] if st.model.cfg.act_embed else None)
                preds.index_copy_(0, di, out)
                coef = regression_coef(
                    idx.numel(), data_types[idx] if use_types else None,
                    None if loss_weights is None else {k: torch.as_tensor(v).reshape(-1)[idx] for k, v in loss_weights.items()},
                    None if dz_regress_masks is None else torch.as_tensor(dz_regress_masks).reshape(-1)[idx], device=dev,
                    allowed_types=allowed if use_types else None)
                g = torch.empty_like(out)
                tgt = targets.index_select(0, di).contiguous()
                _lib.check(_lib.lib.pnvo_mse_loss_coef(_ptr(out), _ptr(tgt), _ptr(coef), int(out.numel()), _ptr(st._loss),
                                                       _ptr(g), stream))
                total += st._loss
                idx_of[act], grads[act] = di, g
            if joint:
                valid = torch.nonzero((actions == TURN_LEFT) | (actions == TURN_RIGHT), as_tuple=True)[0]
                if valid.numel():
                    vt = data_types[valid]
                    if valid.numel() % 2 or not bool((vt[0::2] == CUR_REL_TO_PREV).all() and (vt[1::2] == PREV_REL_TO_CUR).all()):
                        raise AssertionError("inverse_joint_train expects alternating (cur_rel_to_prev, prev_rel_to_cur) entries")
                    dv = valid.to(dev)
                    d = preds.index_select(0, dv).contiguous()
                    a32 = actions[valid].to(dev, torch.int32).contiguous()
                    out4 = torch.empty(4, device=dev)
                    ginv = torch.empty_like(d)
                    _lib.check(_lib.lib.pnvo_geo_inverse_loss(_ptr(d), _ptr(a32), int(d.shape[0]), MOVE_FORWARD,
                                                              C.c_float(self.loss_inv_weight), _ptr(out4), _ptr(ginv), stream))
                    total += out4[0:1]
                    self.last_logs = {"abs_diff_geo_inverse_rot": out4[1], "abs_diff_geo_inverse_pos": out4[2:4]}
                    full = torch.zeros_like(preds)
                    full.index_copy_(0, dv, ginv)
                    for act, di in idx_of.items():
                        grads[act] += full.index_select(0, di)
            for act, st in self.steps.items():                   # one fixed order on every rank: the bucket all-reduces of a
                if act in idx_of:                                #   model start inside its backward ...
                    st.backward(grads[act])
                elif present_anywhere[act]:                      #   ... and at the same point of the sequence on a rank without
                    st.absent_backward()                         #   entries for it (zero gradient)
            for act, st in self.steps.items():
                if act in idx_of or present_anywhere[act]:
                    st.optimizer_step()
                elif st.step_count > 0:
                    # the reference zero_grad()s and step()s EVERY action model each iteration (:855-901); a model without
                    # entries in this batch has all-zero gradients once it has had a backward, so Adam still decays and
                    # applies its moments (torch 1.x zero_grad keeps zero tensors, environment.yml)
                    st.grad.zero_()
                    st.optimizer_step(allreduce=False)           # no rank holds entries: the gradient is zero everywhere
        return total, preds
