"""The non-headline configurations of bench.py (`--config dual_bf16`, `--config train`): same timing contract — preheat,
W warm-up steps, EXACTLY K steps between (barrier + synchronize) brackets, max over ranks, ONE JSON line on rank 0 — on
    dual_bf16: BASELINE configs[2]  act_left_right_inv_joint, geometric-invariance dual forward, 256 pairs, bf16
    train    : BASELINE configs[3] per-GPU shape: 128 pairs, forward + backward + Adam (+ one flat-gradient all-reduce), fp32
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pointnav_vo_amd import model_spec as ms, parallel, synth, vo_cnn  # noqa: E402
from pointnav_vo_amd.registry import baseline_registry  # noqa: E402

# Algorithmic HBM bytes per pair of ONE model forward (fp32 layer streaming) and of the stem launch of the
# dual bf16 forward: the 30-channel fp32 observation tensors read once (7 856 640 B) + two bf16 stem outputs (96*171*32*2 B).
STREAM_BYTES_PER_PAIR = 22493692.0
STEM_DUAL_BYTES_PER_PAIR = 7856640.0 + 2 * 96 * 171 * 32 * 2


def _swapped(obs):
    return {k: np.concatenate([v[..., v.shape[-1] // 2:], v[..., : v.shape[-1] // 2]], axis=-1) for k, v in obs.items()}


def _cpu_baseline_dual(sds, ngroups, budget_s=12.0):
    """The oracle C port (fp32) running BOTH models on a bounded sample (the second on the swapped pair), one pair per host
    thread on every usable core (oracle.forward_pairs_parallel), median of three."""
    from oracle import oracle
    cores = oracle.usable_cores()
    distinct = 8
    obs = synth.make_obs_pairs(distinct, bench.H, bench.W, observation_space=bench.SPACE, dd_bins=bench.BINS, seed=100)
    sw = _swapped(obs)
    idx = [i % distinct for i in range(cores)]
    t0 = time.perf_counter()
    oracle.forward_pairs_parallel(sds[0], obs, ngroups=ngroups, threads=cores, indices=idx)
    t1 = time.perf_counter() - t0
    per_core = int(max(1, min(4, budget_s / 6.0 / max(t1, 1e-3))))
    idx = [i % distinct for i in range(cores * per_core)]
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.forward_pairs_parallel(sds[0], obs, ngroups=ngroups, threads=cores, indices=idx)
        oracle.forward_pairs_parallel(sds[1], sw, ngroups=ngroups, threads=cores, indices=idx)
        rates.append(len(idx) / (time.perf_counter() - t0))
    rates.sort()
    return {"value": rates[1], "unit": "frame-pairs/s (both models)", "cores": cores, "kind": "port", "host_cores": cores, "runs": rates,
            "sample": f"{len(idx)} pairs through both models (fp32 oracle C port, one pair per host thread on {cores} threads, median "
                      "of three; the reference has no bf16 CPU path)"}


def run_dual_bf16(args, rank, world, dist, dev, sync_all, obs=None, emit=True):
    """emit=False: called by the headline run for its `secondary` record — returns the result dictionary on rank 0 (None on
    the others) instead of printing it, re-uses the caller's observation tensors, skips the CPU baseline and leaves the
    process group alone."""
    B = args.batch or 256
    models, sds = [], []
    for seed in (0, 1):                               # the two action models of act_left_right_inv_joint
        m, sd = bench.build_model(dev, seed=seed)
        models.append(m.set_precision("bfloat16"))
        sds.append(sd)
    ma, mb = models
    if obs is None:
        obs = bench.make_inputs(B, dev, rank)
    step = lambda: vo_cnn.dual_forward(ma, mb, obs)

    with torch.no_grad():
        oa, ob = step()
        torch.cuda.synchronize(dev)
        err = None
        if rank == 0:
            from oracle import oracle
            nchk = 2
            host = {k: v[:nchk].cpu().numpy() for k, v in obs.items()}
            ra = oracle.forward(sds[0], host, ngroups=ma.cfg.ngroups, dtype=np.float64)
            rb = oracle.forward(sds[1], _swapped(host), ngroups=mb.cfg.ngroups, dtype=np.float64)
            ea = np.linalg.norm(oa[:nchk].cpu().numpy().astype(np.float64) - ra, axis=1)
            eb = np.linalg.norm(ob[:nchk].cpu().numpy().astype(np.float64) - rb, axis=1)
            err = {"abs_l2_model_a": float(ea.max()), "abs_l2_model_b": float(eb.max()),
                   "ref_l2": float(max(np.linalg.norm(ra, axis=1).max(), np.linalg.norm(rb, axis=1).max())),
                   "rms_abs_l2": float(np.sqrt(np.mean(np.concatenate([ea, eb]) ** 2))), "forwards": int(2 * nchk)}
        pre = (0.0, 0, None) if args.no_preheat else bench.preheat(step, dev)
        for _ in range(args.warmup):
            step()
        dt, per_step = bench.timed_steps(step, args.steps, sync_all, dev)
        ma.timing(True)
        for _ in range(args.steps):
            step()
        sync_all()
        kt = ma.timing_read()
        ma.timing(False)
    dt = parallel.max_over_ranks(dt, dev)
    # the same dual forward from the sensor frames (pnvo_forward_dual_raw): bit-identical outputs, 5.5x fewer input bytes
    raw_rec = None
    if not getattr(args, "no_secondary", False):
        with torch.no_grad():
            rgb_f, dep_f = bench.frames_of(obs)
            ra, rb = vo_cnn.dual_forward_raw(ma, mb, rgb_f, dep_f, obs["top_down_view"])
            same = bool(torch.equal(ra, oa) and torch.equal(rb, ob))
            t_raw = parallel.max_over_ranks(
                bench.time_steps_simple(lambda: vo_cnn.dual_forward_raw(ma, mb, rgb_f, dep_f, obs["top_down_view"]), 10, sync_all), dev)
            ma.timing(True)
            for _ in range(5):
                vo_cnn.dual_forward_raw(ma, mb, rgb_f, dep_f, obs["top_down_view"])
            sync_all()
            kr = ma.timing_read()
            ma.timing(False)
        stem_raw = [k for k in kr if k["name"] == "bf16:stem"]
        raw_rec = {"value": world * B / t_raw, "unit": "frame-pairs/s", "ms_per_step": 1e3 * t_raw, "steps": 10,
                   "bit_identical_outputs": same,
                   "stem_launch_ms": stem_raw[0]["total_ms"] / stem_raw[0]["launches"] if stem_raw else None,
                   "stem_algorithmic_gbs": (stem_raw[0]["bytes"] / (stem_raw[0]["total_ms"] * 1e-3) / 1e9) if stem_raw else None}
        del rgb_f, dep_f
    if rank == 0:
        value = world * B * args.steps / dt
        dom = max((k for k in kt if k["name"].startswith("bf16:")), key=lambda k: k["total_ms"])
        launch_ms = dom["total_ms"] / dom["launches"]
        is_stem = dom["name"] == "bf16:stem"
        alg_bytes = B * STEM_DUAL_BYTES_PER_PAIR if is_stem else dom["bytes"] / dom["launches"]
        ach = alg_bytes / (launch_ms * 1e-3) / 1e9
        traffic, note = bench.measured_traffic("dual_bf16", dom["name"], B)
        total_kernel_ms = sum(k["total_ms"] for k in kt)
        res = {
            "metric": "RGB-D frame-pair VO inferences/s @341x192 (geometric-invariance dual forward: two action models per pair)",
            "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: act_left_right_inv_joint dual forward (vo_cnn_rgb_d_dd_top_down x 2, second "
                                   "model on the channel-swapped pair), 341x192, bf16 operands / fp32 accumulation, seeded random "
                                   "weights", "pairs_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"dp{world} (independent pairs, no collective)"},
            "ms_per_step_events": per_step, "preheat_s": pre[0], "preheat_steps": pre[1], "preheat_converged": pre[2],
            "kernel_ms_per_step": total_kernel_ms / args.steps,
            "pose_abs_err_vs_fp64_oracle": err,
            "frac_hbm_streaming_model": value * 2 * STREAM_BYTES_PER_PAIR / 1e9 / (bench.PEAK_HBM_GBS * world),
            "roofline": {"kernel": dom["name"], "bound": "hbm", "achieved": ach, "peak": bench.PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": ach / bench.PEAK_HBM_GBS, "traffic": traffic, "traffic_note": note,
                         "definition": "algorithmic bytes of the launch (fp32 observation tensors read once + bf16 outputs of "
                                       "both models) / HIP-event launch duration",
                         "launch_ms": launch_ms, "share_of_kernel_time": dom["total_ms"] / total_kernel_ms},
            "kernels": sorted(({"name": k["name"], "ms_per_step": k["total_ms"] / args.steps, "launches": k["launches"] // args.steps,
                                "tflops": (k["flops"] / (k["total_ms"] * 1e-3) / 1e12) if k["flops"] else None,
                                "gbs": (k["bytes"] / (k["total_ms"] * 1e-3) / 1e9) if k["bytes"] else None}
                               for k in kt), key=lambda k: -k["ms_per_step"])[:40],
        }
        res["from_sensor_frames"] = raw_rec
        if not emit:
            return res
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = _cpu_baseline_dual(sds, ma.cfg.ngroups)
        print(json.dumps(res))
    if not emit:
        return None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def backend_is_nccl(args):
    return getattr(args, "backend", "nccl") == "nccl"


def run_train(args, rank, world, dist, dev, sync_all, obs=None, emit=True):
    from pointnav_vo_amd.train import VOTrainStep
    B = args.batch or 128
    model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=bench.SPACE, observation_size=(bench.W, bench.H), hidden_size=512, backbone="resnet18",
        normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=bench.BINS)   # reference p
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=0)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev)
    ts = VOTrainStep(model)
    ts.bucketed = not getattr(args, "no_overlap", False)     # A/B knob: gradient all-reduce in buckets behind the backward, or one flat one after it
    if obs is None:
        obs = bench.make_inputs(B, dev, rank)
    elif next(iter(obs.values())).shape[0] != B:
        obs = {k: v[:B].contiguous() for k, v in obs.items()}
    g = torch.Generator(device=dev)
    g.manual_seed(7 + rank)
    tgt = (torch.rand((B, 3), device=dev, generator=g) - 0.5) * 0.5
    losses = []

    def step():
        _, loss = ts.step(obs, tgt)
        losses.append(loss)

    pre = (0.0, 0, None) if args.no_preheat else bench.preheat(step, dev)
    for _ in range(args.warmup):
        step()
    dt, per_step = bench.timed_steps(step, args.steps, sync_all, dev)
    model.timing(True)
    for _ in range(args.steps):
        step()
    sync_all()
    kt = model.timing_read()
    model.timing(False)
    dt_local = dt
    dt = parallel.max_over_ranks(dt, dev)
    multi = None
    if world > 1 and dist is not None:
        # N > 1: every rank's own step time, the OTHER gradient all-reduce schedule in the same run (A/B: buckets behind the backward
        # vs one flat buffer after it), and the flat 15.85 MB all-reduce alone — their difference is what the overlap hides
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt_local], dtype=torch.float64, device=cdev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank_ms = [1e3 * float(x.item()) / args.steps for x in parts]
        ts.bucketed = not ts.bucketed
        for _ in range(2):
            step()
        dt_other, _ = bench.timed_steps(step, args.steps, sync_all, dev)
        dt_other = parallel.max_over_ranks(dt_other, dev)
        ts.bucketed = not ts.bucketed
        flat = ts.grad.clone() if backend_is_nccl(args) else ts.grad.detach().cpu()
        for _ in range(2):
            dist.all_reduce(flat)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(flat)
        sync_all()
        ar_ms = parallel.max_over_ranks((time.perf_counter() - t0) / 10, dev) * 1e3
        this, other = ("bucketed", "flat") if ts.bucketed else ("flat", "bucketed")
        multi = {"per_rank_ms_per_step": per_rank_ms, "slowest_rank": int(np.argmax(per_rank_ms)),
                 f"ms_per_step_{this}": 1e3 * dt / args.steps, f"ms_per_step_{other}": 1e3 * dt_other / args.steps,
                 "allreduce_alone_ms": ar_ms, "allreduce_bytes": int(ts.grad.numel() * 4),
                 "exposed_allreduce_ms_estimate": max(0.0, 1e3 * (dt if this == "bucketed" else dt_other) / args.steps
                                                      - (1e3 * (dt if this == "flat" else dt_other) / args.steps - ar_ms)),
                 "note": "flat = backward, then ONE all-reduce of the whole gradient; bucketed = three ranges reduced on a communication "
                         "stream while the backward still runs; exposed estimate = bucketed step - (flat step - all-reduce alone)"}
    if rank == 0:
        value = world * B * args.steps / dt
        flops = 3 * 2.0 * ms.macs_per_pair(model.cfg)
        agg = {}
        for k in kt:
            key = k["name"].split(":")[0]
            agg[key] = agg.get(key, 0.0) + k["total_ms"] / args.steps
        dom = max((k for k in kt if k["flops"]), key=lambda k: k["total_ms"])
        launch_ms = dom["total_ms"] / dom["launches"]
        alg = dom["flops"] / dom["launches"] / (launch_ms * 1e-3) / 1e12
        stem_mx = dom["name"].endswith("conv1.0.weight") and model.get_option("wgrad_stem") == "mx"
        if stem_mx:
            # the stem's weight gradient runs on the bf16 matrix cores (wgrad_stem_mx.hip): EXECUTED work = tiles (6 x 13 outputs)
            # x 49 taps x 3 K-chunks x 18 v_mfma_f32_16x16x32_bf16 (3 M-tiles x 2 N-tiles x 3 dY pieces) of 16384 FLOP
            tiles = B * ((bench.H // 2 + 5) // 6) * ((bench.W // 2 + 1 + 12) // 13)
            ach, peak, pipe = tiles * 49 * 3 * 18 * 16384 / (launch_ms * 1e-3) / 1e12, bench.PEAK_BF16_TFLOPS, "bf16 MFMA (exact three-piece operands)"
        else:
            ach, peak, pipe = alg, bench.PEAK_FP32_TFLOPS, "fp32 MFMA"
        lv = [float(x) for x in losses]
        res = {
            "metric": "VO training step (fwd+bwd+Adam) frame-pairs/s @341x192", "value": value, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3] per-GPU shape: VO training step of vo_cnn_rgb_d_dd_top_down, 128 pairs per GPU, "
                                   "341x192, fp32, dropout 0.2, Adam lr 2.5e-4; N > 1: one flat 15.85 MB gradient all-reduce (RCCL)",
                       "pairs_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}"},
            "ms_per_step_events": per_step, "preheat_s": pre[0], "preheat_steps": pre[1], "preheat_converged": pre[2],
            "loss_first_last": [lv[0], lv[-1]], "tflops_3x_fwd": value * flops / 1e12,
            "frac_fp32_peak_3x_fwd": value * flops / 1e12 / (bench.PEAK_FP32_TFLOPS * world),
            "roofline": {"kernel": dom["name"], "bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s", "frac": alg / peak,
                         "pipe": pipe, "definition": "SURVEY 8(d): ALGORITHMIC FLOPs of the layer's weight gradient per launch / HIP-event "
                                                     "launch duration / dense peak of the pipe the kernel runs on",
                         "executed_tflops": ach, "executed_frac": ach / peak, "traffic": None, "launch_ms": launch_ms},
            "ms_by_kernel_class": agg,
            "kernels": sorted(({"name": k["name"], "ms_per_step": k["total_ms"] / args.steps, "launches": k["launches"] // args.steps,
                                "tflops": (k["flops"] / (k["total_ms"] * 1e-3) / 1e12) if k["flops"] else None,
                                "gbs": (k["bytes"] / (k["total_ms"] * 1e-3) / 1e9) if k["bytes"] else None}
                               for k in kt), key=lambda k: -k["ms_per_step"])[:70],
        }
        res["config"]["gradient_allreduce"] = "bucketed behind the backward" if ts.bucketed else "one flat buffer after the backward"
        res["rccl_ranks"] = bench.process_group_record(dist, world)
        if multi is not None:
            res["multi_gpu"] = multi
        if not emit:
            return res
        print(json.dumps(res))
    if not emit:
        return None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
