#!/usr/bin/env python
"""bench.py — RGB-D frame-pair VO inferences/s at 341x192 on N x MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the VO hot path — VisualOdometryCNNBase.forward of `vo_cnn_rgb_d_dd_top_down`
(act_forward model, fp32) — over one batch of 256 synthetic frame pairs whose observation tensors are already
resident in HBM (BASELINE.json configs[1]).  Data parallel: every rank owns its own 256 pairs (weak scaling), no
data-path collective (pairs are independent at inference).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointnav_vo_amd import model_spec as ms  # noqa: E402
from pointnav_vo_amd import parallel, synth  # noqa: E402
from pointnav_vo_amd.registry import baseline_registry  # noqa: E402
from pointnav_vo_amd.trainer import NormalizedDepth2TopDownViewHabitatTorch  # noqa: E402
from pointnav_vo_amd import _lib  # noqa: E402
import ctypes as C  # noqa: E402

W, H, BINS = 341, 192, 10
SPACE = ["rgb", "depth", "discretized_depth", "top_down_view"]
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA (= vector) peak
PEAK_BF16_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
ORACLE_PAIRS = (0, 1, 63, 127, 128, 190, 254, 255)   # pairs of the headline batch checked against the fp64 oracle
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")   # written by tools/round_profile.sh (PMC passes)


# the sources that define the kernels whose HBM traffic is recorded (both records are stem launches): a record measured on other
# versions of THESE files is reported as stale; edits elsewhere (host API, other kernels) do not touch what was measured
TRAFFIC_SOURCES = ("stem_rs.hip", "stem_mx.hip", "stem_tile.h", "pnvo_internal.h")


def source_hash():
    """sha256[:12] over the sources of the recorded kernels (TRAFFIC_SOURCES)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pointnav-vo_amd", "csrc")
    for f in TRAFFIC_SOURCES:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def measured_traffic(config, kernel, batch):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this command (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, MI355X_MICROARCH.md section HBM), as recorded by tools/round_profile.sh; None when no record matches
    the current kernel sources, configuration and batch."""
    try:
        rec = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    e = rec.get(config, {}).get(kernel)
    if not e or e.get("batch") != batch:
        return None, "no PMC record for this kernel / batch"
    if e.get("source_hash") != source_hash():
        return None, f"stale PMC record (kernel sources changed since {e.get('source_hash')})"
    return float(e["bytes_per_launch"]), f"rocprofv3 PMC, {e.get('file', 'profiles/')}"


def rocprof_duration(config, kernel, batch):
    """Average / minimum duration (ms) of `kernel` in the rocprofv3 --kernel-trace --stats pass of this command recorded by
    tools/round_profile.sh next to the PMC traffic (same staleness rule), or None."""
    try:
        e = json.load(open(TRAFFIC_FILE)).get(config, {}).get(kernel)
    except (OSError, ValueError):
        return None
    if not e or e.get("batch") != batch or e.get("source_hash") != source_hash() or "rocprof_avg_us" not in e:
        return None
    return {"avg_ms": e["rocprof_avg_us"] * 1e-3, "min_ms": e["rocprof_min_us"] * 1e-3, "calls": e.get("rocprof_calls"),
            "file": "profiles/" + str(e.get("rocprof_file"))}


def build_model(dev, seed=0):
    model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=SPACE, observation_size=(W, H), hidden_size=512, backbone="resnet18",
        normalize_visual_inputs=True, output_dim=3, dropout_p=0.2, discretized_depth_channels=BINS)
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=seed)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return model.to(dev).eval(), sd


def make_inputs(B, dev, rank):
    """Synthetic pairs generated ON DEVICE: rgb U{0..255}, depth U[0.05,0.95] rounded through fp16,
    discretized_depth / top_down_view produced from that depth by this build's own pre-processing kernels."""
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    rgb = torch.randint(0, 256, (B, H, W, 6), device=dev, generator=g, dtype=torch.int32).to(torch.float32)
    depth = (torch.rand((B, H, W, 2), device=dev, generator=g) * 0.9 + 0.05).to(torch.float16).to(torch.float32)
    dd = torch.empty((B, H, W, 2 * BINS), device=dev, dtype=torch.float32)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for k in range(2):
        _lib.check(_lib.lib.pnvo_discretize_depth(C.c_void_p(depth.data_ptr() + 4 * k), B * H * W, 2, BINS,
                                                  C.c_void_p(dd.data_ptr() + 4 * k * BINS), 2 * BINS, None, stream))
    gen = NormalizedDepth2TopDownViewHabitatTorch(min_depth=0.1, max_depth=10.0, vis_size_h=H, vis_size_w=W, hfov_rad=70)
    tdv = torch.empty((B, H, W, 2), device=dev, dtype=torch.float32)
    for k in range(2):
        gen.gen_top_down_view_batch(depth[..., k], out=tdv, out_channel=k)
    return {"rgb": rgb, "depth": depth, "discretized_depth": dd, "top_down_view": tdv}


def cpu_baseline(sd, ngroups, budget_s=15.0):
    """The oracle (CPU port of the reference forward, fp32) on a bounded sample of the same workload, parallelised over PAIRS:
    one whole single-pair forward per host thread on every usable core (pairs are independent), median of three runs."""
    from oracle import oracle
    cores = oracle.usable_cores()
    distinct = 16
    obs = synth.make_obs_pairs(distinct, H, W, observation_space=SPACE, dd_bins=BINS, seed=100)
    t0 = time.perf_counter()
    oracle.forward_pairs_parallel(sd, obs, ngroups=ngroups, threads=cores, indices=[i % distinct for i in range(cores)])   # warm-up
    t1 = time.perf_counter() - t0                                              # one pair per core
    per_core = int(max(1, min(8, budget_s / 3.0 / max(t1, 1e-3))))
    n = cores * per_core
    idx = [i % distinct for i in range(n)]
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.forward_pairs_parallel(sd, obs, ngroups=ngroups, threads=cores, indices=idx)
        rates.append(n / (time.perf_counter() - t0))
    rates.sort()
    return {"value": rates[1], "unit": "frame-pairs/s", "cores": cores, "kind": "port", "host_cores": cores,
            "runs": rates, "per_core": rates[1] / cores,
            "sample": f"{n} single-pair fp32 forwards of the oracle C port ({distinct} distinct pairs), one pair per host thread on "
                      f"{cores} threads (= every usable core; each thread an OpenMP team of one), median of three runs"}


def preheat(step, dev, cap_s=2.0, tol=0.02):
    """Run steps until three consecutive step times agree within `tol` (clock ramp / first-touch effects of a cold box
    are over) or `cap_s` seconds have passed.  Reported as `preheat_s` / `preheat_steps`; never part of the timed region."""
    t_start = time.perf_counter()
    hist = []
    while True:
        t0 = time.perf_counter()
        step()
        if torch.device(dev).type == "cuda":
            torch.cuda.synchronize(dev)
        hist.append(time.perf_counter() - t0)
        el = time.perf_counter() - t_start
        ok = len(hist) >= 3 and max(hist[-3:]) <= (1.0 + tol) * min(hist[-3:])
        stop = ok or el >= cap_s
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # every rank leaves after the same number of steps (a step may contain collectives): stop when ALL ranks want to
            flag = torch.tensor([0.0 if stop else 1.0, 0.0 if ok else 1.0], device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            stop, ok = flag[0].item() == 0.0, flag[1].item() == 0.0
        if stop:
            return el, len(hist), ok


def timed_steps(step, steps, sync_all, dev):
    """EXACTLY `steps` calls between two (barrier + synchronize) brackets -> wall seconds; one event per step boundary on
    the launch stream gives the per-step distribution without serialising anything."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    # untimed steps bracketed by events first: on a cold box the first event-recorded step of a process sometimes takes
    # ~50 ms of host time (seen as `slowest_step: 0` before this line existed) — first-use costs belong to the warm-up
    # (and on the first process of a fresh box a training step ~10 steps into the run stalls 30-40 ms once: three of them here)
    for i in range(min(3, steps)):
        ev[i].record()
        step()
        ev[i + 1].record()
    sync_all()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    raw = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    per = sorted(raw)
    out = {"p50": per[len(per) // 2], "min": per[0], "max": per[-1]}
    if per[-1] > 1.5 * per[len(per) // 2]:                 # an outlier step: say which one (a one-off stall shows, it is not hidden)
        out["slowest_step"] = raw.index(per[-1])
    return dt, out


def multi_gpu_report(model, out_local, dt_local, steps, B, rank, world, dist, dev, backend):
    """What makes an N > 1 line self-explaining: every rank's own step time (the value uses their MAX), and a proof that sharding
    changed nothing — rank 0 re-generates every rank's shard (inputs depend only on the rank's seed), runs it alone, and compares with
    what that rank computed: bit-identical by construction (fixed-order reductions, no data-path collective)."""
    cdev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([dt_local], dtype=torch.float64, device=cdev)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    per_rank_ms = [1e3 * float(x.item()) / steps for x in parts]
    o = out_local.detach().to(cdev).contiguous()
    outs = [torch.zeros_like(o) for _ in range(world)]
    dist.all_gather(outs, o)
    rep = {"per_rank_ms_per_step": per_rank_ms, "slowest_rank": int(np.argmax(per_rank_ms)),
           "spread": (max(per_rank_ms) - min(per_rank_ms)) / max(per_rank_ms),
           "collectives_in_the_timed_region": "none (independent pairs); barrier + MAX over ranks around it"}
    if rank == 0:
        same, worst = [], 0.0
        with torch.no_grad():
            for r in range(world):
                mine = model(make_inputs(B, dev, r)).to(cdev)
                same.append(bool(torch.equal(mine, outs[r])))
                worst = max(worst, float((mine - outs[r]).abs().max()))
        rep["shards_equal_single_gpu"] = all(same)
        rep["shards_checked"] = world
        rep["max_abs_diff_vs_single_gpu"] = worst
    return rep


def arithmetic_text(model):
    """What the measured forward multiplied with, read from the handle's options (not a fixed string)."""
    conv, stem, pieces = model.get_option("conv"), model.get_option("stem"), model.get_option("pieces")
    if conv in ("auto", "x3") and stem in ("auto", "mx"):
        if pieces == "2":
            how = ("every float32 operand of the stem and of the sixteen 3x3 convs is split into TWO float16 pieces (22-bit operands; "
                   "weights pre-scaled by a power of two per tensor, undone exactly on the accumulators) and multiplied on the float16 "
                   "matrix cores: a*w ~ a0*w0 + a0*w1 + a1*w0, every kept term exact in the float32 accumulator, the dropped a1*w1 "
                   "below 2^-22 of the product (relative error per product <= 3 * 2^-22 = 7e-7); raw stem inputs (integers <= 255, "
                   "{0,1}) are exact in float16, only the stem weight and the float-valued channels are split there; float32 "
                   "accumulation.  NARROWER than the reference's fp32 multiply: `secondary.fwd_fp32_pieces3` (three exact bf16 "
                   "pieces, products exact or within 2^-23) and `secondary.fwd_fp32_pipe` (every conv on the fp32 MFMA pipe) are the "
                   "strict forms, measured in the same run with their own error against the fp64 oracle")
        else:
            how = ("every float32 operand of the stem and of the sixteen 3x3 convs is split into THREE bf16 pieces (hi + mid + lo == the "
                   "float32 value) and multiplied on the bf16 matrix cores: products exact (stem: inputs exact in bf16) or within 2^-23 "
                   "(six of the nine cross terms), float32 accumulation")
        return "float32 activations, weights, accumulators and results; " + how + "; options conv=fp32 / stem=dense select the fp32-MFMA kernels"
    return f"float32 (options conv={conv}, stem={stem}, pieces={pieces}): fp32-MFMA convs where conv=fp32, dense fp32 stem where stem=dense"


def strict_variant(model, obs, ref, chk, opts, B, world, sync_all, dev, note):
    """10 timed steps of the headline batch with `opts` set on the handle (the strict-arithmetic forms), their own error against the
    fp64 oracle on the headline's checked pairs; options restored afterwards."""
    saved = {k: model.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            model.set_option(k, v)
        with torch.no_grad():
            o = model(obs)
            t = parallel.max_over_ranks(time_steps_simple(lambda: model(obs), 10, sync_all), dev)
        rec = {"workload": note, "options": dict(opts), "value": world * B / t, "unit": "frame-pairs/s", "ms_per_step": 1e3 * t, "steps": 10}
        if ref is not None:
            oc = o[chk].cpu().numpy().astype(np.float64)
            rec["pose_rel_err_vs_fp64_oracle"] = float((np.linalg.norm(oc - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)).max())
        return rec
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        for k, v in saved.items():
            model.set_option(k, v.split(" ")[0])


def stem_executed(kt_entry, B, per_launch_ms, sel="auto", pieces="2", form="auto"):
    """Work the stem kernels EXECUTE (not the algorithmic 30-channel conv): matrix-core FLOPs per launch against the
    peak of the pipe they run on."""
    ho, wo = (H + 1) // 2, (W + 1) // 2
    px = B * (-(-ho // 8)) * (-(-wo // 16)) * 128              # pixels of the 8x16 tiles, padding included
    if sel in ("auto", "mx") and pieces == "2":   # stem_mx.hip / stem_rs.hip, float16 pieces: 5 x v_mfma_f32_32x32x16_f16 per tap and
        per_tile = 4 * 49 * 5                                 #   32-pixel tile (2 weight pieces x 2 K-chunks + 1 chunk of remainders)
        if form in ("auto", "fast") and B * (-(-ho // 8)) * (-(-wo // 16)) >= 4 * 256:
            per_tile = 4 * (12 * 16 + 3 * 4 + 5)              # stem_rs FAST: the remainder MFMAs of four taps share a chunk: 836, not 980
        flops = px / 128 * per_tile * (2.0 * 32 * 32 * 16)
        peak, pipe = PEAK_BF16_TFLOPS, "float16 MFMA (two float16 weight pieces, inputs exact in float16 -> float32-grade results)"
    elif sel in ("auto", "mx"):        # stem_mx.hip: 7 x v_mfma_f32_32x32x16_bf16 per tap and 32-pixel tile (3 weight pieces x 2
        flops = px / 32 * 49 * 7 * (2.0 * 32 * 32 * 16)       #   K-chunks + 1 chunk of float-modality remainders)
        peak, pipe = PEAK_BF16_TFLOPS, "bf16 MFMA (three exact bf16 weight pieces -> float32 results)"
    elif sel == "dd":        # stem_dd.hip: K = 12 of 30 channels on the fp32 MFMA pipe, the rest gathered from LDS
        flops = 2.0 * px * 32 * 12 * 49
        peak, pipe = PEAK_FP32_TFLOPS, "fp32 MFMA at K = 12 (+ LDS gather of the one-hot channels)"
    else:                    # stem_lds.hip: all 32 padded channels on the fp32 MFMA pipe
        flops = 2.0 * px * 32 * 32 * 49
        peak, pipe = PEAK_FP32_TFLOPS, "fp32 MFMA"
    tf = flops / (per_launch_ms * 1e-3) / 1e12
    return tf, peak, pipe


def fast_form_layers(model, B):
    """Which of the 17 layers the headline's arithmetic note is about (the stem + the sixteen 3x3 convs of the residual stages) run on
    the two-float16-piece form at batch B.  A layer leaves it for three exact bf16 pieces when a GroupNorm-derived bound on its
    input reaches 6e4 (`range_guarded`); small launches (deep stages at small batches) run on the fp32 MFMA pipe (`on_fp32_pipe`,
    exact fp32 arithmetic): pnvo_layer_kernel."""
    names = [f"visual_encoder.backbone.layer{s}.{b}.convs.{c}" for s in (1, 2, 3, 4) for b in (0, 1) for c in (0, 3)]
    fam = {n: model.layer_kernel(n, B)[0] for n in names}
    stem_fast = model.get_option("pieces") == "2" and model.get_option("stem") in ("auto", "mx")
    fast = int(stem_fast) + sum(1 for f in fam.values() if f == "x2")
    return {"layers_on_fast_form": f"{fast}/{1 + len(names)}", "stem": "two float16 pieces" if stem_fast else "other",
            "range_guarded": sorted(n for n, f in fam.items() if f == "x3"),
            "on_fp32_pipe": sorted(n for n, f in fam.items() if f not in ("x2", "x3"))}


def frames_of(obs):
    """The sensor frames behind synthetic observation pairs: uint8 rgb [B,2,H,W,3], float32 depth [B,2,H,W] (the pair tensors
    hold [prev | cur] on the channel axis)."""
    B = obs["depth"].shape[0]
    rgb = obs["rgb"].to(torch.uint8).reshape(B, H, W, 2, 3).permute(0, 3, 1, 2, 4).contiguous()
    dep = obs["depth"].permute(0, 3, 1, 2).contiguous()
    return rgb, dep


def time_steps_simple(step, steps, sync_all):
    for _ in range(3):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    return (time.perf_counter() - t0) / steps


def slim_secondary(name, full):
    """The `secondary` entry of the headline line: the figures of a 10-step run of another BASELINE configuration."""
    if full is None or "error" in full:
        return full
    rf = full["roofline"]
    out = {"workload": full["config"]["workload"], "value": full["value"], "unit": full["unit"], "ms_per_step": full["ms_per_step"],
           "steps": full["steps"], "dtype": full["dtype"], "pairs_per_gpu": full["config"]["pairs_per_gpu"],
           "ms_per_step_events": full["ms_per_step_events"],
           "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "pipe")}}
    if name == "dual_bf16":
        e = full["pose_abs_err_vs_fp64_oracle"]
        out["from_sensor_frames"] = full.get("from_sensor_frames")
        out["rms_err_vs_fp64"] = e and e["rms_abs_l2"]
        out["err_note"] = e and f"RMS over {e['forwards']} forwards of ||out - ref||_2, reference norm up to {e['ref_l2']:.2f}"
    else:
        out["loss_first_last"] = full["loss_first_last"]
        out["ms_by_kernel_class"] = full["ms_by_kernel_class"]
        out["gradient_allreduce"] = full["config"].get("gradient_allreduce")
    return out


def self_launch(n, port=0):
    """Re-exec this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on 127.0.0.1 (one process
    per GPU); rank 0 of the children prints the JSON line on the inherited stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    argv, skip = [], False
    for a in sys.argv[1:]:                                   # the children get this command line minus --master-port
        if skip or a.startswith("--master-port="):
            skip = False
        elif a == "--master-port":
            skip = True
        else:
            argv.append(a)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.exit(rc)
    return rc


def process_group_record(dist, world):
    """What the driver needs to see N ranks: the process group's own size and backend (`nccl` IS RCCL on ROCm)."""
    if dist is None or not dist.is_initialized():
        return {"world_size": 1, "backend": None}
    be = dist.get_backend()
    return {"world_size": dist.get_world_size(), "backend": be, "is_rccl": be == "nccl", "launched_by": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frame pairs per GPU per step (default: the config's)")
    ap.add_argument("--config", default="fwd_fp32", choices=["fwd_fp32", "dual_bf16", "train"],
                    help="fwd_fp32 = BASELINE configs[1] (the headline); dual_bf16 = configs[2]; train = configs[3] shape")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N>1 launch (nccl = RCCL; gloo only exercises the host logic)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: run the N-rank control flow (barriers, max-over-ranks, JSON) with a stub step")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="testing aid for 1-GPU boxes: every rank uses cuda:0 (with --backend gloo), so the N > 1 code path runs "
                         "on the real kernels; the value it prints is not a scaling measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline run only: skip the 10-step dual_bf16 / train measurements reported under `secondary`")
    ap.add_argument("--no-overlap", action="store_true",
                    help="--config train: ONE flat gradient all-reduce after the backward instead of buckets behind it (A/B)")
    ap.add_argument("--no-preheat", action="store_true")
    ap.add_argument("--secondary-deadline", type=float, default=240.0,
                    help="N > 1: seconds the side measurements (dual_bf16, train with its all-reduce, ...) may take before every rank gives up "
                         "on them and rank 0 prints the headline-only line")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = a free one)")
    args = ap.parse_args()

    if args.gpus > 1 and not args.dry_run and not args.shared_gpu and args.backend == "nccl":
        have = torch.cuda.device_count()
        if args.gpus > have:                   # one line instead of N rank tracebacks (every rank and the launcher check the same thing)
            if int(os.environ.get("RANK", "0")) == 0:
                print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (torch.cuda.device_count()); "
                      "nothing was run", file=sys.stderr, flush=True)
            sys.exit(2)
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (the reference's launch.py:9-32 shells out to
        # torch.distributed.launch the same way); the ranks re-enter this file with RANK / WORLD_SIZE set
        return self_launch(args.gpus, args.master_port)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.shared_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: one process per GPU over RCCL
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run:
        return dry_run(args, rank, world, dist)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.config == "dual_bf16":
        from tools import bench_configs
        return bench_configs.run_dual_bf16(args, rank, world, dist, dev, sync_all)
    if args.config == "train":
        from tools import bench_configs
        return bench_configs.run_train(args, rank, world, dist, dev, sync_all)

    model, sd = build_model(dev)
    B = args.batch or 256
    obs = make_inputs(B, dev, rank)
    step = lambda: model(obs)                      # one pass of the product path as a caller makes it

    with torch.no_grad():
        # parity on the first pairs of this rank's batch (fp64 oracle on the same tensors)
        out = model(obs)
        torch.cuda.synchronize(dev)
        rel = ref = chk = None
        if rank == 0:
            from oracle import oracle
            chk = sorted({i for i in ORACLE_PAIRS if i < B} | {0, B - 1})     # first / last tile of the batch, mid-batch
            ref = oracle.forward(sd, {k: v[chk].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups,
                                 dtype=np.float64)
            o = out[chk].cpu().numpy().astype(np.float64)
            rel = float((np.linalg.norm(o - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)).max())

        pre = (0.0, 0, None) if args.no_preheat else preheat(step, dev)
        for _ in range(args.warmup):
            step()
        dt, per_step = timed_steps(step, args.steps, sync_all, dev)      # ---- the timed region
        # ---- the same K steps once more with a HIP-event pair around every launch (on the launch stream): per-kernel
        #      durations for `roofline` / `kernels`.  Not part of `value`: event pairs serialise the launches.
        model.timing(True)
        for _ in range(args.steps):
            step()
        sync_all()
        kt = model.timing_read()
        model.timing(False)

    dt_local = dt
    dt = parallel.max_over_ranks(dt, dev)     # slowest rank
    multi = None
    if world > 1:
        multi = multi_gpu_report(model, out, dt_local, args.steps, B, rank, world, dist, dev, args.backend)

    res = None
    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        flops_pair = 2.0 * ms.macs_per_pair(model.cfg)
        bytes_pair = float(ms.streaming_bytes_per_pair(model.cfg))
        dom = max((k for k in kt if k["name"].startswith("conv:")), key=lambda k: k["total_ms"])
        per_launch_ms = dom["total_ms"] / dom["launches"]
        alg = dom["flops"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e12      # algorithmic FLOPs of the layer
        total_kernel_ms = sum(k["total_ms"] for k in kt)
        is_stem = dom["name"].endswith("conv1.0")
        if is_stem:
            ach, peak, pipe = stem_executed(dom, B, per_launch_ms, model.get_option("stem"), model.get_option("pieces"), model.get_option("stem_form"))
        else:
            fam, ex = model.layer_kernel(dom["name"][len("conv:"):], B)
            if fam in ("x3", "x2"):                  # six bf16 / three float16 MFMA terms per float32 product, tile padding included
                ach, peak = ex / (per_launch_ms * 1e-3) / 1e12, PEAK_BF16_TFLOPS
                pipe = ("bf16 MFMA (three-piece operands, six terms per product)" if fam == "x3" else
                        "float16 MFMA (two-piece operands, three terms per product)")
            else:                                    # the fp32-MFMA kernels execute the algorithmic FLOPs
                ach, peak, pipe = alg, PEAK_FP32_TFLOPS, "fp32 MFMA"
        traffic, traffic_note = measured_traffic("fwd_fp32", dom["name"], B)
        res = {
            "metric": "RGB-D frame-pair VO inferences/s @341x192", "value": value, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "float32 storage, accumulation and results; what the multipliers see is stated under `arithmetic`",
            "config": {"workload": "BASELINE configs[1]: act_forward VO inference (vo_cnn_rgb_d_dd_top_down, 30 input "
                                   "channels), 341x192, fp32, seeded random weights", "pairs_per_gpu": B,
                       "global_batch": world * B, "parallelism": f"dp{world} (independent pairs, no collective)"},
            "ms_per_step_events": per_step,
            "preheat_s": pre[0], "preheat_steps": pre[1], "preheat_converged": pre[2],
            "kernel_ms_per_step": total_kernel_ms / args.steps,
            "pose_rel_err_vs_fp64_oracle": rel, "oracle_checked_pairs": chk,
            "arithmetic": arithmetic_text(model),
            **{k: v for k, v in fast_form_layers(model, B).items() if k != "stem"},
            "model_tflops": value * flops_pair / 1e12,
            "frac_fp32_peak_whole_path_algorithmic": value * flops_pair / 1e12 / (PEAK_FP32_TFLOPS * world),
            "frac_hbm_streaming_model": value * bytes_pair / 1e9 / (PEAK_HBM_GBS * world),
            "roofline": {"kernel": dom["name"], "bound": "mfma", "achieved": alg, "peak": peak, "unit": "TFLOP/s",
                         "frac": alg / peak, "pipe": pipe,
                         "definition": "SURVEY 8(d): ALGORITHMIC FLOPs of the layer per launch (2 x MACs of the 30-channel fp32 conv x pairs) "
                                       "/ HIP-event launch duration / dense peak of the pipe the kernel runs on",
                         "executed_tflops": ach, "executed_frac": ach / peak,
                         "executed_note": "matrix-core FLOPs the kernel issues (operand splitting: 2-3 MFMA terms per float32 product, "
                                          "K and tile padding) / the same duration / the same peak",
                         "frac_of_fp32_pipe_peak": alg / PEAK_FP32_TFLOPS,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "launch_ms": per_launch_ms, "launch_ms_source": "HIP events on the launch stream, this run",
                         # the committed rocprofv3 --kernel-trace --stats pass of the same command (another box, another day: the
                         # stem launch alone varies 0.73-0.83 ms between boxes) — both durations in the record, `frac` uses the live one
                         "rocprof": rocprof_duration("fwd_fp32", dom["name"], B),
                         "share_of_kernel_time": dom["total_ms"] / total_kernel_ms},
            "kernels": sorted(({"name": k["name"], "ms_per_step": k["total_ms"] / args.steps,
                                "tflops": (k["flops"] / (k["total_ms"] * 1e-3) / 1e12) if k["flops"] else None,
                                "gbs": (k["bytes"] / (k["total_ms"] * 1e-3) / 1e9) if k["bytes"] else None}
                               for k in kt), key=lambda k: -k["ms_per_step"])[:40],
        }
        res["rccl_ranks"] = process_group_record(dist, world)
        if multi is not None:
            res["multi_gpu"] = multi

    # N > 1: the side measurements below walk collectives that no run has ever executed on RCCL hardware.  A rank that hangs in one of
    # them must not cost the scaling run its headline: after `--secondary-deadline` seconds every rank leaves on its own, rank 0 with
    # the headline-only line (marked).  N = 1 has no collectives and no deadline.
    import threading
    watchdog = None
    emit_lock = threading.Lock()              # exactly ONE JSON line: whoever takes the lock first (main path or deadline) prints
    emitted = [False]
    if world > 1 and not args.no_secondary:
        def bail():
            with emit_lock:
                if emitted[0]:                # the main path is printing / has printed: nothing to give up on
                    return
                emitted[0] = True
                if rank == 0:
                    res["secondary"] = {"error": f"side measurements did not finish within {args.secondary_deadline:.0f} s at {world} ranks "
                                                 "(a collective did not return); headline only"}
                    print(json.dumps(res), flush=True)
                else:
                    print(f"bench.py: rank {rank} gave up on the side measurements at the deadline (headline is rank 0's line)",
                          file=sys.stderr, flush=True)
                # every rank exits 0: a non-zero rank would make torch.distributed.run tear rank 0 down, possibly before its line is out,
                # and fail a run whose headline WAS measured; the record itself carries the mark (`secondary.error`)
                os._exit(0)
        watchdog = threading.Timer(args.secondary_deadline, bail)
        watchdog.daemon = True
        watchdog.start()

    # ---- BASELINE configs[2] and configs[3] (per-GPU shape) next to the headline: 10 timed steps each under the same contract
    #      (tools/bench_configs.py), on the headline's own observation tensors; AFTER the headline's timed region, never in it
    secondary = None
    raw_rec = None
    strict = None
    if not args.no_secondary:
        # the same forward from the SENSOR frames (pnvo_forward_raw: no observation-pair tensors): bit-identical results
        with torch.no_grad():
            rgb_f, dep_f = frames_of(obs)
            o_raw = model.forward_raw(rgb_f, dep_f, obs["top_down_view"])
            same = bool(torch.equal(o_raw, out))
            t_raw = parallel.max_over_ranks(time_steps_simple(lambda: model.forward_raw(rgb_f, dep_f, obs["top_down_view"]), 10, sync_all), dev)
        raw_rec = {"workload": "the headline batch handed over as sensor frames (uint8 rgb + float32 depth + top-down view: 1.44 MB per "
                               "pair instead of 7.86 MB of float32 observation tensors)", "value": world * B / t_raw,
                   "unit": "frame-pairs/s", "ms_per_step": 1e3 * t_raw, "steps": 10, "bit_identical_to_headline_outputs": same}
        del rgb_f, dep_f
        # the strict forms of the same forward, beside the headline: three exact bf16 pieces (round 2's arithmetic) and the fp32 pipe
        strict = {
            "fwd_fp32_pieces3": strict_variant(model, obs, ref if rank == 0 else None, chk if rank == 0 else None, {"pieces": "3"}, B, world,
                                               sync_all, dev, "the headline batch with option pieces=3: three exact bf16 pieces per float32 "
                                               "operand, six MFMA terms per product (stem: 7 MFMAs per tap)"),
            "fwd_fp32_pipe": strict_variant(model, obs, ref if rank == 0 else None, chk if rank == 0 else None,
                                            {"conv": "fp32", "stem": "dense", "pool": "separate"}, B, world, sync_all, dev,
                                            "the headline batch with options conv=fp32, stem=dense: every multiply on the fp32 MFMA pipe "
                                            "(v_mfma_f32_32x32x2_f32 / 16x16x4_f32: an exact fp32 FMA chain)"),
        }
    if not args.no_secondary and B >= 128:
        import copy
        from tools import bench_configs
        sa = copy.copy(args)
        sa.steps, sa.warmup, sa.batch = 10, 2, None
        secondary = {}
        for name, fn in (("dual_bf16", bench_configs.run_dual_bf16), ("train", bench_configs.run_train)):
            try:
                full = fn(sa, rank, world, dist, dev, sync_all, obs=obs, emit=False)
            except Exception as e:                      # the headline line must survive a failing side measurement
                full = {"error": f"{type(e).__name__}: {e}"[:300]}
            if rank == 0:
                secondary[name] = slim_secondary(name, full)
        torch.cuda.empty_cache()

    latency = None
    if not args.no_secondary and rank == 0:
        # the reference's own call shape (rl/ppo/ppo_trainer.py:836-841: ONE pair per environment step): model-only latency of a
        # batch-1 forward, observation tensors resident, with the persistent small-batch kernel (csrc/smallnet.hip) and with the
        # per-layer launches it replaces; outputs of the two compared
        try:
            with torch.no_grad():
                o1 = {k: v[:1].contiguous() for k, v in obs.items()}
                fam = model.layer_kernel("visual_encoder.backbone.layer1.0.convs.0", 1)[0]
                y_small = model(o1).clone()
                t_small = time_steps_simple(lambda: model(o1), 200, lambda: torch.cuda.synchronize(dev))
                model.set_option("small_net", "off")
                y_layers = model(o1).clone()
                t_layers = time_steps_simple(lambda: model(o1), 200, lambda: torch.cuda.synchronize(dev))
                model.set_option("small_net", "on")
            latency = {"workload": "one frame pair per call (the navigation loop's call shape), observation tensors resident",
                       "ms_per_call": 1e3 * t_small, "kernel_family": fam, "ms_per_call_per_layer_launches": 1e3 * t_layers,
                       "calls": 200, "max_abs_diff_between_the_two": float((y_small - y_layers).abs().max())}
        except Exception as e:
            latency = {"error": f"{type(e).__name__}: {e}"[:300]}

    navloop = None
    if not args.no_secondary and rank == 0 and B >= 128:
        # BASELINE configs[4] cannot run here (no habitat-sim / Gibson scenes): its GPU side — policy step + batched VO through the
        # boundary (host numpy frames in, PCIe included) + goal update per simulator step, 8 environments of one process
        try:
            from tools import bench_navloop
            nl = bench_navloop.run([8], 20, 150.0, dev)
            navloop = {"workload": "GPU side of BASELINE configs[4]'s loop per simulator step (policy.act + VO boundary call + goal update; "
                                   "the simulator itself is not emulated), 8 environments, host numpy observations",
                       **nl["results"][0], "unit": nl["unit"], "note": nl["note"]}
        except Exception as e:
            navloop = {"error": f"{type(e).__name__}: {e}"[:300]}

    with emit_lock:                            # from here on the deadline no longer fires (bail() returns when it finds the flag)
        emitted[0] = True
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if secondary is not None:
            res["secondary"] = secondary
        if raw_rec is not None:
            res.setdefault("secondary", {})["fwd_fp32_from_sensor_frames"] = raw_rec
        if strict is not None:
            res.setdefault("secondary", {}).update(strict)
        if navloop is not None:
            res.setdefault("secondary", {})["navloop_gpu_side"] = navloop
        if latency is not None:
            res.setdefault("secondary", {})["batch1_latency"] = latency
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, model.cfg.ngroups)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(args, rank, world, dist):
    """`--backend gloo --dry-run`: the N>1 control flow of this file (rendezvous, barrier brackets, max over ranks,
    rank-0 JSON) with a stub in place of the GPU step, so the multi-process path runs where there is no GPU."""
    cpu = torch.device("cpu")

    def sync_all():
        if dist is not None:
            dist.barrier()

    B = args.batch or 256
    # the pre-heat loop with a stub whose step times settle at a rank-dependent moment: all ranks must leave it together
    calls = [0]

    def stub():
        calls[0] += 1
        time.sleep(0.004 if calls[0] <= 2 + 2 * rank else 0.001)

    pre = preheat(stub, cpu, cap_s=1.0, tol=0.5)
    for _ in range(args.warmup):
        time.sleep(0.001)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))               # rank-dependent: the max over ranks must win
    sync_all()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, cpu)
    lo, hi = parallel.shard_bounds(world * B, rank, world)
    mine = torch.full((hi - lo, 1), float(rank))
    allr = parallel.gather_results(mine, world * B)          # the result gather of the sharded inference path
    counts = [int((allr == r).sum()) for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": world * B * args.steps / dt, "unit": "frame-pairs/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "none", "preheat_steps": pre[1], "rccl_ranks": process_group_record(dist, world),
                          "config": {"workload": "dry run of the N-rank control flow", "backend": args.backend,
                                     "shard_counts": counts}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
