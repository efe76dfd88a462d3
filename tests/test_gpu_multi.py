"""GPU (-m gpu), needs >= 2 MI355X (skipped on a 1-GPU box): the data-parallel paths through libpnvo.so under
torch.distributed with backend nccl (RCCL over xGMI) — SURVEY.md section 4(iii) / section 8(e).
  * sharded inference: N-rank gathered result == the same shards on one GPU, bit for bit (and the whole batch to 1e-5)
  * VOTrainStep: 2-rank step (RunningMeanAndVar's three all-reduces, running_mean_and_var.py:27-38, and the ONE flat
    gradient all-reduce) == the single-GPU step on the concatenated batch
The same host logic runs under gloo on CPU in tests/test_distributed_cpu.py, and — on any GPU box — as two ranks that share
cuda:0 with gloo collectives (the `shared_gpu` tests below: everything but RCCL itself)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
need2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least 2 GPUs")


def _run(mode, *extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "mgpu_worker.py"), "--mode", mode, *extra],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]


def test_two_ranks_sharing_one_gpu_sharded_inference():
    """The 2-rank sharding / gather logic on the real kernels of a 1-GPU box (collectives over gloo)."""
    _run("infer", "--shared-gpu")


def test_two_ranks_sharing_one_gpu_train_step_equals_the_step_on_the_concatenated_batch():
    """Data-parallel VOTrainStep (statistics + gradient all-reduces over gloo, both ranks on cuda:0) against the single
    process step on the concatenated batch."""
    _run("train", "--shared-gpu")


def test_two_ranks_sharing_one_gpu_action_models_absent_on_one_rank():
    """GeoInvarianceTrainStep when an action model has entries on one rank only: the set of models that run their
    collectives is decided globally (all-reduced presence mask), the absent rank contributes zeros."""
    _run("geo", "--shared-gpu")


@need2
def test_two_gpu_action_models_absent_on_one_rank():
    _run("geo")


@need2
def test_two_gpu_sharded_inference_equals_one_gpu():
    _run("infer")


@need2
def test_two_gpu_train_step_equals_one_gpu_step_on_the_concatenated_batch():
    _run("train")


def test_bench_headline_with_its_secondary_runs_as_two_ranks_sharing_one_gpu():
    """The driver's default command at N = 2: the headline plus its `secondary` measurements (dual bf16, training step with the
    bucketed gradient all-reduce and the statistics collectives, sensor-frame variants) — every rank walks the same collectives."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--backend", "gloo", "--shared-gpu", "--no-cpu-baseline", "--batch", "128"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    sec = rec["secondary"]
    assert rec["n_gpus"] == 2 and set(sec) >= {"dual_bf16", "train", "fwd_fp32_from_sensor_frames"}
    for k in ("dual_bf16", "train"):
        assert "error" not in sec[k], sec[k]
        assert sec[k]["value"] > 0
    assert sec["fwd_fp32_from_sensor_frames"]["bit_identical_to_headline_outputs"] is True
    assert sec["train"]["gradient_allreduce"].startswith("bucketed")


def test_bench_plain_invocation_self_launches_two_ranks_on_the_real_kernels():
    """`python bench.py --gpus 2 ...` with no launcher in front (the round-4 driver's command shape): bench.py starts its own
    two ranks; here both share cuda:0 over gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--shared-gpu", "--no-cpu-baseline", "--no-secondary", "--batch", "16"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"]["world_size"] == 2 and rec["config"]["global_batch"] == 32
    assert rec["multi_gpu"]["shards_equal_single_gpu"] is True
    rf = rec["roofline"]                                   # the fraction is the ALGORITHMIC one; executed work beside it
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["executed_frac"] >= rf["frac"]


def test_bench_headline_survives_side_measurements_that_do_not_return():
    """N > 1: the side measurements walk collectives (the training step's all-reduce ...).  If one of them never returns, every rank
    leaves after --secondary-deadline and rank 0 still prints the ONE JSON line — the headline, with the side measurements marked as
    not finished.  Forced here with a deadline no side measurement can meet."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--backend", "gloo", "--shared-gpu", "--no-cpu-baseline", "--batch", "128",
                        "--secondary-deadline", "0.05"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["multi_gpu"]["shards_equal_single_gpu"] is True
    assert "did not finish" in rec["secondary"]["error"]


@pytest.mark.parametrize("config", ["fwd_fp32", "train"])
def test_bench_two_ranks_sharing_one_gpu(config):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run), both ranks on cuda:0 over gloo: the real timed
    path of the N > 1 launch — collective pre-heat exit, barrier brackets, max over ranks, ONE JSON line from rank 0."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--backend", "gloo", "--shared-gpu", "--no-cpu-baseline", "--config", config,
                        "--batch", "16"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["config"]["global_batch"] == 32 and rec["value"] > 0
    mg = rec["multi_gpu"]                                             # what makes an N > 1 line self-explaining
    assert len(mg["per_rank_ms_per_step"]) == 2 and min(mg["per_rank_ms_per_step"]) > 0
    assert abs(max(mg["per_rank_ms_per_step"]) - rec["ms_per_step"]) < 0.25 * rec["ms_per_step"] + 0.2   # the value is the slowest rank's
    if config == "fwd_fp32":
        assert mg["shards_equal_single_gpu"] is True and mg["shards_checked"] == 2 and mg["max_abs_diff_vs_single_gpu"] == 0.0
    else:
        assert mg["ms_per_step_bucketed"] > 0 and mg["ms_per_step_flat"] > 0 and mg["allreduce_alone_ms"] > 0
        assert mg["allreduce_bytes"] == 4 * 3962305



def test_more_ranks_than_gpus_is_a_one_line_diagnostic():
    """`bench.py --gpus N` on a node with fewer than N GPUs (the first real multi-GPU run may land on one): exit code 2 and ONE line
    on stderr — no launcher, no rank tracebacks, no JSON line that could be mistaken for a measurement."""
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    lines = [ln for ln in r.stderr.splitlines() if ln.strip() and "amdgpu.ids" not in ln]
    assert len(lines) == 1 and f"--gpus {n}" in lines[0] and "GPU(s)" in lines[0], r.stderr[-800:]
    assert r.stdout.strip() == ""
