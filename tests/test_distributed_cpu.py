"""CPU, world_size 2, gloo: the N>1 path of the inference sharding (contiguous pair shards, shard-consistent synthetic
inputs, ordered gather, MAX-over-ranks timing).  The per-rank compute here is the oracle on a tiny configuration (the HIP
forward needs a GPU); the sharding/gather logic under test is the code bench.py and a multi-GPU caller use."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointnav_vo_amd import model_spec as ms
from pointnav_vo_amd import parallel, synth

SPACE = ["rgb", "depth"]
W, H, TOTAL = 40, 36, 5


def _forward(sd, cfg, obs):
    from oracle import oracle
    oracle.set_threads(1)
    return oracle.forward(sd, obs, ngroups=cfg.ngroups, dtype=np.float64)


def _cfg():
    cfg = ms.config_from_kwargs(observation_space=SPACE, observation_size=(W, H), normalize_visual_inputs=True, output_dim=3)
    return cfg, synth.make_state_dict(ms.state_dict_spec(cfg), seed=11)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sd = _cfg()
    lo, hi = parallel.shard_bounds(TOTAL, rank, world)
    obs = synth.make_obs_pairs(hi - lo, H, W, observation_space=SPACE, seed=5, start=lo)   # this rank's pairs only
    out = torch.from_numpy(_forward(sd, cfg, obs))
    full = parallel.gather_results(out, TOTAL)
    tmax = parallel.max_over_ranks(1.0 + rank)
    g = parallel.allreduce_mean_(torch.arange(6, dtype=torch.float32) * (rank + 1))   # gradient averaging
    if rank == 0:
        q.put((full.numpy(), tmax, g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything_once():
    for total in (1, 5, 256, 1023):
        for world in (1, 2, 3, 8):
            b = [parallel.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_two_rank_sharded_inference_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, tmax, g = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg, sd = _cfg()
    ref = _forward(sd, cfg, synth.make_obs_pairs(TOTAL, H, W, observation_space=SPACE, seed=5))
    np.testing.assert_array_equal(full, ref)        # N-rank result == 1-process result, in pair order
    assert tmax == 2.0                               # MAX over ranks
    np.testing.assert_array_equal(g, np.arange(6, dtype=np.float32) * 1.5)   # mean of rank 0 (x1) and rank 1 (x2)


# ---------------------------------------------------------------------------------------------------------------------
# The PRODUCT's host logic of the distributed training step (train.py, not the oracle): RunningMeanAndVar's train-mode
# update with its three all-reduces (running_mean_and_var.py:27-38) and the flat-gradient mean.  Only the device
# statistics kernel (pnvo_input_moments) is substituted by numpy moments of the same tensors.
class _HostStatsStep:
    """VOTrainStep with the HIP moments kernel replaced: everything else is train.py's own code."""

    def __new__(cls, obs, C):
        from pointnav_vo_amd.train import VOTrainStep

        class _RMV:
            pass

        self = object.__new__(type("HostStatsStep", (VOTrainStep,), {"_input_moments": cls._moments}))
        self.rmv = _RMV()
        self.rmv._mean = torch.zeros(1, C, 1, 1)
        self.rmv._var = torch.zeros(1, C, 1, 1)
        self.rmv._count = torch.zeros(())
        self._m12 = torch.empty(2 * C)
        self._x = _assembled(obs)
        return self

    @staticmethod
    def _moments(self, ptrs, B, center, power, out, stream):
        x = self._x
        c = torch.zeros(x.shape[1]) if center is None else center
        d = x - c.view(1, -1, 1, 1)
        assert power == 3                          # one pass: first and second moment about `center`
        out.copy_(torch.cat([d.mean(dim=(0, 2, 3)), (d * d).mean(dim=(0, 2, 3))]))


def _assembled(obs):
    """[B,C,H,W] in the reference channel order (vo_cnn.py:114-174), rgb / 255."""
    halves = [[], []]
    for key in ("rgb", "depth", "discretized_depth", "top_down_view"):
        if key in obs:
            t = torch.as_tensor(obs[key]).float().permute(0, 3, 1, 2)
            t = t / 255.0 if key == "rgb" else t
            n = t.shape[1] // 2
            halves[0].append(t[:, :n])
            halves[1].append(t[:, n:])
    return torch.cat(halves[0] + halves[1], dim=1)


def _stats_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_bounds(6, rank, world)
    obs = synth.make_obs_pairs(hi - lo, H, W, observation_space=SPACE, seed=9, start=lo)
    st = _HostStatsStep(obs, 8)
    for _ in range(2):                                   # two consecutive training forwards (the merge uses the old state)
        st._update_running_stats(None, hi - lo, None)
    flat = torch.full((3962305,), float(rank + 1))       # the default model's flat gradient buffer (15.85 MB)
    parallel.allreduce_mean_(flat)
    if rank == 0:
        q.put((st.rmv._mean.numpy(), st.rmv._var.numpy(), float(st.rmv._count), float(flat[0]), float(flat[-1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_running_stats_and_gradient_mean_equal_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_stats_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mean, var, count, g0, g1 = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process on the concatenated batch, and the reference's own formula restated by the checker
    from oracle import torch_train_ref as ref
    obs = synth.make_obs_pairs(6, H, W, observation_space=SPACE, seed=9)
    st = _HostStatsStep(obs, 8)
    x = _assembled(obs).double()
    m, v, c = torch.zeros(1, 8, 1, 1, dtype=torch.float64), torch.zeros(1, 8, 1, 1, dtype=torch.float64), torch.zeros((), dtype=torch.float64)
    for _ in range(2):
        st._update_running_stats(None, 6, None)
        m, v, c = ref.running_stats_update(x, m, v, c)
    np.testing.assert_allclose(mean, st.rmv._mean.numpy(), rtol=1e-5, atol=1e-6)     # 2 ranks == 1 process
    np.testing.assert_allclose(var, st.rmv._var.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mean, m.numpy(), rtol=1e-5, atol=1e-6)                 # == the reference formula (fp64)
    np.testing.assert_allclose(var, v.numpy(), rtol=1e-4, atol=1e-6)
    assert count == float(c) == 12.0
    assert g0 == g1 == 1.5


def _absent_worker(rank, world, port, q):
    """Rank 0 holds all 4 entries of an action model, rank 1 none: rank 1 walks participate_absent()'s collective sequence
    (zero contributions) while rank 0 runs its ordinary update; both must finish (no hang) with the SAME statistics."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    obs = synth.make_obs_pairs(4, H, W, observation_space=SPACE, seed=9)
    st = _HostStatsStep(obs, 8)
    for _ in range(2):
        st._update_running_stats(None, 4 if rank == 0 else 0, None)
    q.put((rank, st.rmv._mean.numpy(), st.rmv._var.numpy(), float(st.rmv._count)))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_without_entries_joins_the_running_stats_collectives():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_absent_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (m, v, c)) for r, m, v, c in (q.get(), q.get()))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    obs = synth.make_obs_pairs(4, H, W, observation_space=SPACE, seed=9)
    st = _HostStatsStep(obs, 8)
    for _ in range(2):
        st._update_running_stats(None, 4, None)
    for r in (0, 1):                                     # both ranks == the single process that saw the 4 entries
        np.testing.assert_allclose(got[r][0], st.rmv._mean.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[r][1], st.rmv._var.numpy(), rtol=1e-5, atol=1e-7)
        assert got[r][2] == 8.0


def test_bench_multi_rank_control_flow_runs_under_gloo():
    """`bench.py --backend gloo --dry-run`: rendezvous, barrier brackets, MAX over ranks, result gather and the rank-0
    JSON line of the N > 1 launch, executed as the driver launches it (torch.distributed.run, 2 ranks)."""
    import json
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4",
                        "--warmup", "1", "--backend", "gloo", "--dry-run"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["config"]["shard_counts"] == [256, 256]
    assert rec["ms_per_step"] >= 2.0                                  # the slower rank (2 ms per stub step) sets the time
    # the pre-heat loop is left by all ranks together (its steps may contain collectives): rank 1's stub settles two steps
    # later than rank 0's, and the run did not hang
    assert rec["preheat_steps"] >= 6


def test_bench_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (how the round-4 driver called it; the reference's launch.py:9-32 starts its
    own ranks the same way): bench.py re-enters itself under torch.distributed.run; still ONE JSON line, and the record
    shows the process group's own world size."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--dry-run"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == {"world_size": 2, "backend": "gloo", "is_rccl": False,
                                                        "launched_by": "torch.distributed.run"}
    # a failing rank is not swallowed: the launcher's exit code comes back
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--backend", "gloo",
                        "--dry-run", "--config", "nonsense"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0
