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
