"""Data-parallel plumbing for the VO path: one process per GPU, pairs sharded contiguously, no data-path collective.

Frame pairs are independent at inference (GroupNorm is per sample, RunningMeanAndVar is frozen in eval — SURVEY.md
§8(e)); the reference already runs one process per GPU with its own env shard (/root/reference/launch.py:11-12,
pointnav_vo/rl/ddppo/algo/ddppo_trainer.py:199-216).  The only communication is gathering the 12 B/pair results and,
for timing, a MAX over ranks — both through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous split: rank g owns pairs [lo, hi); sizes differ by at most 1."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(local_out: torch.Tensor, total: int) -> torch.Tensor:
    """All ranks contribute their [n_local, D] results; every rank receives the [total, D] tensor in pair order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_out
    world = dist.get_world_size()
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax, local_out.shape[1]), dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.shape[0]] = local_out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def max_over_ranks(seconds: float, device=None) -> float:
    """Elapsed time of the slowest rank (bench.py contract: barrier + sync on both sides, MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean of a flat buffer over all ranks — the ONE data-path collective of the training step (the
    3 962 305-float gradient buffer of an action model: 15.85 MB; ring all-reduce over xGMI)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat)
        flat /= dist.get_world_size()
    return flat
