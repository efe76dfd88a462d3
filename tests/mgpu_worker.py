"""Worker of tests/test_gpu_multi.py: one process per GPU (torch.distributed.run), backend nccl (= RCCL).
    --mode infer : rank r runs its contiguous shard of the pairs through libpnvo.so; the gathered result must equal, BIT FOR
                   BIT, the same shards evaluated one after the other on one GPU, and the whole batch at once to 1e-5
                   (kernel / tile choices depend on the batch size of a launch)
    --mode train : data-parallel VOTrainStep (RunningMeanAndVar all-reduces + one flat-gradient all-reduce, Adam) on the
                   rank's half of the batch must equal the single-GPU step on the concatenated batch (rank 0 checks)
    --mode geo   : GeoInvarianceTrainStep over {left, right} action models when rank 0's batch holds only `left` entries and
                   rank 1's only `right` ones: both ranks must issue the same collectives (no hang), end with identical
                   parameters / running statistics for BOTH models, and the step counts must agree
    --shared-gpu : both ranks use cuda:0 and the collectives go through gloo — the 2-rank logic on the real kernels of a
                   1-GPU box (everything but RCCL itself)
Exit code 0 = all assertions held."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointnav_vo_amd import model_spec as ms, parallel, synth, vo_cnn  # noqa: E402,F401
from pointnav_vo_amd.registry import baseline_registry  # noqa: E402
from pointnav_vo_amd.train import VOTrainStep  # noqa: E402

SPACE = ["rgb", "depth", "discretized_depth", "top_down_view"]
W, H, BINS = 96, 64, 10


def build(dev, dropout_p=0.0):
    model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=SPACE, observation_size=(W, H), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=dropout_p, discretized_depth_channels=BINS)
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=3)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return model.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["infer", "train", "geo"])
    ap.add_argument("--shared-gpu", action="store_true")
    a = ap.parse_args()
    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    if a.shared_gpu:
        lr = 0
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    if a.shared_gpu:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    if a.mode == "geo":
        return geo(rank, world, dev)
    total = 7 if a.mode == "infer" else 6
    lo, hi = parallel.shard_bounds(total, rank, world)
    shard = {k: torch.from_numpy(v).to(dev) for k, v in
             synth.make_obs_pairs(hi - lo, H, W, observation_space=SPACE, dd_bins=BINS, seed=21, start=lo).items()}
    full = {k: torch.from_numpy(v).to(dev) for k, v in
            synth.make_obs_pairs(total, H, W, observation_space=SPACE, dd_bins=BINS, seed=21).items()}
    if a.mode == "infer":
        model = build(dev).eval()
        with torch.no_grad():
            mine = model(shard)
            gathered = parallel.gather_results(mine, total)
            # kernel and tile choices depend on the batch size of a launch (measured on one GPU: 7 pairs at once vs 4 + 3
            # differ by 9e-7), so the bit-exact reference evaluates the same shards one after the other on this GPU; the
            # whole batch at once must agree to 1e-5
            single = torch.cat([model({k: v[a:b] for k, v in full.items()})
                                for a, b in (parallel.shard_bounds(total, r, world) for r in range(world))])
            whole = model(full)
        assert torch.equal(gathered, single), (rank, (gathered - single).abs().max())
        assert (gathered - whole).abs().max() <= 1e-5 * max(1.0, float(whole.abs().max())), (rank, (gathered - whole).abs().max())
    else:
        tgt_full = torch.from_numpy(np.random.default_rng(4).normal(size=(total, 3)).astype(np.float32) * 0.2).to(dev)
        m_dp, m_one = build(dev), build(dev)
        ts_dp = VOTrainStep(m_dp)
        out_dp, loss_dp = ts_dp.step(shard, tgt_full[lo:hi])          # 3 all-reduces of the statistics + 1 of the gradients
        # single-GPU step on the concatenated batch, with the collectives switched off for this model
        saved = dist.is_initialized
        try:
            torch.distributed.is_initialized = lambda: False
            ts_one = VOTrainStep(m_one)
            out_one, loss_one = ts_one.step(full, tgt_full)
        finally:
            torch.distributed.is_initialized = saved
        rmv_dp, rmv_one = m_dp.visual_encoder.running_mean_and_var, m_one.visual_encoder.running_mean_and_var
        for k in ("_mean", "_var", "_count"):
            torch.testing.assert_close(getattr(rmv_dp, k), getattr(rmv_one, k), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(out_dp, out_one[lo:hi], rtol=2e-4, atol=2e-5)
        # mean over ranks of per-rank mean losses == the full-batch loss (equal shard sizes); gradients likewise
        g_dp, g_one = ts_dp.grad, ts_one.grad
        rel = (g_dp - g_one).norm() / g_one.norm()
        assert rel < 1e-4, rel
        # Adam's first step is lr * sign(g): compare where the sign is not within rounding noise of zero
        sel = g_one.abs() > 1e-6 * g_one.abs().max()
        assert (ts_dp.flat[sel] - ts_one.flat[sel]).abs().max() < 2e-6
    dist.barrier()
    dist.destroy_process_group()


def geo(rank, world, dev):
    from pointnav_vo_amd.train import GeoInvarianceTrainStep, TURN_LEFT, TURN_RIGHT, CUR_REL_TO_PREV
    steps = {TURN_LEFT: VOTrainStep(build(dev)), TURN_RIGHT: VOTrainStep(build(dev))}
    eng = GeoInvarianceTrainStep(steps, invariance_types=())
    n = 3
    batch = {k: torch.from_numpy(v).to(dev) for k, v in
             synth.make_obs_pairs(n, H, W, observation_space=SPACE, dd_bins=BINS, seed=30 + rank).items()}
    tgt = torch.from_numpy(np.random.default_rng(5 + rank).normal(size=(n, 3)).astype(np.float32) * 0.2)
    mine = TURN_LEFT if rank == 0 else TURN_RIGHT
    count0 = {act: float(st.model.visual_encoder.running_mean_and_var._count) for act, st in steps.items()}
    for it in range(2):                                   # second iteration: every model has stepped once on every rank
        total, preds = eng.step(batch, [mine] * n, [CUR_REL_TO_PREV] * n, tgt)
        assert torch.isfinite(total).all() and torch.isfinite(preds).all()
    for act, st in steps.items():
        assert st.step_count == 2, (rank, act, st.step_count)
        rmv = st.model.visual_encoder.running_mean_and_var
        for t in (st.flat, rmv._mean.reshape(-1), rmv._var.reshape(-1), rmv._count.reshape(-1)):
            parts = [torch.empty_like(t.cpu()) for _ in range(world)]
            dist.all_gather(parts, t.cpu().contiguous())
            assert torch.equal(parts[0], parts[1]), (rank, act)
        assert float(rmv._count) == count0[act] + 2 * n                 # the entries of ONE rank per iteration, seen by both
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
