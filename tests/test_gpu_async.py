"""GPU (-m gpu): pnvo_forward is asynchronous with the input-contract check ON ("no hidden host syncs inside
forward").  The reference's call shape enqueues one model per action back to back (base_trainer_with_vo.py:279-292); the decision to
redo a stem on float32 operands is taken on the device (stem_lds_kernel<.., PAIRED> predicated on the stem's flag), so no forward
waits for its stem — whether its inputs keep the contract or break it."""
import time

import numpy as np
import pytest
import torch

import bench
from oracle import oracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _backlog(ms_target=150.0):
    """GPU work of ~ms_target on the current stream, enqueued without waiting; returns an event behind it."""
    a = torch.randn(8192, 8192, device=DEV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (a @ a).sum().item()
    one = (time.perf_counter() - t0) * 1e3
    for _ in range(max(2, int(ms_target / max(one, 0.5)))):
        a = (a @ a) * 1e-4
    ev = torch.cuda.Event()
    ev.record()
    return ev


@pytest.mark.parametrize("B", [1, 16, 64])
def test_three_action_models_enqueue_behind_a_backlog_without_a_host_wait(B):
    models = [bench.build_model(DEV, seed=s)[0] for s in range(3)]          # forward / left / right
    obs = bench.make_inputs(B, DEV, 0)
    with torch.no_grad():
        want = [m(obs).clone() for m in models]                             # warm: workspaces, lazily built operands
        torch.cuda.synchronize()
        for m in models:
            assert m.get_option("input_fallback") == "on" and m.get_option("stem") == "auto"
        behind = _backlog()
        t0 = time.perf_counter()
        outs = [m(obs) for m in models]
        host_ms = (time.perf_counter() - t0) * 1e3
        still_busy = not behind.query()                # the backlog AHEAD of the three forwards has not even finished ...
        torch.cuda.synchronize()
    assert still_busy, "the three forwards returned only after the GPU had drained the work queued ahead of them"
    assert host_ms < 40.0, host_ms                     # ... and the three calls took launch time only (a wait would cost >= the backlog)
    for o, w in zip(outs, want):
        assert torch.equal(o, w)


def test_a_contract_breaking_forward_repairs_itself_on_the_device_without_a_host_wait():
    model, sd = bench.build_model(DEV)
    B = 16
    obs = bench.make_inputs(B, DEV, 0)
    bad = dict(obs)
    bad["rgb"] = obs["rgb"].clone()
    bad["rgb"][5, 100, 200, 2] = 17.3
    with torch.no_grad():
        clean = model(obs).clone()
        torch.cuda.synchronize()
        behind = _backlog()
        o_bad = model(bad)                             # breaks the contract: repaired by the predicated float32 stem behind its stem
        o_clean = model(obs)                           # enqueued before the host can know: its repair launches run too (flag still up)
        still_busy = not behind.query()
        torch.cuda.synchronize()
        assert still_busy
        assert model.get_option("stem") == "dense (fallback)"
        o_after = model(bad).clone()                   # the stand-in launched directly
    ref = oracle.forward(sd, {k: v[4:7].cpu().numpy() for k, v in bad.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    got = o_bad[4:7].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 1e-4, err
    assert torch.equal(o_bad, o_after)                 # same kernel, predicated or not
    assert float((o_clean - clean).abs().max() / clean.abs().max()) < 5e-6 and not torch.equal(o_bad[5], clean[5])


def test_the_repair_is_part_of_a_captured_graph():
    """A forward captured into a hipGraph carries its repair launches: replaying it on contract-breaking values is correct."""
    model, sd = bench.build_model(DEV)
    obs = {k: v.clone() for k, v in bench.make_inputs(8, DEV, 0).items()}
    with torch.no_grad():
        model(obs)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(obs)
        obs["rgb"][3, 50, 60, 1] = 200.5               # the captured forward now reads a fractional rgb value
        g.replay()
        torch.cuda.synchronize()
    ref = oracle.forward(sd, {k: v[3:4].cpu().numpy() for k, v in obs.items()}, ngroups=model.cfg.ngroups, dtype=np.float64)
    got = out[3:4].double().cpu().numpy()
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-2)
    assert err.max() < 1e-4, err
