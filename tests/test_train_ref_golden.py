"""CPU: the training-step checker (oracle/torch_train_ref.py, functional torch restatement) is pinned against golden
vectors captured from the imported reference model in train mode + the reference's own loss code + torch.optim.Adam."""
import numpy as np
import pytest
import torch

from conftest import golden_case, load_golden
from oracle import torch_train_ref as ref


def run_steps(rec, dtype, nsteps):
    cfg, sd, obs, _ = golden_case(rec)
    cur = {k: np.array(v) for k, v in sd.items()}
    state, res = None, []
    for step in range(1, nsteps + 1):
        r = ref.train_step(cur, obs, rec["target"], ngroups=cfg.ngroups, lr=float(rec["lr"]), eps=float(rec["eps"]),
                           dtype=dtype, state=state, step=step)
        res.append(r)
        state = r["state"]
        cur = {**{k: v.numpy() for k, v in r["params"].items()}, **{k: v.numpy() for k, v in r["buffers"].items()}}
    return res


@pytest.mark.parametrize("fname,dtype,tol", [("train_default_45x37_b4_f64.npz", torch.float64, 1e-9),
                                             ("train_default_96x64_b3_f32.npz", torch.float32, 2e-3)])
def test_train_step_matches_reference(fname, dtype, tol):
    rec = load_golden(fname)
    steps = run_steps(rec, dtype, 2 if dtype == torch.float64 else 1)
    for s, r in enumerate(steps, start=1):
        assert abs(float(r["loss"]) - float(rec[f"loss{s}"])) <= tol * max(1.0, abs(float(rec[f"loss{s}"])))
        np.testing.assert_allclose(r["out"].double().numpy(), rec[f"out{s}"], rtol=tol, atol=tol)
        for k, g in r["grads"].items():
            gf = g.reshape(-1).double().numpy()
            nrm = float(rec[f"g{s}norm/{k}"])
            assert abs(np.linalg.norm(gf) - nrm) <= tol * max(nrm, 1e-6) * 5, k
            np.testing.assert_allclose(gf[rec[f"gidx/{k}"]], rec[f"g{s}val/{k}"], rtol=20 * tol, atol=20 * tol * max(nrm, 1e-3), err_msg=k)
            pf = r["params"][k].reshape(-1).double().numpy()
            if dtype == torch.float64:   # Adam's first step is +-lr whatever the gradient: only meaningful in fp64
                np.testing.assert_allclose(pf[rec[f"gidx/{k}"]], rec[f"p{s}val/{k}"], rtol=1e-9, atol=1e-12, err_msg=k)
        for k, b in r["buffers"].items():
            np.testing.assert_allclose(b.double().numpy().reshape(-1), rec[f"buf{s}/{k}"].reshape(-1), rtol=max(tol, 1e-6), atol=1e-7, err_msg=k)


def test_geo_inverse_loss_matches_reference():
    rec = load_golden("train_geo_loss.npz")
    d = torch.from_numpy(rec["deltas"]).requires_grad_(True)
    loss = ref.geo_inverse_loss(d, torch.from_numpy(rec["actions"]))
    loss.backward()
    assert abs(loss.item() - float(rec["loss"])) < 1e-12
    np.testing.assert_allclose(d.grad.numpy(), rec["grad"], rtol=1e-10, atol=1e-14)
