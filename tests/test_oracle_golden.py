"""CPU: the oracle (oracle/*.c restatement) is pinned against golden vectors captured from the imported
reference (tests/golden/gen_golden.py).  No GPU, no /root/reference at run time."""
import numpy as np
import pytest

from conftest import MODEL_FIXTURES, golden_case, load_golden, pair_rel_err
from oracle import oracle


@pytest.mark.parametrize("fname", MODEL_FIXTURES)
def test_forward_fp64_matches_reference(fname):
    rec = load_golden(fname)
    cfg, sd, obs, actions = golden_case(rec)
    taps = {}
    out = oracle.forward(sd, obs, ngroups=cfg.ngroups, dtype=np.float64, actions=actions, taps=taps)
    assert out.shape == rec["out64"].shape
    assert pair_rel_err(out, rec["out64"]).max() < 1e-10
    # intermediate activations: sampled values + statistics of every tap the reference hooks exposed
    for k in [k[7:] for k in rec if k.startswith("tapidx/")]:
        flat = taps[k].reshape(-1)
        np.testing.assert_allclose(flat[rec[f"tapidx/{k}"]], rec[f"tapval/{k}"], rtol=1e-9, atol=1e-11, err_msg=k)
        st = np.array([flat.mean(), np.sqrt((flat ** 2).mean()), np.abs(flat).max()])
        np.testing.assert_allclose(st, rec[f"tapstat/{k}"], rtol=1e-9, atol=1e-12, err_msg=k)
        if f"tap/{k}" in rec:
            np.testing.assert_allclose(taps[k], rec[f"tap/{k}"].reshape(taps[k].shape), rtol=2e-6, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("fname", MODEL_FIXTURES)
def test_forward_fp32_matches_reference(fname):
    rec = load_golden(fname)
    cfg, sd, obs, actions = golden_case(rec)
    out = oracle.forward(sd, obs, ngroups=cfg.ngroups, dtype=np.float32, actions=actions)
    # both are fp32 evaluations of the same function in different summation orders; the reference itself moves
    # by 4.5e-5 elementwise between thread counts
    assert pair_rel_err(out, rec["out64"]).max() < 2e-5
    assert pair_rel_err(out, rec["out32"]).max() < 4e-5


def test_discretize_depth_matches_reference():
    rec = load_golden("preproc.npz")
    dd, fired = oracle.discretize_depth(rec["dd_depth"], int(rec["dd_bins"]))
    assert fired == rec["dd_depth"].size          # the reference's assert at base_trainer_with_vo.py:163
    np.testing.assert_array_equal(dd.argmax(-1).astype(np.uint8), rec["dd_index"])
    assert dd.sum() == rec["dd_depth"].size


TDV_CASES = ["full_uniform", "full_border", "full_near", "full_fp32", "full_zero", "full_one_pixel",
             "full_top_band", "small_odd", "small_border"]


@pytest.mark.parametrize("case", TDV_CASES)
def test_topdown_matches_reference_bit_exact(case):
    rec = load_golden("preproc.npz")
    d = rec[f"tdv_in/{case}"].astype(np.float32)
    H, W = d.shape[:2]
    ref_c = rec[f"tdv_consts/{H}x{W}"]
    c = oracle.topdown_consts(H, W, 70, 0.1, 10.0)
    # closed-form constants == what the reference's torch.inverse / _get_x_range produce
    np.testing.assert_array_equal(c[:7], ref_c[:7])
    assert ref_c[7] == 0.0
    out = oracle.topdown_view(d, c)
    want = np.zeros(H * W, np.float32)
    want[rec[f"tdv_nz/{case}"]] = rec[f"tdv_val/{case}"]
    np.testing.assert_array_equal(out.reshape(-1), want)
