"""CPU: the policy oracle (oracle/policy_oracle.py + the C conv oracle) against the golden vectors captured from the
imported reference PointNavResNetPolicy (tests/golden/gen_golden_policy.py)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import policy_oracle
from pointnav_vo_amd import synth
from pointnav_vo_amd.policy import policy_state_dict_spec


@pytest.mark.parametrize("fname", ["policy_128x96_b2.npz", "policy_341x192_b3.npz"])
def test_policy_oracle_matches_reference(fname):
    rec = load_golden(fname)
    H, W, B, steps = (int(rec[k]) for k in ("H", "W", "B", "steps"))
    sd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=int(rec["weight_seed"]))
    hidden = np.zeros((4, B, 512))
    for t, (depth, goal, prev, mask) in enumerate(synth.make_policy_inputs(H, W, B, steps, int(rec["input_seed"]))):
        out = policy_oracle.policy_step(sd, depth, goal, prev, mask, hidden, dtype=np.float64)
        for key, ref in (("features", "features64"), ("hidden", "hidden64"), ("logits", "logits_raw64"),
                         ("value", "value64")):
            want = rec[f"{ref}/{t}"]
            np.testing.assert_allclose(out[key], want, rtol=1e-9, atol=1e-11, err_msg=f"{key} step {t}")
        assert np.array_equal(out["logits"].argmax(-1)[:, None], rec[f"action64/{t}"])
        hidden = out["hidden"]


def test_mask_resets_only_the_masked_environment():
    rec = load_golden("policy_128x96_b2.npz")
    H, W, B = int(rec["H"]), int(rec["W"]), int(rec["B"])
    sd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=int(rec["weight_seed"]))
    depth, goal, prev, _ = synth.make_policy_inputs(H, W, B, 1, 3)[0]
    hid = synth.uniform(1, "hid", (4, B, 512), -1.0, 1.0)
    a = policy_oracle.policy_step(sd, depth, goal, prev, np.array([1.0, 0.0], np.float32), hid)
    b = policy_oracle.policy_step(sd, depth, goal, prev, np.array([1.0, 0.0], np.float32), hid * np.array([1, 0])[None, :, None])
    c = policy_oracle.policy_step(sd, depth, goal, prev, np.array([1.0, 1.0], np.float32), hid)
    np.testing.assert_allclose(a["hidden"], b["hidden"], rtol=0, atol=0)             # env 1's state was ignored
    np.testing.assert_allclose(a["hidden"][:, 0], c["hidden"][:, 0], rtol=0, atol=0)  # env 0 unaffected by env 1's mask
    assert np.abs(a["hidden"][:, 1] - c["hidden"][:, 1]).max() > 1e-3
