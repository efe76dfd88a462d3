"""GPU (-m gpu): the HIP navigation policy (pnvo_policy_* through the nn.Module mirror) against the golden vectors
captured from the imported reference PointNavResNetPolicy and against the pinned oracle.  fp32 tolerance: 2e-4 of the
tensor's scale for features / hidden state / logits / value; the deterministic action must be identical."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import policy_oracle
from pointnav_vo_amd import synth
from pointnav_vo_amd.policy import PointNavResNetPolicy, policy_state_dict_spec
from pointnav_vo_amd.registry import baseline_registry

pytestmark = pytest.mark.gpu


class Box:
    def __init__(self, shape):
        self.shape = shape


class Space:
    def __init__(self, d):
        self.spaces = d


class Act:
    n = 4


def build(H, W, seed):
    cls = baseline_registry.get_policy("resnet_rnn_policy")
    assert cls is PointNavResNetPolicy
    space = Space({"depth": Box((H, W, 1)), "rgb": Box((H, W, 3)), "pointgoal_with_gps_compass": Box((2,))})
    pol = cls(observation_space=space, action_space=Act(), hidden_size=512, rnn_type="LSTM", num_recurrent_layers=2,
              backbone="resnet18", goal_sensor_uuid="pointgoal_with_gps_compass", normalize_visual_inputs=False,
              obs_transform=None, vis_types=["depth"])
    sd = synth.make_state_dict(policy_state_dict_spec(width=W, height=H), seed=seed)
    assert list(pol.state_dict().keys()) == list(sd.keys())
    pol.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return pol.to("cuda:0").eval(), sd


def close(got, want, tol=2e-4):
    scale = np.abs(want).max() + 1e-6
    return np.abs(got - want).max() / scale < tol


@pytest.mark.parametrize("fname", ["policy_128x96_b2.npz", "policy_341x192_b3.npz"])
def test_policy_steps_match_reference(fname):
    rec = load_golden(fname)
    H, W, B, steps = (int(rec[k]) for k in ("H", "W", "B", "steps"))
    pol, sd = build(H, W, int(rec["weight_seed"]))
    dev = torch.device("cuda", 0)
    hidden = torch.zeros(pol.num_recurrent_layers, B, 512, device=dev)
    hid_o = np.zeros((4, B, 512))
    for t, (depth, goal, prev, mask) in enumerate(synth.make_policy_inputs(H, W, B, steps, int(rec["input_seed"]))):
        obs = {"depth": torch.from_numpy(depth).to(dev), "pointgoal_with_gps_compass": torch.from_numpy(goal).to(dev)}
        pa, mk = torch.from_numpy(prev).view(B, 1).to(dev), torch.from_numpy(mask).view(B, 1).to(dev)
        feats, hnew, logits, value = pol.features_and_logits(obs, hidden, pa, mk)
        v2, action, logp, h2 = pol.act(obs, hidden, pa, mk, deterministic=True)
        torch.cuda.synchronize()
        assert torch.equal(h2, hnew) and torch.equal(v2, value)          # act() is deterministic given the inputs
        assert close(feats.cpu().numpy(), rec[f"features64/{t}"]), t
        assert close(hnew.cpu().numpy(), rec[f"hidden64/{t}"]), t
        assert close(logits.cpu().numpy(), rec[f"logits_raw64/{t}"]), t
        assert close(value.cpu().numpy(), rec[f"value64/{t}"]), t
        np.testing.assert_array_equal(action.cpu().numpy(), rec[f"action64/{t}"])
        np.testing.assert_allclose(logp.cpu().numpy(), rec[f"logp64/{t}"], rtol=0, atol=2e-4)
        assert tuple(value.shape) == (B, 1) and tuple(action.shape) == (B, 1) and action.dtype == torch.int64
        # and against the oracle stepped alongside
        o = policy_oracle.policy_step(sd, depth, goal, prev, mask, hid_o)
        assert close(hnew.cpu().numpy(), o["hidden"]), t
        hid_o = o["hidden"]
        hidden = hnew


def test_policy_sampling_and_value():
    pol, _ = build(96, 128, 3)
    dev = torch.device("cuda", 0)
    B = 4
    depth, goal, prev, mask = synth.make_policy_inputs(96, 128, B, 1, 9)[0]
    obs = {"depth": torch.from_numpy(depth).to(dev), "pointgoal_with_gps_compass": torch.from_numpy(goal).to(dev)}
    hidden = torch.zeros(pol.num_recurrent_layers, B, 512, device=dev)
    pa, mk = torch.from_numpy(prev).view(B, 1).to(dev), torch.from_numpy(mask).view(B, 1).to(dev)
    value, action, logp, hnew = pol.act(obs, hidden, pa, mk, deterministic=False)
    assert action.min() >= 0 and action.max() < 4 and torch.isfinite(logp).all() and (logp <= 0).all()
    v = pol.get_value(obs, hidden, pa, mk)
    torch.testing.assert_close(v, value)
    with pytest.raises(NotImplementedError):
        pol.evaluate_actions(obs, hidden, pa, mk, action)


@pytest.mark.parametrize("B", [5, 8, 9, 16])
def test_policy_batches_of_a_nav_loop_match_the_oracle(B):
    """Environment counts of one nav-loop process (5-16).  Up to 8 quarter-size frames the policy's encoder runs behind its stem as ONE
    persistent launch (smallnet.hip: the same amount of work as 2 VO pairs), above that as per-layer launches; the LSTM layers and the
    two heads are one launch each.  Both sides of the threshold against the pinned oracle, two steps (the second on a carried state)."""
    H, W = 192, 341
    pol, sd = build(H, W, 5)
    dev = torch.device("cuda", 0)
    hidden = torch.zeros(pol.num_recurrent_layers, B, 512, device=dev)
    hid_o = np.zeros((4, B, 512))
    for t, (depth, goal, prev, mask) in enumerate(synth.make_policy_inputs(H, W, B, 2, 21)):
        obs = {"depth": torch.from_numpy(depth).to(dev), "pointgoal_with_gps_compass": torch.from_numpy(goal).to(dev)}
        pa, mk = torch.from_numpy(prev).view(B, 1).to(dev), torch.from_numpy(mask).view(B, 1).to(dev)
        feats, hnew, logits, value = pol.features_and_logits(obs, hidden, pa, mk)
        o = policy_oracle.policy_step(sd, depth, goal, prev, mask, hid_o)
        assert torch.isfinite(hnew).all() and close(hnew.cpu().numpy(), o["hidden"]), (B, t)
        for k, got in (("logits", logits), ("value", value)):
            if k in o:
                assert close(got.cpu().numpy().reshape(np.asarray(o[k]).shape), np.asarray(o[k])), (B, t, k)
        hid_o, hidden = o["hidden"], hnew
