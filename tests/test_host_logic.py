"""CPU: host-side mirrors of the reference interface (registry names, constructor contract, state_dict layout,
error behaviour) and the model/roofline bookkeeping."""
import numpy as np
import pytest
import torch

from pointnav_vo_amd import model_spec as ms
from pointnav_vo_amd import synth
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd import vo_cnn  # noqa: F401

FULL = ["rgb", "depth", "discretized_depth", "top_down_view"]
KW = dict(observation_size=(341, 192), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
          output_dim=3, dropout_p=0.2)


def test_registry_has_every_reference_name():
    names = ["vo_cnn", "vo_cnn_rgb", "vo_cnn_wider", "vo_cnn_deeper", "vo_cnn_rgb_d_dd", "vo_cnn_rgb_d_top_down",
             "vo_cnn_rgb_dd_top_down", "vo_cnn_d_dd_top_down", "vo_cnn_rgb_d_dd_top_down",
             "vo_cnn_discretize_depth_top_down", "vo_cnn_act_embed", "vo_cnn_wider_act_embed"]
    for n in names:
        assert baseline_registry.get_vo_model(n) is not None, n
    assert baseline_registry.get_vo_model("nope") is None


def test_default_model_work_figures_match_survey():
    cfg = ms.config_from_kwargs(observation_space=FULL, discretized_depth_channels=10, **KW)
    assert cfg.in_channels == 30 and cfg.final_hw == (6, 11) and cfg.comp_channels == 31
    assert ms.macs_per_pair(cfg) == 1342236672
    assert ms.streaming_bytes_per_pair(cfg) == 22493692
    spec = ms.state_dict_spec(cfg)
    assert sum(int(np.prod(s)) for n, s in spec if not n.split(".")[-1].startswith("_")) == 3962305


def test_state_dict_round_trip_and_shapes():
    m = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(observation_space=FULL,
                                                                   discretized_depth_channels=10, **KW)
    sd = m.state_dict()
    assert sd["visual_encoder.backbone.conv1.0.weight"].shape == (32, 30, 7, 7)
    assert sd["visual_encoder.backbone.layer4.0.downsample.0.weight"].shape == (256, 128, 1, 1)
    assert sd["visual_encoder.compression.0.weight"].shape == (31, 256, 3, 3)
    assert sd["visual_fc.2.weight"].shape == (512, 2046) and sd["output_head.1.weight"].shape == (3, 512)
    syn = synth.make_state_dict(ms.state_dict_spec(m.cfg), seed=3)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.items()})       # strict
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), syn[k])
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in sd.items() if "conv1" not in k})           # missing keys -> error


def test_constructor_asserts_mirror_reference():
    with pytest.raises(AssertionError):     # vo_cnn.py:254
        baseline_registry.get_vo_model("vo_cnn")(observation_space=FULL, **KW)
    with pytest.raises(AssertionError):     # vo_cnn.py:501
        baseline_registry.get_vo_model("vo_cnn_d_dd_top_down")(observation_space=FULL, discretized_depth_channels=10, **KW)
    with pytest.raises(AssertionError):     # vo_cnn.py:355: the deeper variant is resnet101 only
        baseline_registry.get_vo_model("vo_cnn_deeper")(observation_space=["rgb", "depth"], **KW)
    deeper = baseline_registry.get_vo_model("vo_cnn_deeper")(observation_space=["rgb", "depth"],
                                                             **dict(KW, backbone="resnet101"))
    assert sum(k.endswith("convs.6.weight") for k in deeper.state_dict()) == 3 + 4 + 23 + 3   # Bottleneck blocks


def test_forward_refuses_cpu():
    """No CPU fallback in either mode: the train-mode forward ('rnd' mode / training) and the eval forward both need the
    HIP library and a model on the GPU."""
    m = baseline_registry.get_vo_model("vo_cnn")(observation_space=["rgb", "depth"], **KW)
    obs = {"rgb": torch.zeros(1, 192, 341, 6), "depth": torch.zeros(1, 192, 341, 2)}
    with pytest.raises(RuntimeError, match="MI355X"):
        m(obs)                                              # nn.Module default: training mode
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(obs)


def test_synth_is_deterministic_and_shard_consistent():
    a = synth.make_obs_pairs(3, 40, 48, observation_space=FULL, seed=7)
    b = synth.make_obs_pairs(2, 40, 48, observation_space=FULL, seed=7, start=1)
    for k in a:
        np.testing.assert_array_equal(a[k][1:], b[k])
    assert a["discretized_depth"].sum() == 3 * 40 * 48 * 2
    assert a["rgb"].min() >= 0 and a["rgb"].max() <= 255 and a["rgb"].dtype == np.float32


def test_policy_exposes_what_the_reference_trainers_read():
    """ppo_trainer.py:618 / ddppo_trainer.py:279 read policy.net.num_recurrent_layers (and net.output_size) to allocate the
    recurrent state; ddppo_trainer.py:150 calls policy.net.visual_encoder.load_state_dict(...)."""
    from types import SimpleNamespace
    from pointnav_vo_amd import policy as pol
    space = SimpleNamespace(spaces={"depth": SimpleNamespace(shape=(192, 341, 1)), "pointgoal_with_gps_compass": SimpleNamespace(shape=(2,))})
    p = baseline_registry.get_policy("resnet_rnn_policy")(observation_space=space, action_space=SimpleNamespace(n=4), hidden_size=512,
                                                          num_recurrent_layers=2, rnn_type="LSTM", resnet_baseplanes=32,
                                                          backbone="resnet18", normalize_visual_inputs=False)
    assert p.net.num_recurrent_layers == 4 and p.num_recurrent_layers == 4          # LSTM: h and c per layer
    assert p.net.output_size == 512 and p.net.is_blind is False
    enc_sd = p.net.visual_encoder.state_dict()
    p.net.visual_encoder.load_state_dict(enc_sd)                                     # a Module with the reference's keys
    assert any(k.startswith("backbone.") for k in enc_sd)


def test_set_precision_is_validated_on_the_host():
    m = baseline_registry.get_vo_model("vo_cnn")(observation_space=["rgb", "depth"], **KW)
    assert m.set_precision("bfloat16") is m and m._precision == "bfloat16"
    with pytest.raises(ValueError):
        m.set_precision("float16")


def test_mfma_utilisation_table_regenerates_from_the_committed_profiles():
    """tools/mfma_util.py on the newest committed counter / trace summaries of profiles/ (the evidence DESIGN.md quotes)."""
    import glob
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traces = sorted(glob.glob(os.path.join(root, "profiles", "r2*_kernel_trace_fwd_fp32.md")))
    assert traces
    tag = os.path.basename(traces[-1]).split("_")[0]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_util.py"), traces[-1],
                        os.path.join(root, "profiles", f"{tag}_pmc_fwd_fp32.md")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    stem = [ln for ln in r.stdout.splitlines() if ln.startswith("| stem_mx_kernel")]
    assert stem and 0.3 < float(stem[0].split("|")[5]) < 1.0, r.stdout[:600]
