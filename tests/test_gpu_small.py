"""GPU (-m gpu): the persistent small-batch kernel (csrc/smallnet.hip; options small_net / small_max / small_coop) — everything
behind the stem conv in one launch for the batch sizes of the navigation loop (rl/ppo/ppo_trainer.py:836-841 calls the VO model
with ONE pair per environment step) — against the reference goldens, the fp64 oracle and the per-layer launches it replaces.
Tolerance as in test_gpu_parity.py: per-pair ||out-ref||_2 / max(||ref||_2, 1e-2) < 1e-4 vs the fp64 reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pair_rel_err
from test_gpu_parity import build, dev

pytestmark = pytest.mark.gpu
TOL = 1e-4

# The kernel takes BasicBlock backbones with channel counts that are multiples of 32 up to 256 and a compression layer of <= 64
# channels (the 6 x 11 map of the 341 x 192 frames: 31; the 64 x 48 goldens end in a 2 x 2 map with 512 and keep the per-layer path).
SMALL_FIXTURES = ["model_default_341x192_b2.npz"]
FIRST_CONV = "visual_encoder.backbone.layer1.0.convs.0"
VARIANTS = [("vo_cnn", "rgb,depth", 0), ("vo_cnn_rgb", "rgb", 0), ("vo_cnn_rgb_d_dd", "rgb,depth,discretized_depth", 10),
            ("vo_cnn_d_dd_top_down", "depth,discretized_depth,top_down_view", 10), ("vo_cnn_rgb_d_dd_top_down", "rgb,depth,discretized_depth,top_down_view", 10),
            ("vo_cnn_act_embed", "rgb,depth,discretized_depth,top_down_view", 10)]


def run(model, tobs, tact):
    with torch.no_grad():
        return (model(tobs, tact) if tact is not None else model(tobs)).cpu().numpy()


@pytest.mark.parametrize("fname", SMALL_FIXTURES)
def test_persistent_kernel_matches_goldens_and_the_per_layer_launches(fname):
    rec = load_golden(fname)
    model, cfg, sd, obs, tobs, actions, tact = build(rec)
    B = rec["out64"].shape[0]
    assert model.layer_kernel(FIRST_CONV, B)[0] == "smallnet"
    small = run(model, tobs, tact)
    assert np.isfinite(small).all()
    assert pair_rel_err(small, rec["out64"]).max() < TOL, fname
    model.set_option("small_net", "off")
    assert model.layer_kernel(FIRST_CONV, B)[0] != "smallnet"
    layers = run(model, tobs, tact)
    assert pair_rel_err(small, layers.astype(np.float64)).max() < 2e-5
    # every sub-batch the kernel takes gives the rows of the full batch (its tiles never mix samples)
    model.set_option("small_net", "on")
    for b in range(1, B):
        sub = run(model, {k: v[:b].contiguous() for k, v in tobs.items()}, tact[:b].contiguous() if tact is not None else None)
        assert np.abs(sub - small[:b]).max() <= 2e-6 * max(1.0, np.abs(small).max())


@pytest.mark.parametrize("name,space,bins", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_model_variants_at_a_size_the_kernel_takes(name, space, bins):
    """The registry's BasicBlock variants (vo_cnn.py:236-557, vo_cnn_act_embed.py:17) with seeded weights at 256 x 128 (4 x 8 final
    map, 64 compression channels): persistent kernel vs the fp64 oracle and vs the per-layer launches, batch 3."""
    from oracle import oracle
    from pointnav_vo_amd import model_spec as ms
    from pointnav_vo_amd import synth
    from pointnav_vo_amd.registry import baseline_registry
    W, H, B = 256, 128, 3
    kw = dict(observation_space=space.split(","), observation_size=(W, H), hidden_size=512, backbone="resnet18",
              normalize_visual_inputs=True, output_dim=3, dropout_p=0.2)
    if bins:
        kw["discretized_depth_channels"] = bins
    model = baseline_registry.get_vo_model(name)(**kw)
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=11)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev()).eval()
    obs = synth.make_obs_pairs(B, H, W, observation_space=space.split(","), dd_bins=bins or 10, seed=3)
    tobs = {k: torch.from_numpy(v).to(dev()) for k, v in obs.items()}
    actions = np.array([1, 3, 2], dtype=np.int64) if "act_embed" in name else None
    tact = torch.from_numpy(actions).to(dev()) if actions is not None else None
    assert model.layer_kernel(FIRST_CONV, B)[0] == "smallnet"
    small = run(model, tobs, tact)
    ref = oracle.forward(sd, obs, ngroups=model.cfg.ngroups, dtype=np.float64, actions=actions)
    assert pair_rel_err(small, ref).max() < TOL
    model.set_option("small_net", "off")
    assert pair_rel_err(small, run(model, tobs, tact).astype(np.float64)).max() < 2e-5


def test_large_frames_take_the_kernel_while_the_batch_fits_its_lds():
    """640 x 480 frames: 15 x 20 final map, 9600 activations per pair in front of the Linear layer — three pairs fit the kernel's
    LDS, four do not (they keep the per-layer launches); 1200 GroupNorm slots per group exercise the consumers' slot loop."""
    from pointnav_vo_amd import model_spec as ms
    from pointnav_vo_amd import synth
    from pointnav_vo_amd.registry import baseline_registry
    W, H = 640, 480
    space = ["rgb", "depth", "discretized_depth", "top_down_view"]
    model = baseline_registry.get_vo_model("vo_cnn_rgb_d_dd_top_down")(
        observation_space=space, observation_size=(W, H), hidden_size=512, backbone="resnet18", normalize_visual_inputs=True,
        output_dim=3, dropout_p=0.2, discretized_depth_channels=10)
    sd = synth.make_state_dict(ms.state_dict_spec(model.cfg), seed=2)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev()).eval()
    assert model.layer_kernel(FIRST_CONV, 3)[0] == "smallnet" and model.layer_kernel(FIRST_CONV, 4)[0] != "smallnet"
    obs = synth.make_obs_pairs(4, H, W, observation_space=space, dd_bins=10, seed=8)
    tobs = {k: torch.from_numpy(v).to(dev()) for k, v in obs.items()}
    sub = {k: v[:3].contiguous() for k, v in tobs.items()}
    small = run(model, sub, None)
    four = run(model, tobs, None)                                   # per-layer launches
    model.set_option("small_net", "off")
    layers = run(model, sub, None)
    assert np.isfinite(small).all()
    assert pair_rel_err(small, layers.astype(np.float64)).max() < 2e-5
    assert pair_rel_err(four[:3], layers.astype(np.float64)).max() < 2e-5


def test_reloaded_weights_reach_the_kernels_own_packing():
    """The kernel keeps its own fragment-ordered copy of every weight: a load_state_dict between two forwards must rebuild it."""
    rec = load_golden("model_default_341x192_b2.npz")
    model, cfg, sd, obs, tobs, actions, tact = build(rec)
    one = {k: v[:1].contiguous() for k, v in tobs.items()}
    before = run(model, one, None)
    sd2 = {k: torch.from_numpy(np.array(v)).clone() for k, v in sd.items()}
    for k in ("visual_encoder.backbone.layer3.1.convs.3.weight", "visual_fc.2.weight", "output_head.1.bias",
              "visual_encoder.backbone.layer2.0.downsample.1.weight"):
        sd2[k] = sd2[k] * 1.25 + 0.01
    model.load_state_dict(sd2)
    after = run(model, one, None)
    assert np.abs(after - before).max() > 1e-3
    model.set_option("small_net", "off")
    assert pair_rel_err(after, run(model, one, None).astype(np.float64)).max() < 2e-5


def test_models_it_does_not_fit_keep_the_per_layer_launches():
    for fname in ("model_wider_64x48_b2.npz",          # 512 channels in the last stage: above the kernel's 256
                  "model_deeper_64x48_b2.npz",         # Bottleneck blocks
                  "model_default_45x37_b3.npz"):       # 2 x 2 final map: 512 compression channels
        rec = load_golden(fname)
        model, cfg, sd, obs, tobs, actions, tact = build(rec)
        assert model.layer_kernel(FIRST_CONV, 1)[0] != "smallnet", fname
        assert pair_rel_err(run(model, tobs, tact), rec["out64"]).max() < TOL


def test_batches_above_small_max_and_explicit_kernel_choices_keep_the_per_layer_launches():
    rec = load_golden("model_default_341x192_b2.npz")
    model, cfg, sd, obs, tobs, actions, tact = build(rec)
    assert model.layer_kernel(FIRST_CONV, 3)[0] == "smallnet"     # default small_max = 3 (round 6: four pairs are faster as per-layer launches)
    assert model.layer_kernel(FIRST_CONV, 4)[0] != "smallnet"
    model.set_option("small_max", "4")
    assert model.layer_kernel(FIRST_CONV, 4)[0] == "smallnet"
    assert model.layer_kernel(FIRST_CONV, 5)[0] != "smallnet"
    model.set_option("small_max", "1")
    assert model.layer_kernel(FIRST_CONV, 2)[0] != "smallnet"
    model.set_option("small_max", "3")
    for key, value in (("conv", "fp32"), ("tail", "separate"), ("pool", "separate")):
        model.set_option(key, value)
        assert model.layer_kernel(FIRST_CONV, 1)[0] != "smallnet", key
        model.set_option(key, {"conv": "auto"}.get(key, "fused"))
    assert model.layer_kernel(FIRST_CONV, 1)[0] == "smallnet"


@pytest.mark.parametrize("coop", ["0", "1"])
def test_alternating_inputs_never_see_a_stale_buffer(coop):
    """The phases of the kernel hand tensors from workgroup to workgroup through buffers that every forward re-uses, with
    agent-scope loads / stores instead of cache write-backs: 400 forwards alternating between four different pairs, each result
    against the per-layer path's."""
    rec = load_golden("model_default_341x192_b2.npz")
    model, cfg, sd, obs, tobs, actions, tact = build(rec)
    g = torch.Generator().manual_seed(5)
    pairs = []
    for i in range(4):
        o = {k: v[i % 2:i % 2 + 1].clone() for k, v in tobs.items()}
        if i >= 2:                                         # two more pairs: the golden ones with their frames swapped
            o = {k: torch.cat([v[..., v.shape[-1] // 2:], v[..., :v.shape[-1] // 2]], dim=-1).contiguous() for k, v in o.items()}
        pairs.append(o)
    model.set_option("small_net", "off")
    refs = [run(model, o, None) for o in pairs]
    model.set_option("small_net", "on")
    model.set_option("small_coop", coop)
    order = torch.randint(0, 4, (400,), generator=g).tolist()
    with torch.no_grad():
        outs = [model(pairs[i]) for i in order]
    torch.cuda.synchronize()
    for i, out in zip(order, outs):
        assert pair_rel_err(out.cpu().numpy(), refs[i].astype(np.float64)).max() < 2e-5


def test_two_handles_on_two_streams_run_their_kernels_side_by_side():
    """Two models (the reference keeps one VO model per action, base_trainer_with_vo.py:83-99) launching their persistent
    kernels from two streams at once: both complete (two workgroups of the kernel fit on a compute unit) and agree with the
    serial results."""
    rec = load_golden("model_default_341x192_b2.npz")
    ma, _, _, _, tobs, _, _ = build(rec)
    mb, _, _, _, _, _, _ = build(rec)
    one = {k: v[:1].contiguous() for k, v in tobs.items()}
    with torch.no_grad():
        ra, rb = ma(one).clone(), mb(one).clone()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for _ in range(50):
            with torch.cuda.stream(sa):
                outs.append((ma(one), ra))
            with torch.cuda.stream(sb):
                outs.append((mb(one), rb))
        torch.cuda.synchronize()
    for got, want in outs:
        assert torch.equal(got, want)
