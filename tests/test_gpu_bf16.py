"""GPU (-m gpu): the native-bf16 forward (BASELINE configs[2]: geometric-invariance dual forward, bf16) through the C ABI.

Tolerance.  bf16 carries 8 significant bits and every stored activation / weight is rounded to it, ~20 layers deep: rounding
noise of ~1-2 % of the output norm is inherent and chaotic from pair to pair.  The yardstick is the reference itself cast to
bfloat16 as a whole (model.bfloat16(); BASELINE.md section 2 measured 1.8e-2 max abs on small outputs; on the 341x192 fixtures here,
outputs of norm 1.1-2.0: per-pair L2 error up to 0.049, RMS 0.032 — tests/golden/dual_bf16_341x192_b6.npz).  This path keeps
the accumulation, the GroupNorm statistics, the whitening constants and both Linear layers in float32 and is held to
    per pair  ||out - ref||_2 <= 1e-2 + 4e-2 * ||ref||_2   against the fp64 reference goldens, and
    RMS over the fixture's forwards <= the whole-model cast's RMS  (measured: 0.024 vs 0.032, 26 % lower).
Properties that do not depend on rounding are exact:
the dual forward's first model equals its single forward BIT FOR BIT (the second one sees its stem input channels in a
permuted K order, so it agrees with the forward on the materialised swapped pair to rounding, not bitwise), results do not
depend on the batch a pair travels in, and runs are reproducible."""
import numpy as np
import pytest
import torch

from conftest import golden_case, load_golden
from oracle import oracle
from pointnav_vo_amd import synth
from pointnav_vo_amd import model_spec as ms
from pointnav_vo_amd.registry import baseline_registry
from pointnav_vo_amd import vo_cnn

pytestmark = pytest.mark.gpu
ABS, REL = 1e-2, 4e-2
BF16_FIXTURES = ["model_default_341x192_b2.npz", "model_default_45x37_b3.npz", "model_vo_cnn_64x48_b2.npz",
                 "model_rgb_d_dd_70x40_b2.npz", "model_d_dd_tdv_66x34_b2.npz", "model_act_embed_64x48_b3.npz"]


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda", 0)


def build(rec, seed=None):
    cfg, sd, obs, actions = golden_case(rec)
    if seed is not None:
        sd = synth.make_state_dict(ms.state_dict_spec(cfg), seed=seed)
    space = str(rec["obs_space"]).split(",")
    kw = dict(observation_space=space, observation_size=(cfg.width, cfg.height), hidden_size=512, backbone="resnet18",
              normalize_visual_inputs=True, output_dim=3, dropout_p=0.2)
    if int(rec["dd_bins"]):
        kw["discretized_depth_channels"] = int(rec["dd_bins"])
    model = baseline_registry.get_vo_model(str(rec["model"]))(**kw)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev()).eval().set_precision("bfloat16")
    tobs = {k: torch.from_numpy(v).to(dev()) for k, v in obs.items()}
    tact = torch.from_numpy(actions).to(dev()) if actions is not None else None
    return model, cfg, sd, obs, tobs, actions, tact


def swapped(obs):
    """(prev, cur) -> (cur, prev): every observation tensor holds [prev half | cur half] on its channel axis."""
    out = {}
    for k, v in obs.items():
        h = v.shape[-1] // 2
        out[k] = (torch.cat([v[..., h:], v[..., :h]], dim=-1) if torch.is_tensor(v) else
                  np.concatenate([v[..., h:], v[..., :h]], axis=-1))
    return out


def bound(ref):
    return ABS + REL * np.linalg.norm(ref, axis=-1)


@pytest.mark.parametrize("fname", BF16_FIXTURES)
def test_bf16_forward_within_tolerance_of_fp64_reference(fname):
    rec = load_golden(fname)
    model, cfg, sd, obs, tobs, actions, tact = build(rec)
    with torch.no_grad():
        out = (model(tobs, tact) if tact is not None else model(tobs)).cpu().numpy().astype(np.float64)
        model.set_precision("float32")
        out32 = (model(tobs, tact) if tact is not None else model(tobs)).cpu().numpy().astype(np.float64)
    ref = rec["out64"]
    err = np.linalg.norm(out - ref, axis=-1)
    assert np.isfinite(out).all() and (err <= bound(ref)).all(), (fname, err, bound(ref), out, ref)
    assert not np.array_equal(out, out32), "set_precision('bfloat16') did not select the bf16 path"
    assert np.abs(out32 - ref).max() < 1e-4, "switching back to float32 must restore the float32 path"


def test_dual_forward_equals_two_single_forwards():
    rec = load_golden("model_default_341x192_b2.npz")
    ma, cfg, sda, obs, tobs, _, _ = build(rec)
    mb, _, sdb, _, _, _, _ = build(rec, seed=77)
    with torch.no_grad():
        oa, ob = vo_cnn.dual_forward(ma, mb, tobs)
        sa = ma(tobs)
        sb = mb(swapped(tobs))
        oa2, ob2 = vo_cnn.dual_forward(ma, mb, tobs)
    assert torch.equal(oa, sa), (oa, sa)
    # model b multiplies the same numbers in a different order inside the stem's MFMAs (its input channels are permuted in
    # K instead of in memory): float32 rounding differences, re-rounded to bf16 downstream
    assert (ob - sb).abs().max().item() < 4e-3, (ob, sb)
    assert torch.equal(oa, oa2) and torch.equal(ob, ob2)                 # reproducible run to run
    # and the swapped-pair model against the fp64 oracle fed the materialised swapped observations
    refb = oracle.forward(sdb, swapped(obs), ngroups=cfg.ngroups, dtype=np.float64)
    errb = np.linalg.norm(ob.cpu().numpy().astype(np.float64) - refb, axis=-1)
    assert (errb <= bound(refb)).all(), (errb, bound(refb))
    with pytest.raises(Exception, match="bfloat16"):
        ma.set_precision("float32")
        vo_cnn.dual_forward(ma, mb, tobs)


def test_dual_forward_against_reference_goldens_and_the_whole_model_bf16_cast():
    """tests/golden/dual_bf16_341x192_b6.npz (generated from the imported reference, gen_golden_bf16.py): fp64 outputs of
    model a on six pairs and of model b on the swapped pairs, and the same forwards with the reference cast to bfloat16 as
    a whole.  Every pair must stay within the stated tolerance of the fp64 reference.  bf16 rounding noise is chaotic from
    pair to pair (either implementation wins on individual pairs), so accuracy is compared as the RMS over the twelve
    forwards: the HIP path keeps accumulation, normalisation statistics and the Linear layers in float32 and must not be
    worse than the whole-model cast."""
    g = load_golden("dual_bf16_341x192_b6.npz")
    rec = load_golden("model_default_341x192_b2.npz")
    ma, cfg, _, _, _, _, _ = build(rec, seed=int(g["seed_a"]))
    mb, _, _, _, _, _, _ = build(rec, seed=int(g["seed_b"]))
    obs = synth.make_obs_pairs(int(g["batch"]), cfg.height, cfg.width, observation_space=str(g["obs_space"]).split(","),
                               dd_bins=int(g["dd_bins"]), seed=int(g["obs_seed"]))
    tobs = {k: torch.from_numpy(v).to(dev()) for k, v in obs.items()}
    with torch.no_grad():
        oa, ob = vo_cnn.dual_forward(ma, mb, tobs)
    ours, cast = [], []
    for tag, o in (("a", oa), ("b", ob)):
        ref = g[f"out64_{tag}"]
        err = np.linalg.norm(o.cpu().numpy().astype(np.float64) - ref, axis=-1)
        assert (err <= bound(ref)).all(), (tag, err, bound(ref))
        ours += err.tolist()
        cast += np.linalg.norm(g[f"cast_{tag}"].astype(np.float64) - ref, axis=-1).tolist()
    rms = lambda v: float(np.sqrt(np.mean(np.square(v))))
    print(f"bf16 per-pair L2 error, RMS over 12 forwards: HIP path {rms(ours):.4f}, reference whole-model cast {rms(cast):.4f}; "
          f"max {max(ours):.4f} vs {max(cast):.4f}")
    assert rms(ours) <= rms(cast), (ours, cast)


def test_dual_forward_at_baseline_batch_256():
    """BASELINE configs[2] size: 256 pairs, two action models.  Every pair's result is independent of the batch it
    travels in (GroupNorm is per sample) and of its position: the first / last pairs equal their small-batch results."""
    rec = load_golden("model_default_341x192_b2.npz")
    ma, cfg, _, _, _, _, _ = build(rec)
    mb, _, _, _, _, _, _ = build(rec, seed=77)
    space = str(rec["obs_space"]).split(",")
    small = synth.make_obs_pairs(3, cfg.height, cfg.width, observation_space=space, dd_bins=10, seed=5)
    tsmall = {k: torch.from_numpy(v).to(dev()) for k, v in small.items()}
    big = {k: v.repeat((86,) + (1,) * (v.dim() - 1))[:256].contiguous() for k, v in tsmall.items()}
    with torch.no_grad():
        ra, rb = vo_cnn.dual_forward(ma, mb, tsmall)
        oa, ob = vo_cnn.dual_forward(ma, mb, big)
    assert oa.shape == (256, 3) and torch.isfinite(oa).all() and torch.isfinite(ob).all()
    for k in (0, 1, 2, 129, 255):
        assert torch.equal(oa[k], ra[k % 3]) and torch.equal(ob[k], rb[k % 3]), k


@pytest.mark.parametrize("pairs", [2, 40, 256])
def test_downsample_ride_equals_the_separate_downsample_conv(pairs):
    """Option ds_fuse in the bf16 path (round 6): the 1x1 stride-2 downsample conv of layer3.0 (resnet.py:192-195; layer2.0 / 4.0 keep their own launch) is computed
    by the launch of the block's first 3x3 stride-2 conv from that conv's centre-tap A fragments (conv_bf16.hip DSF) — in the block-tail
    and in the plain-input form (bf16_fuse off).  Same bf16 products in the same k order; only the
    GroupNorm partial sums are grouped by the riding conv's tiles, so the outputs agree to float32 summation order."""
    rec = load_golden("model_default_341x192_b2.npz")
    ma, cfg, _, _, tobs, _, _ = build(rec)
    mb, _, _, _, _, _, _ = build(rec, seed=77)
    big = {k: v.repeat(((pairs + v.shape[0] - 1) // v.shape[0],) + (1,) * (v.dim() - 1))[:pairs].contiguous() for k, v in tobs.items()}
    res = {}
    with torch.no_grad():
        for fuse in ("on", "off"):
            for v in ("on", "off"):
                for m in (ma, mb):
                    m.set_option("bf16_fuse", fuse)
                    m.set_option("ds_fuse", v)
                res[fuse, v] = [o.clone() for o in vo_cnn.dual_forward(ma, mb, big)] + [ma(big).clone()]
        for m in (ma, mb):
            m.set_option("bf16_fuse", "on")
            m.set_option("ds_fuse", "on")
    for fuse in ("on", "off"):
        for a, b in zip(res[fuse, "on"], res[fuse, "off"]):
            d = (a - b).abs().max().item()
            print(f"pairs {pairs} bf16_fuse {fuse}: ride vs separate max |diff| {d:.3e} (|out| max {b.abs().max().item():.3f})")
            assert torch.isfinite(a).all() and d <= 2e-3 * max(1.0, b.abs().max().item()), (fuse, d)


def test_fused_block_tail_equals_the_separate_residual_pass():
    """conv1 of the next block computes relu(GN2(conv2) + skip) while staging (conv_bf16.hip MODE 2) and writes the block
    output; option bf16_fuse=off keeps the separate residual kernel.  Same float32 expression, same bf16 rounding: the
    network outputs must be identical bit for bit (identity and downsample skips, strided and compression consumers)."""
    rec = load_golden("model_default_341x192_b2.npz")
    model, cfg, sd, obs, tobs, _, _ = build(rec)
    with torch.no_grad():
        fused = model(tobs).clone()
        model.set_option("bf16_fuse", "off")
        plain = model(tobs).clone()
        model.set_option("bf16_fuse", "on")
    assert torch.equal(fused, plain), (fused, plain)
