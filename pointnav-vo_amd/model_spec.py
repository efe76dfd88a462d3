"""Architecture description of the reference VO networks, derived from constructor kwargs only.

Mirrors the shape logic of
  ResNetEncoder.__init__            /root/reference/pointnav_vo/vo/models/vo_cnn.py:17-107
  VisualOdometryCNNBase.__init__    vo_cnn.py:183-227
  VisualOdometryCNNActEmbed         /root/reference/pointnav_vo/vo/models/vo_cnn_act_embed.py:17-59
  ResNet / resnet18                 /root/reference/pointnav_vo/model_utils/visual_encoders/resnet.py:153-229
so that the host side can (a) name every tensor exactly as the reference's state_dict does and (b) size the
HIP workspace.  No torch import here.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

MODALITIES = ("rgb", "depth", "discretized_depth", "top_down_view")
RGB_PAIR_CHANNEL = 6          # vo/common/common_vars.py:53
DEPTH_PAIR_CHANNEL = 2        # :54
TOP_DOWN_VIEW_PAIR_CHANNEL = 2  # :55
EMBED_DIM = 32                # :52
N_ACTS = 4                    # :9


def _half(n):
    return (n - 1) // 2 + 1


@dataclass
class VOConfig:
    width: int
    height: int
    n_rgb: int = 0            # PAIR channel counts (6 / 2 / 2*bins / 2) or 0 when the modality is absent
    n_depth: int = 0
    n_dd: int = 0
    n_tdv: int = 0
    baseplanes: int = 32
    hidden: int = 512
    out_dim: int = 3
    normalize: bool = True
    act_embed: bool = False
    n_acts: int = N_ACTS
    after_compression_flat_size: int = 2048
    blocks: Tuple[int, int, int, int] = (2, 2, 2, 2)   # resnet18, resnet.py:226-229
    bottleneck: bool = False                           # resnet50 / resnet101: Bottleneck blocks, expansion 4 (:93-117)
    backbone_depth: int = 18

    @property
    def ngroups(self):        # vo_cnn.py:206
        return self.baseplanes // 2

    @property
    def in_channels(self):    # vo_cnn.py:60-65
        return self.n_rgb + self.n_depth + self.n_dd + self.n_tdv

    @property
    def stem_hw(self):
        return _half(self.height), _half(self.width)

    @property
    def pool_hw(self):
        h, w = self.stem_hw
        return _half(h), _half(w)

    def layer_hw(self, li):   # li in 1..4
        h, w = self.pool_hw
        for _ in range(li - 1):
            h, w = _half(h), _half(w)
        return h, w

    @property
    def final_hw(self):       # == ceil(H/32), ceil(W/32)  (vo_cnn.py:76-81)
        return self.layer_hw(4)

    @property
    def comp_channels(self):  # vo_cnn.py:82-84
        fh, fw = self.final_hw
        return int(round(self.after_compression_flat_size / (fw * fh)))

    @property
    def flat_features(self):
        fh, fw = self.final_hw
        return self.comp_channels * fh * fw

    @property
    def fc_in(self):
        return self.flat_features + (EMBED_DIM if self.act_embed else 0)


def config_from_kwargs(*, observation_space, observation_size, hidden_size=512, resnet_baseplanes=32,
                       backbone="resnet18", normalize_visual_inputs=False, output_dim=4, dropout_p=0.2,
                       discretized_depth_channels=0, after_compression_flat_size=2048,
                       rgb_pair_channel=RGB_PAIR_CHANNEL, depth_pair_channel=DEPTH_PAIR_CHANNEL,
                       top_down_view_pair_channel=TOP_DOWN_VIEW_PAIR_CHANNEL, act_embed=False,
                       n_acts=N_ACTS) -> VOConfig:
    """Same keyword contract as the reference constructors (vo_cnn.py:183-198; called at
    rl/common/base_trainer_with_vo.py:68-80).  dropout_p is accepted and irrelevant in eval()."""
    depths = {"resnet18": (18, (2, 2, 2, 2), False), "resnet50": (50, (3, 4, 6, 3), True),
              "resnet101": (101, (3, 4, 23, 3), True)}               # resnet.py:226-241
    if backbone not in depths:
        raise NotImplementedError(f"backbone {backbone!r}: resnet18 / resnet50 / resnet101 are built (no SE / ResNeXt)")
    depth, blocks, bottleneck = depths[backbone]
    w, h = observation_size
    return VOConfig(
        width=int(w), height=int(h),
        n_rgb=rgb_pair_channel if "rgb" in observation_space else 0,
        n_depth=depth_pair_channel if "depth" in observation_space else 0,
        n_dd=2 * discretized_depth_channels if "discretized_depth" in observation_space else 0,
        n_tdv=top_down_view_pair_channel if "top_down_view" in observation_space else 0,
        baseplanes=int(resnet_baseplanes), hidden=int(hidden_size), out_dim=int(output_dim),
        normalize=bool(normalize_visual_inputs), act_embed=bool(act_embed), n_acts=int(n_acts),
        after_compression_flat_size=int(after_compression_flat_size), blocks=blocks, bottleneck=bottleneck,
        backbone_depth=depth,
    )


@dataclass
class ConvDesc:
    name: str          # state_dict prefix of the conv weight (without ".weight")
    gn: str            # state_dict prefix of the GroupNorm that follows
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    hin: int
    win: int
    groups: int        # GroupNorm groups
    hout: int = field(init=False)
    wout: int = field(init=False)

    def __post_init__(self):
        self.hout = (self.hin + 2 * self.pad - self.k) // self.stride + 1
        self.wout = (self.win + 2 * self.pad - self.k) // self.stride + 1

    @property
    def macs(self):
        return self.hout * self.wout * self.cout * self.cin * self.k * self.k


def conv_plan(cfg: VOConfig) -> List[ConvDesc]:
    """Every conv of the forward in execution order (resnet.py:214-223, vo_cnn.py:177-178)."""
    bb = "visual_encoder.backbone."
    g = cfg.ngroups
    plan = [ConvDesc(bb + "conv1.0", bb + "conv1.1", cfg.in_channels, cfg.baseplanes, 7, 2, 3,
                     cfg.height, cfg.width, g)]
    h, w = cfg.pool_hw
    cin = cfg.baseplanes
    for li in range(1, 5):
        planes = cfg.baseplanes * (2 ** (li - 1))
        for bi in range(cfg.blocks[li - 1]):
            p = bb + f"layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            if cfg.bottleneck:                                  # 1x1 -> 3x3 (stride) -> 1x1 (x4), resnet.py:58-69
                b1 = ConvDesc(p + "convs.0", p + "convs.1", cin, planes, 1, 1, 0, h, w, g)
                b2 = ConvDesc(p + "convs.3", p + "convs.4", planes, planes, 3, stride, 1, h, w, g)
                b3 = ConvDesc(p + "convs.6", p + "convs.7", planes, planes * 4, 1, 1, 0, b2.hout, b2.wout, g)
                plan += [b1, b2, b3]
                if stride != 1 or cin != planes * 4:            # resnet.py:190-195
                    plan.append(ConvDesc(p + "downsample.0", p + "downsample.1", cin, planes * 4, 1, stride, 0, h, w, g))
                h, w, cin = b2.hout, b2.wout, planes * 4
                continue
            c1 = ConvDesc(p + "convs.0", p + "convs.1", cin, planes, 3, stride, 1, h, w, g)
            plan.append(c1)
            plan.append(ConvDesc(p + "convs.3", p + "convs.4", planes, planes, 3, 1, 1, c1.hout, c1.wout, g))
            if stride != 1 or cin != planes:
                plan.append(ConvDesc(p + "downsample.0", p + "downsample.1", cin, planes, 1, stride, 0, h, w, g))
            h, w, cin = c1.hout, c1.wout, planes
    plan.append(ConvDesc("visual_encoder.compression.0", "visual_encoder.compression.1", cin,
                         cfg.comp_channels, 3, 1, 1, h, w, 1))
    return plan


def state_dict_spec(cfg: VOConfig):
    """[(name, shape)] in the reference's state_dict order."""
    spec = []
    if cfg.act_embed:
        spec.append(("action_embedding.weight", (cfg.n_acts + 1, EMBED_DIM)))
    if cfg.normalize:
        c = cfg.in_channels
        pre = "visual_encoder.running_mean_and_var."
        spec += [(pre + "_mean", (1, c, 1, 1)), (pre + "_var", (1, c, 1, 1)), (pre + "_count", ())]
    for cd in conv_plan(cfg):
        spec.append((cd.name + ".weight", (cd.cout, cd.cin, cd.k, cd.k)))
        spec.append((cd.gn + ".weight", (cd.cout,)))
        spec.append((cd.gn + ".bias", (cd.cout,)))
    fc = "hidden_generator.1" if cfg.act_embed else "visual_fc.2"
    spec.append((fc + ".weight", (cfg.hidden, cfg.fc_in)))
    spec.append((fc + ".bias", (cfg.hidden,)))
    spec.append(("output_head.1.weight", (cfg.out_dim, cfg.hidden)))
    spec.append(("output_head.1.bias", (cfg.out_dim,)))
    return spec


def macs_per_pair(cfg: VOConfig) -> int:
    """conv + linear multiply-accumulates per frame pair (1 342 236 672 for the default)."""
    return sum(cd.macs for cd in conv_plan(cfg)) + cfg.fc_in * cfg.hidden + cfg.hidden * cfg.out_dim


def streaming_bytes_per_pair(cfg: VOConfig) -> int:
    """fp32 layer-streaming traffic model of SURVEY.md §8(d): every conv / pool / linear input read once and
    output written once, GroupNorm fused (22 493 692 B for the default model)."""
    total = 0
    for cd in conv_plan(cfg):
        total += 4 * (cd.hin * cd.win * cd.cin + cd.hout * cd.wout * cd.cout)
    sh, sw = cfg.stem_hw
    ph, pw = cfg.pool_hw
    total += 4 * (sh * sw * cfg.baseplanes + ph * pw * cfg.baseplanes)      # maxpool
    total += 4 * (cfg.fc_in + cfg.hidden) + 4 * (cfg.hidden + cfg.out_dim)  # linears
    return total
