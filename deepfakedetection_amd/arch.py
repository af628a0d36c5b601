"""EfficientNet layer plans for the HIP engine.

Two published parameterisations of the same MBConv stack are supported because the
reference uses both lineages:

  flavour "lukemelas"  efficientnet_pytorch 0.7.1 — what the reference's trainer and
      registry instantiate (trainers/efficientnet.py:405, model_registry.py:32-36):
      TF "SAME" padding frozen at construction from the variant's NOMINAL resolution
      (asymmetric: the extra pixel goes bottom/right), BN eps 1e-3 / momentum 0.01,
      SE width int(0.25*block_in), per-block drop-connect 0.2*i/len, head dropout per variant.
  flavour "timm"  timm 1.0.20 `efficientnet_bX` — the BASELINE.json configuration:
      symmetric k//2 padding, BN eps 1e-5 / momentum 0.1, SE width round(0.25*block_in),
      drop_path 0, dropout 0.2 (b0).

A plan is pure data: the modules in efficientnet.py turn it into parameters whose
state-dict keys equal the third-party packages', and the fused functions consume the
per-layer geometry (kernel, stride, leading/trailing padding).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

# stage table of EfficientNet-B0: repeats, kernel, stride, expansion, cin, cout
_B0_STAGES = [
    dict(repeats=1, kernel=3, stride=1, expand=1, cin=32, cout=16),
    dict(repeats=2, kernel=3, stride=2, expand=6, cin=16, cout=24),
    dict(repeats=2, kernel=5, stride=2, expand=6, cin=24, cout=40),
    dict(repeats=3, kernel=3, stride=2, expand=6, cin=40, cout=80),
    dict(repeats=3, kernel=5, stride=1, expand=6, cin=80, cout=112),
    dict(repeats=4, kernel=5, stride=2, expand=6, cin=112, cout=192),
    dict(repeats=1, kernel=3, stride=1, expand=6, cin=192, cout=320),
]

# variant -> width multiplier, depth multiplier, nominal resolution, head dropout
_COMPOUND = {
    "b0": (1.0, 1.0, 224, 0.2),
    "b1": (1.0, 1.1, 240, 0.2),
    "b2": (1.1, 1.2, 260, 0.3),
    "b3": (1.2, 1.4, 300, 0.3),
    "b4": (1.4, 1.8, 380, 0.4),
}


def scale_channels(channels: int, width: float, divisor: int = 8) -> int:
    """Width scaling with the 'never lose more than 10 %' rule."""
    scaled = channels * width
    snapped = max(divisor, (int(scaled + divisor / 2) // divisor) * divisor)
    return snapped + divisor if snapped < 0.9 * scaled else snapped


def scale_repeats(repeats: int, depth: float) -> int:
    return math.ceil(depth * repeats)


def tf_same_padding(size: int, kernel: int, stride: int) -> tuple[int, int]:
    """(leading, trailing) zero padding TensorFlow's SAME uses for an input of `size`."""
    needed = max((math.ceil(size / stride) - 1) * stride + kernel - size, 0)
    return needed // 2, needed - needed // 2


@dataclass(frozen=True)
class ConvGeom:
    kernel: int
    stride: int
    pad_lead: int      # top == left (square inputs are not assumed; sizes come at run time)
    pad_trail: int

    def out_size(self, size: int) -> int:
        return (size + self.pad_lead + self.pad_trail - self.kernel) // self.stride + 1


@dataclass(frozen=True)
class BlockPlan:
    index: int
    stage: int
    pos: int                 # index inside its stage
    cin: int
    cmid: int
    cout: int
    se_width: int
    expand: bool
    dw: ConvGeom
    skip: bool
    drop_connect: float


@dataclass(frozen=True)
class NetPlan:
    variant: str
    flavour: str
    stem_out: int
    stem: ConvGeom
    blocks: tuple[BlockPlan, ...]
    head_out: int
    dropout: float
    bn_eps: float
    bn_momentum: float
    stage_sizes: tuple[int, ...] = field(default=())


def efficientnet_plan(variant: str, flavour: str) -> NetPlan:
    if variant not in _COMPOUND:
        raise KeyError(f"unknown EfficientNet variant '{variant}'")
    if flavour not in ("lukemelas", "timm"):
        raise KeyError(f"unknown flavour '{flavour}'")
    width, depth, nominal, dropout = _COMPOUND[variant]
    same = flavour == "lukemelas"
    res = nominal
    if same:
        stem = ConvGeom(3, 2, *tf_same_padding(res, 3, 2))
    else:
        stem = ConvGeom(3, 2, 1, 1)
    res = math.ceil(res / 2)
    reps = [scale_repeats(s["repeats"], depth) for s in _B0_STAGES]
    n_blocks = sum(reps)
    blocks: list[BlockPlan] = []
    for stage, (spec, count) in enumerate(zip(_B0_STAGES, reps)):
        stage_in = scale_channels(spec["cin"], width)
        stage_out = scale_channels(spec["cout"], width)
        for pos in range(count):
            cin = stage_in if pos == 0 else stage_out
            stride = spec["stride"] if pos == 0 else 1
            k = spec["kernel"]
            if same:
                geom = ConvGeom(k, stride, *tf_same_padding(res, k, stride))
                se_width = max(1, int(cin * 0.25))
                dc = 0.2 * len(blocks) / n_blocks
            else:
                geom = ConvGeom(k, stride, k // 2, k // 2)
                se_width = int(round(cin * 0.25))
                dc = 0.0
            blocks.append(BlockPlan(
                index=len(blocks), stage=stage, pos=pos, cin=cin, cmid=cin * spec["expand"], cout=stage_out,
                se_width=se_width, expand=spec["expand"] != 1, dw=geom,
                skip=(stride == 1 and cin == stage_out), drop_connect=dc,
            ))
            res = math.ceil(res / stride)
    return NetPlan(
        variant=variant, flavour=flavour, stem_out=scale_channels(32, width), stem=stem, blocks=tuple(blocks),
        head_out=scale_channels(1280, width), dropout=dropout,
        bn_eps=1e-3 if same else 1e-5, bn_momentum=0.01 if same else 0.1, stage_sizes=tuple(reps),
    )


__all__ = ["BlockPlan", "ConvGeom", "NetPlan", "efficientnet_plan", "scale_channels", "scale_repeats", "tf_same_padding"]
