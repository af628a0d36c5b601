"""TEST INFRASTRUCTURE — CPU oracle for the EfficientNet forward/backward the reference runs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY PINNING: the arithmetic of the reference's hot path lives in un-vendored
third-party packages that are absent here (efficientnet_pytorch==0.7.1,
timm==1.0.20; requirements.txt:14,95 of the reference) and the reference's own tests
hold no golden vector for it (tests/test_repo_smoke.py:10-22).  This file restates the
published architectures with plain torch.nn.functional CPU ops and is pinned by
  * parameter-count known answers (B0 5,288,548; B3 12,233,232 @1000 classes),
  * state-dict key sets of both naming schemes,
  * an independent topology + logits cross-check against
    transformers.EfficientNetForImageClassification (TF-"SAME" lineage, eps 1e-3),
see tests/test_oracle.py.  Against the pinned third-party packages themselves parity is
UNPINNED (they cannot be imported in this container).

Reference call sites this module stands in for:
  trainers/efficientnet.py:405-407  EfficientNet.from_pretrained("efficientnet-b3"), _fc swap
  orchestration/model_registry.py:32-36  EfficientNet.from_name("efficientnet-b3")
  trainers/efficientnet.py:297,302  forward under autocast, backward
Flavours:
  "lukemelas": efficientnet_pytorch 0.7.1 — TF static-SAME padding from the model's
      nominal image size, BN eps 1e-3 / momentum 0.01, drop-connect, keys _conv_stem/_blocks.N/...
  "timm": timm 1.0.20 efficientnet_bX — symmetric k//2 padding, BN eps 1e-5 / momentum 0.1,
      keys conv_stem/blocks.S.B/...
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import nn

# (repeats, kernel, stride, expand, in, out) — the EfficientNet-B0 base table
BASE_STAGES = (
    (1, 3, 1, 1, 32, 16),
    (2, 3, 2, 6, 16, 24),
    (2, 5, 2, 6, 24, 40),
    (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112),
    (4, 5, 2, 6, 112, 192),
    (1, 3, 1, 6, 192, 320),
)
# name -> (width, depth, nominal resolution, dropout)
SCALING = {
    "b0": (1.0, 1.0, 224, 0.2),
    "b1": (1.0, 1.1, 240, 0.2),
    "b2": (1.1, 1.2, 260, 0.3),
    "b3": (1.2, 1.4, 300, 0.3),
    "b4": (1.4, 1.8, 380, 0.4),
}


def round_filters(filters: int, width: float, divisor: int = 8) -> int:
    f = filters * width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def round_repeats(repeats: int, depth: float) -> int:
    return int(math.ceil(depth * repeats))


@dataclass
class BlockCfg:
    k: int
    stride: int
    expand: int
    cin: int
    cout: int
    se_ch: int
    pad_dw: tuple[int, int, int, int]   # left, right, top, bottom
    drop_connect: float
    stage: int
    index_in_stage: int


def _same_pad_1d(size: int, k: int, s: int) -> tuple[int, int]:
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def build_cfg(variant: str, flavour: str):
    width, depth, res, dropout = SCALING[variant]
    stem = round_filters(32, width)
    head = round_filters(1280, width)
    blocks: list[BlockCfg] = []
    size = res
    # lukemelas: the stem's static padding comes from the nominal resolution
    stem_pad = _same_pad_1d(size, 3, 2) if flavour == "lukemelas" else (1, 1)
    size = -(-size // 2)
    total = sum(round_repeats(r, depth) for r, *_ in BASE_STAGES)
    idx = 0
    for si, (r, k, s, e, ci, co) in enumerate(BASE_STAGES):
        ci, co = round_filters(ci, width), round_filters(co, width)
        for bi in range(round_repeats(r, depth)):
            stride = s if bi == 0 else 1
            cin = ci if bi == 0 else co
            if flavour == "lukemelas":
                a, b = _same_pad_1d(size, k, stride)
                pad = (a, b, a, b)
                se_ch = max(1, int(cin * 0.25))
                dc = 0.2 * idx / total
            else:
                pad = (k // 2,) * 4
                se_ch = int(round(cin * 0.25))
                dc = 0.0
            blocks.append(BlockCfg(k, stride, e, cin, co, se_ch, pad, dc, si, bi))
            size = -(-size // stride)
            idx += 1
    return stem, stem_pad, blocks, head, dropout


def _bn(c: int, flavour: str) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01) if flavour == "lukemelas" else nn.BatchNorm2d(c, eps=1e-5, momentum=0.1)


def _conv(ci: int, co: int, k: int, s: int, groups: int = 1, bias: bool = False) -> nn.Conv2d:
    return nn.Conv2d(ci, co, k, stride=s, padding=0, groups=groups, bias=bias)


def _run_conv(x: torch.Tensor, m: nn.Conv2d, pad=(0, 0, 0, 0)) -> torch.Tensor:
    if any(pad):
        x = F.pad(x, pad)
    return F.conv2d(x, m.weight, m.bias, m.stride, 0, 1, m.groups)


def _run_bn(x: torch.Tensor, m: nn.BatchNorm2d) -> torch.Tensor:
    if m.training and m.num_batches_tracked is not None:
        m.num_batches_tracked.add_(1)
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, m.training, m.momentum, m.eps)


class _SE(nn.Module):
    def __init__(self, ch: int, rd: int, names: tuple[str, str]) -> None:
        super().__init__()
        self.names = names
        setattr(self, names[0], _conv(ch, rd, 1, 1, bias=True))
        setattr(self, names[1], _conv(rd, ch, 1, 1, bias=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        s = x.mean((2, 3), keepdim=True)
        s = _run_conv(F.silu(_run_conv(s, getattr(self, self.names[0]))), getattr(self, self.names[1]))
        return x * torch.sigmoid(s)


class _BlockLM(nn.Module):
    """efficientnet_pytorch MBConvBlock (attribute names are its state-dict keys)."""

    def __init__(self, c: BlockCfg) -> None:
        super().__init__()
        self.c = c
        mid = c.cin * c.expand
        if c.expand != 1:
            self._expand_conv = _conv(c.cin, mid, 1, 1)
            self._bn0 = _bn(mid, "lukemelas")
        self._depthwise_conv = _conv(mid, mid, c.k, c.stride, groups=mid)
        self._bn1 = _bn(mid, "lukemelas")
        self._se_reduce = _conv(mid, c.se_ch, 1, 1, bias=True)
        self._se_expand = _conv(c.se_ch, mid, 1, 1, bias=True)
        self._project_conv = _conv(mid, c.cout, 1, 1)
        self._bn2 = _bn(c.cout, "lukemelas")

    def forward(self, x: torch.Tensor, drop_mask: torch.Tensor | None = None) -> torch.Tensor:
        c, inp = self.c, x
        if c.expand != 1:
            x = F.silu(_run_bn(_run_conv(x, self._expand_conv), self._bn0))
        x = F.silu(_run_bn(_run_conv(x, self._depthwise_conv, c.pad_dw), self._bn1))
        s = x.mean((2, 3), keepdim=True)
        s = _run_conv(F.silu(_run_conv(s, self._se_reduce)), self._se_expand)
        x = torch.sigmoid(s) * x
        x = _run_bn(_run_conv(x, self._project_conv), self._bn2)
        if c.stride == 1 and c.cin == c.cout:
            if drop_mask is not None:
                x = x * drop_mask
            x = x + inp
        return x


class _BlockTimm(nn.Module):
    """timm DepthwiseSeparableConv (expand == 1) / InvertedResidual."""

    def __init__(self, c: BlockCfg) -> None:
        super().__init__()
        self.c = c
        mid = c.cin * c.expand
        if c.expand == 1:
            self.conv_dw = _conv(mid, mid, c.k, c.stride, groups=mid)
            self.bn1 = _bn(mid, "timm")
            self.se = _SE(mid, c.se_ch, ("conv_reduce", "conv_expand"))
            self.conv_pw = _conv(mid, c.cout, 1, 1)
            self.bn2 = _bn(c.cout, "timm")
        else:
            self.conv_pw = _conv(c.cin, mid, 1, 1)
            self.bn1 = _bn(mid, "timm")
            self.conv_dw = _conv(mid, mid, c.k, c.stride, groups=mid)
            self.bn2 = _bn(mid, "timm")
            self.se = _SE(mid, c.se_ch, ("conv_reduce", "conv_expand"))
            self.conv_pwl = _conv(mid, c.cout, 1, 1)
            self.bn3 = _bn(c.cout, "timm")

    def forward(self, x: torch.Tensor, drop_mask: torch.Tensor | None = None) -> torch.Tensor:
        c, inp = self.c, x
        if c.expand == 1:
            x = F.silu(_run_bn(_run_conv(x, self.conv_dw, c.pad_dw), self.bn1))
            x = self.se(x)
            x = _run_bn(_run_conv(x, self.conv_pw), self.bn2)
        else:
            x = F.silu(_run_bn(_run_conv(x, self.conv_pw), self.bn1))
            x = F.silu(_run_bn(_run_conv(x, self.conv_dw, c.pad_dw), self.bn2))
            x = self.se(x)
            x = _run_bn(_run_conv(x, self.conv_pwl), self.bn3)
        if c.stride == 1 and c.cin == c.cout:
            x = x + inp
        return x


class EfficientNetRef(nn.Module):
    """CPU restatement; `variant` in SCALING, `flavour` in {"lukemelas", "timm"}."""

    def __init__(self, variant: str = "b0", flavour: str = "timm", num_classes: int = 1000) -> None:
        super().__init__()
        self.flavour = flavour
        stem, self.stem_pad, cfgs, head, self.dropout = build_cfg(variant, flavour)
        self.cfgs = cfgs
        last = cfgs[-1].cout
        if flavour == "lukemelas":
            self._conv_stem = _conv(3, stem, 3, 2)
            self._bn0 = _bn(stem, flavour)
            self._blocks = nn.ModuleList([_BlockLM(c) for c in cfgs])
            self._conv_head = _conv(last, head, 1, 1)
            self._bn1 = _bn(head, flavour)
            self._fc = nn.Linear(head, num_classes)
        else:
            self.conv_stem = _conv(3, stem, 3, 2)
            self.bn1 = _bn(stem, flavour)
            stages: list[list[nn.Module]] = [[] for _ in BASE_STAGES]
            for c in cfgs:
                stages[c.stage].append(_BlockTimm(c))
            self.blocks = nn.Sequential(*[nn.Sequential(*s) for s in stages])
            self.conv_head = _conv(last, head, 1, 1)
            self.bn2 = _bn(head, flavour)
            self.classifier = nn.Linear(head, num_classes)

    def block_list(self) -> list[nn.Module]:
        if self.flavour == "lukemelas":
            return list(self._blocks)
        return [b for stage in self.blocks for b in stage]

    def forward(self, x: torch.Tensor, drop_masks: list[torch.Tensor | None] | None = None,
                dropout_mask: torch.Tensor | None = None) -> torch.Tensor:
        """drop_masks / dropout_mask: explicit (already 1/keep-scaled) masks so a test can
        feed the SAME randomness to the HIP engine; None disables the stochastic parts."""
        lm = self.flavour == "lukemelas"
        a, b = self.stem_pad
        stem, bn0 = (self._conv_stem, self._bn0) if lm else (self.conv_stem, self.bn1)
        x = F.silu(_run_bn(_run_conv(x, stem, (a, b, a, b)), bn0))
        for i, blk in enumerate(self.block_list()):
            x = blk(x, None if drop_masks is None else drop_masks[i])
        head, bnh, fc = (self._conv_head, self._bn1, self._fc) if lm else (self.conv_head, self.bn2, self.classifier)
        y = _run_conv(x, head)
        for hook in head._forward_hooks.values():          # the third-party modules call conv_head as a module:
            r = hook(head, (x,), y)                        # forward hooks (Grad-CAM, web_ui.py:96-114) see its output
            if r is not None:
                y = r
        x = F.silu(_run_bn(y, bnh))
        x = x.mean((2, 3))
        if dropout_mask is not None:
            x = x * dropout_mask
        return fc(x)


def train_step_ref(model: nn.Module, opt: torch.optim.Optimizer, x: torch.Tensor, y: torch.Tensor,
                   label_smoothing: float = 0.1) -> float:
    """One optimizer step as trainers/efficientnet.py:296-309 does it (accum_steps=1, no AMP on CPU)."""
    model.train()
    opt.zero_grad(set_to_none=True)
    loss = F.cross_entropy(model(x), y, label_smoothing=label_smoothing)
    loss.backward()
    opt.step()
    return float(loss.item())
