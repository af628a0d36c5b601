"""TEST INFRASTRUCTURE — CPU oracle for EfficientFormerV2 (timm 1.0.20 `efficientformerv2_s0/s1/s2/l`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The reference builds this model with `timm.create_model("efficientformerv2_s1", pretrained=True,
num_classes=..., img_size=...)` (trainers/efficientformer_v2.py:327; orchestration/model_registry.py:39-40)
and calls it at trainers/efficientformer_v2.py:244 (train), :215 (evaluate), :369 (warm-up) and
orchestration/orchestrator.py:529,590 (inference).  timm is not installable in the build container
(ordinary ModuleNotFoundError; no network), so this file RESTATES the published architecture
(timm/models/efficientformer_v2.py at the pinned version; Li et al., "Rethinking Vision Transformers for
MobileNet Size and Speed", ICCV 2023) with torch.nn.functional ops only, under timm's parameter names, so
that a timm checkpoint loads with strict=True.

PARITY UNPINNED against timm itself: the reference's tests hold no numeric fixture for this model and the
package is absent.  What pins this restatement instead (tests/test_efformer_oracle.py): the parameter
count of every variant against the published model cards (S0 3.6 M, S1 6.19 M, S2 12.7 M, L 26.3 M with
the distillation head), the MAC count of S1 (0.65-0.67 G), the state-dict key grammar, and the shapes the
trainer relies on (`head` / `head_dist` Linear(224, nc), attention only in the last two blocks of stages
2 and 3, 49 query tokens in every attention).

Architecture digest:
  stem      conv3x3 s2 (3 -> C0/2) + BN + GELU, conv3x3 s2 (C0/2 -> C0) + BN + GELU            -> 1/4 resolution
  stage i   [downsample: conv3x3 s2 + BN (+ attention branch on the last stage)] then `depth_i` blocks
  block     [x + ls1 * Attention2d(x)]  (last `num_vit` blocks of stages 2, 3)  then  x + ls2 * ConvMlp(x)
  ConvMlp   1x1 (+bias) BN GELU -> dw3x3 (+bias) BN GELU -> 1x1 (+bias) BN
  Attention2d   [dw3x3 s2 + BN (stage 2)] -> q, k (8 heads x 32), v (8 x 128): 1x1 + BN;  v_local = dw3x3 + BN on v;
                attn = q k^T * 32^-0.5 + bias[|dy|*W + |dx|] -> talking_head1 (1x1 over heads) -> softmax ->
                talking_head2 -> attn v + v_local -> [bilinear x2] -> GELU -> 1x1 + BN
  tail      BN -> mean(H, W) -> (head(x) + head_dist(x)) / 2
Every convolution inside a ConvNorm / ConvNormAct carries a bias (timm default `bias=True`).
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

WIDTHS = {"s0": (32, 48, 96, 176), "s1": (32, 48, 120, 224), "s2": (32, 64, 144, 288), "l": (40, 80, 192, 384)}
DEPTHS = {"s0": (2, 2, 6, 4), "s1": (3, 3, 9, 6), "s2": (4, 4, 12, 8), "l": (5, 5, 15, 10)}
EXPANSION = {
    "s0": ((4, 4), (4, 4), (4, 3, 3, 3, 4, 4), (4, 3, 3, 4)),
    "s1": ((4, 4, 4), (4, 4, 4), (4, 4, 3, 3, 3, 3, 4, 4, 4), (4, 4, 3, 3, 4, 4)),
    "s2": ((4, 4, 4, 4), (4, 4, 4, 4), (4, 4, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4), (4, 4, 3, 3, 3, 3, 4, 4)),
    "l": ((4,) * 5, (4,) * 5, (4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4), (4, 4, 4, 3, 3, 3, 3, 4, 4, 4)),
}
NUM_VIT = {"s0": 2, "s1": 2, "s2": 4, "l": 6}
DROP_PATH = {"s0": 0.0, "s1": 0.0, "s2": 0.02, "l": 0.1}
BN_EPS = 1e-5
LS_INIT = 1e-5


def variant_of(name: str) -> str:
    key = name.lower().replace("-", "_")
    for v in ("s0", "s1", "s2", "l"):
        if key.endswith("_" + v) or key.endswith("v2" + v) or key.endswith("v2_" + v):
            return v
    raise KeyError(f"not an EfficientFormerV2 name: {name}")


class ConvBN(nn.Module):
    """conv (with bias) + BatchNorm2d [+ GELU].  Children are named `conv` and `bn` like timm's ConvNorm / ConvNormAct."""

    def __init__(self, cin: int, cout: int, k: int = 1, stride: int = 1, groups: int = 1, act: bool = False) -> None:
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=((stride - 1) + (k - 1)) // 2, groups=groups, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=BN_EPS)
        self.act = act

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = F.conv2d(x, self.conv.weight, self.conv.bias, self.conv.stride, self.conv.padding, 1, self.conv.groups)
        x = F.batch_norm(x, self.bn.running_mean, self.bn.running_var, self.bn.weight, self.bn.bias, self.training,
                         self.bn.momentum, self.bn.eps)
        if self.training:
            self.bn.num_batches_tracked += 1
        return F.gelu(x) if self.act else x


class Scale(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.gamma = nn.Parameter(LS_INIT * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x * self.gamma.view(1, -1, 1, 1)


def bias_index(q_res: tuple[int, int], k_res: tuple[int, int], q_step: int) -> torch.Tensor:
    """idx[i, j] = |qy_i - ky_j| * k_width + |qx_i - kx_j| over the flattened query / key grids."""
    ky, kx = torch.meshgrid(torch.arange(k_res[0]), torch.arange(k_res[1]), indexing="ij")
    qy, qx = torch.meshgrid(torch.arange(0, k_res[0], q_step), torch.arange(0, k_res[1], q_step), indexing="ij")
    assert qy.shape == q_res
    dy = (qy.reshape(-1, 1) - ky.reshape(1, -1)).abs()
    dx = (qx.reshape(-1, 1) - kx.reshape(1, -1)).abs()
    return dy * k_res[1] + dx


class Attention(nn.Module):
    """Attention2d: 8 heads, key_dim 32, value dim 4*32 per head; optional stride-2 token reduction."""

    def __init__(self, dim: int, resolution: tuple[int, int], stride: int | None, heads: int = 8, key_dim: int = 32,
                 attn_ratio: int = 4) -> None:
        super().__init__()
        self.heads, self.key_dim, self.scale = heads, key_dim, key_dim ** -0.5
        if stride is not None:
            resolution = tuple(math.ceil(r / stride) for r in resolution)
            self.stride_conv = ConvBN(dim, dim, 3, stride, groups=dim)
        else:
            self.stride_conv = None
        self.stride = stride
        self.resolution = resolution
        self.N = resolution[0] * resolution[1]
        self.d = attn_ratio * key_dim
        self.dh = self.d * heads
        kh = key_dim * heads
        self.q = ConvBN(dim, kh)
        self.k = ConvBN(dim, kh)
        self.v = ConvBN(dim, self.dh)
        self.v_local = ConvBN(self.dh, self.dh, 3, groups=self.dh)
        self.talking_head1 = nn.Conv2d(heads, heads, 1)
        self.talking_head2 = nn.Conv2d(heads, heads, 1)
        self.proj = ConvBN(self.dh, dim)
        self.attention_biases = nn.Parameter(torch.zeros(heads, self.N))
        self.register_buffer("attention_bias_idxs", bias_index(resolution, resolution, 1), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        if self.stride_conv is not None:
            x = self.stride_conv(x)
        q = self.q(x).reshape(B, self.heads, self.key_dim, self.N).transpose(2, 3)        # [B, h, N, dk]
        k = self.k(x).reshape(B, self.heads, self.key_dim, self.N)                        # [B, h, dk, N]
        v = self.v(x)
        v_local = self.v_local(v)
        v = v.reshape(B, self.heads, self.d, self.N).transpose(2, 3)                      # [B, h, N, d]
        attn = (q @ k) * self.scale + self.attention_biases[:, self.attention_bias_idxs]
        attn = F.conv2d(attn, self.talking_head1.weight, self.talking_head1.bias)
        attn = attn.softmax(dim=-1)
        attn = F.conv2d(attn, self.talking_head2.weight, self.talking_head2.bias)
        out = (attn @ v).transpose(2, 3).reshape(B, self.dh, self.resolution[0], self.resolution[1]) + v_local
        if self.stride is not None:
            out = F.interpolate(out, scale_factor=self.stride, mode="bilinear")
        return self.proj(F.gelu(out))


class LocalGlobalQuery(nn.Module):
    def __init__(self, dim: int, out_dim: int) -> None:
        super().__init__()
        self.local = nn.Conv2d(dim, dim, 3, stride=2, padding=1, groups=dim)
        self.proj = ConvBN(dim, out_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # AvgPool2d(kernel 1, stride 2) == every other pixel
        return self.proj(F.conv2d(x, self.local.weight, self.local.bias, 2, 1, 1, x.shape[1]) + x[:, :, ::2, ::2])


class AttentionDownsample(nn.Module):
    """Attention2dDownsample: queries on the stride-2 grid, keys / values on the full grid; key_dim 16."""

    def __init__(self, dim: int, out_dim: int, resolution: tuple[int, int], heads: int = 8, key_dim: int = 16,
                 attn_ratio: int = 4) -> None:
        super().__init__()
        self.heads, self.key_dim, self.scale = heads, key_dim, key_dim ** -0.5
        self.resolution = resolution
        self.resolution2 = tuple(math.ceil(r / 2) for r in resolution)
        self.N, self.N2 = resolution[0] * resolution[1], self.resolution2[0] * self.resolution2[1]
        self.d = attn_ratio * key_dim
        self.dh = self.d * heads
        kh = key_dim * heads
        self.q = LocalGlobalQuery(dim, kh)
        self.k = ConvBN(dim, kh)
        self.v = ConvBN(dim, self.dh)
        self.v_local = ConvBN(self.dh, self.dh, 3, 2, groups=self.dh)
        self.proj = ConvBN(self.dh, out_dim)
        self.attention_biases = nn.Parameter(torch.zeros(heads, self.N))
        self.register_buffer("attention_bias_idxs", bias_index(self.resolution2, resolution, 2), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        q = self.q(x).reshape(B, self.heads, self.key_dim, self.N2).transpose(2, 3)
        k = self.k(x).reshape(B, self.heads, self.key_dim, self.N)
        v = self.v(x)
        v_local = self.v_local(v)
        v = v.reshape(B, self.heads, self.d, self.N).transpose(2, 3)
        attn = ((q @ k) * self.scale + self.attention_biases[:, self.attention_bias_idxs]).softmax(dim=-1)
        out = (attn @ v).transpose(2, 3).reshape(B, self.dh, self.resolution2[0], self.resolution2[1]) + v_local
        return self.proj(F.gelu(out))


class Downsample(nn.Module):
    def __init__(self, cin: int, cout: int, resolution: tuple[int, int], use_attn: bool) -> None:
        super().__init__()
        self.conv = ConvBN(cin, cout, 3, 2)
        self.attn = AttentionDownsample(cin, cout, resolution) if use_attn else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.conv(x)
        return self.attn(x) + out if self.attn is not None else out


class ConvMlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = ConvBN(dim, hidden, 1, act=True)
        self.mid = ConvBN(hidden, hidden, 3, groups=hidden, act=True)
        self.fc2 = ConvBN(hidden, dim, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(self.mid(self.fc1(x)))


def drop_path(x: torch.Tensor, p: float, training: bool, mask: torch.Tensor | None = None) -> torch.Tensor:
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    if mask is None:
        mask = x.new_empty(x.shape[0], 1, 1, 1).bernoulli_(keep) / keep
    return x * mask.view(-1, 1, 1, 1)


class Block(nn.Module):
    def __init__(self, dim: int, mlp_ratio: int, resolution: tuple[int, int], stride: int | None, use_attn: bool,
                 dp: float) -> None:
        super().__init__()
        if use_attn:
            self.token_mixer = Attention(dim, resolution, stride)
            self.ls1 = Scale(dim)
        else:
            self.token_mixer = None
            self.ls1 = None
        self.mlp = ConvMlp(dim, int(dim * mlp_ratio))
        self.ls2 = Scale(dim)
        self.dp = dp

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.token_mixer is not None:
            x = x + drop_path(self.ls1(self.token_mixer(x)), self.dp, self.training)
        return x + drop_path(self.ls2(self.mlp(x)), self.dp, self.training)


class Stage(nn.Module):
    def __init__(self, dim: int, dim_out: int, depth: int, resolution: tuple[int, int], downsample: bool,
                 block_stride: int | None, downsample_attn: bool, block_attn: bool, num_vit: int, ratios, dprs) -> None:
        super().__init__()
        if downsample:
            self.downsample = Downsample(dim, dim_out, resolution, downsample_attn)
            dim = dim_out
            resolution = tuple(math.ceil(r / 2) for r in resolution)
        else:
            assert dim == dim_out
            self.downsample = nn.Identity()
        first_attn = depth - num_vit            # blocks with index > depth - num_vit - 1
        self.blocks = nn.Sequential(*[
            Block(dim, ratios[i], resolution, block_stride, block_attn and i >= first_attn, dprs[i]) for i in range(depth)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.blocks(self.downsample(x))


class Stem(nn.Sequential):
    def __init__(self, cout: int) -> None:
        super().__init__()
        self.conv1 = ConvBN(3, cout // 2, 3, 2, act=True)
        self.conv2 = ConvBN(cout // 2, cout, 3, 2, act=True)


class EfficientFormerV2Ref(nn.Module):
    def __init__(self, variant: str = "s1", num_classes: int = 1000, img_size: int = 224, drop_rate: float = 0.0,
                 drop_path_rate: float | None = None) -> None:
        super().__init__()
        widths, depths, ratios = WIDTHS[variant], DEPTHS[variant], EXPANSION[variant]
        dpr_total = DROP_PATH[variant] if drop_path_rate is None else drop_path_rate
        self.variant, self.num_classes, self.img_size = variant, num_classes, img_size
        self.stem = Stem(widths[0])
        dprs = torch.linspace(0, dpr_total, sum(depths)).split(list(depths))
        stages, prev, stride = [], widths[0], 4
        for i in range(4):
            res = (math.ceil(img_size / stride),) * 2
            stages.append(Stage(prev, widths[i], depths[i], res, downsample=i > 0, block_stride=2 if i == 2 else None,
                                downsample_attn=i >= 3, block_attn=i >= 2, num_vit=NUM_VIT[variant], ratios=ratios[i],
                                dprs=[float(v) for v in dprs[i]]))
            if i > 0:
                stride *= 2
            prev = widths[i]
        self.stages = nn.Sequential(*stages)
        self.num_features = widths[-1]
        self.norm = nn.BatchNorm2d(widths[-1], eps=BN_EPS)
        self.drop_rate = drop_rate
        self.head = nn.Linear(widths[-1], num_classes)
        self.head_dist = nn.Linear(widths[-1], num_classes)
        for m in self.modules():                                   # timm: trunc_normal_(std=.02) on every Linear
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        x = self.stages(self.stem(x))
        x = F.batch_norm(x, self.norm.running_mean, self.norm.running_var, self.norm.weight, self.norm.bias, self.training,
                         self.norm.momentum, self.norm.eps)
        if self.training:
            self.norm.num_batches_tracked += 1
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.forward_features(x).mean(dim=(2, 3))
        x = F.dropout(x, self.drop_rate, self.training)
        return (F.linear(x, self.head.weight, self.head.bias) + F.linear(x, self.head_dist.weight, self.head_dist.bias)) / 2


def count_macs(model: EfficientFormerV2Ref, img_size: int = 224) -> int:
    """Multiply-accumulates of the convolutions, linears and attention products of one forward pass."""
    total = 0
    hooks = []
    model.eval()
    shapes = {}

    def rec(name):
        def hook(m, inp, out):
            shapes[name] = (tuple(inp[0].shape), tuple(out.shape))
        return hook

    for name, m in model.named_modules():
        if isinstance(m, (ConvBN, Attention, AttentionDownsample, LocalGlobalQuery)):
            hooks.append(m.register_forward_hook(rec(name)))
    with torch.no_grad():
        model(torch.zeros(1, 3, img_size, img_size))
    for h in hooks:
        h.remove()
    mods = dict(model.named_modules())
    for name, (i_shape, o_shape) in shapes.items():
        m = mods[name]
        if isinstance(m, ConvBN):
            c = m.conv
            total += o_shape[1] * o_shape[2] * o_shape[3] * (c.in_channels // c.groups) * c.kernel_size[0] * c.kernel_size[1]
        elif isinstance(m, Attention):
            total += m.heads * m.N * m.N * (m.key_dim + m.d) + 2 * m.heads * m.heads * m.N * m.N
        elif isinstance(m, AttentionDownsample):
            total += m.heads * m.N2 * m.N * (m.key_dim + m.d)
        elif isinstance(m, LocalGlobalQuery):
            total += o_shape[2] * o_shape[3] * m.local.in_channels * 9
    total += 2 * model.head.in_features * model.head.out_features
    return total


def train_step_ref(model: nn.Module, opt: torch.optim.Optimizer, x: torch.Tensor, y: torch.Tensor,
                   label_smoothing: float = 0.1) -> float:
    """The loop body of trainers/efficientformer_v2.py:237-249 in f32 (zero_grad, forward, CE, backward, step)."""
    model.train()
    opt.zero_grad(set_to_none=True)
    loss = F.cross_entropy(model(x), y, label_smoothing=label_smoothing)
    loss.backward()
    opt.step()
    return float(loss.detach())


__all__ = ["EfficientFormerV2Ref", "count_macs", "train_step_ref", "variant_of", "WIDTHS", "DEPTHS", "EXPANSION", "NUM_VIT"]
