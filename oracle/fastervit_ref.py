"""TEST INFRASTRUCTURE — CPU oracle for FasterViT (fastervit 1.0.0 `faster_vit_{0,1,2,3}_224`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The reference builds this model with `fastervit.create_model("faster_vit_2_224", pretrained=True)` and swaps
`model.head = nn.Linear(model.head.in_features, num_classes)` (trainers/fastervit.py:371-373;
orchestration/model_registry.py:43-47) and calls it at trainers/fastervit.py:271 (train), :235 (evaluate) and
orchestration/orchestrator.py:529,590 (inference).  The `fastervit` package is not installable in the build
container (ordinary ModuleNotFoundError; no network), so this file RESTATES the published architecture
(fastervit/models/faster_vit.py at the pinned version; Hatamizadeh et al., "FasterViT: Fast Vision Transformers
with Hierarchical Attention", ICLR 2024) with torch.nn.functional ops, under the package's parameter names.

PARITY UNPINNED against the package itself (absent; the reference's tests hold no numeric fixture).  Pinned
instead by tests/test_fastervit_oracle.py: the published parameter counts (FasterViT-0 31.4 M, -1 53.4 M,
-2 75.9 M, -3 159.5 M), the state-dict key grammar, the head width the trainer relies on (512 / 640 / 768), token
bookkeeping identities (window partition / reverse and carrier-token de-window / window are inverse
permutations) and the bias-table construction against a brute-force loop.

Architecture digest:
  patch_embed  conv3x3 s2 (3 -> in_dim) BN(eps 1e-4) ReLU, conv3x3 s2 (in_dim -> dim) BN(eps 1e-4) ReLU     -> 1/4
  level 0, 1   ConvBlock x depth: conv3x3 (+bias) BN GELU conv3x3 (+bias) BN [* gamma] + x (DropPath)
  downsample   LayerNorm2d(eps 1e-6) -> conv3x3 s2 (C -> 2C, no bias)                       (after levels 0, 1, 2)
  level 2      7x7 windows of a 14x14 map; 2x2 carrier tokens per window (16 in total), initialised by
               dw3x3 (+bias) -> AvgPool(5, stride 3); every HAT block: carrier tokens attend globally (16 tokens,
               attention + MLP), are appended to their window's 49 tokens (sequence 53) for the local attention
               + MLP, and are split off again
  level 3      one 7x7 window, plain attention + MLP
  attention    qkv Linear(+bias) -> per head softmax(q k^T / sqrt(32) + bias) v -> proj; the bias is
               16 * sigmoid(MLP(log-spaced relative coordinates)), zero for carrier-token rows / columns
  tail         BN -> global average pool -> head Linear
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn

CONFIGS = {
    #        depths          heads           dim in_dim drop_path layer_scale
    "0": ((2, 3, 6, 5), (2, 4, 8, 16), 64, 64, 0.2, None),
    "1": ((1, 3, 8, 5), (2, 4, 8, 16), 80, 32, 0.2, None),
    "2": ((3, 3, 8, 5), (2, 4, 8, 16), 96, 64, 0.2, None),
    "3": ((3, 3, 12, 5), (2, 4, 8, 16), 128, 64, 0.3, 1e-5),
}
WINDOW, CT_SIZE, MLP_RATIO = 7, 2, 4


def variant_of(name: str) -> str:
    key = name.lower().replace("-", "_")
    parts = key.split("_")
    if len(parts) >= 3 and parts[0] == "faster" and parts[1] == "vit" and parts[2] in CONFIGS:
        return parts[2]
    raise KeyError(f"not a FasterViT name handled here: {name}")


def window_partition(x: torch.Tensor, ws: int) -> torch.Tensor:
    B, C, H, W = x.shape
    x = x.view(B, C, H // ws, ws, W // ws, ws)
    return x.permute(0, 2, 4, 3, 5, 1).reshape(-1, ws * ws, C)


def window_reverse(windows: torch.Tensor, ws: int, H: int, W: int) -> torch.Tensor:
    B = windows.shape[0] // ((H // ws) * (W // ws))
    x = windows.reshape(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 5, 1, 3, 2, 4).reshape(B, windows.shape[2], H, W)


def ct_dewindow(ct: torch.Tensor, W: int, H: int, ws: int) -> torch.Tensor:
    """carrier tokens stored window by window -> row-major image order"""
    bs, N = ct.shape[0], ct.shape[2]
    ct2 = ct.view(-1, W // ws, H // ws, ws, ws, N).permute(0, 5, 1, 3, 2, 4)
    return ct2.reshape(bs, N, W * H).transpose(1, 2)


def ct_window(ct: torch.Tensor, W: int, H: int, ws: int) -> torch.Tensor:
    bs, N = ct.shape[0], ct.shape[2]
    ct = ct.view(bs, H // ws, ws, W // ws, ws, N)
    return ct.permute(0, 1, 3, 2, 4, 5)


class PosEmb1D(nn.Module):
    """PosEmbMLPSwinv1D (rank 2): adds MLP((y, x) / (s // 2) - 1) to every token of an s x s grid."""

    def __init__(self, dim: int, seq_length: int) -> None:
        super().__init__()
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(), nn.Linear(512, dim, bias=False))
        self.register_buffer("relative_bias", torch.zeros(1, seq_length, dim))

    def table(self, seq_length: int) -> torch.Tensor:
        s = int(seq_length ** 0.5)
        ar = torch.arange(0, s, dtype=torch.float32, device=self.cpb_mlp[0].weight.device)
        grid = torch.stack(torch.meshgrid([ar, ar], indexing="ij")).unsqueeze(0)      # [1, 2, s, s]
        grid = (grid - s // 2) / (s // 2)
        return self.cpb_mlp(grid.flatten(2).transpose(1, 2))                         # [1, s*s, dim]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.table(x.shape[1])


class PosEmb2D(nn.Module):
    """PosEmbMLPSwinv2D: additive attention bias 16*sigmoid(MLP(log-spaced relative coordinates)), zero-padded on
    the top / left for carrier tokens."""

    def __init__(self, ws: int, heads: int, seq_length: int) -> None:
        super().__init__()
        self.ws, self.heads = ws, heads
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, heads, bias=False))
        rel = torch.arange(-(ws - 1), ws, dtype=torch.float32)
        table = torch.stack(torch.meshgrid([rel, rel], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
        table = table / (ws - 1) * 8
        table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / math.log2(8)
        self.register_buffer("relative_coords_table", table)
        ar = torch.arange(ws)
        coords = torch.flatten(torch.stack(torch.meshgrid([ar, ar], indexing="ij")), 1)
        rc = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rc[:, :, 0] += ws - 1
        rc[:, :, 1] += ws - 1
        rc[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rc.sum(-1))
        self.register_buffer("relative_bias", torch.zeros(1, heads, seq_length, seq_length))

    def bias(self, n_tokens: int) -> torch.Tensor:
        n_local = self.ws * self.ws
        tab = self.cpb_mlp(self.relative_coords_table).view(-1, self.heads)
        b = tab[self.relative_position_index.view(-1)].view(n_local, n_local, -1).permute(2, 0, 1).contiguous()
        b = 16 * torch.sigmoid(b)
        n_global = n_tokens - n_local
        return F.pad(b, (n_global, 0, n_global, 0)).unsqueeze(0)

    def forward(self, attn: torch.Tensor, local_window_size: int) -> torch.Tensor:
        return attn + self.bias(attn.shape[2])


class WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, resolution: int, seq_length: int) -> None:
        super().__init__()
        self.heads, self.scale, self.resolution = heads, (dim // heads) ** -0.5, resolution
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.pos_emb_funct = PosEmb2D(resolution, heads, seq_length)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, -1, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.pos_emb_funct(attn, self.resolution ** 2).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, -1, C))


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(F.gelu(self.fc1(x)))


def drop_path(x: torch.Tensor, p: float, training: bool, mask: torch.Tensor | None = None) -> torch.Tensor:
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    if mask is None:
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep) / keep
    return x * mask.view((x.shape[0],) + (1,) * (x.dim() - 1))


class HAT(nn.Module):
    def __init__(self, dim: int, heads: int, sr_ratio: int, dp: float, layer_scale: float | None) -> None:
        super().__init__()
        self.pos_embed = PosEmb1D(dim, WINDOW ** 2)
        self.norm1 = nn.LayerNorm(dim)
        per_window = CT_SIZE ** 2 if sr_ratio > 1 else 0
        total = per_window * sr_ratio * sr_ratio
        self.cr_window, self.sr_ratio, self.dp = CT_SIZE, sr_ratio, dp
        self.attn = WindowAttention(dim, heads, WINDOW, WINDOW ** 2 + per_window)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * MLP_RATIO))
        use_ls = layer_scale is not None
        self.gamma3 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
        self.gamma4 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
        if sr_ratio > 1:
            self.hat_norm1 = nn.LayerNorm(dim)
            self.hat_norm2 = nn.LayerNorm(dim)
            self.hat_attn = WindowAttention(dim, heads, int(total ** 0.5), total)
            self.hat_mlp = Mlp(dim, int(dim * MLP_RATIO))
            self.hat_pos_embed = PosEmb1D(dim, total)
            self.gamma1 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
            self.gamma2 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1

    def forward(self, x: torch.Tensor, ct: torch.Tensor | None, masks=None):
        """masks: optional (m_ct, m_win) per-sample DropPath masks for the carrier / window streams (already 1/keep scaled)."""
        B, T, N = x.shape
        m_ct, m_win = masks if masks is not None else (None, None)
        x = self.pos_embed(x)
        if self.sr_ratio > 1:
            Bg, Ng, Hg = ct.shape
            side = self.cr_window * self.sr_ratio
            ct = ct_dewindow(ct, side, side, self.cr_window)
            ct = self.hat_pos_embed(ct)
            ct = ct + drop_path(self.gamma1 * self.hat_attn(self.hat_norm1(ct)), self.dp, self.training, m_ct)
            ct = ct + drop_path(self.gamma2 * self.hat_mlp(self.hat_norm2(ct)), self.dp, self.training, m_ct)
            ct = ct_window(ct, side, side, self.cr_window).reshape(x.shape[0], -1, N)
            x = torch.cat((ct, x), dim=1)
        x = x + drop_path(self.gamma3 * self.attn(self.norm1(x)), self.dp, self.training, m_win)
        x = x + drop_path(self.gamma4 * self.mlp(self.norm2(x)), self.dp, self.training, m_win)
        if self.sr_ratio > 1:
            ctr, x = x.split([x.shape[1] - WINDOW * WINDOW, WINDOW * WINDOW], dim=1)
            ct = ctr.reshape(Bg, Ng, Hg)
        return x, ct


class ConvBlock(nn.Module):
    def __init__(self, dim: int, dp: float, layer_scale: float | None) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, 3, 1, 1)
        self.norm1 = nn.BatchNorm2d(dim, eps=1e-5)
        self.conv2 = nn.Conv2d(dim, dim, 3, 1, 1)
        self.norm2 = nn.BatchNorm2d(dim, eps=1e-5)
        self.layer_scale = layer_scale is not None
        if self.layer_scale:
            self.gamma = nn.Parameter(layer_scale * torch.ones(dim))
        self.dp = dp

    def forward(self, x: torch.Tensor, mask: torch.Tensor | None = None) -> torch.Tensor:
        h = _bn(F.conv2d(x, self.conv1.weight, self.conv1.bias, 1, 1), self.norm1, self.training)
        h = _bn(F.conv2d(F.gelu(h), self.conv2.weight, self.conv2.bias, 1, 1), self.norm2, self.training)
        if self.layer_scale:
            h = h * self.gamma.view(1, -1, 1, 1)
        return x + drop_path(h, self.dp, self.training, mask)


def _bn(x: torch.Tensor, bn: nn.BatchNorm2d, training: bool) -> torch.Tensor:
    out = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
    if training:
        bn.num_batches_tracked += 1
    return out


class LayerNorm2d(nn.LayerNorm):
    """timm's LayerNorm2d: LayerNorm over the channel dimension of an NCHW tensor, eps 1e-6."""

    def __init__(self, dim: int) -> None:
        super().__init__(dim, eps=1e-6)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class Downsample(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.norm = LayerNorm2d(dim)
        self.reduction = nn.Sequential(nn.Conv2d(dim, 2 * dim, 3, 2, 1, bias=False))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.reduction(self.norm(x))


class TokenInitializer(nn.Module):
    def __init__(self, dim: int, input_resolution: int) -> None:
        super().__init__()
        output_size = int(CT_SIZE * input_resolution / WINDOW)
        stride = int(input_resolution / output_size)
        kernel = input_resolution - (output_size - 1) * stride
        self.pos_embed = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        to_global = nn.Sequential()
        to_global.add_module("pos", self.pos_embed)                         # the package registers the conv twice
        to_global.add_module("pool", nn.AvgPool2d(kernel_size=kernel, stride=stride))
        self.to_global_feature = to_global
        self.kernel, self.stride = kernel, stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.to_global_feature(x)
        B, C, H, W = x.shape
        ct = x.view(B, C, H // CT_SIZE, CT_SIZE, W // CT_SIZE, CT_SIZE)
        return ct.permute(0, 2, 4, 3, 5, 1).reshape(-1, H * W, C)


class Level(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, conv: bool, downsample: bool, dprs, input_resolution: int,
                 only_local: bool, layer_scale: float | None) -> None:
        super().__init__()
        self.conv = conv
        if conv:
            self.blocks = nn.ModuleList([ConvBlock(dim, dprs[i], None) for i in range(depth)])
        else:
            sr = input_resolution // WINDOW if not only_local else 1
            self.blocks = nn.ModuleList([HAT(dim, heads, sr, dprs[i], layer_scale) for i in range(depth)])
        self.downsample = Downsample(dim) if downsample else None
        self.do_gt = (not conv) and (not only_local) and input_resolution // WINDOW > 1
        if self.do_gt:
            self.global_tokenizer = TokenInitializer(dim, input_resolution)

    def forward(self, x: torch.Tensor, masks=None) -> torch.Tensor:
        ct = self.global_tokenizer(x) if self.do_gt else None
        B, C, H, W = x.shape
        if not self.conv:
            x = window_partition(x, WINDOW)
        for i, blk in enumerate(self.blocks):
            m = None if masks is None else masks[i]
            if self.conv:
                x = blk(x, m)
            else:
                x, ct = blk(x, ct, m)
        if not self.conv:
            x = window_reverse(x, WINDOW, H, W)
        return x if self.downsample is None else self.downsample(x)


class PatchEmbed(nn.Module):
    def __init__(self, in_dim: int, dim: int) -> None:
        super().__init__()
        self.proj = nn.Identity()
        self.conv_down = nn.Sequential(
            nn.Conv2d(3, in_dim, 3, 2, 1, bias=False), nn.BatchNorm2d(in_dim, eps=1e-4), nn.ReLU(),
            nn.Conv2d(in_dim, dim, 3, 2, 1, bias=False), nn.BatchNorm2d(dim, eps=1e-4), nn.ReLU())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        c = self.conv_down
        x = F.relu(_bn(F.conv2d(x, c[0].weight, None, 2, 1), c[1], self.training))
        return F.relu(_bn(F.conv2d(x, c[3].weight, None, 2, 1), c[4], self.training))


class FasterViTRef(nn.Module):
    def __init__(self, variant: str = "0", num_classes: int = 1000, resolution: int = 224, drop_path_rate: float | None = None) -> None:
        super().__init__()
        depths, heads, dim, in_dim, dpr, layer_scale = CONFIGS[variant]
        dpr = dpr if drop_path_rate is None else drop_path_rate
        self.variant, self.num_classes, self.resolution = variant, num_classes, resolution
        self.patch_embed = PatchEmbed(in_dim, dim)
        rates = [float(v) for v in torch.linspace(0, dpr, sum(depths))]
        hat = (False, False, True, False)
        self.levels = nn.ModuleList()
        for i in range(4):
            self.levels.append(Level(dim * 2 ** i, depths[i], heads[i], conv=i < 2, downsample=i < 3,
                                     dprs=rates[sum(depths[:i]):sum(depths[:i + 1])], input_resolution=int(2 ** (-2 - i) * resolution),
                                     only_local=not hat[i], layer_scale=layer_scale))
        self.num_features = dim * 8
        self.norm = nn.BatchNorm2d(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(self.num_features, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x: torch.Tensor, masks=None) -> torch.Tensor:
        x = self.patch_embed(x)
        for i, level in enumerate(self.levels):
            x = level(x, None if masks is None else masks[i])
        return _bn(x, self.norm, self.training)

    def forward(self, x: torch.Tensor, masks=None) -> torch.Tensor:
        """masks: optional per-level, per-block DropPath masks (None disables the stochastic depth entirely when the
        model is built with drop_path_rate 0; with a rate > 0 and masks=None torch's RNG draws them)."""
        return self.head(self.forward_features(x, masks).mean((2, 3)))


def train_step_ref(model: nn.Module, opt: torch.optim.Optimizer, x: torch.Tensor, y: torch.Tensor, label_smoothing: float = 0.1) -> float:
    model.train()
    opt.zero_grad(set_to_none=True)
    loss = F.cross_entropy(model(x), y, label_smoothing=label_smoothing)
    loss.backward()
    opt.step()
    return float(loss.detach())


__all__ = ["CONFIGS", "FasterViTRef", "train_step_ref", "variant_of", "window_partition", "window_reverse", "ct_dewindow", "ct_window"]
