"""HIP-backed FasterViT: drop-in for `fastervit.create_model("faster_vit_{0,1,2,3}_224")`.

The reference builds `faster_vit_2_224` and replaces its head (trainers/fastervit.py:371-373:
`model.head = nn.Linear(model.head.in_features, num_classes)`; orchestration/model_registry.py:43-47), trains names
containing "head" during the warm-up (:400-402) and everything afterwards (:434-435).  This module carries the
package's parameter and buffer names (patch_embed.conv_down.0.weight, levels.2.blocks.0.hat_attn.pos_emb_funct.
cpb_mlp.0.weight, levels.2.global_tokenizer.to_global_feature.pos.weight, ...), real nn.Conv2d / nn.Linear /
nn.LayerNorm / nn.BatchNorm2d objects as parameter containers, arithmetic in fastervit_functions.py on the kernels
of libdfd_hip.so.  No ATen fallback: a CPU input raises.  Architecture: fastervit 1.0.0 faster_vit.py, restated
independently in oracle/fastervit_ref.py, which this module is tested against.
"""

from __future__ import annotations

import math

import torch
from torch import nn

from ._lib import ACT_RELU
from .efficientnet import compute_dtype
from .fastervit_functions import (AttnSpec, ConvBlockCtx, ConvBlockFunction, FVDownsampleFunction, HATCtx, HATFunction,
                                  TokenInitFunction, derived_weights)
from .functions import BNRef, bn_eval_batch
from .vit_functions import ConvStemCtx, ConvStemFunction, DenseConvBNFunction, DenseConvCtx, TailCtx, TailFunction

_CONFIGS = {
    "0": ((2, 3, 6, 5), (2, 4, 8, 16), 64, 64, 0.2, None),
    "1": ((1, 3, 8, 5), (2, 4, 8, 16), 80, 32, 0.2, None),
    "2": ((3, 3, 8, 5), (2, 4, 8, 16), 96, 64, 0.2, None),
    "3": ((3, 3, 12, 5), (2, 4, 8, 16), 128, 64, 0.3, 1e-5),
}
WINDOW, CT_SIZE, MLP_RATIO = 7, 2, 4


def _bnref(bn: nn.BatchNorm2d) -> BNRef:
    return BNRef(bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps)


class PosEmbMLPSwinv1D(nn.Module):
    def __init__(self, dim: int, seq_length: int) -> None:
        super().__init__()
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(), nn.Linear(512, dim, bias=False))
        self.register_buffer("relative_bias", torch.zeros(1, seq_length, dim))
        s = int(seq_length ** 0.5)
        ar = torch.arange(0, s, dtype=torch.float32)
        grid = torch.stack(torch.meshgrid([ar, ar], indexing="ij"))
        grid = (grid - s // 2) / (s // 2)
        self.register_buffer("_coords", grid.flatten(1).t().contiguous(), persistent=False)          # [s*s, 2]

    def tensors(self):
        return self.cpb_mlp[0].weight, self.cpb_mlp[0].bias, self.cpb_mlp[2].weight


class PosEmbMLPSwinv2D(nn.Module):
    def __init__(self, ws: int, heads: int, seq_length: int) -> None:
        super().__init__()
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
        index = rc.sum(-1)
        self.register_buffer("relative_position_index", index)
        self.register_buffer("relative_bias", torch.zeros(1, heads, seq_length, seq_length))
        self.register_buffer("_coords2d", table.reshape(-1, 2).contiguous(), persistent=False)
        self.register_buffer("_idx32", index.reshape(-1).to(torch.int32), persistent=False)

    def tensors(self):
        return self.cpb_mlp[0].weight, self.cpb_mlp[0].bias, self.cpb_mlp[2].weight


class WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, resolution: int, seq_length: int) -> None:
        super().__init__()
        self.heads, self.resolution, self.seq_length = heads, resolution, seq_length
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.pos_emb_funct = PosEmbMLPSwinv2D(resolution, heads, seq_length)

    def spec(self) -> AttnSpec:
        n_local = self.resolution ** 2
        return AttnSpec(self.heads, n_local, self.seq_length - n_local, self.pos_emb_funct._coords2d, self.pos_emb_funct._idx32)

    def tensors(self, norm: nn.LayerNorm):
        """The attention sub-block's own tensors; the relative-position bias (pos_emb_funct) comes from coord_tables."""
        return (norm.weight, norm.bias, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def tensors(self, norm: nn.LayerNorm):
        return norm.weight, norm.bias, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias


class HipHAT(nn.Module):
    def __init__(self, dim: int, heads: int, sr_ratio: int, drop_path: float, layer_scale: float | None, index: int) -> None:
        super().__init__()
        self.pos_embed = PosEmbMLPSwinv1D(dim, WINDOW ** 2)
        self.norm1 = nn.LayerNorm(dim)
        per_window = CT_SIZE ** 2 if sr_ratio > 1 else 0
        total = per_window * sr_ratio * sr_ratio
        self.sr_ratio, self.drop_path, self.index, self.per_window = sr_ratio, drop_path, index, per_window
        self.attn = WindowAttention(dim, heads, WINDOW, WINDOW ** 2 + per_window)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * MLP_RATIO))
        use_ls = layer_scale is not None
        self.gamma3 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else None
        self.gamma4 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else None
        if sr_ratio > 1:
            self.hat_norm1 = nn.LayerNorm(dim)
            self.hat_norm2 = nn.LayerNorm(dim)
            self.hat_attn = WindowAttention(dim, heads, int(total ** 0.5), total)
            self.hat_mlp = Mlp(dim, int(dim * MLP_RATIO))
            self.hat_pos_embed = PosEmbMLPSwinv1D(dim, total)
            self.gamma1 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else None
            self.gamma2 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else None

    def coord_jobs(self):
        """(jobs, params): this block's coordinate MLPs for fastervit_functions.coord_tables — token position table, window
        attention bias [, carrier position table, carrier attention bias]."""
        from .fastervit_functions import CoordJob

        def cpb(att: WindowAttention) -> CoordJob:
            sp = att.spec()
            return CoordJob("cpb", sp.coords2d, sp.idx, sp.n_local, sp.n_global)

        jobs = [CoordJob("pos", self.pos_embed._coords), cpb(self.attn)]
        params = [*self.pos_embed.tensors(), *self.attn.pos_emb_funct.tensors()]
        if self.sr_ratio > 1:
            jobs += [CoordJob("pos", self.hat_pos_embed._coords), cpb(self.hat_attn)]
            params += [*self.hat_pos_embed.tensors(), *self.hat_attn.pos_emb_funct.tensors()]
        return jobs, params

    def forward(self, x, ct, maps, rng=None, tables=None):
        """tables: this block's outputs of coord_tables (the level computes them for all of its blocks in one call); None:
        computed here (a block used on its own)."""
        from .fastervit_functions import coord_tables

        carrier = self.sr_ratio > 1
        if tables is None:
            tables = coord_tables(*self.coord_jobs())
        rs_win = rs_ct = None
        if self.training and self.drop_path > 0.0 and rng is not None:
            keep = 1.0 - self.drop_path
            rs_win = rng.drop_path_scale(x.shape[0], keep, stream_id=4 * self.index)
            if carrier:
                rs_ct = rng.drop_path_scale(ct.shape[0], keep, stream_id=4 * self.index + 1)
        flat = [tables[0], *self.attn.tensors(self.norm1), tables[1], *self.mlp.tensors(self.norm2)]
        if carrier:
            flat += [tables[2], *self.hat_attn.tensors(self.hat_norm1), tables[3], *self.hat_mlp.tensors(self.hat_norm2)]
        flat += [self.gamma3, self.gamma4]
        if carrier:
            flat += [self.gamma1, self.gamma2]
        flat += [rs_win, rs_ct]
        cfg = HATCtx(self.attn.heads, carrier, self.attn.spec(), self.hat_attn.spec() if carrier else None, self.pos_embed._coords,
                     self.hat_pos_embed._coords if carrier else None, *(maps if carrier else (None, None, None)), self.training)
        return HATFunction.apply(x, ct, cfg, *flat)


class HipConvBlock(nn.Module):
    def __init__(self, dim: int, drop_path: float, layer_scale: float | None, index: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, 3, 1, 1)
        self.norm1 = nn.BatchNorm2d(dim, eps=1e-5)
        self.conv2 = nn.Conv2d(dim, dim, 3, 1, 1)
        self.norm2 = nn.BatchNorm2d(dim, eps=1e-5)
        self.gamma = nn.Parameter(layer_scale * torch.ones(dim)) if layer_scale is not None else None
        self.drop_path, self.index = drop_path, index

    def forward(self, x, rng=None, counters=None):
        rs = None
        if self.training and self.drop_path > 0.0 and rng is not None:
            rs = rng.drop_path_scale(x.shape[0], 1.0 - self.drop_path, stream_id=4 * self.index)
        cfg = ConvBlockCtx(_bnref(self.norm1), _bnref(self.norm2), self.training, counters)
        return ConvBlockFunction.apply(x, self.conv1.weight, self.conv1.bias, self.norm1.weight, self.norm1.bias, self.conv2.weight,
                                       self.conv2.bias, self.norm2.weight, self.norm2.bias, self.gamma, rs, cfg)


class LayerNorm2d(nn.LayerNorm):
    def __init__(self, dim: int) -> None:
        super().__init__(dim, eps=1e-6)


class HipFVDownsample(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.norm = LayerNorm2d(dim)
        self.reduction = nn.Sequential(nn.Conv2d(dim, 2 * dim, 3, 2, 1, bias=False))

    def forward(self, x):
        return FVDownsampleFunction.apply(x, self.norm.weight, self.norm.bias, self.reduction[0].weight)


class HipTokenInitializer(nn.Module):
    def __init__(self, dim: int, input_resolution: int) -> None:
        super().__init__()
        output_size = int(CT_SIZE * input_resolution / WINDOW)
        self.stride = int(input_resolution / output_size)
        self.kernel = input_resolution - (output_size - 1) * self.stride
        self.pos_embed = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        to_global = nn.Sequential()
        to_global.add_module("pos", self.pos_embed)                 # registered twice, as in the package
        to_global.add_module("pool", nn.AvgPool2d(kernel_size=self.kernel, stride=self.stride))
        self.to_global_feature = to_global

    def forward(self, x):
        from .hooks import has_hooks

        if has_hooks(self.pos_embed):
            return self._hooked_forward(x)
        return TokenInitFunction.apply(x, self.pos_embed.weight, self.pos_embed.bias, self.kernel, self.stride)

    def _hooked_forward(self, x):
        """Forward hooks on `pos_embed` (Grad-CAM's "last nn.Conv2d" of FasterViT, web_ui.py:95-114): the depthwise
        convolution runs as its own stage so that its output (+ bias, NCHW) is an autograd tensor wired to the logits
        through the average pool; eval mode only."""
        from .hooks import AvgPoolFunction, DwConvOutFunction, call_hooks

        if self.training:
            raise NotImplementedError("forward hooks on global_tokenizer.pos_embed are supported in eval mode only")
        xd = x.detach()
        y = DwConvOutFunction.apply(xd, self.pos_embed.weight, self.pos_embed.bias)
        y = call_hooks(self.pos_embed, xd, y)
        p = AvgPoolFunction.apply(y, self.kernel, self.stride)
        return p.reshape(p.shape[0], p.shape[1] * p.shape[2], 1, p.shape[3])


def _window_maps(B: int, res: int, device: torch.device):
    """int32 index maps for a batch of B res x res maps cut into 7x7 windows with 2x2 carrier tokens per window:
    (partition: source row of every window-major token; cat_src_ct, cat_dst_ct, cat_dst_x: see HATCtx)."""
    nwy = res // WINDOW
    b, wy, wx, iy, ix = torch.meshgrid(torch.arange(B), torch.arange(nwy), torch.arange(nwy), torch.arange(WINDOW), torch.arange(WINDOW),
                                       indexing="ij")
    part = (b * res * res + (wy * WINDOW + iy) * res + wx * WINDOW + ix).reshape(-1)
    nW = B * nwy * nwy
    side = nwy * CT_SIZE
    b, wy, wx, cy, cx = torch.meshgrid(torch.arange(B), torch.arange(nwy), torch.arange(nwy), torch.arange(CT_SIZE), torch.arange(CT_SIZE),
                                       indexing="ij")
    src_ct = (b * side * side + (wy * CT_SIZE + cy) * side + wx * CT_SIZE + cx).reshape(-1)
    per = CT_SIZE * CT_SIZE
    seq = WINDOW * WINDOW + per
    w = torch.arange(nW)
    dst_ct = (w[:, None] * seq + torch.arange(per)[None, :]).reshape(-1)
    dst_x = (w[:, None] * seq + per + torch.arange(WINDOW * WINDOW)[None, :]).reshape(-1)
    with torch.inference_mode(False):
        return tuple(t.to(torch.int32).to(device) for t in (part, src_ct, dst_ct, dst_x))


class HipFasterViTLayer(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, conv: bool, downsample: bool, dprs, input_resolution: int, only_local: bool,
                 layer_scale: float | None, first_index: int) -> None:
        super().__init__()
        self.conv, self.res = conv, input_resolution
        if conv:
            self.blocks = nn.ModuleList([HipConvBlock(dim, dprs[i], None, first_index + i) for i in range(depth)])
        else:
            sr = input_resolution // WINDOW if not only_local else 1
            self.blocks = nn.ModuleList([HipHAT(dim, heads, sr, dprs[i], layer_scale, first_index + i) for i in range(depth)])
        self.downsample = HipFVDownsample(dim) if downsample else None
        self.do_gt = (not conv) and (not only_local) and input_resolution // WINDOW > 1
        if self.do_gt:
            self.global_tokenizer = HipTokenInitializer(dim, input_resolution)

    def _maps(self, B: int, device: torch.device):
        cache = self.__dict__.setdefault("_map_cache", {})
        key = (B, device.type, device.index)
        if key not in cache:
            cache[key] = _window_maps(B, self.res, device)
        return cache[key]

    def forward(self, x, rng=None, counters=None):
        from . import kernels as K

        if self.conv:
            for blk in self.blocks:
                x = blk(x, rng, counters)
        else:
            B, H, W, C = x.shape
            ct = self.global_tokenizer(x) if self.do_gt else None
            nwin = (H // WINDOW) * (W // WINDOW)
            if nwin > 1:
                part, src_ct, dst_ct, dst_x = self._maps(B, x.device)
                xw = _PermuteRows.apply(x.reshape(B * H * W, C), part, True).view(B * nwin, WINDOW * WINDOW, 1, C)
                maps = (src_ct, dst_ct, dst_x)
            else:
                xw, maps = x.reshape(B, H * W, 1, C), None
            # every coordinate MLP of the level (position tables, attention biases) in one batched call: they depend on
            # parameters only (fastervit_functions.CoordTablesFunction)
            from .fastervit_functions import coord_tables

            jobs, params, spans = [], [], []
            for blk in self.blocks:
                j, p = blk.coord_jobs()
                spans.append((len(jobs), len(jobs) + len(j)))
                jobs += j
                params += p
            tabs = coord_tables(jobs, params)
            for blk, (lo, hi) in zip(self.blocks, spans):
                xw, ct = blk(xw, ct, maps, rng, tabs[lo:hi])
            if nwin > 1:
                x = _PermuteRows.apply(xw.reshape(B * H * W, C), part, False).view(B, H, W, C)
            else:
                x = xw.reshape(B, H, W, C)
        return x if self.downsample is None else self.downsample(x)


class _PermuteRows(torch.autograd.Function):
    """window_partition (gather=True: out[r] = x[idx[r]]) / window_reverse (gather=False: out[idx[r]] = x[r])."""

    @staticmethod
    def forward(ctx, x, idx, gather: bool):
        from . import kernels as K

        out = torch.empty_like(x)
        if gather:
            K.copy_rows(x, idx, out, None, x.shape[0])
        else:
            K.copy_rows(x, None, out, idx, x.shape[0])
        ctx.idx, ctx.gather = idx, gather
        return out

    @staticmethod
    def backward(ctx, g):
        from . import kernels as K

        g = g if g.is_contiguous() else g.contiguous()
        dx = torch.empty_like(g)
        if ctx.gather:
            K.copy_rows(g, None, dx, ctx.idx, g.shape[0])
        else:
            K.copy_rows(g, ctx.idx, dx, None, g.shape[0])
        return dx, None, None


class HipPatchEmbed(nn.Module):
    def __init__(self, in_dim: int, dim: int) -> None:
        super().__init__()
        self.proj = nn.Identity()
        self.conv_down = nn.Sequential(
            nn.Conv2d(3, in_dim, 3, 2, 1, bias=False), nn.BatchNorm2d(in_dim, eps=1e-4), nn.ReLU(),
            nn.Conv2d(in_dim, dim, 3, 2, 1, bias=False), nn.BatchNorm2d(dim, eps=1e-4), nn.ReLU())


class HipFasterViT(nn.Module):
    """FasterViT-{0,1,2,3} at 224 px (7x7 windows) whose forward/backward run on the MI355X kernels."""

    def __init__(self, variant: str = "0", num_classes: int = 1000, resolution: int = 224, drop_path_rate: float | None = None,
                 fp8_weights: bool = False) -> None:
        super().__init__()
        # BASELINE config 5 ("bf16/fp8 weights"): the qkv / proj / fc1 / fc2 weights whose K is a multiple of 128 are held as
        # OCP MX fp8 (e4m3fn + one e8m0 scale per 32) next to the f32 masters and the forward products run on the block-scaled
        # fp8 MFMA (csrc/dfd_mx.hip); off by default: bf16 weights
        self.fp8_weights = bool(fp8_weights)
        if variant not in _CONFIGS:
            raise KeyError(f"unknown FasterViT variant '{variant}'")
        depths, heads, dim, in_dim, dpr, layer_scale = _CONFIGS[variant]
        dpr = dpr if drop_path_rate is None else drop_path_rate
        if resolution % 32 or (resolution // 16) % WINDOW or (resolution // 32) % WINDOW:
            raise ValueError("FasterViT needs a resolution whose 1/16 and 1/32 maps tile into 7x7 windows (224)")
        self.variant, self.num_classes, self.resolution = variant, num_classes, resolution
        self.patch_embed = HipPatchEmbed(in_dim, dim)
        rates = [float(v) for v in torch.linspace(0, dpr, sum(depths))]
        hat = (False, False, True, False)
        self.levels = nn.ModuleList()
        index = 0
        for i in range(4):
            self.levels.append(HipFasterViTLayer(dim * 2 ** i, depths[i], heads[i], i < 2, i < 3, rates[sum(depths[:i]):sum(depths[:i + 1])],
                                                 int(2 ** (-2 - i) * resolution), not hat[i], layer_scale, index))
            index += depths[i]
        self.num_features = dim * 8
        self.norm = nn.BatchNorm2d(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(self.num_features, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def to(self, *args, **kwargs):
        kwargs.pop("memory_format", None)
        args = tuple(a for a in args if not isinstance(a, torch.memory_format))
        return super().to(*args, **kwargs) if (args or kwargs) else self

    def rng(self, device: torch.device):
        from . import kernels as K

        cur = self.__dict__.get("_rng_obj")
        if cur is None or cur.state.device != device:
            cur = self.__dict__["_rng_obj"] = K.DeviceRng(device)
        return cur

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("HipFasterViT runs on a HIP device only (no CPU fallback); move the input with .to('cuda')")
        # the coefficient blocks of every eval-mode BatchNorm and of every Linear's identity statistics (bias, LayerScale):
        # one batched launch per pass instead of ~70 small ones
        with bn_eval_batch(self.__dict__, (self.training, compute_dtype())), derived_weights(self._derived_weights(compute_dtype())):
            return self._forward(x)

    def _derived_weights(self, dt: torch.dtype) -> dict:
        """{weight.data_ptr(): (w_nk, w_kn)} for the qkv / proj / fc1 / fc2 Linear of every attention block, refreshed by ONE
        batched launch per forward pass (kernels.DerivedWeights); one cache entry per activation dtype, never freed (a
        captured hipGraph holds raw pointers into it).  With `fp8_weights` the entries of the weights with K % 128 == 0 are
        (kernels.MxWeight, dequantised [K][N] copy) from one batched quantisation launch (kernels.MxWeights) instead."""
        from . import kernels as K

        weights = [lin.weight for m in self.modules() if isinstance(m, (WindowAttention, Mlp))
                   for lin in ((m.qkv, m.proj) if isinstance(m, WindowAttention) else (m.fc1, m.fc2))]
        fp8 = [w for w in weights if self.fp8_weights and dt == torch.bfloat16 and w.shape[1] % 128 == 0]
        plain = [w for w in weights if not any(w is q for q in fp8)]
        caches = self.__dict__.setdefault("_derived_caches", {})
        table: dict = {}
        if plain:
            cache = caches.get(dt)
            if cache is None or not cache.valid_for(plain, dt):
                with torch.inference_mode(False):
                    cache = caches[dt] = K.DerivedWeights([(w, True, True, False) for w in plain], dt)
            cache.refresh()
            table.update({w.data_ptr(): pair for w, pair in zip(plain, cache.out)})
        if fp8:
            cache = caches.get(("mx", dt))
            if cache is None or not cache.valid_for(fp8, dt):
                with torch.inference_mode(False):
                    cache = caches[("mx", dt)] = K.MxWeights(fp8, dt)
            cache.refresh()
            table.update({w.data_ptr(): pair for w, pair in zip(fp8, cache.out)})
        return table

    def dp_cut_modules(self) -> list[nn.Module]:
        """Where a replayed data-parallel backward may be cut into segments (graph_step.plan_cuts): the four levels (one tensor in,
        one tensor out; the blocks inside levels 2-3 carry a (windows, carrier tokens) pair and are not cut points)."""
        return list(self.levels)

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[2] != self.resolution or x.shape[3] != self.resolution:
            raise ValueError(f"FasterViT was built for {self.resolution}x{self.resolution} inputs, got {tuple(x.shape[2:])}")
        dt = compute_dtype()
        tr = self.training
        counters: list = []
        xh = x.detach().float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        c = self.patch_embed.conv_down
        h = ConvStemFunction.apply(xh, c[0].weight, None, c[1].weight, c[1].bias, ConvStemCtx(2, 1, _bnref(c[1]), dt, tr, ACT_RELU, counters))
        h = DenseConvBNFunction.apply(h, c[3].weight, None, c[4].weight, c[4].bias, DenseConvCtx(3, 2, _bnref(c[4]), tr, ACT_RELU, counters))
        rng = self.rng(x.device) if tr else None
        for level in self.levels:
            h = level(h, rng, counters)
        cfg = TailCtx(_bnref(self.norm), 0.0, tr, counters)
        out = TailFunction.apply(h, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, None, None, None, cfg)
        if tr:
            self.rng(h.device).tick(counters)
        return out


def variant_of(name: str) -> str:
    parts = name.lower().replace("-", "_").split("_")
    if len(parts) >= 3 and parts[0] == "faster" and parts[1] == "vit" and parts[2] in _CONFIGS:
        return parts[2]
    raise KeyError(f"not a FasterViT name handled by the HIP engine: {name}")


def build_fastervit(name: str, num_classes: int, fp8_weights: bool | None = None) -> HipFasterViT:
    """'faster_vit_2_224' (the reference's MODEL_NAME, trainers/fastervit.py:62), 'faster_vit_0_224' (BASELINE config 5), ...
    fp8_weights None: $FP8_WEIGHTS (YAML `training.fp8_weights` / `inference.fp8_weights`, exported by the orchestrator)."""
    import os

    parts = name.lower().replace("-", "_").split("_")
    res = int(parts[3]) if len(parts) > 3 and parts[3].isdigit() else 224
    if fp8_weights is None:
        fp8_weights = os.environ.get("FP8_WEIGHTS", "0").lower() in {"1", "true", "yes", "on"}
    return HipFasterViT(variant_of(name), num_classes, res, fp8_weights=fp8_weights)


__all__ = ["HipFasterViT", "build_fastervit", "variant_of"]
