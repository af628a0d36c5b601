"""HIP-backed EfficientFormerV2: drop-in for `timm.create_model("efficientformerv2_s1", ...)`.

The reference builds this model in trainers/efficientformer_v2.py:327 (`timm.create_model(MODEL_NAME, pretrained=True,
num_classes=num_classes, img_size=img_size)`) and orchestration/model_registry.py:39-40, and touches only:
  * `forward(float[N,3,H,W]) -> float[N,num_classes]` (the average of `head` and `head_dist`);
  * `.named_parameters()` — the warm-up trains names containing `classifier` or `head`
    (trainers/efficientformer_v2.py:351-352; that substring also catches `talking_head1/2` and `head_dist`),
    fine-tuning trains names containing any of UNFREEZE_KEYS (`stages.3`, `blocks.3`, `head`, ...; :66-74,389-393);
  * `.state_dict()/.load_state_dict()` with timm's keys (orchestrator.py:370-375), `.to()`, `.train()/.eval()`.
The module tree below carries exactly timm 1.0.20's parameter names (stem.conv1.conv.weight, stages.2.blocks.7.
token_mixer.talking_head1.weight, stages.3.downsample.attn.q.local.weight, ...): real nn.Conv2d / nn.BatchNorm2d /
nn.Linear objects as parameter containers, arithmetic in vit_functions.py on the kernels of libdfd_hip.so.  No ATen
fallback: a CPU input raises.  Architecture tables: timm efficientformer_v2.py (S0 / S1 / S2 / L), restated
independently in oracle/efformer_ref.py, which this module is tested against.
"""

from __future__ import annotations

import math

import torch
from torch import nn

from .efficientnet import compute_dtype
from .functions import BNRef, bn_eval_batch
from .vit_functions import (AttentionCtx, AttentionFunction, AttnGeom, ConvMlpCtx, ConvMlpFunction, ConvStemCtx, ConvStemFunction,
                            DenseConvBNFunction, DenseConvCtx, DownsampleCtx, DownsampleFunction, TailCtx, TailFunction)
from ._lib import ACT_GELU

_WIDTHS = {"s0": (32, 48, 96, 176), "s1": (32, 48, 120, 224), "s2": (32, 64, 144, 288), "l": (40, 80, 192, 384)}
_DEPTHS = {"s0": (2, 2, 6, 4), "s1": (3, 3, 9, 6), "s2": (4, 4, 12, 8), "l": (5, 5, 15, 10)}
_RATIOS = {
    "s0": ((4, 4), (4, 4), (4, 3, 3, 3, 4, 4), (4, 3, 3, 4)),
    "s1": ((4, 4, 4), (4, 4, 4), (4, 4, 3, 3, 3, 3, 4, 4, 4), (4, 4, 3, 3, 4, 4)),
    "s2": ((4, 4, 4, 4), (4, 4, 4, 4), (4, 4, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4), (4, 4, 3, 3, 3, 3, 4, 4)),
    "l": ((4, 4, 4, 4, 4), (4, 4, 4, 4, 4), (4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4), (4, 4, 4, 3, 3, 3, 3, 4, 4, 4)),
}
_NUM_VIT = {"s0": 2, "s1": 2, "s2": 4, "l": 6}
_DROP_PATH = {"s0": 0.0, "s1": 0.0, "s2": 0.02, "l": 0.1}
_EPS = 1e-5


def _bnref(bn: nn.BatchNorm2d) -> BNRef:
    return BNRef(bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps)


class ConvNorm(nn.Module):
    """Parameter container named like timm's ConvNorm / ConvNormAct: `.conv` (with bias) and `.bn`."""

    def __init__(self, cin: int, cout: int, k: int = 1, stride: int = 1, groups: int = 1) -> None:
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=True)
        self.bn = nn.BatchNorm2d(cout, eps=_EPS)

    def tensors(self):
        return self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias


class LayerScale2d(nn.Module):
    def __init__(self, dim: int, init: float = 1e-5) -> None:
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))


def _bias_index(q_res, k_res, step: int) -> torch.Tensor:
    ky, kx = torch.meshgrid(torch.arange(k_res[0]), torch.arange(k_res[1]), indexing="ij")
    qy, qx = torch.meshgrid(torch.arange(0, k_res[0], step), torch.arange(0, k_res[1], step), indexing="ij")
    dy = (qy.reshape(-1, 1) - ky.reshape(1, -1)).abs()
    dx = (qx.reshape(-1, 1) - kx.reshape(1, -1)).abs()
    return dy * k_res[1] + dx


class HipAttention2d(nn.Module):
    def __init__(self, dim: int, resolution, stride: int | None, heads: int = 8, key_dim: int = 32, attn_ratio: int = 4) -> None:
        super().__init__()
        self.heads, self.key_dim, self.stride = heads, key_dim, stride
        if stride is not None:
            resolution = tuple(math.ceil(r / stride) for r in resolution)
            self.stride_conv = ConvNorm(dim, dim, 3, stride, groups=dim)
        else:
            self.stride_conv = None
        self.resolution = resolution
        self.N = resolution[0] * resolution[1]
        self.d = attn_ratio * key_dim
        self.dh = self.d * heads
        self.q = ConvNorm(dim, key_dim * heads)
        self.k = ConvNorm(dim, key_dim * heads)
        self.v = ConvNorm(dim, self.dh)
        self.v_local = ConvNorm(self.dh, self.dh, 3, groups=self.dh)
        self.talking_head1 = nn.Conv2d(heads, heads, 1)
        self.talking_head2 = nn.Conv2d(heads, heads, 1)
        self.proj = ConvNorm(self.dh, dim)
        self.attention_biases = nn.Parameter(torch.zeros(heads, self.N))
        idx = _bias_index(resolution, resolution, 1)
        self.register_buffer("attention_bias_idxs", idx, persistent=False)
        self.register_buffer("_idx32", idx.reshape(-1).to(torch.int32), persistent=False)

    def run(self, x, ls: LayerScale2d, row_scale, training: bool, counters, derived=None) -> torch.Tensor:
        names = ([] if self.stride_conv is None else ["stride_conv"]) + ["q", "k", "v", "v_local", "proj"]
        mods = {nm: getattr(self, nm) for nm in names}
        flat = [t for nm in names for t in mods[nm].tensors()]
        geo = AttnGeom(self.heads, self.key_dim, self.d, self.N, self.N, self.key_dim ** -0.5)
        cfg = AttentionCtx(geo, self.stride, {nm: _bnref(m.bn) for nm, m in mods.items()}, self._idx32, training, counters, derived)
        return AttentionFunction.apply(x, cfg, *flat, self.talking_head1.weight, self.talking_head1.bias, self.talking_head2.weight,
                                       self.talking_head2.bias, self.attention_biases, ls.gamma, row_scale)


class HipConvMlp(nn.Module):
    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = ConvNorm(dim, hidden, 1)
        self.mid = ConvNorm(hidden, hidden, 3, groups=hidden)
        self.fc2 = ConvNorm(hidden, dim, 1)

    def run(self, x, ls: LayerScale2d, row_scale, training: bool, counters, derived=None) -> torch.Tensor:
        from .hooks import has_hooks

        if has_hooks(self.fc2.conv):
            return self._hooked_run(x, ls, training)
        cfg = ConvMlpCtx(_bnref(self.fc1.bn), _bnref(self.mid.bn), _bnref(self.fc2.bn), training, counters, derived)
        return ConvMlpFunction.apply(x, *self.fc1.tensors(), *self.mid.tensors(), *self.fc2.tensors(), ls.gamma, row_scale, cfg)

    def _hooked_run(self, x, ls: LayerScale2d, training: bool) -> torch.Tensor:
        """Forward hooks on fc2's convolution (Grad-CAM's "last nn.Conv2d" for this family, web_ui.py:95-114): the MLP runs
        unfused in eval mode so that the hooks see (module, (input,), output) as NCHW tensors and the output is wired to the
        logits:  a = GELU(BN(dw(GELU(BN(fc1 x)))))  [no graph]  ->  y = conv(a) + bias  [hooked]  ->  x + ls * BN(y)."""
        from . import kernels as K
        from ._lib import ACT_GELU, ACT_NONE
        from .functions import HeadConvFunction, _bn_state
        from .hooks import ChannelAffineFunction, call_hooks

        if training:
            raise NotImplementedError("forward hooks on mlp.fc2.conv are supported in eval mode only (the training path "
                                      "keeps the whole ConvMlp in one fused stage)")
        N, H, W, C = x.shape
        rows = N * H * W
        w1, b1, g1, be1 = self.fc1.tensors()
        wd, bd, gd, bed = self.mid.tensors()
        w2, b2, g2, be2 = self.fc2.tensors()
        with torch.no_grad():
            w1_nk, _ = K.prep_weights(w1, x.dtype, True, False)
            y1, _, _ = K.pwconv(x, None, w1_nk, None, stats=False)
            st1 = _bn_state(None, 0, rows, _bnref(self.fc1.bn), g1, be1, False, None, conv_bias=b1)
            y2, _, _ = K.dwconv_fwd(y1, st1, ACT_GELU, wd, 3, 1, 1, 1, H, W, stats=False)
            st2 = _bn_state(None, 0, rows, _bnref(self.mid.bn), gd, bed, False, None, conv_bias=bd)
            a = K.bn_act_apply(y2, st2, ACT_GELU)
            bn = self.fc2.bn
            scale = g2 / torch.sqrt(bn.running_var + bn.eps)
            shift = be2 - bn.running_mean * scale
            ones = torch.ones_like(scale)
            bias = b2 if b2 is not None else torch.zeros_like(scale)
        y = HeadConvFunction.apply(a, w2)                                   # raw 1x1 convolution, NHWC
        y = ChannelAffineFunction.apply(y, ones, bias, None)                # + conv bias: what the module's forward returns
        y = call_hooks(self.fc2.conv, a, y)
        return ChannelAffineFunction.apply(y, (ls.gamma * scale).detach(), (ls.gamma * shift).detach(), x.detach())


class HipBlock(nn.Module):
    def __init__(self, dim: int, ratio: int, resolution, stride, use_attn: bool, drop_path: float, index: int) -> None:
        super().__init__()
        if use_attn:
            self.token_mixer = HipAttention2d(dim, resolution, stride)
            self.ls1 = LayerScale2d(dim)
        else:
            self.token_mixer = None
            self.ls1 = None
        self.mlp = HipConvMlp(dim, int(dim * ratio))
        self.ls2 = LayerScale2d(dim)
        self.drop_path = drop_path
        self.index = index

    def _scale(self, x, rng, which: int):
        if not self.training or self.drop_path <= 0.0 or rng is None:
            return None
        return rng.drop_path_scale(x.shape[0], 1.0 - self.drop_path, stream_id=2 * self.index + which)

    def forward(self, x, rng=None, counters=None, derived=None):
        if self.token_mixer is not None:
            x = self.token_mixer.run(x, self.ls1, self._scale(x, rng, 0), self.training, counters, derived)
        return self.mlp.run(x, self.ls2, self._scale(x, rng, 1), self.training, counters, derived)


class HipLocalGlobalQuery(nn.Module):
    def __init__(self, dim: int, out_dim: int) -> None:
        super().__init__()
        self.pool = nn.AvgPool2d(1, 2, 0)
        self.local = nn.Conv2d(dim, dim, 3, stride=2, padding=1, groups=dim)
        self.proj = ConvNorm(dim, out_dim, 1)


class HipAttention2dDownsample(nn.Module):
    def __init__(self, dim: int, out_dim: int, resolution, heads: int = 8, key_dim: int = 16, attn_ratio: int = 4) -> None:
        super().__init__()
        self.heads, self.key_dim = heads, key_dim
        self.resolution = resolution
        self.resolution2 = tuple(math.ceil(r / 2) for r in resolution)
        self.N, self.N2 = resolution[0] * resolution[1], self.resolution2[0] * self.resolution2[1]
        self.d = attn_ratio * key_dim
        self.dh = self.d * heads
        self.q = HipLocalGlobalQuery(dim, key_dim * heads)
        self.k = ConvNorm(dim, key_dim * heads, 1)
        self.v = ConvNorm(dim, self.dh, 1)
        self.v_local = ConvNorm(self.dh, self.dh, 3, 2, groups=self.dh)
        self.proj = ConvNorm(self.dh, out_dim, 1)
        self.attention_biases = nn.Parameter(torch.zeros(heads, self.N))
        idx = _bias_index(self.resolution2, resolution, 2)
        self.register_buffer("attention_bias_idxs", idx, persistent=False)
        self.register_buffer("_idx32", idx.reshape(-1).to(torch.int32), persistent=False)


class HipDownsample(nn.Module):
    def __init__(self, cin: int, cout: int, resolution, use_attn: bool) -> None:
        super().__init__()
        self.conv = ConvNorm(cin, cout, 3, 2)
        self.attn = HipAttention2dDownsample(cin, cout, resolution) if use_attn else None

    def forward(self, x, counters=None, derived=None):
        flat = list(self.conv.tensors())
        if self.attn is None:
            cfg = DownsampleCtx(_bnref(self.conv.bn), self.training, counters)
            return DownsampleFunction.apply(x, cfg, *flat)
        a = self.attn
        mods = {"q_proj": a.q.proj, "k": a.k, "v": a.v, "v_local": a.v_local, "proj": a.proj}
        flat += [a.q.local.weight, a.q.local.bias]
        for nm in ("q_proj", "k", "v", "v_local", "proj"):
            flat += list(mods[nm].tensors())
        flat.append(a.attention_biases)
        geo = AttnGeom(a.heads, a.key_dim, a.d, a.N2, a.N, a.key_dim ** -0.5)
        cfg = DownsampleCtx(_bnref(self.conv.bn), self.training, counters, True, geo, {nm: _bnref(m.bn) for nm, m in mods.items()},
                            a._idx32, derived)
        return DownsampleFunction.apply(x, cfg, *flat)


class HipStage(nn.Module):
    def __init__(self, dim, dim_out, depth, resolution, downsample, block_stride, downsample_attn, block_attn, num_vit, ratios,
                 dprs, first_index) -> None:
        super().__init__()
        if downsample:
            self.downsample = HipDownsample(dim, dim_out, resolution, downsample_attn)
            dim = dim_out
            resolution = tuple(math.ceil(r / 2) for r in resolution)
        else:
            self.downsample = nn.Identity()
        first_attn = depth - num_vit
        self.blocks = nn.Sequential(*[
            HipBlock(dim, ratios[i], resolution, block_stride, block_attn and i >= first_attn, dprs[i], first_index + i)
            for i in range(depth)])

    def forward(self, x, rng=None, counters=None, derived=None):
        if not isinstance(self.downsample, nn.Identity):
            x = self.downsample(x, counters, derived)
        for blk in self.blocks:
            x = blk(x, rng, counters, derived)
        return x


class HipStem4(nn.Module):
    def __init__(self, cout: int) -> None:
        super().__init__()
        self.conv1 = ConvNorm(3, cout // 2, 3, 2)
        self.conv2 = ConvNorm(cout // 2, cout, 3, 2)


class HipEfficientFormerV2(nn.Module):
    """EfficientFormerV2-{S0,S1,S2,L} whose forward/backward run on the MI355X kernels."""

    def __init__(self, variant: str = "s1", num_classes: int = 1000, img_size: int = 224, drop_rate: float = 0.0,
                 drop_path_rate: float | None = None) -> None:
        super().__init__()
        if variant not in _WIDTHS:
            raise KeyError(f"unknown EfficientFormerV2 variant '{variant}'")
        widths, depths, ratios = _WIDTHS[variant], _DEPTHS[variant], _RATIOS[variant]
        dpr_total = _DROP_PATH[variant] if drop_path_rate is None else drop_path_rate
        self.variant, self.num_classes, self.img_size = variant, num_classes, img_size
        self.stem = HipStem4(widths[0])
        dprs = torch.linspace(0, dpr_total, sum(depths)).split(list(depths))
        stages, prev, stride, index = [], widths[0], 4, 0
        for i in range(4):
            res = (math.ceil(img_size / stride),) * 2
            stages.append(HipStage(prev, widths[i], depths[i], res, i > 0, 2 if i == 2 else None, i >= 3, i >= 2, _NUM_VIT[variant],
                                   ratios[i], [float(v) for v in dprs[i]], index))
            index += depths[i]
            if i > 0:
                stride *= 2
            prev = widths[i]
        self.stages = nn.Sequential(*stages)
        self.num_features = self.head_hidden_size = widths[-1]
        self.norm = nn.BatchNorm2d(widths[-1], eps=_EPS)
        self.head_drop = nn.Dropout(drop_rate)
        self.drop_rate = drop_rate
        self.head = nn.Linear(widths[-1], num_classes)
        self.head_dist = nn.Linear(widths[-1], num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def to(self, *args, **kwargs):
        """`.to(memory_format=torch.channels_last)` (trainers/efficientformer_v2.py:328) is accepted and ignored for the
        PARAMETERS (kernels read torch's default weight layouts; activations are NHWC inside the engine)."""
        kwargs.pop("memory_format", None)
        args = tuple(a for a in args if not isinstance(a, torch.memory_format))
        return super().to(*args, **kwargs) if (args or kwargs) else self

    def rng(self, device: torch.device):
        from . import kernels as K

        cur = self.__dict__.get("_rng_obj")
        if cur is None or cur.state.device != device:
            cur = self.__dict__["_rng_obj"] = K.DeviceRng(device)
        return cur

    def dp_cut_modules(self) -> list[nn.Module]:
        """Where a replayed data-parallel backward may be cut into segments (graph_step.plan_cuts): the four stages."""
        return list(self.stages)

    def forward_features_nhwc(self, x: torch.Tensor, counters: list | None = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("HipEfficientFormerV2 runs on a HIP device only (no CPU fallback); move the input with .to('cuda')")
        if x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise ValueError(f"EfficientFormerV2 was built for {self.img_size}x{self.img_size} inputs (attention bias tables "
                             f"depend on the resolution), got {tuple(x.shape[2:])}")
        dt = compute_dtype()
        tr = self.training
        xh = x.detach().float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        c1, c2 = self.stem.conv1, self.stem.conv2
        h = ConvStemFunction.apply(xh, *c1.tensors(), ConvStemCtx(2, 1, _bnref(c1.bn), dt, tr, ACT_GELU, counters))
        h = DenseConvBNFunction.apply(h, *c2.tensors(), DenseConvCtx(3, 2, _bnref(c2.bn), tr, ACT_GELU, counters))
        rng = self.rng(x.device) if tr else None
        derived = self._derived_weights(dt)
        for stage in self.stages:
            h = stage(h, rng, counters, derived)
        return h

    def _derived_weights(self, dt: torch.dtype) -> dict:
        """{weight.data_ptr(): (w_nk, w_kn)} for every 1x1 convolution of the network, refreshed by ONE batched launch per
        forward pass (kernels.DerivedWeights); one cache entry per activation dtype, never freed by a train / eval switch
        (a captured hipGraph holds raw pointers into it)."""
        from . import kernels as K

        weights = [m.conv.weight for m in self.modules() if isinstance(m, ConvNorm) and m.conv.kernel_size == (1, 1)]
        caches = self.__dict__.setdefault("_derived_caches", {})
        cache = caches.get(dt)
        if cache is None or not cache.valid_for(weights, dt):
            with torch.inference_mode(False):
                cache = caches[dt] = K.DerivedWeights([(w, True, True, False) for w in weights], dt)
        cache.refresh()
        return {w.data_ptr(): pair for w, pair in zip(weights, cache.out)}

    def forward(self, x: torch.Tensor, dropout_u: torch.Tensor | None = None) -> torch.Tensor:
        with bn_eval_batch(self.__dict__, (self.training, compute_dtype())):     # eval-mode BN blocks: one batched launch per pass
            return self._forward(x, dropout_u)

    def _forward(self, x: torch.Tensor, dropout_u: torch.Tensor | None = None) -> torch.Tensor:
        counters: list = []
        h = self.forward_features_nhwc(x, counters)
        u = dropout_u
        if u is None and self.training and self.drop_rate > 0:
            u = self.rng(h.device).uniform(h.shape[0] * h.shape[3], stream_id=1 << 20).view(h.shape[0], h.shape[3])
        cfg = TailCtx(_bnref(self.norm), self.drop_rate, self.training, counters)
        out = TailFunction.apply(h, self.norm.weight, self.norm.bias, self.head.weight, self.head.bias, self.head_dist.weight,
                                 self.head_dist.bias, u, cfg)
        if self.training:
            self.rng(h.device).tick(counters)
        return out


def variant_of(name: str) -> str:
    key = name.lower().replace("-", "_")
    for v in ("s0", "s1", "s2", "l"):
        if key.endswith("_" + v) or key.endswith("v2" + v):
            return v
    raise KeyError(f"not an EfficientFormerV2 name: {name}")


def build_efficientformer_v2(name: str, num_classes: int, img_size: int = 224) -> HipEfficientFormerV2:
    """'efficientformerv2_s1' (the reference's MODEL_NAME, trainers/efficientformer_v2.py:54) and its siblings s0 / s2 / l."""
    return HipEfficientFormerV2(variant_of(name), num_classes, img_size)


__all__ = ["HipEfficientFormerV2", "build_efficientformer_v2", "variant_of"]
