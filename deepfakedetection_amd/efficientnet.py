"""HIP-backed EfficientNet modules: drop-in for the nn.Modules the reference builds.

`HipEfficientNet(variant, flavour, num_classes)` exposes the same surface the reference
touches on its third-party models:
  * `forward(float[N,3,H,W]) -> float[N,num_classes]` raw logits, any memory format;
  * `.to()`, `.train()/.eval()`, `.named_parameters()` with the third-party names
    (`_fc` / `classifier` substrings drive the trainers' freeze masks,
    trainers/efficientnet.py:433-437), `.state_dict()/.load_state_dict()` with the
    third-party KEYS so released weights load (orchestrator.py:370-375);
  * lukemelas flavour: attributes `_fc`, `_fc.in_features`, `_conv_head`
    (trainers/efficientnet.py:406-407, web_ui.py:111); `.modules()` yields nn.Conv2d.
The sub-modules are real nn.Conv2d / nn.BatchNorm2d / nn.Linear objects used as
parameter containers; the arithmetic runs in the fused functions of functions.py on the
kernels of libdfd_hip.so.  There is no ATen fallback: a CPU input raises.

Compute dtype: bf16 activations when called under torch autocast (the reference's
AMP region, trainers/efficientnet.py:296), f32 otherwise (evaluate / inference run
without autocast: trainers/efficientnet.py:249-254, orchestrator.py:587-590).
"""

from __future__ import annotations

import math
import os
import warnings

import torch
from torch import nn

from . import functions as F_
from .arch import BlockPlan, NetPlan, efficientnet_plan
from .functions import (BNRef, HeadConvFunction, HeadCtx, HeadFunction, HeadTailEvalFunction, MBConvCtx, MBConvFunction,
                        StemCtx, StemFunction)

# A/B switch for experiments, read once at import: per-layer weight preparation instead of the batched launch
_NO_DERIVED = os.environ.get("DFD_NO_DERIVED") == "1"
_default_rng: dict = {}


def default_rng(device: torch.device):
    """Process-wide Philox state per device, for stages used outside of a HipEfficientNet (tests, plug-ins)."""
    from . import kernels as K

    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    rng = _default_rng.get(key)
    if rng is None:
        rng = _default_rng[key] = K.DeviceRng(device)
    return rng


_LM_NAMES = dict(expand="_expand_conv", expand_bn="_bn0", dw="_depthwise_conv", dw_bn="_bn1",
                 se_reduce="_se_reduce", se_expand="_se_expand", project="_project_conv", project_bn="_bn2")
_TIMM_IR_NAMES = dict(expand="conv_pw", expand_bn="bn1", dw="conv_dw", dw_bn="bn2",
                      se_reduce="se.conv_reduce", se_expand="se.conv_expand", project="conv_pwl", project_bn="bn3")
_TIMM_DS_NAMES = dict(dw="conv_dw", dw_bn="bn1", se_reduce="se.conv_reduce", se_expand="se.conv_expand",
                      project="conv_pw", project_bn="bn2")


def _bnref(bn: nn.BatchNorm2d) -> BNRef:
    pre = None
    held = bn.__dict__.get("_dfd_eval")               # (kernels.EvalBNStates, this layer's block), set by the owning network
    if held is not None and held[0].fresh and not bn.training:
        pre = held[1]
    return BNRef(bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps, pre)


def compute_dtype(device_type: str = "cuda") -> torch.dtype:
    """bf16 inside an autocast region, f32 outside."""
    try:
        enabled = torch.is_autocast_enabled(device_type)
    except TypeError:  # older signature
        enabled = torch.is_autocast_enabled()
    if not enabled:
        return torch.float32
    amp = torch.get_autocast_dtype(device_type) if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
    if amp != torch.bfloat16:
        warnings.warn("dfd HIP engine computes autocast regions in bf16 (fp16 autocast requested)", stacklevel=3)
    return torch.bfloat16


class _SEParams(nn.Module):
    """Parameter holder named like timm's SqueezeExcite (se.conv_reduce / se.conv_expand)."""

    def __init__(self, channels: int, reduced: int) -> None:
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, reduced, 1, bias=True)
        self.conv_expand = nn.Conv2d(reduced, channels, 1, bias=True)


class HipMBConv(nn.Module):
    """One MBConv / depthwise-separable block; parameters named per flavour."""

    def __init__(self, plan: BlockPlan, net: NetPlan) -> None:
        super().__init__()
        self.plan = plan
        lm = net.flavour == "lukemelas"
        self._names = _LM_NAMES if lm else (_TIMM_IR_NAMES if plan.expand else _TIMM_DS_NAMES)
        names = self._names

        def bn(c: int) -> nn.BatchNorm2d:
            return nn.BatchNorm2d(c, eps=net.bn_eps, momentum=net.bn_momentum)

        if plan.expand:
            setattr(self, names["expand"], nn.Conv2d(plan.cin, plan.cmid, 1, bias=False))
            setattr(self, names["expand_bn"], bn(plan.cmid))
        setattr(self, names["dw"], nn.Conv2d(plan.cmid, plan.cmid, plan.dw.kernel, stride=plan.dw.stride,
                                              groups=plan.cmid, bias=False))
        setattr(self, names["dw_bn"], bn(plan.cmid))
        if lm:
            self._se_reduce = nn.Conv2d(plan.cmid, plan.se_width, 1, bias=True)
            self._se_expand = nn.Conv2d(plan.se_width, plan.cmid, 1, bias=True)
        else:
            self.se = _SEParams(plan.cmid, plan.se_width)
        setattr(self, names["project"], nn.Conv2d(plan.cmid, plan.cout, 1, bias=False))
        setattr(self, names["project_bn"], bn(plan.cout))

    def part(self, role: str) -> nn.Module:
        mod: nn.Module = self
        for piece in self._names[role].split("."):
            mod = getattr(mod, piece)
        return mod

    def forward(self, x: torch.Tensor, derived: tuple | None = None, rng=None, counters: list | None = None) -> torch.Tensor:
        """x: NHWC.  Drop-connect (efficientnet_pytorch utils.drop_connect): per-sample Bernoulli(keep) / keep, drawn
        by the Philox kernel from the owning network's device-resident state."""
        p = self.plan
        row_scale = None
        if self.training and p.skip and p.drop_connect > 0:
            rng = rng if rng is not None else default_rng(x.device)
            row_scale = rng.drop_path_scale(x.shape[0], 1.0 - p.drop_connect, stream_id=p.index)
        return self.run(x, row_scale, derived, counters)

    def run(self, x: torch.Tensor, row_scale: torch.Tensor | None, derived: tuple | None = None,
            counters: list | None = None) -> torch.Tensor:
        p = self.plan
        dw, dw_bn = self.part("dw"), self.part("dw_bn")
        ser, see = self.part("se_reduce"), self.part("se_expand")
        proj, proj_bn = self.part("project"), self.part("project_bn")
        if p.expand:
            exp, exp_bn = self.part("expand"), self.part("expand_bn")
            w_exp, g_exp, b_exp, ref_exp = exp.weight, exp_bn.weight, exp_bn.bias, _bnref(exp_bn)
        else:
            w_exp = g_exp = b_exp = ref_exp = None
        cfg = MBConvCtx(p.expand, p.dw, p.skip, ref_exp, _bnref(dw_bn), _bnref(proj_bn), self.training, derived, counters,
                        torch.is_grad_enabled())
        return MBConvFunction.apply(x, w_exp, g_exp, b_exp, dw.weight, dw_bn.weight, dw_bn.bias, ser.weight, ser.bias,
                                    see.weight, see.bias, proj.weight, proj_bn.weight, proj_bn.bias, row_scale, cfg)


class HipEfficientNet(nn.Module):
    """EfficientNet-{b0..b4} whose forward/backward run on the MI355X kernels."""

    def __init__(self, variant: str = "b0", flavour: str = "timm", num_classes: int = 1000,
                 drop_rate: float | None = None) -> None:
        super().__init__()
        plan = efficientnet_plan(variant, flavour)
        self.plan = plan
        self.flavour = flavour
        self.drop_rate = plan.dropout if drop_rate is None else drop_rate
        last = plan.blocks[-1].cout

        def bn(c: int) -> nn.BatchNorm2d:
            return nn.BatchNorm2d(c, eps=plan.bn_eps, momentum=plan.bn_momentum)

        stem = nn.Conv2d(3, plan.stem_out, 3, stride=2, bias=False)
        head = nn.Conv2d(last, plan.head_out, 1, bias=False)
        blocks = [HipMBConv(b, plan) for b in plan.blocks]
        if flavour == "lukemelas":
            self._conv_stem, self._bn0 = stem, bn(plan.stem_out)
            self._blocks = nn.ModuleList(blocks)
            self._conv_head, self._bn1 = head, bn(plan.head_out)
            self._avg_pooling = nn.AdaptiveAvgPool2d(1)
            self._dropout = nn.Dropout(self.drop_rate)
            self._fc = nn.Linear(plan.head_out, num_classes)
        else:
            self.conv_stem, self.bn1 = stem, bn(plan.stem_out)
            stages, at = [], 0
            for count in plan.stage_sizes:
                stages.append(nn.Sequential(*blocks[at:at + count]))
                at += count
            self.blocks = nn.Sequential(*stages)
            self.conv_head, self.bn2 = head, bn(plan.head_out)
            self.global_pool = nn.AdaptiveAvgPool2d(1)
            self.classifier = nn.Linear(plan.head_out, num_classes)
            self._init_timm()
        self.num_features = plan.head_out

    # -- timm's efficientnet_init_weights: fan-out normal convs, unit BN, uniform classifier
    def _init_timm(self) -> None:
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                bound = 1.0 / math.sqrt(m.weight.shape[0])
                nn.init.uniform_(m.weight, -bound, bound)
                nn.init.zeros_(m.bias)

    def to(self, *args, **kwargs):
        """`.to(memory_format=torch.channels_last)` (trainers/efficientnet.py:409) is accepted and
        ignored for the PARAMETERS: the kernels read weights in torch's default [O][I][kh][kw]
        layout, while activations are NHWC inside the engine whatever the input's strides."""
        kwargs.pop("memory_format", None)
        args = tuple(a for a in args if not isinstance(a, torch.memory_format))
        return super().to(*args, **kwargs) if (args or kwargs) else self

    # -- named parts, resolved at call time (the reference swaps `_fc` after construction)
    def _parts(self):
        if self.flavour == "lukemelas":
            return self._conv_stem, self._bn0, list(self._blocks), self._conv_head, self._bn1, self._fc
        flat = [b for stage in self.blocks for b in stage]
        return self.conv_stem, self.bn1, flat, self.conv_head, self.bn2, self.classifier

    def block_list(self) -> list[HipMBConv]:
        return self._parts()[2]

    def rng(self, device: torch.device):
        """This network's Philox state on `device` (created on first use, seeded from torch.initial_seed())."""
        from . import kernels as K

        cur = self.__dict__.get("_rng_obj")
        if cur is None or cur.state.device != device:
            cur = self.__dict__["_rng_obj"] = K.DeviceRng(device)
        return cur

    def dp_cut_modules(self) -> list[nn.Module]:
        """Modules after which a replayed data-parallel backward may be cut into segments (graph_step.plan_cuts): the MBConv blocks —
        one tensor in, one tensor out, in forward order."""
        return list(self.block_list())

    def forward_features_nhwc(self, x: torch.Tensor, drop_masks=None, counters: list | None = None) -> torch.Tensor:
        stem, stem_bn, blocks, _, _, _ = self._parts()
        if not x.is_cuda:
            raise RuntimeError("HipEfficientNet runs on a HIP device only (no CPU fallback); move the input with .to('cuda')")
        dt = compute_dtype()
        xh = x.detach().float().contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        h = StemFunction.apply(xh, stem.weight, stem_bn.weight, stem_bn.bias,
                               StemCtx(self.plan.stem, _bnref(stem_bn), dt, self.training, counters))
        derived = self._derived_weights(dt)
        rng = self.rng(x.device) if self.training else None
        for i, blk in enumerate(blocks):
            h = blk(h, derived[i], rng, counters) if drop_masks is None else blk.run(h, drop_masks[i], derived[i], counters)
        self._head_derived = derived[len(blocks)]
        return h

    def _derived_weights(self, dt: torch.dtype) -> list:
        """One batched launch per forward refreshes every derived weight of the network (1x1 conv weights in
        the activation dtype as [N][K] and [K][N], [R][C] copies of the squeeze-excite expand weights);
        returns, per block, ((wexp_nk, wexp_kn) | None, (wproj_nk, wproj_kn), se_w2t) and last the head's pair.
        The buffers and the job list are rebuilt only when a parameter moved (e.g. after .to())."""
        from . import kernels as K

        _, _, blocks, head, _, _ = self._parts()
        if _NO_DERIVED:                                        # A/B switch: per-layer preparation
            return [None] * (len(blocks) + 1)
        items, layout = [], []
        for blk in blocks:
            p = blk.plan
            if p.expand:
                items.append((blk.part("expand").weight, True, True, False))
            items.append((blk.part("project").weight, True, True, False))
            items.append((blk.part("se_expand").weight, False, True, True))
            layout.append(p.expand)
        items.append((head.weight, True, True, False))
        # one cache entry PER DTYPE: the trainer alternates bf16 training and f32 evaluation, and a captured
        # hipGraph holds raw pointers into its entry, so an entry is never freed by a train/eval switch; the
        # buffers are ordinary (non-inference) tensors even when first built under torch.inference_mode()
        caches = self.__dict__.setdefault("_derived_caches", {})
        cache = caches.get(dt)
        if cache is None or not cache.valid_for([it[0] for it in items], dt):
            with torch.inference_mode(False):
                cache = caches[dt] = K.DerivedWeights(items, dt)
        cache.refresh()
        out, at = [], 0
        for has_exp in layout:
            exp = None
            if has_exp:
                exp = cache.out[at]
                at += 1
            proj = cache.out[at]
            w2t = cache.out[at + 1][1]
            at += 2
            out.append((exp, proj, w2t))
        out.append(cache.out[at])
        return out

    def _eval_bn_states(self):
        """Eval / inference forward: every BatchNorm's coefficient block from one batched launch (kernels.EvalBNStates)."""
        from . import kernels as K

        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        cache = self.__dict__.get("_eval_bn_cache")
        if cache is None or not cache.valid_for(bns):
            with torch.inference_mode(False):
                cache = self.__dict__["_eval_bn_cache"] = K.EvalBNStates(bns)
            for bn, st in zip(bns, cache.states):
                bn.__dict__["_dfd_eval"] = (cache, st)
        cache.refresh()
        return cache

    def forward(self, x: torch.Tensor, drop_masks=None, dropout_u: torch.Tensor | None = None) -> torch.Tensor:
        """drop_masks / dropout_u let a test inject the stochastic-depth masks (already
        1/keep scaled, one [N] tensor or None per block) and the dropout uniforms."""
        if not self.training and x.is_cuda:
            states = self._eval_bn_states()
            states.fresh = True
            try:
                return self._forward(x, drop_masks, dropout_u)
            finally:
                states.fresh = False
        return self._forward(x, drop_masks, dropout_u)

    def _forward(self, x: torch.Tensor, drop_masks=None, dropout_u: torch.Tensor | None = None) -> torch.Tensor:
        _, _, _, head, head_bn, fc = self._parts()
        counters: list = []                 # owned by this call: num_batches_tracked of every BatchNorm that ran
        h = self.forward_features_nhwc(x, drop_masks, counters)
        u = dropout_u
        if u is None and self.training and self.drop_rate > 0 and drop_masks is None:
            u = self.rng(h.device).uniform(h.shape[0] * head.out_channels, stream_id=1 << 20).view(h.shape[0], head.out_channels)
        if head._forward_hooks or head._forward_pre_hooks:
            out = self._hooked_head(h, head, head_bn, fc)
        else:
            cfg = HeadCtx(_bnref(head_bn), self.drop_rate, self.training, self._head_derived, counters)
            out = HeadFunction.apply(h, head.weight, head_bn.weight, head_bn.bias, fc.weight, fc.bias, u, cfg)
        if self.training:
            self.rng(h.device).tick(counters)       # one launch: every counter += 1, Philox offset += 1
        return out


def _hooked_head(self, h, head, head_bn, fc):
    """Forward hooks on the head convolution (Grad-CAM, web_ui.py:96-114): run the head unfused so that the
    hooks see (module, (input,), output) as NCHW tensors wired into autograd.  Eval mode only."""
    if self.training:
        raise NotImplementedError("forward hooks on the head convolution are supported in eval mode only "
                                  "(the training path keeps conv + BN + SiLU + pool + classifier in one fused stage)")
    x_nchw = h.permute(0, 3, 1, 2)
    for hook in head._forward_pre_hooks.values():
        hook(head, (x_nchw,))
    y = HeadConvFunction.apply(h, head.weight)
    y_nchw = y.permute(0, 3, 1, 2)
    for hook in head._forward_hooks.values():
        r = hook(head, (x_nchw,), y_nchw)
        if r is not None:
            y_nchw = r
    y2 = y_nchw.permute(0, 2, 3, 1)
    if not y2.is_contiguous():
        y2 = y2.contiguous()
    return HeadTailEvalFunction.apply(y2, fc.weight, fc.bias, _bnref(head_bn), head_bn.weight, head_bn.bias)


HipEfficientNet._hooked_head = _hooked_head


def build_efficientnet(name: str, num_classes: int) -> HipEfficientNet:
    """'efficientnet_b3' / 'efficientnet-b3' -> lukemelas flavour (the reference's model);
    'efficientnet_b0' and friends with suffix '.timm' or variant b0 -> timm flavour (BASELINE)."""
    key = name.lower().replace("-", "_")
    flavour = None
    if key.endswith(".timm"):
        key, flavour = key[:-5], "timm"
    elif key.endswith(".lukemelas"):
        key, flavour = key[:-10], "lukemelas"
    if not key.startswith("efficientnet_b"):
        raise KeyError(f"not an EfficientNet name: {name}")
    variant = key[len("efficientnet_"):]
    if flavour is None:
        flavour = "timm" if variant == "b0" else "lukemelas"
    return HipEfficientNet(variant, flavour, num_classes)


__all__ = ["HipEfficientNet", "HipMBConv", "build_efficientnet", "compute_dtype"]
