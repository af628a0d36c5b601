"""Forward-hook support on convolutions that normally live inside fused stages.

The reference's Grad-CAM (web_ui.py:95-114) picks `_conv_head`, or else the LAST nn.Conv2d of `model.modules()`, and
pytorch_grad_cam hangs a forward hook on it that keeps the output activation and registers a tensor hook for its gradient.
In the HIP modules those Conv2d objects are parameter holders inside fused autograd stages, so their `forward` never runs.
When (and only when) such a target module carries forward hooks in eval mode, its owner switches to an UNFUSED path built
from the pieces below, in which the convolution's output exists as an NCHW autograd tensor wired to the logits:

    HipEfficientNet          _conv_head / conv_head                      efficientnet._hooked_head (round 2)
    HipEfficientFormerV2     stages.3.blocks.<last>.mlp.fc2.conv         efficientformer_v2.HipConvMlp._hooked_run
    HipFasterViT             levels.2.global_tokenizer.pos_embed         fastervit.HipTokenInitializer._hooked_forward

The unfused paths are differentiable with respect to the hooked activation (what Grad-CAM needs); everything upstream of
the hooked convolution is computed without a graph.  Interactive batch-1 use: speed is irrelevant here, correctness is
tested against the oracle (tests/test_efformer_gpu.py, tests/test_fastervit_gpu.py).
"""

from __future__ import annotations

import torch

from . import kernels as K
from ._lib import ACT_NONE


def has_hooks(module: torch.nn.Module) -> bool:
    return bool(module._forward_hooks or module._forward_pre_hooks)


def call_hooks(module: torch.nn.Module, x_nhwc: torch.Tensor, y_nhwc: torch.Tensor) -> torch.Tensor:
    """Run `module`'s pre-hooks and hooks as nn.Module.__call__ would, on NCHW views; returns the (possibly replaced) output, NHWC."""
    x_nchw = x_nhwc.permute(0, 3, 1, 2)
    for hook in module._forward_pre_hooks.values():
        hook(module, (x_nchw,))
    y_nchw = y_nhwc.permute(0, 3, 1, 2)
    for hook in module._forward_hooks.values():
        r = hook(module, (x_nchw,), y_nchw)
        if r is not None:
            y_nchw = r
    out = y_nchw.permute(0, 2, 3, 1)
    return out if out.is_contiguous() else out.contiguous()


class ChannelAffineFunction(torch.autograd.Function):
    """out = scale[c] * y + shift[c] (+ residual); differentiable with respect to y only (dy = scale[c] * g)."""

    @staticmethod
    def forward(ctx, y, scale, shift, residual):
        C = y.shape[-1]
        st = torch.zeros((4, C), dtype=torch.float32, device=y.device)
        st[0].copy_(scale)
        st[1].copy_(shift)
        st[3].fill_(1.0)
        ctx.save_for_backward(y, st)
        return K.bn_act_apply(y, st, ACT_NONE, residual)

    @staticmethod
    def backward(ctx, g):
        y, st = ctx.saved_tensors
        coef = torch.zeros((3, y.shape[-1]), dtype=torch.float32, device=y.device)
        coef[0].copy_(st[0])
        g = g if g.is_contiguous() else g.contiguous()
        return K.affine2_apply(g, y, coef), None, None, None


class DwConvOutFunction(torch.autograd.Function):
    """depthwise 3x3 stride-1 convolution + bias as a stand-alone stage whose OUTPUT is the hooked activation.  No gradient
    flows further back (the input is computed without a graph); the weight only marks the output as requiring grad."""

    @staticmethod
    def forward(ctx, x, w, b):
        N, H, W, C = x.shape
        y, _, _ = K.dwconv_fwd(x, None, ACT_NONE, w, 3, 1, 1, 1, H, W, stats=False)
        return K.add_rowtable(y, b.view(1, C)) if b is not None else y

    @staticmethod
    def backward(ctx, g):
        return None, None, None


class AvgPoolFunction(torch.autograd.Function):
    """nn.AvgPool2d(kernel, stride) on NHWC, differentiable with respect to its input."""

    @staticmethod
    def forward(ctx, y, kernel: int, stride: int):
        ctx.geom = (tuple(y.shape), kernel, stride)
        return K.avgpool_fwd(y, kernel, stride)

    @staticmethod
    def backward(ctx, g):
        shape, kernel, stride = ctx.geom
        g = g if g.is_contiguous() else g.contiguous()
        return K.avgpool_bwd(g, shape, kernel, stride), None, None


__all__ = ["AvgPoolFunction", "ChannelAffineFunction", "DwConvOutFunction", "call_hooks", "has_hooks"]
