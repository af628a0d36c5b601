"""The ctx -> output cycle guard (tests/_autograd_cycles.py): the detector itself on the CPU, and every custom Function of the three
model families on the GPU."""

from __future__ import annotations

import pytest
import torch

from tests._autograd_cycles import custom_functions, no_cycle_collector, survivors, track


class _Good(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x * 2
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        return g * 2


class _KeepsItsOutput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x * 2
        ctx.out = y                         # the bug: output -> grad_fn -> ctx -> output
        return y

    @staticmethod
    def backward(ctx, g):
        return g * 2


def _run(cls):
    with no_cycle_collector(), track([cls]) as refs:
        x = torch.ones(4, requires_grad=True)
        y = cls.apply(x)
        y.sum().backward()
        del y
        return survivors(refs)


def test_detector_flags_a_function_that_keeps_its_output():
    assert _run(_Good) == []
    assert _run(_KeepsItsOutput) == ["_KeepsItsOutput"]
    import gc

    gc.collect()                            # (clean up the cycle the bad example made on purpose)


def _families():
    from deepfakedetection_amd.efficientformer_v2 import HipEfficientFormerV2
    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.fastervit import HipFasterViT

    return {
        "efficientnet": lambda: HipEfficientNet("b0", "timm", 2),
        "efficientformerv2": lambda: HipEfficientFormerV2("s0", 2, 224),
        "fastervit": lambda: HipFasterViT("0", 2, 224),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["efficientnet", "efficientformerv2", "fastervit"])
@pytest.mark.parametrize("mode", ["train_bf16", "eval_hooked_f32"])
def test_no_custom_function_keeps_its_own_output(family, mode):
    from deepfakedetection_amd import fastervit, fastervit_functions, functions, hooks, vit_functions
    from deepfakedetection_amd.optim import HipCrossEntropyLoss

    classes = custom_functions(functions, vit_functions, fastervit_functions, fastervit, hooks)
    assert len(classes) >= 20
    torch.manual_seed(0)
    model = _families()[family]().cuda()
    x = torch.randn(4, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 2, (4,), device="cuda")
    crit = HipCrossEntropyLoss(label_smoothing=0.1)
    seen: list = []
    handles = []
    if mode == "train_bf16":
        model.train()
    else:
        # eval mode with forward hooks on the Grad-CAM target (hooks.py: the unfused, differentiable eval path of web_ui.py:95-114)
        model.eval()
        target = getattr(model, "_conv_head", None) or [m for m in model.modules() if isinstance(m, torch.nn.Conv2d)][-1]
        handles.append(target.register_forward_hook(lambda mod, inp, out: seen.append(out.shape)))
    with no_cycle_collector(), track(classes) as refs:
        for _ in range(2):                  # twice: a node pinned by the first iteration would still be there after the second
            if mode == "train_bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = crit(model(x), y)
            else:
                loss = model(x)[:, 0].sum()
            loss.backward()
            del loss
            model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        ran = sorted({name for name, _ in refs})
        alive = survivors(refs)
    for h in handles:
        h.remove()
    assert ran, "no custom Function ran"
    assert alive == [], f"outputs of {alive} outlive their iteration: a ctx -> output reference cycle (keep None on ctx, not the output)"
    if mode != "train_bf16":
        assert seen, "the hooked module did not run"
