"""EfficientFormerV2 on the HIP kernels against the CPU oracle (oracle/efformer_ref.py).

Stage level: each fused autograd function (ConvMlp block, attention block with and without the stride-2 /
upsample path, plain and attention downsample, stem, tail) against the oracle's sub-module on the same
weights and inputs — outputs, input gradient and every parameter gradient.
Network level: f32 eval logits rel <= 1e-3 with identical arg-max (the north-star bar), f32 train step (loss,
logits, all parameter gradients, BN running statistics), the reference's truncated fine-tune backward
(UNFREEZE_KEYS, trainers/efficientformer_v2.py:66-74,389-393), bf16 autocast step, bitwise reproducibility.
Tolerances: f32 2e-4 .. 5e-3 of the tensor's max magnitude as stated per assert (summation order differs;
gradients pass through up to 30 BatchNorm layers), bf16 compared by cosine.
"""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _imports():
    from deepfakedetection_amd.efficientformer_v2 import HipEfficientFormerV2
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
    from oracle.efformer_ref import EfficientFormerV2Ref

    return HipEfficientFormerV2, EfficientFormerV2Ref, HipAdamW, HipCrossEntropyLoss


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)


def randomise(model: torch.nn.Module, seed: int) -> None:
    """Non-trivial values everywhere: BN affine / running statistics, LayerScale O(0.3) (timm's 1e-5 init would hide
    the attention and MLP branches from every comparison), attention bias tables, talking heads."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("gamma"):
                p.copy_(0.2 + 0.3 * torch.rand(p.shape, generator=g))
            elif "attention_biases" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif "bn" in name or name.startswith("norm"):
                p.copy_(0.6 + 0.8 * torch.rand(p.shape, generator=g) if name.endswith("weight") else torch.randn(p.shape, generator=g) * 0.2)
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))


def make_pair(variant="s1", nc=2, img=224, seed=0):
    Hip, Ref, _, _ = _imports()
    torch.manual_seed(seed)
    ref = Ref(variant, nc, img)
    randomise(ref, seed + 1)
    hip = Hip(variant, nc, img)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref, hip.cuda()


def nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.permute(0, 2, 3, 1).contiguous()


def check_param_grads(ref_mod, hip_mod, tol, skip=(), train=True):
    worst = ("", 0.0)
    hp = dict(hip_mod.named_parameters())
    rp = dict(ref_mod.named_parameters())
    for name, p in ref_mod.named_parameters():
        if any(s in name for s in skip):
            continue
        if p.grad is None:
            continue
        got = hp[name].grad
        assert got is not None, f"no gradient for {name}"
        scale = max(float(p.grad.abs().max()), 1e-9)
        err = float((got.float().cpu() - p.grad).abs().max())
        # a conv bias in front of a training-mode BatchNorm has an exactly-zero true gradient (the batch mean
        # absorbs it): the oracle holds float cancellation noise there, the engine exact zeros
        if train and name.endswith(".conv.bias"):
            assert float(got.abs().max()) == 0.0, name
            wscale = float(rp[name[:-4] + "weight"].grad.abs().max())
            assert scale <= 2e-3 * max(wscale, 1e-6), f"{name}: oracle bias gradient {scale:.3e} is not noise (weight grad {wscale:.3e})"
            continue
        # other structurally-zero gradients (a BN bias in front of another batch-statistics BN, anything that shifts
        # all keys of a softmax row equally: k's bias terms, talking_head1.bias) hold cancellation noise on BOTH
        # sides: a bias is judged on the scale of its module's joint (weight, bias) gradient
        if name.endswith("bias") and name[:-4] + "weight" in rp and rp[name[:-4] + "weight"].grad is not None:
            scale = max(scale, float(rp[name[:-4] + "weight"].grad.abs().max()))
        if err / scale > worst[1]:
            worst = (name, err / scale)
        assert err / scale <= tol, f"gradient of {name}: rel {err / scale:.3e} > {tol:.1e} (|ref| {scale:.3e})"
    return worst


# ------------------------------------------------------------------ stage level
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("stage,block", [(0, 0), (2, 3), (2, 7), (3, 5)])
def test_block_matches_oracle(stage, block, train):
    """(0,0): ConvMlp at 56x56; (2,3): ConvMlp with 3x expansion; (2,7): stride-2 attention + upsample; (3,5): 7x7 attention."""
    ref, hip = make_pair()
    rb, hb = ref.stages[stage].blocks[block], hip.stages[stage].blocks[block]
    rb.train(train); hb.train(train)
    dim = (32, 48, 120, 224)[stage]
    res = (56, 28, 14, 7)[stage]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, dim, res, res, generator=g).requires_grad_()
    gout = torch.randn(4, dim, res, res, generator=g)
    out = rb(x)
    out.backward(gout)
    xh = nhwc(x.detach()).cuda().requires_grad_()
    oh = hb(xh)
    assert rel_err(oh.permute(0, 3, 1, 2), out) <= 2e-4
    oh.backward(nhwc(gout).cuda())
    assert rel_err(xh.grad.permute(0, 3, 1, 2), x.grad) <= 1e-3
    check_param_grads(rb, hb, 2e-3, train=train)
    if train:
        hbuf = dict(hb.named_buffers())
        for n1, b1 in rb.named_buffers():
            if "running" in n1:
                assert rel_err(hbuf[n1], b1) <= 1e-4, n1


@pytest.mark.parametrize("stage", [1, 3])
def test_downsample_matches_oracle(stage):
    """stage 1: conv3x3 s2 + BN; stage 3: the same plus the Attention2dDownsample branch (196 keys, 49 queries)."""
    ref, hip = make_pair()
    rd, hd = ref.stages[stage].downsample, hip.stages[stage].downsample
    rd.train(); hd.train()
    dim = (32, 48, 120)[stage - 1]
    res = (56, 28, 14)[stage - 1]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, dim, res, res, generator=g).requires_grad_()
    out = rd(x)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    xh = nhwc(x.detach()).cuda().requires_grad_()
    oh = hd(xh)
    assert rel_err(oh.permute(0, 3, 1, 2), out) <= 2e-4
    oh.backward(nhwc(gout).cuda())
    assert rel_err(xh.grad.permute(0, 3, 1, 2), x.grad) <= 1e-3
    check_param_grads(rd, hd, 2e-3)


# ------------------------------------------------------------------ network level
def calibrated_pair(variant="s1", nc=2, img=224, n=8):
    ref, hip = make_pair(variant, nc, img)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, 3, img, img, generator=g)
    y = torch.randint(0, nc, (n,), generator=g)
    return ref, hip, x, y


@pytest.mark.parametrize("variant,img", [("s1", 224), ("s0", 160)])
def test_eval_logits_f32(variant, img):
    ref, hip, x, _ = calibrated_pair(variant, 10, img)
    ref.eval(); hip.eval()
    with torch.no_grad():
        want = ref(x)
    with torch.inference_mode():
        got = hip(x.cuda())
    assert rel_err(got, want) <= 1e-3
    assert torch.equal(got.argmax(1).cpu(), want.argmax(1))
    assert float((want - want.mean(0, keepdim=True)).abs().max()) > 1e-3        # logits differ between images


def test_train_step_f32_all_parameters():
    _, _, _, HipCE = _imports()
    ref, hip, x, y = calibrated_pair("s1", 2, 224)
    ref.train(); hip.train()
    loss_ref = F.cross_entropy(ref(x), y, label_smoothing=0.1)
    loss_ref.backward()
    logits = hip(x.cuda())
    loss = HipCE(0.1)(logits, y.cuda())
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) <= 1e-4 * max(1.0, abs(float(loss_ref)))
    worst = check_param_grads(ref, hip, 5e-3)
    print("worst gradient:", worst)
    hbuf = dict(hip.named_buffers())
    for n1, b1 in ref.named_buffers():
        if "running" in n1:
            assert rel_err(hbuf[n1], b1) <= 2e-4, n1
        if n1.endswith("num_batches_tracked"):
            assert int(hbuf[n1]) == int(b1) == 1, n1


def test_truncated_backward_like_the_reference_fine_tune():
    """trainers/efficientformer_v2.py:389-393 trains only names containing UNFREEZE_KEYS: backward stops at the
    earliest such parameter (stages.2.blocks.3) and frozen parameters get no gradient."""
    _, _, HipAdamW, HipCE = _imports()
    keys = ("stages.3", "blocks.3", "layer4", "bneck", "features.6", "classifier", "head")
    ref, hip, x, y = calibrated_pair("s1", 2, 224, n=4)
    for m in (ref, hip):
        for name, p in m.named_parameters():
            p.requires_grad = any(k in name for k in keys)
    ref.train(); hip.train()
    F.cross_entropy(ref(x), y, label_smoothing=0.1).backward()
    opt = HipAdamW([p for p in hip.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-2)
    HipCE(0.1)(hip(x.cuda()), y.cuda()).backward()
    n_train = 0
    hp = dict(hip.named_parameters())
    for name, p in ref.named_parameters():
        if p.requires_grad:
            n_train += 1
            assert hp[name].grad is not None, name
        else:
            assert hp[name].grad is None, name
    assert n_train == 182
    check_param_grads(ref, hip, 5e-3)
    assert opt.arena.holds_all_grads()
    opt.step()
    torch.cuda.synchronize()


def test_bf16_autocast_step_and_reproducibility():
    _, _, _, HipCE = _imports()
    ref, hip, x, y = calibrated_pair("s1", 2, 224, n=8)
    ref.train(); hip.train()
    F.cross_entropy(ref(x), y, label_smoothing=0.1).backward()
    outs = []
    for _ in range(2):
        hip.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = hip(x.cuda())
            loss = HipCE(0.1)(logits, y.cuda())
        loss.backward()
        outs.append((logits.detach().clone(), hip.stages[3].blocks[5].mlp.fc1.conv.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # no float atomics anywhere
    assert torch.isfinite(loss).item()
    # direction of the large late-layer gradients against the f32 oracle
    for name in ("stages.3.blocks.5.mlp.fc1.conv.weight", "stages.3.blocks.4.token_mixer.q.conv.weight", "head.weight"):
        a = dict(hip.named_parameters())[name].grad.float().cpu().flatten()
        b = dict(ref.named_parameters())[name].grad.flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        assert cos >= 0.97, f"{name}: cosine {cos:.4f}"


def test_refuses_cpu_and_wrong_resolution():
    Hip, _, _, _ = _imports()
    m = Hip("s1", 2, 224)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.zeros(1, 3, 224, 224))
    with pytest.raises(ValueError, match="224x224"):
        m.cuda()(torch.zeros(1, 3, 192, 192).cuda())


def test_grad_cam_hooks_on_the_last_conv2d():
    """web_ui.py:95-114: without a `_conv_head` the Grad-CAM target is the LAST nn.Conv2d of model.modules() — for
    EfficientFormerV2 `stages.3.blocks.<last>.mlp.fc2.conv`, a convolution in the middle of a fused stage.  With forward
    hooks on it the HIP module runs that MLP unfused (eval mode): the hook sees the conv output (NCHW, bias included) and
    its gradient with respect to a class logit; both must match the oracle's."""
    import torch.nn.functional as F

    ref, hip = make_pair("s1", 2, 224, seed=4)
    ref.eval(); hip.eval()
    last = lambda m: [c for c in m.modules() if isinstance(c, torch.nn.Conv2d)][-1]     # noqa: E731 - web_ui._find_last_conv_layer
    names = {id(c): n for n, c in hip.named_modules()}
    assert names[id(last(hip))] == "stages.3.blocks.5.mlp.fc2.conv"
    # the oracle's ConvBN calls F.conv2d on the weights: route this one through the module so that its hooks fire
    fc2 = ref.stages[3].blocks[-1].mlp.fc2

    def via_module(x):
        y = fc2.conv(x)
        return F.batch_norm(y, fc2.bn.running_mean, fc2.bn.running_var, fc2.bn.weight, fc2.bn.bias, False, fc2.bn.momentum, fc2.bn.eps)

    fc2.forward = via_module
    kept = {}

    def make_hook(tag):
        def hook(module, inputs, output):
            kept[tag + "_in"] = inputs[0]
            kept[tag + "_act"] = output
            if output.requires_grad:
                output.register_hook(lambda grad: kept.__setitem__(tag + "_grad", grad))
        return hook

    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(12))
    h1 = last(ref).register_forward_hook(make_hook("ref"))
    h2 = last(hip).register_forward_hook(make_hook("hip"))
    try:
        ref_logits = ref(x)
        ref_logits[:, 1].sum().backward()
        hip_logits = hip(x.cuda())
        hip_logits[:, 1].sum().backward()
    finally:
        h1.remove(); h2.remove()
    assert rel_err(hip_logits, ref_logits) <= 1e-3
    assert tuple(kept["hip_act"].shape) == tuple(kept["ref_act"].shape) == (2, 224, 7, 7)
    assert rel_err(kept["hip_in"], kept["ref_in"]) <= 1e-3
    assert rel_err(kept["hip_act"], kept["ref_act"]) <= 1e-3
    assert rel_err(kept["hip_grad"], kept["ref_grad"]) <= 2e-3
    with torch.no_grad():
        assert rel_err(hip(x.cuda()), ref_logits) <= 1e-3                 # hooks gone: the fused stage is back
    hip.train()
    h3 = last(hip).register_forward_hook(make_hook("t"))
    try:
        with pytest.raises(NotImplementedError):
            hip(x.cuda())
    finally:
        h3.remove()
