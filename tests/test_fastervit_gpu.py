"""FasterViT on the HIP kernels against the CPU oracle (oracle/fastervit_ref.py).

Stage level: ConvBlock, Downsample, the carrier-token initialiser and HAT blocks with and without carrier tokens
against the oracle's modules on the same weights and inputs (outputs, input gradients, every parameter gradient).
Network level: f32 eval logits rel <= 1e-3 with identical arg-max, f32 train step (loss, all gradients, BN running
statistics), bf16 autocast step + bitwise reproducibility.  Stochastic depth: the oracle and the engine draw from
different generators, so the network-level comparisons build both with drop_path_rate 0 and the DropPath arithmetic
(per-sample scale on the window and carrier streams) is checked on a block with injected masks.
"""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _imports():
    from deepfakedetection_amd.fastervit import HipFasterViT
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
    from oracle.fastervit_ref import FasterViTRef

    return HipFasterViT, FasterViTRef, HipAdamW, HipCrossEntropyLoss


def rel_err(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)


def randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            last = name.rsplit(".", 1)[-1]
            if "norm" in name or ".conv_down.1" in name or ".conv_down.4" in name:
                p.copy_(0.6 + 0.8 * torch.rand(p.shape, generator=g) if last == "weight" else torch.randn(p.shape, generator=g) * 0.2)
            elif last == "bias":
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif "cpb_mlp" in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.shape[-1] == 2 else 0.05))
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))


def make_pair(variant="0", nc=2, seed=0, dpr=0.0):
    Hip, Ref, _, _ = _imports()
    torch.manual_seed(seed)
    ref = Ref(variant, nc, 224, drop_path_rate=dpr)
    randomise(ref, seed + 1)
    hip = Hip(variant, nc, 224, drop_path_rate=dpr)
    hip.load_state_dict(ref.state_dict(), strict=True)
    return ref, hip.cuda()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def check_param_grads(ref_mod, hip_mod, tol, train=True):
    hp = dict(hip_mod.named_parameters())
    rp = dict(ref_mod.named_parameters())
    worst = ("", 0.0)
    for name, p in rp.items():
        if p.grad is None:
            continue
        got = hp[name].grad
        assert got is not None, f"no gradient for {name}"
        scale = max(float(p.grad.abs().max()), 1e-9)
        err = float((got.float().cpu() - p.grad).abs().max())
        sib = name[:-4] + "weight"
        if name.endswith("bias") and sib in rp and rp[sib].grad is not None:
            scale = max(scale, float(rp[sib].grad.abs().max()))     # structurally-zero bias gradients hold noise on both sides
        if err / scale > worst[1]:
            worst = (name, err / scale)
        assert err / scale <= tol, f"gradient of {name}: rel {err / scale:.3e} > {tol:.1e} (|ref| {scale:.3e})"
    return worst


@pytest.mark.parametrize("train", [True, False])
def test_conv_block_and_downsample(train):
    ref, hip = make_pair()
    rb, hb = ref.levels[0].blocks[1], hip.levels[0].blocks[1]
    rd_, hd = ref.levels[0].downsample, hip.levels[0].downsample
    for m in (rb, hb, rd_, hd):
        m.train(train)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 64, 56, 56, generator=g).requires_grad_()
    mid = rb(x)
    out = rd_(mid)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    xh = nhwc(x.detach()).cuda().requires_grad_()
    oh = hd(hb(xh))
    assert rel_err(oh.permute(0, 3, 1, 2), out) <= 2e-4
    oh.backward(nhwc(gout).cuda())
    assert rel_err(xh.grad.permute(0, 3, 1, 2), x.grad) <= 1e-3
    check_param_grads(rb, hb, 2e-3, train)
    check_param_grads(rd_, hd, 2e-3, train)


def test_token_initializer():
    ref, hip = make_pair()
    rt, ht = ref.levels[2].global_tokenizer, hip.levels[2].global_tokenizer
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 256, 14, 14, generator=g).requires_grad_()
    ct = rt(x)                                                  # [2, 16, 256], window-major order
    from oracle.fastervit_ref import ct_dewindow

    want = ct_dewindow(ct, 4, 4, 2)                             # row-major, the order the engine keeps carrier tokens in
    gout = torch.randn(want.shape, generator=g)
    want.backward(gout)
    xh = nhwc(x.detach()).cuda().requires_grad_()
    got = ht(xh)
    assert rel_err(got.view(2, 16, 256), want) <= 2e-4
    got.backward(gout.view(2, 16, 1, 256).cuda())
    assert rel_err(xh.grad.permute(0, 3, 1, 2), x.grad) <= 1e-3
    check_param_grads(rt, ht, 2e-3)


@pytest.mark.parametrize("level,with_masks", [(2, False), (2, True), (3, False)])
def test_hat_block(level, with_masks):
    """level 2: carrier tokens attend globally, join their windows (sequence 53) and are split off again;
    level 3: one plain 49-token window.  with_masks: DropPath scales on both streams."""
    from oracle.fastervit_ref import ct_dewindow, ct_window
    from deepfakedetection_amd.fastervit import _window_maps

    ref, hip = make_pair(dpr=0.3 if with_masks else 0.0)
    rb, hb = ref.levels[level].blocks[1], hip.levels[level].blocks[1]
    rb.train(); hb.train()
    dim = 256 if level == 2 else 512
    B = 3
    nW = B * (4 if level == 2 else 1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(nW, 49, dim, generator=g).requires_grad_()
    ct = torch.randn(B, 16, dim, generator=g).requires_grad_() if level == 2 else None      # row-major carrier grid
    masks = None
    if with_masks:
        keep = 1.0 - rb.dp
        masks = (torch.floor(keep + torch.rand(B, generator=g)) / keep, torch.floor(keep + torch.rand(nW, generator=g)) / keep)
    ct_in = ct_window(ct, 4, 4, 2).reshape(B, 16, dim) if ct is not None else None            # the package's window-major storage
    ox, oc = rb(x, ct_in, masks)
    gx = torch.randn(ox.shape, generator=g)
    loss = (ox * gx).sum()
    if oc is not None:
        oc_rm = ct_dewindow(oc, 4, 4, 2)
        gc = torch.randn(oc_rm.shape, generator=g)
        loss = loss + (oc_rm * gc).sum()
    loss.backward()
    xh = x.detach().view(nW, 49, 1, dim).cuda().requires_grad_()
    cth = ct.detach().view(B, 16, 1, dim).cuda().requires_grad_() if ct is not None else None
    maps = _window_maps(B, 14, torch.device("cuda"))[1:] if level == 2 else None

    class FakeRng:
        def drop_path_scale(self, n, keep, stream_id):
            return (masks[1] if n == nW and stream_id % 4 == 0 else masks[0]).cuda()

    hx, hc = hb(xh, cth, maps, FakeRng() if with_masks else None)
    assert rel_err(hx.view(nW, 49, dim), ox) <= 3e-4
    if hc is not None:
        assert rel_err(hc.view(B, 16, dim), oc_rm) <= 3e-4
        torch.autograd.backward([hx, hc], [gx.view(nW, 49, 1, dim).cuda(), gc.view(B, 16, 1, dim).cuda()])
        assert rel_err(cth.grad.view(B, 16, dim), ct.grad) <= 2e-3
    else:
        hx.backward(gx.view(nW, 49, 1, dim).cuda())
    assert rel_err(xh.grad.view(nW, 49, dim), x.grad) <= 2e-3
    print("worst gradient:", check_param_grads(rb, hb, 3e-3))


def calibrated_pair(variant="0", nc=2, n=4):
    ref, hip = make_pair(variant, nc)
    g = torch.Generator().manual_seed(1)
    return ref, hip, torch.randn(n, 3, 224, 224, generator=g), torch.randint(0, nc, (n,), generator=g)


def test_eval_logits_f32():
    ref, hip, x, _ = calibrated_pair("0", 10, n=4)
    ref.eval(); hip.eval()
    with torch.no_grad():
        want = ref(x)
    with torch.inference_mode():
        got = hip(x.cuda())
    assert rel_err(got, want) <= 1e-3, rel_err(got, want)
    assert torch.equal(got.argmax(1).cpu(), want.argmax(1))
    assert float((want - want.mean(0, keepdim=True)).abs().max()) > 1e-3


def test_train_step_f32_all_parameters():
    _, _, HipAdamW, HipCE = _imports()
    ref, hip, x, y = calibrated_pair("0", 2, n=4)
    ref.train(); hip.train()
    loss_ref = F.cross_entropy(ref(x), y, label_smoothing=0.1)
    loss_ref.backward()
    opt = HipAdamW(hip.parameters(), lr=1e-4, weight_decay=5e-2)
    loss = HipCE(0.1)(hip(x.cuda()), y.cuda())
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) <= 1e-4 * max(1.0, abs(float(loss_ref)))
    print("worst gradient:", check_param_grads(ref, hip, 5e-3))
    hbuf = dict(hip.named_buffers())
    for n1, b1 in ref.named_buffers():
        if "running" in n1:
            assert rel_err(hbuf[n1], b1) <= 2e-4, n1
    assert opt.arena.holds_all_grads()
    opt.step()
    torch.cuda.synchronize()


def test_bf16_autocast_step_with_drop_path_and_reproducibility():
    Hip, _, _, HipCE = _imports()
    torch.manual_seed(0)
    hip = Hip("0", 2).cuda().train()                     # the package's drop_path_rate 0.2: the engine's own Philox masks
    g = torch.Generator().manual_seed(2)
    x, y = torch.randn(8, 3, 224, 224, generator=g).cuda(), torch.randint(0, 2, (8,), generator=g).cuda()
    losses = []
    for _ in range(2):
        hip.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = HipCE(0.1)(hip(x), y)
        loss.backward()
        losses.append(float(loss))
    assert all(torch.isfinite(p.grad).all() for p in hip.parameters())
    assert losses[0] != losses[1]                        # the Philox offset advanced: different DropPath masks
    hip2 = Hip("0", 2, drop_path_rate=0.0).cuda().train()
    outs = []
    for _ in range(2):
        hip2.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = hip2(x)
            HipCE(0.1)(logits, y).backward()
        outs.append((logits.detach().clone(), hip2.levels[2].blocks[0].attn.qkv.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_refuses_cpu():
    Hip, _, _, _ = _imports()
    with pytest.raises(RuntimeError, match="HIP device"):
        Hip("0", 2)(torch.zeros(1, 3, 224, 224))


@pytest.mark.gpu
def test_batched_eval_coefficients_replay_equals_per_layer_path():
    """functions.bn_eval_batch: the first pass records the eval-mode coefficient requests (per-layer kernels), later
    passes compute them with one batched launch; results are those of the per-layer path, also after biases, LayerScale-free
    Linear parameters and running statistics changed in place, in eval and in training mode."""
    _, hip = make_pair("0", 2, seed=3)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(4)).cuda()
    hip.eval()
    with torch.inference_mode():
        first, replay, plain = hip(x), hip(x), hip._forward(x)
        assert torch.equal(first, replay) and torch.equal(first, plain)
        batch = hip.__dict__["_bn_eval_batches"][(False, torch.float32)]
        assert batch.keys is not None and len(batch.keys) > 40
    with torch.no_grad():
        for name, p in hip.named_parameters():
            if name.endswith("bias"):
                p.add_(0.05)
        for name, b in hip.named_buffers():
            if name.endswith("running_var"):
                b.mul_(1.3)
    with torch.inference_mode():
        again, plain2 = hip(x), hip._forward(x)
        assert torch.equal(again, plain2) and not torch.equal(again, first)
    hip.train()
    torch.manual_seed(0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = hip(x)                          # records the training pass's requests (the Linear layers' identity statistics)
        b = hip(x)                          # replays them
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert (True, torch.bfloat16) in hip.__dict__["_bn_eval_batches"]


def test_grad_cam_hooks_on_the_last_conv2d():
    """web_ui.py:95-114 targets the last nn.Conv2d of model.modules(): for FasterViT that is the depthwise
    `levels.2.global_tokenizer.pos_embed` (level 3 has no convolution).  With forward hooks on it the tokenizer runs
    unfused in eval mode; activation (NCHW, bias included) and its gradient with respect to a logit match the oracle's."""
    ref, hip = make_pair("0", 2, seed=6)
    ref.eval(); hip.eval()
    last = lambda m: [c for c in m.modules() if isinstance(c, torch.nn.Conv2d)][-1]     # noqa: E731
    names = {id(c): n for n, c in hip.named_modules()}
    assert names[id(last(hip))].endswith("global_tokenizer.pos_embed") or names[id(last(hip))].endswith("to_global_feature.pos")
    kept = {}

    def make_hook(tag):
        def hook(module, inputs, output):
            kept[tag + "_act"] = output
            if output.requires_grad:
                output.register_hook(lambda grad: kept.__setitem__(tag + "_grad", grad))
        return hook

    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(13))
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
    assert tuple(kept["hip_act"].shape) == tuple(kept["ref_act"].shape) == (2, 256, 14, 14)
    assert rel_err(kept["hip_act"], kept["ref_act"]) <= 1e-3
    assert rel_err(kept["hip_grad"], kept["ref_grad"]) <= 2e-3
    with torch.no_grad():
        assert rel_err(hip(x.cuda()), ref_logits) <= 1e-3
