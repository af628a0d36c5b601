"""BASELINE config 2 at its own size (EfficientNet-B0, batch 256, 224 px, bf16) — the shapes the benchmark runs.

The kernel tests of tests/test_ops_gpu.py use shapes the CPU oracle finishes in milliseconds; the persistent-grid
kernels, their 20-bit magic divisions and `int` work counters are exercised here at M = 3,211,264 rows and
616 MB tensors (round-1 review, "What's weak" item 3):

  * block 1 of B0 (16 -> 96 -> 24 channels, 112 -> 56 px, k3 s2) kernel by kernel at N = 256 through the same
    reference functions and tolerances as test_ops_gpu.py;
  * one full bf16-autocast training step at N = 256 against the f32 oracle: logits, loss, BatchNorm running
    statistics of the stem and blocks 0-2 (f32 statistics of the large tensors);
  * a 20-step f32 loss curve against the oracle;
  * every MBConv block of B0 in bf16 against an emulation of the kernels' rounding points, two bf16 ulps.
"""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F

from oracle import ops_ref as R
from tests import test_ops_gpu as T

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
N_BASE = 256


def _hip():
    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    return HipEfficientNet, HipAdamW, HipCrossEntropyLoss


# ------------------------------------------------------------------ block 1 kernel by kernel at N = 256
@pytest.mark.parametrize("mode", [0, 3])
def test_expand_gemm_3_2_million_rows(mode):
    """16 -> 96 at 112x112x256: forward (mode 0, with BN statistics) and the data-gradient form (mode 3: affine2
    prologue + residual) of the wave-autonomous kernel."""
    T._pwconv_fwd_case((N_BASE, 112 * 112, 16, 96), mode, BF)


def test_expand_dgrad_shape_3_2_million_rows():
    T._pwconv_fwd_case((N_BASE, 112 * 112, 96, 16), 3, BF)


def test_project_gemm_with_gate_800k_rows():
    T._pwconv_fwd_case((N_BASE, 56 * 56, 96, 24), 2, BF)


def test_depthwise_forward_616_megabyte_input():
    T.test_dwconv_fwd((N_BASE, 112, 112, 96, 3, 2, 1, 1), BF)


def test_depthwise_backward_616_megabyte_input():
    T.test_dwconv_bwd((N_BASE, 112, 112, 96, 3, 2, 1, 1), BF)


@pytest.mark.parametrize("case", [(N_BASE, 112 * 112, 96, 16), (N_BASE, 56 * 56, 24, 96)])
def test_weight_gradients_at_baseline_rows(case):
    T.test_pwconv_wgrad_large_m(case)


def test_rowpasses_at_baseline_rows():
    T.test_rowpass((N_BASE, 112, 112, 96), BF)


# ------------------------------------------------------------------ the whole step at N = 256
def _pair(seed=5):
    from oracle.effnet_ref import EfficientNetRef

    Hip, _, _ = _hip()
    torch.manual_seed(seed)
    ref = EfficientNetRef("b0", "timm", 2)
    hip = Hip("b0", "timm", 2)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.cuda()


def rel_err(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)


def test_full_training_step_at_the_benchmark_configuration():
    _, _, HipCE = _hip()
    ref, hip = _pair()
    ref.train(); hip.train()
    g = torch.Generator().manual_seed(1)                                  # bench.py's generator seed
    x = torch.randn(N_BASE, 3, 224, 224, generator=g).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 2, (N_BASE,), generator=g)
    ref_logits = ref(x)
    ref_loss = F.cross_entropy(ref_logits, y, label_smoothing=0.1)
    with torch.autocast("cuda", dtype=BF):
        logits = hip(x.cuda(), [None] * len(hip.block_list()), None)
        loss = HipCE(0.1)(logits, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all()
    # bf16 yardstick: the oracle itself under CPU bf16 autocast against its f32 self.  ONE such run is one realisation of bf16
    # rounding noise amplified through 80 random-init layers; another summation order anywhere in the network is another realisation
    # of the same noise (VERDICT r3 "what's weak" 2), so the yardstick is drawn several times — two memory formats (different oneDNN
    # kernels) x two batch orders (different order of the BatchNorm sums) — and the bound is their maximum plus their spread (at
    # least 15 % of their mean: four samples under-estimate the spread of a maximum over 512 logits).
    import copy

    yards = []
    with torch.no_grad(), torch.autocast("cpu", dtype=BF):
        for fmt in (torch.channels_last, torch.contiguous_format):
            for flip in (False, True):
                xi = (x.flip(0) if flip else x).contiguous(memory_format=fmt)
                auto = copy.deepcopy(ref)(xi).float()
                yards.append(rel_err(auto.flip(0) if flip else auto, ref_logits))
    spread = max(max(yards) - min(yards), 0.15 * sum(yards) / len(yards))
    bound = max(yards) + spread
    err = rel_err(logits, ref_logits)
    print(f"logits rel err {err:.4f}; the oracle's own bf16 autocast, four realisations: {[round(v, 4) for v in yards]} -> bound {bound:.4f}; "
          f"loss {float(loss):.5f} vs {float(ref_loss):.5f}")
    assert err <= max(bound, 2e-2), (err, yards, bound)
    assert abs(float(loss) - float(ref_loss)) <= 2e-2 * max(1.0, abs(float(ref_loss)))
    # BatchNorm running statistics of the large early layers: f32 sums over 3.2 M bf16 values per channel
    rb, hb = dict(ref.named_buffers()), dict(hip.named_buffers())
    for name in ("bn1", "blocks.0.0.bn1", "blocks.0.0.bn2", "blocks.1.0.bn1", "blocks.1.0.bn2", "blocks.1.0.bn3", "blocks.1.1.bn1",
                 "blocks.1.1.bn2", "blocks.2.0.bn1"):
        for stat in ("running_mean", "running_var"):
            a, b = hb[f"{name}.{stat}"].float().cpu(), rb[f"{name}.{stat}"]
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-2 * scale + 1e-4, (name, stat, float((a - b).abs().max()), scale)
    # every parameter received a finite gradient through the persistent grids
    for name, p in hip.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    ga = torch.cat([p.grad.float().cpu().flatten() for _, p in hip.named_parameters()])
    ref_loss.backward()
    rp = dict(ref.named_parameters())
    gb = torch.cat([rp[n].grad.flatten() for n, _ in hip.named_parameters()])
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
    print(f"gradient cosine vs f32 oracle at N=256: {cos:.4f}")
    assert cos >= 0.98, cos


def test_twenty_step_loss_curve_f32():
    """Same seed, same inputs, same dropout uniforms, f32 on both sides: the loss curves must track each other
    (SURVEY.md section 7 step 7).  AdamW lr 1e-3 so that 20 steps move the loss visibly."""
    _, HipAdamW, HipCE = _hip()
    ref, hip = _pair(seed=9)
    ref.train(); hip.train()
    N, size = 8, 96
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(N, 3, size, size, generator=g) for _ in range(4)]
    ys = [torch.randint(0, 2, (N,), generator=g) for _ in range(4)]
    opt_r = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=5e-2)
    opt_h = HipAdamW(hip.parameters(), lr=1e-3, weight_decay=5e-2)
    crit = HipCE(0.1)
    p = ref.dropout
    curve = []
    for step in range(20):
        x, y = xs[step % 4], ys[step % 4]
        u = torch.rand((N, 1280), generator=g)
        opt_r.zero_grad(set_to_none=True)
        lr_ = F.cross_entropy(ref(x, None, (u >= p).float() / (1 - p)), y, label_smoothing=0.1)
        lr_.backward()
        opt_r.step()
        opt_h.zero_grad(set_to_none=True)
        lh = crit(hip(x.cuda(), [None] * len(hip.block_list()), u.cuda()), y.cuda())
        lh.backward()
        opt_h.step()
        curve.append((float(lr_), float(lh)))
    worst = max(abs(a - b) / max(1.0, abs(a)) for a, b in curve)
    print("loss curve (oracle, hip):", [(round(a, 4), round(b, 4)) for a, b in curve[::4]], "worst rel dev", worst)
    assert curve[-1][0] < curve[0][0]                      # it trained
    assert worst <= 5e-3, curve


# ------------------------------------------------------------------ per-block bf16 with the kernels' rounding points
def _block_ref_bf16(blk, x, eps):
    """MBConv forward (training-mode BN, no drop-connect) with a rounding to bf16 wherever the kernels store or
    stage a value: raw conv outputs, the activated tensor staged in LDS / loaded into MFMA fragments, the gated
    project operand, the block output.  Statistics are f32 sums over the ROUNDED raw tensors, as in the kernels."""
    rd = BF
    c = blk.c

    def st_of(y, bn):
        return R.bn_state(y, bn.weight.detach(), bn.bias.detach(), eps)

    N, H, W, _ = x.shape
    names = dict(blk.named_children())
    if c.expand != 1:
        w = blk.conv_pw.weight.detach().flatten(1)
        y1 = R.rnd(x @ R.rnd(w, rd).t(), rd)
        st1 = st_of(y1, blk.bn1)
        dw_bn, proj, proj_bn = blk.bn2, blk.conv_pwl, blk.bn3
        dw_in, dw_st, dw_act = y1, st1, R.ACT_SILU
    else:
        dw_bn, proj, proj_bn = blk.bn1, blk.conv_pw, blk.bn2
        dw_in, dw_st, dw_act = x, None, 0
    k, s = c.k, c.stride
    Ho = (H + 2 * (k // 2) - k) // s + 1
    y2 = R.dwconv_fwd(dw_in, dw_st, dw_act, blk.conv_dw.weight.detach(), k, s, k // 2, k // 2, Ho, Ho, rd)
    st2 = st_of(y2, dw_bn)
    a2 = R.rnd(R.act_fwd(st2[0] * y2 + st2[1], R.ACT_SILU), rd)
    pooled = a2.mean((1, 2))
    se = blk.se
    _, gate = R.se_fc(pooled, se.conv_reduce.weight.detach().flatten(1), se.conv_reduce.bias.detach(),
                      se.conv_expand.weight.detach().flatten(1), se.conv_expand.bias.detach(), R.ACT_SILU)
    A = R.rnd(a2 * gate[:, None, None, :], rd)
    y3 = R.rnd(A @ R.rnd(proj.weight.detach().flatten(1), rd).t(), rd)
    st3 = st_of(y3, proj_bn)
    out = st3[0] * y3 + st3[1]
    if c.stride == 1 and c.cin == c.cout:
        out = out + x
    assert "se" in names
    return R.rnd(out, rd)


@pytest.mark.parametrize("index", list(range(16)))
def test_every_b0_block_in_bf16_within_two_ulps_of_the_rounding_emulation(index):
    ref, hip = _pair(seed=21)
    rblk, hblk = ref.block_list()[index], hip.block_list()[index]
    c = rblk.c
    res = {0: 56, 1: 56, 2: 28, 3: 28, 4: 14, 5: 14, 6: 14, 7: 14, 8: 14, 9: 14, 10: 14, 11: 14, 12: 7, 13: 7, 14: 7, 15: 7}[index]
    g = torch.Generator().manual_seed(100 + index)
    x = torch.randn(8, res, res, c.cin, generator=g).to(BF)
    # non-trivial BN affine parameters on both sides
    with torch.no_grad():
        hp = dict(hblk.named_parameters())
        for n1, p1 in rblk.named_parameters():
            if ".bn" in n1 or n1.startswith("bn"):
                v = 0.6 + 0.8 * torch.rand(p1.shape, generator=g) if n1.endswith("weight") else torch.randn(p1.shape, generator=g) * 0.2
                p1.copy_(v); hp[n1].copy_(v.cuda())
    rblk.train(); hblk.train()
    want = _block_ref_bf16(rblk, x.float(), 1e-5)
    got = hblk.run(x.cuda(), None, None)
    assert got.dtype == BF
    scale = float(want.abs().max())
    err = float((got.float().cpu() - want).abs().max()) / scale
    assert err <= 1.6e-2, f"block {index}: {err:.4f} of max |out| {scale:.3f} (two bf16 ulps = 1.6e-2)"
