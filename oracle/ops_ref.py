"""TEST INFRASTRUCTURE — CPU oracle for the individual fused kernels.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Each function restates, with plain torch CPU ops in f32, what one entry point of
include/dfd_hip.h computes.  The third-party modules the reference calls
(efficientnet_pytorch 0.7.1 `MBConvBlock.forward`, timm 1.0.20 `InvertedResidual.forward`;
call sites trainers/efficientnet.py:297 and orchestration/orchestrator.py:590 of the
reference) are thin compositions of exactly these torch.nn.functional ops, so torch's
CPU kernels are the op-level oracle.

Tensors are NHWC ([N,H,W,C]) like the kernels'.  `rd` is the storage dtype of the
activations (torch.float32 or torch.bfloat16): values are rounded through it wherever
the kernel stores or would store them.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU = 0, 1, 2, 3


def rnd(t: torch.Tensor, rd: torch.dtype) -> torch.Tensor:
    return t.to(rd).to(torch.float32)


def act_fwd(z: torch.Tensor, act: int) -> torch.Tensor:
    if act == ACT_SILU:
        return z * torch.sigmoid(z)
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_GELU:
        return F.gelu(z)
    return z


def act_grad(z: torch.Tensor, act: int) -> torch.Tensor:
    if act == ACT_SILU:
        s = torch.sigmoid(z)
        return s * (1 + z * (1 - s))
    if act == ACT_RELU:
        return (z > 0).to(z.dtype)
    if act == ACT_GELU:
        cdf = 0.5 * (1 + torch.erf(z / 2 ** 0.5))
        pdf = torch.exp(-0.5 * z * z) / (2 * torch.pi) ** 0.5
        return cdf + z * pdf
    return torch.ones_like(z)


def bn_state(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """[4,C] = scale, shift, mean, rstd from batch statistics of y (f32 values)."""
    C = y.shape[-1]
    flat = y.reshape(-1, C).double()
    mean = flat.mean(0)
    var = flat.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + eps)
    scale = gamma.double() * rstd
    shift = beta.double() - mean * scale
    return torch.stack([scale, shift, mean, rstd]).float()


def stats_sums(y: torch.Tensor) -> torch.Tensor:
    C = y.shape[-1]
    flat = y.reshape(-1, C).double()
    return torch.stack([flat.sum(0), (flat * flat).sum(0)]).float()


def bn_bwd_coef(g: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, state: torch.Tensor):
    """coef [3,C] (dy = a*g + b*y + c), dgamma, dbeta for training-mode BN."""
    C = y.shape[-1]
    gf, yf = g.reshape(-1, C).double(), y.reshape(-1, C).double()
    mean, rstd = state[2].double(), state[3].double()
    xhat = (yf - mean) * rstd
    s1, s2 = gf.sum(0), (gf * xhat).sum(0)
    M = gf.shape[0]
    gm = gamma.double()
    a = gm * rstd
    b = -gm * rstd * rstd * s2 / M
    c = -gm * rstd * s1 / M + gm * rstd * rstd * mean * s2 / M
    return torch.stack([a, b, c]).float(), s2.float(), s1.float()


def _same_pad(H: int, Ho: int, k: int, s: int, p0: int) -> int:
    return max((Ho - 1) * s + k - p0 - H, 0)


def dwconv_fwd(x, state, act, w, k, s, pt, pl, Ho, Wo, rd):
    """y = dwconv(rnd(act(scale*x+shift))) with weights rounded through rd."""
    a = x if state is None else rnd(act_fwd(state[0] * x + state[1], act), rd)
    N, H, W, C = a.shape
    pb, pr = _same_pad(H, Ho, k, s, pt), _same_pad(W, Wo, k, s, pl)
    an = F.pad(a.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(an, rnd(w, rd), stride=s, groups=C)[:, :, :Ho, :Wo]
    return rnd(y.permute(0, 2, 3, 1).contiguous(), rd)


def dwconv_bwd(dy, xact, w, k, s, pt, pl, rd):
    """(d xact, dw) of y = dwconv(xact); dy, xact f32 NHWC."""
    xa = xact.clone().requires_grad_(True)
    wr = rnd(w, rd).clone().requires_grad_(True)
    N, H, W, C = xa.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    pb, pr = _same_pad(H, Ho, k, s, pt), _same_pad(W, Wo, k, s, pl)
    an = F.pad(xa.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(an, wr, stride=s, groups=C)[:, :, :Ho, :Wo].permute(0, 2, 3, 1)
    y.backward(dy)
    return xa.grad, wr.grad


def stem_conv_fwd(x, w, s, pt, pl, Ho, Wo, rd):
    N, H, W, _ = x.shape
    k = w.shape[2]
    pb, pr = _same_pad(H, Ho, k, s, pt), _same_pad(W, Wo, k, s, pl)
    xn = F.pad(rnd(x, rd).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, rnd(w, rd), stride=s)[:, :, :Ho, :Wo]
    return rnd(y.permute(0, 2, 3, 1).contiguous(), rd)


def stem_conv_wgrad(x, dy, k, s, pt, pl, rd):
    Cout = dy.shape[3]
    w = torch.zeros(Cout, 3, k, k, requires_grad=True)
    N, H, W, _ = x.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    pb, pr = _same_pad(H, Ho, k, s, pt), _same_pad(W, Wo, k, s, pl)
    xn = F.pad(rnd(x, rd).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w, stride=s)[:, :, :Ho, :Wo].permute(0, 2, 3, 1)
    y.backward(dy)
    return w.grad


def prologue(a, mode, rd, act=ACT_NONE, coef=None, a2=None, gate=None):
    """A operand after the GEMM prologue; a is [N,HW,K] (f32 values)."""
    if mode == 0:
        return a
    if mode == 3:
        return rnd(coef[0] * a + coef[1] * a2 + coef[2], rd)
    v = act_fwd(coef[0] * a + coef[1], act)
    if mode == 2:
        v = rnd(v, rd) * gate[:, None, :]
    return rnd(v, rd)


def se_fc(pooled, w1, b1, w2, b2, act):
    hpre = pooled @ w1.t() + b1
    gate = torch.sigmoid(act_fwd(hpre, act) @ w2.t() + b2)
    return hpre, gate


def ce_label_smooth(logits, targets, eps):
    return F.cross_entropy(logits, targets, label_smoothing=eps)


# ---------------------------------------------------------------------------------------------------------------------
# OCP microscaling (MX) FP8 — the format of the FasterViT fp8-weight path (include/dfd_hip.h "MX fp8"; BASELINE config 5).
# Restates the OCP MX v1.0 conversion rule for element type e4m3 (emax_elem = 8) and block size 32:
#   X = 2^(floor(log2(max|v|)) - 8) clamped to the E8M0 range, q = RNE_saturate_e4m3fn(v / X).
# torch.float8_e4m3fn is OCP e4m3fn (not the MI300X fnuz variant); its f32 conversion rounds to nearest even.
def mx_quant(v: torch.Tensor):
    """v [..., K] f32 (K % 32 == 0) -> (q uint8 [..., K] e4m3fn bytes, scale uint8 [..., K/32] e8m0 bytes)."""
    shape = v.shape
    blocks = v.detach().to(torch.float32).reshape(-1, 32)
    amax = blocks.abs().amax(dim=1)
    _, ex = torch.frexp(amax)                                   # amax = m * 2^ex, m in [0.5, 1): floor(log2 amax) = ex - 1
    e = (ex.to(torch.int32) - 1 - 8).clamp(-127, 127)
    e = torch.where(amax > 0, e, torch.full_like(e, -127))
    inv = torch.ldexp(torch.ones_like(amax), -e)                # 2^-e (e = -127 only for all-zero blocks: 0 * 2^127 = 0)
    scaled = (blocks * inv[:, None]).clamp(-448.0, 448.0)
    q = scaled.to(torch.float8_e4m3fn).view(torch.uint8)
    sb = (e + 127).to(torch.uint8)
    # a block with a NaN or an infinity is a NaN block: the OCP MX NaN scale 0xff, every element the e4m3fn NaN 0x7f
    bad = ~torch.isfinite(blocks).all(dim=1)
    q = torch.where(bad[:, None], torch.full_like(q, 0x7F), q)
    sb = torch.where(bad, torch.full_like(sb, 0xFF), sb)
    return q.reshape(shape), sb.reshape(*shape[:-1], shape[-1] // 32)


def mx_dequant(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """(q, scale) of mx_quant -> f32 values (exact)."""
    vals = q.contiguous().view(torch.float8_e4m3fn).to(torch.float32).reshape(-1, 32)
    sc = torch.ldexp(torch.ones(vals.shape[0]), scale.reshape(-1).to(torch.int32) - 127)
    sc = torch.where(scale.reshape(-1) == 0xFF, torch.full_like(sc, float("nan")), sc)
    return (vals * sc[:, None]).reshape(q.shape)


def mx_linear(a: torch.Tensor, w: torch.Tensor, quantise_a: bool = True) -> torch.Tensor:
    """deq(mx(a)) @ deq(mx(w))^T in f32 — what dfd_mx_quant_rows + dfd_mx_gemm compute (f32 accumulation; the products of
    two e4m3 values are exact in f32).  quantise_a=False: weights-only quantisation (the looser, model-level yardstick)."""
    wd = mx_dequant(*mx_quant(w))
    ad = mx_dequant(*mx_quant(a)) if quantise_a else a.to(torch.float32)
    return ad @ wd.t()
