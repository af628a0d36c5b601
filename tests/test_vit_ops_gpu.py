"""Kernel-level parity of the ABI-110 set (token mixers, extended BatchNorm, bookkeeping) against torch CPU ops.

Same yardsticks as tests/test_ops_gpu.py: f32 kernels rel 2e-4 of the tensor's max magnitude, bf16 kernels
1.6e-2 (two bf16 ulps), stated per assert.  The references are the torch.nn.functional ops that timm 1.0.20
`efficientformer_v2.py` / fastervit 1.0.0 `faster_vit.py` compose (nn.Upsample(bilinear), softmax, 1x1 convs over
heads, F.layer_norm, F.conv2d, F.unfold).
"""

from __future__ import annotations

import math

import pytest
import torch
import torch.nn.functional as F

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

DT = [torch.float32, torch.bfloat16]


def _k():
    from deepfakedetection_amd import kernels

    return kernels


def tol(rd):
    return 2e-4 if rd == torch.float32 else 1.6e-2


def close(got, want, rel, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = max(float(want.abs().max()), 1e-6)
    err = float((got - want).abs().max()) / scale
    assert err <= rel, f"{what}: max err {err:.3e} of max |ref| {scale:.3e} > {rel:.1e}"


def gen(shape, seed, rd=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(rd)


def rand_state(C, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([0.5 + torch.rand(C, generator=g), torch.randn(C, generator=g) * 0.3,
                        torch.randn(C, generator=g) * 0.2, 0.5 + torch.rand(C, generator=g)])


# ------------------------------------------------------------------ BatchNorm with conv bias + LayerScale
def test_bn_finalize_ex_train_and_eval():
    K = _k()
    C, M = 48, 600
    y = gen((M, C), 1) * 1.5 + 0.3
    gamma, beta, cb, ls = 0.5 + torch.rand(C), gen((C,), 2), gen((C,), 3), gen((C,), 4, scale=0.1)
    parts = torch.stack([y.sum(0), (y * y).sum(0)]).reshape(1, 2, C)
    rm, rv = gen((C,), 5) * 0.1, 0.5 + torch.rand(C)
    bn = K.BNParams(gamma.cuda(), beta.cuda(), rm.clone().cuda(), rv.clone().cuda(), 0.1, 1e-5, cb.cuda(), ls.cuda())
    st = K.bn_finalize(parts.cuda().contiguous(), 1, M, bn).cpu()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    close(st[0], ls * gamma * rstd, 1e-5, "scale")
    close(st[1], ls * (beta - mean * gamma * rstd), 1e-4, "shift")
    close(st[2], mean, 1e-5, "mean")
    # running mean sees the biased convolution output, as F.batch_norm(conv(x) + b) would
    close(bn.running_mean, 0.9 * rm + 0.1 * (mean + cb), 1e-5, "running_mean")
    close(bn.running_var, 0.9 * rv + 0.1 * y.var(0, unbiased=True), 1e-5, "running_var")
    # eval: ls * BN(y + b) with running statistics
    bn2 = K.BNParams(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 0.1, 1e-5, cb.cuda(), ls.cuda())
    st2 = K.bn_eval_coeffs(bn2).cpu()
    want = ls * F.batch_norm(y + cb, rm, rv, gamma, beta, False, 0.1, 1e-5)
    close(st2[0] * y + st2[1], want, 1e-5, "eval affine")


@pytest.mark.parametrize("train", [True, False])
def test_bn_bwd_finalize_ex_matches_autograd(train):
    K = _k()
    C, M = 32, 400
    y = (gen((M, C), 1) * 1.3 + 0.2).requires_grad_()
    gamma, beta = (0.5 + torch.rand(C)).requires_grad_(), gen((C,), 2).requires_grad_()
    cb, ls = gen((C,), 3).requires_grad_(), gen((C,), 4, scale=0.3).requires_grad_()
    rm, rv = gen((C,), 5) * 0.1, 0.5 + torch.rand(C)
    out = ls * F.batch_norm(y + cb, rm.clone(), rv.clone(), gamma, beta, train, 0.1, 1e-5)
    g = gen((M, C), 6)
    out.backward(g)
    with torch.no_grad():
        if train:
            mean, var = y.mean(0), y.var(0, unbiased=False)
        else:
            mean, var = rm - cb, rv
        rstd = 1 / torch.sqrt(var + 1e-5)
        xhat = (y - mean) * rstd
        parts = torch.stack([g.sum(0), (g * xhat).sum(0)]).reshape(1, 2, C).contiguous()
        st = torch.stack([ls * gamma * rstd, ls * (beta - mean * gamma * rstd), mean, rstd]).contiguous()
    coef, dg, db, dls, dbias = K.bn_bwd_finalize_ex(parts.cuda(), 1, M, gamma.detach().cuda(), beta.detach().cuda(),
                                                    ls.detach().cuda(), st.cuda(), train, True, True, True)
    close(dg, gamma.grad, 2e-4, "dgamma")
    close(db, beta.grad, 2e-4, "dbeta")
    close(dls, ls.grad, 2e-4, "dls")
    if train:
        assert float(dbias.abs().max()) == 0.0          # exactly zero through batch statistics
        assert float(cb.grad.abs().max()) < 1e-3
    else:
        close(dbias, cb.grad, 2e-4, "dbias")
    dy = K.affine2_apply(g.cuda(), y.detach().cuda().contiguous(), coef)
    close(dy, y.grad, 5e-4, "dy = a*g + b*y + c")


# ------------------------------------------------------------------ elementwise passes
@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("act", [R.ACT_NONE, R.ACT_GELU, R.ACT_SILU])
def test_bn_add_act_fwd_bwd(rd, act):
    K = _k()
    shape = (3, 7, 7, 40)
    y, other, g = gen(shape, 1, rd), gen(shape, 2, rd), gen(shape, 3, rd)
    st = rand_state(40, 4)
    out = K.bn_add_act(y.cuda(), st.cuda(), other.cuda(), act)
    pre = st[0] * y.float() + st[1] + other.float()
    close(out, R.act_fwd(pre, act), tol(rd), "bn_add_act")
    d, parts, n = K.bn_add_act_bwd(g.cuda(), y.cuda(), st.cuda(), other.cuda(), act)
    want = R.rnd(g.float() * R.act_grad(pre, act), rd)
    close(d, want, tol(rd), "bn_add_act_bwd d")
    xhat = (y.float() - st[2]) * st[3]
    got = parts[: n * 2 * 40].view(n, 2, 40).double().sum(0).float().cpu()
    close(got[0], want.reshape(-1, 40).sum(0), 5e-3 if rd == torch.bfloat16 else 5e-4, "sum d")
    close(got[1], (want * xhat).reshape(-1, 40).sum(0), 5e-3 if rd == torch.bfloat16 else 5e-4, "sum d*xhat")
    # without BN and without addend
    out2 = K.bn_add_act(y.cuda(), None, None, act)
    close(out2, R.act_fwd(y.float(), act), tol(rd), "plain act")


@pytest.mark.parametrize("rd", DT)
def test_channel_stats_and_sum_rows(rd):
    K = _k()
    x = gen((5, 14, 14, 224), 1, rd)
    parts, n = K.channel_stats(x.cuda())
    both = torch.empty((2, 224), device="cuda")
    K.sum_rows(parts, n, 2 * 224, both.view(-1))
    flat = x.float().reshape(-1, 224)
    close(both[0], flat.sum(0), 1e-4, "sum")
    close(both[1], (flat * flat).sum(0), 1e-4, "sumsq")


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("act", [R.ACT_NONE, R.ACT_GELU])
@pytest.mark.parametrize("hw", [(7, 7), (3, 5)])
def test_up2_matches_nn_upsample_bilinear(rd, act, hw):
    K = _k()
    h, w = hw
    s = gen((2, h, w, 16), 1, rd)
    g = gen((2, 2 * h, 2 * w, 16), 2, rd)
    sn = s.float().permute(0, 3, 1, 2).clone().requires_grad_()
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear")(sn)          # what timm's Attention2d holds
    ref = R.act_fwd(up, act)
    ref.backward(g.float().permute(0, 3, 1, 2))
    out = K.up2_act_fwd(s.cuda(), act)
    close(out, ref.permute(0, 2, 3, 1), tol(rd), "up2 fwd")
    ds = K.up2_act_bwd(g.cuda(), s.cuda(), act)
    close(ds, sn.grad.permute(0, 2, 3, 1), tol(rd), "up2 bwd")


@pytest.mark.parametrize("rd", DT)
def test_subsample_add(rd):
    K = _k()
    x = gen((2, 14, 14, 24), 1, rd)
    a = gen((2, 7, 7, 24), 2, rd)
    bias = gen((24,), 3)
    out = K.subsample_add(a.cuda(), bias.cuda(), x.cuda(), 2)
    close(out, a.float() + bias + x.float()[:, ::2, ::2], tol(rd), "local + bias + pool")      # nn.AvgPool2d(1, 2, 0)
    dx = gen((2, 14, 14, 24), 4, rd)
    want = dx.float().clone()
    want[:, ::2, ::2] += a.float()
    got = K.subsample_add_bwd(a.cuda(), dx.cuda(), 2)
    close(got, want, tol(rd), "subsample bwd")


# ------------------------------------------------------------------ batched GEMM and attention rows
@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                      (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)])
def test_bgemm_reads_heads_in_place_from_nhwc(din, dout):
    """q [B*N][H*dk] and k [B*N][H*dk] as the 1x1 projections leave them -> S[b,h] = scale * q_h k_h^T + bias[h]."""
    K = _k()
    B, H, N, dk = 3, 8, 49, 32
    q, k = gen((B * N, H * dk), 1, din), gen((B * N, H * dk), 2, din)
    bias = gen((H, N, N), 3)
    S = torch.empty((B, H, N, N), dtype=dout, device="cuda")
    K.bgemm(q.cuda(), (N * H * dk, dk, H * dk, 1), k.cuda(), (N * H * dk, dk, 1, H * dk), S, (H * N * N, N * N, N, 1), B, H, N, N, dk,
            alpha=dk ** -0.5, bias=bias.cuda())
    qh = q.float().view(B, N, H, dk).permute(0, 2, 1, 3)
    kh = k.float().view(B, N, H, dk).permute(0, 2, 1, 3)
    want = qh @ kh.transpose(-1, -2) * dk ** -0.5 + bias
    close(S, want, 1e-4 if dout == torch.float32 else 1.6e-2, "S = QK^T*scale + bias")
    # P.V written straight into NHWC [B*N][H*dv]
    dv = 16
    P = torch.softmax(want, -1)
    v = gen((B * N, H * dv), 4, din)
    O = torch.empty((B * N, H * dv), dtype=dout, device="cuda")
    K.bgemm(P.cuda().contiguous(), (H * N * N, N * N, N, 1), v.cuda(), (N * H * dv, dv, H * dv, 1), O, (N * H * dv, dv, H * dv, 1),
            B, H, N, dv, N)
    vh = v.float().view(B, N, H, dv).permute(0, 2, 1, 3)
    close(O, (P @ vh).permute(0, 2, 1, 3).reshape(B * N, H * dv), 1e-4 if dout == torch.float32 else 1.6e-2, "O = PV")
    # transposed A through strides: dV = P^T dO
    dO = gen((B * N, H * dv), 5, din)
    dV = torch.empty((B * N, H * dv), dtype=dout, device="cuda")
    K.bgemm(P.cuda().contiguous(), (H * N * N, N * N, 1, N), dO.cuda(), (N * H * dv, dv, H * dv, 1), dV, (N * H * dv, dv, H * dv, 1),
            B, H, N, dv, N)
    dOh = dO.float().view(B, N, H, dv).permute(0, 2, 1, 3)
    close(dV, (P.transpose(-1, -2) @ dOh).permute(0, 2, 1, 3).reshape(B * N, H * dv), 1e-4 if dout == torch.float32 else 1.6e-2, "dV")


@pytest.mark.parametrize("shape", [(3, 8, 49, 49, 32, 128), (2, 8, 49, 37, 32, 64), (5, 3, 16, 64, 64, 32), (1, 8, 64, 1, 96, 96),
                                   (3, 8, 49, 196, 16, 64), (2, 5, 130, 65, 8, 24), (1, 2, 256, 256, 48, 120)])
def test_attention_products_on_the_matrix_cores(shape):
    """dfd_attn_scores / dfd_attn_apply (one wave per (image, head), csrc/dfd_attn.hip) against the f32 products of the same bf16
    inputs — the six GEMMs around the talking-head softmax of timm's Attention2d.  The f32 operand of dfd_attn_apply is rounded to
    bf16 for the product: the bound is that rounding (2^-9 relative per term), as for P and dS in the window attention."""
    K = _k()
    B, H, Nq, Nk, dk, dv = shape
    assert K.attn_mfma_supported(torch.bfloat16, Nq, Nk, dk, dv)
    bf = torch.bfloat16
    q, k, v = gen((B, Nq, 1, H * dk), 1, bf), gen((B, Nk, 1, H * dk), 2, bf), gen((B, Nk, 1, H * dv), 3, bf)
    bias = gen((H, Nq * Nk), 4)
    heads = lambda t, T, D: t.float().view(B, T, H, D).permute(0, 2, 1, 3)
    back = lambda t, T, D: t.permute(0, 2, 1, 3).reshape(B, T, 1, H * D)
    qh, kh, vh = heads(q, Nq, dk), heads(k, Nk, dk), heads(v, Nk, dv)
    S = K.attn_scores(q.cuda(), k.cuda(), H, dk ** -0.5, bias.cuda())
    want_S = qh @ kh.transpose(-1, -2) * dk ** -0.5 + bias.view(H, Nq, Nk)
    close(S, want_S, 1e-5, "S = scale q k^T + bias")
    T2 = torch.softmax(want_S, -1) + 0.1 * gen((B, H, Nq, Nk), 5)
    O = K.attn_apply(T2.cuda(), v.cuda(), (B, Nq, 1, H * dv), H)
    close(O, back(T2 @ vh, Nq, dv), 1.2e-2, "O = T2 v")
    gO = gen((B, Nq, 1, H * dv), 6, bf)
    gOh = heads(gO, Nq, dv)
    close(K.attn_scores(gO.cuda(), v.cuda(), H), gOh @ vh.transpose(-1, -2), 1e-5, "dT2 = dO v^T")
    close(K.attn_apply(T2.cuda(), gO.cuda(), (B, Nk, 1, H * dv), H, transpose=True), back(T2.transpose(-1, -2) @ gOh, Nk, dv), 1.2e-2, "dV = T2^T dO")
    dS = gen((B, H, Nq, Nk), 7)
    close(K.attn_apply(dS.cuda(), k.cuda(), (B, Nq, 1, H * dk), H, alpha=0.25), back(0.25 * (dS @ kh), Nq, dk), 1.2e-2, "dQ = scale dS k")
    close(K.attn_apply(dS.cuda(), q.cuda(), (B, Nk, 1, H * dk), H, alpha=0.25, transpose=True), back(0.25 * (dS.transpose(-1, -2) @ qh), Nk, dk), 1.2e-2,
          "dK = scale dS^T q")
    # the same calls give the same bits
    assert torch.equal(K.attn_scores(q.cuda(), k.cuda(), H, dk ** -0.5, bias.cuda()), S)
    assert torch.equal(K.attn_apply(T2.cuda(), v.cuda(), (B, Nq, 1, H * dv), H), O)


def test_bgemm_large_k_and_rect():
    K = _k()
    B, H, Nq, Nk, dk = 2, 8, 49, 196, 16
    q, k = gen((B, H, Nq, dk), 1), gen((B, H, Nk, dk), 2)
    S = torch.empty((B, H, Nq, Nk), device="cuda")
    K.bgemm(q.cuda(), (H * Nq * dk, Nq * dk, dk, 1), k.cuda(), (H * Nk * dk, Nk * dk, 1, dk), S, (H * Nq * Nk, Nq * Nk, Nk, 1), B, H, Nq, Nk, dk)
    close(S, q @ k.transpose(-1, -2), 1e-4, "49x196")
    v = gen((B, H, Nk, 64), 3)
    O = torch.empty((B, H, Nq, 64), device="cuda")
    K.bgemm(S, (H * Nq * Nk, Nq * Nk, Nk, 1), v.cuda(), (H * Nk * 64, Nk * 64, 64, 1), O, (H * Nq * 64, Nq * 64, 64, 1), B, H, Nq, 64, Nk)
    close(O, (q @ k.transpose(-1, -2)) @ v, 1e-4, "K = 196")


@pytest.mark.parametrize("talk", [True, False])
@pytest.mark.parametrize("shape", [(3, 8, 49, 49), (2, 8, 49, 196), (5, 4, 53, 53), (2, 16, 49, 49)])
def test_attn_softmax_rows_match_autograd(talk, shape):
    K = _k()
    B, H, Nq, Nk = shape
    S = (gen(shape, 1) * 2).requires_grad_()
    if talk:
        w1, b1 = (gen((H, H), 2) * 0.4).requires_grad_(), gen((H,), 3).requires_grad_()
        w2, b2 = (gen((H, H), 4) * 0.4).requires_grad_(), gen((H,), 5).requires_grad_()
        T1 = F.conv2d(S, w1.view(H, H, 1, 1), b1)                 # timm Attention2d.talking_head1
        Pr = T1.softmax(-1)
        T2 = F.conv2d(Pr, w2.view(H, H, 1, 1), b2)                # talking_head2
        th = tuple(t.detach().cuda() for t in (w1, b1, w2, b2))
    else:
        Pr = S.softmax(-1)
        T2, th = Pr, None
    g = gen(shape, 6)
    T2.backward(g)
    P, T2g = K.attn_softmax_fwd(S.detach().cuda(), th)
    close(P, Pr, 2e-5, "P")
    close(T2g, T2, 2e-5, "T2")
    dT1, dS = K.attn_softmax_bwd(g.cuda(), P, th)
    close(dS, S.grad, 2e-4, "dS")
    if talk:
        # the talking-head weight gradients are contractions of (dT2, P) and (dT1, S) over batch and positions
        want_w2 = torch.einsum("bgij,bhij->gh", g, Pr.detach())
        close(torch.einsum("bgij,bhij->gh", g.cuda(), P), want_w2, 2e-4, "dW2 ingredients")
        close(w2.grad, want_w2, 2e-4, "dW2 definition")
        close(torch.einsum("bgij,bhij->gh", dT1, S.detach().cuda()), w1.grad, 5e-4, "dW1 from dT1")
        assert float(b1.grad.abs().max()) < 1e-4                 # softmax is shift-invariant: db1 == 0


def test_bias_gather_scatter():
    K = _k()
    H, res = 8, 7
    N = res * res
    pos = torch.stack(torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")).flatten(1)
    rel = (pos[..., :, None] - pos[..., None, :]).abs()
    idx = (rel[0] * res + rel[1]).to(torch.int32)                 # timm Attention2d.attention_bias_idxs
    table = gen((H, N), 1)
    full = K.bias_gather(table.cuda(), idx.cuda().contiguous().view(-1))
    close(full.view(H, N, N), table[:, idx.long()], 0.0, "gather")
    dfull = gen((H, N * N), 2)
    want = torch.zeros(H, N).index_add_(1, idx.view(-1).long(), dfull)
    close(K.bias_scatter(dfull.cuda(), idx.cuda().contiguous().view(-1), N), want, 1e-5, "scatter")


# ------------------------------------------------------------------ dense convolution pieces
@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("cfg", [(3, 2, 1, 16, 12), (3, 1, 1, 24, 9), (3, 2, 1, 48, 7)])
def test_im2col_col2im(rd, cfg):
    K = _k()
    k, s, p, C, H = cfg
    Ho = (H + 2 * p - k) // s + 1
    x = gen((2, H, H, C), 1, rd)
    st = rand_state(C, 2)
    col = K.im2col(x.cuda(), st.cuda(), R.ACT_GELU, k, s, p, Ho, Ho)
    a = R.rnd(F.gelu(st[0] * x.float() + st[1]), rd)
    unf = F.unfold(a.permute(0, 3, 1, 2), k, padding=p, stride=s)                  # [N, C*k*k, L] with (c, kh, kw) order
    want = unf.view(2, C, k * k, Ho, Ho).permute(0, 3, 4, 2, 1).reshape(2, Ho, Ho, k * k * C)
    close(col, want, tol(rd), "im2col")
    col_plain = K.im2col(x.cuda(), None, R.ACT_NONE, k, s, p, Ho, Ho)
    unf = F.unfold(x.float().permute(0, 3, 1, 2), k, padding=p, stride=s)
    close(col_plain, unf.view(2, C, k * k, Ho, Ho).permute(0, 3, 4, 2, 1).reshape(2, Ho, Ho, k * k * C), 0.0, "im2col plain")
    dcol = gen((2, Ho, Ho, k * k * C), 3, rd)
    dx = K.col2im(dcol.cuda(), (2, H, H, C), k, s, p)
    back = dcol.float().view(2, Ho * Ho, k * k, C).permute(0, 3, 2, 1).reshape(2, C * k * k, Ho * Ho)
    want_dx = F.fold(back, (H, H), k, padding=p, stride=s).permute(0, 2, 3, 1)
    close(dx, want_dx, tol(rd), "col2im")


def test_conv_weight_perm_roundtrip_and_dense_conv_through_gemm():
    K = _k()
    O, I, k = 32, 16, 3
    w = gen((O, I, k, k), 1)
    wg = K.conv_weight_to_gemm(w.cuda())
    close(wg, w.permute(0, 2, 3, 1).reshape(O, k * k * I), 0.0, "to gemm")
    close(K.conv_wgrad_from_gemm(wg, (O, I, k, k)), w, 0.0, "back")
    x = gen((2, 12, 12, I), 2)
    col = K.im2col(x.cuda(), None, R.ACT_NONE, k, 2, 1, 6, 6)
    y, _, _ = K.pwconv(col, None, wg, None, stats=False)
    want = F.conv2d(x.permute(0, 3, 1, 2), w, stride=2, padding=1).permute(0, 2, 3, 1)
    close(y, want, 2e-4, "conv3x3 s2 = im2col + 1x1 GEMM")


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("C", [64, 256, 512, 1024, 2048])      # 1, 1, 1 | 2, 2 | 0 (streaming) vectors per lane
def test_layernorm_fwd_bwd(rd, C):
    K = _k()
    x = (gen((37, C), 1, rd).float() * 1.5 + 0.2).to(rd)
    gamma, beta = 0.5 + torch.rand(C), gen((C,), 2)
    xr = x.float().clone().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    g = gen((37, C), 3, rd)
    ref.backward(g.float())
    y, stats = K.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    close(y, ref, tol(rd), "ln fwd")
    dx, dg, db = K.layernorm_bwd(g.cuda(), x.cuda(), gamma.cuda(), stats)
    close(dx, xr.grad, tol(rd), "ln dx")
    close(dg, gr.grad, 2e-4 if rd == torch.float32 else 5e-3, "ln dgamma")
    close(db, br.grad, 2e-4 if rd == torch.float32 else 5e-3, "ln dbeta")
    # residual fused into the kernel (the skip connection's gradient) and (dgamma, dbeta) summed straight into a [2, C]
    # destination: same values as the separate add / the temporary
    res = gen((37, C), 4, rd)
    out2 = torch.empty((2, C), dtype=torch.float32, device="cuda")
    dx2, dg2, db2 = K.layernorm_bwd(g.cuda(), x.cuda(), gamma.cuda(), stats, res.cuda(), out2)
    assert torch.equal(dx2, K.add(dx, res.cuda())), "fused residual differs from dx + residual"
    assert torch.equal(dg2, dg) and torch.equal(db2, db) and dg2.data_ptr() == out2.data_ptr()


# ------------------------------------------------------------------ bookkeeping
def test_axpby_add():
    K = _k()
    x, y = gen((1000,), 1), gen((1000,), 2)
    s = torch.tensor([0.25], device="cuda")
    close(K.axpby(x.cuda(), y.cuda(), 2.0, -1.0, a_dev=s), 0.5 * x - y, 1e-6, "axpby")
    close(K.axpby(x.cuda(), None, 3.0, 0.0), 3 * x, 1e-6, "scale")
    for rd in DT:
        a, b = gen((4, 6, 6, 16), 3, rd), gen((4, 6, 6, 16), 4, rd)
        close(K.add(a.cuda(), b.cuda()), R.rnd(a.float() + b.float(), rd), tol(rd), "add")


def test_philox_rand_and_tick():
    K = _k()
    rng = K.DeviceRng(torch.device("cuda"), seed=1234)
    u1 = rng.uniform(100_000, 7)
    u1b = rng.uniform(100_000, 7)
    assert torch.equal(u1, u1b)                                  # pure function of (seed, offset, stream, index)
    u2 = rng.uniform(100_000, 8)
    assert not torch.equal(u1, u2)
    assert 0.0 <= float(u1.min()) and float(u1.max()) < 1.0
    assert abs(float(u1.mean()) - 0.5) < 5e-3 and abs(float(u1.var()) - 1 / 12) < 2e-3
    counters = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(5)]
    rng.tick(counters)
    rng.tick(counters)
    assert all(int(c) == 2 for c in counters)
    assert int(rng.state[1]) == 2
    u3 = rng.uniform(100_000, 7)
    assert not torch.equal(u1, u3)                               # the offset advanced
    keep = 0.8
    rs = rng.drop_path_scale(50_000, keep, 3)
    vals = torch.unique(rs).tolist()
    assert all(v == 0.0 or abs(v - 1 / keep) < 1e-6 for v in vals)
    assert abs(float((rs > 0).float().mean()) - keep) < 1e-2
    # same seed -> same stream on another state object
    rng2 = K.DeviceRng(torch.device("cuda"), seed=1234)
    assert torch.equal(rng2.uniform(1000, 7), u1[:1000])


# ------------------------------------------------------------------ GELU prologues of the existing kernels
@pytest.mark.parametrize("rd", DT)
def test_pwconv_gelu_prologue_and_wgrad(rd):
    K = _k()
    M, Kd, Nout = 2 * 14 * 14, 120, 480
    a = gen((2, 14, 14, Kd), 1, rd)
    st = rand_state(Kd, 2)
    w = gen((Nout, Kd), 3) * 0.1
    w_nk, _ = K.prep_weights(w.cuda().view(Nout, Kd, 1, 1), rd, True, False)
    out, parts, n = K.pwconv(a.cuda(), K.pro_bn_act(st.cuda(), R.ACT_GELU), w_nk, None, stats=True)
    act = R.rnd(F.gelu(st[0] * a.float() + st[1]), rd)
    want = act.reshape(M, Kd) @ R.rnd(w, rd).t()
    close(out.reshape(M, Nout), want, tol(rd), "pwconv GELU prologue")
    g = gen((2, 14, 14, Nout), 4, rd)
    dw = K.pwconv_wgrad(g.cuda(), None, a.cuda(), K.pro_bn_act(st.cuda(), R.ACT_GELU))
    close(dw, g.float().reshape(M, Nout).t() @ act.reshape(M, Kd), 5e-3 if rd == torch.bfloat16 else 5e-4, "wgrad GELU prologue")


@pytest.mark.parametrize("rd", DT)
def test_dwconv_gelu_prologue(rd):
    K = _k()
    C = 128
    x = gen((2, 14, 14, C), 1, rd)
    st = rand_state(C, 2)
    w = gen((C, 1, 3, 3), 3) * 0.3
    y, _, _ = K.dwconv_fwd(x.cuda(), st.cuda(), R.ACT_GELU, w.cuda(), 3, 1, 1, 1, 14, 14, stats=False)
    act = R.rnd(F.gelu(st[0] * x.float() + st[1]), rd)
    want = F.conv2d(act.permute(0, 3, 1, 2), R.rnd(w, rd), padding=1, groups=C).permute(0, 2, 3, 1)
    close(y, want, tol(rd), "dwconv GELU prologue")
    assert math.isfinite(float(y.float().abs().max()))


# ------------------------------------------------------------------ FasterViT token bookkeeping
@pytest.mark.parametrize("rd", DT)
def test_copy_rows_partition_and_inverse(rd):
    K = _k()
    from deepfakedetection_amd.fastervit import _window_maps
    from oracle.fastervit_ref import window_partition, window_reverse

    B, res, C = 3, 14, 24
    x = gen((B, res, res, C), 1, rd)
    part, src_ct, dst_ct, dst_x = _window_maps(B, res, torch.device("cuda"))
    out = torch.empty((B * res * res, C), dtype=rd, device="cuda")
    K.copy_rows(x.cuda().view(-1, C), part, out, None, B * res * res)
    want = window_partition(x.float().permute(0, 3, 1, 2), 7)                        # [B*4, 49, C]
    close(out.view(B * 4, 49, C), want, 0.0, "window_partition")
    back = torch.empty_like(out)
    K.copy_rows(out, None, back, part, B * res * res)
    close(back.view(B, res, res, C), window_reverse(want, 7, res, res).permute(0, 2, 3, 1), 0.0, "window_reverse")
    # carrier tokens (row-major 4x4 grid per image) in front of each window's 49 tokens
    ct = gen((B, 16, C), 2, rd)
    cat = torch.zeros((B * 4, 53, C), dtype=rd, device="cuda")
    K.copy_rows(ct.cuda().view(-1, C), src_ct, cat.view(-1, C), dst_ct, B * 16)
    K.copy_rows(out, None, cat.view(-1, C), dst_x, B * 196)
    from oracle.fastervit_ref import ct_window

    ctw = ct_window(ct.float(), 4, 4, 2).reshape(B * 4, 4, C)                         # the package's ct_window on row-major tokens
    close(cat[:, :4], ctw, 0.0, "carrier tokens per window")
    close(cat[:, 4:], want, 0.0, "window tokens")


@pytest.mark.parametrize("rd", DT)
def test_rowtable_avgpool_relpos(rd):
    K = _k()
    x = gen((6, 49, 1, 64), 1, rd)
    tab = gen((49, 64), 2)
    close(K.add_rowtable(x.cuda(), tab.cuda()), R.rnd(x.float() + tab[None, :, None, :], rd), tol(rd), "add_rowtable")
    close(K.rowtable_grad(x.cuda(), 49), x.float().sum(0).view(49, 64), 1e-3 if rd == torch.bfloat16 else 1e-5, "rowtable_grad")
    y = gen((2, 14, 14, 32), 3, rd)
    yn = y.float().permute(0, 3, 1, 2).clone().requires_grad_()
    ref = F.avg_pool2d(yn, 5, 3)
    g = gen((2, 4, 4, 32), 4, rd)
    ref.backward(g.float().permute(0, 3, 1, 2))
    close(K.avgpool_fwd(y.cuda(), 5, 3), ref.permute(0, 2, 3, 1), tol(rd), "avgpool fwd")
    close(K.avgpool_bwd(g.cuda(), (2, 14, 14, 32), 5, 3), yn.grad.permute(0, 2, 3, 1), tol(rd), "avgpool bwd")
    if rd == torch.float32:
        from oracle.fastervit_ref import PosEmb2D

        torch.manual_seed(0)
        pe = PosEmb2D(7, 8, 53)
        want = pe.bias(53)[0]                                                         # [8, 53, 53]
        tabl = pe.cpb_mlp(pe.relative_coords_table).view(-1, 8).detach()
        idx = pe.relative_position_index.reshape(-1).to(torch.int32)
        got = K.relpos_bias_fwd(tabl.cuda().contiguous(), idx.cuda(), 49, 4)
        close(got, want, 1e-5, "relpos bias")
        assert float(got[:, :4].abs().max()) == 0.0 and float(got[:, :, :4].abs().max()) == 0.0
        t2 = tabl.clone().requires_grad_()
        b = 16 * torch.sigmoid(t2[pe.relative_position_index.view(-1)].view(49, 49, 8).permute(2, 0, 1))
        dfull = gen((8, 53, 53), 5)
        b.backward(dfull[:, 4:, 4:])
        close(K.relpos_bias_bwd(dfull.cuda(), tabl.cuda().contiguous(), idx.cuda(), 49, 4), t2.grad, 1e-4, "relpos bias bwd")


def test_sum_rows_beyond_1024_partial_rows():
    K = _k()
    P, L = 2500, 96
    parts = torch.randn(P + 100, L, generator=torch.Generator().manual_seed(1))
    out = torch.empty(L, device="cuda")
    K.sum_rows(parts.cuda().view(-1), P, L, out)
    close(out, parts[:P].double().sum(0).float(), 1e-5, "multi-group row sum")


@pytest.mark.parametrize("P", [1, 31, 32, 33, 36, 64, 65, 100, 128, 129, 200, 256, 257, 700])
def test_sum_rows_has_one_order_whatever_kernel_adds_them(P):
    """The fixed order of every partial-row sum: rows in groups of 32, each group added in row order from 0.f, then the group sums in group
    order from 0.f — by one launch (P <= 32), by the one-pass kernel (33..256: both stages in one launch), by the two-launch form (above),
    and by the batched form inside sum_batch().  All of them must give the bits of that order (restated here in torch f32), and accumulate
    on top of what the destination holds."""
    K = _k()
    L = 1003
    parts = torch.randn(P + (P + 31) // 32 + 1, L, generator=torch.Generator().manual_seed(P))
    groups = []
    for p0 in range(0, P, 32):
        g = torch.zeros(L)
        for r in range(p0, min(p0 + 32, P)):
            g = g + parts[r]
        groups.append(g)
    if len(groups) == 1:
        want = groups[0]
    else:
        want = torch.zeros(L)
        for g in groups:
            want = want + g
    out = torch.empty(L, device="cuda")
    K.sum_rows(parts.cuda().view(-1), P, L, out)
    assert torch.equal(out.cpu(), want), f"immediate sum of {P} rows differs from the fixed order"
    base = torch.randn(L, generator=torch.Generator().manual_seed(7))
    acc = base.cuda()
    K.sum_rows(parts.cuda().view(-1), P, L, acc, accumulate=True)
    assert torch.equal(acc.cpu(), base + want), "accumulating sum"
    slab = K.scratch(torch.device("cuda"), "test_sum_rows", parts.numel() * 4)[:parts.numel()]
    out2 = torch.empty(L, device="cuda")
    with K.sum_batch():
        slab = K.scratch(torch.device("cuda"), "test_sum_rows", parts.numel() * 4)[:parts.numel()]
        slab.copy_(parts.view(-1))
        K.sum_rows(slab, P, L, out2, deferred=True)
        K._sum_batch.temp_dest = True               # out2 is read right after the block: the batch is summed at its end, not as passengers
    assert torch.equal(out2.cpu(), want), "batched sum"


@pytest.mark.parametrize("act", [R.ACT_GELU, R.ACT_SILU])
def test_wave_autonomous_wgrad_with_bn_act_prologue(act):
    """EfficientFormerV2 ConvMlp fc2 weight gradient at the 56x56 stage: narrow operand through the BN-backward affine
    map, wide operand through BN + activation (no gate), M above the wave-autonomous kernel's threshold."""
    K = _k()
    rd = torch.bfloat16
    N, HW, Ni, Nj = 4, 50003, 32, 128
    p, p2 = gen((N, HW, 1, Ni), 1, rd, 0.5), gen((N, HW, 1, Ni), 2, rd, 0.5)
    q = gen((N, HW, 1, Nj), 3, rd, 0.5)
    coef3 = rand_state(Ni, 4)[:3].contiguous()
    st = rand_state(Nj, 5)
    P = R.prologue(p.float().view(N, HW, Ni), 3, rd, coef=coef3, a2=p2.float().view(N, HW, Ni))
    Q = R.prologue(q.float().view(N, HW, Nj), 1, rd, act, st)
    want = P.reshape(-1, Ni).t().double() @ Q.reshape(-1, Nj).double()
    dp2, dcoef, dst = p2.cuda(), coef3.cuda(), st.cuda()
    got = K.pwconv_wgrad(p.cuda(), K.pro_affine2(dp2, dcoef), q.cuda(), K.pro_bn_act(dst, act))
    close(got, want.float(), 4e-3, "wgrad affine2 x bn_act")


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("cfg", [(3, 1, 1, 64, 64, 28, 6), (3, 2, 1, 24, 48, 15, 3), (3, 1, 1, 128, 128, 9, 5), (3, 2, 1, 96, 192, 14, 2),
                                 (3, 1, 1, 96, 64, 14, 3), (3, 1, 1, 64, 128, 56, 2), (3, 1, 1, 128, 192, 17, 1), (3, 1, 1, 64, 48, 12, 2)])
def test_dense_conv_as_implicit_gemm(rd, cfg):
    """dfd_conv_fwd against the im2col + GEMM pair it replaces and against torch's conv2d; with and without the producer's
    BN + GELU, with BN statistics.  Two kernels serve it: the implicit GEMM (the GEMM kernel gathers its A operand from the
    image: same kernel, same summation order as im2col + GEMM -> identical) and, for bf16 3x3 stride-1 layers with 64 / 96 /
    128 input channels and a multiple of 64 output channels, the direct kernel with register-resident weights (dfd_conv3.hip:
    another summation order -> rounding tolerance; its statistics are sums of its own rounded outputs)."""
    K = _k()
    k, s, p, C, Co, H, N = cfg
    Ho = (H + 2 * p - k) // s + 1
    direct = rd == torch.bfloat16 and k == 3 and s == 1 and p == 1 and C in (64, 96, 128) and Co % 64 == 0
    x = gen((N, H, H, C), 1, rd).cuda()
    st = rand_state(C, 2).cuda()
    w = gen((Co, C, k, k), 3, torch.float32, C ** -0.5)
    w_nk, _ = K.prep_weights(K.conv_weight_to_gemm(w.cuda()), rd, True, False)
    for state, act in ((None, R.ACT_NONE), (st, R.ACT_GELU)):
        col = K.im2col(x, state, act, k, s, p, Ho, Ho)
        y0, parts0, n0 = K.pwconv(col, None, w_nk, None, stats=True)
        y1, parts1, n1 = K.conv_fwd(x, state, act, w_nk, k, s, p, Ho, Ho, stats=True)
        if direct:
            close(y1, y0.float().cpu(), tol(rd), "direct 3x3 conv vs im2col + GEMM")
            sums = parts1[:n1 * 2 * Co].view(n1, 2, Co).double().sum(0).cpu()
            yf = y1.double().view(-1, Co).cpu()
            assert torch.allclose(sums[0], yf.sum(0), rtol=1e-5, atol=1e-3), float((sums[0] - yf.sum(0)).abs().max())
            assert torch.allclose(sums[1], (yf * yf).sum(0), rtol=1e-5, atol=1e-3)
        else:
            assert torch.equal(y0, y1), float((y0.float() - y1.float()).abs().max())
            assert n0 == n1 and torch.equal(parts0[:n0 * 2 * Co], parts1[:n1 * 2 * Co])
        y2, _, _ = K.conv_fwd(x, state, act, w_nk, k, s, p, Ho, Ho, stats=False)
        assert torch.equal(y1, y2)
        a = x.float().cpu() if state is None else R.rnd(F.gelu(st.cpu()[0] * x.float().cpu() + st.cpu()[1]), rd)
        want = F.conv2d(a.permute(0, 3, 1, 2), R.rnd(w, rd), stride=s, padding=p).permute(0, 2, 3, 1)
        close(y1, want, tol(rd), "conv vs conv2d")


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("cfg", [(3, 1, 1, 64, 64, 28, 6), (3, 2, 1, 24, 48, 15, 3), (3, 1, 1, 128, 128, 9, 5), (3, 2, 1, 96, 192, 14, 2),
                                 (3, 1, 1, 64, 128, 56, 3), (3, 1, 1, 128, 64, 33, 2), (3, 1, 1, 192, 64, 7, 40), (3, 1, 1, 96, 64, 14, 3)])
def test_dense_conv_weight_gradient_as_implicit_gemm(rd, cfg):
    """dfd_conv_wgrad against im2col + dfd_pwconv_wgrad; with the BN-backward map on the gradient operand and the producer's
    BN + GELU on the gathered one.  The TN kernel that gathers the im2col operand: same kernel, same split of the rows, same
    summation order -> identical.  bf16 3x3 stride-1 layers with channel counts that are multiples of 64 run the direct kernel
    (dfd_conv3.hip, output block resident in registers): another f32 summation order -> 1e-5 of the largest entry."""
    K = _k()
    k, s, p, C, Co, H, N = cfg
    direct = rd == torch.bfloat16 and k == 3 and s == 1 and p == 1 and C % 64 == 0 and Co % 64 == 0
    Ho = (H + 2 * p - k) // s + 1
    x = gen((N, H, H, C), 1, rd).cuda()
    st = rand_state(C, 2).cuda()
    dz, y = gen((N, Ho, Ho, Co), 3, rd).cuda(), gen((N, Ho, Ho, Co), 4, rd).cuda()
    coef = rand_state(Co, 5)[:3].contiguous().cuda()
    for pro_p in (None, K.pro_affine2(y, coef)):
        for state, act in ((None, R.ACT_NONE), (st, R.ACT_GELU)):
            col = K.im2col(x, state, act, k, s, p, Ho, Ho)
            want = K.pwconv_wgrad(dz, pro_p, col, None)
            got = K.conv_wgrad(dz, pro_p, x, state, act, k, s, p)
            if direct:
                assert float((want - got).abs().max()) <= 1e-5 * float(want.abs().max()), (float((want - got).abs().max()), float(want.abs().max()))
            else:
                assert torch.equal(want, got), float((want - got).abs().max())


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("cfg", [(3, 64, 96, 14, 3), (3, 128, 128, 9, 2), (5, 32, 48, 11, 2)])
def test_stride1_conv_data_gradient_is_a_forward_conv_with_the_flipped_weight(rd, cfg):
    """dx of a stride-1 'same' convolution = dfd_conv_fwd of the output gradient with the [I][(flipped tap, o)] weight
    (dfd_conv_weight_perm mode 2): against torch autograd and against the column-matrix path it replaces."""
    K = _k()
    k, C, Co, H, N = cfg
    p = k // 2
    w = gen((Co, C, k, k), 1, torch.float32, (C * k * k) ** -0.5)
    g = gen((N, H, H, Co), 2, rd)
    x = torch.zeros((N, C, H, H), requires_grad=True)
    F.conv2d(x, R.rnd(w, rd), padding=p).backward(g.float().permute(0, 3, 1, 2))
    want = x.grad.permute(0, 2, 3, 1)
    wd_nk, _ = K.prep_weights(K.conv_weight_to_dgrad_gemm(w.cuda()), rd, True, False)
    got, _, _ = K.conv_fwd(g.cuda(), None, R.ACT_NONE, wd_nk, k, 1, p, H, H, stats=False)
    close(got, want, tol(rd), "dgrad as forward conv")
    _, w_kn = K.prep_weights(K.conv_weight_to_gemm(w.cuda()), rd, False, True)
    dcol, _, _ = K.pwconv(g.cuda(), None, w_kn, None, stats=False)
    old = K.col2im(dcol, (N, H, H, C), k, 1, p)
    close(got, old.float().cpu(), tol(rd), "dgrad: implicit GEMM vs column matrix + col2im")


# ---------------------------------------------------------------------------------------------------------------------
# fused window attention on the bf16 matrix cores (csrc/dfd_attn.hip) against plain torch f32 attention
@pytest.mark.parametrize("n,T,H,with_bias", [(5, 53, 8, True), (4, 49, 16, True), (3, 16, 8, True), (9, 49, 4, False), (2, 64, 2, True),
                                              (1, 1, 1, True), (6, 33, 3, True)])
def test_fused_window_attention_forward_and_backward(n, T, H, with_bias):
    """o = softmax(scale q k^T + bias) v read in place from the qkv projection output; backward recomputes P from the saved
    log-sum-exp.  Operands are bf16 (exact inputs for both sides); P and dS are rounded to bf16 for their products, so the
    tolerance is bf16's (two ulps of the tensor's largest entry).  Ragged window counts (n % 4 != 0), padded token tiles
    (T = 49, 53, 33 of 64; 16 of 32; a single token) and the bias gradient's cross-window sum are all exercised."""
    from deepfakedetection_amd import kernels as K

    hd, C = 32, H * 32
    g = torch.Generator().manual_seed(n * 1000 + T)
    qkv = (torch.randn(n, T, 3 * C, generator=g) * 1.5).to(BF)
    bias = torch.randn(H, T, T, generator=g) if with_bias else None
    dO = torch.randn(n, T, C, generator=g).to(BF)
    scale = hd ** -0.5
    # reference in f32 on the bf16-exact operands
    q, k, v = [t.float().view(n, T, H, hd).permute(0, 2, 1, 3).requires_grad_() for t in qkv.split(C, dim=-1)]
    bref = bias.clone().requires_grad_() if with_bias else None
    S = q @ k.transpose(-1, -2) * scale
    if with_bias:
        S = S + bref
    P = S.softmax(-1)
    o_ref = (P @ v).permute(0, 2, 1, 3).reshape(n, T, C)
    o_ref.backward(dO.float())
    dqkv_ref = torch.cat([t.grad.permute(0, 2, 1, 3).reshape(n, T, C) for t in (q, k, v)], dim=-1)
    L_ref = torch.logsumexp(S.detach(), dim=-1)

    qd = qkv.cuda().view(n, T, 1, 3 * C)
    bd = bias.cuda() if with_bias else None
    o, L = K.wattn_fwd(qd, bd, H, scale)
    dqkv, dbias = K.wattn_bwd(qd, dO.cuda().view(n, T, 1, C), L, bd, H, scale, with_bias)
    torch.cuda.synchronize()
    tol = 1.6e-2

    def err(got, want):
        return float((got.float().cpu().reshape(want.shape) - want).abs().max()) / max(float(want.abs().max()), 1e-12)

    assert err(L, L_ref) <= 1e-5
    assert err(o, o_ref.detach()) <= tol, err(o, o_ref.detach())
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        e = err(dqkv.view(n, T, 3 * C)[..., sl], dqkv_ref[..., sl])
        assert e <= tol, (name, e)
    if with_bias:
        assert err(dbias, bref.grad) <= tol, err(dbias, bref.grad)
    # bitwise reproducible (fixed-order bias-gradient sum)
    dqkv2, dbias2 = K.wattn_bwd(qd, dO.cuda().view(n, T, 1, C), L, bd, H, scale, with_bias)
    assert torch.equal(dqkv, dqkv2) and (not with_bias or torch.equal(dbias, dbias2))


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rd", DT)
def test_packed_pair_gelu_is_the_scalar_gelu_bit_for_bit(rd):
    """The hot kernels evaluate GELU / GELU' on float2 pairs (v_pk_*_f32: dfd_common.h act_fwd2 / act_grad2); the row kernels
    dfd_bn_add_act / dfd_bn_add_act_bwd still use the scalar forms.  Same operations in the same order -> identical bits, over
    the whole range where the erfc polynomial, the exp underflow and the sign select matter."""
    K = _k()
    N, H, W, C = 4, 9, 7, 64
    y = (gen((N, H, W, C), 71, rd) * 3.0)
    y.view(-1)[:8] = torch.tensor([0.0, -0.0, 1e-8, -1e-8, 12.0, -12.0, 40.0, -40.0]).to(rd)
    g = gen((N, H, W, C), 72, rd)
    st = rand_state(C, 73)
    y, g, st = y.cuda(), g.cuda(), st.cuda()
    packed = K.bn_act_apply(y, st, R.ACT_GELU)
    scalar = K.bn_add_act(y, st, None, R.ACT_GELU)
    assert torch.equal(packed, scalar), float((packed.float() - scalar.float()).abs().max())
    d_packed, _, _ = K.act_bn_bwd(g, y, None, None, st, R.ACT_GELU)
    d_scalar, _, _ = K.bn_add_act_bwd(g, y, st, None, R.ACT_GELU, stats=False)
    assert torch.equal(d_packed, d_scalar), float((d_packed.float() - d_scalar.float()).abs().max())


@pytest.mark.parametrize("rd", DT)
def test_packed_pair_silu_is_the_scalar_silu_bit_for_bit(rd):
    """SiLU on float2 pairs (round 4: act_fwd2<SILU>, every GEMM / row-pass prologue) against the scalar form the bn_add_act row
    kernel still uses: z * rcp(1 + exp2(-z * log2 e)), same operations in the same order."""
    K = _k()
    N, H, W, C = 4, 9, 7, 64
    y = (gen((N, H, W, C), 81, rd) * 4.0)
    y.view(-1)[:8] = torch.tensor([0.0, -0.0, 1e-8, -1e-8, 20.0, -20.0, 90.0, -90.0]).to(rd)
    st = rand_state(C, 83)
    y, st = y.cuda(), st.cuda()
    packed = K.bn_act_apply(y, st, R.ACT_SILU)
    scalar = K.bn_add_act(y, st, None, R.ACT_SILU)
    assert torch.equal(packed, scalar), float((packed.float() - scalar.float()).abs().max())
    g = gen((N, H, W, C), 82, rd).cuda()
    d_packed, _, _ = K.act_bn_bwd(g, y, None, None, st, R.ACT_SILU)
    d_scalar, _, _ = K.bn_add_act_bwd(g, y, st, None, R.ACT_SILU, stats=False)
    assert torch.equal(d_packed, d_scalar), float((d_packed.float() - d_scalar.float()).abs().max())


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(256 * 80 + 17, 192, 520), (256 * 170, 128, 256), (256 * 41 + 255, 1024, 1032)])
def test_plain_bf16_product_on_256_tiles_with_lds_dma(shape):
    """Plain bf16 products with >= 160 tiles of 256 x 256 and K % 64 == 0 run `k_gemm_nt_dma` (csrc/dfd_gemm.hip: LDS-DMA staging,
    swizzle on the source address, eight waves): ragged last row / column tiles (clamped sources, masked stores), K from two
    stages up.  Against the f32 product of the same bf16 operands: one bf16 rounding of the result."""
    K = _k()
    M, Kd, N = shape
    from deepfakedetection_amd._lib import load

    # K.pwconv falls back to the 128 x 128 kernel when the planner declines: make sure this shape IS the LDS-DMA kernel's
    assert load().dfd_gemm_plan(M, Kd, N) in (128, 256), "the LDS-DMA planner no longer serves this shape: the test would run k_pw_nt"
    assert load().dfd_gemm_plan(M, Kd + 8, N) == 0 and load().dfd_gemm_plan(200, Kd, N) == 0
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn((M, 1, 1, Kd), generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn((N, Kd), generator=g) * Kd ** -0.5).cuda()
    w_nk, _ = K.prep_weights(w, torch.bfloat16, True, False)
    out, _, _ = K.pwconv(a, None, w_nk, None, stats=False)
    want = a.view(M, Kd).float() @ w.to(torch.bfloat16).float().t()
    err = (out.view(M, N).float() - want).abs().max()
    assert float(err) <= 2.0 ** -8 * float(want.abs().max()) + 1e-6, (float(err), float(want.abs().max()))
    # rows far apart and the ragged edges, exactly as stored
    rows = torch.tensor([0, 1, 255, 256, M // 2, M - 2, M - 1], device="cuda")
    assert torch.equal(out.view(M, N)[rows], want[rows].to(torch.bfloat16)) or float(
        (out.view(M, N)[rows].float() - want[rows]).abs().max()) <= 2.0 ** -8 * float(want.abs().max())


@pytest.mark.parametrize("case", [(256 * 170, 256, 256, R.ACT_GELU, False, False, True), (256 * 80 + 17, 192, 520, R.ACT_NONE, True, True, False),
                                  (49 * 256, 512, 512, R.ACT_NONE, True, True, True), (256 * 170, 128, 768, R.ACT_NONE, False, False, False)])
def test_linear_with_fused_bias_activation_residual_epilogue_is_bitwise_the_two_kernel_form(case):
    """dfd_gemm_bias_act: act(scale * y + shift) [* row_scale] [+ residual] in the product's store loop, on the bf16-rounded y —
    identical bits to dfd_pwconv_fwd followed by dfd_bn_act_apply, and the optional raw y equals the plain product.
    (A comparison of two HIP paths on purpose: both sides are checked against the oracle on their own —
    test_plain_bf16_product_on_256_tiles_with_lds_dma above, test_ops_gpu.py::test_rowpass — this test pins the FUSION.)"""
    K = _k()
    M, Kd, N, act, with_res, with_rs, want_raw = case
    T = 53 if M % 53 == 0 else (49 if M % 49 == 0 else 1)
    n = M // T
    g = torch.Generator().manual_seed(M + N + act)
    x = torch.randn((n, T, 1, Kd), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((N, Kd), generator=g) * Kd ** -0.5).cuda()
    w_nk, _ = K.prep_weights(w, torch.bfloat16, True, False)
    st = rand_state(N, 91).cuda()
    res = torch.randn((n, T, 1, N), generator=g).to(torch.bfloat16).cuda() if with_res else None
    rs = (0.5 + torch.rand(n, generator=g)).cuda() if with_rs else None
    fused = K.gemm_bias_act(x, w_nk, st, act, res, rs, want_raw=want_raw)
    assert fused is not None, "shape expected to be served by the fused kernel"
    out, raw = fused
    y, _, _ = K.pwconv(x, None, w_nk, None, stats=False)
    want = K.bn_act_apply(y, st, act, res, rs)
    assert torch.equal(out, want), float((out.float() - want.float()).abs().max())
    assert (raw is None) == (not want_raw)
    if raw is not None:
        assert torch.equal(raw, y)


def test_eval_mode_backward_of_a_batchnormed_1x1_at_a_fused_shape():
    """in eval mode `pwbn_fwd` may serve a layer with the fused product + BN-apply kernel, and a layer with a REAL
    BatchNorm still needs the raw product in its backward (bn_bwd_reduce, the BN-backward prologue) — frozen-BN fine-tuning and
    attribution runs differentiate in eval mode.  A shape the fused kernel accepts (K % 64 == 0, N >= 256, >= 160 tiles): the raw y
    must be returned and the backward must match the eval-mode BatchNorm arithmetic."""
    from deepfakedetection_amd._lib import load
    from deepfakedetection_amd.functions import BNRef
    from deepfakedetection_amd.vit_functions import pwbn_bwd, pwbn_fwd

    K = _k()
    n, T, Kd, N = 160, 128, 128, 256
    assert load().dfd_gemm_plan(n * T, Kd, N) != 0
    g0 = torch.Generator().manual_seed(11)
    x = torch.randn((n, T, 1, Kd), generator=g0).to(torch.bfloat16).cuda()
    w = (torch.randn((N, Kd, 1, 1), generator=g0) * Kd ** -0.5).cuda()
    gamma, beta = (0.5 + torch.rand(N, generator=g0)).cuda(), (torch.randn(N, generator=g0) * 0.2).cuda()
    rm, rv = (torch.randn(N, generator=g0) * 0.1).cuda(), (0.5 + torch.rand(N, generator=g0)).cuda()
    bn = BNRef(rm, rv, None, 0.1, 1e-5)
    w_nk, w_kn = K.prep_weights(w, torch.bfloat16, True, True)
    out, y, st = pwbn_fwd(x, w_nk, None, gamma, beta, bn, False, None)
    assert y is not None, "a BatchNormed layer lost its raw product: its eval-mode backward would dereference None"
    g = torch.randn(out.shape, generator=g0).to(torch.bfloat16).cuda()
    dx, dw, _, dgamma, dbeta, _ = pwbn_bwd(g, x, y, st, w_kn, tuple(w.shape), w, None, gamma, beta, None, R.ACT_NONE, False, True, True, True)
    # reference: eval BatchNorm is the affine map out = a * y + c with a = gamma * rstd
    xf, wf, gf = x.float().view(-1, Kd), w.view(N, Kd).to(torch.bfloat16).float(), g.float().view(-1, N)
    yf = xf @ wf.t()
    rstd = (rv + 1e-5).rsqrt()
    want_out = (yf - rm) * rstd * gamma + beta
    dy = (gf * (gamma * rstd)).to(torch.bfloat16).float()
    close(out.view(-1, N), want_out, 1.6e-2, "eval pw+bn out")
    close(y.view(-1, N), yf, 1.6e-2, "eval pw+bn raw y")
    close(dx.view(-1, Kd), dy @ wf, 2e-2, "eval pw+bn dx")
    close(dw.view(N, Kd), dy.t() @ xf, 2e-2, "eval pw+bn dw")
    close(dbeta, gf.sum(0), 2e-2, "eval pw+bn dbeta")
    close(dgamma, (gf * ((y.float().view(-1, N) - rm) * rstd)).sum(0), 2e-2, "eval pw+bn dgamma")
