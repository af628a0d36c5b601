"""Fused autograd functions of the EfficientFormerV2 engine (and the pieces FasterViT shares).

One torch.autograd.Function per network stage, built from plain forward/backward helper pairs (no autograd
inside a stage, so a tensor with several consumers gets its gradient summed by the `residual` operand of a
GEMM launch, never by an ATen add):

  ConvStemFunction     conv3x3 s2 on the f32 image (+bias) + BN + GELU                    timm Stem4.conv1
  DenseConvBNFunction  dense 3x3 conv (+bias) + BN [+ GELU] as im2col + MFMA GEMM         Stem4.conv2, Downsample.conv
  ConvMlpFunction      1x1 BN GELU -> dw3x3 BN GELU -> 1x1 BN -> *ls + x                  EfficientFormerV2Block.mlp/ls2
  AttentionFunction    [dw3x3 s2 BN] -> q,k,v 1x1 BN -> talking-head attention + v_local -> [bilinear x2] -> GELU
                       -> 1x1 BN -> *ls + x                                              .token_mixer/ls1
  DownsampleFunction   conv3x3 s2 BN  [+ Attention2dDownsample branch]                    EfficientFormerV2Stage.downsample
  TailFunction         BN -> mean(H, W) -> dropout -> (head + head_dist) / 2             EfficientFormerV2.norm / forward_head

As in functions.py every convolution writes its RAW output plus per-channel partial sums; BatchNorm, the
convolution bias (folded into the BN mean), LayerScale (folded into the BN affine map) and the activation are
applied by the consumer.  Reference call sites: trainers/efficientformer_v2.py:244 (train), :215 (evaluate),
:369 (warm-up), orchestration/orchestrator.py:529,590 (inference); arithmetic per timm 1.0.20
efficientformer_v2.py (restated in oracle/efformer_ref.py).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch

from . import kernels as K
from ._lib import ACT_GELU, ACT_NONE
from .arena import grad_dest
from .functions import BNRef, _bn_state, _c


def _slot(t: torch.Tensor | None, need: bool, shape):
    """Gradient-arena slot of parameter `t` if it wants a gradient and its slot is free."""
    if not need or t is None:
        return None
    return grad_dest(t.data_ptr(), shape)


def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1]


def _prep(w: torch.Tensor, dt: torch.dtype, derived, need_bwd: bool = True):
    """(w_nk, w_kn) of a 1x1 convolution weight [O, I, 1, 1] (or a 2-D GEMM weight) in the activation dtype.
    derived: {weight.data_ptr(): (w_nk, w_kn)} refreshed for the whole network by one batched launch per forward
    (kernels.DerivedWeights), or None: prepare this layer on the spot."""
    if derived is not None:
        hit = derived.get(w.data_ptr())
        if hit is not None:
            return hit
    return K.prep_weights(w, dt, True, need_bwd)


# =========================================================================== helper pairs (no autograd)
def pwbn_fwd(x, w_nk, b, gamma, beta, bn: BNRef, training, counters, act=ACT_NONE, ls=None, residual=None, row_scale=None,
             raw_unused=False):
    """1x1 conv (+bias) -> BN -> act, [* ls] [* row_scale] [+ residual]; returns (materialised out, raw y, bn state).
    raw_unused: the caller's backward never reads y (the identity-statistics Linear layers of FasterViT, whose backward is
    pwbn_bwd's `plain` path) — only then may the fused eval kernel skip writing it.  Every caller with a real BatchNorm needs y in
    its backward even in eval mode (bn_bwd_reduce / the BN-backward prologue read it: frozen-BN fine-tuning, attribution runs)."""
    if not training and act in (ACT_NONE, ACT_GELU):
        # eval statistics are known before the product: one kernel (csrc/dfd_gemm.hip) where the shape is its own.  The raw y is
        # produced only where a backward reads it (activation, LayerScale); otherwise None is returned in its place — NOT `out`:
        # a block Function that kept its own output among its saved tensors would close the reference cycle that once crashed
        # hipStreamEndCapture.
        st = _bn_state(None, 0, _rows(x), bn, gamma, beta, False, counters, conv_bias=b, ls=ls)
        fused = K.gemm_bias_act(x, w_nk, st, act, residual, row_scale, want_raw=(not raw_unused or act != ACT_NONE or ls is not None))
        if fused is not None:
            out, raw = fused
            return out, raw, st
        y, _, _ = K.pwconv(x, None, w_nk, None, stats=False)
        return K.bn_act_apply(y, st, act, residual, row_scale), y, st
    y, parts, n = K.pwconv(x, None, w_nk, None, stats=training)
    st = _bn_state(parts, n, _rows(y), bn, gamma, beta, training, counters, conv_bias=b, ls=ls)
    out = K.bn_act_apply(y, st, act, residual, row_scale)
    return out, y, st


def pwbn_bwd(g, x, y, st, w_kn, w_shape, w, b, gamma, beta, ls, act, training, need_dx, need_w, need_bn, need_ls=False,
             dx_residual=None, row_scale=None, need_b=None, identity=False):
    """Backward of pwbn_fwd for the gradient g of its output.  Returns (dx, dw, db, dgamma, dbeta, dls); dx already
    includes `dx_residual` (the running sum of the other consumers' gradients of x).
    identity: the "BatchNorm" is the identity statistic of a Linear layer (fastervit_functions.ident): its backward map
    is dy = ls * dz, so the two GEMMs take dz as it is (or scaled per channel) instead of the two-operand BN-backward
    prologue, which would read y only to multiply it by zero."""
    if row_scale is not None:
        g = K.scale_rows(g, row_scale)
    # a Linear layer without layer scale: the BatchNorm-shaped backward degenerates to dz = g and dbias = column sums of g
    plain = act == ACT_NONE and identity and not training and ls is None and not need_ls and b is None
    if plain:
        dbeta = K.bias_grad(g, None, _slot(beta, need_bn, (gamma.numel(),))) if (need_bn and beta is not None) else None
        dx = dx_residual
        if need_dx:
            dx, _, _ = K.pwconv(g, None, w_kn, dx_residual, stats=False)
        dw = None
        if need_w:
            O, I = w_shape[0], w_shape[1]
            dw = K.pwconv_wgrad(g, None, x, None, _slot(w, True, (O, I))).view(w_shape)
        return dx, dw, None, None, dbeta, None
    if act == ACT_NONE:
        parts, n = K.bn_bwd_reduce(g, y, st, None)
        dz = g
    else:
        dz, parts, n = K.act_bn_bwd(g, y, None, None, st, act)
    C = gamma.numel()
    want_b = (need_w if need_b is None else need_b) and b is not None
    outs = (_slot(gamma, need_bn, (C,)), _slot(beta, need_bn, (C,)), _slot(ls, need_ls, (C,)), _slot(b, want_b, (C,)))
    coef, dgamma, dbeta, dls, db = K.bn_bwd_finalize_ex(parts, n, _rows(y), gamma, beta, ls, st, training, need_bn, need_ls,
                                                        want_b, outs)
    if identity and not training:
        pro_d = pro_w = None
        if ls is not None:
            # rows 0, 1 of coef = (ls, 0): a plain per-channel scale (the GEMM kernels add a residual only behind the
            # prologues the EfficientNet path uses, so with one the two-operand form stays)
            pro_w = K.pro_affine2(y, coef)
            pro_d = K.pro_bn_act(coef, ACT_NONE) if dx_residual is None else pro_w
    else:
        pro_d = pro_w = K.pro_affine2(y, coef)
        if need_dx and need_w and K.pwconv_bwd_fused_ok(dz, x):
            # narrow input, wide BatchNormed output at many rows (EfficientFormerV2 stage-1 fc1: 32 -> 128 at 802,816 rows): data and
            # weight gradient from one pass over (dz, y) — csrc/dfd_pwtnw.hip, DG; bit-identical to the two kernels below
            O, I = w_shape[0], w_shape[1]
            dx, dw = K.pwconv_bwd_fused(dz, y, coef, x, w_kn, dx_residual, _slot(w, True, (O, I)))
            return dx, dw.view(w_shape), db, dgamma, dbeta, dls
    dx = dx_residual
    if need_dx:
        dx, _, _ = K.pwconv(dz, pro_d, w_kn, dx_residual, stats=False)
    dw = None
    if need_w:
        O, I = w_shape[0], w_shape[1]
        dw = K.pwconv_wgrad(dz, pro_w, x, None, _slot(w, True, (O, I))).view(w_shape)
    return dx, dw, db, dgamma, dbeta, dls


def dwbn_fwd(x, w, b, gamma, beta, bn: BNRef, training, counters, k: int, stride: int):
    """depthwise k x k (+bias) -> raw y and the BN state (the consumer applies it)."""
    N, H, W, C = x.shape
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
    y, parts, n = K.dwconv_fwd(x, None, ACT_NONE, w, k, stride, p, p, Ho, Wo, stats=training)
    st = _bn_state(parts, n, N * Ho * Wo, bn, gamma, beta, training, counters, conv_bias=b)
    return y, st


def dwbn_bwd(dz, parts, n, x, y, st, w, b, gamma, beta, training, k, stride, need_dx, need_w, need_bn):
    """dz: gradient at the BN output (with its partial sums).  Returns (dx, dw, db, dgamma, dbeta)."""
    C = gamma.numel()
    p = k // 2
    outs = (_slot(gamma, need_bn, (C,)), _slot(beta, need_bn, (C,)), None, _slot(b, need_w, (C,)))
    coef, dgamma, dbeta, _, db = K.bn_bwd_finalize_ex(parts, n, _rows(y), gamma, beta, None, st, training, need_bn, False,
                                                      need_w and b is not None, outs)
    dx = dw = None
    if need_dx:
        dx, _, _ = K.dwconv_bwd_data(dz, y, coef, w, None, None, ACT_NONE, tuple(x.shape), k, stride, p, p)
    if need_w:
        dw = K.dwconv_bwd_weight(dz, y, coef, x, None, ACT_NONE, k, stride, p, p, _slot(w, True, (C, 1, k, k)))
    return dx, dw, db, dgamma, dbeta


@dataclass
class AttnGeom:
    heads: int
    dk: int
    dv: int
    Nq: int
    Nk: int
    scale: float


def attn_core_fwd(q, k, v, table, idx, th, geo: AttnGeom):
    """q [B,hq,wq,H*dk], k [B,hk,wk,H*dk], v [B,hk,wk,H*dv] (NHWC, read in place head by head) ->
    O [B,hq,wq,H*dv] and the saved (S, P, T2) f32 tensors."""
    B = q.shape[0]
    H, dk, dv, Nq, Nk = geo.heads, geo.dk, geo.dv, geo.Nq, geo.Nk
    bias_full = K.bias_gather(table, idx)                                        # [H, Nq*Nk]
    if K.attn_mfma_supported(q.dtype, Nq, Nk, dk, dv):
        # bf16, <= 64 tokens: the two products on the matrix cores, one wave per (image, head) (csrc/dfd_attn.hip)
        S = K.attn_scores(q, k, H, geo.scale, bias_full)
        P, T2 = K.attn_softmax_fwd(S, th)
        return K.attn_apply(T2, v, (*q.shape[:3], H * dv), H), S, P, T2
    S = torch.empty((B, H, Nq, Nk), dtype=torch.float32, device=q.device)
    K.bgemm(q, (Nq * H * dk, dk, H * dk, 1), k, (Nk * H * dk, dk, 1, H * dk), S, (H * Nq * Nk, Nq * Nk, Nk, 1),
            B, H, Nq, Nk, dk, alpha=geo.scale, bias=bias_full)
    P, T2 = K.attn_softmax_fwd(S, th)
    O = torch.empty((*q.shape[:3], H * dv), dtype=q.dtype, device=q.device)
    K.bgemm(T2, (H * Nq * Nk, Nq * Nk, Nk, 1), v, (Nk * H * dv, dv, H * dv, 1), O, (Nq * H * dv, dv, H * dv, 1), B, H, Nq, dv, Nk)
    return O, S, P, T2


def _partial_rows(B: int) -> int:
    return B + (B + 31) // 32 + 1


def attn_core_bwd(gO, q, k, v, S, P, T2, table, idx, th, geo: AttnGeom, need_table: bool, need_th: bool, th_params=None):
    """Returns (dQ, dK, dV, dtable, (dw1, db1, dw2, db2) | None).  th_params: the talking-head parameter tensors
    (weight1, bias1, weight2, bias2) whose gradient-arena slots receive the results."""
    B = q.shape[0]
    H, dk, dv, Nq, Nk = geo.heads, geo.dk, geo.dv, geo.Nq, geo.Nk
    L = Nq * Nk
    dev = q.device
    if B > 1024:
        raise RuntimeError("attention backward sums per-image partials in one two-stage pass: batch <= 1024")
    mfma = K.attn_mfma_supported(q.dtype, Nq, Nk, dk, dv) and gO.dtype == q.dtype
    if mfma:
        gO = _c(gO)
        dT2 = K.attn_scores(gO, v, H)
        dV = K.attn_apply(T2, gO, v.shape, H, transpose=True)
    else:
        dT2 = torch.empty((B, H, Nq, Nk), dtype=torch.float32, device=dev)
        K.bgemm(gO, (Nq * H * dv, dv, H * dv, 1), v, (Nk * H * dv, dv, 1, H * dv), dT2, (H * L, L, Nk, 1), B, H, Nq, Nk, dv)
        dV = torch.empty_like(v)
        K.bgemm(T2, (H * L, L, 1, Nk), gO, (Nq * H * dv, dv, H * dv, 1), dV, (Nk * H * dv, dv, H * dv, 1), B, H, Nk, dv, Nq)
    # dS gets room for the two-stage row sum that turns it into the bias gradient
    dS_buf = torch.empty((_partial_rows(B), H, Nq, Nk), dtype=torch.float32, device=dev)
    dS = dS_buf[:B]
    dT1 = torch.empty_like(dT2) if th is not None else None
    from ._lib import check  # local: the front end takes whole tensors, here dS is a prefix view

    lib = K._L()
    if th is None:
        check(lib.dfd_attn_softmax_bwd(dT2.data_ptr(), P.data_ptr(), None, None, None, dS.data_ptr(), B, H, Nq, Nk, K._stream()),
              "dfd_attn_softmax_bwd")
    else:
        check(lib.dfd_attn_softmax_bwd(dT2.data_ptr(), P.data_ptr(), th[0].data_ptr(), th[2].data_ptr(), dT1.data_ptr(),
                                       dS.data_ptr(), B, H, Nq, Nk, K._stream()), "dfd_attn_softmax_bwd")
    if mfma:
        dQ = K.attn_apply(dS, k, q.shape, H, alpha=geo.scale)
        dK = K.attn_apply(dS, q, k.shape, H, alpha=geo.scale, transpose=True)
    else:
        dQ = torch.empty_like(q)
        K.bgemm(dS, (H * L, L, Nk, 1), k, (Nk * H * dk, dk, H * dk, 1), dQ, (Nq * H * dk, dk, H * dk, 1), B, H, Nq, dk, Nk, alpha=geo.scale)
        dK = torch.empty_like(k)
        K.bgemm(dS, (H * L, L, 1, Nk), q, (Nq * H * dk, dk, H * dk, 1), dK, (Nk * H * dk, dk, H * dk, 1), B, H, Nk, dk, Nq, alpha=geo.scale)
    dth = None
    if th is not None and need_th:
        w1, b1, w2, b2 = th
        pw1, pb1, pw2, pb2 = th_params if th_params is not None else (None, None, None, None)

        def dst(param, n):
            # -> (destination, deferred): a gradient-arena slot is read by nobody before the optimizer, so the final sum of the per-image
            # rows may join the enclosing block's batched sums (kernels.sum_batch); a fresh tensor is handed to autograd and summed now
            slot = _slot(param, param is not None, tuple(param.shape)) if param is not None else None
            return (slot.view(-1), True) if slot is not None else (torch.empty(n, dtype=torch.float32, device=dev), False)

        def rows_buf(name, n, deferred):
            # (a deferred sum's slab must outlive this function: scratch() keeps the slabs of an open batch until they are summed)
            if deferred:
                return K.scratch(dev, name, _partial_rows(B) * n * 4)[:_partial_rows(B) * n].view(_partial_rows(B), n)
            return torch.empty((_partial_rows(B), n), dtype=torch.float32, device=dev)

        # dW2[g][h] = sum_{b,l} dT2[b,g,l] * P[b,h,l]   (per-image partials, then a fixed-order row sum)
        d2, defer2 = dst(pw2, H * H)
        part = rows_buf("th_dw2", H * H, defer2)
        K.bgemm(dT2, (H * L, 0, L, 1), P, (H * L, 0, 1, L), part, (H * H, 0, H, 1), B, 1, H, H, L)
        dw2 = K.sum_rows(part.view(-1), B, H * H, d2, deferred=defer2).view(H, H, 1, 1)
        d1, defer1 = dst(pw1, H * H)
        part1 = rows_buf("th_dw1", H * H, defer1)
        K.bgemm(dT1, (H * L, 0, L, 1), S, (H * L, 0, 1, L), part1, (H * H, 0, H, 1), B, 1, H, H, L)
        dw1 = K.sum_rows(part1.view(-1), B, H * H, d1, deferred=defer1).view(H, H, 1, 1)
        # db2[g] = sum dT2[b,g,:]: the same contraction against a column of ones (all strides 0)
        ones = _ones(dev)
        e2, deferb = dst(pb2, H)
        partb = rows_buf("th_db2", H, deferb)
        K.bgemm(dT2, (H * L, 0, L, 1), ones, (0, 0, 0, 0), partb, (H, 0, 1, 1), B, 1, H, 1, L)
        db2 = K.sum_rows(partb.view(-1), B, H, e2, deferred=deferb)
        db1 = K.axpby(b1, None, 0.0, 0.0, out=dst(pb1, H)[0])      # softmax is invariant to a per-head shift: exactly zero
        dth = (dw1, db1, dw2, db2)
    dtable = None
    if need_table:
        dfull = torch.empty(H * L, dtype=torch.float32, device=dev)
        K.sum_rows(dS_buf.view(-1), B, H * L, dfull)
        dtable = K.bias_scatter(dfull.view(H, L), idx, table.shape[1], _slot(table, True, tuple(table.shape)))
    return dQ, dK, dV, dtable, dth


_ones_cache: dict = {}


def _ones(device: torch.device) -> torch.Tensor:
    key = (device.type, device.index)
    t = _ones_cache.get(key)
    if t is None:
        with torch.inference_mode(False):
            t = torch.ones(8, dtype=torch.float32, device=device)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _ones_cache[key] = t                # a tensor born inside a capture belongs to that graph's pool: not cached
    return t


# =========================================================================== stem conv1
@dataclass
class ConvStemCtx:
    stride: int
    pad: int
    bn: BNRef
    dtype: torch.dtype
    training: bool
    act: int = ACT_GELU
    counters: list | None = None


class ConvStemFunction(torch.autograd.Function):
    """conv k3 s2 (3 -> C) + bias + BN + act on the f32 NHWC image; the 3-channel direct kernel of the stem."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, cfg: ConvStemCtx):
        N, H, W, _ = x.shape
        k = weight.shape[2]
        Ho, Wo = (H + 2 * cfg.pad - k) // cfg.stride + 1, (W + 2 * cfg.pad - k) // cfg.stride + 1
        y, parts, n = K.stem_conv_fwd(x, weight, cfg.dtype, cfg.stride, cfg.pad, cfg.pad, Ho, Wo, stats=cfg.training)
        st = _bn_state(parts, n, N * Ho * Wo, cfg.bn, gamma, beta, cfg.training, cfg.counters, conv_bias=bias)
        out = K.bn_act_apply(y, st, cfg.act)
        ctx.cfg = cfg
        ctx.save_for_backward(x, y, st, weight, bias, gamma, beta)
        return out

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        cfg: ConvStemCtx = ctx.cfg
        x, y, st, weight, bias, gamma, beta = ctx.saved_tensors
        need = ctx.needs_input_grad
        C, k = weight.shape[0], weight.shape[2]
        dz, parts, n = K.act_bn_bwd(_c(g), y, None, None, st, cfg.act)
        need_bn = need[3] or need[4]
        outs = (_slot(gamma, need_bn, (C,)), _slot(beta, need_bn, (C,)), None, _slot(bias, need[2], (C,)))
        coef, dgamma, dbeta, _, db = K.bn_bwd_finalize_ex(parts, n, _rows(y), gamma, beta, None, st, cfg.training, need_bn,
                                                          False, need[2], outs)
        dw = None
        if need[1]:
            dw = K.stem_conv_wgrad(x, dz, y, coef, k, cfg.stride, cfg.pad, cfg.pad, _slot(weight, True, tuple(weight.shape)))
        return None, dw, db, dgamma, dbeta, None


# =========================================================================== dense conv + BN (+ act)
def dense_conv_fwd(x, wg_nk, b, gamma, beta, bn: BNRef, training, counters, k, stride):
    N, H, W, C = x.shape
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
    y, parts, n = K.conv_fwd(x, None, ACT_NONE, wg_nk, k, stride, p, Ho, Wo, stats=training)
    st = _bn_state(parts, n, N * Ho * Wo, bn, gamma, beta, training, counters, conv_bias=b)
    return y, st


def dense_conv_bwd(dz, parts, n, x, y, st, wg_kn, w, b, gamma, beta, training, k, stride, need_dx, need_w, need_bn):
    """dz: gradient at the BN output.  The im2col matrix is rebuilt instead of kept (9x the input)."""
    C = gamma.numel()
    p = k // 2
    outs = (_slot(gamma, need_bn, (C,)), _slot(beta, need_bn, (C,)), None, _slot(b, need_w, (C,)))
    coef, dgamma, dbeta, _, db = K.bn_bwd_finalize_ex(parts, n, _rows(y), gamma, beta, None, st, training, need_bn, False,
                                                      need_w and b is not None, outs)
    pro = K.pro_affine2(y, coef)
    dx = dw = None
    if need_dx:
        dcol, _, _ = K.pwconv(dz, pro, wg_kn, None, stats=False)
        dx = K.col2im(dcol, tuple(x.shape), k, stride, p)
    if need_w:
        dwg = K.conv_wgrad(dz, pro, x, None, ACT_NONE, k, stride, p)
        dw = K.conv_wgrad_from_gemm(dwg, tuple(w.shape), _slot(w, True, tuple(w.shape)))
    return dx, dw, db, dgamma, dbeta


def _gemm_weight(w: torch.Tensor, dt: torch.dtype, need_bwd: bool):
    wg = K.conv_weight_to_gemm(w)
    return K.prep_weights(wg, dt, True, need_bwd)


@dataclass
class DenseConvCtx:
    k: int
    stride: int
    bn: BNRef
    training: bool
    act: int = ACT_NONE
    counters: list | None = None


class DenseConvBNFunction(torch.autograd.Function):
    """Dense k x k convolution (+bias) + BN [+ act], output materialised.  x is a materialised NHWC activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, cfg: DenseConvCtx):
        need_bwd = any(ctx.needs_input_grad)
        wg_nk, wg_kn = _gemm_weight(weight, x.dtype, need_bwd)
        y, st = dense_conv_fwd(x, wg_nk, bias, gamma, beta, cfg.bn, cfg.training, cfg.counters, cfg.k, cfg.stride)
        out = K.bn_act_apply(y, st, cfg.act)
        ctx.cfg = cfg
        ctx.save_for_backward(x, y, st, wg_kn, weight, bias, gamma, beta)
        return out

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        cfg: DenseConvCtx = ctx.cfg
        x, y, st, wg_kn, weight, bias, gamma, beta = ctx.saved_tensors
        need = ctx.needs_input_grad
        g = _c(g)
        if cfg.act == ACT_NONE:
            parts, n = K.bn_bwd_reduce(g, y, st, None)
            dz = g
        else:
            dz, parts, n = K.act_bn_bwd(g, y, None, None, st, cfg.act)
        dx, dw, db, dgamma, dbeta = dense_conv_bwd(dz, parts, n, x, y, st, wg_kn, weight, bias, gamma, beta, cfg.training, cfg.k,
                                                   cfg.stride, need[0], need[1], need[3] or need[4])
        return dx, dw, db if need[2] else None, dgamma, dbeta, None


# =========================================================================== ConvMlp block
@dataclass
class ConvMlpCtx:
    bn1: BNRef
    bnd: BNRef
    bn2: BNRef
    training: bool
    counters: list | None = None
    derived: dict | None = None           # see _prep


class ConvMlpFunction(torch.autograd.Function):
    """x + [row_scale *] ls * BN(fc2(GELU(BN(dw3x3(GELU(BN(fc1(x))))))))  — EfficientFormerV2Block's MLP half.

    The same three-kernel chain as MBConvFunction without squeeze-excite: each convolution writes its raw output
    and partial statistics, the consumer applies BN + GELU while loading; biases and LayerScale live in the BN
    coefficient kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, wd, bd, gd, bed, w2, b2, g2, be2, ls, row_scale, cfg: ConvMlpCtx):
        tr = cfg.training
        N, H, W, C = x.shape
        need_bwd = any(ctx.needs_input_grad)
        w1_nk, w1_kn = _prep(w1, x.dtype, cfg.derived, need_bwd)
        y1, parts, n = K.pwconv(x, None, w1_nk, None, stats=tr)
        st1 = _bn_state(parts, n, N * H * W, cfg.bn1, g1, be1, tr, cfg.counters, conv_bias=b1)
        # GELU(BN(y1)) is materialised once: as a prologue of the depthwise kernels it sits between a tile's loads and its LDS
        # stores on every workgroup's critical path — measured per layer (scripts/dw_shapes.py, batch 256): forward 202 -> 97 us
        # and weight gradient 292 -> 171 us at 56 x 56 x 128 against 78 us for this pass; a net gain at every level
        a1 = K.bn_act_apply(y1, st1, ACT_GELU)
        y2, parts, n = K.dwconv_fwd(a1, None, ACT_NONE, wd, 3, 1, 1, 1, H, W, stats=tr)
        st2 = _bn_state(parts, n, N * H * W, cfg.bnd, gd, bed, tr, cfg.counters, conv_bias=bd)
        w2_nk, w2_kn = _prep(w2, x.dtype, cfg.derived, need_bwd)
        a2 = K.bn_act_apply(y2, st2, ACT_GELU)               # likewise for fc2 and its weight gradient (S1: 18.6 -> 18.35 ms)
        y3, parts, n = K.pwconv(a2, None, w2_nk, None, stats=tr)
        st3 = _bn_state(parts, n, N * H * W, cfg.bn2, g2, be2, tr, cfg.counters, conv_bias=b2, ls=ls)
        out = K.bn_act_apply(y3, st3, ACT_NONE, x, row_scale)
        ctx.cfg = cfg
        ctx.has_rs = row_scale is not None
        ctx.save_for_backward(x, y1, a1, y2, a2, y3, st1, st2, st3, w1_kn, w2_kn, w1, b1, g1, be1, wd, bd, gd, bed, w2, b2, g2, be2, ls,
                              row_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        with K.sum_batch():        # the three weight gradients' final sums in one pair of launches at the end
            return ConvMlpFunction._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        cfg: ConvMlpCtx = ctx.cfg
        (x, y1, a1, y2, a2, y3, st1, st2, st3, w1_kn, w2_kn, w1, b1, g1, be1, wd, bd, gd, bed, w2, b2, g2, be2, ls,
         row_scale) = ctx.saved_tensors
        need = ctx.needs_input_grad
        tr = cfg.training
        rows = _rows(y1)
        C, Cm = x.shape[3], y1.shape[3]
        g = _c(g)
        gb = K.scale_rows(g, row_scale) if ctx.has_rs else g
        # ---- fc2's BN (+ LayerScale) backward, folded into the two GEMMs that consume dy3
        parts, n = K.bn_bwd_reduce(gb, y3, st3, None)
        nb2 = need[11] or need[12]
        outs = (_slot(g2, nb2, (C,)), _slot(be2, nb2, (C,)), _slot(ls, need[13], (C,)), _slot(b2, need[10], (C,)))
        coef3, dg2, dbe2, dls, db2 = K.bn_bwd_finalize_ex(parts, n, rows, g2, be2, ls, st3, tr, nb2, need[13], need[10], outs)
        # the mapped gradient [rows][C] is materialised once for its two GEMMs (as a prologue it is re-evaluated per 128-column
        # tile of the Cm-wide data gradient; S1: 18.48 -> 18.35 ms).  The Cm-wide one below stays a prologue (measured neutral)
        gm3 = K.affine2_apply(gb, y3, coef3)
        D, _, _ = K.pwconv(gm3, None, w2_kn, None, stats=False)
        dw2 = None
        if need[9]:
            dw2 = K.pwconv_wgrad(gm3, None, a2, None, _slot(w2, True, (C, Cm))).view(C, Cm, 1, 1)
        # ---- GELU' and the depthwise BN backward
        dz2, parts, n = K.act_bn_bwd(D, y2, None, None, st2, ACT_GELU)
        nbd = need[7] or need[8]
        outs = (_slot(gd, nbd, (Cm,)), _slot(bed, nbd, (Cm,)), None, _slot(bd, need[6], (Cm,)))
        coef2, dgd, dbed, _, dbd = K.bn_bwd_finalize_ex(parts, n, rows, gd, bed, None, st2, tr, nbd, False, need[6], outs)
        dwd = None
        fused = need[5] and K.dwconv_bwd_fused_ok(3, 1)
        if fused:       # data and weight gradient from one staging of (dz2, y2, y1): csrc/dfd_dwbwdf.hip
            dz1, parts, n, dwd = K.dwconv_bwd_fused(dz2, y2, coef2, wd, y1, st1, ACT_GELU, 3, 1, 1, 1, _slot(wd, True, (Cm, 1, 3, 3)))
        else:
            dz1, parts, n = K.dwconv_bwd_data(dz2, y2, coef2, wd, y1, st1, ACT_GELU, tuple(y1.shape), 3, 1, 1, 1)
        nb1 = need[3] or need[4]
        outs = (_slot(g1, nb1, (Cm,)), _slot(be1, nb1, (Cm,)), None, _slot(b1, need[2], (Cm,)))
        coef1, dg1, dbe1, _, db1 = K.bn_bwd_finalize_ex(parts, n, rows, g1, be1, None, st1, tr, nb1, False, need[2], outs)
        if need[5] and not fused:
            dwd = K.dwconv_bwd_weight(dz2, y2, coef2, a1, None, ACT_NONE, 3, 1, 1, 1, _slot(wd, True, (Cm, 1, 3, 3)))
        pro1 = K.pro_affine2(y1, coef1)
        dw1 = dx = None
        if need[0] and need[1] and K.pwconv_bwd_fused_ok(dz1, x):
            # fc1 of the early stages (32 -> 128 at 802,816 rows): both gradients from ONE pass over (dz1, y1) — csrc/dfd_pwtnw.hip, DG
            dx, dw1 = K.pwconv_bwd_fused(dz1, y1, coef1, x, w1_kn, g, _slot(w1, True, (Cm, C)))
            dw1 = dw1.view(Cm, C, 1, 1)
        else:
            if need[1]:
                dw1 = K.pwconv_wgrad(dz1, pro1, x, None, _slot(w1, True, (Cm, C))).view(Cm, C, 1, 1)
            if need[0]:
                dx, _, _ = K.pwconv(dz1, pro1, w1_kn, g, stats=False)
        return (dx, dw1, db1, dg1, dbe1, dwd, dbd, dgd, dbed, dw2, db2, dg2, dbe2, dls, None, None)


# =========================================================================== attention block
@dataclass
class AttentionCtx:
    geo: AttnGeom
    stride: int | None                 # 2: dw3x3 s2 in front, bilinear x2 behind (stage 2)
    bns: dict                          # name -> BNRef for stride_conv, q, k, v, v_local, proj
    idx: torch.Tensor                  # int32 [Nq*Nk] bias index
    training: bool
    counters: list | None = None
    derived: dict | None = None


_ATT_ORDER = ("stride_conv", "q", "k", "v", "v_local", "proj")


class AttentionFunction(torch.autograd.Function):
    """x + [row_scale *] ls1 * Attention2d(x).  Tensor inputs after x, in order:
    for name in (stride_conv?, q, k, v, v_local, proj): weight, bias, bn.weight, bn.bias;
    then talking_head1.weight, .bias, talking_head2.weight, .bias, attention_biases, ls1, row_scale."""

    @staticmethod
    def forward(ctx, x, cfg: AttentionCtx, *t):
        tr, cn, geo = cfg.training, cfg.counters, cfg.geo
        names = [nm for nm in _ATT_ORDER if nm != "stride_conv" or cfg.stride is not None]
        P4 = {nm: t[4 * i: 4 * i + 4] for i, nm in enumerate(names)}
        at = 4 * len(names)
        th_w1, th_b1, th_w2, th_b2, table, ls, row_scale = t[at: at + 7]
        H = geo.heads
        th = (th_w1.reshape(H, H), th_b1, th_w2.reshape(H, H), th_b2)
        dt = x.dtype
        xs, ys, sts = x, None, None
        if cfg.stride is not None:
            w, b, gm, be = P4["stride_conv"]
            ys, sts = dwbn_fwd(x, w, b, gm, be, cfg.bns["stride_conv"], tr, cn, 3, cfg.stride)
            xs = K.bn_act_apply(ys, sts, ACT_NONE)
        kn = {}
        mats = {}
        for nm in ("q", "k", "v"):
            w, b, gm, be = P4[nm]
            w_nk, kn[nm] = _prep(w, dt, cfg.derived)
            mats[nm] = pwbn_fwd(xs, w_nk, b, gm, be, cfg.bns[nm], tr, cn)
        q, yq, stq = mats["q"]
        k, yk, stk = mats["k"]
        v, yv, stv = mats["v"]
        O, S, P, T2 = attn_core_fwd(q, k, v, table, cfg.idx, th, geo)
        w, b, gm, be = P4["v_local"]
        yl, stl = dwbn_fwd(v, w, b, gm, be, cfg.bns["v_local"], tr, cn, 3, 1)
        if cfg.stride is not None:
            s = K.bn_add_act(yl, stl, O, ACT_NONE)
            a = K.up2_act_fwd(s, ACT_GELU)
        else:
            s = None
            a = K.bn_add_act(yl, stl, O, ACT_GELU)
        w, b, gm, be = P4["proj"]
        w_nk, kn["proj"] = _prep(w, dt, cfg.derived)
        out, yp, stp = pwbn_fwd(a, w_nk, b, gm, be, cfg.bns["proj"], tr, cn, ACT_NONE, ls, x, row_scale)
        ctx.cfg = cfg
        ctx.names = names
        ctx.nt = len(t)
        ctx.save_for_backward(x, xs, ys, sts, q, yq, stq, k, yk, stk, v, yv, stv, S, P, T2, O, yl, stl, s, a, yp, stp,
                              kn["q"], kn["k"], kn["v"], kn["proj"], *t)
        return out

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        cfg: AttentionCtx = ctx.cfg
        sv = ctx.saved_tensors
        (x, xs, ys, sts, q, yq, stq, k, yk, stk, v, yv, stv, S, P, T2, O, yl, stl, s, a, yp, stp, knq, knk, knv, knp) = sv[:27]
        t = sv[27:]
        names = ctx.names
        need = ctx.needs_input_grad                       # [x, cfg, *t]
        nt = {nm: need[2 + 4 * i: 2 + 4 * i + 4] for i, nm in enumerate(names)}
        P4 = {nm: t[4 * i: 4 * i + 4] for i, nm in enumerate(names)}
        at = 4 * len(names)
        th_w1, th_b1, th_w2, th_b2, table, ls, row_scale = t[at: at + 7]
        n_th = need[2 + at: 2 + at + 4]
        n_table, n_ls = need[2 + at + 4], need[2 + at + 5]
        tr, geo = cfg.training, cfg.geo
        H = geo.heads
        th = (th_w1.reshape(H, H), th_b1, th_w2.reshape(H, H), th_b2)
        grads: dict = {}
        g = _c(g)
        # ---- proj (1x1 + BN, LayerScale, drop-path scale); its input gradient is needed by everything upstream
        upstream = need[0] or any(any(nt[nm]) for nm in names if nm != "proj") or any(n_th) or n_table
        w, b, gm, be = P4["proj"]
        da, dw, db, dgm, dbe, dls = pwbn_bwd(g, a, yp, stp, knp, tuple(w.shape), w, b, gm, be, ls, ACT_NONE, tr, upstream,
                                             nt["proj"][0], nt["proj"][2] or nt["proj"][3], n_ls, None, row_scale)
        grads["proj"] = (dw, db if nt["proj"][1] else None, dgm, dbe)
        dx = None
        dth = None
        dtable = None
        if upstream:
            # ---- GELU [+ upsample] and the sum O + BN(v_local)
            if cfg.stride is not None:
                d = K.up2_act_bwd(da, s, ACT_GELU)
                parts, n = K.bn_bwd_reduce(d, yl, stl, None)
            else:
                d, parts, n = K.bn_add_act_bwd(da, yl, stl, O, ACT_GELU)
            w, b, gm, be = P4["v_local"]
            nv = nt["v_local"]
            dv_loc, dw, db, dgm, dbe = dwbn_bwd(d, parts, n, v, yl, stl, w, b, gm, be, tr, 3, 1, True, nv[0], nv[2] or nv[3])
            grads["v_local"] = (dw, db if nv[1] else None, dgm, dbe)
            # ---- attention core
            dQ, dK, dV, dtable, dth = attn_core_bwd(d, q, k, v, S, P, T2, table, cfg.idx, th, geo, n_table, any(n_th),
                                                    (th_w1, th_b1, th_w2, th_b2))
            dVt = K.add(dV, dv_loc)
            # ---- q, k, v projections: the gradient of their common input accumulates through `residual`
            need_xs = need[0] or (cfg.stride is not None and any(nt["stride_conv"]))
            acc = g if (cfg.stride is None and need[0]) else None          # the skip connection's share of dx
            for nm, dz, y_, st_, kn_ in (("q", dQ, yq, stq, knq), ("k", dK, yk, stk, knk), ("v", dVt, yv, stv, knv)):
                w, b, gm, be = P4[nm]
                nn_ = nt[nm]
                acc, dw, db, dgm, dbe, _ = pwbn_bwd(dz, xs, y_, st_, kn_, tuple(w.shape), w, b, gm, be, None, ACT_NONE, tr,
                                                    need_xs, nn_[0], nn_[2] or nn_[3], False, acc, None)
                grads[nm] = (dw, db if nn_[1] else None, dgm, dbe)
            if cfg.stride is not None:
                w, b, gm, be = P4["stride_conv"]
                ns = nt["stride_conv"]
                if need_xs:
                    parts, n = K.bn_bwd_reduce(acc, ys, sts, None)
                    dxs, dw, db, dgm, dbe = dwbn_bwd(acc, parts, n, x, ys, sts, w, b, gm, be, tr, 3, cfg.stride, need[0], ns[0],
                                                     ns[2] or ns[3])
                    grads["stride_conv"] = (dw, db if ns[1] else None, dgm, dbe)
                    dx = K.add(dxs, g) if need[0] else None
                else:
                    grads["stride_conv"] = (None, None, None, None)
            else:
                dx = acc if need[0] else None
        else:
            for nm in names:
                grads.setdefault(nm, (None, None, None, None))
        flat = []
        for nm in names:
            flat.extend(grads.get(nm, (None, None, None, None)))
        if dth is not None:
            dw1, db1, dw2, db2 = dth
            flat.extend([dw1 if n_th[0] else None, db1 if n_th[1] else None, dw2 if n_th[2] else None, db2 if n_th[3] else None])
        else:
            flat.extend([None, None, None, None])
        flat.extend([dtable, dls, None])
        return (dx, None, *flat)


# =========================================================================== downsample (conv [+ attention branch])
@dataclass
class DownsampleCtx:
    bn_conv: BNRef
    training: bool
    counters: list | None = None
    attn: bool = False
    geo: AttnGeom | None = None
    bns: dict | None = None            # q_proj, k, v, v_local, proj
    idx: torch.Tensor | None = None
    derived: dict | None = None


_DS_ORDER = ("q_proj", "k", "v", "v_local", "proj")


class DownsampleFunction(torch.autograd.Function):
    """conv3x3 s2 (+bias) + BN, plus (last stage) the Attention2dDownsample branch on the same input.
    Tensor inputs after x: conv.weight, conv.bias, bn.weight, bn.bias; with attention:
    q.local.weight, q.local.bias, then for name in (q_proj, k, v, v_local, proj): weight, bias, bn.weight, bn.bias;
    then attention_biases."""

    @staticmethod
    def forward(ctx, x, cfg: DownsampleCtx, *t):
        tr, cn = cfg.training, cfg.counters
        wc, bc, gc, bec = t[:4]
        need_bwd = any(ctx.needs_input_grad)
        wg_nk, wg_kn = _gemm_weight(wc, x.dtype, need_bwd)
        yc, stc = dense_conv_fwd(x, wg_nk, bc, gc, bec, cfg.bn_conv, tr, cn, 3, 2)
        conv_out = K.bn_act_apply(yc, stc, ACT_NONE)
        ctx.cfg = cfg
        if not cfg.attn:
            ctx.save_for_backward(x, yc, stc, wg_kn, *t)
            return conv_out
        geo = cfg.geo
        wl, bl = t[4:6]
        P4 = {nm: t[6 + 4 * i: 6 + 4 * i + 4] for i, nm in enumerate(_DS_ORDER)}
        table = t[6 + 4 * len(_DS_ORDER)]
        dt = x.dtype
        N, H, W, C = x.shape
        lq, _, _ = K.dwconv_fwd(x, None, ACT_NONE, wl, 3, 2, 1, 1, (H + 1) // 2, (W + 1) // 2, stats=False)
        qin = K.subsample_add(lq, bl, x, 2)
        kn = {}
        w, b, gm, be = P4["q_proj"]
        w_nk, kn["q_proj"] = _prep(w, dt, cfg.derived)
        q, yq, stq = pwbn_fwd(qin, w_nk, b, gm, be, cfg.bns["q_proj"], tr, cn)
        w, b, gm, be = P4["k"]
        w_nk, kn["k"] = _prep(w, dt, cfg.derived)
        k, yk, stk = pwbn_fwd(x, w_nk, b, gm, be, cfg.bns["k"], tr, cn)
        w, b, gm, be = P4["v"]
        w_nk, kn["v"] = _prep(w, dt, cfg.derived)
        v, yv, stv = pwbn_fwd(x, w_nk, b, gm, be, cfg.bns["v"], tr, cn)
        O, S, P, T2 = attn_core_fwd(q, k, v, table, cfg.idx, None, geo)
        w, b, gm, be = P4["v_local"]
        yl, stl = dwbn_fwd(v, w, b, gm, be, cfg.bns["v_local"], tr, cn, 3, 2)
        a = K.bn_add_act(yl, stl, O, ACT_GELU)
        w, b, gm, be = P4["proj"]
        w_nk, kn["proj"] = _prep(w, dt, cfg.derived)
        out, yp, stp = pwbn_fwd(a, w_nk, b, gm, be, cfg.bns["proj"], tr, cn, ACT_NONE, None, conv_out, None)
        ctx.save_for_backward(x, yc, stc, wg_kn, qin, q, yq, stq, k, yk, stk, v, yv, stv, S, P, O, yl, stl, a, yp, stp,
                              kn["q_proj"], kn["k"], kn["v"], kn["proj"], *t)
        return out

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        cfg: DownsampleCtx = ctx.cfg
        sv = ctx.saved_tensors
        need = ctx.needs_input_grad                   # [x, cfg, *t]
        tr = cfg.training
        g = _c(g)
        if not cfg.attn:
            x, yc, stc, wg_kn = sv[:4]
            wc, bc, gc, bec = sv[4:8]
            parts, n = K.bn_bwd_reduce(g, yc, stc, None)
            dx, dw, db, dgm, dbe = dense_conv_bwd(g, parts, n, x, yc, stc, wg_kn, wc, bc, gc, bec, tr, 3, 2, need[0], need[2],
                                                  need[4] or need[5])
            return (dx, None, dw, db if need[3] else None, dgm, dbe)
        (x, yc, stc, wg_kn, qin, q, yq, stq, k, yk, stk, v, yv, stv, S, P, O, yl, stl, a, yp, stp, knq, knk, knv, knp) = sv[:26]
        t = sv[26:]
        wc, bc, gc, bec = t[:4]
        wl, bl = t[4:6]
        P4 = {nm: t[6 + 4 * i: 6 + 4 * i + 4] for i, nm in enumerate(_DS_ORDER)}
        nt = {nm: need[2 + 6 + 4 * i: 2 + 6 + 4 * i + 4] for i, nm in enumerate(_DS_ORDER)}
        table = t[6 + 4 * len(_DS_ORDER)]
        n_table = need[2 + 6 + 4 * len(_DS_ORDER)]
        n_local = need[2 + 4: 2 + 6]
        geo = cfg.geo
        grads = {}
        # ---- conv branch (its BN output was the `residual` of the attention branch's last apply: gradient g)
        parts, n = K.bn_bwd_reduce(g, yc, stc, None)
        dx, dwc, dbc, dgc, dbec = dense_conv_bwd(g, parts, n, x, yc, stc, wg_kn, wc, bc, gc, bec, tr, 3, 2, need[0], need[2],
                                                 need[4] or need[5])
        # ---- attention branch
        w, b, gm, be = P4["proj"]
        da, dw, db, dgm, dbe, _ = pwbn_bwd(g, a, yp, stp, knp, tuple(w.shape), w, b, gm, be, None, ACT_NONE, tr, True,
                                           nt["proj"][0], nt["proj"][2] or nt["proj"][3])
        grads["proj"] = (dw, db if nt["proj"][1] else None, dgm, dbe)
        d, parts, n = K.bn_add_act_bwd(da, yl, stl, O, ACT_GELU)
        w, b, gm, be = P4["v_local"]
        nv = nt["v_local"]
        dv_loc, dw, db, dgm, dbe = dwbn_bwd(d, parts, n, v, yl, stl, w, b, gm, be, tr, 3, 2, True, nv[0], nv[2] or nv[3])
        grads["v_local"] = (dw, db if nv[1] else None, dgm, dbe)
        dQ, dK, dV, dtable, _ = attn_core_bwd(d, q, k, v, S, P, P, table, cfg.idx, None, geo, n_table, False)
        dVt = K.add(dV, dv_loc)
        acc = dx                                       # running sum of dx over the consumers of x
        for nm, dz, y_, st_, kn_ in (("k", dK, yk, stk, knk), ("v", dVt, yv, stv, knv)):
            w, b, gm, be = P4[nm]
            nn_ = nt[nm]
            acc, dw, db, dgm, dbe, _ = pwbn_bwd(dz, x, y_, st_, kn_, tuple(w.shape), w, b, gm, be, None, ACT_NONE, tr, need[0],
                                                nn_[0], nn_[2] or nn_[3], False, acc, None)
            grads[nm] = (dw, db if nn_[1] else None, dgm, dbe)
        w, b, gm, be = P4["q_proj"]
        nq = nt["q_proj"]
        need_qin = need[0] or any(n_local)
        dqin, dw, db, dgm, dbe, _ = pwbn_bwd(dQ, qin, yq, stq, knq, tuple(w.shape), w, b, gm, be, None, ACT_NONE, tr, need_qin,
                                             nq[0], nq[2] or nq[3])
        grads["q_proj"] = (dw, db if nq[1] else None, dgm, dbe)
        dwl = dbl = None
        C = x.shape[3]
        if need_qin:
            if need[0]:
                dxl, _, _ = K.dwconv_bwd_data(dqin, None, None, wl, None, None, ACT_NONE, tuple(x.shape), 3, 2, 1, 1)
                K.subsample_add_bwd(dqin, dxl, 2)                  # the AvgPool2d(1, 2) branch
                acc = K.add(acc, dxl) if acc is not None else dxl
            if n_local[0]:
                dwl = K.dwconv_bwd_weight(dqin, None, None, x, None, ACT_NONE, 3, 2, 1, 1, _slot(wl, True, (C, 1, 3, 3)))
            if n_local[1]:
                parts, n = K.channel_stats(dqin)
                both = torch.empty(2 * C, dtype=torch.float32, device=x.device)
                K.sum_rows(parts, n, 2 * C, both)
                slot = _slot(bl, True, (C,))
                dbl = K.axpby(both[:C], None, 1.0, 0.0, out=slot) if slot is not None else both[:C]
        flat = [dwc if need[2] else None, dbc if need[3] else None, dgc, dbec, dwl, dbl]
        for nm in _DS_ORDER:
            flat.extend(grads[nm])
        flat.append(dtable)
        return (acc if need[0] else None, None, *flat)


# =========================================================================== tail
@dataclass
class TailCtx:
    bn: BNRef
    dropout: float
    training: bool
    counters: list | None = None


class TailFunction(torch.autograd.Function):
    """BN -> mean over (H, W) -> dropout -> (head(x) + head_dist(x)) / 2 -> f32 logits
    (EfficientFormerV2.forward_features' norm + forward_head with distillation averaging)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w_h, b_h, w_d, b_d, drop_u, cfg: TailCtx):
        N, H, W, C = x.shape
        tr = cfg.training
        parts, n = K.channel_stats(x) if tr else (None, 0)
        st = _bn_state(parts, n, N * H * W, cfg.bn, gamma, beta, tr, cfg.counters)
        pooled = K.pool_act(x, st, ACT_NONE)
        feat = K.dropout(pooled, drop_u, cfg.dropout) if drop_u is not None else pooled
        if w_d is None:                       # a single classifier (FasterViT); two heads averaged: EfficientFormerV2
            logits = K.linear_fwd(feat, w_h, b_h)
        else:
            logits = K.axpby(K.linear_fwd(feat, w_h, b_h), K.linear_fwd(feat, w_d, b_d), 0.5, 0.5)
        ctx.cfg = cfg
        ctx.save_for_backward(x, st, feat, gamma, beta, w_h, b_h, w_d, b_d, drop_u)
        return logits

    @staticmethod
    @K.batched_sums
    def backward(ctx, dlogits):
        cfg: TailCtx = ctx.cfg
        x, st, feat, gamma, beta, w_h, b_h, w_d, b_d, drop_u = ctx.saved_tensors
        need = ctx.needs_input_grad
        N, H, W, C = x.shape
        J = w_h.shape[0]
        backbone = need[0] or need[1] or need[2]
        two = w_d is not None
        half = K.axpby(_c(dlogits.float()), None, 0.5, 0.0) if two else _c(dlogits.float())
        d1, dw_h, db_h = K.linear_bwd(half, feat, w_h, backbone, need[3], need[4], _slot(w_h, need[3], (J, C)), _slot(b_h, need[4], (J,)))
        d2 = dw_d = db_d = None
        if two:
            d2, dw_d, db_d = K.linear_bwd(half, feat, w_d, backbone, need[5], need[6], _slot(w_d, need[5], (J, C)), _slot(b_d, need[6], (J,)))
        dx = dgamma = dbeta = None
        if backbone:
            dfeat = K.axpby(d1, d2, 1.0, 1.0) if two else d1
            dpooled = K.dropout(dfeat, drop_u, cfg.dropout) if drop_u is not None else dfeat
            dz, parts, n = K.act_bn_bwd(None, x, None, dpooled, st, ACT_NONE)
            nb = need[1] or need[2]
            coef, dgamma, dbeta, _, _ = K.bn_bwd_finalize_ex(parts, n, N * H * W, gamma, beta, None, st, cfg.training, nb, False,
                                                             False, (_slot(gamma, nb, (C,)), _slot(beta, nb, (C,)), None, None))
            if need[0]:
                dx = K.affine2_apply(dz, x, coef)
        return dx, dgamma, dbeta, dw_h, db_h, dw_d, db_d, None, None


__all__ = ["AttentionCtx", "AttentionFunction", "AttnGeom", "ConvMlpCtx", "ConvMlpFunction", "ConvStemCtx", "ConvStemFunction",
           "DenseConvBNFunction", "DenseConvCtx", "DownsampleCtx", "DownsampleFunction", "TailCtx", "TailFunction"]
