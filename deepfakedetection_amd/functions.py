"""Fused autograd functions of the HIP engine: stem, MBConv block, classifier head, loss.

One torch.autograd.Function per network stage.  Inside a stage the tensors that an
unfused framework would write to HBM after every BatchNorm / SiLU / SE multiply never
exist: each conv kernel writes its RAW output plus per-channel partial sums, a tiny
finalize kernel turns the sums into (scale, shift, mean, rstd), and the NEXT kernel
applies normalisation + activation (+ SE gate) while it loads its operand.  Backward
re-derives activations from the saved raw tensors the same way and uses the identity
    d(conv out) = a[c]*dz + b[c]*y + c[c]
for training-mode BatchNorm, so BN-backward is also folded into the consumers.

Stage arithmetic mirrors efficientnet_pytorch 0.7.1 `MBConvBlock.forward` /
timm 1.0.20 `InvertedResidual.forward` (reference call sites:
trainers/efficientnet.py:297,302; orchestration/orchestrator.py:529,590).
"""

from __future__ import annotations

import contextlib
import os
import threading
from dataclasses import dataclass

import torch

from . import kernels as K
from ._lib import ACT_NONE, ACT_SILU
from .arch import ConvGeom
from .arena import grad_dest


@dataclass
class BNRef:
    """A BatchNorm2d's buffers + hyper-parameters (its weight/bias travel as Function inputs)."""

    running_mean: torch.Tensor
    running_var: torch.Tensor
    num_batches_tracked: torch.Tensor | None
    momentum: float
    eps: float
    pre: torch.Tensor | None = None       # eval mode: this layer's coefficient block, already computed (kernels.EvalBNStates)

    def params(self, weight: torch.Tensor, bias: torch.Tensor) -> K.BNParams:
        return K.BNParams(weight, bias, self.running_mean, self.running_var, self.momentum, self.eps)


def _bn_state(parts, nparts, count: int, bn: BNRef, weight, bias, training: bool, counters: list | None = None,
              conv_bias=None, ls=None) -> torch.Tensor:
    """BatchNorm coefficients of one layer.  `counters`: the list owned by the calling network's forward pass; the
    layer's num_batches_tracked is appended and the network bumps all of them with ONE launch at the end
    (kernels.DeviceRng.tick).  A stage used on its own (counters None) bumps its counter itself."""
    params = bn.params(weight, bias)
    params.conv_bias, params.ls = conv_bias, ls
    if training:
        if bn.num_batches_tracked is not None:
            if counters is not None:
                counters.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
        return K.bn_finalize(parts, nparts, count, params)
    if bn.pre is not None and conv_bias is None and ls is None:
        return bn.pre
    batch = getattr(_bn_batch_tls, "batch", None)
    if batch is not None:
        return batch.request(params)
    return K.bn_eval_coeffs(params)


_bn_batch_tls = threading.local()


def eval_fused_enabled(elems: int, dtype: torch.dtype) -> bool:
    """Whether an MBConv block whose depthwise input has `elems` elements of `dtype` runs the inference form
    (MBConvFunction._forward_eval).  Measured (profiles/r02_eval_forms.md): the SiLU of a producer epilogue is
    exposed VALU time (two half-rate transcendentals per element), the same SiLU in a consumer's staging phase hides under
    its loads — so the inference form only wins where launches, not VALU, bound the block: f32 tensors up to 32 M elements
    (batch <= 64 at 224 px: -2..3 % replayed, -6 % eager); in bf16 it never does.  DFD_EVAL_FUSED=0 / 1 forces the
    training-form chain / the inference form (A/B switch of scripts/eval_small.py and of the parity test)."""
    flag = os.environ.get("DFD_EVAL_FUSED")
    if flag is not None:
        return flag != "0"
    return dtype == torch.float32 and elems <= int(os.environ.get("DFD_EVAL_FUSED_MAX_ELEMS", EVAL_FUSED_MAX_ELEMS))


EVAL_FUSED_MAX_ELEMS = 32 << 20
# DFD_SE_PASSENGER=0: the squeeze-excite FC weight gradients as a launch of their own (A/B switch)
SE_WGRAD_PASSENGER = os.environ.get("DFD_SE_PASSENGER", "1") != "0"


@contextlib.contextmanager
def bn_eval_batch(owner: dict, tag):
    """Inside the block, eval-mode BatchNorm coefficient blocks come from `owner`'s kernels.BNEvalBatch for `tag` (one per
    network and mode): recorded on the first pass, one batched launch from the second pass on."""
    batches = owner.setdefault("_bn_eval_batches", {})
    batch = batches.get(tag)
    if batch is None:
        batch = batches[tag] = K.BNEvalBatch()
    prev = getattr(_bn_batch_tls, "batch", None)
    _bn_batch_tls.batch = batch
    batch.begin()
    try:
        yield batch
    finally:
        _bn_batch_tls.batch = prev
        batch.end()


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _ptrs(*tensors) -> tuple[int, ...]:
    """Addresses of a Function's tensor inputs: keys into the gradient arena."""
    return tuple(t.data_ptr() if isinstance(t, torch.Tensor) else 0 for t in tensors)


def _dest(ctx, index: int, shape):
    """Arena slot for input `index` if that input wants a gradient and owns a free slot."""
    if not ctx.needs_input_grad[index] or not ctx.pptr[index]:
        return None
    return grad_dest(ctx.pptr[index], shape)


# =========================================================================== stem
@dataclass
class StemCtx:
    geom: ConvGeom
    bn: BNRef
    dtype: torch.dtype
    training: bool
    counters: list | None = None          # see _bn_state


class StemFunction(torch.autograd.Function):
    """conv k3 s2 (3 -> C) + BN + SiLU.  x: [N,H,W,3] f32."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, cfg: StemCtx):
        g = cfg.geom
        N, H, W, _ = x.shape
        Ho, Wo = g.out_size(H), g.out_size(W)
        y, parts, n = K.stem_conv_fwd(x, weight, cfg.dtype, g.stride, g.pad_lead, g.pad_lead, Ho, Wo, stats=cfg.training)
        st = _bn_state(parts, n, N * Ho * Wo, cfg.bn, gamma, beta, cfg.training, cfg.counters)
        out = K.bn_act_apply(y, st, ACT_SILU)
        ctx.cfg = cfg
        ctx.pptr = _ptrs(x, weight, gamma, beta)
        ctx.save_for_backward(x, y, st, gamma)
        return out

    @staticmethod
    def backward(ctx, g):
        cfg: StemCtx = ctx.cfg
        x, y, st, gamma = ctx.saved_tensors
        geom = cfg.geom
        N, Ho, Wo, _ = y.shape
        dz, parts, n = K.act_bn_bwd(_c(g), y, None, None, st, ACT_SILU)
        want_bn = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        C = gamma.numel()
        coef, dgamma, dbeta = K.bn_bwd_finalize(parts, n, N * Ho * Wo, gamma, st, cfg.training, want_bn,
                                                _dest(ctx, 2, (C,)), _dest(ctx, 3, (C,)))
        dw = None
        if ctx.needs_input_grad[1]:
            dw = K.stem_conv_wgrad(x, dz, y, coef, geom.kernel, geom.stride, geom.pad_lead, geom.pad_lead,
                                   _dest(ctx, 1, (C, 3, geom.kernel, geom.kernel)))
        return None, dw, dgamma, dbeta, None


# =========================================================================== MBConv
@dataclass
class MBConvCtx:
    expand: bool
    dw: ConvGeom
    skip: bool
    bn_expand: BNRef | None
    bn_dw: BNRef
    bn_project: BNRef
    training: bool
    # derived weights prepared for the whole network in one launch (kernels.DerivedWeights):
    # ((wexp_nk, wexp_kn) | None, (wproj_nk, wproj_kn), se_w2t); None: prepare per layer
    derived: tuple | None = None
    counters: list | None = None          # see _bn_state
    # autograd state at the call site (inside Function.forward grad mode is always off, and needs_input_grad reflects
    # requires_grad alone): False under no_grad / inference_mode, where nothing has to be kept for a backward pass
    grad_enabled: bool = True


class MBConvFunction(torch.autograd.Function):
    """[1x1 expand + BN + SiLU] -> dw kxk + BN + SiLU -> SE -> 1x1 project + BN [-> *mask + x].

    Tensor inputs (None where a block has no expand conv):
      x, w_exp, g_exp, b_exp, w_dw, g_dw, b_dw, se_w1, se_b1, se_w2, se_b2, w_proj, g_proj, b_proj, row_scale
    """

    @staticmethod
    def forward(ctx, x, w_exp, g_exp, b_exp, w_dw, g_dw, b_dw, se_w1, se_b1, se_w2, se_b2, w_proj, g_proj, b_proj,
                row_scale, cfg: MBConvCtx):
        tr = cfg.training
        N, H, W, Cin = x.shape
        dt = x.dtype
        geom = cfg.dw
        Ho, Wo = geom.out_size(H), geom.out_size(W)
        need_bwd = any(ctx.needs_input_grad)   # grad mode is off inside forward(); this is the autograd view
        if not tr and not (need_bwd and cfg.grad_enabled) and eval_fused_enabled(
                N * H * W * (w_exp.shape[0] if cfg.expand else Cin), dt):
            return MBConvFunction._forward_eval(x, w_exp, g_exp, b_exp, w_dw, g_dw, b_dw, se_w1, se_b1, se_w2, se_b2, w_proj,
                                                g_proj, b_proj, cfg, Ho, Wo)
        if cfg.expand:
            wexp_nk, wexp_kn = cfg.derived[0] if cfg.derived is not None else K.prep_weights(w_exp, dt, True, need_bwd)
            y1, parts, n = K.pwconv(x, None, wexp_nk, None, stats=tr)
            st1 = _bn_state(parts, n, N * H * W, cfg.bn_expand, g_exp, b_exp, tr, cfg.counters)
            y2, parts, n = K.dwconv_fwd(y1, st1, ACT_SILU, w_dw, geom.kernel, geom.stride, geom.pad_lead, geom.pad_lead,
                                        Ho, Wo, stats=tr)
        else:
            wexp_kn, y1, st1 = None, None, None
            y2, parts, n = K.dwconv_fwd(x, None, ACT_NONE, w_dw, geom.kernel, geom.stride, geom.pad_lead, geom.pad_lead,
                                        Ho, Wo, stats=tr)
        st2 = _bn_state(parts, n, N * Ho * Wo, cfg.bn_dw, g_dw, b_dw, tr, cfg.counters)
        w1, w2 = se_w1.reshape(se_w1.shape[0], -1), se_w2.reshape(se_w2.shape[0], -1)
        pooled, hpre, gate, w2t = K.se_fwd(y2, st2, ACT_SILU, w1, se_b1, w2, se_b2, ACT_SILU,
                                           cfg.derived[2] if cfg.derived is not None else None)
        wproj_nk, wproj_kn = cfg.derived[1] if cfg.derived is not None else K.prep_weights(w_proj, dt, True, need_bwd)
        pro = K.pro_bn_act_gate(st2, ACT_SILU, gate, Ho * Wo)
        y3, parts, n = K.pwconv(y2, pro, wproj_nk, None, stats=tr)
        st3 = _bn_state(parts, n, N * Ho * Wo, cfg.bn_project, g_proj, b_proj, tr, cfg.counters)
        out = K.bn_act_apply(y3, st3, ACT_NONE, x if cfg.skip else None, row_scale if cfg.skip else None)
        ctx.cfg = cfg
        ctx.pptr = _ptrs(x, w_exp, g_exp, b_exp, w_dw, g_dw, b_dw, se_w1, se_b1, se_w2, se_b2, w_proj, g_proj, b_proj)
        ctx.in_shape = (N, H, W, Cin)
        ctx.has_rs = cfg.skip and row_scale is not None
        ctx.save_for_backward(x, y1, y2, y3, st1, st2, st3, pooled, hpre, gate, wexp_kn, wproj_kn, w_dw, w1, w2t,
                              g_exp, g_dw, g_proj, row_scale if ctx.has_rs else None)
        return out

    @staticmethod
    def _forward_eval(x, w_exp, g_exp, b_exp, w_dw, g_dw, b_dw, se_w1, se_b1, se_w2, se_b2, w_proj, g_proj, b_proj, cfg, Ho, Wo):
        """Inference form (running statistics, nothing kept for a backward pass): every BatchNorm is an affine map known up
        front, so each PRODUCER stores its activated output and the depthwise kernel also leaves the squeeze-excite channel
        sums — no consumer prologues, no pooling pass.  Per stage the values are those of the reference's autocast graph
        (conv -> bn -> SiLU each rounded to the activation dtype; timm InvertedResidual.forward / efficientnet_pytorch
        MBConvBlock.forward)."""
        N, H, W, Cin = x.shape
        dt = x.dtype
        geom = cfg.dw
        a1 = x
        if cfg.expand:
            wexp_nk, _ = cfg.derived[0] if cfg.derived is not None else K.prep_weights(w_exp, dt, True, False)
            st1 = _bn_state(None, 0, N * H * W, cfg.bn_expand, g_exp, b_exp, False, cfg.counters)
            a1 = K.pwconv_eval(x, wexp_nk, st1, ACT_SILU)
        st2 = _bn_state(None, 0, N * Ho * Wo, cfg.bn_dw, g_dw, b_dw, False, cfg.counters)
        a2, pool_parts, _ = K.dwconv_eval(a1, w_dw, st2, ACT_SILU, geom.kernel, geom.stride, geom.pad_lead, geom.pad_lead, Ho, Wo)
        w1, w2 = se_w1.reshape(se_w1.shape[0], -1), se_w2.reshape(se_w2.shape[0], -1)
        _, gate, _ = K.se_fwd_parts(pool_parts, Ho * Wo, w1, se_b1, w2, se_b2, ACT_SILU,
                                    cfg.derived[2] if cfg.derived is not None else None)
        wproj_nk, _ = cfg.derived[1] if cfg.derived is not None else K.prep_weights(w_proj, dt, True, False)
        pro = K.pro_bn_act_gate(K.identity_state(x.device, a2.shape[3]), ACT_NONE, gate, Ho * Wo)     # gate only
        y3, _, _ = K.pwconv(a2, pro, wproj_nk, None, stats=False)
        st3 = _bn_state(None, 0, N * Ho * Wo, cfg.bn_project, g_proj, b_proj, False, cfg.counters)
        return K.bn_act_apply(y3, st3, ACT_NONE, x if cfg.skip else None, None)

    @staticmethod
    def backward(ctx, g):
        cfg: MBConvCtx = ctx.cfg
        (x, y1, y2, y3, st1, st2, st3, pooled, hpre, gate, wexp_kn, wproj_kn, w_dw, w1, w2t,
         g_exp, g_dw, g_proj, row_scale) = ctx.saved_tensors
        need = ctx.needs_input_grad
        tr = cfg.training
        geom = cfg.dw
        N, H, W, Cin = ctx.in_shape
        _, Ho, Wo, Cmid = y2.shape
        with K.sum_batch():        # the three weight gradients' final sums: one pair of launches at the end of the block
            return MBConvFunction._backward(ctx, g, cfg, x, y1, y2, y3, st1, st2, st3, pooled, hpre, gate, wexp_kn, wproj_kn,
                                            w_dw, w1, w2t, g_exp, g_dw, g_proj, row_scale, need, tr, geom, N, H, W, Cin, Ho, Wo, Cmid)

    @staticmethod
    def _backward(ctx, g, cfg, x, y1, y2, y3, st1, st2, st3, pooled, hpre, gate, wexp_kn, wproj_kn, w_dw, w1, w2t, g_exp, g_dw,
                  g_proj, row_scale, need, tr, geom, N, H, W, Cin, Ho, Wo, Cmid):
        g = _c(g)
        gb = K.scale_rows(g, row_scale) if ctx.has_rs else g
        # ---- project BN backward, folded into the two GEMMs that consume dy3
        parts, n = K.bn_bwd_reduce(gb, y3, st3, None)
        Cout, R = y3.shape[3], w1.shape[0]
        coef3, dg_proj, db_proj = K.bn_bwd_finalize(parts, n, N * Ho * Wo, g_proj, st3, tr, need[12] or need[13],
                                                    _dest(ctx, 12, (Cout,)), _dest(ctx, 13, (Cout,)))
        # the BN-backward-mapped gradient [rows][Cout] is materialised once for its two GEMMs: as a prologue it is re-evaluated
        # (and y3 re-read) per 128-column tile of the Cmid-wide data gradient (B0: 13.99 -> 13.90 ms per step; same bits)
        gm3 = K.affine2_apply(gb, y3, coef3)
        D, _, _ = K.pwconv(gm3, None, wproj_kn, None, stats=False)                   # d(act*gate) [.., Cmid]
        dw_proj = None
        if need[11]:
            pro_q = K.pro_bn_act_gate(st2, ACT_SILU, gate, Ho * Wo)
            with K.side_stream(N * Ho * Wo):
                dw_proj = K.pwconv_wgrad(gm3, None, y2, pro_q, _dest(ctx, 11, (Cout, Cmid))).view(Cout, Cmid, 1, 1)
        # ---- squeeze-excite backward
        want_se = need[7] or need[8] or need[9] or need[10]
        se_outs = (_dest(ctx, 7, (R, Cmid)), _dest(ctx, 8, (R,)), _dest(ctx, 9, (Cmid, R)), _dest(ctx, 10, (Cmid,)))
        # the FC weight gradients (read by the optimizer only) ride along with the next launch instead of being one of their own
        se_res = K.se_bwd(D, y2, st2, ACT_SILU, gate, hpre, pooled, w1, w2t, ACT_SILU, want_se, se_outs, defer_wgrad=SE_WGRAD_PASSENGER)
        dpooled, dw1, db1, dw2, db2 = se_res[:5]
        se_job = se_res[5] if SE_WGRAD_PASSENGER else None
        # ---- SiLU' and depthwise BN backward
        dz2, parts, n = K.act_bn_bwd(D, y2, gate, dpooled, st2, ACT_SILU, se_job=se_job)
        if dw1 is not None:
            dw1, dw2 = dw1.view(dw1.shape[0], -1, 1, 1), dw2.view(dw2.shape[0], -1, 1, 1)
        coef2, dg_dw, db_dw = K.bn_bwd_finalize(parts, n, N * Ho * Wo, g_dw, st2, tr, need[5] or need[6],
                                                _dest(ctx, 5, (Cmid,)), _dest(ctx, 6, (Cmid,)))
        kk = geom.kernel
        dw_dw = dx = dw_exp = dg_exp = db_exp = None
        if cfg.expand:
            fused = need[4] and K.dwconv_bwd_fused_ok(geom.kernel, geom.stride)
            if fused:
                # 3x3 stride 1: data and weight gradient from one staging of (dz2, y2, y1) — csrc/dfd_dwbwdf.hip
                dz1, parts, n, dw_dw = K.dwconv_bwd_fused(dz2, y2, coef2, w_dw, y1, st1, ACT_SILU, geom.kernel, geom.stride,
                                                          geom.pad_lead, geom.pad_lead, _dest(ctx, 4, (Cmid, 1, kk, kk)))
            else:
                dz1, parts, n = K.dwconv_bwd_data(dz2, y2, coef2, w_dw, y1, st1, ACT_SILU, y1.shape, geom.kernel,
                                                  geom.stride, geom.pad_lead, geom.pad_lead)
            coef1, dg_exp, db_exp = K.bn_bwd_finalize(parts, n, N * H * W, g_exp, st1, tr, need[2] or need[3],
                                                      _dest(ctx, 2, (Cmid,)), _dest(ctx, 3, (Cmid,)))
            pro_dy1 = K.pro_affine2(y1, coef1)
            # blocks 1-3: data and weight gradient of the expand layer from ONE pass over (dz1, y1), the widest tensors of the
            # network (csrc/dfd_pwtnw.hip, DG) — bit-identical to the two kernels below
            both = K.pwconv_bwd_fused(dz1, y1, coef1, x, wexp_kn, g if cfg.skip else None, _dest(ctx, 1, (Cmid, Cin))) \
                if (need[0] and need[1] and K.pwconv_bwd_fused_ok(dz1, x)) else None
            with K.side_stream(N * Ho * Wo):
                if need[4] and not fused:
                    dw_dw = K.dwconv_bwd_weight(dz2, y2, coef2, y1, st1, ACT_SILU, geom.kernel, geom.stride,
                                                geom.pad_lead, geom.pad_lead, _dest(ctx, 4, (Cmid, 1, kk, kk)))
                if need[1] and both is None:
                    dw_exp = K.pwconv_wgrad(dz1, pro_dy1, x, None, _dest(ctx, 1, (Cmid, Cin))).view(Cmid, Cin, 1, 1)
            if both is not None:
                dx, dw_exp = both[0], both[1].view(Cmid, Cin, 1, 1)
            elif need[0]:
                dx, _, _ = K.pwconv(dz1, pro_dy1, wexp_kn, g if cfg.skip else None, stats=False)
        else:
            if need[4]:
                with K.side_stream(N * Ho * Wo):
                    dw_dw = K.dwconv_bwd_weight(dz2, y2, coef2, x, None, ACT_NONE, geom.kernel, geom.stride,
                                                geom.pad_lead, geom.pad_lead, _dest(ctx, 4, (Cmid, 1, kk, kk)))
            if need[0]:
                dx, _, _ = K.dwconv_bwd_data(dz2, y2, coef2, w_dw, None, None, ACT_NONE, ctx.in_shape, geom.kernel,
                                             geom.stride, geom.pad_lead, geom.pad_lead)
                if cfg.skip:
                    dx = K.add(dx, g)
        K.join_side()
        return (dx, dw_exp, dg_exp, db_exp, dw_dw, dg_dw, db_dw, dw1, db1, dw2, db2, dw_proj, dg_proj, db_proj,
                None, None)


# =========================================================================== head
@dataclass
class HeadCtx:
    bn: BNRef
    dropout: float
    training: bool
    derived: tuple | None = None          # (w_nk, w_kn) from kernels.DerivedWeights, or None
    counters: list | None = None          # see _bn_state


class HeadFunction(torch.autograd.Function):
    """1x1 conv + BN + SiLU -> global average pool -> dropout -> Linear.  Returns f32 logits."""

    @staticmethod
    def forward(ctx, x, w_head, gamma, beta, w_fc, b_fc, drop_u, cfg: HeadCtx):
        N, H, W, Cin = x.shape
        need_bwd = any(ctx.needs_input_grad)
        w_nk, w_kn = cfg.derived if cfg.derived is not None else K.prep_weights(w_head, x.dtype, True, need_bwd)
        y, parts, n = K.pwconv(x, None, w_nk, None, stats=cfg.training)
        st = _bn_state(parts, n, N * H * W, cfg.bn, gamma, beta, cfg.training, cfg.counters)
        pooled = K.pool_act(y, st, ACT_SILU)
        feat = K.dropout(pooled, drop_u, cfg.dropout) if drop_u is not None else pooled
        logits = K.linear_fwd(feat, w_fc, b_fc)
        ctx.cfg = cfg
        ctx.pptr = _ptrs(x, w_head, gamma, beta, w_fc, b_fc)
        ctx.has_bias = b_fc is not None
        ctx.save_for_backward(x, y, st, feat, w_kn, w_fc, gamma, drop_u)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        cfg: HeadCtx = ctx.cfg
        x, y, st, feat, w_kn, w_fc, gamma, drop_u = ctx.saved_tensors
        need = ctx.needs_input_grad
        N, H, W, Cin = x.shape
        Chead = y.shape[3]
        backbone = need[0] or need[1] or need[2] or need[3]
        J = w_fc.shape[0]
        dfeat, dw_fc, db_fc = K.linear_bwd(_c(dlogits.float()), feat, w_fc, backbone, need[4], ctx.has_bias and need[5],
                                           _dest(ctx, 4, (J, Chead)), _dest(ctx, 5, (J,)) if ctx.has_bias else None)
        dx = dw_head = dgamma = dbeta = None
        if backbone:
            dpooled = K.dropout(dfeat, drop_u, cfg.dropout) if drop_u is not None else dfeat
            dz, parts, n = K.act_bn_bwd(None, y, None, dpooled, st, ACT_SILU)
            coef, dgamma, dbeta = K.bn_bwd_finalize(parts, n, N * H * W, gamma, st, cfg.training, need[2] or need[3],
                                                    _dest(ctx, 2, (Chead,)), _dest(ctx, 3, (Chead,)))
            pro = K.pro_affine2(y, coef)
            if need[0]:
                dx, _, _ = K.pwconv(dz, pro, w_kn, None, stats=False)
            if need[1]:
                dw_head = K.pwconv_wgrad(dz, pro, x, None, _dest(ctx, 1, (Chead, Cin))).view(Chead, Cin, 1, 1)
        return dx, dw_head, dgamma, dbeta, dw_fc, db_fc, None, None

# =========================================================================== hooked head (Grad-CAM)
class HeadConvFunction(torch.autograd.Function):
    """The head's 1x1 convolution alone (raw output, NHWC): used when forward hooks sit on the head-conv
    module, so that its OUTPUT exists as an autograd tensor (web_ui.py:96-114 attaches Grad-CAM there)."""

    @staticmethod
    def forward(ctx, x, w_head):
        w_nk, w_kn = K.prep_weights(w_head, x.dtype, True, True)
        y, _, _ = K.pwconv(x, None, w_nk, None, stats=False)
        ctx.save_for_backward(x, w_kn)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w_kn = ctx.saved_tensors
        g = _c(g)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx, _, _ = K.pwconv(g, None, w_kn, None, stats=False)
        if ctx.needs_input_grad[1]:
            dw = K.pwconv_wgrad(g, None, x, None).view(g.shape[3], x.shape[3], 1, 1)
        return dx, dw


class HeadTailEvalFunction(torch.autograd.Function):
    """Eval-mode BN + SiLU -> global average pool -> Linear on the raw head-conv output.  Differentiable
    with respect to that activation only (the Grad-CAM use): parameters get no gradient here."""

    @staticmethod
    def forward(ctx, y, w_fc, b_fc, bn: BNRef, gamma, beta):
        st = K.bn_eval_coeffs(bn.params(gamma, beta))
        pooled = K.pool_act(y, st, ACT_SILU)
        logits = K.linear_fwd(pooled, w_fc, b_fc)
        ctx.save_for_backward(y, st, pooled, w_fc)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        y, st, pooled, w_fc = ctx.saved_tensors
        dy = None
        if ctx.needs_input_grad[0]:
            dpooled, _, _ = K.linear_bwd(_c(dlogits.float()), pooled, w_fc, True, False, False, None, None)
            dz, _, _ = K.act_bn_bwd(None, y, None, dpooled, st, ACT_SILU)          # d loss / d (BN output)
            C = y.shape[3]
            coef = torch.empty((3, C), dtype=torch.float32, device=y.device)        # eval BN: dy = scale * dz
            coef[0].copy_(st[0])
            coef[1:].zero_()
            dy = K.affine2_apply(dz, y, coef)
        return dy, None, None, None, None, None



# =========================================================================== loss
class CrossEntropyFunction(torch.autograd.Function):
    """Mean label-smoothed cross entropy of f32 logits (trainers/efficientnet.py:412)."""

    @staticmethod
    def forward(ctx, logits, targets, label_smoothing: float):
        loss, dlogits = K.ce_loss(_c(logits), targets, label_smoothing, 1.0, ctx.needs_input_grad[0])
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (dlogits,) = ctx.saved_tensors
        return K.axpby(dlogits, None, 1.0, 0.0, a_dev=gloss.reshape(1).float()), None, None


__all__ = ["BNRef", "CrossEntropyFunction", "HeadConvFunction", "HeadCtx", "HeadFunction", "HeadTailEvalFunction", "MBConvCtx", "MBConvFunction", "StemCtx",
           "StemFunction"]
