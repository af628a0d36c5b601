"""Fused autograd functions of the FasterViT engine.

  ConvBlockFunction        conv3x3 (+bias) BN GELU conv3x3 (+bias) BN [* gamma] + x          faster_vit.py ConvBlock
  FVDownsampleFunction     LayerNorm2d -> conv3x3 s2                                          Downsample
  TokenInitFunction        dw3x3 (+bias) -> AvgPool2d(k, s)  -> carrier tokens                TokenInitializer
  HATFunction              hierarchical attention block: carrier-token attention + MLP, window attention + MLP
                           over [carrier ; window] tokens, split                              HAT
(the stem uses vit_functions.ConvStemFunction / DenseConvBNFunction with ReLU, the tail TailFunction with one head).

Dense 3x3 convolutions run as im2col + the MFMA GEMM kernels; Linear layers run on the same GEMM kernels with the
bias (and LayerScale, residual, DropPath scale) applied by the BatchNorm-apply pass under an identity statistic
(mean 0, variance 1, eps 0), so `Linear + bias` needs no kernel of its own and its bias gradient is the BN-beta
gradient.  Window bookkeeping (partition, carrier-token concatenation and split) is `dfd_copy_rows` with index
maps built once per batch size.  Reference call sites: trainers/fastervit.py:271 (train), :235 (evaluate),
orchestration/orchestrator.py:529,590; arithmetic per fastervit 1.0.0 faster_vit.py (restated in
oracle/fastervit_ref.py).
"""

from __future__ import annotations

import contextlib
import threading
from dataclasses import dataclass

import torch

from . import kernels as K
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU
from .functions import BNRef, _bn_state, _c
from .vit_functions import _gemm_weight, _partial_rows, _rows, _slot, pwbn_bwd, pwbn_fwd

_ident_cache: dict = {}


def ident(device: torch.device, C: int):
    """(ones [C], BNRef with mean 0 / variance 1 / eps 0): the identity statistic that turns the BN-apply kernels into
    `+ bias`."""
    key = (device.type, device.index, C)
    hit = _ident_cache.get(key)
    if hit is None:
        with torch.inference_mode(False):
            ones = torch.ones(C, dtype=torch.float32, device=device)
            zeros = torch.zeros(C, dtype=torch.float32, device=device)
        hit = (ones, BNRef(zeros, ones, None, 0.0, 0.0))
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _ident_cache[key] = hit             # tensors born inside a capture belong to that graph's pool: not cached
    return hit


def tok4(t: torch.Tensor) -> torch.Tensor:
    """[n, T, C] token tensor as the [n, T, 1, C] NHWC view the row kernels take."""
    return t if t.dim() == 4 else t.view(t.shape[0], t.shape[1], 1, t.shape[2])


# =========================================================================== Linear on the GEMM kernels
def lin_fwd(x, w_nk, bias, act=ACT_NONE, ls=None, residual=None, row_scale=None):
    ones, ref = ident(x.device, w_nk.shape[0])
    return pwbn_fwd(x, w_nk, None, ones, bias, ref, False, None, act, ls, residual, row_scale, raw_unused=True)


def lin_bwd(g, x, y, st, w_kn, w, bias, ls, act, need_dx, need_w, need_b, need_ls=False, dx_residual=None, row_scale=None):
    """-> (dx, dw, dbias, dls)"""
    ones, _ = ident(x.device, w.shape[0])
    dx, dw, _, _, dbeta, dls = pwbn_bwd(g, x, y, st, w_kn, tuple(w.shape), w, None, ones, bias, ls, act, False, need_dx, need_w,
                                        need_b and bias is not None, need_ls, dx_residual, row_scale, identity=True)
    return dx, dw, (dbeta if need_b else None), dls


# =========================================================================== coordinate MLPs, batched per level (f32)
@dataclass
class CoordJob:
    """One PosEmbMLPSwinv1D ("pos": table [T, C] added to the tokens) or PosEmbMLPSwinv2D ("cpb": relative-position attention
    bias [H, S, S] = 16 sigmoid(table[idx])) of a hierarchical-attention block."""

    kind: str                       # "pos" | "cpb"
    coords: torch.Tensor            # [T, 2] f32 constant
    idx: torch.Tensor | None = None     # cpb: int32 [n_local^2]
    n_local: int = 0
    n_global: int = 0


class CoordTablesFunction(torch.autograd.Function):
    """Every coordinate MLP of a level: Linear(2, 512) -> ReLU -> Linear(512, D, bias=False) on constant coordinates, and for
    the "cpb" jobs the gather + 16 sigmoid that turns the table into the attention bias.  Inputs: (w0, b0, w2) per job.
    Outputs: one tensor per job ("pos": table [T, D]; "cpb": bias [H, S, S]).  2 launches forward, 2 backward
    (csrc/dfd_coord.hip) — the per-layer form cost ~16 launches per block and sat on every block's critical path."""

    @staticmethod
    def forward(ctx, jobs, *params):
        from ._lib import CmlpJob, RelposJob

        dev = params[0].device
        n = len(jobs)
        tables, outs = [], []
        cj = (CmlpJob * n)()
        rp = []
        for i, job in enumerate(jobs):
            w0, b0, w2 = params[3 * i:3 * i + 3]
            T, D = job.coords.shape[0], w2.shape[0]
            tab = torch.empty((T, D), dtype=torch.float32, device=dev)
            tables.append(tab)
            cj[i] = CmlpJob(K._p(job.coords), K._p(w0), K._p(b0), K._p(w2), K._p(tab), None, None, None, None, T, D, w0.shape[0], 0)
            if job.kind == "cpb":
                S = job.n_local + job.n_global
                full = torch.empty((D, S, S), dtype=torch.float32, device=dev)
                rp.append(RelposJob(K._p(tab), K._p(job.idx), K._p(full), None, None, D, T, job.n_local, job.n_global))
                outs.append(full)
            else:
                outs.append(tab)
        K.check(K._L().dfd_coord_mlp_fwd_multi(cj, n, K._stream()), "dfd_coord_mlp_fwd_multi")
        if rp:
            arr = (RelposJob * len(rp))(*rp)
            K.check(K._L().dfd_relpos_bias_fwd_multi(arr, len(rp), K._stream()), "dfd_relpos_bias_fwd_multi")
        # never keep an OUTPUT on ctx: output -> grad_fn (this node) -> ctx -> output is a cycle Python's collector cannot see
        # through; it would pin this node and the AccumulateGrad nodes behind it across iterations (and a stale AccumulateGrad
        # node on the legacy stream breaks a later hipGraph capture).  Only the raw tables of the "cpb" jobs are needed.
        ctx.jobs, ctx.params = jobs, params
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.raw = [t if job.kind == "cpb" else None for t, job in zip(tables, jobs)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        from ._lib import CmlpJob, RelposJob

        jobs, params, shapes, raw = ctx.jobs, ctx.params, ctx.shapes, ctx.raw
        need = ctx.needs_input_grad[1:]
        dev = params[0].device
        grads: list = [None] * len(params)
        cj, rp, keep = [], [], []
        for i, job in enumerate(jobs):
            g = gouts[i]
            nw0, nb0, nw2 = need[3 * i:3 * i + 3]
            if g is None or not (nw0 or nb0 or nw2):
                continue
            w0, b0, w2 = params[3 * i:3 * i + 3]
            T, D = shapes[i]
            g = _c(g.float())
            if job.kind == "cpb":
                dtab = torch.empty((T, D), dtype=torch.float32, device=dev)
                rp.append(RelposJob(K._p(raw[i]), K._p(job.idx), None, K._p(g), K._p(dtab), D, T, job.n_local, job.n_global))
                keep.append(g)
                g = dtab
            dw0 = (_slot(w0, True, tuple(w0.shape)) if nw0 else None)
            db0 = (_slot(b0, True, tuple(b0.shape)) if nb0 else None)
            dw2 = (_slot(w2, True, tuple(w2.shape)) if nw2 else None)
            if nw0 and dw0 is None:
                dw0 = torch.empty_like(w0)
            if nb0 and db0 is None:
                db0 = torch.empty_like(b0)
            if nw2 and dw2 is None:
                dw2 = torch.empty_like(w2)
            grads[3 * i], grads[3 * i + 1], grads[3 * i + 2] = dw0, db0, dw2
            keep.append(g)
            cj.append(CmlpJob(K._p(job.coords), K._p(w0), K._p(b0), K._p(w2), None, K._p(g), K._p(dw0), K._p(db0), K._p(dw2),
                              T, D, w0.shape[0], 0))
        if rp:
            arr = (RelposJob * len(rp))(*rp)
            K.check(K._L().dfd_relpos_bias_bwd_multi(arr, len(rp), K._stream()), "dfd_relpos_bias_bwd_multi")
        if cj:
            arr = (CmlpJob * len(cj))(*cj)
            K.check(K._L().dfd_coord_mlp_bwd_multi(arr, len(cj), K._stream()), "dfd_coord_mlp_bwd_multi")
        return (None, *grads)


def coord_tables(jobs: list, params: list) -> list:
    """[table | bias per job] for `jobs` (CoordJob) with their (w0, b0, w2) parameter triples flattened in `params`."""
    return list(CoordTablesFunction.apply(jobs, *params))


# =========================================================================== attention / MLP sub-blocks on [n, T, C]
_derived_tls = threading.local()


@contextlib.contextmanager
def derived_weights(table: dict | None):
    """Inside the block `_prep` serves the (w_nk, w_kn) pairs of `table` ({weight.data_ptr(): pair}, refreshed for the whole
    network by ONE batched launch per forward pass — kernels.DerivedWeights) instead of one small launch per Linear."""
    prev = getattr(_derived_tls, "table", None)
    _derived_tls.table = table
    try:
        yield
    finally:
        _derived_tls.table = prev


def _prep(w: torch.Tensor, dt: torch.dtype):
    table = getattr(_derived_tls, "table", None)
    if table is not None:
        hit = table.get(w.data_ptr())
        if hit is not None and (isinstance(hit[0], K.MxWeight) or hit[0].dtype == dt):
            return hit
    return K.prep_weights(w, dt, True, True)


@dataclass
class AttnSpec:
    heads: int
    n_local: int            # tokens of the window grid (49, or 16 for the carrier grid)
    n_global: int           # carrier tokens in front (4 in level 2's window attention, else 0)
    coords2d: torch.Tensor  # [(2w-1)^2, 2] f32 log-spaced relative coordinates
    idx: torch.Tensor       # int32 [n_local^2] relative position index


def attn_sub_fwd(x, P, spec: AttnSpec, ls, row_scale):
    """x + [rs *] [ls *] proj(softmax(q k^T * scale + bias) v) with q, k, v = qkv(LN(x)).
    P: dict of tensors (ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, bias); bias = the [H, T, T] relative-position bias of
    CoordTablesFunction."""
    n, T, _, C = x.shape
    H = spec.heads
    hd = C // H
    dt = x.dtype
    xn, lnst = K.layernorm_fwd(x, P["ln_w"], P["ln_b"], 1e-5)
    wq_nk, wq_kn = _prep(P["qkv_w"], dt)
    qkv, yq, stq = lin_fwd(xn, wq_nk, P["qkv_b"])
    bias_full = P["bias"]
    if K.wattn_supported(dt, T, hd):
        # fused MFMA attention (csrc/dfd_attn.hip): S and P stay in registers; the backward recomputes P from L
        O, Pm = K.wattn_fwd(qkv, bias_full, H, hd ** -0.5)
    else:
        S = torch.empty((n, H, T, T), dtype=torch.float32, device=x.device)
        q, k, v = qkv.view(n * T, 3 * C)[:, 0:C], qkv.view(n * T, 3 * C)[:, C:2 * C], qkv.view(n * T, 3 * C)[:, 2 * C:]
        K.bgemm(q, (T * 3 * C, hd, 3 * C, 1), k, (T * 3 * C, hd, 1, 3 * C), S, (H * T * T, T * T, T, 1), n, H, T, T, hd,
                alpha=hd ** -0.5, bias=bias_full)
        Pm, _ = K.attn_softmax_fwd(S, None)
        O = torch.empty((n, T, 1, C), dtype=dt, device=x.device)
        K.bgemm(Pm, (H * T * T, T * T, T, 1), v, (T * 3 * C, hd, 3 * C, 1), O, (T * C, hd, C, 1), n, H, T, hd, T)
    wp_nk, wp_kn = _prep(P["proj_w"], dt)
    out, yp, stp = lin_fwd(O, wp_nk, P["proj_b"], ACT_NONE, ls, x, row_scale)
    return out, (x, xn, lnst, qkv, yq, stq, Pm, O, yp, stp, wq_kn, wp_kn, bias_full)


def attn_sub_bwd(g, saved, P, spec: AttnSpec, ls, row_scale, need: dict, need_dx: bool):
    """-> (dx, grads dict).  need: name -> bool for the entries of P and 'ls'."""
    x, xn, lnst, qkv, yq, stq, Pm, O, yp, stp, wq_kn, wp_kn, bias_full = saved
    n, T, _, C = x.shape
    H = spec.heads
    hd = C // H
    dev = x.device
    grads = {}
    upstream = need_dx or any(need[k] for k in ("ln_w", "ln_b", "qkv_w", "qkv_b", "bias"))
    dO, grads["proj_w"], grads["proj_b"], grads["ls"] = lin_bwd(g, O, yp, stp, wp_kn, P["proj_w"], P["proj_b"], ls, ACT_NONE, upstream,
                                                                 need["proj_w"], need["proj_b"], need["ls"], None, row_scale)
    if not upstream:
        return None, grads
    need_cpb = need["bias"]
    L = T * T
    if K.wattn_supported(x.dtype, T, hd):
        dqkv, dfull = K.wattn_bwd(qkv, dO, Pm, bias_full, H, hd ** -0.5, need_cpb)       # Pm holds the log-sum-exp rows here
    else:
        q, k, v = qkv.view(n * T, 3 * C)[:, 0:C], qkv.view(n * T, 3 * C)[:, C:2 * C], qkv.view(n * T, 3 * C)[:, 2 * C:]
        dT2 = torch.empty((n, H, T, T), dtype=torch.float32, device=dev)
        K.bgemm(dO, (T * C, hd, C, 1), v, (T * 3 * C, hd, 1, 3 * C), dT2, (H * L, L, T, 1), n, H, T, T, hd)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv.view(n * T, 3 * C)[:, 0:C], dqkv.view(n * T, 3 * C)[:, C:2 * C], dqkv.view(n * T, 3 * C)[:, 2 * C:]
        K.bgemm(Pm, (H * L, L, 1, T), dO, (T * C, hd, C, 1), dv, (T * 3 * C, hd, 3 * C, 1), n, H, T, hd, T)
        dS_buf = torch.empty((_partial_rows(n), H, T, T), dtype=torch.float32, device=dev)
        from ._lib import check

        check(K._L().dfd_attn_softmax_bwd(dT2.data_ptr(), Pm.data_ptr(), None, None, None, dS_buf.data_ptr(), n, H, T, T, K._stream()),
              "dfd_attn_softmax_bwd")
        dS = dS_buf[:n]
        scale = hd ** -0.5
        K.bgemm(dS, (H * L, L, T, 1), k, (T * 3 * C, hd, 3 * C, 1), dq, (T * 3 * C, hd, 3 * C, 1), n, H, T, hd, T, alpha=scale)
        K.bgemm(dS, (H * L, L, 1, T), q, (T * 3 * C, hd, 3 * C, 1), dk, (T * 3 * C, hd, 3 * C, 1), n, H, T, hd, T, alpha=scale)
        dfull = None
        if need_cpb:
            dfull = torch.empty(H * L, dtype=torch.float32, device=dev)
            K.sum_rows(dS_buf.view(-1), n, H * L, dfull)
    if need_cpb:
        grads["bias"] = dfull.view(H, T, T)
    need_xn = need_dx or need["ln_w"] or need["ln_b"]
    dxn, grads["qkv_w"], grads["qkv_b"], _ = lin_bwd(dqkv, xn, yq, stq, wq_kn, P["qkv_w"], P["qkv_b"], None, ACT_NONE, need_xn,
                                                     need["qkv_w"], need["qkv_b"])
    dx = None
    if need_xn:
        dx, grads["ln_w"], grads["ln_b"] = _ln_bwd(dxn, x, P["ln_w"], P["ln_b"], lnst, need["ln_w"], need["ln_b"],
                                                   g if need_dx else None)
        if not need_dx:
            dx = None
    return dx, grads


def _ln_bwd(dxn, x, ln_w, ln_b, lnst, need_w: bool, need_b: bool, residual=None):
    """LayerNorm backward with the skip connection's gradient added in the same kernel; (dgamma, dbeta) are summed straight
    into the parameters' gradient-arena slots when both are wanted and the slots are adjacent (weight, bias: they are)."""
    C = ln_w.numel()
    sw = _slot(ln_w, need_w, (C,)) if need_w else None
    sb = _slot(ln_b, need_b, (C,)) if need_b else None
    if sw is not None and sb is not None and sb.data_ptr() == sw.data_ptr() + 4 * C:
        both = torch.as_strided(sw, (2, C), (C, 1))
        dx, dg, db = K.layernorm_bwd(dxn, x, ln_w, lnst, residual, both)
        return dx, dg, db
    dx, dg, db = K.layernorm_bwd(dxn, x, ln_w, lnst, residual)

    def place(val, slot, need):
        if not need:
            return None
        if slot is None:
            return val
        return K.axpby(val.reshape(-1), None, 1.0, 0.0, out=slot.view(-1)).view(slot.shape)

    return dx, place(dg, sw, need_w), place(db, sb, need_b)


def _to_slot(val: torch.Tensor, param: torch.Tensor, need: bool):
    """Move a freshly computed f32 gradient into the parameter's arena slot when it has one."""
    if not need:
        return None
    slot = _slot(param, True, tuple(param.shape))
    if slot is None:
        return val.view(param.shape)
    return K.axpby(val.reshape(-1), None, 1.0, 0.0, out=slot.view(-1)).view(param.shape)


def mlp_sub_fwd(x, P, ls, row_scale):
    """x + [rs *] [ls *] fc2(GELU(fc1(LN(x)))).  P: ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b."""
    dt = x.dtype
    xn, lnst = K.layernorm_fwd(x, P["ln_w"], P["ln_b"], 1e-5)
    w1_nk, w1_kn = _prep(P["fc1_w"], dt)
    hid = P["fc1_w"].shape[0]
    ones, ref = ident(x.device, hid)
    st1 = _bn_state(None, 0, _rows(xn), ref, ones, P["fc1_b"], False)
    w2_nk, w2_kn = _prep(P["fc2_w"], dt)
    C = P["fc2_w"].shape[0]
    ones2, ref2 = ident(x.device, C)
    st2 = _bn_state(None, 0, _rows(xn), ref2, ones2, P["fc2_b"], False, None, None, ls)
    # fc1 + bias + GELU and fc2 + bias (+ LayerScale, DropPath row scale, skip connection) are one kernel each where the shape
    # is dfd_gemm's (the activated hidden tensor `a` and the pre-activation `h` both leave fc1's epilogue: `h` for GELU');
    # otherwise the activated tensor is materialised by one pass — as a GEMM prologue the GELU was evaluated once per
    # 128-column output tile by fc2's forward and again by its weight gradient
    f1 = K.gemm_bias_act(xn, w1_nk, st1, ACT_GELU, None, None, want_raw=True)
    if f1 is not None:
        a, h = f1
    else:
        h, _, _ = K.pwconv(xn, None, w1_nk, None, stats=False)
        a = K.bn_act_apply(h, st1, ACT_GELU)
    f2 = K.gemm_bias_act(a, w2_nk, st2, ACT_NONE, x, row_scale, want_raw=ls is not None)
    if f2 is not None:
        out, y2 = f2                                        # y2 None without LayerScale: the backward never reads it (and `out`
                                                            # must not stand in: a Function must not save its own output)
    else:
        y2, _, _ = K.pwconv(a, None, w2_nk, None, stats=False)
        out = K.bn_act_apply(y2, st2, ACT_NONE, x, row_scale)
    return out, (x, xn, lnst, h, a, st1, y2, st2, w1_kn, w2_kn)


def mlp_sub_bwd(g, saved, P, ls, row_scale, need: dict, need_dx: bool):
    x, xn, lnst, h, a, st1, y2, st2, w1_kn, w2_kn = saved
    grads = {}
    C, hid = P["fc2_w"].shape[0], P["fc1_w"].shape[0]
    dev = x.device
    gb = K.scale_rows(g, row_scale) if row_scale is not None else g
    # fc2 has no BatchNorm: the backward map of its identity statistic is dz = ls * g.  Without layer scale that is g itself and
    # the bias gradient is the column sums of g (one reduction launch, its final summation in the block's batch); with it the
    # BatchNorm-shaped pair runs and the scaled gradient is materialised once (as a GEMM prologue it was evaluated once per
    # 128-column tile of the data gradient — 8 times per element at hidden 1024 — and read y2 only to multiply it by zero)
    if ls is None:
        grads["fc2_b"] = K.bias_grad(gb, None, _slot(P["fc2_b"], True, (C,))) if need["fc2_b"] else None
        grads["ls"] = None
        dz2 = gb
    else:
        parts, n = K.bn_bwd_reduce(gb, y2, st2, None)
        ones2, _ = ident(dev, C)
        outs = (None, _slot(P["fc2_b"], need["fc2_b"], (C,)), _slot(ls, need["ls"], (C,)), None)
        coef2, _, db2, dls, _ = K.bn_bwd_finalize_ex(parts, n, _rows(y2), ones2, P["fc2_b"], ls, st2, False, need["fc2_b"] or True,
                                                     need["ls"], False, outs)
        grads["fc2_b"], grads["ls"] = (db2 if need["fc2_b"] else None), dls
        dz2 = K.affine2_apply(gb, y2, coef2)
    upstream = need_dx or any(need[k] for k in ("ln_w", "ln_b", "fc1_w", "fc1_b"))
    if need["fc2_w"]:
        grads["fc2_w"] = K.pwconv_wgrad(dz2, None, a, None, _slot(P["fc2_w"], True, (C, hid))).view(P["fc2_w"].shape)
    if not upstream:
        return None, grads
    D, _, _ = K.pwconv(dz2, None, w2_kn, None, stats=False)
    dz1, parts, n = K.act_bn_bwd(D, h, None, None, st1, ACT_GELU)
    if need["fc1_b"]:
        ones1, _ = ident(dev, hid)
        _, _, db1, _, _ = K.bn_bwd_finalize_ex(parts, n, _rows(h), ones1, P["fc1_b"], None, st1, False, True, False, False,
                                               (None, _slot(P["fc1_b"], True, (hid,)), None, None))
        grads["fc1_b"] = db1
    if need["fc1_w"]:
        grads["fc1_w"] = K.pwconv_wgrad(dz1, None, xn, None, _slot(P["fc1_w"], True, (hid, C))).view(P["fc1_w"].shape)
    dx = None
    if need_dx or need["ln_w"] or need["ln_b"]:
        dxn, _, _ = K.pwconv(dz1, None, w1_kn, None, stats=False)
        dx, grads["ln_w"], grads["ln_b"] = _ln_bwd(dxn, x, P["ln_w"], P["ln_b"], lnst, need["ln_w"], need["ln_b"],
                                                   g if need_dx else None)
        if not need_dx:
            dx = None
    return dx, grads


# =========================================================================== HAT block
_ATTN_KEYS = ("ln_w", "ln_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "bias")
_MLP_KEYS = ("ln_w", "ln_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")
_POS_KEYS = ("table",)


@dataclass
class HATCtx:
    heads: int
    carrier: bool                       # level 2: carrier-token stream + concatenation
    win_spec: AttnSpec
    ct_spec: AttnSpec | None
    coords_win: torch.Tensor            # [49, 2] PosEmbMLPSwinv1D coordinates of the window grid
    coords_ct: torch.Tensor | None      # [16, 2]
    cat_src_ct: torch.Tensor | None     # int32 [nW*4]: carrier row of every (window, slot)
    cat_dst_ct: torch.Tensor | None     # int32 [nW*4]: its row in the concatenated tensor
    cat_dst_x: torch.Tensor | None      # int32 [nW*49]: row of every window token in the concatenated tensor
    training: bool


def _layout(carrier: bool, has_ls: bool) -> list[str]:
    """Order of the tensor inputs after (x, ct)."""
    names = [f"pos.{k}" for k in _POS_KEYS] + [f"attn.{k}" for k in _ATTN_KEYS] + [f"mlp.{k}" for k in _MLP_KEYS]
    if carrier:
        names += [f"hpos.{k}" for k in _POS_KEYS] + [f"hattn.{k}" for k in _ATTN_KEYS] + [f"hmlp.{k}" for k in _MLP_KEYS]
    names += ["gamma3", "gamma4"] + (["gamma1", "gamma2"] if carrier else [])
    names += ["rs_win", "rs_ct"]
    return names


class HATFunction(torch.autograd.Function):
    """One hierarchical-attention block.  x: [nW, 49, 1, C] window tokens; ct: [B, 16, 1, C] carrier tokens in row-major
    image order (None without carrier tokens).  Returns (x_out, ct_out)."""

    @staticmethod
    def forward(ctx, x, ct, cfg: HATCtx, *t):
        names = _layout(cfg.carrier, True)
        T = dict(zip(names, t))
        nW, Tw, _, C = x.shape

        def sub(prefix, keys):
            return {k: T[f"{prefix}.{k}"] for k in keys}

        x1 = K.add_rowtable(x, T["pos.table"])
        saved_ct = None
        if cfg.carrier:
            B, Tc = ct.shape[0], ct.shape[1]
            c1 = K.add_rowtable(ct, T["hpos.table"])
            c2, hattn_saved = attn_sub_fwd(c1, sub("hattn", _ATTN_KEYS), cfg.ct_spec, T["gamma1"], T["rs_ct"])
            c3, hmlp_saved = mlp_sub_fwd(c2, sub("hmlp", _MLP_KEYS), T["gamma2"], T["rs_ct"])
            per = Tc * B // nW                                      # carrier tokens per window (4)
            xc = torch.empty((nW, Tw + per, 1, C), dtype=x.dtype, device=x.device)
            K.copy_rows(c3.view(-1, C), cfg.cat_src_ct, xc.view(-1, C), cfg.cat_dst_ct, nW * per)
            K.copy_rows(x1.view(-1, C), None, xc.view(-1, C), cfg.cat_dst_x, nW * Tw)
            saved_ct = (hattn_saved, hmlp_saved, per)
        else:
            xc = x1
        xa, attn_saved = attn_sub_fwd(xc, sub("attn", _ATTN_KEYS), cfg.win_spec, T["gamma3"], T["rs_win"])
        xm, mlp_saved = mlp_sub_fwd(xa, sub("mlp", _MLP_KEYS), T["gamma4"], T["rs_win"])
        if cfg.carrier:
            per = saved_ct[2]
            x_out = torch.empty_like(x)
            ct_out = torch.empty_like(ct)
            K.copy_rows(xm.view(-1, C), cfg.cat_dst_x, x_out.view(-1, C), None, nW * Tw)
            K.copy_rows(xm.view(-1, C), cfg.cat_dst_ct, ct_out.view(-1, C), cfg.cat_src_ct, nW * per)
        else:
            x_out, ct_out = xm, None
        ctx.cfg = cfg
        ctx.names = names
        ctx.T = T
        ctx.saved = (saved_ct, attn_saved, mlp_saved)
        ctx.shapes = (tuple(x.shape), tuple(ct.shape) if ct is not None else None, x.dtype)
        if ct_out is None:
            ctx.mark_non_differentiable()
            return x_out, None
        return x_out, ct_out

    @staticmethod
    @K.batched_sums
    def backward(ctx, gx, gct):
        cfg: HATCtx = ctx.cfg
        names, T = ctx.names, ctx.T
        saved_ct, attn_saved, mlp_saved = ctx.saved
        x_shape, ct_shape, dt = ctx.shapes
        need = dict(zip(["x", "ct", "cfg"] + names, ctx.needs_input_grad))
        nW, Tw, _, C = x_shape
        dev = gx.device
        grads: dict = {}

        def sub(prefix, keys):
            return {k: T[f"{prefix}.{k}"] for k in keys}

        def sub_need(prefix, keys, gamma):
            d = {k: need[f"{prefix}.{k}"] for k in keys}
            d["ls"] = need[gamma]
            return d

        carrier_up = cfg.carrier and (need["ct"] or any(need[n] for n in names if n.startswith(("hpos", "hattn", "hmlp", "gamma1", "gamma2"))))
        need_x1 = need["x"] or any(need[f"pos.{k}"] for k in _POS_KEYS)
        gx = _c(gx)
        if cfg.carrier:
            per = saved_ct[2]
            gm = torch.empty((nW, Tw + per, 1, C), dtype=dt, device=dev)
            K.copy_rows(gx.view(-1, C), None, gm.view(-1, C), cfg.cat_dst_x, nW * Tw)
            if gct is not None:
                K.copy_rows(_c(gct).view(-1, C), cfg.cat_src_ct, gm.view(-1, C), cfg.cat_dst_ct, nW * per)
            else:
                zero = torch.zeros((nW * per, C), dtype=dt, device=dev)
                K.copy_rows(zero, None, gm.view(-1, C), cfg.cat_dst_ct, nW * per)
        else:
            gm = gx
        need_xa = True
        dxa, g_mlp = mlp_sub_bwd(gm, mlp_saved, sub("mlp", _MLP_KEYS), T["gamma4"], T["rs_win"], sub_need("mlp", _MLP_KEYS, "gamma4"), need_xa)
        for k, v in g_mlp.items():
            grads["gamma4" if k == "ls" else f"mlp.{k}"] = v
        need_xc = need_x1 or carrier_up
        dxc, g_attn = attn_sub_bwd(dxa, attn_saved, sub("attn", _ATTN_KEYS), cfg.win_spec, T["gamma3"], T["rs_win"],
                                   sub_need("attn", _ATTN_KEYS, "gamma3"), need_xc)
        for k, v in g_attn.items():
            grads["gamma3" if k == "ls" else f"attn.{k}"] = v
        dx = dct = None
        if cfg.carrier:
            hattn_saved, hmlp_saved, per = saved_ct
            dx1 = None
            if need_x1:
                dx1 = torch.empty(x_shape, dtype=dt, device=dev)
                K.copy_rows(dxc.view(-1, C), cfg.cat_dst_x, dx1.view(-1, C), None, nW * Tw)
            if carrier_up:
                dc3 = torch.empty(ct_shape, dtype=dt, device=dev)
                K.copy_rows(dxc.view(-1, C), cfg.cat_dst_ct, dc3.view(-1, C), cfg.cat_src_ct, nW * per)
                dc2, g_hm = mlp_sub_bwd(dc3, hmlp_saved, sub("hmlp", _MLP_KEYS), T["gamma2"], T["rs_ct"], sub_need("hmlp", _MLP_KEYS, "gamma2"), True)
                for k, v in g_hm.items():
                    grads["gamma2" if k == "ls" else f"hmlp.{k}"] = v
                need_c1 = need["ct"] or any(need[f"hpos.{k}"] for k in _POS_KEYS)
                dc1, g_ha = attn_sub_bwd(dc2, hattn_saved, sub("hattn", _ATTN_KEYS), cfg.ct_spec, T["gamma1"], T["rs_ct"],
                                         sub_need("hattn", _ATTN_KEYS, "gamma1"), need_c1)
                for k, v in g_ha.items():
                    grads["gamma1" if k == "ls" else f"hattn.{k}"] = v
                if need_c1:
                    if need["hpos.table"]:
                        grads["hpos.table"] = K.rowtable_grad(dc1, ct_shape[1])
                    dct = dc1 if need["ct"] else None
        else:
            dx1 = dxc
        if need_x1 and dx1 is not None:
            if need["pos.table"]:
                grads["pos.table"] = K.rowtable_grad(dx1, Tw)
            dx = dx1 if need["x"] else None
        flat = [grads.get(nm) if need[nm] else None for nm in names]
        return (dx, dct, None, *flat)


# =========================================================================== ConvBlock (levels 0, 1)
@dataclass
class ConvBlockCtx:
    bn1: BNRef
    bn2: BNRef
    training: bool
    counters: list | None = None


class ConvBlockFunction(torch.autograd.Function):
    """x + [rs *] [gamma *] BN(conv3x3(GELU(BN(conv3x3(x)))))  (both convolutions with bias, folded into the BNs)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, gamma, row_scale, cfg: ConvBlockCtx):
        tr = cfg.training
        N, H, W, C = x.shape
        need_bwd = any(ctx.needs_input_grad)
        w1_nk, _ = _gemm_weight(w1, x.dtype, False)
        w1_kn = _dgrad_weight(w1, x.dtype) if need_bwd else None          # [C][(flipped tap, co)]: stride-1 data gradient
        y1, parts, n = K.conv_fwd(x, None, ACT_NONE, w1_nk, 3, 1, 1, H, W, stats=tr)
        st1 = _bn_state(parts, n, N * H * W, cfg.bn1, g1, be1, tr, cfg.counters, conv_bias=b1)
        w2_nk, _ = _gemm_weight(w2, x.dtype, False)
        w2_kn = _dgrad_weight(w2, x.dtype) if need_bwd else None
        # GELU(BN(y1)) is materialised once (one streaming pass, ~45 us at level 0): as the prologue of the second convolution and
        # of its weight gradient it was evaluated on every staged tile with its halo, per output-channel group, on the
        # workgroup's critical path — +50 us and +54 us per block against the plain kernels (same bits either way)
        a1 = K.bn_act_apply(y1, st1, ACT_GELU)
        y2, parts, n = K.conv_fwd(a1, None, ACT_NONE, w2_nk, 3, 1, 1, H, W, stats=tr)
        st2 = _bn_state(parts, n, N * H * W, cfg.bn2, g2, be2, tr, cfg.counters, conv_bias=b2, ls=gamma)
        out = K.bn_act_apply(y2, st2, ACT_NONE, x, row_scale)
        ctx.cfg = cfg
        ctx.save_for_backward(x, y1, a1, y2, st1, st2, w1_kn, w2_kn, w1, b1, g1, be1, w2, b2, g2, be2, gamma, row_scale)
        return out

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        cfg: ConvBlockCtx = ctx.cfg
        x, y1, a1, y2, st1, st2, w1_kn, w2_kn, w1, b1, g1, be1, w2, b2, g2, be2, gamma, row_scale = ctx.saved_tensors
        need = ctx.needs_input_grad
        tr = cfg.training
        N, H, W, C = x.shape
        rows = N * H * W
        g = _c(g)
        gb = K.scale_rows(g, row_scale) if row_scale is not None else g
        parts, n = K.bn_bwd_reduce(gb, y2, st2, None)
        nb2 = need[7] or need[8]
        outs = (_slot(g2, nb2, (C,)), _slot(be2, nb2, (C,)), _slot(gamma, need[9], (C,)), _slot(b2, need[6], (C,)))
        coef2, dg2, dbe2, dgam, db2 = K.bn_bwd_finalize_ex(parts, n, rows, g2, be2, gamma, st2, tr, nb2, need[9] and gamma is not None,
                                                           need[6], outs)
        dz2 = K.affine2_apply(gb, y2, coef2)              # the BN-backward-mapped gradient, once for both consumers
        dw2 = None
        if need[5]:
            dwg = K.conv_wgrad(dz2, None, a1, None, ACT_NONE, 3, 1, 1)
            dw2 = K.conv_wgrad_from_gemm(dwg, tuple(w2.shape), _slot(w2, True, tuple(w2.shape)))
        # data gradient of a stride-1 convolution = the forward convolution of the BN-backward-mapped gradient with the
        # flipped, transposed weight: no [M][9C] column matrix, no col2im
        da1, _, _ = K.conv_fwd(dz2, None, ACT_NONE, w2_kn, 3, 1, 1, H, W, stats=False)
        dz1, parts, n = K.act_bn_bwd(da1, y1, None, None, st1, ACT_GELU)
        nb1 = need[3] or need[4]
        outs = (_slot(g1, nb1, (C,)), _slot(be1, nb1, (C,)), None, _slot(b1, need[2], (C,)))
        coef1, dg1, dbe1, _, db1 = K.bn_bwd_finalize_ex(parts, n, rows, g1, be1, None, st1, tr, nb1, False, need[2], outs)
        dw1 = dx = None
        dzm1 = K.affine2_apply(dz1, y1, coef1) if (need[0] or need[1]) else None
        if need[1]:
            dwg = K.conv_wgrad(dzm1, None, x, None, ACT_NONE, 3, 1, 1)
            dw1 = K.conv_wgrad_from_gemm(dwg, tuple(w1.shape), _slot(w1, True, tuple(w1.shape)))
        if need[0]:
            dx0, _, _ = K.conv_fwd(dzm1, None, ACT_NONE, w1_kn, 3, 1, 1, H, W, stats=False)
            dx = K.add(dx0, g)
        return (dx, dw1, db1 if need[2] else None, dg1, dbe1, dw2, db2 if need[6] else None, dg2, dbe2, dgam, None, None)


def _dgrad_weight(w: torch.Tensor, dt: torch.dtype):
    w_nk, _ = K.prep_weights(K.conv_weight_to_dgrad_gemm(w), dt, True, False)
    return w_nk


# =========================================================================== Downsample (LayerNorm2d + conv3x3 s2)
class FVDownsampleFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w):
        N, H, W, C = x.shape
        need_bwd = any(ctx.needs_input_grad)
        xn, lnst = K.layernorm_fwd(x, ln_w, ln_b, 1e-6)
        w_nk, w_kn = _gemm_weight(w, x.dtype, need_bwd)
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y, _, _ = K.conv_fwd(xn, None, ACT_NONE, w_nk, 3, 2, 1, Ho, Wo, stats=False)
        ctx.save_for_backward(x, xn, lnst, w_kn, ln_w, ln_b, w)
        return y

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        x, xn, lnst, w_kn, ln_w, ln_b, w = ctx.saved_tensors
        need = ctx.needs_input_grad
        N, H, W, C = x.shape
        g = _c(g)
        dw = dx = dlw = dlb = None
        if need[3]:
            dwg = K.conv_wgrad(g, None, xn, None, ACT_NONE, 3, 2, 1)
            dw = K.conv_wgrad_from_gemm(dwg, tuple(w.shape), _slot(w, True, tuple(w.shape)))
        if need[0] or need[1] or need[2]:
            dcol, _, _ = K.pwconv(g, None, w_kn, None, stats=False)
            dxn = K.col2im(dcol, (N, H, W, C), 3, 2, 1)
            dx, dgm, dbt = K.layernorm_bwd(dxn, x, ln_w, lnst)
            dlw, dlb = _to_slot(dgm, ln_w, need[1]), _to_slot(dbt, ln_b, need[2])
            if not need[0]:
                dx = None
        return dx, dlw, dlb, dw


# =========================================================================== carrier-token initialiser
class TokenInitFunction(torch.autograd.Function):
    """dw3x3 (+bias) -> AvgPool2d(kernel, stride): [B, H, W, C] -> [B, h*w, 1, C] carrier tokens in row-major order."""

    @staticmethod
    def forward(ctx, x, w, b, kernel: int, stride: int):
        N, H, W, C = x.shape
        y, _, _ = K.dwconv_fwd(x, None, ACT_NONE, w, 3, 1, 1, 1, H, W, stats=False)
        p = K.avgpool_fwd(y, kernel, stride)
        out = K.add_rowtable(p, b.view(1, C))
        ctx.save_for_backward(x, w, b)
        ctx.geom = (kernel, stride, tuple(y.shape))
        return out.view(N, p.shape[1] * p.shape[2], 1, C)

    @staticmethod
    @K.batched_sums
    def backward(ctx, g):
        x, w, b = ctx.saved_tensors
        kernel, stride, yshape = ctx.geom
        need = ctx.needs_input_grad
        N, H, W, C = x.shape
        ho = (H - kernel) // stride + 1
        g4 = _c(g).view(N, ho, -1, C)
        db = _to_slot(K.rowtable_grad(g4, 1).view(-1), b, need[2])
        dx = dw = None
        if need[0] or need[1]:
            dy = K.avgpool_bwd(g4, yshape, kernel, stride)
            if need[0]:
                dx, _, _ = K.dwconv_bwd_data(dy, None, None, w, None, None, ACT_NONE, tuple(x.shape), 3, 1, 1, 1)
            if need[1]:
                dw = K.dwconv_bwd_weight(dy, None, None, x, None, ACT_NONE, 3, 1, 1, 1, _slot(w, True, (C, 1, 3, 3)))
        return dx, dw, db, None, None


__all__ = ["AttnSpec", "ConvBlockCtx", "ConvBlockFunction", "FVDownsampleFunction", "HATCtx", "HATFunction", "TokenInitFunction"]
