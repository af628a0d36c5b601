"""Kernel-level parity: every C-ABI entry point against the CPU oracle (oracle/ops_ref.py).

Tolerances: f32 kernels rel 2e-4 of the tensor's max magnitude (summation order differs);
bf16 kernels 1.6e-2 (two bf16 ulps: one rounding in the kernel, one in the oracle's
emulation), stated per assert.
"""

from __future__ import annotations

import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _k():
    from deepfakedetection_amd import kernels

    return kernels


def tol(rd):
    return 2e-4 if rd == torch.float32 else 1.6e-2


def close(got: torch.Tensor, want: torch.Tensor, rel: float, what: str = "") -> None:
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = max(float(want.abs().max()), 1e-6)
    err = float((got - want).abs().max()) / scale
    assert err <= rel, f"{what}: max err {err:.3e} of max |ref| {scale:.3e} > {rel:.1e}"


def gen(shape, seed, rd, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(rd)


def dev(t):
    return t.cuda() if t is not None else None


def sum_parts(parts, n, C):
    return parts[: n * 2 * C].view(n, 2, C).double().sum(0).float().cpu()


def rand_state(C, seed):
    g = torch.Generator().manual_seed(seed)
    scale = 0.5 + torch.rand(C, generator=g)
    shift = torch.randn(C, generator=g) * 0.3
    mean = torch.randn(C, generator=g) * 0.2
    rstd = 0.5 + torch.rand(C, generator=g)
    return torch.stack([scale, shift, mean, rstd])


DW_CASES = [
    # N, H, W, C, k, s, pt, pl
    (2, 14, 14, 40, 3, 1, 1, 1),
    (2, 15, 13, 24, 3, 2, 0, 0),     # TF-SAME asymmetric, odd sizes
    (1, 28, 28, 144, 5, 1, 2, 2),
    (2, 16, 16, 48, 5, 2, 1, 1),     # asymmetric (1,2)
    (2, 7, 7, 1152, 5, 1, 2, 2),
    (1, 56, 56, 96, 3, 2, 1, 1),     # timm symmetric stride 2
    (3, 9, 9, 8, 3, 1, 1, 1),        # smallest channel count
    (5, 56, 56, 144, 3, 1, 1, 1),    # several tiles per image, channel chunks sharing cache lines (B0 block 2)
    (4, 7, 7, 896, 3, 1, 1, 1),      # EfficientFormerV2 stage 3 ConvMlp
    (2, 29, 31, 192, 3, 1, 1, 1),    # ragged tile edges
]


def out_size(H, k, s, pt, flavour_same):
    return -(-H // s) if flavour_same else (H + 2 * pt - k) // s + 1


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_fwd(case, rd):
    K = _k()
    N, H, W, C, k, s, pt, pl = case
    Ho, Wo = -(-H // s), -(-W // s)
    x = gen((N, H, W, C), 1, rd)
    w = gen((C, 1, k, k), 2, torch.float32, 0.3)
    st = rand_state(C, 3)
    want = R.dwconv_fwd(x.float(), st, R.ACT_SILU, w, k, s, pt, pl, Ho, Wo, rd)
    y, parts, n = K.dwconv_fwd(dev(x), dev(st), R.ACT_SILU, dev(w), k, s, pt, pl, Ho, Wo, stats=True)
    close(y, want, tol(rd), "dwconv_fwd y")
    close(sum_parts(parts, n, C), R.stats_sums(y.float().cpu()), 1e-3, "dwconv_fwd stats")
    # no prologue, no stats
    want2 = R.dwconv_fwd(x.float(), None, 0, w, k, s, pt, pl, Ho, Wo, rd)
    y2, _, _ = K.dwconv_fwd(dev(x), None, 0, dev(w), k, s, pt, pl, Ho, Wo, stats=False)
    close(y2, want2, tol(rd), "dwconv_fwd raw")


# the matrix-core form of the forward (csrc/dfd_dwmm.hip) serves bf16 layers with C % 16 == 0; by default only where it measured
# faster, so these cases switch it on for every shape (dfd_tune key 0, bit 3): chunks of 1 / 2 / 3 / 4 groups, both strides and kernel
# sizes, whole-picture tiles with several images per item and a batch tail, ragged tile edges, asymmetric padding
DW_MM_CASES = [c for c in DW_CASES if c[3] % 16 == 0] + [
    (7, 7, 7, 1152, 5, 1, 2, 2),     # 3 pictures per item, 7 = 2 * 3 + 1: a partial last item
    (3, 14, 14, 16, 3, 1, 1, 1),     # one group
    (2, 30, 30, 32, 5, 1, 2, 2),     # two groups
    (2, 20, 18, 112, 3, 2, 0, 0),    # 4 + 3 groups, TF-SAME asymmetric stride 2
    (2, 33, 35, 80, 5, 2, 1, 1),     # 4 + 1 groups, ragged
    (1, 112, 112, 32, 3, 1, 1, 1),   # B0 block 0: many tiles per picture
]


@pytest.mark.parametrize("case", DW_MM_CASES)
def test_dwconv_fwd_matrix_core_form(case):
    import ctypes

    from deepfakedetection_amd._lib import DwShape, load

    K = _k()
    rd = torch.bfloat16
    N, H, W, C, k, s, pt, pl = case
    Ho, Wo = -(-H // s), -(-W // s)
    x = gen((N, H, W, C), 1, rd)
    w = gen((C, 1, k, k), 2, torch.float32, 0.3)
    st = rand_state(C, 3)
    lib = load()
    plan = (ctypes.c_int * 12)()
    assert lib.dfd_dw_mm_plan(ctypes.byref(DwShape(N, H, W, C, Ho, Wo, k, s, pt, pl)), 1, plan) == 0, "the matrix-core planner declined the shape"
    want = R.dwconv_fwd(x.float(), st, R.ACT_SILU, w, k, s, pt, pl, Ho, Wo, rd)
    want2 = R.dwconv_fwd(x.float(), None, 0, w, k, s, pt, pl, Ho, Wo, rd)
    assert lib.dfd_tune(0, 0) == 0
    y_v, parts_v, n_v = K.dwconv_fwd(dev(x), dev(st), R.ACT_SILU, dev(w), k, s, pt, pl, Ho, Wo, stats=True)
    sums_v = sum_parts(parts_v, n_v, C)
    assert lib.dfd_tune(0, 9) == 0
    try:
        y, parts, n = K.dwconv_fwd(dev(x), dev(st), R.ACT_SILU, dev(w), k, s, pt, pl, Ho, Wo, stats=True)
        sums = sum_parts(parts, n, C)
        y2, _, _ = K.dwconv_fwd(dev(x), None, 0, dev(w), k, s, pt, pl, Ho, Wo, stats=False)
    finally:
        lib.dfd_tune(0, 1)
    close(y, want, tol(rd), "matrix-core dwconv_fwd y")
    close(sums, R.stats_sums(y.float().cpu()), 1e-3, "matrix-core dwconv_fwd stats")
    close(y2, want2, tol(rd), "matrix-core dwconv_fwd raw")
    # the two forms round the same f32 sums of the same bf16 products: they may differ in the order of the additions only
    close(y, y_v.float().cpu(), 8e-3, "matrix-core against vector-unit form")
    close(sums, sums_v, 2e-3, "matrix-core against vector-unit statistics")
    assert lib.dfd_tune(99, 0) != 0


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_bwd(case, rd):
    K = _k()
    N, H, W, C, k, s, pt, pl = case
    Ho, Wo = -(-H // s), -(-W // s)
    xin = gen((N, H, W, C), 4, rd)
    dz = gen((N, Ho, Wo, C), 5, rd)
    yraw = gen((N, Ho, Wo, C), 6, rd)
    w = gen((C, 1, k, k), 7, torch.float32, 0.3)
    st = rand_state(C, 8)
    g = torch.Generator().manual_seed(9)
    coef = torch.stack([0.5 + torch.rand(C, generator=g), torch.randn(C, generator=g) * 0.1,
                        torch.randn(C, generator=g) * 0.05])
    dy = R.rnd(coef[0] * dz.float() + coef[1] * yraw.float() + coef[2], rd)
    z = st[0] * xin.float() + st[1]
    xact = R.rnd(R.act_fwd(z, R.ACT_SILU), rd)
    da, dw = R.dwconv_bwd(dy, xact, w, k, s, pt, pl, rd)
    want_dzin = R.rnd(da * R.act_grad(z, R.ACT_SILU), rd)
    dzin, parts, n = K.dwconv_bwd_data(dev(dz), dev(yraw), dev(coef), dev(w), dev(xin), dev(st), R.ACT_SILU,
                                       (N, H, W, C), k, s, pt, pl)
    close(dzin, want_dzin, tol(rd), "dwconv_bwd_data")
    got = dzin.float().cpu()
    xhat = (xin.float() - st[2]) * st[3]
    want_sums = torch.stack([got.reshape(-1, C).double().sum(0), (got * xhat).reshape(-1, C).double().sum(0)]).float()
    close(sum_parts(parts, n, C), want_sums, 2e-3, "dwconv_bwd_data stats")
    # plain variant: no coef, no epilogue
    da2, _ = R.dwconv_bwd(dz.float(), xact, w, k, s, pt, pl, rd)
    dzin2, _, _ = K.dwconv_bwd_data(dev(dz), None, None, dev(w), None, None, 0, (N, H, W, C), k, s, pt, pl)
    close(dzin2, R.rnd(da2, rd), tol(rd), "dwconv_bwd_data plain")
    got_dw = K.dwconv_bwd_weight(dev(dz), dev(yraw), dev(coef), dev(xin), dev(st), R.ACT_SILU, k, s, pt, pl)
    close(got_dw, dw, 5e-3 if rd == torch.bfloat16 else 2e-4, "dwconv_bwd_weight")
    if k == 3 and s == 1:
        # the one-kernel form (csrc/dfd_dwbwdf.hip): same per-element arithmetic -> dzin bit for bit; its sums follow its own tiling
        fz, fparts, fn, fdw = K.dwconv_bwd_fused(dev(dz), dev(yraw), dev(coef), dev(w), dev(xin), dev(st), R.ACT_SILU, k, s, pt, pl)
        assert torch.equal(fz, dzin), "fused data gradient differs from the two-kernel path"
        close(sum_parts(fparts, fn, C), want_sums, 2e-3, "dwconv_bwd_fused stats")
        close(fdw, dw, 5e-3 if rd == torch.bfloat16 else 2e-4, "dwconv_bwd_fused weight gradient")
        fz2, _, _, fdw2 = K.dwconv_bwd_fused(dev(dz), dev(yraw), dev(coef), dev(w), dev(xin), dev(st), R.ACT_SILU, k, s, pt, pl)
        assert torch.equal(fz, fz2) and torch.equal(fdw, fdw2), "fused backward is not reproducible"


@pytest.mark.parametrize("case", [(5, 28, 28, 240, 5, 1, 2, 2), (6, 30, 30, 144, 3, 1, 1, 1), (12, 33, 35, 96, 3, 2, 1, 1), (7, 14, 14, 672, 5, 1, 2, 2)])
def test_dwconv_grid_knobs_change_the_partial_rows_not_the_tensors(case):
    """dfd_tune keys 8-11 size the grids of the vector-unit depthwise kernels.  The small cases above run one work item per workgroup;
    here the grid is squeezed so that every workgroup walks several items (what a batch-256 launch does): the tensors must not change
    by a bit, the statistics and the weight gradient (other partial rows, other summation order) within rounding."""
    from deepfakedetection_amd._lib import load

    K = _k()
    lib = load()
    rd = torch.bfloat16
    N, H, W, C, k, s, pt, pl = case
    Ho, Wo = -(-H // s), -(-W // s)
    x, dz, yraw = dev(gen((N, H, W, C), 21, rd)), dev(gen((N, Ho, Wo, C), 22, rd)), dev(gen((N, Ho, Wo, C), 23, rd))
    w = dev(gen((C, 1, k, k), 24, torch.float32, 0.3))
    st = dev(rand_state(C, 25))
    coef = dev(torch.stack([torch.full((C,), 0.8), torch.full((C,), 0.05), torch.full((C,), -0.01)]))

    def run():
        y, p, n = K.dwconv_fwd(x, st, R.ACT_SILU, w, k, s, pt, pl, Ho, Wo, stats=True)
        fs = sum_parts(p, n, C)                          # (the partial rows live in a scratch buffer the next call reuses)
        dx, q, m = K.dwconv_bwd_data(dz, yraw, coef, w, x, st, R.ACT_SILU, (N, H, W, C), k, s, pt, pl)
        bs = sum_parts(q, m, C)
        dw = K.dwconv_bwd_weight(dz, yraw, coef, x, st, R.ACT_SILU, k, s, pt, pl)
        return y, fs, n, dx, bs, m, dw.float().cpu()

    lib.dfd_tune(0, 0)                                   # the vector-unit forward for every shape
    try:
        ref = run()
        for key in (8, 9, 10):
            assert lib.dfd_tune(key, 16) == 0
        assert lib.dfd_tune(11, 3) == 0
        got = run()
    finally:
        for key in (8, 9, 10):
            lib.dfd_tune(key, 1024)
        lib.dfd_tune(11, 32)
        lib.dfd_tune(0, 1)
    assert got[2] < ref[2] and got[5] < ref[5], f"the squeezed grid has as many partial rows as the default one: {got[2]} / {ref[2]}, {got[5]} / {ref[5]}"
    assert torch.equal(got[0], ref[0]) and torch.equal(got[3], ref[3]), "depthwise results depend on the grid"
    close(got[1], ref[1], 1e-3, "forward statistics")
    close(got[4], ref[4], 2e-3, "data-gradient statistics")
    close(got[6], ref[6], 5e-3, "weight gradient")


PW_CASES = [
    # N, HW, K, Nout
    (2, 49, 16, 96),
    (3, 100, 96, 24),
    (2, 196, 240, 40),
    (1, 333, 1152, 320),
    (4, 64, 24, 144),
    (2, 49, 320, 1280),
    (2, 130, 40, 8),
]


# rows well above one pass of the persistent grids (wave-autonomous NT kernel: several 128/256-row tiles
# per workgroup, two-deep register prefetch), ragged so that the last tile and the last wave are partial
PW_LARGE_CASES = [(3, 70001, 32, 16), (2, 100003, 16, 96), (3, 66001, 144, 24), (3, 66003, 24, 144), (2, 99001, 40, 240)]


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("case", PW_CASES)
def test_pwconv_fwd(case, mode, rd):
    _pwconv_fwd_case(case, mode, rd)


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("case", PW_LARGE_CASES)
def test_pwconv_fwd_large_m(case, mode, rd):
    _pwconv_fwd_case(case, mode, rd)


def _pwconv_fwd_case(case, mode, rd):
    K = _k()
    N, HW, Kd, No = case
    a = gen((N, HW, 1, Kd), 11, rd)
    a2 = gen((N, HW, 1, Kd), 12, rd)
    w = gen((No, Kd), 13, torch.float32, Kd ** -0.5)
    st = rand_state(Kd, 14)
    gate = torch.rand((N, Kd), generator=torch.Generator().manual_seed(15))
    coef3 = rand_state(Kd, 16)[:3].contiguous()
    dst = dev(st)  # the Prologue struct holds raw pointers: keep the tensors alive
    if mode == 0:
        pro, A = None, a.float()
    elif mode == 1:
        pro, A = K.pro_bn_act(dst, R.ACT_SILU), R.prologue(a.float().view(N, HW, Kd), 1, rd, R.ACT_SILU, st)
    elif mode == 2:
        dgate = dev(gate)
        pro = K.pro_bn_act_gate(dst, R.ACT_SILU, dgate, HW)
        A = R.prologue(a.float().view(N, HW, Kd), 2, rd, R.ACT_SILU, st, gate=gate)
    else:
        da2, dcoef = dev(a2), dev(coef3)
        pro = K.pro_affine2(da2, dcoef)
        A = R.prologue(a.float().view(N, HW, Kd), 3, rd, coef=coef3, a2=a2.float().view(N, HW, Kd))
    w_nk, _ = K.prep_weights(dev(w), rd, True, False)
    want = R.rnd(A.reshape(N, HW, 1, Kd) @ R.rnd(w, rd).t(), rd)
    stats = mode != 3
    out, parts, n = K.pwconv(dev(a), pro, w_nk, None, stats=stats)
    close(out, want, tol(rd), f"pwconv mode {mode}")
    if stats:
        close(sum_parts(parts, n, No), R.stats_sums(out.float().cpu()), 1e-3, "pwconv stats")
    if mode in (0, 3):
        res = gen((N, HW, 1, No), 17, rd)
        out2, _, _ = K.pwconv(dev(a), pro, w_nk, dev(res), stats=False)
        close(out2, R.rnd(want + res.float(), rd), tol(rd), "pwconv residual")


# mid-size layers (8 k .. 64 k rows): the LDS-DMA ring kernel k_pw_ntd (csrc/dfd_pwntd.hip).  Ragged last row tiles, K with a partial
# last 64-wide step (80, 112, 240, 672), one and several column tiles, a 64-row tile that spans three images (HW 49 / 25)
PW_MID_CASES = [(256, 49, 1152, 192), (67, 196, 480, 80), (67, 196, 80, 480), (50, 197, 112, 672), (131, 64, 672, 112),
                (401, 25, 240, 40), (200, 49, 192, 320), (180, 49, 320, 1280),
                # the smallest shapes it takes: 17 row tiles, K = 128 + 8, outputs that are not a multiple of 16 channels (three K steps or
                # more: with fewer the panel-resident kernel k_pw_ntw, which dfd_pwconv_fwd asks first, keeps such small shapes)
                (17, 64, 136, 24), (10, 103, 192, 8), (21, 50, 200, 56)]


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("case", PW_MID_CASES)
def test_pwconv_fwd_mid_m_ring_kernel(case, mode):
    """Against the oracle (as every pwconv case), AND against the register-staged kernel bit for bit: same prologue arithmetic, same
    order of the 32-wide MFMA steps over k.  The plan function is asserted so that a silent fallback cannot pass for the ring kernel."""
    K = _k()
    lib = K._L()
    N, HW, Kd, No = case
    assert lib.dfd_pw_ntd_plan(N * HW, Kd, No) > 0
    _pwconv_fwd_case(case, mode, torch.bfloat16)
    a = gen((N, HW, 1, Kd), 11, torch.bfloat16)
    a2 = gen((N, HW, 1, Kd), 12, torch.bfloat16)
    w = gen((No, Kd), 13, torch.float32, Kd ** -0.5)
    w_nk, _ = K.prep_weights(dev(w), torch.bfloat16, True, False)
    dst, dgate, da2, dcoef = dev(rand_state(Kd, 14)), dev(torch.rand((N, Kd), generator=torch.Generator().manual_seed(15))), dev(a2), dev(rand_state(Kd, 16)[:3].contiguous())
    pro = None if mode == 0 else (K.pro_bn_act_gate(dst, R.ACT_SILU, dgate, HW) if mode == 2 else K.pro_affine2(da2, dcoef))
    res = dev(gen((N, HW, 1, No), 17, torch.bfloat16)) if mode == 3 else None
    stats = mode != 3
    def run():                                  # (the partial rows live in a scratch buffer the next call reuses: copy them)
        out, parts, n = K.pwconv(dev(a), pro, w_nk, res, stats=stats)
        return out, (parts[: n * 2 * No].clone() if stats else None), n

    ring, ring2 = run(), run()
    try:
        assert lib.dfd_tune(4, 0) == 0
        tiled = run()
    finally:
        lib.dfd_tune(4, 1)
    assert torch.equal(ring[0], tiled[0]), "ring kernel and tile kernel disagree"
    assert torch.equal(ring[0], ring2[0])
    if stats:
        assert ring[2] == (N * HW + 63) // 64
        assert torch.equal(ring[1], ring2[1]), "statistics are not reproducible"
        close(sum_parts(ring[1], ring[2], No), sum_parts(tiled[1], tiled[2], No), 1e-4, "ring vs tile statistics")


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("qmode", [0, 1, 2])
@pytest.mark.parametrize("case", PW_CASES)
def test_pwconv_wgrad(case, qmode, rd):
    K = _k()
    N, HW, Nj, Ni = case
    p = gen((N, HW, 1, Ni), 21, rd)
    p2 = gen((N, HW, 1, Ni), 22, rd)
    q = gen((N, HW, 1, Nj), 23, rd)
    coef3 = rand_state(Ni, 24)[:3].contiguous()
    st = rand_state(Nj, 25)
    gate = torch.rand((N, Nj), generator=torch.Generator().manual_seed(26))
    dp2, dcoef, dst, dgate = dev(p2), dev(coef3), dev(st), dev(gate)
    P = R.prologue(p.float().view(N, HW, Ni), 3, rd, coef=coef3, a2=p2.float().view(N, HW, Ni))
    if qmode == 0:
        pq, Q = None, q.float().view(N, HW, Nj)
    elif qmode == 1:
        pq, Q = K.pro_bn_act(dst, R.ACT_SILU), R.prologue(q.float().view(N, HW, Nj), 1, rd, R.ACT_SILU, st)
    else:
        pq = K.pro_bn_act_gate(dst, R.ACT_SILU, dgate, HW)
        Q = R.prologue(q.float().view(N, HW, Nj), 2, rd, R.ACT_SILU, st, gate=gate)
    want = P.reshape(-1, Ni).t().double() @ Q.reshape(-1, Nj).double()
    got = K.pwconv_wgrad(dev(p), K.pro_affine2(dp2, dcoef), dev(q), pq)
    close(got, want.float(), 3e-3 if rd == torch.bfloat16 else 2e-4, f"pwconv_wgrad q{qmode}")
    # plain p
    want2 = p.float().reshape(-1, Ni).t().double() @ Q.reshape(-1, Nj).double()
    got2 = K.pwconv_wgrad(dev(p), None, dev(q), pq)
    close(got2, want2.float(), 3e-3 if rd == torch.bfloat16 else 2e-4, "pwconv_wgrad plain p")

# (N, HW, Ni, Nj): row counts above the wave-autonomous weight-gradient kernel's threshold (196,608 rows),
# ragged so that the last 32-row steps and the last workgroup are partial
TNW_CASES = [(4, 50000, 16, 32), (3, 70001, 24, 96), (4, 50003, 96, 16), (2, 100003, 32, 8), (3, 66001, 24, 144),
             (3, 66003, 144, 24), (2, 99001, 8, 136)]


@pytest.mark.parametrize("case", TNW_CASES)
def test_pwconv_wgrad_large_m(case):
    """bf16, M >= 196,608: the narrow x wide layers of EfficientNet blocks 0-1 run k_pw_tnw (wave-private
    tiles, no workgroup barrier); both operand orders, every prologue the engine uses there."""
    K = _k()
    rd = torch.bfloat16
    N, HW, Ni, Nj = case
    p = gen((N, HW, 1, Ni), 61, rd, 0.5)
    p2 = gen((N, HW, 1, Ni), 62, rd, 0.5)
    q = gen((N, HW, 1, Nj), 63, rd, 0.5)
    coef3 = rand_state(Ni, 64)[:3].contiguous()
    st = rand_state(Nj, 65)
    gate = torch.rand((N, Nj), generator=torch.Generator().manual_seed(66))
    P = R.prologue(p.float().view(N, HW, Ni), 3, rd, coef=coef3, a2=p2.float().view(N, HW, Ni))
    Qg = R.prologue(q.float().view(N, HW, Nj), 2, rd, R.ACT_SILU, st, gate=gate)
    pro_p = K.pro_affine2(dev(p2), dev(coef3))
    pro_g = K.pro_bn_act_gate(dev(st), R.ACT_SILU, dev(gate), HW)
    # project-conv form: affine2 on p, BN + SiLU + gate on q
    want = P.reshape(-1, Ni).t().double() @ Qg.reshape(-1, Nj).double()
    close(K.pwconv_wgrad(dev(p), pro_p, dev(q), pro_g), want.float(), 4e-3, "wgrad affine2 x gate")
    # expand-conv form: affine2 on p, raw q
    want = P.reshape(-1, Ni).t().double() @ q.float().reshape(-1, Nj).double()
    close(K.pwconv_wgrad(dev(p), pro_p, dev(q), None), want.float(), 4e-3, "wgrad affine2 x raw")
    # no prologue at all, and repeatability (fixed-order reductions)
    want = p.float().reshape(-1, Ni).t().double() @ q.float().reshape(-1, Nj).double()
    got = K.pwconv_wgrad(dev(p), None, dev(q), None)
    close(got, want.float(), 4e-3, "wgrad raw x raw")
    assert torch.equal(got, K.pwconv_wgrad(dev(p), None, dev(q), None)), "weight gradient is not reproducible"


ROW_CASES = [(2, 7, 7, 1152), (3, 14, 14, 40), (2, 28, 28, 144), (1, 56, 56, 24), (4, 5, 3, 672), (2, 7, 7, 1280)]


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", ROW_CASES)
def test_rowpass(case, rd):
    K = _k()
    N, H, W, C = case
    y = gen((N, H, W, C), 31, rd)
    g = gen((N, H, W, C), 32, rd)
    res = gen((N, H, W, C), 33, rd)
    gamma = 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(34))
    beta = torch.randn(C, generator=torch.Generator().manual_seed(35)) * 0.2
    rs = 0.5 + torch.rand(N, generator=torch.Generator().manual_seed(36))
    st = R.bn_state(y.float(), gamma, beta, 1e-3)
    # apply (+res, +row scale)
    z = st[0] * y.float() + st[1]
    close(K.bn_act_apply(dev(y), dev(st), R.ACT_SILU), R.rnd(R.act_fwd(z, 1), rd), tol(rd), "apply silu")
    want = R.rnd(z * rs[:, None, None, None] + res.float(), rd)
    close(K.bn_act_apply(dev(y), dev(st), 0, dev(res), dev(rs)), want, tol(rd), "apply res rs")
    # bn_bwd_reduce + finalize
    parts, n = K.bn_bwd_reduce(dev(g), dev(y), dev(st), None)
    coef, dgam, dbet = K.bn_bwd_finalize(parts, n, N * H * W, dev(gamma), dev(st), True)
    wc, wdg, wdb = R.bn_bwd_coef(g.float(), y.float(), gamma, st)
    close(coef, wc, 2e-3, "bn bwd coef")
    close(dgam, wdg, 2e-3, "dgamma")
    close(dbet, wdb, 2e-3, "dbeta")
    parts, n = K.bn_bwd_reduce(dev(g), dev(y), dev(st), dev(rs))
    coef2, _, _ = K.bn_bwd_finalize(parts, n, N * H * W, dev(gamma), dev(st), True)
    wc2, _, _ = R.bn_bwd_coef(g.float() * rs[:, None, None, None], y.float(), gamma, st)
    close(coef2, wc2, 2e-3, "bn bwd coef rs")
    # bias gradient of a layer without statistics: column sums of g (times the row scale), into a given destination too
    for scale in (None, rs):
        want_db = (g.float() * (1.0 if scale is None else scale[:, None, None, None])).sum((0, 1, 2))
        got_db = K.bias_grad(dev(g), None if scale is None else dev(scale))
        close(got_db, want_db, 2e-3, "bias grad")
        slot = torch.full((C,), 7.0, device="cuda")
        assert K.bias_grad(dev(g), None if scale is None else dev(scale), out=slot) is slot and torch.equal(slot, got_db)
    # act_bn_bwd, three modes
    gate = torch.rand((N, C), generator=torch.Generator().manual_seed(37))
    dpool = torch.randn((N, C), generator=torch.Generator().manual_seed(38))
    xhat = (y.float() - st[2]) * st[3]
    for mode, (D, gt, dp) in enumerate([(g, None, None), (g, gate, dpool), (None, None, dpool)]):
        if mode == 0:
            da = g.float()
        elif mode == 1:
            da = g.float() * gate[:, None, None, :] + dpool[:, None, None, :] / (H * W)
        else:
            da = (dpool[:, None, None, :] / (H * W)).expand(N, H, W, C)
        want = R.rnd(da * R.act_grad(z, 1), rd)
        dz, parts, n = K.act_bn_bwd(dev(D), dev(y), dev(gt), dev(dp), dev(st), R.ACT_SILU)
        close(dz, want, tol(rd), f"act_bn_bwd mode {mode}")
        got = dz.float().cpu()
        ws = torch.stack([got.reshape(-1, C).double().sum(0), (got * xhat).reshape(-1, C).double().sum(0)]).float()
        close(sum_parts(parts, n, C), ws, 2e-3, f"act_bn_bwd sums {mode}")
    # pools
    a = R.rnd(R.act_fwd(z, 1), rd)
    close(K.pool_act(dev(y), dev(st), R.ACT_SILU), a.mean((1, 2)), 1e-3, "pool_act")
    close(K.pool_bwd_reduce(dev(g), dev(y), dev(st), R.ACT_SILU), (a * g.float()).sum((1, 2)), 2e-3, "pool_bwd")
    close(K.scale_rows(dev(g), dev(rs)), R.rnd(g.float() * rs[:, None, None, None], rd), tol(rd), "scale_rows")

@pytest.mark.parametrize("rd", DT)
def test_pool_split_over_workgroups(rd):
    """Images large enough that the pooling kernels split H*W over several workgroups (partial
    vectors + last-arrival sum); repeated calls must leave the arrival counters re-armed."""
    K = _k()
    N, H, W, C = 3, 56, 56, 96
    y, g = gen((N, H, W, C), 91, rd), gen((N, H, W, C), 92, rd)
    st = torch.zeros((4, C))
    st[0] = torch.rand(C, generator=torch.Generator().manual_seed(93)) + 0.5
    st[1] = torch.randn(C, generator=torch.Generator().manual_seed(94)) * 0.1
    st[3] = 1.0
    assert int(K._L().dfd_pool_ws(K._dt(dev(y)), N, H * W, C)) > 0, "shape was meant to trigger the split path"
    z = y.float() * st[0] + st[1]
    a = R.rnd(R.act_fwd(z, 1), rd)
    for _ in range(3):
        close(K.pool_act(dev(y), dev(st), R.ACT_SILU), a.mean((1, 2)), 1e-3, "pool_act split")
        close(K.pool_bwd_reduce(dev(g), dev(y), dev(st), R.ACT_SILU), (a * g.float()).sum((1, 2)), 2e-3, "pool_bwd split")


@pytest.mark.parametrize("case", [(4, 32, 8), (3, 1152, 48), (2, 96, 4), (5, 2304, 96)])
def test_bn_finalize_and_eval(case):
    K = _k()
    N, C, _ = case
    y = gen((N, 6, 5, C), 41, torch.float32, 2.0) + 0.7
    w = gen((C, 1, 3, 3), 42, torch.float32)
    gamma = 0.5 + torch.rand(C)
    beta = torch.randn(C) * 0.1
    rm, rv = torch.randn(C) * 0.1, 0.5 + torch.rand(C)
    bn = K.BNParams(dev(gamma), dev(beta), dev(rm.clone()), dev(rv.clone()), 0.01, 1e-3)
    yy, parts, n = K.dwconv_fwd(dev(y), None, 0, dev(w), 3, 1, 1, 1, 6, 5, stats=True)
    st = K.bn_finalize(parts, n, N * 30, bn)
    yc = yy.float().cpu()
    close(st, R.bn_state(yc, gamma, beta, 1e-3), 1e-4, "bn_finalize state")
    flat = yc.reshape(-1, C)
    close(bn.running_mean, 0.99 * rm + 0.01 * flat.mean(0), 1e-5, "running_mean")
    close(bn.running_var, 0.99 * rv + 0.01 * flat.var(0, unbiased=True), 1e-4, "running_var")
    bn2 = K.BNParams(dev(gamma), dev(beta), dev(rm), dev(rv), 0.01, 1e-3)
    st2 = K.bn_eval_coeffs(bn2)
    rstd = 1 / torch.sqrt(rv + 1e-3)
    close(st2, torch.stack([gamma * rstd, beta - rm * gamma * rstd, rm, rstd]), 1e-5, "bn_eval")


@pytest.mark.parametrize("rd", DT)
def test_bn_statistics_with_a_large_mean(rd):
    """Variance is E[x^2] - mean^2 from f32 partial sums combined in double: bound the loss for a
    channel whose mean is 30x its standard deviation, at the largest row count of the benchmark (N*H*W = 3.2 M)."""
    K = _k()
    N, H, W, C = 16, 448, 448, 16
    g = torch.Generator().manual_seed(61)
    x = (torch.randn((N, H, W, C), generator=g) + 30.0).to(rd)
    w = torch.zeros((C, 1, 3, 3))
    w[:, 0, 1, 1] = 1.0                                    # identity depthwise conv: y = x, statistics of x
    gamma, beta = torch.ones(C), torch.zeros(C)
    bn = K.BNParams(dev(gamma), dev(beta), dev(torch.zeros(C)), dev(torch.ones(C)), 0.1, 1e-5)
    y, parts, n = K.dwconv_fwd(dev(x), None, 0, dev(w), 3, 1, 1, 1, H, W, stats=True)
    st = K.bn_finalize(parts, n, N * H * W, bn).cpu()
    flat = x.float().reshape(-1, C).double()
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    assert torch.allclose(st[2].double(), mean, rtol=0, atol=1e-4), (st[2].double() - mean).abs().max()
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    rel = ((st[3].double() - rstd).abs() / rstd).max().item()
    assert rel < 2e-3, rel


@pytest.mark.parametrize("case", [(4, 32, 8), (3, 1152, 48), (2, 96, 4), (5, 2304, 96)])
def test_se_fc(case):
    K = _k()
    N, C, Rr = case
    g = torch.Generator().manual_seed(51)
    pooled = torch.rand((N, C), generator=g)
    w1 = (torch.randn((Rr, C), generator=g) * C ** -0.5).requires_grad_(True)
    b1 = (torch.randn(Rr, generator=g) * 0.1).requires_grad_(True)
    w2 = (torch.randn((C, Rr), generator=g) * Rr ** -0.5).requires_grad_(True)
    b2 = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    pr = pooled.clone().requires_grad_(True)
    hpre, gate = R.se_fc(pr, w1, b1, w2, b2, R.ACT_SILU)
    dgate = torch.randn((N, C), generator=g)
    gate.backward(dgate)
    h2, g2, w2t = K.se_fc_fwd(dev(pooled), dev(w1.detach()), dev(b1.detach()), dev(w2.detach()), dev(b2.detach()), R.ACT_SILU)
    close(w2t, w2.detach().t(), 0.0, "se w2t")
    close(h2, hpre, 1e-4, "se hpre")
    close(g2, gate, 1e-4, "se gate")
    # the kernel takes d(loss)/d(gate) and applies sigmoid' itself
    dp, dw1, db1, dw2, db2 = K.se_fc_bwd(dev(dgate), g2, h2, dev(pooled), dev(w1.detach()), w2t, R.ACT_SILU)
    close(dp, pr.grad, 2e-4, "se dpooled")
    close(dw1, w1.grad, 2e-4, "se dw1")
    close(db1, b1.grad, 2e-4, "se db1")
    close(dw2, w2.grad, 2e-4, "se dw2")
    close(db2, b2.grad, 2e-4, "se db2")


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", [(3, 56, 56, 96, 4), (5, 7, 7, 1152, 48), (4, 14, 14, 480, 20), (2, 112, 112, 32, 8)])
def test_se_branch_in_one_call_each_way(rd, case):
    """dfd_se_fwd / dfd_se_bwd (pooling + one workgroup per image) against the separate entry points: same
    summation orders, so the results are identical, with and without the H*W split of the pooling kernel."""
    K = _k()
    N, H, W, C, Rr = case
    g = torch.Generator().manual_seed(77)
    y, D = dev(gen((N, H, W, C), 78, rd)), dev(gen((N, H, W, C), 79, rd))
    st = torch.zeros((4, C))
    st[0] = torch.rand(C, generator=g) + 0.5
    st[1] = torch.randn(C, generator=g) * 0.1
    st[3] = 1.0
    st = dev(st)
    w1 = dev(torch.randn((Rr, C), generator=g) * C ** -0.5)
    b1 = dev(torch.randn(Rr, generator=g) * 0.1)
    w2 = dev(torch.randn((C, Rr), generator=g) * Rr ** -0.5)
    b2 = dev(torch.randn(C, generator=g) * 0.1)
    pooled0 = K.pool_act(y, st, R.ACT_SILU)
    h0, g0, w2t = K.se_fc_fwd(pooled0, w1, b1, w2, b2, R.ACT_SILU)
    for prepared in (None, w2t):
        pooled, hpre, gate, w2t1 = K.se_fwd(y, st, R.ACT_SILU, w1, b1, w2, b2, R.ACT_SILU, prepared)
        for a, b, name in ((pooled, pooled0, "pooled"), (hpre, h0, "hpre"), (gate, g0, "gate"), (w2t1, w2t, "w2t")):
            assert torch.equal(a, b), name
    dgate = K.pool_bwd_reduce(D, y, st, R.ACT_SILU)
    want = K.se_fc_bwd(dgate, g0, h0, pooled0, w1, w2t, R.ACT_SILU)
    got = K.se_bwd(D, y, st, R.ACT_SILU, g0, h0, pooled0, w1, w2t, R.ACT_SILU)
    for a, b, name in zip(got, want, ("dpooled", "dw1", "db1", "dw2", "db2")):
        assert torch.equal(a, b), name
    dp_only = K.se_bwd(D, y, st, R.ACT_SILU, g0, h0, pooled0, w1, w2t, R.ACT_SILU, want_param_grads=False)
    assert torch.equal(dp_only[0], want[0]) and dp_only[1] is None
    # the FC weight gradients as passenger workgroups of the next launch (dfd_act_bn_bwd_se): same bits as the launch of their own,
    # and the host launch's own results (dz, statistics) are untouched
    dz0, parts0, n0 = K.act_bn_bwd(D, y, g0, want[0], st, R.ACT_SILU)
    parts0 = parts0[: n0 * 2 * C].clone()
    later = K.se_bwd(D, y, st, R.ACT_SILU, g0, h0, pooled0, w1, w2t, R.ACT_SILU, defer_wgrad=True)
    assert torch.equal(later[0], want[0]) and later[5] is not None
    for t in later[1:5]:
        t.fill_(float("nan"))                   # (nothing has computed them yet)
    dz1, parts1, n1 = K.act_bn_bwd(D, y, g0, later[0], st, R.ACT_SILU, se_job=later[5])
    assert n1 == n0 and torch.equal(dz1, dz0) and torch.equal(parts1[: n1 * 2 * C], parts0)
    for a, b, name in zip(later[1:5], want[1:], ("dw1", "db1", "dw2", "db2")):
        assert torch.equal(a, b), f"passenger {name}"
    none_job = K.se_bwd(D, y, st, R.ACT_SILU, g0, h0, pooled0, w1, w2t, R.ACT_SILU, want_param_grads=False, defer_wgrad=True)
    assert none_job[5] is None and none_job[1] is None


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", [(2, 32, 32, 32, 0), (2, 33, 31, 40, 1), (1, 224, 224, 32, 1), (3, 64, 60, 64, 1), (2, 50, 36, 16, 1),
                                  (2, 40, 44, 48, 0), (5, 224, 224, 64, 1), (2, 38, 300, 32, 1)])
def test_stem(case, rd):
    """stem convolution and its weight gradient.  bf16 with Cout % 16 == 0 and W % 4 == 0 runs the matrix-core kernels
    (dfd_stem.hip: ragged last 16- / 32-pixel tile, odd row counts, 48 channels, rows wider than one staging pass), everything
    else the f32-FMA kernels."""
    K = _k()
    N, H, W, Co, pad = case
    Ho, Wo = -(-H // 2), -(-W // 2)
    x = gen((N, H, W, 3), 61, torch.float32)
    w = gen((Co, 3, 3, 3), 62, torch.float32, 0.3)
    want = R.stem_conv_fwd(x, w, 2, pad, pad, Ho, Wo, rd)
    y, parts, n = K.stem_conv_fwd(dev(x), dev(w), rd, 2, pad, pad, Ho, Wo)
    close(y, want, tol(rd), "stem fwd")
    close(sum_parts(parts, n, Co), R.stats_sums(y.float().cpu()), 1e-3, "stem stats")
    dz = gen((N, Ho, Wo, Co), 63, rd)
    yraw = gen((N, Ho, Wo, Co), 64, rd)
    coef = rand_state(Co, 65)[:3].contiguous()
    dy = R.rnd(coef[0] * dz.float() + coef[1] * yraw.float() + coef[2], rd)
    want_dw = R.stem_conv_wgrad(x, dy, 3, 2, pad, pad, rd)
    got = K.stem_conv_wgrad(dev(x), dev(dz), dev(yraw), dev(coef), 3, 2, pad, pad)
    close(got, want_dw, 5e-3 if rd == torch.bfloat16 else 2e-4, "stem wgrad")
    want_plain = R.stem_conv_wgrad(x, dz.float(), 3, 2, pad, pad, rd)
    got_plain = K.stem_conv_wgrad(dev(x), dev(dz), None, None, 3, 2, pad, pad)
    close(got_plain, want_plain, 5e-3 if rd == torch.bfloat16 else 2e-4, "stem wgrad without the BN-backward map")


@pytest.mark.parametrize("J", [2, 10, 1000])
def test_head_loss(J):
    K = _k()
    N, Kd = 6, 1280
    g = torch.Generator().manual_seed(71)
    x = torch.randn((N, Kd), generator=g)
    w = (torch.randn((J, Kd), generator=g) * Kd ** -0.5).requires_grad_(True)
    b = (torch.randn(J, generator=g) * 0.1).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    logits = xr @ w.t() + b
    tg = torch.randint(0, J, (N,), generator=g)
    loss = R.ce_label_smooth(logits, tg, 0.1)
    loss.backward()
    lg = K.linear_fwd(dev(x), dev(w.detach()), dev(b.detach()))
    close(lg, logits, 1e-4, "linear fwd")
    l2, dlog = K.ce_loss(lg, dev(tg), 0.1, 1.0, True)
    close(l2.reshape(1), loss.detach().reshape(1), 1e-5, "ce loss")
    dx, dw, db = K.linear_bwd(dlog, dev(x), dev(w.detach()), True, True, True)
    close(dx, xr.grad, 2e-4, "linear dx")
    close(dw, w.grad, 2e-4, "linear dw")
    close(db, b.grad, 2e-4, "linear db")
    probs, preds = K.softmax_argmax(lg, True)
    want_p = torch.softmax(logits.detach(), 1)
    close(probs, want_p, 1e-5, "softmax")
    assert torch.equal(preds.cpu(), want_p.argmax(1))
    u = torch.rand((N, Kd), generator=g)
    close(K.dropout(dev(x), dev(u), 0.2), torch.where(u >= 0.2, x / 0.8, torch.zeros_like(x)), 1e-6, "dropout")


def test_adamw_matches_torch():
    K = _k()
    g = torch.Generator().manual_seed(81)
    shapes = [(1280, 10), (10,), (32, 3, 3, 3), (70001,)]
    params = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in params]
    opt = torch.optim.AdamW(ref, lr=3e-3, weight_decay=5e-2)
    dparams = [p.cuda() for p in params]
    ms = [torch.zeros_like(p) for p in dparams]
    vs = [torch.zeros_like(p) for p in dparams]
    for step in range(1, 4):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
        dgrads = [gr.cuda() for gr in grads]
        rows = []
        for p, gr, m, v in zip(dparams, dgrads, ms, vs):
            for off in range(0, p.numel(), 65536):
                cnt = min(65536, p.numel() - off)
                rows.append([p.data_ptr() + 4 * off, gr.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, cnt])
        table = torch.tensor(rows, dtype=torch.int64).cuda()
        hp = torch.tensor([3e-3, 0.9, 0.999, 1e-8, 5e-2, 1 - 0.9 ** step, 1 - 0.999 ** step, 1.0]).cuda()
        K.adamw_step(table, hp)
        torch.cuda.synchronize()
    for p, r in zip(dparams, ref):
        close(p, r.detach(), 1e-5, "adamw")


def test_image_prep_matches_cpu_transforms_bit_for_bit():
    """dfd_image_prep = RandomHorizontalFlip -> ToTensor -> Normalize -> RandomErasing(value=0) with the random
    decisions given: identical bits to the CPU transforms of deepfakedetection_amd.data."""
    from deepfakedetection_amd import data as D

    K = _k()
    N, H, W = 5, 37, 29
    g = torch.Generator().manual_seed(71)
    src = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    flip = torch.tensor([1, 0, 1, 0, 1], dtype=torch.uint8)
    erase = torch.tensor([[3, 4, 10, 7], [0, 0, 0, 0], [30, 20, 7, 9], [0, 0, 36, 28], [5, 5, 0, 3]], dtype=torch.int32)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    norm = D.Normalize(mean, std)
    want = []
    for n in range(N):
        img = src[n]
        if flip[n]:
            img = img.flip(1)
        t = norm(img.permute(2, 0, 1).to(torch.float32).div_(255.0))
        top, left, eh, ew = (int(v) for v in erase[n])
        if eh > 0:
            t = t.clone()
            t[:, top:top + eh, left:left + ew] = 0.0
        want.append(t)
    want = torch.stack(want)
    got = K.image_prep(src.cuda(), mean, std, flip.cuda(), erase.cuda())
    assert got.shape == (N, 3, H, W) and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got.cpu(), want), float((got.cpu() - want).abs().max())
    # no flip / erase: plain ToTensor + Normalize
    got2 = K.image_prep(src.cuda(), mean, std, None, None)
    want2 = torch.stack([norm(src[n].permute(2, 0, 1).to(torch.float32).div_(255.0)) for n in range(N)])
    assert torch.equal(got2.cpu(), want2)


@pytest.mark.parametrize("rd", DT)
def test_prep_weights_multi_matches_per_layer(rd):
    """dfd_prep_weights_multi (one launch for every derived weight of a network, >32 jobs = two launches) against
    the per-layer dfd_pw_prep_weights and a plain transpose for the squeeze-excite copies."""
    K = _k()
    g = torch.Generator().manual_seed(81)
    items = []
    for i in range(37):
        n, k = int(torch.randint(1, 40, (1,), generator=g)) * 8, int(torch.randint(1, 30, (1,), generator=g)) * 8
        w = torch.randn((n, k, 1, 1), generator=g).cuda()
        items.append((w, i % 3 != 2, True, i % 5 == 4))
    dw = K.DerivedWeights(items, rd)
    dw.refresh()
    for (w, want_nk, want_kn, is_se), (nk, kn) in zip(items, dw.out):
        if is_se:
            assert kn.dtype == torch.float32 and torch.equal(kn, w.reshape(w.shape[0], -1).t().contiguous())
            continue
        ref_nk, ref_kn = K.prep_weights(w, rd, True, True)
        if want_nk:
            assert torch.equal(nk, ref_nk)
        else:
            assert nk is None
        assert torch.equal(kn, ref_kn)
    assert dw.valid_for([w for w, *_ in items], rd) and not dw.valid_for([w for w, *_ in items][:-1], rd)


def test_batched_partial_sums_are_the_unbatched_ones():
    """kernels.sum_batch(): the final summations of the weight gradients launched inside are recorded and added by one
    pair of launches at the exit (dfd_sum_batch_begin / _end) — same bits as without; more than eight jobs flush early;
    a nested batch is a no-op; an unmatched end is refused."""
    K = _k()
    rd = torch.bfloat16
    cases = [(4096, 16, 96), (12544, 192, 1152), (50176, 80, 480), (300, 8, 8), (8192, 24, 144)] * 2     # ten jobs
    ops = [(dev(gen((1, M, 1, Ni), 10 + i, rd)), dev(gen((1, M, 1, Nj), 30 + i, rd))) for i, (M, Ni, Nj) in enumerate(cases)]
    dz, y = dev(gen((3, 14, 14, 48), 50, rd)), dev(gen((3, 14, 14, 48), 51, rd))
    x = dev(gen((3, 14, 14, 48), 52, rd))
    want = [K.pwconv_wgrad(p, None, q, None) for p, q in ops]
    want_dw = K.dwconv_bwd_weight(dz, None, None, x, None, R.ACT_NONE, 3, 1, 1, 1)
    with K.sum_batch():
        with K.sum_batch():                                   # nested: no-op
            got = [K.pwconv_wgrad(p, None, q, None) for p, q in ops]
        got_dw = K.dwconv_bwd_weight(dz, None, None, x, None, R.ACT_NONE, 3, 1, 1, 1)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert torch.equal(want_dw, got_dw)
    assert K._L().dfd_sum_batch_end() != 0                   # nothing open
    again = K.pwconv_wgrad(*ops[1][:1], None, ops[1][1], None)
    assert torch.equal(again, want[1])                       # the unbatched path is untouched afterwards


# ---- eval / inference form of the MBConv block: the producer applies its own BatchNorm + activation
@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", DW_CASES + [(4, 112, 112, 32, 3, 1, 1, 1), (3, 28, 28, 240, 5, 2, 1, 1)])
def test_dwconv_eval_form(case, rd):
    K = _k()
    N, H, W, C, k, s, pt, pl = case
    Ho, Wo = -(-H // s), -(-W // s)
    x = gen((N, H, W, C), 1, rd)
    w = gen((C, 1, k, k), 2, torch.float32, 0.3)
    st = rand_state(C, 3)
    # oracle: raw convolution rounded to the activation dtype, then act(scale*y+shift) rounded again
    raw = R.dwconv_fwd(x.float(), None, R.ACT_NONE, w, k, s, pt, pl, Ho, Wo, rd)
    want = R.rnd(R.act_fwd(raw.float() * st[0] + st[1], R.ACT_SILU), rd)
    a, parts, tiles = K.dwconv_eval(dev(x), dev(w), dev(st), R.ACT_SILU, k, s, pt, pl, Ho, Wo)
    close(a, want, tol(rd), "dwconv_eval a")
    # the training-form kernels on the same input: identical bits
    y, _, _ = K.dwconv_fwd(dev(x), None, R.ACT_NONE, dev(w), k, s, pt, pl, Ho, Wo, stats=False)
    a_ref = K.bn_act_apply(y, dev(st), R.ACT_SILU)
    assert torch.equal(a, a_ref), "eval-form depthwise output differs from dwconv_fwd + bn_act_apply"
    # channel sums of the stored tensor, one row per (tile, image)
    assert parts.shape == (tiles, N, C)
    got = parts.double().sum(0).cpu()
    ref = a.double().sum((1, 2)).cpu()
    assert float((got - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0) * (1 if rd == torch.float32 else 1)


@pytest.mark.parametrize("rd", DT)
@pytest.mark.parametrize("case", [(2 * 112 * 112, 16, 96), (3 * 56 * 56, 24, 144), (5 * 14 * 14, 80, 480), (7 * 7 * 7, 192, 1152),
                                  (64 * 112 * 112, 16, 96), (64 * 28 * 28, 40, 240)])
def test_pwconv_eval_form(case, rd):
    K = _k()
    M, Kd, Nout = case
    a = dev(gen((1, 1, M, Kd), 5, rd))
    w = dev(gen((Nout, Kd), 6, rd, 0.2))
    st = dev(rand_state(Nout, 7))
    got = K.pwconv_eval(a, w, st, R.ACT_SILU)
    y, _, _ = K.pwconv(a, None, w, None, stats=False)
    ref = K.bn_act_apply(y, st, R.ACT_SILU)
    assert torch.equal(got, ref), "eval-form 1x1 convolution differs from pwconv + bn_act_apply"
    if M <= 5 * 14 * 14:
        want = R.rnd(R.act_fwd(R.rnd(a.float().cpu() @ w.float().cpu().T, rd).float() * st[0].cpu() + st[1].cpu(), R.ACT_SILU), rd)
        close(got, want, tol(rd), "pwconv_eval")


@pytest.mark.parametrize("case", [(3, 56 * 56, 96, 4, 5), (5, 49, 1152, 48, 1), (2, 112 * 112, 32, 8, 16)])
def test_se_from_tile_sums(case):
    K = _k()
    N, HW, C, Rr, tiles = case
    parts = dev(gen((tiles, N, C), 1, torch.float32, 3.0))
    w1, b1 = dev(gen((Rr, C), 2, torch.float32, 0.1)), dev(gen((Rr,), 3, torch.float32, 0.1))
    w2, b2 = dev(gen((C, Rr), 4, torch.float32, 0.1)), dev(gen((C,), 5, torch.float32, 0.1))
    pooled, gate, _ = K.se_fwd_parts(parts, HW, w1, b1, w2, b2, R.ACT_SILU)
    want_pooled = (parts.double().sum(0) / HW).float().cpu()
    close(pooled, want_pooled, 1e-6, "pooled")
    _, want_gate = R.se_fc(want_pooled, w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu(), R.ACT_SILU)
    close(gate, want_gate, 1e-5, "gate")


def test_resize_crop_matches_pillow_bit_for_bit():
    """csrc/dfd_resize.hip restates Pillow's bilinear Image.resize (anti-aliased two-pass resampling with an 8-bit
    intermediate, fixed-point coefficients): BYTE work, so the device result must equal PIL's exactly — for the eval pipeline
    (Resize(shorter side) + CenterCrop, trainers/efficientnet.py:196-203), RandomResizedCrop boxes, up- and down-scaling,
    odd sizes, images smaller than the crop (black padding) and a 20x shrink."""
    import numpy as np
    from PIL import Image

    from deepfakedetection_amd import data as D

    K = _k()
    rng = np.random.default_rng(5)
    sizes = [(500, 375), (375, 500), (224, 224), (257, 257), (640, 427), (97, 131), (60, 40), (1, 1), (3000, 2000), (301, 299)]
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for w, h in sizes]
    # smooth content too (random noise hides off-by-one coefficient errors less well than gradients hide rounding ones)
    yy, xx = np.mgrid[0:333, 0:517]
    imgs.append(Image.fromarray(np.stack([(xx * 255 // 516), (yy * 255 // 332), ((xx + yy) % 256)], -1).astype(np.uint8)))
    out, enlarged = 224, 257
    # ---- eval pipeline: Resize(enlarged) -> CenterCrop(out)
    plan = D.PlanGeometry("center", out, enlarged)
    want = [np.array(D.CenterCrop(out)(D.Resize(enlarged)(im))) for im in imgs]
    (flat, jobs, meta), _ = D.collate_raw([(plan(im), 0) for im in imgs])
    got = K.resize_crop_u8(flat.cuda(), jobs.cuda(), len(imgs), out, out, int(meta[2])).cpu().numpy()
    for i, w in enumerate(want):
        assert np.array_equal(got[i], w), (i, imgs[i].size, int(np.abs(got[i].astype(int) - w.astype(int)).max()))
    # ---- RandomResizedCrop: same RNG calls, same boxes
    rrc = D.RandomResizedCrop(out, scale=(0.3, 1.0))
    torch.manual_seed(3); import random; random.seed(3); np.random.seed(3)
    state = torch.get_rng_state()
    want = [np.array(rrc(im)) for im in imgs]
    torch.set_rng_state(state); random.seed(3); np.random.seed(3)
    planr = D.PlanGeometry("rrc", out, rrc=rrc)
    (flat, jobs, meta), _ = D.collate_raw([(planr(im), 0) for im in imgs])
    got = K.resize_crop_u8(flat.cuda(), jobs.cuda(), len(imgs), out, out, int(meta[2])).cpu().numpy()
    for i, w in enumerate(want):
        assert np.array_equal(got[i], w), ("rrc", i, imgs[i].size)
    # ---- small-image branch: Resize(size + 4) + RandomCrop(size)
    torch.manual_seed(4)
    state = torch.get_rng_state()
    want = [np.array(D.RandomCrop(32)(D.Resize(36)(im))) for im in imgs[:6]]
    torch.set_rng_state(state)
    plans = D.PlanGeometry("random", 32, 36)
    (flat, jobs, meta), _ = D.collate_raw([(plans(im), 0) for im in imgs[:6]])
    got = K.resize_crop_u8(flat.cuda(), jobs.cuda(), 6, 32, 32, int(meta[2])).cpu().numpy()
    for i, w in enumerate(want):
        assert np.array_equal(got[i], w), ("small", i, imgs[i].size)
    # ---- and through the whole tail: same f32 tensor as ToTensor + Normalize of the PIL result
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    tail = D.GpuInputTail(mean, std)
    (raw, _) = D.collate_raw([(plan(im), 0) for im in imgs[:4]])
    x = tail(raw, "cuda").cpu()
    ref = torch.stack([D.Normalize(mean, std)(D.ToTensor()(D.CenterCrop(out)(D.Resize(enlarged)(im)))) for im in imgs[:4]])
    assert torch.equal(x, ref)


def test_oversize_image_is_preshrunk_by_the_worker_and_still_bit_exact():
    """An image whose shrink factor exceeds the device kernel's filter length (~46x) is resized in the worker by the PIL calls the
    kernel restates and travels with an identity plan (data.PlanGeometry): the batch no longer fails with DFD_EUNSUPPORTED and the
    result still equals the PIL pipeline byte for byte — mixed with ordinary images in one batch."""
    import numpy as np
    from PIL import Image

    from deepfakedetection_amd import data as D

    K = _k()
    rng = np.random.default_rng(9)
    sizes = [(4000, 3100), (400, 300), (64, 9000)]            # 64x / 6x / 140x shrink to a 48-pixel shorter side
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for w, h in sizes]
    out, enlarged = 40, 48
    plan = D.PlanGeometry("center", out, enlarged)
    want = [np.array(D.CenterCrop(out)(D.Resize(enlarged)(im))) for im in imgs]
    (flat, jobs, meta), _ = D.collate_raw([(plan(im), 0) for im in imgs])
    assert int(meta[2]) <= D.PlanGeometry.MAX_DEVICE_SHRINK
    got = K.resize_crop_u8(flat.cuda(), jobs.cuda(), len(imgs), out, out, int(meta[2])).cpu().numpy()
    for i, w in enumerate(want):
        assert np.array_equal(got[i], w), (i, imgs[i].size)
    rrc = D.RandomResizedCrop(out, scale=(0.9, 1.0))
    torch.manual_seed(2); import random; random.seed(2); np.random.seed(2)
    state = torch.get_rng_state()
    want = [np.array(rrc(im)) for im in imgs]
    torch.set_rng_state(state); random.seed(2); np.random.seed(2)
    planr = D.PlanGeometry("rrc", out, rrc=rrc)
    (flat, jobs, meta), _ = D.collate_raw([(planr(im), 0) for im in imgs])
    got = K.resize_crop_u8(flat.cuda(), jobs.cuda(), len(imgs), out, out, int(meta[2])).cpu().numpy()
    for i, w in enumerate(want):
        assert np.array_equal(got[i], w), ("rrc", i, imgs[i].size)


AUG_SIZES = [(224, 224), (37, 53), (1, 1), (228, 228), (64, 64), (5, 200)]


@pytest.mark.parametrize("size", AUG_SIZES)
def test_device_rotation_and_colour_jitter_match_the_oracle_byte_for_byte(size):
    """csrc/dfd_augment.hip against oracle/image_ref.py (itself pinned against Pillow): rotation modes incl. the 0 / 90 / 180 / 270
    special cases, every permutation class of the four colour operations, factors below and above 1, disabled operations."""
    import numpy as np

    from deepfakedetection_amd import data as D
    from oracle import image_ref as IR

    K = _k()
    h, w = size
    rng = np.random.default_rng(h * 1000 + w)
    n = 26
    imgs = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    imgs[1, : max(1, h // 2)] = 0
    imgs[2] = 255
    jobs = np.zeros((n, 16), dtype=np.int32)
    fl = jobs.view(np.float32)
    want = []
    perms = [list(p) for p in __import__("itertools").permutations(range(4))]
    for i in range(n):
        angle = [0.0, 180.0, 90.0, 270.0, 45.0, -7.3, 9.99, 0.001][i % 8] if i < 16 else float(rng.uniform(-10, 10))
        mode, coef = D.rotate_plan(w, h, angle)
        assert (mode, tuple(coef)) == IR.rotate_plan(w, h, angle)
        order = perms[i % 24]
        fb, fc, fs = (float(rng.uniform(0.0, 2.0)) for _ in range(3))
        dh = float(rng.uniform(-0.5, 0.5))
        enable = 15 if i % 5 else int(rng.integers(0, 16))
        jobs[i, 0] = mode; jobs[i, 1:7] = coef; jobs[i, 7:11] = order
        fl[i, 11], fl[i, 12], fl[i, 13] = fb, fc, fs
        jobs[i, 14] = IR.hue_delta(dh); jobs[i, 15] = enable
        # the oracle takes the f32 factors the job carries (the PIL transform hands Blend.c a C float too)
        ref = IR.rotate(imgs[i], angle)
        ref = IR.jitter(ref, order, float(fl[i, 11]) if enable & 1 else None, float(fl[i, 12]) if enable & 2 else None,
                        float(fl[i, 13]) if enable & 4 else None, dh if enable & 8 else None)
        want.append(ref)
    got = K.augment_u8(torch.from_numpy(imgs).cuda(), torch.from_numpy(jobs).cuda()).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], want[i]), (i, size, int((got[i] != want[i]).sum()), jobs[i].tolist())


def test_device_augment_draws_what_the_pil_transforms_draw():
    """GpuInputTail.sample_augment makes the RNG calls of data.RandomRotation followed by data.ColorJitter, so with one seed the
    device result equals the PIL pipeline's image exactly (the whole tail: + ToTensor + Normalize)."""
    import numpy as np
    from PIL import Image

    from deepfakedetection_amd import data as D

    rng = np.random.default_rng(4)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    tail = D.GpuInputTail(mean, std, rotate_degrees=10.0, jitter=(0.2, 0.2, 0.2, 0.05))
    rot, cj = D.RandomRotation(10), D.ColorJitter(0.2, 0.2, 0.2, 0.05)
    for trial in range(6):
        img = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
        torch.manual_seed(100 + trial); import random; random.seed(100 + trial); np.random.seed(100 + trial)
        state = torch.get_rng_state()
        want = D.Normalize(mean, std)(D.ToTensor()(cj(rot(Image.fromarray(img)))))
        torch.set_rng_state(state); random.seed(100 + trial); np.random.seed(100 + trial)
        got = tail(torch.from_numpy(img)[None], "cuda").cpu()[0]
        assert torch.equal(got, want), (trial, float((got - want).abs().max()))
    big = torch.zeros((1, 300, 300, 3), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="DFD_EUNSUPPORTED"):
        _k().augment_u8(big.cuda(), torch.zeros((1, 16), dtype=torch.int32).cuda())


@pytest.mark.parametrize("case", [(256 * 28 * 28, 96, 16, True), (256 * 28 * 28 + 37, 144, 24, True), (3 * 256 * 256 + 5, 144, 24, False),
                                  (2048 * 96 + 3, 128, 32, True), (2048 * 96, 64, 8, False),
                                  (256 * 28 * 28, 240, 40, True), (256 * 28 * 28 + 11, 192, 48, False), (2048 * 96 + 5, 160, 24, True)])
def test_expand_backward_in_one_pass(case):
    """dfd_pwconv_bwd_fused (csrc/dfd_pwtnw.hip, DG): data and weight gradient of the expand 1x1 layer from one pass over (dz, y) —
    against the f32 arithmetic of the oracle's ops (BN-backward map, two products) and, bit for bit, against the two kernels it
    replaces; ragged row counts, both step sizes (Cm <= 96: 32 rows, else 16), with and without the skip connection's gradient."""
    K = _k()
    M, Cm, Cin, with_res = case
    K._FUSE_EXPAND_WIDE = True           # the 192- / 240-wide instances are off by default (measured slower): tested all the same
    g = torch.Generator().manual_seed(M % 1000 + Cm)
    dz = (torch.randn((M, 1, 1, Cm), generator=g) * 0.5).to(torch.bfloat16).cuda()
    y = torch.randn((M, 1, 1, Cm), generator=g).to(torch.bfloat16).cuda()
    x = torch.randn((M, 1, 1, Cin), generator=g).to(torch.bfloat16).cuda()
    res = torch.randn((M, 1, 1, Cin), generator=g).to(torch.bfloat16).cuda() if with_res else None
    w = (torch.randn((Cm, Cin), generator=g) * Cin ** -0.5).cuda()
    coef = torch.stack([0.5 + torch.rand(Cm, generator=g), torch.randn(Cm, generator=g) * 0.1, torch.randn(Cm, generator=g) * 0.05]).cuda()
    w_nk, w_kn = K.prep_weights(w, torch.bfloat16, True, True)
    both = K.pwconv_bwd_fused(dz, y, coef, x, w_kn, res)
    assert both is not None, "shape expected to be served by the fused kernel"
    dx, dw = both
    pro = K.pro_affine2(y, coef)
    dx2, _, _ = K.pwconv(dz, pro, w_kn, res, stats=False)
    dw2 = K.pwconv_wgrad(dz, pro, x, None)
    assert torch.equal(dx, dx2), float((dx.float() - dx2.float()).abs().max())
    if Cm <= 144 and Cin <= 32:
        assert torch.equal(dw, dw2), float((dw - dw2).abs().max())      # the separate weight gradient is the same wave-autonomous kernel
    else:
        close(dw, dw2, 2e-5, "fused dw against the tiled kernel's (another split and summation order of the same f32 products)")
    # and against plain f32 arithmetic on the bf16-rounded d
    d = R.rnd(coef[0].cpu() * dz.float().cpu().view(M, Cm) + coef[1].cpu() * y.float().cpu().view(M, Cm) + coef[2].cpu(), torch.bfloat16)
    want_dx = d @ w.cpu().to(torch.bfloat16).float()
    if with_res:
        want_dx = R.rnd(want_dx, torch.bfloat16) + res.float().cpu().view(M, Cin)
    close(dx.view(M, Cin), want_dx, 1.6e-2, "fused expand backward dx")
    close(dw, d.t() @ x.float().cpu().view(M, Cin), 5e-3, "fused expand backward dw")
    K._FUSE_EXPAND_WIDE = False


def test_expand_backward_fused_declines_other_shapes():
    K = _k()
    assert not K._FUSE_EXPAND_WIDE
    for M, Cm, Cin in [(1000, 96, 16), (256 * 28 * 28, 256, 40), (256 * 28 * 28, 152, 24), (256 * 28 * 28, 240, 56)]:
        dz = torch.zeros((M, 1, 1, Cm), dtype=torch.bfloat16, device="cuda")
        x = torch.zeros((M, 1, 1, Cin), dtype=torch.bfloat16, device="cuda")
        w_kn = torch.zeros((Cin, Cm), dtype=torch.bfloat16, device="cuda")
        assert K.pwconv_bwd_fused(dz, dz, torch.zeros((3, Cm), device="cuda"), x, w_kn, None) is None
