"""A/B of the 1x1 NT kernels on the MID-SIZE layers of EfficientNet-B0 at batch 256 (14 x 14 and 7 x 7 maps): the register-staged
tile kernel k_pw_nt (dfd_tune key 4 = 0) against the LDS-DMA ring kernel k_pw_ntd (key 4 = 1, ring depth by key 5).

    python scripts/pw_mid_shapes.py [ns ...]        (default ring depths: 0 = automatic)

Per layer and variant (expand forward = plain operand + statistics, project forward = BN + SiLU + gate prologue + statistics,
project data gradient = plain, expand data gradient = BN-backward map of two tensors + residual): microseconds, and the rate
on the bytes every tensor moves once.  Also checks that both kernels return the same bits."""
from __future__ import annotations

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from deepfakedetection_amd import kernels as K  # noqa: E402
from deepfakedetection_amd._lib import ACT_SILU  # noqa: E402

DT = torch.bfloat16
L = K._L()


def timeit(fn, reps=20):
    """Microseconds per call with the calls REPLAYED from a hipGraph: from Python a call costs 13-19 us of host time (allocation,
    ctypes), which hides every kernel shorter than that."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


def main() -> None:
    depths = [int(v) for v in sys.argv[1:]] or [0]
    # (rows, HW, Cin, Cmid, Cout) of blocks 5..15 + head (distinct shapes)
    shapes = [(50176, 196, 80, 480, 80), (50176, 196, 80, 480, 112), (50176, 196, 112, 672, 112), (12544, 49, 112, 672, 192),
              (12544, 49, 192, 1152, 192), (12544, 49, 192, 1152, 320), (12544, 49, 320, 1280, 0)]
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    print(f"{'variant':<12} {'shape':<26} {'k_pw_nt us':>10} " + " ".join(f"{'ntd ns=' + str(d):>12}" for d in depths) + f" {'GB/s best':>10}  same bits")
    tot = {"nt": 0.0, **{d: 0.0 for d in depths}}
    for (M, HW, Cin, Cm, Co) in shapes:
        N = M // HW
        variants = []
        x = rnd(M, Cin).to(DT)
        wexp = (rnd(Cm, Cin) * Cin ** -0.5).to(DT)
        variants.append(("expand", x, None, wexp, None, True, M * (Cin + Cm) * 2))
        if Co:
            y2 = rnd(M, Cm).to(DT)
            st = torch.stack([0.5 + torch.rand(Cm, device="cuda", generator=g), rnd(Cm) * 0.1, rnd(Cm), 1 + torch.rand(Cm, device="cuda", generator=g)])
            gate = torch.rand(N, Cm, device="cuda", generator=g)
            wproj = (rnd(Co, Cm) * Cm ** -0.5).to(DT)
            variants.append(("project", y2, K.pro_bn_act_gate(st, ACT_SILU, gate, HW), wproj, None, True, M * (Cm + Co) * 2))
            gm = rnd(M, Co).to(DT)
            variants.append(("proj_dgrad", gm, None, wproj.t().contiguous(), None, False, M * (Cm + Co) * 2))
        dz = rnd(M, Cm).to(DT)
        y1 = rnd(M, Cm).to(DT)
        coef = torch.stack([0.5 + torch.rand(Cm, device="cuda", generator=g), rnd(Cm) * 0.1, rnd(Cm) * 0.05])
        res = rnd(M, Cin).to(DT)
        variants.append(("exp_dgrad", dz, K.pro_affine2(y1, coef), wexp.t().contiguous(), res, False, M * (2 * Cm + 2 * Cin) * 2))
        for name, a, pro, w, r, stats, nbytes in variants:
            fn = lambda: K.pwconv(a, pro, w, r, stats=stats)
            L.dfd_tune(4, 0)
            ref = fn()
            t_nt = timeit(fn)
            tot["nt"] += t_nt
            cells, best, same = [], t_nt, True
            for d in depths:
                L.dfd_tune(4, 1)
                L.dfd_tune(5, d)
                if not L.dfd_pw_ntd_plan(a.shape[0], a.shape[1], w.shape[0]):
                    cells.append(f"{'-':>12}")
                    tot[d] += t_nt
                    continue
                got = fn()
                same = same and torch.equal(got[0], ref[0])
                td = timeit(fn)
                tot[d] += td
                best = min(best, td)
                cells.append(f"{td:12.1f}")
            print(f"{name:<12} {f'{M}x{a.shape[1]}->{w.shape[0]}':<26} {t_nt:10.1f} " + " ".join(cells) + f" {nbytes / best / 1e3:10.0f}  {same}")
    L.dfd_tune(4, 1)
    L.dfd_tune(5, 0)
    print("totals (us): k_pw_nt", round(tot["nt"], 1), {d: round(tot[d], 1) for d in depths})


if __name__ == "__main__":
    main()
