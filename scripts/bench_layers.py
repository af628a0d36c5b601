"""Per-layer kernel timings at the BASELINE shapes (EfficientNet-B0, batch 256, 224 px, bf16).

    python scripts/bench_layers.py [dw|pw|all] [batch]

Prints, for every MBConv block, the duration and the algorithmic HBM rate (bytes every
tensor must move once / time) of the depthwise and pointwise kernels, forward and backward.
Used to steer kernel optimisation; numbers quoted in DESIGN.md come from here.
"""

from __future__ import annotations

import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

from deepfakedetection_amd import kernels as K  # noqa: E402
from deepfakedetection_amd._lib import ACT_SILU  # noqa: E402
from deepfakedetection_amd.arch import efficientnet_plan  # noqa: E402

DT = torch.bfloat16
ES = 2


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def state(C):
    st = torch.zeros((4, C), device="cuda")
    st[0] = 1.0
    st[3] = 1.0
    return st


def main() -> None:
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    plan = efficientnet_plan("b0", "timm")
    H = 112
    tot = {}
    print(f"{'blk':>3} {'op':<14} {'shape':<28} {'us':>9} {'GB/s':>8} {'TF/s':>7}")
    for b in plan.blocks:
        Ho = b.dw.out_size(H)
        Cm = b.cmid
        rows = []
        if what in ("dw", "all"):
            x = torch.randn((N, H, H, Cm), device="cuda").to(DT)
            w = torch.randn((Cm, 1, b.dw.kernel, b.dw.kernel), device="cuda") * 0.2
            st = state(Cm)
            coef = torch.zeros((3, Cm), device="cuda"); coef[0] = 1.0
            dz = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
            y = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
            g = b.dw
            bin_, bout = x.numel() * ES, dz.numel() * ES
            t = timeit(lambda: K.dwconv_fwd(x, st, ACT_SILU, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, True))
            rows.append(("dw_fwd", f"{H}->{Ho} C{Cm} k{g.kernel}s{g.stride}", t, bin_ + bout, 0))
            t = timeit(lambda: K.dwconv_bwd_data(dz, y, coef, w, x, st, ACT_SILU, tuple(x.shape), g.kernel, g.stride, g.pad_lead, g.pad_lead))
            rows.append(("dw_bwd_data", "", t, 2 * bout + 2 * bin_, 0))
            t = timeit(lambda: K.dwconv_bwd_weight(dz, y, coef, x, st, ACT_SILU, g.kernel, g.stride, g.pad_lead, g.pad_lead))
            rows.append(("dw_bwd_weight", "", t, 2 * bout + bin_, 0))
            del x, dz, y
        if what in ("pw", "all"):
            M_in, M_out = N * H * H, N * Ho * Ho
            if b.expand:
                a = torch.randn((N, H, H, b.cin), device="cuda").to(DT)
                w = torch.randn((Cm, b.cin), device="cuda") * 0.1
                w_nk, w_kn = K.prep_weights(w, DT)
                t = timeit(lambda: K.pwconv(a, None, w_nk, None, True))
                fl = 2.0 * M_in * b.cin * Cm
                rows.append(("pw_expand", f"M{M_in} {b.cin}->{Cm}", t, M_in * (b.cin + Cm) * ES, fl))
                dz1 = torch.randn((N, H, H, Cm), device="cuda").to(DT)
                y1 = torch.randn((N, H, H, Cm), device="cuda").to(DT)
                coef = torch.zeros((3, Cm), device="cuda"); coef[0] = 1.0
                pro = K.pro_affine2(y1, coef)
                t = timeit(lambda: K.pwconv(dz1, pro, w_kn, a if b.skip else None, False))
                rows.append(("pw_exp_dgrad", "", t, M_in * (2 * Cm + b.cin * (2 if b.skip else 1)) * ES, fl))
                t = timeit(lambda: K.pwconv_wgrad(dz1, pro, a, None))
                rows.append(("pw_exp_wgrad", "", t, M_in * (2 * Cm + b.cin) * ES, fl))
                del a, dz1, y1
            y2 = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
            w = torch.randn((b.cout, Cm), device="cuda") * 0.1
            w_nk, w_kn = K.prep_weights(w, DT)
            st = state(Cm)
            gate = torch.rand((N, Cm), device="cuda")
            pro = K.pro_bn_act_gate(st, ACT_SILU, gate, Ho * Ho)
            if os.environ.get("NOPRO") == "1":          # what-if: activated and gated tensor materialised
                pro = None
            elif os.environ.get("NOPRO") == "2":        # what-if: activated tensor materialised, gate applied here
                pro = K.pro_bn_act_gate(st, 0, gate, Ho * Ho)
            t = timeit(lambda: K.pwconv(y2, pro, w_nk, None, True))
            fl = 2.0 * M_out * b.cout * Cm
            rows.append(("pw_project", f"M{M_out} {Cm}->{b.cout}", t, M_out * (Cm + b.cout) * ES, fl))
            gb = torch.randn((N, Ho, Ho, b.cout), device="cuda").to(DT)
            y3 = torch.randn((N, Ho, Ho, b.cout), device="cuda").to(DT)
            coef3 = torch.zeros((3, b.cout), device="cuda"); coef3[0] = 1.0
            pro3 = K.pro_affine2(y3, coef3)
            t = timeit(lambda: K.pwconv(gb, pro3, w_kn, None, False))
            rows.append(("pw_proj_dgrad", "", t, M_out * (2 * b.cout + Cm) * ES, fl))
            t = timeit(lambda: K.pwconv_wgrad(gb, pro3, y2, pro))
            rows.append(("pw_proj_wgrad", "", t, M_out * (2 * b.cout + Cm) * ES, fl))
            del y2, gb, y3
        for name, shape, t, nbytes, fl in rows:
            print(f"{b.index:>3} {name:<14} {shape:<28} {t * 1e6:9.1f} {nbytes / t / 1e9:8.0f} {fl / t / 1e12:7.1f}")
            a = tot.setdefault(name, [0.0, 0.0])
            a[0] += t
            a[1] += nbytes
        H = Ho
        torch.cuda.empty_cache()
    print("---- totals")
    for name, (t, nb) in tot.items():
        print(f"    {name:<14} {t * 1e3:8.3f} ms  {nb / t / 1e9:8.0f} GB/s")


if __name__ == "__main__":
    main()
