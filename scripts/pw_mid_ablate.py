"""Timing-only ablations of the LDS-DMA ring kernel k_pw_ntd (dfd_tune key 3, bits 8..12: no prologue arithmetic, no MFMAs, no
weight DMA, no activation DMA, no epilogue) on three EfficientNet-B0 shapes.  Results are WRONG while a bit is set."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from deepfakedetection_amd import kernels as K  # noqa: E402
from deepfakedetection_amd._lib import ACT_SILU  # noqa: E402
from pw_mid_shapes import timeit  # noqa: E402

L = K._L()
DT = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
for k, v in (kv.split("=") for kv in sys.argv[1:]):
    L.dfd_tune(int(k), int(v))
cases = []
for (M, HW, Kd, No, mode) in [(12544, 49, 1152, 192, 2), (50176, 196, 480, 80, 2), (50176, 196, 80, 480, 0), (50176, 196, 480, 80, 3), (12544, 49, 1152, 192, 3)]:
    a = rnd(M, Kd).to(DT)
    w = (rnd(No, Kd) * Kd ** -0.5).to(DT)
    if mode == 2:
        st = torch.stack([0.5 + torch.rand(Kd, device="cuda", generator=g), rnd(Kd) * 0.1, rnd(Kd), 1 + torch.rand(Kd, device="cuda", generator=g)])
        gate = torch.rand(M // HW, Kd, device="cuda", generator=g)
        pro, res, stats = K.pro_bn_act_gate(st, ACT_SILU, gate, HW), None, True
        keep = (st, gate)
    elif mode == 3:
        a2 = rnd(M, Kd).to(DT)
        coef = torch.stack([0.5 + torch.rand(Kd, device="cuda", generator=g), rnd(Kd) * 0.1, rnd(Kd) * 0.05])
        pro, res, stats = K.pro_affine2(a2, coef), rnd(M, No).to(DT), False
        keep = (a2, coef)
    else:
        pro, res, stats, keep = None, None, True, None
    cases.append((f"{M}x{Kd}->{No} mode {mode}", a, pro, w, res, stats, keep))
masks = [0, 1, 2, 3, 4, 8, 12, 16, 15, 31]
print(f"{'shape':<28}" + "".join(f"{m:>8}" for m in masks) + "   (bits: 1 no prologue, 2 no MFMA, 4 no W DMA, 8 no A DMA, 16 no epilogue)")
for name, a, pro, w, res, stats, keep in cases:
    row = []
    for m in masks:
        L.dfd_tune(3, m << 8)
        row.append(timeit(lambda: K.pwconv(a, pro, w, res, stats=stats)))
    L.dfd_tune(3, 0)
    print(f"{name:<28}" + "".join(f"{v:8.1f}" for v in row))
