"""How does the LDS-DMA ring kernel's time step with the number of 64-row tiles around the co-resident workgroup count?
(14x14 layers of EfficientNet-B0: 50176 rows = 784 tiles against 768 slots at three workgroups per CU.)   python scripts/pw_mid_tail.py"""
from __future__ import annotations

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from deepfakedetection_amd import kernels as K  # noqa: E402
from deepfakedetection_amd._lib import ACT_SILU  # noqa: E402
from scripts.pw_mid_shapes import timeit  # noqa: E402

DT = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
for (HW, Cin, Cm, Co) in [(196, 80, 480, 80), (196, 112, 672, 112), (49, 192, 1152, 192)]:
    for N in ([224, 240, 244, 248, 252, 256, 260, 272, 320, 384] if HW == 196 else [256, 320, 334, 336, 400, 512, 668, 672, 1002, 1024]):
        M = N * HW
        y2 = rnd(M, Cm).to(DT)
        st = torch.stack([0.5 + torch.rand(Cm, device="cuda", generator=g), rnd(Cm) * 0.1, rnd(Cm), 1 + torch.rand(Cm, device="cuda", generator=g)])
        gate = torch.rand(N, Cm, device="cuda", generator=g)
        wproj = (rnd(Co, Cm) * Cm ** -0.5).to(DT)
        pro = K.pro_bn_act_gate(st, ACT_SILU, gate, HW)
        t_proj = timeit(lambda: K.pwconv(y2, pro, wproj, None, stats=True))
        dz, y1 = rnd(M, Cm).to(DT), rnd(M, Cm).to(DT)
        coef = torch.stack([0.5 + torch.rand(Cm, device="cuda", generator=g), rnd(Cm) * 0.1, rnd(Cm) * 0.05])
        wexp_t = (rnd(Cin, Cm) * Cm ** -0.5).to(DT)
        res = rnd(M, Cin).to(DT)
        pa = K.pro_affine2(y1, coef)
        t_dg = timeit(lambda: K.pwconv(dz, pa, wexp_t, res, stats=False))
        tiles = (M + 63) // 64
        print(f"HW{HW} {Cm}->{Co}  N={N:5d} tiles={tiles:5d} ({tiles / 256:5.2f} per CU)  project {t_proj:6.1f} us ({t_proj / tiles * 1e3:6.1f} ns/tile)   exp_dgrad {t_dg:6.1f} us ({t_dg / tiles * 1e3:6.1f} ns/tile)", flush=True)
        del y2, dz, y1, res
