"""A/B timings of kernel options per EfficientNet-B0 layer (batch 256): 1x1 forward with / without BN statistics."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU
from deepfakedetection_amd.arch import efficientnet_plan


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


plan = efficientnet_plan("b0", "timm")
H, N, DT = 112, 256, torch.bfloat16
tot = [0.0] * 4
for b in plan.blocks:
    Ho, Cm = b.dw.out_size(H), b.cmid
    row = f"{b.index:2d}"
    if b.expand:
        a = torch.randn((N, H, H, b.cin), device="cuda").to(DT)
        w_nk, _ = K.prep_weights(torch.randn((Cm, b.cin), device="cuda") * 0.1, DT)
        t1, t0 = timeit(lambda: K.pwconv(a, None, w_nk, None, True)), timeit(lambda: K.pwconv(a, None, w_nk, None, False))
        row += f"  expand M{N * H * H:8d} {b.cin:4d}->{Cm:4d} stats {t1:6.1f} no-stats {t0:6.1f}"
        tot[0] += t1; tot[1] += t0
        del a
    y2 = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
    w_nk, _ = K.prep_weights(torch.randn((b.cout, Cm), device="cuda") * 0.1, DT)
    st = torch.zeros((4, Cm), device="cuda"); st[0] = 1; st[3] = 1
    pro = K.pro_bn_act_gate(st, ACT_SILU, torch.rand((N, Cm), device="cuda"), Ho * Ho)
    t1, t0 = timeit(lambda: K.pwconv(y2, pro, w_nk, None, True)), timeit(lambda: K.pwconv(y2, pro, w_nk, None, False))
    row += f"  | project {Cm:4d}->{b.cout:4d} stats {t1:6.1f} no-stats {t0:6.1f}"
    tot[2] += t1; tot[3] += t0
    print(row)
    H = Ho
    torch.cuda.empty_cache()
print(f"totals: expand {tot[0]:.0f} / {tot[1]:.0f} us   project {tot[2]:.0f} / {tot[3]:.0f} us")
