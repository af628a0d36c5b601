import sys, torch
sys.path.insert(0, "/root/repo")
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU, ACT_NONE
from deepfakedetection_amd.arch import efficientnet_plan
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
plan = efficientnet_plan("b0", "timm"); H = 112; N = 256; DT = torch.bfloat16
tot = [0.0]*6
for b in plan.blocks:
    Ho, Cm, g = b.dw.out_size(H), b.cmid, b.dw
    x = torch.randn((N, H, H, Cm), device="cuda").to(DT)
    w = torch.randn((Cm, 1, g.kernel, g.kernel), device="cuda") * 0.2
    st = torch.zeros((4, Cm), device="cuda"); st[0] = 1; st[3] = 1
    cf = torch.zeros((3, Cm), device="cuda"); cf[0] = 1
    dz = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT); y = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
    a = (g.kernel, g.stride, g.pad_lead, g.pad_lead)
    t = [timeit(lambda: K.dwconv_bwd_data(dz, y, cf, w, x, st, ACT_SILU, tuple(x.shape), *a)),
         timeit(lambda: K.dwconv_bwd_data(dz, y, cf, w, None, None, ACT_NONE, tuple(x.shape), *a)),
         timeit(lambda: K.dwconv_bwd_data(dz, None, None, w, None, None, ACT_NONE, tuple(x.shape), *a)),
         timeit(lambda: K.dwconv_bwd_weight(dz, y, cf, x, st, ACT_SILU, *a)),
         timeit(lambda: K.dwconv_bwd_weight(dz, None, None, x, st, ACT_SILU, *a)),
         timeit(lambda: K.dwconv_bwd_weight(dz, None, None, x, None, ACT_NONE, *a))]
    for i, v in enumerate(t): tot[i] += v
    print(f"{b.index:2d} k{g.kernel}s{g.stride} C{Cm:4d} {H:3d}  data full {t[0]:6.1f}  no-epi {t[1]:6.1f}  plain {t[2]:6.1f} | weight full {t[3]:6.1f}  no-coef {t[4]:6.1f}  plain {t[5]:6.1f}")
    H = Ho
print("totals", " ".join(f"{v:.0f}" for v in tot))
