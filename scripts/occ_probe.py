import sys, torch
sys.path.insert(0, "/root/repo")
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU
from deepfakedetection_amd.arch import efficientnet_plan
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
plan = efficientnet_plan("b0", "timm"); H = 112; N = 256
for b in plan.blocks:
    Ho, Cm, g = b.dw.out_size(H), b.cmid, b.dw
    x = torch.randn((N, H, H, Cm), device="cuda").to(torch.bfloat16)
    w = torch.randn((Cm, 1, g.kernel, g.kernel), device="cuda") * 0.2
    st = torch.zeros((4, Cm), device="cuda"); st[0] = 1; st[3] = 1
    t1 = timeit(lambda: K.dwconv_fwd(x, st, ACT_SILU, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, True))
    t0 = timeit(lambda: K.dwconv_fwd(x, st, ACT_SILU, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, False))
    print(f"{b.index:2d} k{g.kernel}s{g.stride} C{Cm:4d} {H:3d}->{Ho:3d}  stats {t1:7.1f} us   no-stats {t0:7.1f} us   ratio {t0/t1:.2f}")
    H = Ho
