"""Per-shape timing of the three depthwise kernels (batch 256, bf16) with the fusions the models use: time, algorithmic
bytes and TB/s.  python scripts/dw_shapes.py [ef|b0]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU, ACT_GELU


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


which = sys.argv[1] if len(sys.argv) > 1 else "ef"
N, DT = 256, torch.bfloat16
if which == "ef":
    shapes = [(128, 56, 3, 1, 3), (192, 28, 3, 1, 3), (480, 14, 3, 1, 9), (896, 7, 3, 1, 6)]
    act = ACT_GELU
else:
    from deepfakedetection_amd.arch import efficientnet_plan
    shapes, H = [], 112
    for b in efficientnet_plan("b0", "timm").blocks:
        shapes.append((b.cmid, H, b.dw.kernel, b.dw.stride, 1))
        H = b.dw.out_size(H)
    act = ACT_SILU
tot = [0.0, 0.0, 0.0]
for C, H, k, s, count in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn((N, H, H, C), device="cuda").to(DT)
    w = torch.randn((C, 1, k, k), device="cuda") * 0.2
    st = torch.zeros((4, C), device="cuda"); st[0] = 1; st[3] = 1
    cf = torch.zeros((3, C), device="cuda"); cf[0] = 1
    dz = torch.randn((N, Ho, Ho, C), device="cuda").to(DT)
    y = torch.randn((N, Ho, Ho, C), device="cuda").to(DT)
    tf = timeit(lambda: K.dwconv_fwd(x, st, act, w, k, s, p, p, Ho, Ho, stats=True))
    td = timeit(lambda: K.dwconv_bwd_data(dz, y, cf, w, x, st, act, tuple(x.shape), k, s, p, p))
    tw = timeit(lambda: K.dwconv_bwd_weight(dz, y, cf, x, st, act, k, s, p, p))
    tf0 = timeit(lambda: K.dwconv_fwd(x, None, 0, w, k, s, p, p, Ho, Ho, stats=True))
    tw0 = timeit(lambda: K.dwconv_bwd_weight(dz, y, cf, x, None, 0, k, s, p, p))
    ta = timeit(lambda: K.bn_act_apply(x, st, act))
    td0 = timeit(lambda: K.dwconv_bwd_data(dz, None, None, w, x, st, act, tuple(x.shape), k, s, p, p))
    tw1 = timeit(lambda: K.dwconv_bwd_weight(dz, None, None, x, None, 0, k, s, p, p))
    tm = timeit(lambda: K.affine2_apply(dz, y, cf))
    bi, bo = x.numel() * 2, dz.numel() * 2
    bf, bd, bw = bi + bo, 2 * bo + 2 * bi, 2 * bo + bi
    print(f"C{C:5d} {H:3d}x{H:<3d} k{k}s{s} x{count}  fwd {tf:6.1f} us {bf / tf / 1e6:5.2f} TB/s | data {td:6.1f} us {bd / td / 1e6:5.2f} TB/s | "
          f"weight {tw:6.1f} us {bw / tw / 1e6:5.2f} TB/s || no prologue: fwd {tf0:6.1f} weight {tw0:6.1f}  apply pass {ta:6.1f} || no BN map: data {td0:6.1f} weight(no pro, no map) {tw1:6.1f} map pass {tm:6.1f}", flush=True)
    for i, t in enumerate((tf, td, tw)):
        tot[i] += t * count
    del x, dz, y
    torch.cuda.empty_cache()
print("totals per step (us): fwd %.0f data %.0f weight %.0f" % tuple(tot))
