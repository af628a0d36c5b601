"""A/B of the depthwise kernels per EfficientNet-B0 layer: vector-unit kernels (dfd_tune DW_MFMA = 0) against the matrix-core
form (dfd_dwmm*.hip).   python scripts/dw_ab.py [batch] [lds_kb] [grid] [valu_grid] [valu_grid_min]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU, load
from deepfakedetection_amd.arch import efficientnet_plan

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = load()
if len(sys.argv) > 2: L.dfd_tune(1, int(sys.argv[2]))
if len(sys.argv) > 3: L.dfd_tune(2, int(sys.argv[3]))
if len(sys.argv) > 4:                     # workgroups the vector-unit kernels aim for (forward, data gradient, weight gradient)
    for key in (8, 9, 10): L.dfd_tune(key, int(sys.argv[4]))
if len(sys.argv) > 5: L.dfd_tune(11, int(sys.argv[5]))
DT = torch.bfloat16

def timeit(fn, reps=10):
    """device time per call: `reps` calls captured into one hipGraph (no host launch gaps), replayed 3 times"""
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3

def st(C):
    s = torch.zeros((4, C), device="cuda"); s[0] = 1 + 0.1 * torch.randn(C, device="cuda"); s[1] = 0.1 * torch.randn(C, device="cuda"); s[3] = 1; return s

plan = efficientnet_plan("b0", "timm")
H = 112
tot = {}
print(f"{'blk':>3} {'shape':<24} {'op':<10} {'valu us':>9} {'mfma us':>9} {'ratio':>6} {'GB/s':>7}  maxdiff")
for b in plan.blocks:
    g = b.dw; Ho = g.out_size(H); C = b.cmid
    x = torch.randn((N, H, H, C), device="cuda").to(DT)
    w = torch.randn((C, 1, g.kernel, g.kernel), device="cuda") * 0.2
    s = st(C)
    dz = torch.randn((N, Ho, Ho, C), device="cuda").to(DT); y = torch.randn((N, Ho, Ho, C), device="cuda").to(DT)
    coef = torch.zeros((3, C), device="cuda"); coef[0] = 1.0; coef[1] = 0.05
    ops = {
        "fwd": (lambda: K.dwconv_fwd(x, s, ACT_SILU, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, True), (x.numel() + dz.numel()) * 2, 1),
        "bwd_data": (lambda: K.dwconv_bwd_data(dz, y, coef, w, x, s, ACT_SILU, tuple(x.shape), g.kernel, g.stride, g.pad_lead, g.pad_lead), (2 * dz.numel() + 2 * x.numel()) * 2, 2),
        "bwd_weight": (lambda: K.dwconv_bwd_weight(dz, y, coef, x, s, ACT_SILU, g.kernel, g.stride, g.pad_lead, g.pad_lead), (2 * dz.numel() + x.numel()) * 2, 4),
    }
    for name, (fn, nbytes, bit) in ops.items():
        L.dfd_tune(0, 0)
        ref = fn(); ref = ref[0] if isinstance(ref, tuple) else ref
        t0 = timeit(fn)
        L.dfd_tune(0, 9)
        got = fn(); got = got[0] if isinstance(got, tuple) else got
        t1 = timeit(fn)
        d = (got.float() - ref.float()).abs().max().item() / max(ref.float().abs().max().item(), 1e-9)
        print(f"{b.index:>3} {f'{H}->{Ho} C{C} k{g.kernel}s{g.stride}':<24} {name:<10} {t0:9.1f} {t1:9.1f} {t0 / t1:6.2f} {nbytes / t1 / 1e3:7.0f}  {d:.2e}")
        a = tot.setdefault(name, [0.0, 0.0]); a[0] += t0; a[1] += t1
    H = Ho
    del x, dz, y
    torch.cuda.empty_cache()
for name, (a, b_) in tot.items():
    print(f"total {name:<10} valu {a / 1e3:.3f} ms   mfma {b_ / 1e3:.3f} ms")
