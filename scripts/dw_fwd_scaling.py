"""Does the vector-unit depthwise forward get faster per image with more work items per workgroup (phases of the co-resident workgroups
drifting apart), and what does the BN + SiLU prologue cost?   python scripts/dw_fwd_scaling.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU, ACT_NONE, load

def timeit(fn, reps=10):
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

L = load()
L.dfd_tune(0, 0)
DT = torch.bfloat16
for (H, C, k, s) in [(14, 480, 3, 1), (14, 672, 5, 1), (7, 1152, 5, 1), (28, 240, 5, 1), (56, 144, 3, 1)]:
    Ho = H // s
    pad = (k - 1) // 2
    for N in (128, 256, 512, 1024):
        if N * H * H * C > 3e8: continue
        x = torch.randn((N, H, H, C), device="cuda").to(DT)
        w = torch.randn((C, 1, k, k), device="cuda") * 0.2
        sta = st(C)
        t_pro = timeit(lambda: K.dwconv_fwd(x, sta, ACT_SILU, w, k, s, pad, pad, Ho, Ho, True))
        t_raw = timeit(lambda: K.dwconv_fwd(x, None, ACT_NONE, w, k, s, pad, pad, Ho, Ho, True))
        nb = 2 * x.numel() * 2
        print(f"{H}x{H} C{C} k{k} N{N}: pro {t_pro:7.1f} us ({t_pro / N * 256:6.1f} per 256, {nb / t_pro / 1e3:5.0f} GB/s)   raw {t_raw:7.1f} us ({t_raw / N * 256:6.1f} per 256, {nb / t_raw / 1e3:5.0f} GB/s)", flush=True)
        del x
