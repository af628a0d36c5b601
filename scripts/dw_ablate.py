"""Timing-only ablations of the matrix-core depthwise forward (dfd_tune key 3) on selected B0 layers."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU, DwShape, load
from deepfakedetection_amd.arch import efficientnet_plan

N = 256
L = load()
L.dfd_tune(0, 9)
blocks = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 4, 9, 12]
for kv in sys.argv[2:]:
    k, v = kv.split("="); L.dfd_tune(int(k), int(v))
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
plan = efficientnet_plan("b0", "timm")
H = 112
for b in plan.blocks:
    g = b.dw; Ho = g.out_size(H); C = b.cmid
    if b.index in blocks:
        x = torch.randn((N, H, H, C), device="cuda").to(DT)
        w = torch.randn((C, 1, g.kernel, g.kernel), device="cuda") * 0.2
        s = torch.zeros((4, C), device="cuda"); s[0] = 1; s[3] = 1
        shp = DwShape(N, H, H, C, Ho, Ho, g.kernel, g.stride, g.pad_lead, g.pad_lead)
        out = (ctypes.c_int * 12)()
        L.dfd_dw_mm_plan(ctypes.byref(shp), 1, out)
        fn = lambda: K.dwconv_fwd(x, s, ACT_SILU, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, True)
        fn0 = lambda: K.dwconv_fwd(x, None, 0, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, True)
        res = []
        for dbg in (0, 1, 2, 4, 8, 3, 7, 15, 14):
            L.dfd_tune(3, dbg); res.append(f"{dbg}:{timeit(fn):.0f}")
        L.dfd_tune(3, 0)
        nb = (x.numel() + N * Ho * Ho * C) * 2
        print(f"blk{b.index} {H}->{Ho} C{C} k{g.kernel}s{g.stride} plan NI,TH,TW,R,IH,IW,P,plane,nwork,whole,ldsin,ldsout={list(out)}  ideal@5TB/s {nb / 5e6:.0f}us  nopro-entry {timeit(fn0):.0f}  dbg " + " ".join(res))
        del x
    H = Ho
