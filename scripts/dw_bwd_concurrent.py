"""How much would co-scheduling a layer's depthwise data gradient and weight gradient buy?  Both read (dz, y, x); today they are two
launches back to back.  Per EfficientNet-B0 layer (batch 256): each kernel alone, the pair back to back on one stream, and the pair
on TWO streams (concurrent: what a dual-role launch could reach at best through shared L2 lines and filled stalls)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from deepfakedetection_amd import kernels as K  # noqa: E402
from deepfakedetection_amd._lib import ACT_SILU  # noqa: E402
from deepfakedetection_amd.arch import efficientnet_plan  # noqa: E402

DT = torch.bfloat16
N = 256
plan = efficientnet_plan("b0", "timm")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t_ms(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


H = 112
print(f"{'blk':>3} {'shape':<26} {'data us':>9} {'weight us':>9} {'serial':>9} {'2 streams':>10} {'gain':>6}")
tot = [0.0, 0.0]
for b in plan.blocks:
    Ho = b.dw.out_size(H)
    Cm, g = b.cmid, b.dw
    if b.expand:
        x = torch.randn((N, H, H, Cm), device="cuda").to(DT)
        w = torch.randn((Cm, 1, g.kernel, g.kernel), device="cuda") * 0.2
        st = torch.zeros((4, Cm), device="cuda"); st[0] = 1.0; st[3] = 1.0
        coef = torch.zeros((3, Cm), device="cuda"); coef[0] = 1.0
        dz = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
        y = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
        fd = lambda: K.dwconv_bwd_data(dz, y, coef, w, x, st, ACT_SILU, tuple(x.shape), g.kernel, g.stride, g.pad_lead, g.pad_lead)
        fw = lambda: K.dwconv_bwd_weight(dz, y, coef, x, st, ACT_SILU, g.kernel, g.stride, g.pad_lead, g.pad_lead)

        def both():
            fd(); fw()

        def conc():
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(s1):
                s1.wait_event(ev); fd(); e1 = torch.cuda.Event(); e1.record()
            with torch.cuda.stream(s2):
                s2.wait_event(ev); fw(); e2 = torch.cuda.Event(); e2.record()
            torch.cuda.current_stream().wait_event(e1)
            torch.cuda.current_stream().wait_event(e2)

        # the scratch buffers are keyed by stream: warm both up
        td, tw, ts, tc = t_ms(fd), t_ms(fw), t_ms(both), t_ms(conc)
        tot[0] += ts; tot[1] += tc
        print(f"{b.index:>3} {f'{H}->{Ho} C{Cm} k{g.kernel}s{g.stride}':<26} {td:9.1f} {tw:9.1f} {ts:9.1f} {tc:10.1f} {1 - tc / ts:6.1%}")
        del x, dz, y
    H = Ho
    torch.cuda.empty_cache()
print(f"totals: serial {tot[0] / 1e3:.3f} ms, two streams {tot[1] / 1e3:.3f} ms")
