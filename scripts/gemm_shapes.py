"""FasterViT-0 level-2/3 linear shapes (batch 256): the engine's NT GEMM against torch.mm (hipBLASLt / rocBLAS) as a yardstick."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


DT = torch.bfloat16
M2, M3 = 256 * 212, 256 * 49 + 256 * 0
shapes = [(M2, 256, 768), (M2, 256, 256), (M2, 256, 1024), (M2, 1024, 256), (M2, 768, 256),
          (256 * 49, 512, 1536), (256 * 49, 512, 512), (256 * 49, 512, 2048), (256 * 49, 2048, 512)]
for M, Kd, N in shapes:
    a = torch.randn((M, 1, 1, Kd), device="cuda").to(DT)
    w = torch.randn((N, Kd), device="cuda") * 0.05
    w_nk, w_kn = K.prep_weights(w, DT, True, True)
    t_mine = timeit(lambda: K.pwconv(a, None, w_nk, None, stats=False))
    a2 = a.view(M, Kd)
    wt = w.to(DT).t().contiguous()                      # [K, N]
    wn = w.to(DT)                                        # [N, K]
    t_mm = timeit(lambda: torch.mm(a2, wt))
    t_lin = timeit(lambda: torch.nn.functional.linear(a2, wn))
    # weight gradient: dW[N, K] = g^T a
    g = torch.randn((M, 1, 1, N), device="cuda").to(DT)
    t_wg = timeit(lambda: K.pwconv_wgrad(g, None, a, None))
    g2 = g.view(M, N)
    t_wg_mm = timeit(lambda: torch.mm(g2.t(), a2))
    fl = 2.0 * M * Kd * N
    print(f"M{M:6d} K{Kd:5d} N{N:5d}  NT mine {t_mine:6.1f} us ({fl / t_mine / 1e6:6.0f} TF)  mm {t_mm:6.1f}  linear {t_lin:6.1f} ({fl / min(t_mm, t_lin) / 1e6:6.0f} TF) |"
          f" wgrad mine {t_wg:6.1f}  mm {t_wg_mm:6.1f}", flush=True)
