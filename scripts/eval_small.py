"""Inference forward of EfficientNet-B0 over batch sizes, eager and as a replayed hipGraph: python eval_small.py [f32|bf16]."""
import contextlib, sys, time, torch
sys.path.insert(0, "/root/repo")
from deepfakedetection_amd.efficientnet import HipEfficientNet
amp = len(sys.argv) > 1 and sys.argv[1] == "bf16"
cast = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if amp else contextlib.nullcontext
torch.manual_seed(0)
m = HipEfficientNet("b0", "timm", 2).cuda().eval()
for B in (16, 32, 64, 128, 256):
    x = torch.randn(B, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
    with torch.inference_mode(), cast():
        for _ in range(3): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        g = torch.cuda.CUDAGraph()
        sx = x.clone()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = m(sx)
        g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / 20
    print(f"{'bf16' if amp else 'f32 '} batch {B:4d}: eager {dt*1e3:7.3f} ms ({B/dt:8.0f} img/s)   graph {dg*1e3:7.3f} ms ({B/dg:8.0f} img/s)", flush=True)
