"""Bisect a hipGraph capture crash of the FasterViT step: python scripts/debug/capture_fv.py <mode>
mode: fwd (capture forward only) | fwdbwd (forward + backward) | step (GraphedTrainStep)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faulthandler; faulthandler.enable()
import torch
from deepfakedetection_amd.fastervit import build_fastervit
from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

mode = sys.argv[1]
torch.manual_seed(0)
model = build_fastervit("faster_vit_0_224", 2).cuda().train()
opt = HipAdamW(model.parameters(), lr=1e-3)
crit = HipCrossEntropyLoss(0.1)
x = torch.randn(4, 3, 224, 224).cuda(); y = torch.randint(0, 2, (4,)).cuda()

def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return crit(model(x), y)

def fwdbwd():
    opt.zero_grad(set_to_none=True)
    loss = fwd(); loss.backward(); return loss.detach()

for _ in range(2):
    fwdbwd(); opt.step()
torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
if mode == "fwd":
    with torch.no_grad():
        fwd()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = fwd()
elif mode == "fwdbwd":
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = fwdbwd()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", float(out), flush=True)
