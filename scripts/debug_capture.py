"""Debug helper: which part of the train step breaks hipGraph stream capture?"""
import sys, traceback
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd.efficientnet import HipEfficientNet
from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

dev = torch.device("cuda:0")
model = HipEfficientNet("b0", "timm", 2).to(dev).train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
crit = HipCrossEntropyLoss(0.1)
NB, SZ = int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(NB, 3, SZ, SZ, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.zeros(NB, dtype=torch.int64, device=dev)

def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return crit(model(x), y)

def stage(name, fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(name, "capture OK", flush=True)
    except Exception as e:
        print(name, "capture FAILED:", type(e).__name__, str(e).splitlines()[0], flush=True)
        sys.exit(1)

def f_fwd_nograd():
    with torch.no_grad():
        fwd()
def f_fwd_bwd():
    opt.zero_grad(set_to_none=True)
    fwd().backward()
def f_full():
    opt.zero_grad(set_to_none=True)
    fwd().backward()
    opt.step()

which = sys.argv[1]
if which == "fwd": stage("fwd", f_fwd_nograd)
if which == "bwd": stage("fwd+bwd", f_fwd_bwd)
if which == "full":
    opt.prepare_step(); stage("full", f_full)
