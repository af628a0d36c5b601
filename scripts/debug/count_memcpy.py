"""Memcpy / memset activity inside one eager train step (what would become copy nodes of the captured graph)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from torch.profiler import profile, ProfilerActivity
from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
from deepfakedetection_amd.orchestration.model_registry import get_model_spec

name = sys.argv[1] if len(sys.argv) > 1 else "faster_vit_0_224"
model = get_model_spec(name).builder(name, 2).cuda().train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
crit = HipCrossEntropyLoss(0.1)
x = torch.randn(64, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
y = torch.randint(0, 2, (64,), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), y)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "emset" in n or n.startswith("aten::copy_") or n.startswith("aten::to") or "_to_copy" in n:
        c[n] += 1
for k, v in c.most_common(20):
    print(v, k)
