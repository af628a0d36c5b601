import sys, collections, torch
sys.path.insert(0, "/root/repo")
from torch.utils._python_dispatch import TorchDispatchMode
from deepfakedetection_amd.efficientnet import HipEfficientNet
from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
class Count(TorchDispatchMode):
    def __init__(self): super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.c[str(func)] += 1
        return func(*args, **(kwargs or {}))
torch.manual_seed(0)
m = HipEfficientNet("b0", "timm", 2).cuda().train()
opt = HipAdamW(m.parameters(), lr=1e-4, weight_decay=5e-2)
crit = HipCrossEntropyLoss(0.1)
x = torch.randn(32, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last); y = torch.randint(0, 2, (32,), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(m(x), y)
    loss.backward(); opt.step()
for _ in range(2): step()
with Count() as c:
    step()
for k, v in c.c.most_common(25): print(v, k)
