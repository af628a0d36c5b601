"""Which lines of the engine issue ATen copy / fill kernels inside one train step (they become copy / memset nodes of the graph)."""
import collections
import sys
import traceback
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
from deepfakedetection_amd.orchestration.model_registry import get_model_spec

name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b0"
model = get_model_spec(name).builder(name, 2).cuda().train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
crit = HipCrossEntropyLoss(0.1)
x = torch.randn(32, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
y = torch.randint(0, 2, (32,), device="cuda")


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), y)
    loss.backward()
    opt.step()


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func)
        if not any(k in n for k in ("aten.empty", "view", "reshape", "as_strided", "detach", "alias", "permute", "transpose", "slice", "select",
                                    "unsqueeze", "squeeze", "expand", "aten.t.", "unbind", "split", "is_", "_local_scalar", "sym_", "stride", "size")):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "deepfakedetection_amd" in fr.filename and "aten_sites" not in fr.filename:
                    site = f"{Path(fr.filename).name}:{fr.lineno} {fr.name}"
                    break
            shape = tuple(args[0].shape) if args and isinstance(args[0], torch.Tensor) else ()
            self.c[(n, site, str(shape)[:40])] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)      # the dispatch mode is thread-local: keep the backward on this thread
for _ in range(3):
    step()
torch.cuda.synchronize()
with Sites() as s:
    step()
for (n, site, shape), v in sorted(s.c.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v:4d}  {n:<34} {site:<50} {shape}")
