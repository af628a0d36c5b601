"""The drop-in loop's device_batches() prefetch against resident inputs, per model."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from deepfakedetection_amd.graph_step import GraphedTrainStep
from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
from deepfakedetection_amd.orchestration.model_registry import get_model_spec
from deepfakedetection_amd.trainers.efficientnet import device_batches

name = sys.argv[1] if len(sys.argv) > 1 else "faster_vit_0_224"
torch.manual_seed(0)
model = get_model_spec(name).builder(name, 2).cuda().train()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
st = GraphedTrainStep(model, HipCrossEntropyLoss(0.1), opt, accum_steps=1)
g = torch.Generator().manual_seed(1)
batches = [(torch.randn(256, 3, 224, 224, generator=g).pin_memory(), torch.randint(0, 2, (256,), generator=g)) for _ in range(4)]
class DL:
    def __init__(s, n): s.n = n
    def __iter__(s):
        for i in range(s.n): yield batches[i % 4]
def run(n, prefetch, item):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc = torch.zeros((), dtype=torch.float64, device="cuda")
    for x, y in device_batches(DL(n), "cuda", None, prefetch=prefetch):
        loss = st.micro_batch(x, y, first=True, last=True)
        st.optimizer_step()
        acc += loss.detach().double()
        if item: float(acc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(6, True, False)
for pf, item in ((True, False), (False, False), (True, True)):
    print(name, "prefetch" if pf else "in-stream", "item-per-step" if item else "", round(run(30, pf, item), 3), "ms/step", flush=True)
