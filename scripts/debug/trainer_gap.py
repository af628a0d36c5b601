"""Where the drop-in loop's time goes beyond the replayed step: resident inputs vs the loop's own H2D path."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from deepfakedetection_amd.graph_step import GraphedTrainStep
from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
from deepfakedetection_amd.orchestration.model_registry import get_model_spec

name = sys.argv[1] if len(sys.argv) > 1 else "faster_vit_0_224"
torch.manual_seed(0)
model = get_model_spec(name).builder(name, 2).cuda()
opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
crit = HipCrossEntropyLoss(0.1)
st = GraphedTrainStep(model, crit, opt, accum_steps=1)
g = torch.Generator().manual_seed(1)
xs = [torch.randn(256, 3, 224, 224, generator=g).pin_memory() for _ in range(2)]
ys = [torch.randint(0, 2, (256,), generator=g) for _ in range(2)]
xd = [x.cuda().to(memory_format=torch.channels_last) for x in xs]
yd = [y.cuda() for y in ys]
model.train()
def run(n, mode):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if mode == "resident":
            x, y = xd[i % 2], yd[i % 2]
        else:
            x = xs[i % 2].to("cuda", non_blocking=True).to(memory_format=torch.channels_last); y = ys[i % 2].to("cuda", non_blocking=True)
        loss = st.micro_batch(x, y, first=True, last=True)
        st.optimizer_step()
        if mode == "resident+item":
            float(loss)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    run(4, "resident")
for mode in ("resident", "h2d", "resident"):
    print(name, mode, round(run(30, mode), 3), "ms/step", flush=True)
