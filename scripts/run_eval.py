"""The evaluate() forward of EfficientNet-B0 (batch 256, 224 px), repeated — for rocprofv3 --kernel-trace --stats.
python run_eval.py [reps] [f32|bf16]; DFD_EVAL_FUSED=0 runs the training-form kernel chain instead of the inference form."""
import contextlib
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd.efficientnet import HipEfficientNet

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
amp = len(sys.argv) > 2 and sys.argv[2] == "bf16"
torch.manual_seed(0)
model = HipEfficientNet("b0", "timm", 2).cuda().eval()
x = torch.randn(256, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
with torch.inference_mode(), (torch.autocast("cuda", dtype=torch.bfloat16) if amp else contextlib.nullcontext()):
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        model(x)
    e1.record()
    torch.cuda.synchronize()
print(f"eval forward ({'bf16' if amp else 'f32'}): {e0.elapsed_time(e1) / reps:.3f} ms per batch of 256 ({256 * reps / e0.elapsed_time(e1) * 1e3:.0f} images/s)")
