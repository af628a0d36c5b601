"""Graphed eval forward at batch 256 in a fresh process: the fused linear kernels' first launch may fall inside a capture."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from deepfakedetection_amd.graph_step import GraphedForward
from deepfakedetection_amd.orchestration.model_registry import get_model_spec

name = sys.argv[1] if len(sys.argv) > 1 else "faster_vit_0_224"
model = get_model_spec(name).builder(name, 2).cuda().eval()
x = torch.randn(256, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
fwd = GraphedForward(model, amp_dtype=torch.bfloat16) if "amp_dtype" in GraphedForward.__init__.__code__.co_varnames else GraphedForward(model)
with torch.inference_mode():
    a = fwd(x).float().clone()
    b = fwd(x).float().clone()
    c = fwd(x).float().clone()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref = model(x).float()
print(name, "replay == replay:", torch.equal(b, c), " first == replay:", torch.equal(a, b), " max|graph - eager|:", float((c - ref).abs().max()))
