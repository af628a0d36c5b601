"""Reference streaming kernel for PMC comparisons: act_bn_bwd on the block-2 depthwise tensor."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU
N, H, C = 256, 56, 144
D = torch.randn((N, H, H, C), device="cuda").bfloat16(); y = torch.randn((N, H, H, C), device="cuda").bfloat16()
st = torch.zeros((4, C), device="cuda"); st[0] = 1; st[3] = 1
for _ in range(6):
    K.act_bn_bwd(D, y, None, None, st, ACT_SILU)
torch.cuda.synchronize()
