"""Run ONE kernel of one EfficientNet-B0 block repeatedly (for rocprofv3 --pmc / --kernel-trace).
    python scripts/run_one.py <op> <block> [reps] [batch]
op: dw_fwd | dw_bwd_data | dw_bwd_weight | pw_expand | pw_project | pw_exp_wgrad | pw_proj_wgrad | pw_exp_dgrad | pw_proj_dgrad
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_SILU
from deepfakedetection_amd.arch import efficientnet_plan

op, blk = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
plan = efficientnet_plan("b0", "timm")
H = 112
for b in plan.blocks:
    if b.index == blk:
        break
    H = b.dw.out_size(H)
Ho, Cm, g = b.dw.out_size(H), b.cmid, b.dw
DT = torch.bfloat16
def st(C):
    s = torch.zeros((4, C), device="cuda"); s[0] = 1; s[3] = 1; return s
def cf(C):
    c = torch.zeros((3, C), device="cuda"); c[0] = 1; return c
if op.startswith("dw"):
    x = torch.randn((N, H, H, Cm), device="cuda").to(DT)
    w = torch.randn((Cm, 1, g.kernel, g.kernel), device="cuda") * 0.2
    dz = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT); y = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
    s, c = st(Cm), cf(Cm)
    fn = {"dw_fwd": lambda: K.dwconv_fwd(x, s, ACT_SILU, w, g.kernel, g.stride, g.pad_lead, g.pad_lead, Ho, Ho, True),
          "dw_bwd_data": lambda: K.dwconv_bwd_data(dz, y, c, w, x, s, ACT_SILU, tuple(x.shape), g.kernel, g.stride, g.pad_lead, g.pad_lead),
          "dw_bwd_weight": lambda: K.dwconv_bwd_weight(dz, y, c, x, s, ACT_SILU, g.kernel, g.stride, g.pad_lead, g.pad_lead)}[op]
else:
    a = torch.randn((N, H, H, b.cin), device="cuda").to(DT)
    w1 = torch.randn((Cm, b.cin), device="cuda") * 0.1
    w1_nk, w1_kn = K.prep_weights(w1, DT)
    dz1 = torch.randn((N, H, H, Cm), device="cuda").to(DT); y1 = torch.randn((N, H, H, Cm), device="cuda").to(DT)
    pro1 = K.pro_affine2(y1, cf(Cm))
    y2 = torch.randn((N, Ho, Ho, Cm), device="cuda").to(DT)
    w3 = torch.randn((b.cout, Cm), device="cuda") * 0.1
    w3_nk, w3_kn = K.prep_weights(w3, DT)
    gate = torch.rand((N, Cm), device="cuda")
    prog = K.pro_bn_act_gate(st(Cm), ACT_SILU, gate, Ho * Ho)
    gb = torch.randn((N, Ho, Ho, b.cout), device="cuda").to(DT); y3 = torch.randn((N, Ho, Ho, b.cout), device="cuda").to(DT)
    pro3 = K.pro_affine2(y3, cf(b.cout))
    fn = {"pw_expand": lambda: K.pwconv(a, None, w1_nk, None, True),
          "pw_exp_dgrad": lambda: K.pwconv(dz1, pro1, w1_kn, None, False),
          "pw_exp_wgrad": lambda: K.pwconv_wgrad(dz1, pro1, a, None),
          "pw_project": lambda: K.pwconv(y2, prog, w3_nk, None, True),
          "pw_proj_dgrad": lambda: K.pwconv(gb, pro3, w3_kn, None, False),
          "pw_proj_wgrad": lambda: K.pwconv_wgrad(gb, pro3, y2, prog)}[op]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"{op} blk{blk}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
