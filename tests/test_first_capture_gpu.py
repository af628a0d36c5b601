"""A kernel's FIRST launch in a process may fall inside a hipGraph capture (graph_step.GraphedForward captures evaluate()'s forward
on the second sight of a shape; a fused layer that only serves large batches is then launched for the first time under capture).
The host-side attribute caches (hipFuncSetAttribute / hipFuncGetAttributes behind std::call_once) must tolerate that:
in a FRESH process, the first launch of each such kernel happens inside a capture and the replay must equal an eager call bit
for bit (the round-3 debug script scripts/debug/eval_graph_first_capture.py turned into a test)."""

from __future__ import annotations

import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from deepfakedetection_amd import kernels as K
from deepfakedetection_amd._lib import ACT_GELU, ACT_SILU, load
lib = load()
g = torch.Generator().manual_seed(3)
# (a) the fused linear of csrc/dfd_gemm.hip (> 64 KB of dynamic LDS: hipFuncSetAttribute on first use)
M, Kd, N = 256 * 170, 256, 256
x = torch.randn((M // 49 if M % 49 == 0 else M, 49 if M % 49 == 0 else 1, 1, Kd), generator=g).to(torch.bfloat16).cuda()
w = (torch.randn((N, Kd), generator=g) * Kd ** -0.5).cuda()
w_nk, _ = K.prep_weights(w, torch.bfloat16, True, False)
st = torch.stack([0.5 + torch.rand(N, generator=g), torch.randn(N, generator=g) * 0.3, torch.zeros(N), torch.ones(N)]).cuda()
assert lib.dfd_gemm_plan(x.shape[0] * x.shape[1], Kd, N) != 0
# (b) the wave-autonomous 1x1 kernel (occupancy from hipFuncGetAttributes on first use) and (c) the matrix-core depthwise forward
a = torch.randn((64, 28, 28, 40), generator=g).to(torch.bfloat16).cuda()
w2 = (torch.randn((240, 40), generator=g) * 0.1).cuda()
w2_nk, _ = K.prep_weights(w2, torch.bfloat16, True, False)
xd = torch.randn((8, 7, 7, 1152), generator=g).to(torch.bfloat16).cuda()
wd = (torch.randn((1152, 1, 5, 5), generator=g) * 0.2).cuda()
sd = torch.stack([0.5 + torch.rand(1152, generator=g), torch.randn(1152, generator=g) * 0.3, torch.zeros(1152), torch.ones(1152)]).cuda()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    f_out, _ = K.gemm_bias_act(x, w_nk, st, ACT_GELU, None, None, want_raw=False)
    p_out, _, _ = K.pwconv(a, None, w2_nk, None, stats=False)
    d_out, d_parts, d_n = K.dwconv_fwd(xd, sd, ACT_SILU, wd, 5, 1, 2, 2, 7, 7, True)
graph.replay()
torch.cuda.synchronize()
got = [t.clone() for t in (f_out, p_out, d_out)]
graph.replay()
torch.cuda.synchronize()
again = [t.clone() for t in (f_out, p_out, d_out)]
e_f, _ = K.gemm_bias_act(x, w_nk, st, ACT_GELU, None, None, want_raw=False)
e_p, _, _ = K.pwconv(a, None, w2_nk, None, stats=False)
e_d, _, _ = K.dwconv_fwd(xd, sd, ACT_SILU, wd, 5, 1, 2, 2, 7, 7, True)
torch.cuda.synchronize()
for name, r1, r2, e in zip(("fused linear", "1x1 wave-autonomous", "depthwise matrix-core"), got, again, (e_f, e_p, e_d)):
    assert torch.isfinite(r1.float()).all(), name
    assert torch.equal(r1, r2), name + ": two replays differ"
    assert torch.equal(r1, e), name + ": the replay of a first-launch-under-capture differs from an eager launch"
print("FIRST-CAPTURE-OK")
"""


def test_first_launch_inside_a_capture_replays_like_an_eager_launch():
    root = str(Path(__file__).resolve().parents[1])
    proc = subprocess.run([sys.executable, "-c", CHILD.format(root=root)], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "FIRST-CAPTURE-OK" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]
