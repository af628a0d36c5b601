"""BASELINE configs 3 and 5 at their own size: EfficientFormerV2-S1 and FasterViT-0, batch 256, 224 px, bf16 autocast —
the shapes `bench.py --model ...` runs (VERDICT r2 "what's weak" 2: the network-level bf16 checks of these two families
stopped at batch 8, where the persistent grids, magic divisions and window / carrier-token index maps are far from the
sizes the benchmark uses).

One full training step per family against the f32 CPU oracle on the same seeded batch (DropPath off on both sides: the
masks come from different generators):
  * logits within the oracle's OWN bf16 yardstick — the oracle under torch's CPU bf16 autocast against its f32 self —
    (floor 2e-2 of the largest logit), loss within 2e-2;
  * every parameter gets a finite gradient and the flattened gradient has cosine >= 0.98 with the oracle's;
  * BatchNorm running statistics of the first layers within 2e-2 (f32 sums over 0.8-3.2 M bf16 values per channel);
  * arg-max of the logits equal to the oracle's wherever the oracle's top-2 margin exceeds the yardstick.
"""

from __future__ import annotations

import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
N = 256


def rel_err(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-12)


def _step(ref, hip, bn_names):
    from deepfakedetection_amd.optim import HipCrossEntropyLoss

    ref.train(); hip.train()
    g = torch.Generator().manual_seed(1)                                  # bench.py's generator seed
    x = torch.randn(N, 3, 224, 224, generator=g).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 2, (N,), generator=g)
    with torch.no_grad(), torch.autocast("cpu", dtype=BF):
        auto = copy.deepcopy(ref)(x).float()                              # yardstick first: it must not see updated BN buffers
    ref_logits = ref(x)
    ref_loss = F.cross_entropy(ref_logits, y, label_smoothing=0.1)
    ref_loss.backward()
    with torch.autocast("cuda", dtype=BF):
        logits = hip(x.cuda())
        loss = HipCrossEntropyLoss(0.1)(logits, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all()
    yard, err = rel_err(auto, ref_logits), rel_err(logits, ref_logits)
    print(f"N={N}: logits rel err {err:.4f} (oracle's own bf16 autocast: {yard:.4f}); loss {float(loss):.5f} vs {float(ref_loss):.5f}")
    assert err <= max(yard, 2e-2), (err, yard)
    assert abs(float(loss) - float(ref_loss)) <= 2e-2 * max(1.0, abs(float(ref_loss)))
    top2 = ref_logits.detach().topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * max(yard, 2e-2) * float(ref_logits.abs().max())
    assert torch.equal(logits.float().cpu().argmax(1)[clear], ref_logits.argmax(1)[clear])
    rb, hb = dict(ref.named_buffers()), dict(hip.named_buffers())
    for name in bn_names:
        for stat in ("running_mean", "running_var"):
            a, b = hb[f"{name}.{stat}"].float().cpu(), rb[f"{name}.{stat}"]
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-2 * scale + 1e-4, (name, stat, float((a - b).abs().max()), scale)
    rp = dict(ref.named_parameters())
    for name, p in hip.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    ga = torch.cat([p.grad.float().cpu().flatten() for _, p in hip.named_parameters()])
    gb = torch.cat([rp[n].grad.flatten() for n, _ in hip.named_parameters()])
    ga, gb = ga.double(), gb.double()                         # 31 M entries: f32 dot / norm accumulations drift above 1
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
    print(f"gradient cosine vs f32 oracle at N={N}: {cos:.4f}")
    assert cos >= 0.98, cos


def test_efficientformerv2_s1_training_step_at_the_benchmark_configuration():
    from tests.test_efformer_gpu import make_pair

    ref, hip = make_pair("s1", 2, 224, seed=5)
    names = [n[: -len(".running_mean")] for n, _ in ref.named_buffers() if n.endswith("running_mean")][:8]
    _step(ref, hip, names)


def test_fastervit_0_training_step_at_the_benchmark_configuration():
    from tests.test_fastervit_gpu import make_pair

    ref, hip = make_pair("0", 2, seed=5, dpr=0.0)
    names = [n[: -len(".running_mean")] for n, _ in ref.named_buffers() if n.endswith("running_mean")][:8]
    _step(ref, hip, names)
