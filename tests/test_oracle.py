"""Pins the CPU oracle (oracle/effnet_ref.py) — runs without a GPU.

The reference's own tests hold no numeric fixture for this path
(tests/test_repo_smoke.py of the reference only byte-compiles the tree), and the
packages that carry the arithmetic are not installable here, so the oracle is pinned by:
  1. published parameter counts (known answers),
  2. the third-party state-dict key sets (both naming schemes),
  3. an independent implementation of the same published architecture that IS
     installed: transformers.EfficientNetForImageClassification (TF-SAME lineage,
     BN eps 1e-3) — logits must agree after copying the weights across,
  4. committed golden logits (tests/golden/effnet_logits.json, written by
     tests/golden/make_golden.py from this oracle) so later edits cannot drift silently.
"""

from __future__ import annotations

import json
from pathlib import Path

import pytest
import torch

from oracle.effnet_ref import EfficientNetRef, build_cfg, round_filters

GOLDEN = Path(__file__).parent / "golden"


@pytest.mark.parametrize("variant,flavour,classes,want", [
    ("b0", "timm", 1000, 5_288_548),
    ("b0", "lukemelas", 1000, 5_288_548),
    ("b0", "timm", 2, 4_010_110),
    ("b0", "timm", 10, 4_020_358),
    ("b3", "lukemelas", 1000, 12_233_232),
])
def test_parameter_count_known_answers(variant, flavour, classes, want):
    model = EfficientNetRef(variant, flavour, classes)
    assert sum(p.numel() for p in model.parameters()) == want


def test_b3_structure_matches_published_digest():
    stem, stem_pad, blocks, head, dropout = build_cfg("b3", "lukemelas")
    assert (stem, head, dropout, len(blocks)) == (40, 1536, 0.3, 26)
    outs = sorted({b.cout for b in blocks})
    assert outs == [24, 32, 48, 96, 136, 232, 384]
    reps = [sum(1 for b in blocks if b.stage == s) for s in range(7)]
    assert reps == [2, 3, 3, 5, 5, 6, 2]
    # static SAME padding from the nominal 300 px: 300->150 and 150->75 pad (0,1); the k5 s2
    # layer at 75 px pads (2,2) — NOT what TF-SAME would give for a 224 px input
    assert stem_pad == (0, 1)
    s2 = [b for b in blocks if b.stride == 2]
    assert [b.pad_dw[:2] for b in s2] == [(0, 1), (2, 2), (0, 1), (2, 2)]
    assert round_filters(32, 1.2) == 40


def test_se_widths_timm_b0():
    _, _, blocks, _, _ = build_cfg("b0", "timm")
    assert [b.se_ch for b in blocks] == [8, 4, 6, 6, 10, 10, 20, 20, 20, 28, 28, 28, 48, 48, 48, 48]


def test_state_dict_keys_follow_third_party_names():
    lm = set(EfficientNetRef("b3", "lukemelas", 2).state_dict())
    for key in ("_conv_stem.weight", "_bn0.running_mean", "_blocks.0._depthwise_conv.weight", "_blocks.0._se_reduce.bias",
                "_blocks.2._expand_conv.weight", "_blocks.25._project_conv.weight", "_blocks.25._bn2.num_batches_tracked",
                "_conv_head.weight", "_bn1.weight", "_fc.weight", "_fc.bias"):
        assert key in lm, key
    assert "_blocks.0._expand_conv.weight" not in lm          # expand_ratio 1 has no expand conv
    tm = set(EfficientNetRef("b0", "timm", 2).state_dict())
    for key in ("conv_stem.weight", "bn1.weight", "blocks.0.0.conv_dw.weight", "blocks.0.0.se.conv_reduce.bias",
                "blocks.0.0.conv_pw.weight", "blocks.1.0.conv_pw.weight", "blocks.1.0.conv_pwl.weight", "blocks.1.0.bn3.bias",
                "blocks.6.0.se.conv_expand.weight", "conv_head.weight", "bn2.running_var", "classifier.bias"):
        assert key in tm, key


def test_logits_match_independent_hf_implementation():
    """Topology, padding, SE width and BN-eps cross-check on an installed, independent
    implementation of the same architecture (not the reference's dependency)."""
    transformers = pytest.importorskip("transformers")
    cfg = transformers.EfficientNetConfig(width_coefficient=1.0, depth_coefficient=1.0, image_size=224, hidden_dim=1280,
                                          dropout_rate=0.2, num_labels=10)
    hf = transformers.EfficientNetForImageClassification(cfg).eval()
    torch.manual_seed(7)
    ours = EfficientNetRef("b0", "lukemelas", 10).eval()
    # non-trivial BN statistics so eps / running stats matter
    with torch.no_grad():
        for m in ours.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    src, dst = ours.state_dict(), hf.state_dict()
    assert len(src) == len(dst)
    mapped = {}
    for (ks, vs), (kd, vd) in zip(src.items(), dst.items()):
        assert vs.shape == vd.shape, (ks, kd, vs.shape, vd.shape)
        mapped[kd] = vs
    hf.load_state_dict(mapped)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a, b = ours(x), hf(pixel_values=x).logits
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), float((a - b).abs().max())


def _hf_pair(variant: str, classes: int, **cfg_kw):
    """(oracle in the lukemelas flavour, HF model holding the oracle's weights)."""
    transformers = pytest.importorskip("transformers")
    cfg = transformers.EfficientNetConfig(num_labels=classes, **cfg_kw)
    hf = transformers.EfficientNetForImageClassification(cfg)
    torch.manual_seed(7)
    ours = EfficientNetRef(variant, "lukemelas", classes)
    with torch.no_grad():
        for m in ours.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    src, dst = ours.state_dict(), hf.state_dict()
    assert len(src) == len(dst)
    mapped = {}
    for (ks, vs), (kd, vd) in zip(src.items(), dst.items()):
        assert vs.shape == vd.shape, (ks, kd, vs.shape, vd.shape)
        mapped[kd] = vs.clone()
    hf.load_state_dict(mapped)
    return ours, hf


def test_b3_static_same_padding_at_224_matches_independent_hf_implementation():
    """The reference's own model: EfficientNet-B3 (efficientnet_pytorch), whose TF-"SAME" padding is frozen at the
    NOMINAL 300 px and then evaluated on 224 px inputs (trainers/efficientnet.py:405, img_size 224).  HF's model has the
    same static rule built differently: ZeroPad2d((k//2 - 1, k//2)) in front of every stride-2 depthwise convolution
    except the block indices in `depthwise_padding` (HF's published b3 config: [5, 18] — the two k5 s2 layers that
    see an ODD nominal resolution, 75 and 19 px), which pad symmetrically."""
    ours, hf = _hf_pair("b3", 7, width_coefficient=1.2, depth_coefficient=1.4, image_size=300, hidden_dim=1536,
                        dropout_rate=0.3, depthwise_padding=[5, 18])
    assert sum(p.numel() for p in hf.parameters()) == sum(p.numel() for p in ours.parameters())
    ours.eval(); hf.eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        a, b = ours(x), hf(pixel_values=x).logits
    assert torch.allclose(a, b, rtol=1e-4, atol=2e-5), float((a - b).abs().max())
    # and the rule matters: with TF-SAME computed for the 224 px input instead, those layers would pad differently
    _, _, blocks, _, _ = build_cfg("b3", "lukemelas")
    assert blocks[5].pad_dw[:2] == (2, 2) and blocks[18].pad_dw[:2] == (2, 2)


def test_training_mode_matches_independent_hf_implementation():
    """Batch-statistics BatchNorm, running-statistics update (momentum 0.01 in torch's convention, eps 1e-3) and the
    gradient of every parameter, B0 at 96 px.  Stochastic parts off on both sides (HF applies elementwise dropout
    where efficientnet_pytorch applies per-sample drop-connect: not comparable, and not what is pinned here)."""
    ours, hf = _hf_pair("b0", 5, width_coefficient=1.0, depth_coefficient=1.0, image_size=224, hidden_dim=1280, dropout_rate=0.0,
                        drop_connect_rate=0.0, batch_norm_momentum=0.01)
    ours.train(); hf.train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 3, 96, 96, generator=g)
    y = torch.randint(0, 5, (6,), generator=g)
    bns_o = [m for m in ours.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    bns_h = [m for m in hf.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    before = [(m.running_mean.clone(), m.running_var.clone()) for m in bns_o]
    a = ours(x)
    b = hf(pixel_values=x).logits
    assert torch.allclose(a, b, rtol=1e-4, atol=2e-5), float((a - b).abs().max())
    torch.nn.functional.cross_entropy(a, y, label_smoothing=0.1).backward()
    torch.nn.functional.cross_entropy(b, y, label_smoothing=0.1).backward()
    mine = dict(ours.named_parameters())
    for (n1, p1), (n2, p2) in zip(ours.named_parameters(), hf.named_parameters()):
        assert p1.shape == p2.shape, (n1, n2)
        scale = max(float(p1.grad.abs().max()), 1e-8)
        # a BN bias in front of another batch-statistics BN (block without skip) has a structurally zero gradient:
        # both sides hold cancellation noise, so a bias is judged on the scale of its layer's (weight, bias) gradient
        sib = n1[:-4] + "weight"
        if n1.endswith("bias") and sib in mine:
            scale = max(scale, float(mine[sib].grad.abs().max()))
        assert float((p1.grad - p2.grad).abs().max()) <= 2e-3 * scale + 1e-7, (n1, n2)
    # running statistics: new = (1 - m) * old + m * batch statistic.  efficientnet_pytorch uses m = 0.01 everywhere;
    # HF forgets the momentum argument on its expansion BatchNorms (torch default 0.1), so the batch statistic each
    # side folded in is compared, with each module's own momentum
    assert len(bns_o) == len(bns_h) == 49
    for mo, mh, (m0, v0) in zip(bns_o, bns_h, before):
        assert mo.momentum == 0.01 and mo.eps == mh.eps == 1e-3
        for new_o, new_h, old in ((mo.running_mean, mh.running_mean, m0), (mo.running_var, mh.running_var, v0)):
            stat_o = (new_o - (1 - mo.momentum) * old) / mo.momentum
            stat_h = (new_h - (1 - mh.momentum) * old) / mh.momentum
            assert torch.allclose(stat_o, stat_h, rtol=2e-3, atol=2e-4), float((stat_o - stat_h).abs().max())
        assert int(mo.num_batches_tracked) == int(mh.num_batches_tracked) == 1


def test_golden_logits_fixture():
    path = GOLDEN / "effnet_logits.json"
    data = json.loads(path.read_text())
    for case in data["cases"]:
        torch.manual_seed(case["seed"])
        model = EfficientNetRef(case["variant"], case["flavour"], case["classes"]).eval()
        g = torch.Generator().manual_seed(case["input_seed"])
        x = torch.randn(case["batch"], 3, case["size"], case["size"], generator=g)
        with torch.no_grad():
            got = model(x)
        want = torch.tensor(case["logits"])
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (case["variant"], case["flavour"])


def test_train_mode_updates_running_stats_and_grads_flow():
    torch.manual_seed(3)
    model = EfficientNetRef("b0", "timm", 2).train()
    x = torch.randn(4, 3, 64, 64)
    before = model.bn1.running_mean.clone()
    out = model(x)
    out.sum().backward()
    assert not torch.equal(before, model.bn1.running_mean)
    assert int(model.bn1.num_batches_tracked) == 1
    assert all(p.grad is not None for p in model.parameters())


# ---------------------------------------------------------------------------------------------------------------------
# The timm flavour (BASELINE config 2: what bench.py measures) against the same independent implementation.
# timm differs from the TF lineage in three places and HF's model can be configured into each:
#   * BatchNorm eps 1e-5 / momentum 0.1                     -> batch_norm_eps=1e-5, batch_norm_momentum=0.1
#   * stride-2 depthwise convolutions pad k//2 on all sides -> every stride-2 block index (B0: 1, 3, 5, 11) listed in
#                                                              `depthwise_padding` (HF: "adjust_padding = idx not in list")
#   * the stem pads 1 on all sides; HF's stem pads (0, 1)   -> HF is fed the image shifted by one pixel (zero first
#                                                              row / column) and the image has a zero last row / column,
#                                                              so both stems multiply the same 3x3 windows.
# SE widths (timm round(cin/4) vs HF int(cin/4)) coincide for every B0 block.


def _hf_timm_twin(classes: int, calibrate_on=None):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.EfficientNetConfig(width_coefficient=1.0, depth_coefficient=1.0, image_size=224, hidden_dim=1280,
                                          dropout_rate=0.0, drop_connect_rate=0.0, num_labels=classes,
                                          batch_norm_eps=1e-5, batch_norm_momentum=0.1, depthwise_padding=[1, 3, 5, 11])
    hf = transformers.EfficientNetForImageClassification(cfg)
    torch.manual_seed(17)
    ours = EfficientNetRef("b0", "timm", classes)
    ours.drop_rate = 0.0
    with torch.no_grad():
        for m in ours.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    if calibrate_on is not None:
        # running statistics = this batch's statistics: with the fresh (0, 1) statistics the signal dies out over 80
        # layers in eval mode and every image gets the same logits (a check that could not see the input)
        bns = [m for m in ours.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        saved = [m.momentum for m in bns]
        for m in bns:
            m.momentum = 1.0
        ours.train()
        with torch.no_grad():
            ours(calibrate_on)
        for m, mom in zip(bns, saved):
            m.momentum = mom
            m.num_batches_tracked.zero_()
    src, dst = ours.state_dict(), hf.state_dict()
    assert len(src) == len(dst)
    mapped = {}
    for (ks, vs), (kd, vd) in zip(src.items(), dst.items()):
        assert vs.shape == vd.shape, (ks, kd, vs.shape, vd.shape)
        mapped[kd] = vs.clone()
    hf.load_state_dict(mapped)
    return ours, hf


def _stem_aligned_inputs(n: int, size: int, seed: int):
    """(x for the timm-flavour oracle, x shifted by one pixel for HF's (0, 1)-padded stem)."""
    x = torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(seed))
    x[:, :, -1, :] = 0
    x[:, :, :, -1] = 0
    shifted = torch.zeros_like(x)
    shifted[:, :, 1:, 1:] = x[:, :, :-1, :-1]
    return x, shifted


def test_timm_flavour_b0_eval_logits_match_independent_hf_implementation():
    x, xs = _stem_aligned_inputs(4, 224, 5)
    ours, hf = _hf_timm_twin(10, calibrate_on=x)
    assert sum(p.numel() for p in hf.parameters()) == sum(p.numel() for p in ours.parameters()) == 4_007_548 + 1280 * 10 + 10
    ours.eval(); hf.eval()
    with torch.no_grad():
        a, b = ours(x), hf(pixel_values=xs).logits
    assert float((a - a.mean(0, keepdim=True)).abs().max()) > 1e-2, "degenerate input: identical logits for every image"
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), float((a - b).abs().max())
    assert torch.equal(a.argmax(1), b.argmax(1))
    # the flavour matters: the TF-lineage oracle with the same weights gives other logits (eps and padding differ)
    other = EfficientNetRef("b0", "lukemelas", 10).eval()
    other.load_state_dict({k2: v for (k2, _), v in zip(other.state_dict().items(), ours.state_dict().values())})
    with torch.no_grad():
        c = other(x)
    assert float((a - c).abs().max()) > 1e-3 * float(a.abs().max())


def test_timm_flavour_b0_training_mode_matches_independent_hf_implementation():
    """Batch-statistic BatchNorm with eps 1e-5, running-statistic update with momentum 0.1, and every parameter gradient."""
    ours, hf = _hf_timm_twin(5)
    ours.train(); hf.train()
    x, xs = _stem_aligned_inputs(6, 96, 6)
    y = torch.randint(0, 5, (6,), generator=torch.Generator().manual_seed(7))
    bns_o = [m for m in ours.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    bns_h = [m for m in hf.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    a = ours(x)
    b = hf(pixel_values=xs).logits
    assert torch.allclose(a, b, rtol=1e-4, atol=2e-5), float((a - b).abs().max())
    torch.nn.functional.cross_entropy(a, y, label_smoothing=0.1).backward()
    torch.nn.functional.cross_entropy(b, y, label_smoothing=0.1).backward()
    mine = dict(ours.named_parameters())
    for (n1, p1), (n2, p2) in zip(ours.named_parameters(), hf.named_parameters()):
        assert p1.shape == p2.shape, (n1, n2)
        scale = max(float(p1.grad.abs().max()), 1e-8)
        sib = n1[:-4] + "weight"
        if n1.endswith("bias") and sib in mine:       # structurally zero BN-bias gradients: judged on the layer's scale
            scale = max(scale, float(mine[sib].grad.abs().max()))
        assert float((p1.grad - p2.grad).abs().max()) <= 2e-3 * scale + 1e-7, (n1, n2)
    assert len(bns_o) == len(bns_h) == 49
    for mo, mh in zip(bns_o, bns_h):
        assert mo.momentum == mh.momentum == 0.1 and mo.eps == mh.eps == 1e-5
        assert torch.allclose(mo.running_mean, mh.running_mean, rtol=2e-3, atol=2e-5)
        assert torch.allclose(mo.running_var, mh.running_var, rtol=2e-3, atol=2e-5)
        assert int(mo.num_batches_tracked) == int(mh.num_batches_tracked) == 1
