"""Opt-in parity against the packages that carry the reference's arithmetic (hard part 1).

`timm==1.0.20`, `efficientnet_pytorch==0.7.1` and `fastervit==1.0.0` are what the reference imports
(requirements.txt:14,16,95; trainers/efficientnet.py:405, trainers/efficientformer_v2.py:327, trainers/fastervit.py:371)
and none of them is installed in the build container or on the GPU box — every test below is SKIPPED there.  Where one
of the packages is present the test loads the oracle's state dict into the real third-party module (strict key match)
and compares logits on a seeded batch: that is the one check that would turn "parity unpinned" into "pinned".
Nothing is vendored; nothing here runs on the product path.
"""

from __future__ import annotations

import pytest
import torch


def _calibrated(model, x):
    """Running statistics := the batch's statistics (fresh (0, 1) statistics make every image's logits identical)."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    for m, mom in zip(bns, saved):
        m.momentum = mom
    return model.eval()


def _compare(ours, theirs, x, rel=1e-3):
    with torch.no_grad():
        a, b = ours(x), theirs(x)
    scale = max(float(a.abs().max()), 1e-12)
    assert float((a - b).abs().max()) <= rel * scale, float((a - b).abs().max()) / scale
    assert torch.equal(a.argmax(1), b.argmax(1))


def test_timm_efficientnet_b0_takes_the_oracle_state_dict():
    timm = pytest.importorskip("timm")
    from oracle.effnet_ref import EfficientNetRef

    torch.manual_seed(0)
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ours = _calibrated(EfficientNetRef("b0", "timm", 2), x)
    theirs = timm.create_model("efficientnet_b0", pretrained=False, num_classes=2)
    theirs.load_state_dict(ours.state_dict(), strict=True)
    _compare(ours, theirs.eval(), x)


def test_timm_efficientformerv2_s1_takes_the_oracle_state_dict():
    timm = pytest.importorskip("timm")
    from oracle.efformer_ref import EfficientFormerV2Ref

    torch.manual_seed(0)
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ours = _calibrated(EfficientFormerV2Ref("s1", 2, 224), x)
    theirs = timm.create_model("efficientformerv2_s1", pretrained=False, num_classes=2, img_size=224)
    theirs.load_state_dict(ours.state_dict(), strict=True)
    _compare(ours, theirs.eval(), x)


def test_efficientnet_pytorch_b3_takes_the_oracle_state_dict():
    enp = pytest.importorskip("efficientnet_pytorch")
    from oracle.effnet_ref import EfficientNetRef

    torch.manual_seed(0)
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ours = _calibrated(EfficientNetRef("b3", "lukemelas", 2), x)
    theirs = enp.EfficientNet.from_name("efficientnet-b3", num_classes=2)      # model_registry.py:32-36
    theirs.load_state_dict(ours.state_dict(), strict=True)
    _compare(ours, theirs.eval(), x)


def test_fastervit_0_takes_the_oracle_state_dict():
    fastervit = pytest.importorskip("fastervit")
    from oracle.fastervit_ref import FasterViTRef

    torch.manual_seed(0)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ours = _calibrated(FasterViTRef("0", 2, 224), x)
    theirs = fastervit.create_model("faster_vit_0_224", pretrained=False)
    theirs.head = torch.nn.Linear(theirs.head.in_features, 2)                  # trainers/fastervit.py:372-375
    theirs.load_state_dict(ours.state_dict(), strict=True)
    _compare(ours, theirs.eval(), x)
