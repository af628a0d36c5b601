"""No-GPU checks of the native boundary and host logic: the C-ABI library loads and
exports every symbol include/dfd_hip.h declares, the ctypes table covers them, the layer
plans match the published digests, and the data pipeline produces what the loaders need."""

from __future__ import annotations

import re
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

from deepfakedetection_amd import _lib, data as D
from deepfakedetection_amd.arch import efficientnet_plan, scale_channels, tf_same_padding

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    if not _lib.LIB_PATH.exists():
        from deepfakedetection_amd.build import build

        build()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    header = (ROOT / "include" / "dfd_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|size_t)\s+(dfd_\w+)\s*\(", header, flags=re.M))
    assert declared, "header parse"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dfd_version() >= 100            # a call that needs no GPU


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    assert lib.dfd_bn_finalize(None, 0, 0, 0.0, None, None, None, None, 0.1, 1e-5, None, None) == -1     # DFD_EINVAL
    assert lib.dfd_adamw_step(None, 0, None, None) == -1
    assert lib.dfd_pwconv_wgrad_ws(0, 8, 8) == 0
    assert lib.dfd_pwconv_wgrad_ws(1000, 16, 96) > 0


def test_kernel_planners_answer_without_a_gpu(lib):
    """Host-side kernel selection (no launch): which column-tile width the LDS-DMA ring kernel picks for EfficientNet-B0's mid-size
    1x1 layers at batch 256, what it leaves to the other kernels, and that the planner knobs switch it."""
    plan = lib.dfd_pw_ntd_plan
    assert plan(12544, 1152, 192) == 192 and plan(50176, 480, 80) == 96 and plan(50176, 672, 112) == 128
    assert plan(12544, 192, 1152) == 192 and plan(12544, 1152, 320) == 192 and plan(6272, 240, 40) == 64
    assert plan(960, 480, 80) == 0            # fewer than 16 row tiles
    assert plan(200704, 240, 40) == 0         # more row tiles than partial rows
    assert plan(50176, 40, 240) == 0          # K below one 64-wide step
    try:
        assert lib.dfd_tune(4, 0) == 0 and plan(12544, 1152, 192) == 0
        assert lib.dfd_tune(4, 1) == 0 and lib.dfd_tune(6, 96) == 0 and plan(12544, 1152, 192) == 96
    finally:
        lib.dfd_tune(4, 1)
        lib.dfd_tune(6, 0)
    assert lib.dfd_tune(99, 0) != 0
    assert lib.dfd_sum_passengers_discard() == 0 and lib.dfd_sum_batch_end_deferred() != 0      # no open batch: DFD_EINVAL


def test_missing_library_raises_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_plans_match_published_digests():
    b3 = efficientnet_plan("b3", "lukemelas")
    assert (b3.stem_out, b3.head_out, b3.dropout, len(b3.blocks), b3.stage_sizes) == (40, 1536, 0.3, 26, (2, 3, 3, 5, 5, 6, 2))
    assert sorted({b.cout for b in b3.blocks}) == [24, 32, 48, 96, 136, 232, 384]
    assert (b3.bn_eps, b3.bn_momentum) == (1e-3, 0.01)
    strided = [b.dw for b in b3.blocks if b.dw.stride == 2]
    assert [(g.pad_lead, g.pad_trail) for g in strided] == [(0, 1), (2, 2), (0, 1), (2, 2)]     # frozen at 300 px
    assert abs(b3.blocks[13].drop_connect - 0.2 * 13 / 26) < 1e-12 and not b3.blocks[0].expand and b3.blocks[1].skip
    b0 = efficientnet_plan("b0", "timm")
    assert [b.se_width for b in b0.blocks] == [8, 4, 6, 6, 10, 10, 20, 20, 20, 28, 28, 28, 48, 48, 48, 48]
    assert all((b.dw.pad_lead, b.dw.pad_trail) == (b.dw.kernel // 2,) * 2 for b in b0.blocks)
    size = 224
    size = b0.stem.out_size(size)
    for b in b0.blocks:
        size = b.dw.out_size(size)
    assert size == 7
    assert scale_channels(32, 1.2) == 40 and scale_channels(1280, 1.2) == 1536 and tf_same_padding(75, 5, 2) == (2, 2)
    with pytest.raises(KeyError):
        efficientnet_plan("b9", "timm")


def test_image_folder_and_transforms(tmp_path):
    rng = np.random.default_rng(1)
    for cls in ("b_fake", "a_real"):
        (tmp_path / cls / "nested").mkdir(parents=True)
        for i in range(3):
            Image.fromarray((rng.random((50, 70, 3)) * 255).astype(np.uint8)).save(tmp_path / cls / f"{i}.png")
        Image.fromarray((rng.random((30, 30)) * 255).astype(np.uint8)).save(tmp_path / cls / "nested" / "g.jpg")
        (tmp_path / cls / "notes.txt").write_text("skip me")
    ds = D.ImageFolder(tmp_path)
    assert ds.classes == ["a_real", "b_fake"] and ds.class_to_idx == {"a_real": 0, "b_fake": 1}
    assert len(ds) == 8 and ds.targets == [0] * 4 + [1] * 4
    img, target = ds[0]
    assert img.mode == "RGB" and target == 0
    torch.manual_seed(0)
    pipe = D.Compose([D.Resize(40), D.RandomResizedCrop(32, scale=(0.9, 1.0)), D.RandomRotation(10), D.RandomHorizontalFlip(),
                      D.ColorJitter(0.2, 0.2, 0.2, 0.05), D.ToTensor(), D.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
                      D.RandomErasing(p=1.0)])
    t = pipe(img)
    assert t.shape == (3, 32, 32) and t.dtype == torch.float32 and torch.isfinite(t).all()
    assert D.Resize(40)(img).size == (56, 40)                                   # shorter side -> 40, aspect kept (70x50)
    assert D.CenterCrop(32)(img).size == (32, 32) and D.RandomCrop(64)(img).size == (64, 64)   # pads when smaller
    x = D.ToTensor()(Image.fromarray(np.full((2, 2, 3), 255, np.uint8)))
    assert torch.equal(x, torch.ones(3, 2, 2))
    n = D.Normalize([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])(x)
    assert torch.allclose(n, torch.ones(3, 2, 2))
    loader = torch.utils.data.DataLoader(D.ImageFolder(tmp_path, transform=D.Compose([D.Resize((16, 16)), D.ToTensor()])), batch_size=4)
    xb, yb = next(iter(loader))
    assert xb.shape == (4, 3, 16, 16) and yb.dtype == torch.int64
    with pytest.raises(FileNotFoundError):
        D.ImageFolder(tmp_path / "a_real" / "nested")
