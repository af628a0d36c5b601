"""oracle/image_ref.py against Pillow itself (the reference's rotation / colour-jitter arithmetic lives in Pillow via torchvision):
byte work, so every comparison is exact."""

from __future__ import annotations

import numpy as np
import pytest
from PIL import Image, ImageEnhance

from oracle import image_ref as R


def test_blends_match_imageenhance_byte_for_byte():
    rng = np.random.default_rng(0)
    for trial in range(40):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if trial % 3 == 0:
            img[: h // 2] = 0
        pil = Image.fromarray(img)
        f = float(rng.uniform(0.0, 2.0)) if trial % 5 else [0.0, 1.0, 0.8, 1.2][trial % 4]
        for fn, enh in ((R.brightness, ImageEnhance.Brightness), (R.contrast, ImageEnhance.Contrast), (R.color, ImageEnhance.Color)):
            assert np.array_equal(np.array(enh(pil).enhance(f)), fn(img, f)), (fn.__name__, f, (h, w))


@pytest.mark.parametrize("step", [5])
def test_hsv_conversions_match_pillow_on_a_lattice(step):
    v = np.arange(0, 256, step, dtype=np.uint8)
    v = np.unique(np.concatenate([v, np.array([1, 2, 127, 128, 254, 255], dtype=np.uint8)]))
    a, b, c = np.meshgrid(v, v, v, indexing="ij")
    tri = np.stack([a.ravel(), b.ravel(), c.ravel()], -1)[None]
    assert np.array_equal(np.array(Image.fromarray(tri, "RGB").convert("HSV")), R.rgb2hsv(tri))
    assert np.array_equal(np.array(Image.fromarray(tri, "HSV").convert("RGB")), R.hsv2rgb(tri))


def test_rotate_matches_pillow_nearest():
    rng = np.random.default_rng(1)
    for trial in range(60):
        h, w = (224, 224) if trial % 4 == 0 else (int(rng.integers(1, 260)), int(rng.integers(1, 260)))
        if trial % 7 == 3:
            w = h
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ang = float(rng.uniform(-10, 10)) if trial % 6 else [0.0, 180.0, 90.0, 270.0, 45.0, -0.0, 360.0, -180.0][trial % 8]
        want = np.array(Image.fromarray(img).rotate(ang, resample=Image.NEAREST, expand=False))
        assert np.array_equal(want, R.rotate(img, ang)), ((h, w), ang)


def test_jitter_chain_matches_the_pil_transform_with_the_same_draws():
    """data.ColorJitter draws a permutation and four factors; given those, the oracle's chain equals the PIL chain"""
    import torch

    from deepfakedetection_amd import data as D

    rng = np.random.default_rng(2)
    cj = D.ColorJitter(0.2, 0.2, 0.2, 0.05)
    for trial in range(12):
        img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
        torch.manual_seed(trial)
        state = torch.get_rng_state()
        want = np.array(cj(Image.fromarray(img)))
        torch.set_rng_state(state)
        order = torch.randperm(4).tolist()
        draws = {}
        for which in order:
            lo, hi = ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.05, 0.05))[which]
            draws[which] = D._uniform(lo, hi)
        assert np.array_equal(want, R.jitter(img, order, draws[0], draws[1], draws[2], draws[3])), trial
