"""CPU restatement (numpy, integer / IEEE arithmetic spelled out) of the PIL transforms between the crop and ToTensor in the
reference's DEFAULT 224-pixel training pipeline — RandomRotation(10) and ColorJitter(0.2, 0.2, 0.2, 0.05),
/root/reference/trainers/efficientnet.py:173-181 (same in trainers/efficientformer_v2.py and trainers/fastervit.py) — as
deepfakedetection_amd/data.py performs them with Pillow (RandomRotation.__call__, ColorJitter.__call__, _shift_hue).

TEST INFRASTRUCTURE: only tests/ may import this module.  It is the oracle of csrc/dfd_augment.hip.

Pinned against Pillow itself (tests/test_image_oracle.py, `-m "not gpu"`): rotate on random sizes / angles incl. the 0 / 90 / 180 /
270 special cases, the three ImageEnhance blends for factors below and above 1, and RGB <-> HSV on a lattice of the 2^24 triples
(all 2^24 were checked once in both directions when this file was written: 0 mismatches).  The algorithms restated:
  * Image.rotate(angle, NEAREST, expand=False): the affine matrix of Image.rotate (cos / sin rounded to 15 digits, centre w/2, h/2)
    and Geometry.c's affine_fixed — 16.16 fixed point, FIX(v) = floor(v * 65536 + 0.5), sample at pixel centres, fill 0;
  * ImageEnhance.{Brightness, Contrast, Color}.enhance(f) = Image.blend(degenerate, image, f): Blend.c's float expression
    (int)a + alpha * ((int)b - (int)a) in f32, truncated for 0 <= f <= 1, clipped then truncated otherwise; the degenerates are
    black, the rounded mean of convert("L"), and convert("L") per pixel, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
  * convert("HSV") / convert("RGB") of Convert.c (rgb2hsv_row / hsv2rgb following colorsys, float variables, double constants).
"""

from __future__ import annotations

import math

import numpy as np


def to_l(rgb: np.ndarray) -> np.ndarray:
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg: np.ndarray, img: np.ndarray, factor: float) -> np.ndarray:
    a = np.float32(factor)
    d, i = deg.astype(np.int32), img.astype(np.int32)
    temp = d.astype(np.float32) + a * (i - d).astype(np.float32)
    if 0.0 <= factor <= 1.0:
        return temp.astype(np.int32).astype(np.uint8)
    return np.where(temp <= 0, 0, np.where(temp >= 255, 255, temp.astype(np.int32))).astype(np.uint8)


def brightness(img: np.ndarray, f: float) -> np.ndarray:
    return blend(np.zeros_like(img), img, f)


def contrast_mean(img: np.ndarray) -> int:
    lum = to_l(img)
    return int(int(lum.astype(np.int64).sum()) / lum.size + 0.5)


def contrast(img: np.ndarray, f: float) -> np.ndarray:
    return blend(np.full_like(img, contrast_mean(img)), img, f)


def color(img: np.ndarray, f: float) -> np.ndarray:
    return blend(np.repeat(to_l(img)[..., None], 3, -1), img, f)


def rgb2hsv(rgb: np.ndarray) -> np.ndarray:
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc, gc, bc = ((maxc - c).astype(np.float32) / safe for c in (r, g, b))
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    hh = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((hh.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb(hsv: np.ndarray) -> np.ndarray:
    h, s, vv = hsv[..., 0].astype(np.float32), hsv[..., 1], hsv[..., 2]
    hd = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int32)
    f = (hd - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vd = vv.astype(np.float32).astype(np.float64)
    p, q, t = (np.clip(np.floor(vd * e + 0.5), 0, 255).astype(np.uint8) for e in (1.0 - fs, 1.0 - fs * f, 1.0 - fs * (1.0 - f)))
    k = i % 6
    out = np.stack([np.choose(k, [vv, q, p, p, t, vv]), np.choose(k, [t, vv, vv, q, p, p]), np.choose(k, [p, p, t, vv, vv, q])], -1)
    gray = s == 0
    out[gray] = np.stack([vv, vv, vv], -1)[gray]
    return out


def hue_delta(delta: float) -> int:
    """data._shift_hue: the shift added to the 8-bit hue, modulo 256"""
    return int(round(delta * 255)) % 256


def shift_hue(img: np.ndarray, delta: float) -> np.ndarray:
    hsv = rgb2hsv(img)
    hsv[..., 0] = ((hsv[..., 0].astype(np.int16) + int(round(delta * 255))) % 256).astype(np.uint8)
    return hsv2rgb(hsv)


def rotate_plan(w: int, h: int, angle: float) -> tuple[int, tuple[int, ...]]:
    """(mode, six 16.16 coefficients): mode 0 copy, 1 affine, 2 rotate 180, 3 / 4 rotate 90 / 270 (square images only) — the cases
    Image.rotate distinguishes for expand=False, no centre, no translation"""
    angle = angle % 360.0
    if angle == 0:
        return 0, (0,) * 6
    if angle == 180:
        return 2, (0,) * 6
    if angle in (90, 270) and w == h:
        return (3 if angle == 90 else 4), (0,) * 6
    cx, cy = w / 2.0, h / 2.0
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v: float) -> int:
        return int(math.floor(v * 65536.0 + 0.5))

    return 1, (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate(img: np.ndarray, angle: float) -> np.ndarray:
    h, w = img.shape[:2]
    mode, (a0, a1, a2, a3, a4, a5) = rotate_plan(w, h, angle)
    if mode == 0:
        return img.copy()
    if mode == 2:
        return img[::-1, ::-1].copy()
    if mode in (3, 4):
        return np.rot90(img, 1 if mode == 3 else 3).copy()
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xin, yin = (a2 + ys * a1 + xs * a0) >> 16, (a5 + ys * a4 + xs * a3) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def jitter(img: np.ndarray, order, fb, fc, fs, dh) -> np.ndarray:
    """data.ColorJitter.__call__ with its random draws given: order = the permutation of (0 brightness, 1 contrast, 2 saturation,
    3 hue); a factor of None skips that operation (strength 0 in the transform)"""
    for which in order:
        if which == 0 and fb is not None:
            img = brightness(img, fb)
        elif which == 1 and fc is not None:
            img = contrast(img, fc)
        elif which == 2 and fs is not None:
            img = color(img, fs)
        elif which == 3 and dh is not None:
            img = shift_hue(img, dh)
    return img
