"""ImageFolder dataset and the image transforms the reference's loaders use.

The reference builds its pipelines from torchvision (trainers/efficientnet.py:111-234,
orchestration/orchestrator.py:316-347,380-395); torchvision is not part of this stack, so
the same operations are provided here on PIL images / torch tensors, with torch's RNG as
the source of randomness (so `apply_seed` governs augmentation):

  Lambda, Resize(shorter side, bilinear), CenterCrop, RandomCrop,
  RandomResizedCrop(size, scale, ratio), RandomRotation(degrees), RandomHorizontalFlip(p),
  ColorJitter(brightness, contrast, saturation, hue), ToTensor, Normalize(mean, std),
  RandomErasing(p, scale, ratio, value), Compose.

Batches leave the loaders as float32 [N,3,H,W] + int64 [N] exactly like the reference's.
"""

from __future__ import annotations

import math
import os
from collections.abc import Callable, Sequence
from pathlib import Path
from typing import Any

import numpy as np
import torch
from PIL import Image, ImageEnhance

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def _rand() -> float:
    return float(torch.rand(1).item())


def _uniform(lo: float, hi: float) -> float:
    return float(torch.empty(1).uniform_(lo, hi).item())


class Compose:
    def __init__(self, ops: Sequence[Callable]) -> None:
        self.ops = list(ops)

    def __call__(self, img: Any) -> Any:
        for op in self.ops:
            img = op(img)
        return img


class Lambda:
    def __init__(self, fn: Callable) -> None:
        self.fn = fn

    def __call__(self, img: Any) -> Any:
        return self.fn(img)


class Resize:
    """Shorter side -> `size` (int) keeping the aspect ratio, or exact (h, w); bilinear."""

    def __init__(self, size: int | tuple[int, int], interpolation: int = Image.BILINEAR) -> None:
        self.size, self.interpolation = size, interpolation

    def __call__(self, img: Image.Image) -> Image.Image:
        w, h = img.size
        if isinstance(self.size, int):
            short, long = (w, h) if w <= h else (h, w)
            if short == self.size:
                return img
            new_short, new_long = self.size, int(self.size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = self.size
        return img.resize((nw, nh), self.interpolation)


def _pad_to(img: Image.Image, th: int, tw: int) -> Image.Image:
    w, h = img.size
    if w >= tw and h >= th:
        return img
    canvas = Image.new(img.mode, (max(w, tw), max(h, th)))
    canvas.paste(img, ((max(w, tw) - w) // 2, (max(h, th) - h) // 2))
    return canvas


class CenterCrop:
    def __init__(self, size: int | tuple[int, int]) -> None:
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img: Image.Image) -> Image.Image:
        th, tw = self.size
        img = _pad_to(img, th, tw)
        w, h = img.size
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return img.crop((left, top, left + tw, top + th))


class RandomCrop:
    def __init__(self, size: int | tuple[int, int]) -> None:
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img: Image.Image) -> Image.Image:
        th, tw = self.size
        img = _pad_to(img, th, tw)
        w, h = img.size
        top = int(torch.randint(0, h - th + 1, (1,)).item())
        left = int(torch.randint(0, w - tw + 1, (1,)).item())
        return img.crop((left, top, left + tw, top + th))


class RandomResizedCrop:
    """Random area fraction in `scale`, log-uniform aspect in `ratio`, resized to size x size."""

    def __init__(self, size: int, scale: tuple[float, float] = (0.08, 1.0), ratio: tuple[float, float] = (3 / 4, 4 / 3),
                 interpolation: int = Image.BILINEAR) -> None:
        self.size, self.scale, self.ratio, self.interpolation = size, scale, ratio, interpolation

    def _box(self, w: int, h: int) -> tuple[int, int, int, int]:
        area = w * h
        log_lo, log_hi = math.log(self.ratio[0]), math.log(self.ratio[1])
        for _ in range(10):
            target = area * _uniform(*self.scale)
            aspect = math.exp(_uniform(log_lo, log_hi))
            cw, ch = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if 0 < cw <= w and 0 < ch <= h:
                top = int(torch.randint(0, h - ch + 1, (1,)).item())
                left = int(torch.randint(0, w - cw + 1, (1,)).item())
                return left, top, cw, ch
        in_ratio = w / h
        if in_ratio < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        return (w - cw) // 2, (h - ch) // 2, cw, ch

    def __call__(self, img: Image.Image) -> Image.Image:
        left, top, cw, ch = self._box(*img.size)
        return img.crop((left, top, left + cw, top + ch)).resize((self.size, self.size), self.interpolation)


class RandomRotation:
    def __init__(self, degrees: float) -> None:
        self.degrees = float(degrees)

    def __call__(self, img: Image.Image) -> Image.Image:
        return img.rotate(_uniform(-self.degrees, self.degrees), resample=Image.NEAREST, expand=False)


class RandomHorizontalFlip:
    def __init__(self, p: float = 0.5) -> None:
        self.p = p

    def __call__(self, img: Image.Image) -> Image.Image:
        return img.transpose(Image.FLIP_LEFT_RIGHT) if _rand() < self.p else img


def _shift_hue(img: Image.Image, delta: float) -> Image.Image:
    if img.mode not in ("RGB", "L"):
        return img
    if img.mode == "L":
        return img
    hsv = np.array(img.convert("HSV"), dtype=np.uint8)
    hsv[..., 0] = (hsv[..., 0].astype(np.int16) + int(round(delta * 255))) % 256
    return Image.fromarray(hsv, "HSV").convert("RGB")


class ColorJitter:
    def __init__(self, brightness: float = 0.0, contrast: float = 0.0, saturation: float = 0.0, hue: float = 0.0) -> None:
        self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

    def __call__(self, img: Image.Image) -> Image.Image:
        order = torch.randperm(4).tolist()
        for which in order:
            if which == 0 and self.brightness > 0:
                img = ImageEnhance.Brightness(img).enhance(_uniform(max(0.0, 1 - self.brightness), 1 + self.brightness))
            elif which == 1 and self.contrast > 0:
                img = ImageEnhance.Contrast(img).enhance(_uniform(max(0.0, 1 - self.contrast), 1 + self.contrast))
            elif which == 2 and self.saturation > 0:
                img = ImageEnhance.Color(img).enhance(_uniform(max(0.0, 1 - self.saturation), 1 + self.saturation))
            elif which == 3 and self.hue > 0:
                img = _shift_hue(img, _uniform(-self.hue, self.hue))
        return img


class ToTensor:
    """PIL image (uint8) -> float32 CHW in [0, 1]."""

    def __call__(self, img: Image.Image) -> torch.Tensor:
        arr = np.asarray(img, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).to(torch.float32).div_(255.0)


class Normalize:
    def __init__(self, mean: Sequence[float], std: Sequence[float]) -> None:
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        return (t - self.mean) / self.std


class RandomErasing:
    def __init__(self, p: float = 0.5, scale: tuple[float, float] = (0.02, 0.33), ratio: tuple[float, float] = (0.3, 3.3),
                 value: float = 0.0) -> None:
        self.p, self.scale, self.ratio, self.value = p, scale, ratio, value

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        if _rand() >= self.p:
            return t
        _, h, w = t.shape
        area = h * w
        log_lo, log_hi = math.log(self.ratio[0]), math.log(self.ratio[1])
        for _ in range(10):
            target = area * _uniform(*self.scale)
            aspect = math.exp(_uniform(log_lo, log_hi))
            eh, ew = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if eh < h and ew < w:
                top = int(torch.randint(0, h - eh + 1, (1,)).item())
                left = int(torch.randint(0, w - ew + 1, (1,)).item())
                t = t.clone()
                t[:, top:top + eh, left:left + ew] = self.value
                return t
        return t

class ToUint8HWC:
    """PIL image -> uint8 [H, W, 3] tensor: what the loader ships when the tail of the pipeline
    (flip, to-float, normalise, erasing) runs on the GPU (`GpuInputTail`)."""

    def __call__(self, img: Image.Image) -> torch.Tensor:
        return torch.from_numpy(np.array(img.convert("RGB"), dtype=np.uint8))       # np.array: a writable copy


# ---------------------------------------------------------------------------------------------------------------------
# Resize / crop on the device (csrc/dfd_resize.hip).  The worker decodes the file and PLANS the
# geometric transform — same decisions, same RNG calls as the PIL transforms above — but leaves the pixels alone; the batch
# crosses PCIe as the decoded uint8 images packed back to back plus one 56-byte descriptor per image, and ONE kernel resizes
# and crops every image, bit-exact with Pillow's bilinear `Image.resize`.
_JOB_DTYPE = np.dtype([("offset", "<i8"), ("H", "<i4"), ("W", "<i4"), ("bx", "<i4"), ("by", "<i4"), ("bw", "<i4"), ("bh", "<i4"),
                       ("rw", "<i4"), ("rh", "<i4"), ("cx", "<i4"), ("cy", "<i4"), ("_p0", "<i4"), ("_p1", "<i4")])
assert _JOB_DTYPE.itemsize == 56            # struct dfd_resize_job (include/dfd_hip.h)


def plan_resize(w: int, h: int, size: int | tuple[int, int] | None) -> tuple[int, int]:
    """(new width, new height) of `Resize(size)` applied to a w x h image (None: unchanged)."""
    if size is None:
        return w, h
    if isinstance(size, int):
        short, long = (w, h) if w <= h else (h, w)
        if short == size:
            return w, h
        new_short, new_long = size, int(size * long / short)
        return (new_short, new_long) if w <= h else (new_long, new_short)
    return size[1], size[0]


def plan_window(rw: int, rh: int, th: int, tw: int, *, random: bool) -> tuple[int, int]:
    """Origin (cx, cy) of the th x tw crop window in a rw x rh image after `_pad_to` — CenterCrop (random=False) or
    RandomCrop; negative when the image had to be padded (the kernel writes zeros there, as the black canvas does)."""
    pw, ph = max(rw, tw), max(rh, th)
    pad_left, pad_top = (pw - rw) // 2, (ph - rh) // 2
    if random:
        top = int(torch.randint(0, ph - th + 1, (1,)).item())
        left = int(torch.randint(0, pw - tw + 1, (1,)).item())
    else:
        top, left = int(round((ph - th) / 2.0)), int(round((pw - tw) / 2.0))
    return left - pad_left, top - pad_top


class PlanGeometry:
    """Stand-in for [Resize] -> [CenterCrop | RandomCrop] or RandomResizedCrop at the end of a worker pipeline: returns
    (uint8 [H, W, 3] tensor of the DECODED image, int64 plan [bx, by, bw, bh, rw, rh, cx, cy, OH, OW]).
    mode: "center" (Resize(resize) + CenterCrop(out)), "random" (Resize(resize) + RandomCrop(out)), "rrc" (RandomResizedCrop)."""

    def __init__(self, mode: str, out: int, resize: int | None = None, rrc: "RandomResizedCrop | None" = None) -> None:
        if mode not in ("center", "random", "rrc") or (mode == "rrc") != (rrc is not None):
            raise ValueError("PlanGeometry: mode is center | random | rrc (the last one with its RandomResizedCrop)")
        self.mode, self.out, self.resize, self.rrc = mode, out, resize, rrc

    # dfd_resize_crop_u8 holds a filter of at most 96 taps per axis: shrink factors up to ~46.  An image beyond that (a 12,000-pixel
    # scan fed to a 224-pixel model) is resized HERE, by the very PIL calls the device kernel restates bit for bit, and travels with
    # an identity plan — one oversize image no longer aborts the epoch with DFD_EUNSUPPORTED.
    MAX_DEVICE_SHRINK = 40
    # Pillow 12 runs its VERTICAL pass first for some very long, narrow images (measured: 64 x 8000 -> 48 x 6000 vertical-first,
    # 64 x 6000 and 100 x 9000 horizontal-first; the rule is not in its documentation), and the 8-bit intermediate makes the two
    # orders differ by a unit in ~20 % of the bytes.  The device kernel is horizontal-first; images beyond this side length take the
    # PIL calls in the worker too, so the pipeline stays bit-exact with PIL whatever that rule is.
    MAX_DEVICE_SIDE = 4096

    def __call__(self, img: Image.Image):
        img = img.convert("RGB")
        w, h = img.size
        if self.mode == "rrc":
            bx, by, bw, bh = self.rrc._box(w, h)
            plan = (bx, by, bw, bh, self.out, self.out, 0, 0)
            if max(-(-bw // self.out), -(-bh // self.out)) > self.MAX_DEVICE_SHRINK or max(bw, bh) > self.MAX_DEVICE_SIDE:
                img = img.crop((bx, by, bx + bw, by + bh)).resize((self.out, self.out), Image.BILINEAR)
                plan = (0, 0, self.out, self.out, self.out, self.out, 0, 0)
        else:
            rw, rh = plan_resize(w, h, self.resize)
            cx, cy = plan_window(rw, rh, self.out, self.out, random=self.mode == "random")
            plan = (0, 0, w, h, rw, rh, cx, cy)
            if max(-(-w // rw), -(-h // rh)) > self.MAX_DEVICE_SHRINK or max(w, h) > self.MAX_DEVICE_SIDE:
                img = img.resize((rw, rh), Image.BILINEAR)
                plan = (0, 0, rw, rh, rw, rh, cx, cy)
        return torch.from_numpy(np.array(img, dtype=np.uint8)), torch.tensor([*plan, self.out, self.out], dtype=torch.int64)


def collate_raw(batch):
    """[((uint8 HWC, plan), label), ...] -> ((flat uint8 bytes, job bytes, meta int64 [OH, OW, max_shrink]), labels)."""
    imgs = [item[0][0] for item in batch]
    plans = torch.stack([item[0][1] for item in batch]).numpy()
    labels = torch.tensor([item[1] for item in batch], dtype=torch.int64)
    jobs = np.zeros(len(batch), dtype=_JOB_DTYPE)
    at, shrink = 0, 1
    for i, (im, pl) in enumerate(zip(imgs, plans)):
        H, W = int(im.shape[0]), int(im.shape[1])
        bx, by, bw, bh, rw, rh, cx, cy = (int(v) for v in pl[:8])
        jobs[i] = (at, H, W, bx, by, bw, bh, rw, rh, cx, cy, 0, 0)
        at += H * W * 3
        shrink = max(shrink, -(-bw // rw), -(-bh // rh))
    flat = torch.cat([im.reshape(-1) for im in imgs])
    meta = torch.tensor([int(plans[0][8]), int(plans[0][9]), shrink], dtype=torch.int64)
    return (flat, torch.from_numpy(jobs.view(np.uint8).copy()), meta), labels


AUGMENT_MAX_BYTES = 156 * 1024          # dfd_augment_u8 keeps one picture in a CU's LDS (csrc/dfd_augment.hip)


def rotate_plan(w: int, h: int, angle: float) -> tuple[int, tuple[int, ...]]:
    """What Image.rotate(angle, NEAREST, expand=False) will do to a w x h picture, for the device kernel: (mode, coefficients) with
    mode 0 copy, 1 affine (six 16.16 fixed-point coefficients, formed as Image.rotate + Geometry.c affine_fixed form them), 2 the
    180-degree flip, 3 / 4 the 90 / 270-degree transposes Pillow takes for square pictures."""
    angle = angle % 360.0
    if angle == 0:
        return 0, (0,) * 6
    if angle == 180:
        return 2, (0,) * 6
    if angle in (90, 270) and w == h:
        return (3 if angle == 90 else 4), (0,) * 6
    cx, cy = w / 2.0, h / 2.0
    rad = -math.radians(angle)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v: float) -> int:
        return int(math.floor(v * 65536.0 + 0.5))

    return 1, (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


class GpuInputTail:
    """[RandomRotation] -> RandomHorizontalFlip -> [ColorJitter] -> ToTensor -> Normalize -> RandomErasing(value=0) on the device, for a
    uint8 NHWC batch.  The random decisions use the same distributions and the same
    host RNG calls as the CPU transforms above, one image at a time, in the pipeline's order; the arithmetic is the kernels
    dfd_augment_u8 (rotation + colour jitter, byte-exact with Pillow) and dfd_image_prep (bit-identical to ToTensor + Normalize;
    the flip commutes with the per-pixel colour operations and with the mean Contrast takes, so it stays in this last kernel).
    4x fewer bytes cross PCIe than with f32 batches, and the worker processes skip every pixel pass but the decode."""

    def __init__(self, mean: Sequence[float], std: Sequence[float], flip_p: float = 0.0, erase_p: float = 0.0,
                 erase_scale: tuple[float, float] = (0.02, 0.33), erase_ratio: tuple[float, float] = (0.3, 3.3),
                 rotate_degrees: float = 0.0, jitter: Sequence[float] | None = None) -> None:
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.flip_p, self.erase_p, self.erase_scale, self.erase_ratio = flip_p, erase_p, erase_scale, erase_ratio
        self.rotate_degrees = float(rotate_degrees)
        self.jitter = tuple(float(v) for v in jitter) if jitter is not None and any(float(v) > 0 for v in jitter) else None

    @property
    def augments(self) -> bool:
        return self.rotate_degrees > 0 or self.jitter is not None

    def sample_augment(self, n: int, h: int, w: int) -> torch.Tensor:
        """One dfd_augment_job (16 int32) per picture: RandomRotation's angle and ColorJitter's permutation + factors, drawn as
        data.RandomRotation / data.ColorJitter draw them (same calls, same order: angle; permutation; one uniform per enabled
        operation in the permuted order)."""
        jobs = np.zeros((n, 16), dtype=np.int32)
        fl = jobs.view(np.float32)
        for i in range(n):
            if self.rotate_degrees > 0:
                mode, coef = rotate_plan(w, h, _uniform(-self.rotate_degrees, self.rotate_degrees))
                jobs[i, 0] = mode
                jobs[i, 1:7] = coef
            jobs[i, 7:11] = -1
            if self.jitter is not None:
                b, c, s, hue = self.jitter
                order = torch.randperm(4).tolist()
                jobs[i, 7:11] = order
                for which in order:
                    if which == 0 and b > 0:
                        fl[i, 11] = _uniform(max(0.0, 1 - b), 1 + b)
                    elif which == 1 and c > 0:
                        fl[i, 12] = _uniform(max(0.0, 1 - c), 1 + c)
                    elif which == 2 and s > 0:
                        fl[i, 13] = _uniform(max(0.0, 1 - s), 1 + s)
                    elif which == 3 and hue > 0:
                        jobs[i, 14] = int(round(_uniform(-hue, hue) * 255)) % 256
                jobs[i, 15] = (1 if b > 0 else 0) | (2 if c > 0 else 0) | (4 if s > 0 else 0) | (8 if hue > 0 else 0)
        return torch.from_numpy(jobs)

    def sample(self, n: int, h: int, w: int) -> tuple[torch.Tensor | None, torch.Tensor | None]:
        flip = erase = None
        if self.flip_p > 0:
            flip = torch.tensor([1 if _rand() < self.flip_p else 0 for _ in range(n)], dtype=torch.uint8)
        if self.erase_p > 0:
            boxes = torch.zeros((n, 4), dtype=torch.int32)
            log_lo, log_hi = math.log(self.erase_ratio[0]), math.log(self.erase_ratio[1])
            for i in range(n):
                if _rand() >= self.erase_p:
                    continue
                for _ in range(10):
                    target = h * w * _uniform(*self.erase_scale)
                    aspect = math.exp(_uniform(log_lo, log_hi))
                    eh, ew = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
                    if eh < h and ew < w:
                        top = int(torch.randint(0, h - eh + 1, (1,)).item())
                        left = int(torch.randint(0, w - ew + 1, (1,)).item())
                        boxes[i] = torch.tensor([top, left, eh, ew], dtype=torch.int32)
                        break
            erase = boxes
        return flip, erase

    def __call__(self, batch_u8: torch.Tensor, device) -> torch.Tensor:
        from . import kernels as K

        if isinstance(batch_u8, (tuple, list)):          # collate_raw: decoded images + plans -> resize / crop on the device
            flat, jobs, meta = batch_u8
            oh, ow, shrink = (int(v) for v in meta)
            n = jobs.numel() // _JOB_DTYPE.itemsize
            aug = self.sample_augment(n, oh, ow) if self.augments else None
            flip, erase = self.sample(n, oh, ow)
            dev = K.resize_crop_u8(flat.to(device, non_blocking=True), jobs.to(device, non_blocking=True), n, oh, ow, shrink)
            if aug is not None:
                dev = K.augment_u8(dev, aug.to(device, non_blocking=True))
            return K.image_prep(dev, self.mean, self.std,
                                flip.to(device, non_blocking=True) if flip is not None else None,
                                erase.to(device, non_blocking=True) if erase is not None else None)
        if batch_u8.dim() != 4 or batch_u8.shape[3] != 3 or batch_u8.dtype != torch.uint8:
            raise ValueError("GpuInputTail expects a uint8 [N, H, W, 3] batch (ToUint8HWC at the end of the CPU pipeline)")
        n, h, w, _ = batch_u8.shape
        aug = self.sample_augment(n, h, w) if self.augments else None
        flip, erase = self.sample(n, h, w)
        dev = batch_u8.to(device, non_blocking=True).contiguous()
        if aug is not None:
            dev = K.augment_u8(dev, aug.to(device, non_blocking=True))
        return K.image_prep(dev, self.mean, self.std,
                            flip.to(device, non_blocking=True) if flip is not None else None,
                            erase.to(device, non_blocking=True) if erase is not None else None)



def pil_rgb_loader(path: str | os.PathLike) -> Image.Image:
    with open(path, "rb") as handle:
        return Image.open(handle).convert("RGB")


class ImageFolder(torch.utils.data.Dataset):
    """root/<class>/<image> layout; classes are the sorted sub-directory names."""

    def __init__(self, root: str | os.PathLike, transform: Callable | None = None,
                 loader: Callable[[str], Any] = pil_rgb_loader) -> None:
        self.root = Path(root)
        self.transform, self.loader = transform, loader
        self.classes = sorted(entry.name for entry in os.scandir(self.root) if entry.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {self.root}.")
        self.class_to_idx = {name: i for i, name in enumerate(self.classes)}
        self.samples: list[tuple[str, int]] = []
        for name in self.classes:
            for folder, _, files in sorted(os.walk(self.root / name, followlinks=True)):
                for fname in sorted(files):
                    if fname.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(folder, fname), self.class_to_idx[name]))
        self.targets = [target for _, target in self.samples]
        self.imgs = self.samples

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, index: int):
        path, target = self.samples[index]
        sample = self.loader(path)
        if self.transform is not None:
            sample = self.transform(sample)
        return sample, target


__all__ = [
    "CenterCrop", "ColorJitter", "Compose", "ImageFolder", "Lambda", "Normalize", "RandomCrop", "RandomErasing",
    "RandomHorizontalFlip", "RandomResizedCrop", "RandomRotation", "Resize", "ToTensor", "pil_rgb_loader",
]
