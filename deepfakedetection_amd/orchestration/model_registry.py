"""Model registry: name -> (trainer module, builder, default resolution).

Same surface as the reference's orchestration/model_registry.py (`ModelSpec` :21-29,
`get_model_spec` :78-98: exact names first, then `startswith` prefixes with the spec
re-labelled to the requested name, unknown -> KeyError with the reference's message), but
the builders return the HIP-backed modules of this package:

  efficientnet_b3   efficientnet_pytorch-compatible B3 (state-dict keys, static SAME padding) —
                    the reference's own key (model_registry.py:50-58)
  efficientnet_b0   timm-compatible B0 — the BASELINE.json configuration (new key)
  efficientformer*  timm-compatible EfficientFormerV2-{S0,S1,S2,L} on the HIP kernels (the reference's
                    prefix entry, model_registry.py:60-66); the builder takes an optional img_size like
                    timm.create_model does in trainers/efficientformer_v2.py:327
  faster_vit*       fastervit-compatible FasterViT-{0,1,2,3} at 224 px on the HIP kernels (the reference's prefix
                    entry, model_registry.py:67-75; it builds faster_vit_2_224 and swaps the head, :43-47).
Any other name under these prefixes (e.g. faster_vit_4_21k_224) raises a loud NotImplementedError from the builder:
there is no ATen fallback.
"""

from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass, replace

from torch import nn

_EFFNET_TRAINER = "deepfakedetection_amd.trainers.efficientnet"


@dataclass(frozen=True)
class ModelSpec:
    name: str
    train_module: str
    weights_key: str
    default_image_size: int
    builder: Callable[[str, int], nn.Module]


def _effnet(variant: str, flavour: str) -> Callable[[str, int], nn.Module]:
    def build(_: str, num_classes: int) -> nn.Module:
        from ..efficientnet import HipEfficientNet  # deferred: importing the registry must not need the GPU library

        return HipEfficientNet(variant, flavour, num_classes)

    return build


def _efformer(model_name: str, num_classes: int, img_size: int = 224) -> nn.Module:
    from ..efficientformer_v2 import build_efficientformer_v2

    return build_efficientformer_v2(model_name, num_classes, img_size)


def _fastervit(model_name: str, num_classes: int) -> nn.Module:
    from ..fastervit import build_fastervit

    try:
        return build_fastervit(model_name, num_classes)
    except KeyError as exc:
        raise NotImplementedError(
            f"FasterViT ('{model_name}') is registered but its MI355X engine is not built: only faster_vit_0/1/2/3_224 "
            "run on the HIP kernels (there is no ATen fallback).") from exc


_exact: dict[str, ModelSpec] = {
    "efficientnet_b3": ModelSpec("efficientnet_b3", _EFFNET_TRAINER, "efficientnet_b3", 224, _effnet("b3", "lukemelas")),
    "efficientnet_b0": ModelSpec("efficientnet_b0", _EFFNET_TRAINER, "efficientnet_b0", 224, _effnet("b0", "timm")),
}

_by_prefix: dict[str, ModelSpec] = {
    "efficientformer": ModelSpec("efficientformerv2_s1", "deepfakedetection_amd.trainers.efficientformer_v2",
                                 "efficientformerv2_s1", 224, _efformer),
    "faster_vit": ModelSpec("faster_vit_2_224", "deepfakedetection_amd.trainers.fastervit", "faster_vit_2_224", 224, _fastervit),
}


def register_model_spec(spec: ModelSpec, *, prefix: bool = False) -> None:
    """Add or replace a registry entry (plug-ins, tests)."""
    (_by_prefix if prefix else _exact)[spec.name] = spec


def get_model_spec(model_name: str) -> ModelSpec:
    spec = _exact.get(model_name)
    if spec is not None:
        return spec
    for prefix, template in _by_prefix.items():
        if model_name.startswith(prefix):
            return replace(template, name=model_name, weights_key=model_name)
    raise KeyError(f"Unknown model '{model_name}'. Add it to model_registry.py.")


__all__ = ["ModelSpec", "get_model_spec", "register_model_spec"]
