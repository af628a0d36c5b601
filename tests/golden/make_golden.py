"""Writes the committed golden fixtures.  Run from the repo root IN THE BUILD CONTAINER
(it reads /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

Outputs
  tests/golden/reference_contract.json  — values produced by IMPORTING the reference's own
      orchestration/config_schema.py and orchestration/train_env.py (the two modules of the
      reference that import here) on the reference's shipped YAML files: the plug-in
      surface the host side must reproduce bit for bit.
  tests/golden/effnet_logits.json — logits of the CPU oracle (oracle/effnet_ref.py) on
      seeded inputs; the oracle itself is pinned by tests/test_oracle.py.
  tests/golden/vit_logits.json — the same for oracle/efformer_ref.py and oracle/fastervit_ref.py
      (drift guards; their pinning is in tests/test_efformer_oracle.py / test_fastervit_oracle.py).
Only data is written: inputs and expected outputs, no reference source text.
"""

from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def reference_contract() -> dict:
    sys.path.insert(0, str(REF))
    from orchestration import train_env as ref_env                      # noqa: PLC0415
    from orchestration.config_schema import OrchestratorConfig         # noqa: PLC0415
    from pydantic import ValidationError                               # noqa: PLC0415

    out: dict = {"configs": {}, "config_inputs": {}, "validation_errors": {}, "env": {}}
    for name in ("train.yaml", "train_imagenette.yaml", "inference.yaml"):
        raw = yaml.safe_load((REF / "config" / name).read_text())
        out["config_inputs"][name] = raw                      # the parsed YAML (data), fed to our schema in the test
        out["configs"][name] = OrchestratorConfig(**raw).model_dump()
    base = {"data": {"root": "d"}, "models": {"a": {}}}
    cases = {
        "empty_models": {"data": {"root": "d"}, "models": {}},
        "unknown_selection": {**base, "selection": ["b"]},
        "missing_data": {"models": {"a": {}}},
    }
    for key, raw in cases.items():
        try:
            OrchestratorConfig(**raw)
            out["validation_errors"][key] = None
        except ValidationError as exc:
            out["validation_errors"][key] = [e["msg"] for e in exc.errors()]
    out["defaults"] = OrchestratorConfig(**base).model_dump()
    out["validation_inputs"] = cases
    extra = {"data": {"root": "d", "bogus": 1}, "models": {"a": {"training": {"lr": 0.5, "x": [1]}, "zzz": 3}}, "top_extra": True}
    out["extras_input"] = extra
    out["extras_output"] = OrchestratorConfig(**extra).model_dump()

    env = out["env"]
    os.environ["TRANSFORMS"] = '{"train_to_tensor":false,"train_color_jitter":"yes","x":0}'
    env["toggles"] = ref_env.load_transform_toggles({"train_to_tensor": True, "train_color_jitter": False},
                                                    required=("train_to_tensor",))
    os.environ["TRANSFORMS"] = "not json"
    env["toggles_bad_json"] = ref_env.load_transform_toggles({"a": True, "b": False})
    del os.environ["TRANSFORMS"]
    for value in ("abc", "3", " 4", "5.0"):
        os.environ["EPOCHS"] = value
        env[f"env_int[{value}]"] = ref_env.env_int("EPOCHS", 7)
        env[f"env_float[{value}]"] = ref_env.env_float("EPOCHS", 7.5)
    del os.environ["EPOCHS"]
    env["env_int[unset]"] = ref_env.env_int("EPOCHS", 7)
    env["env_str[unset]"] = ref_env.env_str("TRAIN_SPLIT", "Train")
    env["as_bool"] = {repr(v): ref_env._as_bool(v) for v in (True, False, 1, 0, 2.5, "1", "true", " YES ", "on", "off", "", None, [1])}
    env["__all__"] = sorted(ref_env.__all__)

    class FakeDataset:
        classes = ["a", "b", "c", "d", "e", "f", "g"]

    try:
        ref_env.require_num_classes(FakeDataset(), 2, split="train", dataset_root="/data/x")
        env["require_num_classes_error"] = None
    except ValueError as exc:
        env["require_num_classes_error"] = str(exc)
    try:
        ref_env.require_num_classes(FakeDataset(), 0, split="train")
    except ValueError as exc:
        env["require_num_classes_nonpositive"] = str(exc)
    return out


def oracle_logits() -> dict:
    from oracle.effnet_ref import EfficientNetRef  # noqa: PLC0415

    cases = []
    for variant, flavour, classes, size, batch in (("b0", "timm", 2, 224, 2), ("b3", "lukemelas", 2, 224, 2),
                                                   ("b0", "timm", 10, 160, 2), ("b3", "lukemelas", 10, 160, 1)):
        seed, input_seed = 11, 1
        torch.manual_seed(seed)
        model = EfficientNetRef(variant, flavour, classes).eval()
        x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(input_seed))
        with torch.no_grad():
            logits = model(x)
        cases.append(dict(variant=variant, flavour=flavour, classes=classes, size=size, batch=batch, seed=seed,
                          input_seed=input_seed, logits=logits.tolist()))
    return {"generator": "tests/golden/make_golden.py", "torch": torch.__version__, "cases": cases}


def vit_logits() -> dict:
    from oracle.efformer_ref import EfficientFormerV2Ref  # noqa: PLC0415
    from oracle.fastervit_ref import FasterViTRef  # noqa: PLC0415

    cases = []
    for family, variant, classes, size, batch in (("efficientformerv2", "s1", 2, 224, 2), ("efficientformerv2", "s0", 10, 96, 2),
                                                  ("fastervit", "0", 2, 224, 2), ("fastervit", "1", 10, 224, 1)):
        seed, input_seed = 13, 2
        torch.manual_seed(seed)
        model = (EfficientFormerV2Ref(variant, classes, img_size=size) if family == "efficientformerv2"
                 else FasterViTRef(variant, classes, resolution=size)).eval()
        x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(input_seed))
        with torch.no_grad():
            logits = model(x)
        cases.append(dict(family=family, variant=variant, classes=classes, size=size, batch=batch, seed=seed,
                          input_seed=input_seed, logits=logits.tolist()))
    return {"generator": "tests/golden/make_golden.py", "torch": torch.__version__, "cases": cases}


if __name__ == "__main__":
    (OUT / "reference_contract.json").write_text(json.dumps(reference_contract(), indent=1, sort_keys=True))
    (OUT / "effnet_logits.json").write_text(json.dumps(oracle_logits(), indent=1))
    (OUT / "vit_logits.json").write_text(json.dumps(vit_logits(), indent=1))
    print("wrote", sorted(p.name for p in OUT.glob("*.json")))
