"""The drop-in surface on the real device: `orchestrate(train yaml)` then
`orchestrate(inference yaml)` with the HIP-backed EfficientNet-B0 / -B3 on cuda, tiny
generated ImageFolder.  Everything between the YAML and the kernels is the product path."""

from __future__ import annotations

import json
from pathlib import Path

import pytest
import torch
import yaml

from tests.test_plumbing_cpu import _make_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_name,weights_file,gpu_tail", [("efficientnet_b0", "EfficientNetModel.pth", False),
                                                             ("efficientnet_b3", "EfficientNetModel.pth", False),
                                                             ("efficientnet_b0", "EfficientNetModel.pth", True),
                                                             ("efficientformerv2_s1", "EfficientFormerV2_S1.pth", False),
                                                             ("faster_vit_0_224", "FasterVitModel.pth", False)])
def test_orchestrated_training_and_inference_on_gpu(tmp_path, monkeypatch, model_name, weights_file, gpu_tail):
    from deepfakedetection_amd.orchestration.orchestrator import orchestrate

    monkeypatch.chdir(tmp_path)
    former = model_name.startswith("efficientformer")
    fvit = model_name.startswith("faster_vit")
    img = 224 if fvit else (128 if former else 64)   # EfficientFormerV2 at 128 px: 16-token attention; FasterViT: 7x7 windows need 224
    _make_dataset(tmp_path / "data", classes=("fake", "real"), per_class=8, size=img + 8)
    base = {
        "seed": 1, "device": "cuda",
        "data": {"root": str(tmp_path / "data"), "train_split": "train", "val_split": "val", "test_split": "test",
                 "num_classes": 2, "img_size": img},
    }
    out_dir = str(tmp_path / "runs" / model_name)
    train_cfg = {**base, "models": {model_name: {"output_dir": out_dir, "training": {
        "epochs": 1, "batch_size": 8, "ft_batch_size": 8, "accum_steps": 2, "num_workers": 0, "resume": "auto", "pretrained": False,
        "gpu_input_tail": gpu_tail}}}}       # True: loaders ship uint8, flip / normalise / erasing run in dfd_image_prep
    path = tmp_path / "train.yaml"
    path.write_text(yaml.safe_dump(train_cfg))
    orchestrate(path, mode="training")
    run = sorted(Path(out_dir).iterdir())[0]
    ckpt = torch.load(run / "checkpoints" / "latest.ckpt", map_location="cpu")
    assert ckpt["epoch"] == 1 and set(ckpt["optimizer"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    head_key = "head_dist.weight" if former else ("head.weight" if fvit else ("_fc.weight" if model_name == "efficientnet_b3" else "classifier.weight"))
    assert head_key in ckpt["model"]
    log = (run / "logs" / "train.log").read_text()
    assert "Warmup (head only)" in log and "val_acc=" in log
    if former:
        # fine-tuning trains the UNFREEZE_KEYS subset only (trainers/efficientformer_v2.py:389-393): 182 tensors
        assert len(ckpt["optimizer"]["state"]) == 182
        rows = [json.loads(line) for line in (run / "logs" / "throughput.jsonl").read_text().splitlines()]
        assert [r["phase"] for r in rows] == ["warmup", "fine-tune"] and all(r["images_per_sec"] > 0 for r in rows)

    infer_cfg = {**base, "models": {model_name: {"output_dir": out_dir, "inference": {
        # best weights exist only if the fine-tune epoch beat the warm-up (reference behaviour);
        # latest.ckpt always exists and load_model unwraps its "model" entry (orchestrator.py:370-374)
        "weights": str(run / "checkpoints" / "latest.ckpt"), "split": "test", "batch_size": 16, "num_workers": 0, "img_size": img}}}}
    path2 = tmp_path / "infer.yaml"
    path2.write_text(yaml.safe_dump(infer_cfg))
    orchestrate(path2, mode="inference")
    run2 = sorted(Path(out_dir).iterdir())[-1]
    row = json.loads((run2 / "logs" / "metrics.jsonl").read_text().splitlines()[0])
    assert row["model"] == model_name and 0.0 <= row["accuracy"] <= 1.0 and "threshold" in row
    assert sum(map(sum, row["confusion_matrix"])) == 16
