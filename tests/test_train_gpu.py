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
    # opt-in bf16 inference (`inference.amp: bf16`, an extra key of this engine): same plumbing
    infer_cfg["models"][model_name]["inference"]["amp"] = "bf16"
    path2.write_text(yaml.safe_dump(infer_cfg))
    orchestrate(path2, mode="inference")
    run3 = sorted(Path(out_dir).iterdir())[-1]
    row3 = json.loads((run3 / "logs" / "metrics.jsonl").read_text().splitlines()[-1])
    assert row3["model"] == model_name and 0.0 <= row3["accuracy"] <= 1.0
    assert sum(map(sum, row["confusion_matrix"])) == 16


@pytest.mark.parametrize("accum", [1, 3])
def test_graphed_step_is_bitwise_the_eager_step(accum):
    """graph_step.GraphedTrainStep replays exactly the kernels the eager loop body launches (same order, same
    fixed-order reductions, Philox masks from the same device-resident state): after several optimizer cycles the
    parameters, BatchNorm statistics and counters are bit-identical, also across an epoch boundary where the trainer
    drops the gradients (zero_grad(set_to_none=True))."""
    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.graph_step import GraphedTrainStep
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 2, (8,), generator=g).cuda()) for _ in range(5 * accum)]

    def run(graph: bool):
        torch.manual_seed(11)
        model = HipEfficientNet("b0", "timm", 2).cuda().train()
        opt = HipAdamW(model.parameters(), lr=1e-3, weight_decay=5e-2)
        step = GraphedTrainStep(model, HipCrossEntropyLoss(0.1), opt, accum_steps=accum)
        if not graph:
            step.failed = True                              # the object's own eager path
        losses = []
        for i, (x, y) in enumerate(batches):
            if i == 3 * accum:
                opt.zero_grad(set_to_none=True)              # what train_one_epoch does at the start of an epoch
            losses.append(step.micro_batch(x, y, first=i % accum == 0).clone())
            if (i + 1) % accum == 0:
                step.optimizer_step()
        torch.cuda.synchronize()
        return model, torch.stack(losses).cpu(), step

    m_e, l_e, _ = run(False)
    m_g, l_g, step = run(True)
    assert step.replays >= 4 * accum - accum and not step.failed and step.step_graph is not None
    assert torch.equal(l_e, l_g), (l_e, l_g)
    for (n1, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), n1
    assert float((m_e.conv_stem.weight - HipEfficientNet("b0", "timm", 2).conv_stem.weight.cuda()).abs().max()) > 0   # it trained
