"""BASELINE.json configs[0] — the reference's CPU-runnable plumbing case: the whole
registry / YAML / env-var / trainer / checkpoint / inference surface end to end on CPU with
a tiny generated ImageFolder.  The HIP engine cannot (and must not) run on CPU, so the
model is a plug-in stub registered through the public registry hook; everything around it
is the product code.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml
from PIL import Image
from torch import nn

from deepfakedetection_amd.orchestration import model_registry as reg
from deepfakedetection_amd.orchestration.orchestrator import orchestrate


class TinyNet(nn.Module):
    """Stand-in classifier with the head naming the trainers' freeze mask looks for."""

    def __init__(self, num_classes: int) -> None:
        super().__init__()
        self.conv_stem = nn.Conv2d(3, 8, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(8)
        self.classifier = nn.Linear(8, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.classifier(torch.relu(self.bn1(self.conv_stem(x))).mean((2, 3)))


def _make_dataset(root: Path, classes=("cat", "dog", "eel"), per_class=6, size=40) -> None:
    rng = np.random.default_rng(0)
    for split in ("train", "val", "test"):
        for ci, name in enumerate(classes):
            folder = root / split / name
            folder.mkdir(parents=True)
            for i in range(per_class):
                arr = (rng.random((size, size + 8, 3)) * 255).astype(np.uint8)
                arr[..., ci % 3] //= 2
                mode_l = (i == 0)
                Image.fromarray(arr[..., 0] if mode_l else arr).save(folder / f"{i}.png")   # one grayscale: ensure_rgb path


@pytest.fixture()
def workspace(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    _make_dataset(tmp_path / "data")
    reg.register_model_spec(reg.ModelSpec("tinynet_stub", "deepfakedetection_amd.trainers.efficientnet", "tinynet_stub", 32,
                                          lambda _name, nc: TinyNet(nc)))
    return tmp_path


def _config(tmp_path: Path, **model_block) -> Path:
    cfg = {
        "seed": 1, "device": "cpu",
        "data": {"root": str(tmp_path / "data"), "train_split": "train", "val_split": "val", "test_split": "test",
                 "num_classes": 3, "img_size": 32},
        "models": {"tinynet_stub": {"output_dir": str(tmp_path / "runs" / "tiny"), **model_block}},
        "selection": ["tinynet_stub"],
    }
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    return path


def test_training_then_inference_end_to_end(workspace):
    tmp = workspace
    cfg = _config(tmp, training={"epochs": 2, "batch_size": 6, "num_workers": 0, "resume": "auto", "accum_steps": 2,
                                 "ft_batch_size": 4, "pretrained": False},
                  transforms={"train": {"train_random_resized_crop": False, "train_center_crop": True, "train_random_rotation": True,
                                        "train_color_jitter": True, "train_random_erasing": True},
                              "eval": {"val_resize": True}})
    orchestrate(cfg, mode="training")
    runs = sorted((tmp / "runs" / "tiny").iterdir())
    assert len(runs) == 1
    run = runs[0]
    for rel in ("checkpoints/latest.ckpt", "checkpoints/best.ckpt", "EfficientNetModel.pth", "logs/train.log", "config_snapshot.yaml"):
        assert (run / rel).exists(), rel
    ckpt = torch.load(run / "checkpoints" / "latest.ckpt")
    assert ckpt["epoch"] == 2 and ckpt["warmup_done"] is True and "optimizer" in ckpt and "scheduler" in ckpt
    assert set(torch.load(run / "EfficientNetModel.pth")) == set(TinyNet(3).state_dict())
    log = (run / "logs" / "train.log").read_text()
    assert "Warmup (head only)" in log and "Fine-tune" in log and "val_acc=" in log and "img/s" not in log.split("Data")[0]
    snap = yaml.safe_load((run / "config_snapshot.yaml").read_text())
    assert snap["model"]["name"] == "tinynet_stub" and snap["global"]["seed"] == 1

    weights = run / "EfficientNetModel.pth"
    cfg2 = _config(tmp, inference={"weights": str(weights), "split": "test", "batch_size": 5, "num_workers": 0, "img_size": 32})
    orchestrate(cfg2, mode="inference")
    run2 = sorted((tmp / "runs" / "tiny").iterdir())[-1]
    rows = [json.loads(line) for line in (run2 / "logs" / "metrics.jsonl").read_text().splitlines()]
    assert rows[0]["model"] == "tinynet_stub" and rows[0]["split"] == "test" and 0.0 <= rows[0]["accuracy"] <= 1.0
    assert np.array(rows[0]["confusion_matrix"]).sum() == 18
    assert (run2 / "logs" / "inference.log").exists()


class TinyHeadNet(nn.Module):
    """Stub with the names the EfficientFormerV2 / FasterViT freeze masks look for (`head`, `stages.3`)."""

    def __init__(self, num_classes: int) -> None:
        super().__init__()
        self.stem = nn.Conv2d(3, 8, 3, stride=2, padding=1)
        self.stages = nn.Sequential(*[nn.Conv2d(8, 8, 1) for _ in range(4)])
        self.norm = nn.BatchNorm2d(8)
        self.head = nn.Linear(8, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.head(self.norm(torch.relu(self.stages(self.stem(x)))).mean((2, 3)))


@pytest.mark.parametrize("trainer,weights,img_arg", [("efficientformer_v2", "EfficientFormerV2_S1.pth", True),
                                                     ("fastervit", "FasterVitModel.pth", False)])
def test_other_two_trainers_end_to_end_on_cpu(workspace, trainer, weights, img_arg):
    """trainers/efficientformer_v2.py and trainers/fastervit.py counterparts (shared engine) through the orchestrator:
    warm-up on the head, the reference's fine-tune subsets, file names, throughput.jsonl."""
    built = {}

    def builder(_name, nc, img_size=None):
        built["img_size"] = img_size
        return TinyHeadNet(nc)

    if not img_arg:
        two_arg = builder
        builder = lambda name, nc: two_arg(name, nc)  # noqa: E731  (FasterViT's builder takes no img_size)
    name = f"stub_{trainer}"
    reg.register_model_spec(reg.ModelSpec(name, f"deepfakedetection_amd.trainers.{trainer}", name, 32, builder))
    cfg = yaml.safe_load(_config(workspace, training={"epochs": 1, "batch_size": 6, "num_workers": 0, "pretrained": False}).read_text())
    cfg["models"] = {name: {**cfg["models"]["tinynet_stub"], "output_dir": str(workspace / "runs" / name)}}
    cfg["selection"] = [name]
    path = workspace / f"{trainer}.yaml"
    path.write_text(yaml.safe_dump(cfg))
    orchestrate(path, mode="training")
    run = sorted((workspace / "runs" / name).iterdir())[0]
    ckpt = torch.load(run / "checkpoints" / "latest.ckpt")
    assert ckpt["epoch"] == 1 and ckpt["warmup_done"] is True
    log = (run / "logs" / "train.log").read_text()
    assert "Warmup (head only)" in log and "val_acc=" in log
    rows = [json.loads(line) for line in (run / "logs" / "throughput.jsonl").read_text().splitlines()]
    assert [r["phase"] for r in rows] == ["warmup", "fine-tune"] and rows[0]["images"] == 18
    n_state = len(ckpt["optimizer"]["state"])
    if trainer == "efficientformer_v2":
        assert built["img_size"] == 32                        # timm.create_model(..., img_size=img_size) in the reference
        assert n_state == 4                                   # stages.3.{weight,bias} + head.{weight,bias}
        assert "Fine-tune" not in log                         # BATCH_SIZE batches, no accumulation
    else:
        assert n_state == len(list(TinyHeadNet(3).parameters())) and rows[1]["batch_size"] == 32 and rows[1]["accum_steps"] == 4
        assert "Fine-tune" in log and "accum_steps=4" in log
    assert (run / "checkpoints" / "best.ckpt").exists() == (run / weights).exists()


def test_reference_train_yaml_fails_loudly_for_the_engine_that_is_not_built(workspace):
    """config/train.yaml of the reference selects all three models.  A registered model without a HIP engine must
    stop with the registry's message — not with ModuleNotFoundError from a dangling trainer path (round-1 review)."""
    import importlib

    for key in ("efficientnet_b3", "efficientformerv2_s1", "faster_vit_2_224"):
        spec = reg.get_model_spec(key)
        assert hasattr(importlib.import_module(spec.train_module), "main")
    cfg = yaml.safe_load(_config(workspace, training={"epochs": 1, "batch_size": 6, "num_workers": 0, "pretrained": False}).read_text())
    # every model of the reference's YAMLs now has an engine; a name under a registered prefix without one still
    # stops with the registry's message
    name = "faster_vit_4_21k_224"
    cfg["models"] = {name: dict(cfg["models"]["tinynet_stub"])}
    cfg["selection"] = [name]
    path = workspace / "fv.yaml"
    path.write_text(yaml.safe_dump(cfg))
    with pytest.raises(NotImplementedError, match="registered but its MI355X engine is not built"):
        orchestrate(path, mode="training")
    for key, cls in (("faster_vit_2_224", "HipFasterViT"), ("faster_vit_0_224", "HipFasterViT"), ("efficientformerv2_s1", "HipEfficientFormerV2")):
        assert type(reg.get_model_spec(key).builder(key, 2)).__name__ == cls


def test_class_count_mismatch_exits(workspace):
    cfg = yaml.safe_load(_config(workspace, training={"epochs": 1, "batch_size": 4, "num_workers": 0}).read_text())
    cfg["data"]["num_classes"] = 2
    path = workspace / "bad.yaml"
    path.write_text(yaml.safe_dump(cfg))
    with pytest.raises(SystemExit) as err:
        orchestrate(path, mode="training")
    assert err.value.code == 1


def test_missing_dataset_exits(workspace):
    cfg = yaml.safe_load(_config(workspace, training={"epochs": 1, "batch_size": 4, "num_workers": 0}).read_text())
    cfg["data"]["root"] = str(workspace / "nowhere")
    path = workspace / "bad2.yaml"
    path.write_text(yaml.safe_dump(cfg))
    with pytest.raises(SystemExit):
        orchestrate(path, mode="training")


def test_hip_model_refuses_cpu_input():
    """The product model has no CPU path: a CPU tensor raises instead of falling back."""
    model = reg.get_model_spec("efficientnet_b0").builder("efficientnet_b0", 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 3, 32, 32))


def test_gpu_input_tail_sampler_and_uint8_pipeline():
    """Host side of the GPU input tail: the train/val pipelines end in uint8 HWC tensors, the tail samples
    flips and erase boxes with RandomErasing's rules (box strictly inside the image, height 0 = none)."""
    import torch
    from PIL import Image

    from deepfakedetection_amd import data as D
    from deepfakedetection_amd.trainers.efficientnet import build_transforms

    train_t, val_t, train_tail, val_tail = build_transforms(96, gpu_tail=True)
    img = Image.fromarray((torch.rand(120, 100, 3) * 255).to(torch.uint8).numpy())
    for t in (train_t, val_t):
        out = t(img)
        assert out.dtype == torch.uint8 and tuple(out.shape) == (96, 96, 3)
    assert val_tail.flip_p == 0 and val_tail.erase_p == 0 and train_tail.flip_p == 0.5
    torch.manual_seed(0)
    tail = D.GpuInputTail([0.5] * 3, [0.25] * 3, flip_p=0.5, erase_p=0.5)
    flip, erase = tail.sample(200, 64, 48)
    assert flip.dtype == torch.uint8 and 60 <= int(flip.sum()) <= 140
    assert erase.dtype == torch.int32 and tuple(erase.shape) == (200, 4)
    on = erase[:, 2] > 0
    assert 60 <= int(on.sum()) <= 140
    e = erase[on]
    assert bool(((e[:, 0] >= 0) & (e[:, 1] >= 0) & (e[:, 0] + e[:, 2] <= 64) & (e[:, 1] + e[:, 3] <= 48)).all())
    assert bool((e[:, 2] < 64).all()) and bool((e[:, 3] < 48).all())
    area = (e[:, 2] * e[:, 3]).float() / (64 * 48)
    assert float(area.min()) > 0.01 and float(area.max()) < 0.40


def test_hip_adamw_state_dict_round_trips_into_torch_adamw():
    """HipAdamW shares ONE step counter per group internally; the checkpoint must still carry an
    independent `step` per parameter, or torch.optim.AdamW (the reference's optimizer, trainers/efficientnet.py:487-491)
    advances the shared tensor once per parameter after a resume."""
    import pickle

    import torch

    from deepfakedetection_amd.optim import HipAdamW

    params = [torch.nn.Parameter(torch.randn(3, 2)) for _ in range(5)]
    opt = HipAdamW(params, lr=1e-3)
    for _ in range(3):
        opt.prepare_step()                      # host side of a step: counters + hyper-parameters (no kernel)
    sd = pickle.loads(pickle.dumps(opt.state_dict()))
    steps = [st["step"] for st in sd["state"].values()]
    assert len(steps) == 5 and all(float(s) == 3.0 for s in steps)
    assert len({id(s) for s in steps}) == 5, "per-parameter step tensors must not alias"
    ref = torch.optim.AdamW(params, lr=1e-3)
    ref.load_state_dict(sd)
    for p in params:
        p.grad = torch.ones_like(p)
    ref.step()
    assert all(float(ref.state[p]["step"]) == 4.0 for p in params)
    # and back: HipAdamW re-shares the loaded counters
    opt2 = HipAdamW(params, lr=1e-3)
    opt2.load_state_dict(ref.state_dict())
    opt2.prepare_step()
    assert all(float(opt2.state[p]["step"]) == 5.0 for p in params)


def test_rccl_log_summary_never_raises_and_picks_the_decisions(tmp_path):
    """bench.py --gpus N records what RCCL chose (config.rccl): the summariser must cope with any log, or none."""
    from deepfakedetection_amd.dp import rccl_log_summary

    log = tmp_path / "rccl.log"
    log.write_text("h:1:1 [0] NCCL INFO RCCL version 2.22.3\nh:1:1 [0] NCCL INFO Channel 00/32 : 0 1 2 3 4 5 6 7\nnoise\n"
                   "h:1:1 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1\nh:1:1 [0] NCCL INFO Channel 00/32 : 0 1 2 3 4 5 6 7\n")
    got = rccl_log_summary(str(log))
    assert got["log_lines"] == 5 and got["selected"] == ["RCCL version 2.22.3", "Channel 00/32 : 0 1 2 3 4 5 6 7", "Trees [0] 1/-1/-1->0->-1"]
    assert "note" in rccl_log_summary(str(tmp_path / "missing.log"))
    assert rccl_log_summary(None) is None


def test_default_224_training_pipeline_qualifies_for_the_device_path(monkeypatch):
    """the reference's default toggles at 224 pixels switch rotation AND colour jitter on
    (/root/reference/trainers/efficientnet.py:134-135); with training.gpu_resize the whole training pipeline must still move to the
    device (workers decode + plan only, GpuInputTail rotates / jitters), and fall back to PIL for pictures too large for the kernel."""
    from deepfakedetection_amd import data as D
    from deepfakedetection_amd.trainers.efficientnet import build_transforms

    monkeypatch.delenv("TRANSFORMS", raising=False)
    train, val, train_tail, val_tail = build_transforms(224, gpu_tail=True, gpu_resize=True)
    assert isinstance(train.ops[-1], D.PlanGeometry) and train.ops[-1].mode == "rrc"
    assert not any(isinstance(op, (D.RandomRotation, D.ColorJitter)) for op in train.ops)
    assert train_tail.augments and train_tail.rotate_degrees == 10.0 and train_tail.jitter == (0.2, 0.2, 0.2, 0.05)
    assert train_tail.flip_p == 0.5 and train_tail.erase_p == 0.5 and not val_tail.augments
    big_train, _, big_tail, _ = build_transforms(300, gpu_tail=True, gpu_resize=True)
    assert any(isinstance(op, D.RandomRotation) for op in big_train.ops) and any(isinstance(op, D.ColorJitter) for op in big_train.ops)
    assert not big_tail.augments
    # the sampler: one 16-int job per picture, rotation plan + permutation + factors inside their ranges
    import torch

    torch.manual_seed(0)
    jobs = train_tail.sample_augment(64, 224, 224)
    assert jobs.shape == (64, 16) and jobs.dtype == torch.int32
    f = jobs.view(torch.float32)
    assert set(jobs[:, 0].tolist()) <= {0, 1} and (jobs[:, 15] == 15).all()
    assert all(sorted(row) == [0, 1, 2, 3] for row in jobs[:, 7:11].tolist())
    for col in (11, 12, 13):
        assert float(f[:, col].min()) >= 0.8 and float(f[:, col].max()) <= 1.2
    dh = jobs[:, 14]
    assert ((dh <= 13) | (dh >= 243)).all()                     # |0.05 * 255| = 12.75 -> shifts of at most 13 either way, modulo 256


def test_rotate_plan_is_the_oracles():
    import numpy as np

    from deepfakedetection_amd import data as D
    from oracle import image_ref as IR

    rng = np.random.default_rng(3)
    for _ in range(200):
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        a = float(rng.uniform(-400, 400)) if rng.random() < 0.8 else float(rng.choice([0, 90, 180, 270, -90, 360, 450]))
        assert D.rotate_plan(w, h, a) == IR.rotate_plan(w, h, a)
        assert D.rotate_plan(h, h, a) == IR.rotate_plan(h, h, a)
