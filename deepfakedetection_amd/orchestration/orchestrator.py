"""Orchestration layer: YAML -> per-model run directory -> training or inference job.

Counterpart of the reference's orchestration/orchestrator.py with the same public
surface and behaviour on the hot-path rows of SURVEY.md section 8:

  load_config              :112-125   YAML -> validated plain dict
  ensure_run_dirs          :138-145   <output_dir>/<YYYYmmdd-HHMMSS>/{checkpoints,logs,plots}
  snapshot_config          :148-159   config_snapshot.yaml
  resolve_transform_mapping:162-180
  build_env_overrides      :183-283   the YAML -> environment-variable contract
  run_training_job         :294-307   in-process `main()` of the spec's trainer under patched env
  build_eval_transforms    :316-347
  load_model               :350-377   builder -> device -> eval -> load_state_dict(strict=False)
  run_inference_job        :418-658   threshold sweep (binary), test pass, metrics.jsonl, plots
  orchestrate / run_cli    :661-713

What differs: models come from this package's registry (HIP kernels underneath), the data
pipeline is deepfakedetection_amd.data (no torchvision), soft-max/arg-max of the inference
loops run in the dfd_softmax_argmax kernel, the 501-threshold balanced-accuracy sweep is
vectorised, and three extra variables are exported for the trainers (MODEL_NAME,
FT_BATCH_SIZE, PRETRAINED) on top of the reference's set.
"""

from __future__ import annotations

import argparse
import contextlib
import importlib
import io
import json
import os
import sys
from collections.abc import Iterator
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from time import perf_counter
from typing import Any

import numpy as np
import torch
import yaml
from rich.console import Console
from rich.progress import BarColumn, MofNCompleteColumn, Progress, TextColumn, TimeElapsedColumn, TimeRemainingColumn
from torch import nn
from torch.utils.data import DataLoader

from .. import data as D
from .config_schema import OrchestratorConfig
from .model_registry import get_model_spec
from .train_env import apply_seed, as_bool, require_num_classes

console = Console()

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
RELEASE_URL = "https://github.com/thourihan/DeepfakeDetection/releases/download/v0.3.0/"
RELEASE_FILES = {
    "efficientnet_b3": "efficientnet_b3_v0.3.0.pth",
    "efficientformerv2_s1": "efficientformerv2_s1_v0.3.0.pth",
    "faster_vit_2_224": "faster_vit_2_224_v0.3.0.pth",
}


@dataclass(frozen=True)
class RunPaths:
    run_dir: Path
    checkpoints: Path
    logs: Path
    plots: Path


@contextlib.contextmanager
def patched_environ(overrides: dict[str, str]) -> Iterator[None]:
    """Set environment variables for the duration of a trainer call, then restore."""
    saved = {key: os.environ.get(key) for key in overrides}
    os.environ.update(overrides)
    try:
        yield
    finally:
        for key, old in saved.items():
            if old is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = old


@contextlib.contextmanager
def tee_output(log_path: Path) -> Iterator[None]:
    """Mirror stdout and stderr into `log_path` (truncated) while the block runs."""
    log_path.parent.mkdir(parents=True, exist_ok=True)
    out, err = sys.stdout, sys.stderr
    with log_path.open("w", encoding="utf-8") as log:

        class _Both(io.TextIOBase):
            def write(self, text: str) -> int:
                out.write(text)
                log.write(text)
                return len(text)

            def flush(self) -> None:
                out.flush()
                log.flush()

            def isatty(self) -> bool:
                probe = getattr(out, "isatty", None)
                return bool(probe()) if callable(probe) else False

            @property
            def encoding(self) -> str:  # type: ignore[override]
                return getattr(out, "encoding", "utf-8")

        sys.stdout = sys.stderr = _Both()  # type: ignore[assignment]
        try:
            yield
        finally:
            sys.stdout, sys.stderr = out, err
            log.flush()


def load_config(path: Path) -> dict[str, Any]:
    with Path(path).open("r", encoding="utf-8") as handle:
        raw = yaml.safe_load(handle)
    return OrchestratorConfig(**raw).model_dump()


def ensure_run_dirs(base: Path, timestamp: str) -> RunPaths:
    run = Path(base) / timestamp
    paths = RunPaths(run, run / "checkpoints", run / "logs", run / "plots")
    for folder in (paths.run_dir, paths.checkpoints, paths.logs, paths.plots):
        folder.mkdir(parents=True, exist_ok=True)
    return paths


def snapshot_config(run_paths: RunPaths, *, config: dict[str, Any], model_cfg: dict[str, Any]) -> None:
    snap = {
        "timestamp": datetime.now().isoformat(),
        "global": {key: val for key, val in config.items() if key not in ("models", "selection")},
        "model": model_cfg,
    }
    with (run_paths.run_dir / "config_snapshot.yaml").open("w", encoding="utf-8") as handle:
        yaml.safe_dump(snap, handle)


def resolve_transform_mapping(model_cfg: dict[str, Any], *, phase: str) -> dict[str, Any] | None:
    """Toggles for `phase` ('train' | 'eval'): models.<m>.transforms.<phase>, else a flat
    models.<m>.transforms mapping of scalars, else <training|inference>.transforms."""
    block = model_cfg.get("transforms")
    if isinstance(block, dict):
        scoped = block.get(phase)
        if isinstance(scoped, dict):
            return scoped
        if all(isinstance(v, (bool, int, float, str)) for v in block.values()):
            return block
    section = model_cfg.get("training" if phase == "train" else "inference") or {}   # None when the YAML lacks the block
    nested = section.get("transforms")
    return nested if isinstance(nested, dict) else None


_DATA_ENV = (("train_split", "TRAIN_SPLIT"), ("val_split", "VAL_SPLIT"), ("test_split", "TEST_SPLIT"),
             ("img_size", "IMG_SIZE"), ("num_classes", "NUM_CLASSES"))
_TRAIN_ENV = (("batch_size", "BATCH_SIZE"), ("epochs", "EPOCHS"), ("num_workers", "NUM_WORKERS"), ("lr", "LR"),
              ("weight_decay", "WEIGHT_DECAY"), ("accum_steps", "ACCUM_STEPS"), ("warmup_epochs", "WARMUP_EPOCHS"),
              ("early_stop_patience", "EARLY_STOP_PATIENCE"))
_EXTRA_TRAIN_ENV = (("ft_batch_size", "FT_BATCH_SIZE"), ("pretrained", "PRETRAINED"), ("gpu_input_tail", "GPU_INPUT_TAIL"),
                    ("graph_step", "GRAPH_STEP"), ("fp8_weights", "FP8_WEIGHTS"), ("gpu_resize", "GPU_RESIZE"))


def _first_set(*values: Any) -> Any:
    for value in values:
        if value is not None:
            return value
    return None


def build_env_overrides(*, config: dict[str, Any], model_cfg: dict[str, Any], run_paths: RunPaths,
                        training: bool) -> dict[str, str]:
    data_cfg = config.get("data") or {}
    train_cfg = model_cfg.get("training") or {}
    infer_cfg = model_cfg.get("inference") or {}
    env: dict[str, str] = {"OUTPUT_DIR": str(run_paths.run_dir), "MODEL_NAME": str(model_cfg["name"])}
    if config.get("seed") is not None:
        env["SEED"] = str(config["seed"])
    if config.get("device"):
        env["DEVICE"] = str(config["device"])
    if data_cfg.get("root"):
        env["DATA_ROOT"] = str(Path(data_cfg["root"]).expanduser().resolve())
    for key, var in _DATA_ENV:
        if data_cfg.get(key) is not None:
            env[var] = str(data_cfg[key])
    classes = infer_cfg.get("num_classes", model_cfg.get("num_classes", data_cfg.get("num_classes")))
    if classes is not None:
        env["NUM_CLASSES"] = str(classes)

    if training:
        for key, var in _TRAIN_ENV + _EXTRA_TRAIN_ENV:
            if train_cfg.get(key) is not None:
                env[var] = str(train_cfg[key])
        if train_cfg.get("img_size") is not None:
            env["IMG_SIZE"] = str(train_cfg["img_size"])
        env["RESUME_AUTO"] = "1" if str(train_cfg.get("resume", "")).lower() in ("1", "true", "auto") else "0"
    else:
        spec = get_model_spec(model_cfg["name"])
        if infer_cfg.get("split"):
            env["TEST_SPLIT"] = str(infer_cfg["split"])
        env["BATCH_SIZE"] = str(_first_set(infer_cfg.get("batch_size"), train_cfg.get("batch_size"), 64))
        env["NUM_WORKERS"] = str(_first_set(infer_cfg.get("num_workers"), train_cfg.get("num_workers"),
                                            data_cfg.get("num_workers", 0)))
        env["IMG_SIZE"] = str(_first_set(infer_cfg.get("img_size"), train_cfg.get("img_size"),
                                         data_cfg.get("img_size", spec.default_image_size)))

    toggles = resolve_transform_mapping(model_cfg, phase="train" if training else "eval")
    if toggles:
        env["TRANSFORMS"] = json.dumps(toggles)
    return env


def import_trainer(module_name: str) -> Any:
    module = importlib.import_module(module_name)
    if not hasattr(module, "main"):
        raise AttributeError(f"Trainer module '{module_name}' must expose a main() function.")
    return module.main


def run_training_job(config: dict[str, Any], model_cfg: dict[str, Any], run_paths: RunPaths) -> None:
    spec = get_model_spec(model_cfg["name"])
    env = build_env_overrides(config=config, model_cfg=model_cfg, run_paths=run_paths, training=True)
    log_path = run_paths.logs / "train.log"
    log_path.unlink(missing_ok=True)
    env["LOG_PATH"] = str(log_path)
    console.print(f"[bold]→ training {model_cfg['name']}[/]")
    with patched_environ(env):
        import_trainer(spec.train_module)()


def _to_rgb(image):
    return image if image.mode == "RGB" else image.convert("RGB")


def build_eval_transforms(image_size: int, *, toggles: dict[str, Any] | None = None) -> D.Compose:
    on = {"ensure_rgb": True, "val_resize": True, "val_center_crop": True, "val_to_tensor": True, "val_normalize": True}
    on.update({key: as_bool(val) for key, val in (toggles or {}).items()})
    ops: list[Any] = []
    if on.get("ensure_rgb", True):
        ops.append(D.Lambda(_to_rgb))
    if on.get("val_resize", True):
        ops.append(D.Resize(image_size))
    if on.get("val_center_crop", True):
        ops.append(D.CenterCrop(image_size))
    if on.get("val_to_tensor", True):
        ops.append(D.ToTensor())
    if on.get("val_normalize", True):
        ops.append(D.Normalize(IMAGENET_MEAN, IMAGENET_STD))
    return D.Compose(ops)


def load_model(model_name: str, num_classes: int, weights_path: Path | None, device: torch.device,
               img_size: int | None = None) -> nn.Module:
    """Reference orchestrator.py:350-377.  One addition: a builder that takes `img_size` (EfficientFormerV2: its
    attention-bias tables depend on the resolution; the reference's registry builder omits it, model_registry.py:40,
    and so only works at the default 224) receives the job's image size."""
    import inspect

    builder = get_model_spec(model_name).builder
    if img_size is not None and "img_size" in inspect.signature(builder).parameters:
        model = builder(model_name, num_classes, img_size)
    else:
        model = builder(model_name, num_classes)
    model.to(device)
    model.eval()
    if weights_path is not None:
        if not weights_path.exists():
            console.print(f"[bold red]Weights not found:[/] {weights_path}")
            raise SystemExit(1)
        console.print(f"[bold green]Loading weights[/]: {weights_path} ({weights_path.stat().st_size / 2**20:.2f} MiB)")
        state = torch.load(weights_path, map_location=device)
        if isinstance(state, dict) and "state_dict" in state:
            state = state["state_dict"]
        elif isinstance(state, dict) and "model" in state:
            state = state["model"]
        model.load_state_dict(state, strict=False)
    return model


_GPU_EVAL_TAIL = D.GpuInputTail(IMAGENET_MEAN, IMAGENET_STD)


def build_inference_loader(*, dataset, batch_size: int, num_workers: int) -> DataLoader:
    extra = {"prefetch_factor": 2} if num_workers > 0 else {}
    tf = getattr(dataset, "transform", None)
    if tf is not None and any(isinstance(op, D.PlanGeometry) for op in getattr(tf, "ops", ())):
        extra["collate_fn"] = D.collate_raw         # device-side resize: decoded images + plans (data.PlanGeometry)
    return DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=True,
                      persistent_workers=num_workers > 0, **extra)


def class_probabilities(model: nn.Module, images: torch.Tensor, device: torch.device, amp: bool = False):
    """logits -> (softmax probabilities, arg-max) for one batch (orchestrator.py:589-592).  amp: run the forward in a
    bf16 autocast region (opt-in `inference.amp: bf16`; the reference's inference is f32, which stays the default)."""
    with torch.inference_mode(), torch.autocast(device_type=device.type, dtype=torch.bfloat16, enabled=amp and device.type == "cuda"):
        if isinstance(images, (tuple, list)):          # data.collate_raw batch: resize / crop / normalise on the device
            images = _GPU_EVAL_TAIL(images, device)
        logits = model(images.to(device, non_blocking=True))
        if logits.is_cuda:
            from .. import kernels  # HIP softmax/argmax epilogue

            probs, preds = kernels.softmax_argmax(logits.float().contiguous(), True)
        else:
            probs = torch.softmax(logits, dim=1)
            preds = torch.argmax(probs, dim=1)
    return probs, preds


def best_balanced_accuracy_threshold(scores: np.ndarray, truth: np.ndarray, steps: int = 501) -> float:
    """First threshold on linspace(0,1,steps) that maximises balanced accuracy of
    (scores >= thr); same result as the reference's per-threshold sklearn loop (:533-544)."""
    thresholds = np.linspace(0.0, 1.0, steps, dtype=np.float64)
    pred = scores[None, :] >= thresholds[:, None]                  # [steps, n]
    pos, neg = truth == 1, truth == 0
    recall_pos = (pred & pos[None, :]).sum(1) / max(int(pos.sum()), 1)
    recall_neg = (~pred & neg[None, :]).sum(1) / max(int(neg.sum()), 1)
    return float(thresholds[int(np.argmax((recall_pos + recall_neg) / 2.0))])


def confusion_counts(truth: np.ndarray, pred: np.ndarray) -> np.ndarray:
    labels = np.unique(np.concatenate([truth, pred]))
    index = {int(label): i for i, label in enumerate(labels)}
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    for t, p in zip(truth.tolist(), pred.tolist()):
        cm[index[int(t)], index[int(p)]] += 1
    return cm


def _save_plots(cm: np.ndarray, labels: list[str], truth: np.ndarray, scores: np.ndarray | None, plots: Path) -> None:
    """confusion_matrix.png / roc_curve.png; skipped quietly when matplotlib/sklearn are absent."""
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # noqa: BLE001
        return
    fig, ax = plt.subplots(figsize=(6, 5))
    ax.imshow(cm, cmap="Blues")
    ticks = range(cm.shape[0])
    ax.set_xticks(list(ticks), labels=[labels[i] if i < len(labels) else str(i) for i in ticks])
    ax.set_yticks(list(ticks), labels=[labels[i] if i < len(labels) else str(i) for i in ticks])
    for i in ticks:
        for j in ticks:
            ax.text(j, i, int(cm[i, j]), ha="center", va="center")
    ax.set_xlabel("Predicted label")
    ax.set_ylabel("True label")
    fig.tight_layout()
    fig.savefig(plots / "confusion_matrix.png")
    plt.close(fig)
    if scores is not None:
        try:
            from sklearn.metrics import roc_curve

            fpr, tpr, _ = roc_curve(truth, scores)
        except Exception:  # noqa: BLE001
            return
        fig, ax = plt.subplots(figsize=(6, 5))
        ax.plot(fpr, tpr)
        ax.set_title("ROC Curve")
        ax.set_xlabel("False Positive Rate")
        ax.set_ylabel("True Positive Rate")
        fig.tight_layout()
        fig.savefig(plots / "roc_curve.png")
        plt.close(fig)


def run_inference_job(*, config_path: Path, config: dict[str, Any], model_cfg: dict[str, Any], run_paths: RunPaths) -> None:
    console.print(f"[bold]→ inference {model_cfg['name']}[/]")
    log_path = run_paths.logs / "inference.log"
    log_path.unlink(missing_ok=True)
    with tee_output(log_path):
        _run_inference_job(config_path=config_path, config=config, model_cfg=model_cfg, run_paths=run_paths)


def _resolve_weights(infer_cfg: dict[str, Any], model_name: str, out: Console) -> Path | None:
    value = infer_cfg.get("weights")
    if not value:
        return None
    path = Path(value).expanduser()
    if not path.is_absolute():
        path = (Path.cwd() / path).resolve()
    if not path.exists() and sys.stdin is not None and sys.stdin.isatty():
        answer = input(f"Missing weights at '{path}'. Download from GitHub Releases? [Y/N]: ").strip().lower()
        if answer == "y":
            if model_name in RELEASE_FILES:
                import urllib.request

                path.parent.mkdir(parents=True, exist_ok=True)
                urllib.request.urlretrieve(RELEASE_URL + RELEASE_FILES[model_name], str(path))
            else:
                out.print("[bold yellow]No download URL mapped for this model.[/]")
    return path


def _run_inference_job(*, config_path: Path, config: dict[str, Any], model_cfg: dict[str, Any], run_paths: RunPaths) -> None:
    tty = getattr(sys.stdout, "isatty", None)
    out = Console(file=sys.stdout, force_terminal=bool(tty()) if callable(tty) else False)
    name = model_cfg["name"]
    spec = get_model_spec(name)
    data_cfg = config.get("data") or {}
    infer_cfg = model_cfg.get("inference") or {}        # reference uses .get("inference", {}) and breaks on None
    split = infer_cfg.get("split") or data_cfg.get("test_split", "test")
    batch_size = int(infer_cfg.get("batch_size", 64))
    amp = str(infer_cfg.get("amp", "")).lower() in ("bf16", "bfloat16", "true", "1")       # extra key, default: f32 as the reference
    out.print(f"[bold]Model[/]: {name} | split={split} | batch={batch_size}")
    num_classes = int(model_cfg.get("num_classes", data_cfg.get("num_classes", 2)))
    image_size = int(_first_set(infer_cfg.get("img_size"), data_cfg.get("img_size"), spec.default_image_size))
    num_workers = int(infer_cfg.get("num_workers", 4))

    device_name = config.get("device") or "cuda"
    if device_name.startswith("cuda") and not torch.cuda.is_available():
        out.print("[bold yellow]⚠️  CUDA requested but unavailable[/]: using CPU")
        device_name = "cpu"
    device = torch.device(device_name)

    model = load_model(name, num_classes, _resolve_weights(infer_cfg, name, out), device, image_size)
    if str(infer_cfg.get("fp8_weights", "")).lower() in ("1", "true", "yes", "on") and hasattr(model, "fp8_weights"):
        model.fp8_weights = True       # extra key (FasterViT, with `amp: bf16`): Linear weights as MX fp8 on the scaled fp8 MFMA
    eval_toggles = resolve_transform_mapping(model_cfg, phase="eval")
    transform = build_eval_transforms(image_size, toggles=eval_toggles)
    if str(infer_cfg.get("gpu_resize", "")).lower() in ("1", "true", "yes", "on") and device.type == "cuda":
        # extra key: the workers only decode; Resize + CenterCrop + ToTensor + Normalize run on the device (bit-exact with the
        # PIL pipeline above, tests/test_ops_gpu.py::test_resize_crop_matches_pillow_bit_for_bit).  The device pipeline IS the default
        # eval pipeline: with any eval toggle switched off the results would silently differ from the PIL path, so it only engages
        # when the resolved toggles are the defaults.
        switched_off = sorted(k for k in ("ensure_rgb", "val_resize", "val_center_crop", "val_to_tensor", "val_normalize")
                              if not as_bool((eval_toggles or {}).get(k, True)))
        if switched_off:
            out.print(f"[bold yellow]inference.gpu_resize ignored[/]: eval toggles {switched_off} are off; the PIL pipeline honours them")
        else:
            transform = D.Compose([D.Lambda(_to_rgb), D.PlanGeometry("center", image_size, image_size)])
    root = Path(data_cfg.get("root")).expanduser()
    if not root.is_absolute():
        root = (Path.cwd() / root).resolve()

    threshold = 0.5
    if num_classes == 2:
        val_dir = root / data_cfg.get("val_split", "val")
        if val_dir.exists():
            val_set = D.ImageFolder(val_dir, transform=transform)
            if len(val_set) > 0:
                scores, truth = [], []
                for images, targets in build_inference_loader(dataset=val_set, batch_size=batch_size, num_workers=num_workers):
                    probs, _ = class_probabilities(model, images, device, amp)
                    scores.append(probs[:, 1].cpu())
                    truth.append(targets.cpu())
                s, t = torch.cat(scores).numpy(), torch.cat(truth).numpy()
                if s.size and np.unique(t).size > 1:
                    threshold = best_balanced_accuracy_threshold(s, t)

    split_dir = root / split
    if not split_dir.exists():
        out.print(f"[bold red]Split not found:[/] {split_dir}")
        raise SystemExit(1)
    dataset = D.ImageFolder(split_dir, transform=transform)
    require_num_classes(dataset, num_classes, split=split, dataset_root=split_dir)
    if len(dataset) == 0:
        out.print(f"[bold yellow]No images found in[/] {split_dir}")
        return
    loader = build_inference_loader(dataset=dataset, batch_size=batch_size, num_workers=num_workers)
    progress = Progress(TextColumn("[bold blue]{task.description}"), BarColumn(bar_width=None), MofNCompleteColumn(),
                        TimeElapsedColumn(), TimeRemainingColumn(), TextColumn("{task.fields[speed]}"), console=out)
    all_probs, all_preds, all_targets = [], [], []
    seen, start = 0, perf_counter()
    with progress:
        task = progress.add_task("inference", total=len(loader), speed="")
        for images, targets in loader:
            probs, preds = class_probabilities(model, images, device, amp)
            all_probs.append(probs.cpu())
            all_preds.append(preds.cpu())
            all_targets.append(targets.cpu())
            seen += targets.size(0)
            progress.update(task, advance=1, speed=f"{seen / max(perf_counter() - start, 1e-6):.1f} img/s")

    probs_t, preds_t, targets_t = torch.cat(all_probs), torch.cat(all_preds), torch.cat(all_targets)
    if num_classes == 2:
        preds_t = (probs_t[:, 1] >= threshold).long()
    accuracy = (preds_t == targets_t).float().mean().item()
    metrics: dict[str, Any] = {"model": name, "split": split, "accuracy": accuracy, "timestamp": datetime.now().isoformat()}
    truth_np, pred_np = targets_t.numpy(), preds_t.numpy()
    multi_label = torch.unique(targets_t).numel() > 1
    if multi_label:
        try:
            from sklearn.metrics import roc_auc_score

            if num_classes == 2:
                metrics["roc_auc"] = float(roc_auc_score(truth_np, probs_t[:, 1].numpy()))
            else:
                metrics["roc_auc"] = float(roc_auc_score(truth_np, probs_t.numpy(), multi_class="ovr"))
        except (ImportError, ValueError):
            pass
    if num_classes == 2:
        metrics["threshold"] = float(threshold)
    cm = confusion_counts(truth_np, pred_np)
    metrics["confusion_matrix"] = cm.tolist()
    _save_plots(cm, list(dataset.classes), truth_np, probs_t[:, 1].numpy() if num_classes == 2 and multi_label else None,
                run_paths.plots)
    with (run_paths.logs / "metrics.jsonl").open("a", encoding="utf-8") as handle:
        handle.write(json.dumps(metrics) + "\n")
    extras = " ".join(f"{k}={v:.4f}" for k, v in metrics.items() if isinstance(v, float) and k != "accuracy")
    out.print(f"[bold]Accuracy[/]: {accuracy:.4f} {extras}")


def orchestrate(config_path: Path, *, mode: str) -> None:
    config = load_config(config_path)
    apply_seed(config.get("seed"))
    models_cfg = config.get("models", {})
    if not isinstance(models_cfg, dict):
        raise TypeError("models section must be a mapping of name -> config")
    selection = config.get("selection")
    names = list(models_cfg) if selection is None else [str(n) for n in selection]
    for name in names:
        base = models_cfg.get(name)
        if base is None:
            console.print(f"[bold yellow]Skipping unknown model[/]: {name}")
            continue
        model_cfg = {"name": name, **base}
        run_paths = ensure_run_dirs(Path(model_cfg.get("output_dir") or f"runs/{name}"), datetime.now().strftime("%Y%m%d-%H%M%S"))
        snapshot_config(run_paths, config=config, model_cfg=model_cfg)
        if mode == "training":
            run_training_job(config, model_cfg, run_paths)
        elif mode == "inference":
            run_inference_job(config_path=config_path, config=config, model_cfg=model_cfg, run_paths=run_paths)
        else:
            raise ValueError(f"Unknown mode '{mode}'")


def run_cli() -> None:
    parser = argparse.ArgumentParser(description="DeepfakeDetection orchestrator (MI355X engine)")
    parser.add_argument("--mode", choices=["training", "inference"], default="training")
    parser.add_argument("--config", type=Path)
    args = parser.parse_args()
    path = args.config or Path("config/train.yaml" if args.mode == "training" else "config/inference.yaml")
    orchestrate(path.resolve(), mode=args.mode)


if __name__ == "__main__":
    run_cli()
