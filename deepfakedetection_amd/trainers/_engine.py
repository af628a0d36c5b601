"""Shared body of the EfficientFormerV2 and FasterViT trainers on the MI355X engine.

The reference keeps three near-identical scripts (trainers/efficientnet.py, efficientformer_v2.py, fastervit.py).
`deepfakedetection_amd.trainers.efficientnet` mirrors the first one function by function; the other two differ
from it only in the points SURVEY.md section 3.1 tabulates, captured here as a `TrainerSpec`:

  behaviour                  efficientformer_v2.py                           fastervit.py
  warm-up trainable set      "classifier" or "head" in name (:351-352)       "head" in name (:400-402)
  fine-tune trainable set    UNFREEZE_KEYS substrings (:66-74, :389-393)     all (:434-435)
  fine-tune batch / accum    BATCH_SIZE, none (:419-430)                     32 x 4 hard-coded (:437-453)
  zero_grad                  before the forward (:242)                       after the step (:278-283)
  evaluate                   accuracy only (:206-219)                        accuracy only (:224-240)
  early stop                 no                                              EARLY_STOP_PATIENCE (:322, :526)
  transforms                 rotation / erasing off, jitter 0.1 (:105-165)   same (:119-180)
  model                      timm.create_model(name, num_classes, img_size)  create_model(name); head = Linear(in, nc)

Everything else — env contract, phases, checkpoints, file names, console lines, SystemExit paths — is the
reference's.  Deliberate differences are those of the EfficientNet trainer (bf16 autocast, disabled GradScaler
kept for API parity, HIP loss / AdamW, one host sync per LOG_EVERY steps, local pretrained weights, DP over
WORLD_SIZE ranks, `logs/throughput.jsonl`).
"""

from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path
from time import perf_counter, time

import torch
from rich.progress import BarColumn, MofNCompleteColumn, Progress, TaskID, TextColumn, TimeElapsedColumn, TimeRemainingColumn
from torch import nn, optim
from torch.utils.data import DataLoader

from ..dp import GradAllReducer, all_reduce_counts, broadcast_module_state, init_distributed
from ..orchestration.model_registry import get_model_spec
from ..orchestration.train_env import (
    apply_seed, create_console, env_float, env_int, env_path, env_str, maybe_load_checkpoint, prepare_training_environment,
    save_best_checkpoint, save_latest_checkpoint,
)
from . import efficientnet as _base

DATA_ROOT = Path.home() / "code" / "DeepfakeDetection" / "data" / "Dataset"
LOG_EVERY = 10


@dataclass(frozen=True)
class TrainerSpec:
    model_name: str
    best_weights_name: str
    default_epochs: int
    default_batch_size: int
    warmup_keys: tuple[str, ...]
    unfreeze_keys: tuple[str, ...] | None           # None: fine-tune everything
    ft_batch_size: int | None = None                # None: BATCH_SIZE
    ft_accum_steps: int = 1
    zero_grad_first: bool = False
    early_stop: bool = False
    default_patience: int = 4
    default_img_size: int = 224
    default_num_workers: int = 8
    head_lr: float = 3e-4
    head_wd: float = 5e-2
    ft_lr: float = 1e-4
    ft_wd: float = 5e-2
    pass_img_size: bool = False                      # builder takes img_size (EfficientFormerV2 bias tables)
    transform_kwargs: dict = field(default_factory=lambda: dict(rotation_default=False, erasing_default=False,
                                                                jitter=(0.1, 0.1, 0.1, 0.05), rotation_after_flip=True))


@dataclass(frozen=True)
class EvalResult:
    acc: float
    total: int
    correct: int


def evaluate(model: nn.Module, dl: DataLoader, device: str, tail=None) -> EvalResult:
    """Top-1 accuracy; f32, no autocast (efficientformer_v2.py:206-219, fastervit.py:224-240).  The counter stays on
    the device and is read once at the end (and summed over ranks)."""
    model.eval()
    correct = torch.zeros((), dtype=torch.float64, device=device)
    total = 0
    with torch.inference_mode():
        # (large validation batches: the forward is GPU-bound and the in-stream copy would add ~40 % to it)
        fwd = _base.eval_forward(model, device)
        for inputs, targets in _base.device_batches(dl, device, tail, prefetch=(getattr(dl, "batch_size", 0) or 0) >= 128):
            correct += (fwd(inputs).argmax(1) == targets).sum()
            total += targets.numel()
    n_correct, n_total = all_reduce_counts(float(correct), float(total), device=device)
    return EvalResult(acc=n_correct / max(1, n_total), total=int(n_total), correct=int(n_correct))


def train_one_epoch(model: nn.Module, dl: DataLoader, opt: optim.Optimizer, scaler, criterion: nn.Module, device: str, *,
                    use_cuda_amp: bool, progress: Progress, task: TaskID, accum_steps: int = 1, zero_grad_first: bool = False,
                    reducer: GradAllReducer | None = None, tail=None, label: str = "train", ips_in_extra: bool = False,
                    stepper=None) -> dict:
    """One epoch (efficientformer_v2.py:222-257; fastervit.py:243-300 when accum_steps > 1).  Returns throughput
    figures for logs/throughput.jsonl."""
    model.train()
    start = perf_counter()
    if not zero_grad_first:
        opt.zero_grad(set_to_none=True)
    seen_total = pending = 0
    shown = float("nan")
    for i, (inputs, targets) in enumerate(_base.device_batches(dl, device, tail, prefetch=stepper is not None), 1):
        if stepper is not None:
            # hipGraph replay of the same body; zero_grad belongs to the first micro-batch of a cycle either way
            loss = stepper.micro_batch(inputs, targets, first=pending == 0, last=pending + 1 == accum_steps)
            pending += 1
            if pending == accum_steps:
                stepper.optimizer_step()
                pending = 0
        else:
            if zero_grad_first:
                opt.zero_grad(set_to_none=True)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=use_cuda_amp):
                loss = criterion(model(inputs), targets)
                if accum_steps > 1:
                    loss = loss / accum_steps
            if reducer is not None and pending + 1 == accum_steps:
                reducer.arm()
            scaler.scale(loss).backward()
            pending += 1
            if pending == accum_steps:
                if reducer is not None:
                    reducer.finish()
                scaler.step(opt)
                scaler.update()
                if not zero_grad_first:
                    opt.zero_grad(set_to_none=True)
                pending = 0
        bsz = targets.size(0)
        seen_total += bsz
        if i % LOG_EVERY == 0 or i == len(dl):
            shown = float(loss.detach()) * max(1, accum_steps)          # the only host sync of the loop
        seen = min(i * (dl.batch_size or bsz), len(dl.sampler) if dl.sampler is not None else len(dl.dataset))
        ips = seen / max(1e-6, perf_counter() - start)
        if ips_in_extra:
            progress.update(task, advance=1, description=f"{label} | loss={shown:.4f}", extra=f"{ips:.0f} img/s")
        else:
            progress.update(task, advance=1, description=f"{label} | loss={shown:.4f} | {ips:.0f} img/s")
    if pending > 0:
        if stepper is not None:
            stepper.optimizer_step()
        else:
            if reducer is not None:
                reducer.finish()
            scaler.step(opt)
            scaler.update()
        opt.zero_grad(set_to_none=True)
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    seconds = perf_counter() - start
    return {"images": seen_total, "seconds": seconds, "images_per_sec": seen_total / max(1e-9, seconds),
            "launch": "hipgraph" if (stepper is not None and stepper.replays > 0 and not stepper.failed) else "eager"}


def log_throughput(env, chief: bool, world: int, **record) -> None:
    """One JSON line per phase in OUTPUT_DIR/logs/throughput.jsonl (machine-readable img/s)."""
    if not chief:
        return
    path = Path(env.logs_dir) / "throughput.jsonl"
    path.parent.mkdir(parents=True, exist_ok=True)
    record = {"timestamp": time(), "n_gpus": world, **record}
    if "images_per_sec" in record:
        record["images_per_sec_all_ranks"] = record["images_per_sec"] * world
    with path.open("a", encoding="utf-8") as fh:
        fh.write(json.dumps(record) + "\n")


def run(spec: TrainerSpec) -> None:  # noqa: PLR0915
    console = create_console()
    _base.console = console
    env = prepare_training_environment(weights_name=spec.best_weights_name, best_checkpoint_name="best.ckpt",
                                       latest_checkpoint_name="latest.ckpt")
    apply_seed(env.seed)
    data_root = env_path("DATA_ROOT", DATA_ROOT)
    train_split, val_split = env_str("TRAIN_SPLIT", "Train"), env_str("VAL_SPLIT", "Validation")
    batch_size, epochs = env_int("BATCH_SIZE", spec.default_batch_size), env_int("EPOCHS", spec.default_epochs)
    img_size, num_workers = env_int("IMG_SIZE", spec.default_img_size), env_int("NUM_WORKERS", spec.default_num_workers)
    num_classes = env_int("NUM_CLASSES", 2)
    ft_lr, ft_wd = env_float("LR", spec.ft_lr), env_float("WEIGHT_DECAY", spec.ft_wd)
    patience = env_int("EARLY_STOP_PATIENCE", spec.default_patience)
    # the reference's trainers ignore the YAML's model name; this one honours MODEL_NAME when the
    # orchestrator exports it, so the registry's prefix entries (faster_vit_0_224, efficientformerv2_s0, ...) train too
    model_name = env_str("MODEL_NAME", spec.model_name)

    use_cuda = torch.cuda.is_available()
    device = "cuda" if use_cuda else "cpu"
    if env.device_override:
        if env.device_override.startswith("cuda") and not torch.cuda.is_available():
            console.print("[bold yellow]⚠️  Requested CUDA device not available[/]; falling back to CPU")
            device, use_cuda = "cpu", False
        else:
            device, use_cuda = env.device_override, env.device_override.startswith("cuda")
    rank, local_rank, world = init_distributed() if use_cuda else (0, 0, 1)
    if use_cuda and world > 1:
        device = f"cuda:{local_rank}"
    chief = rank == 0
    torch.backends.cudnn.benchmark = use_cuda and env.seed is None

    if not (data_root / train_split).exists() or not (data_root / val_split).exists():
        console.print(f"[bold red]Dataset not found under[/] {data_root}")
        console.print(f"Expected: {data_root}/{train_split}/<class> and {data_root}/{val_split}/<class>")
        raise SystemExit(1)
    try:
        gpu_tail = use_cuda and (env_str("GPU_INPUT_TAIL", "0").lower() in {"1", "true", "yes"}
                                 or env_str("GPU_RESIZE", "0").lower() in {"1", "true", "yes"})     # device resize implies the device tail
        train_dl, val_dl, *tails = _base.get_loaders(data_root, train_split, val_split, img_size, batch_size, num_workers,
                                                     expected_classes=num_classes, rank=rank, world=world, seed=env.seed or 0,
                                                     gpu_tail=gpu_tail, transform_kwargs=spec.transform_kwargs)
        train_tail, val_tail = tails if tails else (None, None)
    except ValueError as exc:
        console.print("[bold red]Class configuration mismatch[/]", f"→ {exc}")
        console.print("Update `data.num_classes` in your YAML to match the dataset. For MNIST, set it to 10.")
        raise SystemExit(1) from exc
    console.print(f"[bold]Data[/]: train={len(train_dl.dataset)} | val={len(val_dl.dataset)} | bs={batch_size} | "
                  f"steps/epoch={len(train_dl)}" + (f" | ranks={world}" if world > 1 else ""))

    builder = get_model_spec(model_name).builder
    model = builder(model_name, num_classes, img_size) if spec.pass_img_size else builder(model_name, num_classes)
    _base._load_pretrained(model, model_name)
    model.to(memory_format=torch.channels_last)
    model = model.to(device)
    broadcast_module_state(model)
    criterion, make_opt = _base._make_criterion_and_optimizer(use_cuda)
    scaler = torch.amp.GradScaler(enabled=False)
    opt_extra = {"grad_scale": 1.0 / world} if use_cuda else {}

    progress = Progress(TextColumn("[bold blue]{task.description}"), BarColumn(bar_width=None), MofNCompleteColumn(),
                        TimeElapsedColumn(), TimeRemainingColumn(), TextColumn("{task.fields[extra]}"), console=console,
                        transient=False, disable=not chief)
    best_val_acc, best_epoch, epochs_no_improve = -1.0, -1, 0
    warmup_done = env.resume_checkpoint is not None

    with progress:
        if not warmup_done:
            for name, p in model.named_parameters():
                p.requires_grad = any(key in name for key in spec.warmup_keys)
            head = [p for p in model.parameters() if p.requires_grad]
            warm_opt = make_opt(head, lr=spec.head_lr, weight_decay=spec.head_wd, **opt_extra)
            reducer = GradAllReducer(head, arena=getattr(warm_opt, "arena", None)) if world > 1 else None
            if reducer is not None:
                reducer.attach()
            task = progress.add_task("warmup", total=len(train_dl), extra="")
            console.print("[bold]Warmup (head only)[/]")
            # the reference's inline warm-up loop (efficientformer_v2.py:362-382 / fastervit.py:408-428) is
            # train_one_epoch with zero_grad first, no accumulation and the rate in the `extra` column
            stats = train_one_epoch(model, train_dl, warm_opt, scaler, criterion, device, use_cuda_amp=use_cuda, progress=progress,
                                    task=task, accum_steps=1, zero_grad_first=True, reducer=reducer, tail=train_tail,
                                    label="warmup", ips_in_extra=True,
                                    stepper=_base.make_stepper(model, criterion, warm_opt, accum_steps=1, use_cuda=use_cuda, world=world,
                                                                 reducer=reducer))
            log_throughput(env, chief, world, phase="warmup", epoch=0, model=model_name, batch_size=batch_size, **stats)
            if reducer is not None:
                reducer.detach()
            res = evaluate(model, val_dl, device, val_tail)
            best_val_acc, best_epoch, warmup_done = res.acc, 0, True
            console.print(f"[bold cyan]warmup[/] | val_acc={best_val_acc:.4f}")
            if getattr(warm_opt, "arena", None) is not None:
                warm_opt.zero_grad()
                warm_opt.arena.release()

        for name, p in model.named_parameters():
            p.requires_grad = spec.unfreeze_keys is None or any(key in name for key in spec.unfreeze_keys)
        ft_dl, accum = train_dl, 1
        if spec.ft_batch_size is not None:
            accum = spec.ft_accum_steps
            console.print(f"[bold]Fine-tune[/]: bs={spec.ft_batch_size}, accum_steps={accum} "
                          f"(effective ≈ {spec.ft_batch_size * accum * world})")
            ft_dl = _base.make_loader(train_dl.dataset, spec.ft_batch_size, num_workers, shuffle=True, rank=rank, world=world,
                                      seed=env.seed or 0)
        opt = make_opt([p for p in model.parameters() if p.requires_grad], lr=ft_lr, weight_decay=ft_wd, **opt_extra)
        reducer = GradAllReducer([p for p in model.parameters() if p.requires_grad],
                                 arena=getattr(opt, "arena", None)) if world > 1 else None
        if reducer is not None:
            reducer.attach()
        scheduler = optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(1, epochs - 1))
        stepper = _base.make_stepper(model, criterion, opt, accum_steps=accum, use_cuda=use_cuda, world=world, reducer=reducer)
        start_epoch = 0
        resume_state = maybe_load_checkpoint(env, model=model, optimizer=opt, scheduler=scheduler)
        if resume_state is not None:
            start_epoch = int(resume_state.get("epoch", 0))
            best_val_acc = float(resume_state.get("best_val_acc", best_val_acc))
            best_epoch = int(resume_state.get("best_epoch", best_epoch))
            warmup_done = bool(resume_state.get("warmup_done", warmup_done))
            epochs_no_improve = max(0, start_epoch - best_epoch)
            console.print(f"[bold green]Resumed[/] from epoch {start_epoch} using {env.resume_checkpoint}")

        for epoch in range(start_epoch + 1, epochs + 1):
            if hasattr(ft_dl.sampler, "set_epoch"):
                ft_dl.sampler.set_epoch(epoch)
            task = progress.add_task(f"epoch {epoch}", total=len(ft_dl), extra="")
            stats = train_one_epoch(model, ft_dl, opt, scaler, criterion, device, use_cuda_amp=use_cuda, progress=progress, task=task,
                                    accum_steps=accum, zero_grad_first=spec.zero_grad_first, reducer=reducer, tail=train_tail,
                                    stepper=stepper)
            log_throughput(env, chief, world, phase="fine-tune", epoch=epoch, model=model_name,
                           batch_size=ft_dl.batch_size, accum_steps=accum, **stats)
            scheduler.step()
            res = evaluate(model, val_dl, device, val_tail)
            if spec.early_stop:
                console.print(f"[bold cyan]epoch {epoch}[/] | val_acc={res.acc:.4f} ({res.correct}/{res.total}) | "
                              f"lr={scheduler.get_last_lr()[0]:.2e}")
            else:
                console.print(f"[bold cyan]epoch {epoch}[/] | val_acc={res.acc:.4f}")
            improved = res.acc > best_val_acc + 1e-4
            if improved:
                best_val_acc, best_epoch, epochs_no_improve = res.acc, epoch, 0
            else:
                epochs_no_improve += 1
            if chief:
                state = save_latest_checkpoint(env, model=model, optimizer=opt, scheduler=scheduler, epoch=epoch,
                                               best_val_acc=best_val_acc, best_epoch=best_epoch,
                                               extra={"warmup_done": warmup_done})
                if improved:
                    save_best_checkpoint(env, state)
                    console.print(f"[bold green]new best[/] val_acc={best_val_acc:.4f} (epoch {best_epoch}) → saved "
                                  f"{env.best_weights_path.name}")
            if spec.early_stop and not improved and epochs_no_improve >= patience:
                console.print(f"[bold yellow]Early stopping[/]: no improvement for {patience} epoch(s). "
                              f"Best at epoch {best_epoch} with val_acc={best_val_acc:.4f}.")
                break

    console.print(f"[bold green]Best weights saved →[/] {env.best_weights_path.resolve()}")
    console.print(f"[bold green]Best checkpoint saved →[/] {env.best_checkpoint_path.resolve()}")


__all__ = ["EvalResult", "TrainerSpec", "evaluate", "log_throughput", "run", "train_one_epoch"]
