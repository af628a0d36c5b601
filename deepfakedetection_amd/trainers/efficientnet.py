"""EfficientNet trainer on the MI355X engine — counterpart of the reference's
trainers/efficientnet.py (same `main()` contract, env variables, phases, file outputs):

  get_loaders       :111-234   ImageFolder + toggleable transform pipelines
  evaluate          :237-262   eval-mode f32 forward, arg-max accuracy + CE loss
  train_one_epoch   :265-333   AMP forward, CE(eps=0.1)/accum, backward, step every accum_steps
  main              :336-569   head-only warm-up epoch -> fine-tune with cosine LR, early stop,
                               latest/best checkpoints, <EfficientNetModel.pth>

Deliberate differences, all keeping the reference's defaults:
  * the model is `get_model_spec($MODEL_NAME).builder` (default efficientnet_b3) instead of a
    hard-coded `EfficientNet.from_pretrained`, pretrained weights come from $PRETRAINED /
    weights/<name>.pth if present (no network), else random init with a warning;
  * autocast dtype is bf16 and the GradScaler object is constructed disabled (MI355X);
  * criterion / optimizer are the HIP kernels on a cuda device;
  * the per-iteration `loss.item()` host syncs are replaced by one sync per LOG_EVERY steps;
  * fine-tune micro-batch is $FT_BATCH_SIZE (default 32 = the reference's constant) and
    `prefetch_factor` is only passed with workers (the reference raises with num_workers=0);
  * WORLD_SIZE>1 (torchrun): minibatches are sharded over ranks, gradients all-reduced over
    RCCL, rank 0 logs and writes checkpoints.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from pathlib import Path
from time import perf_counter

import torch
from rich.progress import BarColumn, MofNCompleteColumn, Progress, TaskID, TextColumn, TimeElapsedColumn, TimeRemainingColumn
from torch import nn, optim
from torch.utils.data import DataLoader

from .. import data as D
from ..dp import GradAllReducer, ShardedSampler, all_reduce_counts, broadcast_module_state, init_distributed
from ..orchestration.model_registry import get_model_spec
from ..orchestration.train_env import (
    apply_seed, create_console, env_float, env_int, env_path, env_str, load_transform_toggles, maybe_load_checkpoint,
    prepare_training_environment, require_num_classes, save_best_checkpoint, save_latest_checkpoint,
)

DATA_ROOT = Path.home() / "code" / "DeepfakeDetection" / "data" / "Dataset"
DEFAULT_MODEL = "efficientnet_b3"
DEFAULT_EPOCHS, DEFAULT_BATCH_SIZE, DEFAULT_IMG_SIZE, DEFAULT_NUM_WORKERS = 25, 64, 224, 8
HEAD_LR, HEAD_WD, FT_LR, FT_WD = 3e-4, 5e-2, 1e-4, 5e-2
DEFAULT_PATIENCE = 4
BEST_WEIGHTS_NAME, BEST_CKPT_NAME, LATEST_CKPT_NAME = "EfficientNetModel.pth", "best.ckpt", "latest.ckpt"
FT_BATCH_SIZE, EFFECTIVE_BATCH = 32, 128
DEFAULT_ACCUM_STEPS = max(1, EFFECTIVE_BATCH // FT_BATCH_SIZE)
LOG_EVERY = 10
HEAD_KEYS = ("_fc", "classifier")       # parameter-name substrings of the classification head

console = create_console()


def _rgb(image):
    return image if getattr(image, "mode", "RGB") == "RGB" else image.convert("RGB")


@dataclass(frozen=True)
class EvalResult:
    acc: float
    loss: float
    total: int
    correct: int


def build_transforms(img_size: int, gpu_tail: bool = False, *, rotation_default: bool | None = None,
                     erasing_default: bool | None = None, jitter=(0.2, 0.2, 0.2, 0.05), rotation_after_flip: bool = False,
                     gpu_resize: bool = False):
    """(train, val) pipelines from the toggle defaults of the reference + $TRANSFORMS.
    gpu_tail=True: the pipelines end in uint8 HWC tensors and (train, val, train_tail, val_tail) is
    returned, the tails being `D.GpuInputTail`s that do flip / to-float / normalise / erasing on the GPU.
    The keyword arguments carry what differs between the reference's three trainers (efficientnet.py:128-187 vs
    efficientformer_v2.py:105-165 / fastervit.py:119-180): rotation / erasing off by default, ColorJitter 0.1,
    rotation placed after the horizontal flip.
    gpu_resize (with gpu_tail): Resize / CenterCrop / RandomCrop / RandomResizedCrop move onto the device as well
    (D.PlanGeometry + csrc/dfd_resize.hip, bit-exact with PIL): always for the validation pipeline, and for the training
    pipeline too — RandomRotation and ColorJitter, which the reference's DEFAULT toggles at 224 pixels switch on
    (trainers/efficientnet.py:134-135), run on the device as well (csrc/dfd_augment.hip, byte-exact with Pillow) as long as
    one picture fits a CU's LDS (img_size <= 228); larger pictures with rotation / jitter keep those two in the PIL workers."""
    small = img_size <= 64
    toggles = load_transform_toggles(
        {
            "ensure_rgb": True, "train_resize": True, "train_random_crop": small, "train_center_crop": False,
            "train_random_resized_crop": not small, "train_random_horizontal_flip": True,
            "train_random_rotation": (not small) if rotation_default is None else rotation_default,
            "train_color_jitter": not small,
            "train_random_erasing": (not small) if erasing_default is None else erasing_default,
            "train_to_tensor": True, "train_normalize": True, "val_resize": True, "val_center_crop": True,
            "val_to_tensor": True, "val_normalize": True,
        },
        required=("train_to_tensor", "train_normalize", "val_to_tensor", "val_normalize"),
    )
    on = toggles.get
    normalize = D.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    enlarged = max(img_size + 32, int(img_size * 1.15))
    train: list = [D.Lambda(_rgb)] if on("ensure_rgb", True) else []
    if small:
        if on("train_resize", True):
            train.append(D.Resize(img_size + 4))
        if on("train_random_crop", True):
            train.append(D.RandomCrop(img_size))
        elif on("train_center_crop", False):
            train.append(D.CenterCrop(img_size))
    else:
        if on("train_random_resized_crop", True):
            train.append(D.RandomResizedCrop(img_size, scale=(0.9, 1.0)))
        else:
            if on("train_resize", True):
                train.append(D.Resize(enlarged))
            if on("train_center_crop", True):
                train.append(D.CenterCrop(img_size))
        if not rotation_after_flip and on("train_random_rotation", True):
            train.append(D.RandomRotation(10))
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    if rotation_after_flip and not gpu_tail:
        # efficientformer_v2.py:157-160 / fastervit.py:166-170: flip, then rotation (also for small images)
        if on("train_random_horizontal_flip", True):
            train.append(D.RandomHorizontalFlip())
        if on("train_random_rotation", False):
            train.append(D.RandomRotation(10))
    want_rot = on("train_random_rotation", False) and (rotation_after_flip or not small)
    want_jit = on("train_color_jitter", False)
    aug_fits = img_size * img_size * 3 <= D.AUGMENT_MAX_BYTES
    train_on_gpu = gpu_tail and gpu_resize and (aug_fits or not (want_rot or want_jit))
    if train_on_gpu:
        # the geometric head of the pipeline as a PLAN (same decisions, same RNG calls), pixels untouched
        train = [D.Lambda(_rgb)] if on("ensure_rgb", True) else []
        if small:
            mode = "random" if on("train_random_crop", True) else "center"
            train.append(D.PlanGeometry(mode, img_size, img_size + 4 if on("train_resize", True) else None))
        elif on("train_random_resized_crop", True):
            train.append(D.PlanGeometry("rrc", img_size, rrc=D.RandomResizedCrop(img_size, scale=(0.9, 1.0))))
        else:
            train.append(D.PlanGeometry("center", img_size, enlarged if on("train_resize", True) else None))
        train_tail = D.GpuInputTail(mean if on("train_normalize", True) else [0.0] * 3,
                                    std if on("train_normalize", True) else [1.0] * 3,
                                    flip_p=0.5 if on("train_random_horizontal_flip", True) else 0.0,
                                    erase_p=0.5 if on("train_random_erasing", False) else 0.0,
                                    rotate_degrees=10.0 if want_rot else 0.0, jitter=jitter if want_jit else None)
    elif gpu_tail:
        if rotation_after_flip and on("train_random_rotation", False):
            train.append(D.RandomRotation(10))      # rotation by a random angle commutes in distribution with the flip
        # flip commutes with the per-pixel colour jitter, so it can move behind it onto the device
        if on("train_color_jitter", False):
            train.append(D.ColorJitter(*jitter))
        train.append(D.ToUint8HWC())
        train_tail = D.GpuInputTail(mean if on("train_normalize", True) else [0.0] * 3,
                                    std if on("train_normalize", True) else [1.0] * 3,
                                    flip_p=0.5 if on("train_random_horizontal_flip", True) else 0.0,
                                    erase_p=0.5 if on("train_random_erasing", False) else 0.0)
    else:
        if not rotation_after_flip and on("train_random_horizontal_flip", True):
            train.append(D.RandomHorizontalFlip())
        if on("train_color_jitter", False):
            train.append(D.ColorJitter(*jitter))
        if on("train_to_tensor", True):
            train.append(D.ToTensor())
        if on("train_normalize", True):
            train.append(normalize)
        if on("train_random_erasing", False):
            train.append(D.RandomErasing(p=0.5, scale=(0.02, 0.33), ratio=(0.3, 3.3), value=0))

    val: list = [D.Lambda(_rgb)] if on("ensure_rgb", True) else []
    if gpu_tail and gpu_resize and on("val_center_crop", True):
        val.append(D.PlanGeometry("center", img_size, (img_size if small else enlarged) if on("val_resize", True) else None))
        val_tail = D.GpuInputTail(mean if on("val_normalize", True) else [0.0] * 3, std if on("val_normalize", True) else [1.0] * 3)
        return D.Compose(train), D.Compose(val), train_tail, val_tail
    if on("val_resize", True):
        val.append(D.Resize(img_size if small else enlarged))
    if on("val_center_crop", True):
        val.append(D.CenterCrop(img_size))
    if gpu_tail:
        val.append(D.ToUint8HWC())
        val_tail = D.GpuInputTail(mean if on("val_normalize", True) else [0.0] * 3, std if on("val_normalize", True) else [1.0] * 3)
        return D.Compose(train), D.Compose(val), train_tail, val_tail
    if on("val_to_tensor", True):
        val.append(D.ToTensor())
    if on("val_normalize", True):
        val.append(normalize)
    return D.Compose(train), D.Compose(val)


def make_loader(dataset, batch_size: int, num_workers: int, *, shuffle: bool, rank: int = 0, world: int = 1,
                seed: int = 0) -> DataLoader:
    extra = {"prefetch_factor": 2} if num_workers > 0 else {}
    # training shards are padded to equal length (equal step counts for the all-reduce); validation shards are not
    sampler = ShardedSampler(len(dataset), rank, world, shuffle=shuffle, seed=seed, pad=shuffle) if world > 1 else None
    # pipelines that end in D.PlanGeometry ship variable-size decoded images: packed by D.collate_raw
    tf = getattr(dataset, "transform", None)
    if tf is not None and any(isinstance(op, D.PlanGeometry) for op in getattr(tf, "ops", ())):
        extra["collate_fn"] = D.collate_raw
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                      num_workers=num_workers, pin_memory=True, persistent_workers=num_workers > 0, **extra)


def get_loaders(data_root: Path, train_split: str, val_split: str, img_size: int, batch_size: int, num_workers: int, *,
                expected_classes: int | None = None, rank: int = 0, world: int = 1, seed: int = 0, gpu_tail: bool = False,
                transform_kwargs: dict | None = None, gpu_resize: bool | None = None):
    """(train loader, val loader); with gpu_tail also (train tail, val tail) to apply to each uint8 batch.
    gpu_resize (default $GPU_RESIZE, YAML training.gpu_resize): resize / crop on the device too (implies the GPU tail)."""
    tails = ()
    tk = transform_kwargs or {}
    if gpu_resize is None:
        gpu_resize = env_str("GPU_RESIZE", "0").lower() in {"1", "true", "yes"}
    if gpu_tail:
        train_t, val_t, *tails = build_transforms(img_size, gpu_tail=True, gpu_resize=gpu_resize, **tk)
    else:
        train_t, val_t = build_transforms(img_size, **tk)
    train_ds = D.ImageFolder(data_root / train_split, transform=train_t)
    if expected_classes is not None:
        require_num_classes(train_ds, expected_classes, split=train_split)
    val_ds = D.ImageFolder(data_root / val_split, transform=val_t)
    return (make_loader(train_ds, batch_size, num_workers, shuffle=True, rank=rank, world=world, seed=seed),
            make_loader(val_ds, batch_size, num_workers, shuffle=False, rank=rank, world=world, seed=seed), *tails)


def _to_device(batch_x: torch.Tensor, device: str, tail) -> torch.Tensor:
    if tail is not None:
        return tail(batch_x, device)                         # uint8 NHWC -> normalised f32 on the GPU
    return batch_x.to(device, non_blocking=True).to(memory_format=torch.channels_last)


def device_batches(dl, device: str, tail, prefetch: bool = False):
    """(inputs, targets) on the device for every batch of `dl`, with the host-to-device copy of batch i+1 issued on a copy
    stream BEFORE the caller enqueues the work of batch i (the reference's loop, trainers/efficientnet.py:283-287, copies
    in-stream: at batch 256 that is 154 MB, ~3 ms of PCIe time the kernels wait for).  Asked for by the hipGraph-replayed
    loop only (`prefetch`): measured on MI355X, B0, 256 x 1: 14.6 k -> 17.1 k images/s, 32 x 4: 5.8 k -> 6.1 k; the eager
    loop is host-bound and loses 2..10 % to the extra stream bookkeeping.  The GPU input tail (`tail`: uint8 batches +
    dfd_image_prep) and CPU runs keep the in-stream path.  PREFETCH_H2D=0 switches the copy stream off."""
    use = prefetch and tail is None and str(device).startswith("cuda") and os.environ.get("PREFETCH_H2D", "1") != "0"
    if not use:
        for batch_x, batch_y in dl:
            yield _to_device(batch_x, device, tail), batch_y.to(device, non_blocking=True)
        return
    copy = torch.cuda.Stream(device=device)

    def stage(batch):
        with torch.cuda.stream(copy):
            x = batch[0].to(device, non_blocking=True)
            y = batch[1].to(device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy)
        return x, y, done

    it = iter(dl)
    try:
        nxt = stage(next(it))
    except StopIteration:
        return
    while nxt is not None:
        x, y, done = nxt
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        x.record_stream(cur)                    # allocated on the copy stream, consumed on this one
        y.record_stream(cur)
        try:
            nxt = stage(next(it))               # requested before the caller enqueues this batch's kernels
        except StopIteration:
            nxt = None
        yield x.to(memory_format=torch.channels_last), y


def evaluate(model: nn.Module, dl: DataLoader, device: str, criterion: nn.Module, tail=None) -> EvalResult:
    """Top-1 accuracy and mean loss; f32, no autocast (reference :237-262).  Counters stay
    on the device and are read once at the end (and summed over ranks)."""
    model.eval()
    correct = torch.zeros((), dtype=torch.float64, device=device)
    loss_sum = torch.zeros((), dtype=torch.float64, device=device)
    total = 0
    with torch.inference_mode():
        # (large validation batches: the forward is GPU-bound and the in-stream copy would add ~40 % to it)
        fwd = eval_forward(model, device)
        for inputs, targets in device_batches(dl, device, tail, prefetch=(getattr(dl, "batch_size", 0) or 0) >= 128):
            logits = fwd(inputs)
            loss_sum += criterion(logits, targets).double() * targets.size(0)
            correct += (logits.argmax(1) == targets).sum()
            total += targets.numel()
    n_correct, n_total, s_loss = all_reduce_counts(float(correct), float(total), float(loss_sum), device=device)
    return EvalResult(acc=n_correct / max(1, n_total), loss=s_loss / max(1, n_total), total=int(n_total), correct=int(n_correct))


def eval_forward(model: nn.Module, device: str):
    """The callable evaluate() runs per batch: the model itself, or — on a HIP device, unless GRAPH_STEP is off — a
    graph_step.GraphedForward kept on the model, which replays the eval-mode forward per batch shape (bit-identical to
    the eager forward; at the reference's validation batch sizes the eager forward is host-bound)."""
    if not str(device).startswith("cuda") or env_str("GRAPH_STEP", "1").lower() in {"0", "false", "no", "off"}:
        return model
    fwd = model.__dict__.get("_graphed_eval")
    if fwd is None:
        from ..graph_step import GraphedForward

        fwd = model.__dict__["_graphed_eval"] = GraphedForward(model)
    return fwd


def make_stepper(model: nn.Module, criterion: nn.Module, opt, *, accum_steps: int, use_cuda: bool, world: int, reducer=None):
    """hipGraph replay of the loop body ($GRAPH_STEP, YAML training.graph_step; default on) on a HIP device with the
    HIP optimizer; with `world` > 1 the object also drives the gradient exchange (`reducer`): graph(zero_grad + forward +
    backward) -> all-reduce of the flat gradient arena (RCCL, outside of capture) -> graph(AdamW).  Otherwise None: the
    loop runs eagerly as the reference's does."""
    if not use_cuda or getattr(opt, "arena", None) is None or (world > 1 and reducer is None):
        return None
    if env_str("GRAPH_STEP", "1").lower() in {"0", "false", "no", "off"}:
        return None
    from ..graph_step import GraphedTrainStep

    return GraphedTrainStep(model, criterion, opt, accum_steps=accum_steps, use_amp=True, reducer=reducer)


def train_one_epoch(model: nn.Module, dl: DataLoader, opt: optim.Optimizer, scaler, criterion: nn.Module, device: str, *,
                    use_cuda_amp: bool, progress: Progress, task: TaskID, accum_steps: int = 1,
                    reducer: GradAllReducer | None = None, tail=None, stepper=None, stats: dict | None = None) -> float:
    """One epoch; returns the mean training loss (reference :265-333).  `stepper` (graph_step.GraphedTrainStep)
    replays the captured loop body instead of dispatching it; `stats` receives throughput figures."""
    model.train()
    start = perf_counter()
    opt.zero_grad(set_to_none=True)
    loss_sum = torch.zeros((), dtype=torch.float64, device=device)
    seen_total = pending = 0
    shown = float("nan")
    for i, (inputs, targets) in enumerate(device_batches(dl, device, tail, prefetch=stepper is not None), 1):
        if stepper is not None:
            # zero_grad is part of the "first" body; `last` lets an eager micro-batch overlap the DP exchange with its backward
            loss = stepper.micro_batch(inputs, targets, first=pending == 0, last=pending + 1 == accum_steps)
            pending += 1
            if pending == accum_steps:
                stepper.optimizer_step()
                pending = 0
        else:
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=use_cuda_amp):
                loss = criterion(model(inputs), targets)
                if accum_steps > 1:
                    loss = loss / accum_steps
            if reducer is not None and pending + 1 == accum_steps:
                reducer.arm()                   # this backward completes the step: buckets leave as they fill
            scaler.scale(loss).backward()
            pending += 1
            if pending == accum_steps:
                if reducer is not None:
                    reducer.finish()
                scaler.step(opt)
                scaler.update()
                opt.zero_grad(set_to_none=True)
                pending = 0
        bsz = targets.size(0)
        seen_total += bsz
        loss_sum += loss.detach().double() * (bsz * max(1, accum_steps))
        if i % LOG_EVERY == 0 or i == len(dl):
            shown = float(loss.detach()) * max(1, accum_steps)          # the only host sync of the loop
        seen = min(i * (dl.batch_size or bsz), len(dl.sampler) if dl.sampler is not None else len(dl.dataset))
        ips = seen / max(1e-6, perf_counter() - start)
        progress.update(task, advance=1, description=f"train | loss={shown:.4f} | {ips:.0f} img/s")
    if pending > 0:
        if stepper is not None:
            stepper.optimizer_step()
        else:
            if reducer is not None:
                reducer.finish()
            scaler.step(opt)
            scaler.update()
        opt.zero_grad(set_to_none=True)
    if stats is not None:
        if str(device).startswith("cuda"):
            torch.cuda.synchronize()
        seconds = perf_counter() - start
        stats.update(images=seen_total, seconds=seconds, images_per_sec=seen_total / max(1e-9, seconds),
                     launch="hipgraph" if (stepper is not None and stepper.replays > 0 and not stepper.failed) else "eager")
    (total_loss,) = all_reduce_counts(float(loss_sum), device=device)
    (total_seen,) = all_reduce_counts(float(seen_total), device=device)
    return total_loss / max(1.0, total_seen)


def _load_pretrained(model: nn.Module, name: str) -> None:
    hint = env_str("PRETRAINED", "")
    if hint.lower() in ("0", "false", "no", "none"):
        return
    candidates = [Path(hint)] if hint else [Path("weights") / f"{name}.pth", Path("weights") / f"{name}_v0.3.0.pth"]
    for path in candidates:
        if path.is_file():
            state = torch.load(path, map_location="cpu")
            if isinstance(state, dict) and "state_dict" in state:
                state = state["state_dict"]
            elif isinstance(state, dict) and "model" in state:
                state = state["model"]
            own = model.state_dict()
            usable = {k: v for k, v in state.items() if k in own and v.shape == own[k].shape}   # head may differ in classes
            model.load_state_dict(usable, strict=False)
            console.print(f"[bold green]Loaded pretrained weights[/] {path} ({len(usable)}/{len(own)} tensors)")
            return
    console.print("[bold yellow]⚠️  No local pretrained weights[/] (set training.pretrained); starting from random init")


def _log_throughput(env, chief: bool, world: int, **record) -> None:
    """One JSON line per phase in OUTPUT_DIR/logs/throughput.jsonl: the machine-readable twin of the progress bar's
    `img/s` (reference :317-325)."""
    if not chief or "images_per_sec" not in record:
        return
    import json
    from time import time

    path = Path(env.logs_dir) / "throughput.jsonl"
    path.parent.mkdir(parents=True, exist_ok=True)
    record = {"timestamp": time(), "n_gpus": world, **record, "images_per_sec_all_ranks": record["images_per_sec"] * world}
    with path.open("a", encoding="utf-8") as fh:
        fh.write(json.dumps(record) + "\n")


def _make_criterion_and_optimizer(use_cuda: bool):
    if use_cuda:
        from ..optim import HipAdamW, HipCrossEntropyLoss

        return HipCrossEntropyLoss(label_smoothing=0.1), HipAdamW
    # device: cpu — the reference's own torch path; only non-HIP (plug-in) modules can run there
    return nn.CrossEntropyLoss(label_smoothing=0.1), optim.AdamW


def main() -> None:  # noqa: PLR0915
    global console
    console = create_console()      # re-bound per call: LOG_PATH differs per orchestrated run
    env = prepare_training_environment(weights_name=BEST_WEIGHTS_NAME, best_checkpoint_name=BEST_CKPT_NAME,
                                       latest_checkpoint_name=LATEST_CKPT_NAME)
    apply_seed(env.seed)
    data_root = env_path("DATA_ROOT", DATA_ROOT)
    train_split, val_split = env_str("TRAIN_SPLIT", "Train"), env_str("VAL_SPLIT", "Validation")
    batch_size, epochs = env_int("BATCH_SIZE", DEFAULT_BATCH_SIZE), env_int("EPOCHS", DEFAULT_EPOCHS)
    img_size, num_workers = env_int("IMG_SIZE", DEFAULT_IMG_SIZE), env_int("NUM_WORKERS", DEFAULT_NUM_WORKERS)
    num_classes = env_int("NUM_CLASSES", 2)
    accum_steps = env_int("ACCUM_STEPS", DEFAULT_ACCUM_STEPS)
    ft_lr, ft_wd = env_float("LR", FT_LR), env_float("WEIGHT_DECAY", FT_WD)
    patience = env_int("EARLY_STOP_PATIENCE", DEFAULT_PATIENCE)
    ft_batch = env_int("FT_BATCH_SIZE", FT_BATCH_SIZE)
    model_name = env_str("MODEL_NAME", DEFAULT_MODEL)

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
        # $GPU_INPUT_TAIL (YAML training.gpu_input_tail): loaders ship uint8 batches, the device does
        # flip / to-float / normalise / erasing
        gpu_tail = use_cuda and (env_str("GPU_INPUT_TAIL", "0").lower() in {"1", "true", "yes"}
                                 or env_str("GPU_RESIZE", "0").lower() in {"1", "true", "yes"})     # device resize implies the device tail
        train_dl, val_dl, *tails = get_loaders(data_root, train_split, val_split, img_size, batch_size, num_workers,
                                               expected_classes=num_classes, rank=rank, world=world, seed=env.seed or 0,
                                               gpu_tail=gpu_tail)
        train_tail, val_tail = tails if tails else (None, None)
    except ValueError as exc:
        console.print("[bold red]Class configuration mismatch[/]", f"→ {exc}")
        console.print("Update `data.num_classes` in your YAML to match the dataset. For MNIST, set it to 10.")
        raise SystemExit(1) from exc
    console.print(f"[bold]Data[/]: train={len(train_dl.dataset)} | val={len(val_dl.dataset)} | bs={batch_size} | "
                  f"steps/epoch={len(train_dl)}" + (f" | ranks={world}" if world > 1 else ""))

    model = get_model_spec(model_name).builder(model_name, num_classes)
    _load_pretrained(model, model_name)
    model.to(memory_format=torch.channels_last)
    model = model.to(device)
    broadcast_module_state(model)
    criterion, make_opt = _make_criterion_and_optimizer(use_cuda)
    scaler = torch.amp.GradScaler(enabled=False)        # bf16 needs no loss scaling; calls kept for parity
    opt_extra = {"grad_scale": 1.0 / world} if use_cuda else {}

    progress = Progress(TextColumn("[bold blue]{task.description}"), BarColumn(bar_width=None), MofNCompleteColumn(),
                        TimeElapsedColumn(), TimeRemainingColumn(), TextColumn("{task.fields[extra]}"), console=console,
                        transient=False, disable=not chief)
    best_val_acc, best_epoch, epochs_no_improve = -1.0, -1, 0
    warmup_done = env.resume_checkpoint is not None

    with progress:
        if not warmup_done:
            for name, p in model.named_parameters():
                p.requires_grad = any(key in name for key in HEAD_KEYS)
            head = [p for p in model.parameters() if p.requires_grad]
            warm_opt = make_opt(head, lr=HEAD_LR, weight_decay=HEAD_WD, **opt_extra)
            reducer = GradAllReducer(head, arena=getattr(warm_opt, "arena", None)) if world > 1 else None
            if reducer is not None:
                reducer.attach()
            task = progress.add_task("warmup (head only)", total=len(train_dl), extra="")
            console.print("[bold]Warmup (head only)[/]")
            stats: dict = {}
            train_one_epoch(model, train_dl, warm_opt, scaler, criterion, device, use_cuda_amp=use_cuda, progress=progress,
                            task=task, accum_steps=1, reducer=reducer, tail=train_tail, stats=stats,
                            stepper=make_stepper(model, criterion, warm_opt, accum_steps=1, use_cuda=use_cuda, world=world, reducer=reducer))
            _log_throughput(env, chief, world, phase="warmup", epoch=0, model=model_name, batch_size=batch_size, **stats)
            if reducer is not None:
                reducer.detach()
            res = evaluate(model, val_dl, device, criterion, val_tail)
            console.print(f"[bold cyan]warmup[/] | val_acc={res.acc:.4f} | val_loss={res.loss:.4f} ({res.correct}/{res.total})")
            best_val_acc, best_epoch, warmup_done = res.acc, 0, True
            if getattr(warm_opt, "arena", None) is not None:
                warm_opt.zero_grad()
                warm_opt.arena.release()

        for p in model.parameters():
            p.requires_grad = True
        console.print(f"[bold]Fine-tune[/]: bs={ft_batch}, accum_steps={accum_steps} (effective ≈ {ft_batch * accum_steps * world})")
        train_dl_ft = make_loader(train_dl.dataset, ft_batch, num_workers, shuffle=True, rank=rank, world=world,
                                  seed=env.seed or 0)
        opt = make_opt([p for p in model.parameters() if p.requires_grad], lr=ft_lr, weight_decay=ft_wd, **opt_extra)
        reducer = GradAllReducer(model.parameters(), arena=getattr(opt, "arena", None)) if world > 1 else None
        if reducer is not None:
            reducer.attach()
        scheduler = optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(1, epochs - 1))
        stepper = make_stepper(model, criterion, opt, accum_steps=accum_steps, use_cuda=use_cuda, world=world, reducer=reducer)
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
            if hasattr(train_dl_ft.sampler, "set_epoch"):
                train_dl_ft.sampler.set_epoch(epoch)
            task = progress.add_task(f"epoch {epoch}", total=len(train_dl_ft), extra="")
            stats = {}
            train_loss = train_one_epoch(model, train_dl_ft, opt, scaler, criterion, device, use_cuda_amp=use_cuda,
                                         progress=progress, task=task, accum_steps=accum_steps, reducer=reducer,
                                         tail=train_tail, stepper=stepper, stats=stats)
            _log_throughput(env, chief, world, phase="fine-tune", epoch=epoch, model=model_name, batch_size=ft_batch,
                            accum_steps=accum_steps, **stats)
            scheduler.step()
            res = evaluate(model, val_dl, device, criterion, val_tail)
            console.print(f"[bold cyan]epoch {epoch}[/] | train_loss={train_loss:.4f} | val_loss={res.loss:.4f} | "
                          f"val_acc={res.acc:.4f} ({res.correct}/{res.total}) | lr={scheduler.get_last_lr()[0]:.2e}")
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
            if not improved and epochs_no_improve >= patience:
                console.print(f"[bold yellow]Early stopping[/]: no improvement for {patience} epoch(s). "
                              f"Best at epoch {best_epoch} with val_acc={best_val_acc:.4f}.")
                break

    console.print(f"[bold green]Best weights saved →[/] {env.best_weights_path.resolve()}")
    console.print(f"[bold green]Best checkpoint saved →[/] {env.best_checkpoint_path.resolve()}")


if __name__ == "__main__":
    main()
