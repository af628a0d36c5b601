"""The environment-variable contract between the orchestrator and the trainers —
behaviour-identical mirror of the reference's orchestration/train_env.py:

  TrainingEnvironment            :31-43      resolved directories, resume, seed, device
  create_console                 :81-95      rich console teed into LOG_PATH
  load_transform_toggles         :110-147    JSON toggles from TRANSFORMS, `required` forced on
  prepare_training_environment   :150-200    OUTPUT_DIR/{checkpoints,logs}, RESUME_AUTO, SEED, DEVICE
  apply_seed                     :203-213
  env_path / env_str / env_int / env_float   :216-251   silent fallback on bad values
  save_latest_checkpoint / save_best_checkpoint / maybe_load_checkpoint   :254-306
  require_num_classes            :309-341

Checkpoint layout (file names, dict keys) is kept so runs are interchangeable:
latest.ckpt / best.ckpt = {epoch, model, optimizer, scheduler, best_val_acc, best_epoch, **extra};
<weights_name> = bare state_dict.  Pinned by tests/test_contract.py against values
captured from the reference module (tests/golden/reference_contract.json).
"""

from __future__ import annotations

import atexit
import io
import json
import os
import random
import sys
from collections.abc import Mapping, Sequence
from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np
import torch
from rich.console import Console

_TRUE_WORDS = frozenset({"1", "true", "yes", "on"})
_open_logs: list[io.TextIOBase] = []


@dataclass(frozen=True)
class TrainingEnvironment:
    output_dir: Path
    checkpoints_dir: Path
    logs_dir: Path
    best_weights_path: Path
    best_checkpoint_path: Path
    latest_checkpoint_path: Path
    resume_checkpoint: Path | None
    seed: int | None
    device_override: str | None


class _Tee(io.TextIOBase):
    """Text stream that writes to the terminal and to a log file."""

    def __init__(self, terminal: Any, log: io.TextIOBase) -> None:
        super().__init__()
        self._terminal, self._log = terminal, log

    def write(self, text: str) -> int:
        self._terminal.write(text)
        self._log.write(text)
        return len(text)

    def flush(self) -> None:
        self._terminal.flush()
        self._log.flush()

    def isatty(self) -> bool:
        probe = getattr(self._terminal, "isatty", None)
        return bool(probe()) if callable(probe) else False

    @property
    def encoding(self) -> str:  # type: ignore[override]
        return getattr(self._terminal, "encoding", "utf-8")


def create_console(*, width: int | None = None) -> Console:
    """Rich console on stdout; mirrored (append mode) into $LOG_PATH when that is set."""
    target: Any = sys.stdout
    log_path = os.environ.get("LOG_PATH")
    if log_path:
        path = Path(log_path).expanduser()
        path.parent.mkdir(parents=True, exist_ok=True)
        handle = path.open("a", encoding="utf-8")
        _open_logs.append(handle)
        atexit.register(handle.close)
        target = _Tee(sys.stdout, handle)
    tty = getattr(sys.stdout, "isatty", None)
    return Console(file=target, force_terminal=bool(tty()) if callable(tty) else False, width=width)


def as_bool(value: Any) -> bool:
    """bool / number / {'1','true','yes','on'} (case- and space-insensitive) -> bool; else False."""
    if isinstance(value, bool):
        return value
    if isinstance(value, (int, float)):
        return value != 0
    if isinstance(value, str):
        return value.strip().lower() in _TRUE_WORDS
    return False


def load_transform_toggles(defaults: Mapping[str, bool], *, env_var: str = "TRANSFORMS",
                           required: Sequence[str] | None = None) -> dict[str, bool]:
    """defaults, overridden by the JSON object in $env_var (ignored when not a JSON object),
    with every key in `required` forced back to True."""
    toggles = dict(defaults)
    raw = os.environ.get(env_var)
    if raw:
        try:
            overrides = json.loads(raw)
        except json.JSONDecodeError:
            overrides = None
        if isinstance(overrides, dict):
            toggles.update({key: as_bool(val) for key, val in overrides.items()})
    for key in required or ():
        if not toggles.get(key, False):
            toggles[key] = True
    return toggles


def prepare_training_environment(*, weights_name: str, default_output_dir: Path | None = None,
                                 best_checkpoint_name: str = "best.ckpt",
                                 latest_checkpoint_name: str = "latest.ckpt") -> TrainingEnvironment:
    root = Path(os.environ.get("OUTPUT_DIR", default_output_dir or Path.cwd())).expanduser().resolve()
    ckpt_dir, log_dir = root / "checkpoints", root / "logs"
    for folder in (root, ckpt_dir, log_dir):
        folder.mkdir(parents=True, exist_ok=True)
    latest = ckpt_dir / latest_checkpoint_name
    resume = latest if (os.environ.get("RESUME_AUTO", "").strip() == "1" and latest.exists()) else None
    return TrainingEnvironment(
        output_dir=root,
        checkpoints_dir=ckpt_dir,
        logs_dir=log_dir,
        best_weights_path=root / weights_name,
        best_checkpoint_path=ckpt_dir / best_checkpoint_name,
        latest_checkpoint_path=latest,
        resume_checkpoint=resume,
        seed=int(os.environ["SEED"]) if "SEED" in os.environ else None,
        device_override=os.environ.get("DEVICE"),
    )


def apply_seed(seed: int | None) -> None:
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def env_path(name: str, default: Path) -> Path:
    raw = os.environ.get(name)
    return Path(raw).expanduser().resolve() if raw else default


def env_str(name: str, default: str) -> str:
    raw = os.environ.get(name)
    return default if raw is None else raw


def _env_number(name: str, default, cast):
    raw = os.environ.get(name)
    if raw is None:
        return default
    try:
        return cast(raw)
    except ValueError:
        return default


def env_int(name: str, default: int) -> int:
    return _env_number(name, default, int)


def env_float(name: str, default: float) -> float:
    return _env_number(name, default, float)


def save_latest_checkpoint(env: TrainingEnvironment, *, model: torch.nn.Module, optimizer, scheduler, epoch: int,
                           best_val_acc: float, best_epoch: int, extra: dict[str, Any] | None = None) -> dict[str, Any]:
    state: dict[str, Any] = {
        "epoch": epoch,
        "model": model.state_dict(),
        "optimizer": None if optimizer is None else optimizer.state_dict(),
        "scheduler": None if scheduler is None else scheduler.state_dict(),
        "best_val_acc": best_val_acc,
        "best_epoch": best_epoch,
    }
    state.update(extra or {})
    torch.save(state, env.latest_checkpoint_path)
    return state


def save_best_checkpoint(env: TrainingEnvironment, state: dict[str, Any]) -> None:
    torch.save(state, env.best_checkpoint_path)
    torch.save(state["model"], env.best_weights_path)


def maybe_load_checkpoint(env: TrainingEnvironment, *, model: torch.nn.Module, optimizer=None,
                          scheduler=None) -> dict[str, Any] | None:
    if env.resume_checkpoint is None:
        return None
    state = torch.load(env.resume_checkpoint, map_location="cpu")
    model.load_state_dict(state["model"])
    if optimizer is not None and state.get("optimizer") is not None:
        optimizer.load_state_dict(state["optimizer"])
    if scheduler is not None and state.get("scheduler") is not None:
        scheduler.load_state_dict(state["scheduler"])
    return state


def require_num_classes(dataset: Any, expected: int, *, split: str, dataset_root: Path | str | None = None) -> None:
    if expected <= 0:
        raise ValueError("expected number of classes must be positive")
    classes = getattr(dataset, "classes", None)
    if classes is None or len(classes) == expected:
        return
    found = len(classes)
    shown = ", ".join(str(c) for c in classes[: min(5, found)]) + (", …" if found > 5 else "")
    where = f" at {Path(dataset_root)}" if dataset_root is not None else ""
    raise ValueError(
        f"Class count mismatch for split '{split}'{where}: dataset exposes {found} classes ({shown}) "
        f"but configuration sets NUM_CLASSES={expected}. "
        "Update config.data.num_classes (e.g., match it to the true number of categories in your ImageFolder)."
    )


# the reference's __all__ omits create_console / env_float / load_transform_toggles although its
# trainers import them (train_env.py:344-355); kept as is, on purpose
__all__ = [
    "TrainingEnvironment", "apply_seed", "env_int", "env_path", "env_str", "maybe_load_checkpoint",
    "prepare_training_environment", "require_num_classes", "save_best_checkpoint", "save_latest_checkpoint",
]
