"""Loss and optimizer objects of the hot loop, backed by libdfd_hip.so.

`HipCrossEntropyLoss` replaces `nn.CrossEntropyLoss(label_smoothing=0.1)`
(trainers/efficientnet.py:412); `HipAdamW` replaces `optim.AdamW(params, lr, weight_decay)`
(trainers/efficientnet.py:440,487-491) with ONE fused multi-tensor kernel per step and
keeps torch.optim.AdamW's state layout (`step`, `exp_avg`, `exp_avg_sq`) so the
checkpoints written by train_env.save_latest_checkpoint stay interchangeable.
"""

from __future__ import annotations

import torch
from torch import nn

from . import kernels as K
from ._lib import ADAMW_HP_LEN
from .arena import GradArena
from .functions import CrossEntropyFunction

_CHUNK = 4096          # elements per workgroup of the fused kernel (~1000 workgroups for EfficientNet-B0)


class HipCrossEntropyLoss(nn.Module):
    def __init__(self, label_smoothing: float = 0.0) -> None:
        super().__init__()
        self.label_smoothing = float(label_smoothing)

    def forward(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        if not logits.is_cuda:
            raise RuntimeError("HipCrossEntropyLoss needs logits on a HIP device (no CPU fallback)")
        return CrossEntropyFunction.apply(logits.float(), targets, self.label_smoothing)


class HipAdamW(torch.optim.Optimizer):
    """AdamW (decoupled weight decay, bias correction) as one kernel launch per group."""

    def __init__(self, params, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, grad_scale: float = 1.0, use_arena: bool = True) -> None:
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)
        super().__init__(params, defaults)
        self._tables: dict[int, tuple[tuple[int, ...], torch.Tensor, torch.Tensor]] = {}
        self._hp: dict[int, torch.Tensor] = {}
        self._shared_step: dict[int, torch.Tensor] = {}
        # gradients of the trainable parameters live at fixed addresses (see arena.py): the
        # pointer table below is then built once and the backward kernels write in place
        self.arena: GradArena | None = None
        trainable = [p for g in self.param_groups for p in g["params"] if p.requires_grad and p.is_cuda]
        if use_arena and trainable:
            self.arena = GradArena(trainable)

    def state_dict(self):
        """torch.optim.AdamW's format: every parameter gets its OWN `step` tensor.  Internally all parameters of a
        group share one counter object (see prepare_step); pickling that aliasing would make torch.optim.AdamW advance
        the shared counter once per parameter after loading the checkpoint (train_env.save_latest_checkpoint stores this
        dict as is, reference train_env.py:254-278)."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (v.clone() if n == "step" and isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._shared_step.clear()                  # the loaded per-parameter counters are re-shared at the next step

    def zero_grad(self, set_to_none: bool = True) -> None:
        """Drop the gradients (always set-to-none: an arena slot must not be both the
        destination of a backward kernel and the accumulator autograd adds into)."""
        super().zero_grad(set_to_none=True)
        if self.arena is not None:
            self.arena.reset()

    def _table(self, gi: int, entries: list[tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]]) -> torch.Tensor:
        key = tuple(t.data_ptr() for e in entries for t in e)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[2]
        rows = []
        for p, g, m, v in entries:
            base = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr())
            for off in range(0, p.numel(), _CHUNK):
                rows.append([b + 4 * off for b in base] + [min(_CHUNK, p.numel() - off)])
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        dev = cached[2] if cached is not None and cached[2].shape == host.shape else torch.empty_like(host, device=entries[0][0].device)
        dev.copy_(host, non_blocking=True)          # pinned + async: legal under stream capture
        self._tables[gi] = (key, host, dev)
        return dev

    def _ensure_state(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if not st:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def prepare_step(self) -> None:
        """Advance the step counters and upload this step's hyper-parameters
        {lr, betas, eps, wd, bias corrections, grad_scale} to device memory.

        `step()` does this itself in eager mode.  When the whole training step is a
        captured hipGraph, call prepare_step() before every replay: the kernel reads the
        values from memory, so the replayed launch sees the new learning rate / step."""
        for gi, group in enumerate(self.param_groups):
            # one step counter per group: every trainable parameter's state["step"] is the SAME CPU tensor, advanced by
            # one in-place add per optimizer step (per-parameter counters were 2 x len(params) host-side tensor ops per
            # step).  After load_state_dict the entries are separate tensors again: they are re-shared here.
            shared, device = self._shared_step.get(gi), None
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                st = self._ensure_state(p)
                if shared is None:
                    shared = self._shared_step[gi] = st["step"] if isinstance(st["step"], torch.Tensor) else torch.tensor(float(st["step"]))
                if st["step"] is not shared:
                    st["step"] = shared
                device = p.device
            if device is None:
                continue
            shared += 1
            step_no = float(shared)
            b1, b2 = group["betas"]
            hp_vals = [group["lr"], b1, b2, group["eps"], group["weight_decay"], 1.0 - b1 ** step_no,
                       1.0 - b2 ** step_no, group["grad_scale"]]
            assert len(hp_vals) == ADAMW_HP_LEN
            dev = self._hp.get(gi)
            if dev is None:
                dev = torch.empty(ADAMW_HP_LEN, dtype=torch.float32, device=device)
                self._hp[gi] = dev
            dev.copy_(torch.tensor(hp_vals, dtype=torch.float32))   # pageable source: host-synchronous staging

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.prepare_step()
        for gi, group in enumerate(self.param_groups):
            entries = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("HipAdamW needs contiguous f32 parameters on a HIP device (no CPU fallback)")
                st = self._ensure_state(p)
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                entries.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            if not entries:
                continue
            if gi not in self._hp:
                raise RuntimeError("HipAdamW.step() under stream capture needs prepare_step() before the capture")
            K.journal_note([t for e in entries for t in e])     # the table carries these addresses
            K.adamw_step(self._table(gi, entries), self._hp[gi])
        return loss


__all__ = ["HipAdamW", "HipCrossEntropyLoss"]
