"""Gradient arena: one flat f32 buffer that the backward kernels write parameter
gradients into directly.

Why: (1) the weight-gradient kernels then need no extra copy or accumulate pass —
autograd adopts the returned view as `param.grad` as is; (2) every gradient has a FIXED
address, so the fused AdamW's pointer table is built once (and stays valid inside a
captured hipGraph, where pinned-memory uploads are not permitted); (3) data-parallel
training all-reduces the flat buffer in a few large xGMI-friendly messages with no
flatten/unflatten copies.

Protocol: a slot may be written directly once per accumulation cycle.  The first backward
after `reset()` (called by HipAdamW.zero_grad) gets the slot; further micro-batches of a
gradient-accumulation cycle get `None`, allocate an ordinary tensor and let autograd add
it into the slot in place.
"""

from __future__ import annotations

import weakref

import torch

# parameter address -> (weak reference to the arena that owns a slot for it, slot index).  Weak: an arena lives as long
# as its optimizer; a dead optimizer must neither keep its model's parameters alive nor answer for a new tensor that
# happens to get the same address.
_registry: dict[int, tuple["weakref.ReferenceType[GradArena]", int]] = {}


class GradArena:
    def __init__(self, params: list[torch.nn.Parameter]) -> None:
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradArena needs at least one trainable parameter")
        dev = self.params[0].device
        self.offsets: list[int] = []
        total = 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4            # keep every slot 16-byte aligned
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slots = [self.flat[o:o + p.numel()] for o, p in zip(self.offsets, self.params)]
        self.written = [False] * len(self.params)
        me = weakref.ref(self)
        for i, p in enumerate(self.params):
            _registry[p.data_ptr()] = (me, i)

    def reset(self) -> None:
        self.written = [False] * len(self.params)

    def mark_written(self) -> None:
        """Every slot holds this cycle's gradient (a hipGraph replay wrote them behind Python's back): a later eager
        micro-batch of the same cycle must allocate its own tensor and let autograd ADD it into the slot."""
        self.written = [True] * len(self.params)

    def holds_all_grads(self) -> bool:
        """True when every parameter's .grad IS its slot (static pointer table valid)."""
        return all(p.grad is not None and p.grad.data_ptr() == s.data_ptr() for p, s in zip(self.params, self.slots))

    def release(self) -> None:
        for p in self.params:
            hit = _registry.get(p.data_ptr())
            if hit is not None and hit[0]() is self:
                del _registry[p.data_ptr()]

    def __del__(self) -> None:
        try:
            for ptr in [k for k, (ref, _) in _registry.items() if ref() is None or ref() is self]:
                del _registry[ptr]
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def grad_dest(param_ptr: int, shape) -> torch.Tensor | None:
    """The arena slot for the parameter at `param_ptr`, viewed as `shape`, if it may be
    written directly now; None otherwise."""
    hit = _registry.get(param_ptr)
    if hit is None:
        return None
    arena, i = hit[0](), hit[1]
    if arena is None:
        del _registry[param_ptr]
        return None
    if arena.written[i]:
        return None
    arena.written[i] = True
    return arena.slots[i].view(shape)


__all__ = ["GradArena", "grad_dest"]
