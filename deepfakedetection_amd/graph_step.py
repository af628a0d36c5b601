"""hipGraph replay of the trainers' loop body.

The reference's hot loop (trainers/efficientnet.py:290-309 and its two siblings) launches ~600 kernels per micro-batch
from Python; at the reference's fine-tune micro-batch of 32 the GPU finishes them long before the host has enqueued
them.  `GraphedTrainStep` captures the loop body once per (batch shape, role) and replays it:

    role "first"  zero_grad + forward + loss (/ accum_steps) + backward        (gradients land in the arena slots)
    role "next"   forward + loss + backward of a later micro-batch              (autograd adds into the slots)
    "step"        the fused AdamW launch                                        (hyper-parameters live in device memory:
                                                                                 HipAdamW.prepare_step() runs before each replay)

Protocol: the first `eager_cycles` optimizer cycles run eagerly — they build every lazily created buffer (derived-weight
caches, index maps, optimizer state and pointer tables, the Philox state) — and each graph is captured the first time its
(shape, role) shows up afterwards.  A capture records but does not execute, so nothing (BatchNorm statistics, gradient
accumulators, counters) is touched twice; the captured graph is then replayed for that very batch.  Inputs are copied
into static buffers; the loss of the last replay stays in a static tensor the caller reads only when it prints it.
Whatever does not fit — a ragged last batch beyond the few shapes kept, a capture failure, CPU runs, gradients outside
the arena — runs the ordinary eager path, so the trainers behave identically with GRAPH_STEP=0.  Everything random in
the step (dropout, drop-connect / DropPath) comes from Philox kernels whose state advances on the device
(kernels.DeviceRng), so replays draw fresh numbers.
"""

from __future__ import annotations

import warnings

import torch


class GraphedTrainStep:
    MAX_SHAPES = 4

    def __init__(self, model: torch.nn.Module, criterion, opt, *, accum_steps: int = 1, use_amp: bool = True,
                 eager_cycles: int = 1) -> None:
        self.model, self.criterion, self.opt = model, criterion, opt
        self.accum, self.use_amp = max(1, accum_steps), use_amp
        self.graphs: dict = {}          # (x shape, x dtype, y shape, role) -> (graph, static_x, static_y, static_loss)
        self.step_graph = None
        self.failed = False
        self.pool = None
        self.cycles_done = 0            # optimizer steps taken through this object
        self.eager_cycles = eager_cycles
        self.replays = 0

    # ------------------------------------------------------------------ the loop body (eager, and what gets captured)
    def _fwd_bwd(self, x, y, first: bool) -> torch.Tensor:
        if first:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.use_amp):
            loss = self.criterion(self.model(x), y)
            if self.accum > 1:
                loss = loss / self.accum
        loss.backward()
        return loss.detach()

    def _graphable(self, x: torch.Tensor) -> bool:
        if self.failed or not x.is_cuda or self.cycles_done < self.eager_cycles:
            return False
        return getattr(self.opt, "arena", None) is not None and hasattr(self.opt, "prepare_step")

    def _give_up(self, exc: Exception) -> None:
        self.failed = True
        torch.cuda.synchronize()
        warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running eagerly", stacklevel=3)

    # ------------------------------------------------------------------ public
    def micro_batch(self, x: torch.Tensor, y: torch.Tensor, first: bool) -> torch.Tensor:
        """forward + backward of one micro-batch; returns the (accumulation-scaled) loss as a device tensor."""
        if not self._graphable(x):
            return self._fwd_bwd(x, y, first)
        key = (tuple(x.shape), x.dtype, tuple(y.shape), "first" if first else "next")
        entry = self.graphs.get(key)
        if entry is None:
            if len(self.graphs) >= self.MAX_SHAPES:         # ragged tail batches: do not hoard graphs
                return self._fwd_bwd(x, y, first)
            try:
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                sx, sy = x.clone(), y.clone()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                    sloss = self._fwd_bwd(sx, sy, first)
                # after the capture every trainable parameter's .grad IS its arena slot; a replay rewrites the slots but
                # cannot re-attach them if Python code in between (zero_grad(set_to_none=True) at the start of an epoch)
                # dropped the references, so the views are kept and re-attached after each replay
                views = [(p, p.grad) for grp in self.opt.param_groups for p in grp["params"] if p.grad is not None]
                entry = self.graphs[key] = (g, sx, sy, sloss, views)
            except Exception as exc:  # noqa: BLE001 - any capture problem means: run eagerly from now on
                self._give_up(exc)
                return self._fwd_bwd(x, y, first)
        g, sx, sy, sloss, views = entry
        sx.copy_(x, non_blocking=True)
        sy.copy_(y, non_blocking=True)
        g.replay()
        self.replays += 1
        if views and views[0][0].grad is None:
            for p, gv in views:
                p.grad = gv
        return sloss

    def optimizer_step(self) -> None:
        self.cycles_done += 1
        arena = getattr(self.opt, "arena", None)
        if self.failed or self.cycles_done <= self.eager_cycles or arena is None or not arena.holds_all_grads():
            self.opt.step()                                 # eager (also: gradients outside the arena have no static address)
            return
        if self.step_graph is None:
            try:
                self.opt.prepare_step()                     # uploads this step's hyper-parameters; the capture reads them
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                    self.opt.step()
                self.step_graph = g
            except Exception as exc:  # noqa: BLE001
                self._give_up(exc)
                self.opt.step()
                return
            self.step_graph.replay()                        # the capture recorded, this executes the step just prepared
            return
        self.opt.prepare_step()
        self.step_graph.replay()


__all__ = ["GraphedTrainStep"]
