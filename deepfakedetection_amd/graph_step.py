"""hipGraph replay of the trainers' loop body and of the evaluate() forward.

The reference's hot loop (trainers/efficientnet.py:290-309 and its two siblings) launches ~300-900 kernels per micro-batch
from Python; at the reference's fine-tune micro-batch of 32 the GPU finishes them long before the host has enqueued
them.  `GraphedTrainStep` captures the loop body once per (batch shape, role) and replays it:

    role "first"  zero_grad + forward + loss (/ accum_steps) + backward        (gradients land in the arena slots)
    role "next"   forward + loss + backward of a later micro-batch              (autograd adds into the slots)
    [exchange]    data parallel: sum-all-reduce of the flat gradient arena     (RCCL, OUTSIDE of any capture)
    "step"        the fused AdamW launch                                        (hyper-parameters live in device memory:
                                                                                 HipAdamW.prepare_step() runs before each replay)

Protocol.  The first `eager_cycles` optimizer cycles run eagerly — they build every lazily created buffer (derived-weight
caches, optimizer state and pointer tables, the Philox state).  A (shape, role) key is run eagerly the FIRST time it
shows up (so that every batch-size-keyed cache — FasterViT's window maps, bias index maps — is built by ordinary code:
a pageable host-to-device copy is illegal under stream capture) and captured on its second occurrence.  A capture
records but does not execute, so nothing (BatchNorm statistics, gradient accumulators, counters) is touched twice; the
captured graph is then replayed for that very batch.  Inputs are copied into static buffers; the loss of the last
replay stays in a static tensor the caller reads only when it prints it.  Whatever does not fit — ragged shapes beyond
the few kept, a capture failure, CPU runs, gradients outside the arena — runs the ordinary eager path, so the trainers
behave identically with GRAPH_STEP=0.  Everything random in the step (dropout, drop-connect / DropPath) comes from
Philox kernels whose state advances on the device (kernels.DeviceRng), so replays draw fresh numbers.

Data parallel (`reducer`, dp.GradAllReducer): eager micro-batches that complete a cycle arm the reducer's hooks, so
their buckets leave from inside backward.  A REPLAYED micro-batch that completes a cycle is captured as SEVERAL graphs —
forward, then the backward cut into segments at module outputs chosen from the model's `dp_cut_modules()` so that every
gradient bucket becomes complete at the end of a segment — and replayed segment by segment: after each one the buckets
it completed are handed to RCCL, whose all-reduce runs on its own stream beside the segments still to come (VERDICT r3
item 6: one graph for the whole backward left the exchange un-overlapped — harmless for EfficientNet-B0's 16 MB, not for
FasterViT-0's 125 MB).  The cut is a forward hook that replaces a module's output by a detached leaf; the segments are
`loss.backward()` and then `output.backward(leaf.grad)` back to front — the same kernels in the same order as one
backward, so the result equals the eager step bit for bit.  (HIP 7.2 rejects external event-record nodes under stream
capture — scripts/probes/external_event_probe.py — so "one graph + events" is not available.)  Models without
`dp_cut_modules()` keep the single graph and the one-shot exchange after it.  Either way `optimizer_step()` waits for
the exchange and then runs (replays) AdamW with grad_scale = 1/world — bench.py --gpus N and the trainers drive this
same object.

By-address hazards.  A graph records raw addresses.  Every tensor whose address reached the library during a capture
is journalled (kernels.capture_journal); the ones owned by something outside the captured body (parameters, buffers,
derived-weight / BN-coefficient caches, index maps, optimizer state and tables, the Philox state) are checked before
EVERY replay: if one was freed or moved (a cache rebuilt after .to(), a regrown table) the replay raises `StaleGraphError`
instead of reading or writing memory it no longer owns.
"""

from __future__ import annotations

import os
import warnings

import torch

from . import kernels as K


class StaleGraphError(RuntimeError):
    """A captured hipGraph refers to device memory that has been freed or replaced since the capture."""


def _check_guard(guard, what: str) -> None:
    bad = K.stale_entries(guard)
    if bad:
        raise StaleGraphError(f"{what}: {len(bad)} recorded tensor(s) changed since the capture, e.g. {bad[0]}; "
                              "rebuild the GraphedTrainStep / GraphedForward after moving or re-creating model state")


class _Cuts:
    """While active, the chosen modules' outputs are replaced by detached leaves: (output, leaf) pairs in forward order."""

    def __init__(self, modules) -> None:
        self.modules, self.pairs, self._handles = list(modules), [], []

    def _hook(self, module, inputs, output):
        if not isinstance(output, torch.Tensor) or not output.requires_grad:
            return None
        leaf = output.detach().requires_grad_(True)
        self.pairs.append((output, leaf))
        return leaf

    def __enter__(self):
        self._handles = [m.register_forward_hook(self._hook) for m in self.modules]
        return self

    def __exit__(self, *exc) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []


def plan_cuts(model: torch.nn.Module, reducer) -> tuple[list, dict[int, int]]:
    """(modules to cut after, id(parameter) -> backward segment index) so that each gradient bucket of `reducer` is complete
    at the end of one segment.  Candidates come from model.dp_cut_modules() (forward order, single-tensor outputs); parameters are
    taken in registration order, which is the forward order for the engine's models (the reducer's buckets rely on it too).
    Segment s holds the parameters between cut s and cut s + 1; the backward runs the segments from the highest down."""
    cands = list(getattr(model, "dp_cut_modules", lambda: [])())
    params = [p for p in model.parameters() if p.requires_grad]
    index = {id(p): i for i, p in enumerate(params)}
    if not cands or reducer is None or len(reducer.buckets) < 2:
        return [], {id(p): 0 for p in params}
    # position of a candidate: one past the last parameter registered inside it (or before it)
    pos = []
    for m in cands:
        inside = [index[id(p)] for p in m.parameters() if id(p) in index]
        pos.append(max(inside) + 1 if inside else 0)
    chosen: list[int] = []
    for bucket in reducer.buckets[:-1]:                          # the last bucket ends at parameter 0: nothing left to overlap
        lo = min(index[id(p)] for p in bucket)
        best = max((j for j in range(len(cands)) if 0 < pos[j] <= lo), key=lambda j: pos[j], default=None)
        if best is not None and best not in chosen:
            chosen.append(best)
    chosen.sort()
    cut_pos = [pos[j] for j in chosen]
    seg_of = {id(p): sum(1 for c in cut_pos if c <= i) for p, i in ((p, index[id(p)]) for p in params)}
    return [cands[j] for j in chosen], seg_of


class GraphedTrainStep:
    MAX_SHAPES = 4

    def __init__(self, model: torch.nn.Module, criterion, opt, *, accum_steps: int = 1, use_amp: bool = True,
                 eager_cycles: int = 1, reducer=None) -> None:
        self.model, self.criterion, self.opt = model, criterion, opt
        self.accum, self.use_amp = max(1, accum_steps), use_amp
        self.graphs: dict = {}          # (x shape, x dtype, y shape, role) -> (graph, static_x, static_y, static_loss, views, guard)
        self.seen: set = set()          # keys that have run eagerly once (every lazily built cache of that shape exists)
        self.step_graph = None
        self.step_guard: list = []
        self.failed = False
        self.pool = None
        self.cycles_done = 0            # optimizer steps taken through this object
        self.eager_cycles = eager_cycles
        self.replays = 0
        self.reducer = reducer if (reducer is not None and getattr(reducer, "world", 1) > 1) else None
        self.cut_modules, self.needs = [], []
        if self.reducer is not None and os.environ.get("DFD_DP_SEGMENTS", "1") != "0":
            self.cut_modules, seg_of = plan_cuts(model, self.reducer)
            if self.cut_modules:
                params = {id(p) for b in self.reducer.buckets for p in b}
                self.needs = self.reducer.bucket_needs({k: v for k, v in seg_of.items() if k in params})
        self.segmented_replays = 0

    # ------------------------------------------------------------------ the loop body (eager, and what gets captured)
    def _fwd_bwd(self, x, y, first: bool, arm: bool = False) -> torch.Tensor:
        if first:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.use_amp):
            loss = self.criterion(self.model(x), y)
            if self.accum > 1:
                loss = loss / self.accum
        if arm and self.reducer is not None:
            self.reducer.arm()          # eager backward that completes the cycle: buckets leave as they fill
        loss.backward()
        return loss.detach()

    def _capture_segments(self, sx, sy, first: bool):
        """forward | backward of the last segment | ... | backward of the first segment, one hipGraph each (one shared pool: they are
        replayed in this order, so memory freed by one and reused by the next is reused in the same order at replay)."""
        graphs = []
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
            if first:
                self.opt.zero_grad(set_to_none=True)
            with _Cuts(self.cut_modules) as cuts:
                with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.use_amp):
                    loss = self.criterion(self.model(sx), sy)
                    if self.accum > 1:
                        loss = loss / self.accum
        graphs.append(g)
        if len(cuts.pairs) != len(self.cut_modules):
            raise RuntimeError(f"{len(self.cut_modules)} cut modules produced {len(cuts.pairs)} cuts (a module ran twice, or returned no tensor)")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
            loss.backward()
        graphs.append(g)
        for out, leaf in reversed(cuts.pairs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                out.backward(leaf.grad)
            graphs.append(g)
        self._seg_keep = getattr(self, "_seg_keep", []) + [cuts.pairs]      # the leaves and their .grad tensors are graph memory in use
        return graphs, loss.detach()

    def _graphable(self, x: torch.Tensor) -> bool:
        if self.failed or not x.is_cuda:
            return False
        return getattr(self.opt, "arena", None) is not None and hasattr(self.opt, "prepare_step")

    def _give_up(self, exc: Exception) -> None:
        self.failed = True
        torch.cuda.synchronize()
        if os.environ.get("DFD_GRAPH_DEBUG"):
            import traceback

            traceback.print_exception(type(exc), exc, exc.__traceback__)
        warnings.warn(f"hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); running eagerly", stacklevel=3)

    # ------------------------------------------------------------------ public
    def micro_batch(self, x: torch.Tensor, y: torch.Tensor, first: bool, last: bool | None = None) -> torch.Tensor:
        """forward + backward of one micro-batch; returns the (accumulation-scaled) loss as a device tensor.
        `last`: this micro-batch completes an optimizer cycle (default: accum_steps == 1) — only used to overlap the
        data-parallel exchange with an eager backward."""
        last = self.accum == 1 if last is None else last
        if not self._graphable(x):
            return self._fwd_bwd(x, y, first, arm=last)
        segmented = bool(self.cut_modules) and last
        key = (tuple(x.shape), x.dtype, tuple(y.shape), "first" if first else "next", segmented)
        entry = self.graphs.get(key)
        if entry is None:
            fresh = key not in self.seen
            self.seen.add(key)
            if fresh or self.cycles_done < self.eager_cycles or len(self.graphs) >= self.MAX_SHAPES:
                return self._fwd_bwd(x, y, first, arm=last)       # first sight / warm-up cycles / do not hoard graphs
            try:
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                sx, sy = x.clone(), y.clone()
                torch.cuda.synchronize()
                with K.capture_journal() as notes:
                    if segmented:
                        g, sloss = self._capture_segments(sx, sy, first)
                    else:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                            sloss = self._fwd_bwd(sx, sy, first)
                # after the capture every trainable parameter's .grad IS its arena slot; a replay rewrites the slots but
                # cannot re-attach them if Python code in between (zero_grad(set_to_none=True) at the start of an epoch)
                # dropped the references, so the views are kept and re-attached after each replay
                views = [(p, p.grad) for grp in self.opt.param_groups for p in grp["params"] if p.grad is not None]
                guard = K.journal_guard(notes, x.device)
                entry = self.graphs[key] = (g, sx, sy, sloss, views, guard)
            except Exception as exc:  # noqa: BLE001 - any capture problem means: run eagerly from now on
                self._give_up(exc)
                return self._fwd_bwd(x, y, first, arm=last)
        g, sx, sy, sloss, views, guard = entry
        _check_guard(guard, "training-step graph")
        sx.copy_(x, non_blocking=True)
        sy.copy_(y, non_blocking=True)
        if isinstance(g, list):
            # forward, then the backward segments from the back; behind each one the buckets it completed start their exchange
            self.reducer.begin_cycle()
            g[0].replay()
            top = len(g) - 2                                    # segment index of the first backward graph
            for j, seg_graph in enumerate(g[1:]):
                seg_graph.replay()
                if j < top:                                     # (the buckets of the last segment leave in optimizer_step -> finish())
                    self.reducer.launch_completed(self.needs, top - j)
            self.segmented_replays += 1
        else:
            g.replay()
        self.replays += 1
        if views and views[0][0].grad is None:
            for p, gv in views:
                p.grad = gv
        arena = getattr(self.opt, "arena", None)
        if arena is not None:
            arena.mark_written()        # the slots hold this cycle's gradients: a later eager micro-batch must ADD to them
        return sloss

    def _exchange(self) -> None:
        if self.reducer is not None:
            self.reducer.finish()       # armed hooks: wait for the buckets; otherwise the one-shot arena all-reduce

    def optimizer_step(self) -> None:
        self.cycles_done += 1
        self._exchange()
        arena = getattr(self.opt, "arena", None)
        if self.failed or self.cycles_done <= self.eager_cycles or arena is None or not arena.holds_all_grads():
            self.opt.step()                                 # eager (also: gradients outside the arena have no static address)
            return
        if self.step_graph is None:
            try:
                self.opt.prepare_step()                     # uploads this step's hyper-parameters; the capture reads them
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with K.capture_journal() as notes:
                    with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                        self.opt.step()
                self.step_graph, self.step_guard = g, K.journal_guard(notes, arena.flat.device)
            except Exception as exc:  # noqa: BLE001
                self._give_up(exc)
                self.opt.step()
                return
            self.step_graph.replay()                        # the capture recorded, this executes the step just prepared
            return
        _check_guard(self.step_guard, "optimizer-step graph")
        self.opt.prepare_step()
        self.step_graph.replay()


class GraphedForward:
    """evaluate()'s forward (eval mode, f32 or autocast, no gradient) replayed per batch shape — trainers/efficientnet.py
    :237-262 runs ~150-450 small launches per validation batch, host-bound below batch ~32.  Same protocol as the training
    step: a shape runs eagerly on first sight, is captured on the second, falls back to eager on any problem; the
    returned logits are a static tensor that the NEXT call overwrites (evaluate() reduces them to counters at once)."""

    MAX_SHAPES = 4

    def __init__(self, model: torch.nn.Module, amp_dtype: torch.dtype | None = None) -> None:
        self.model, self.amp_dtype = model, amp_dtype
        self.graphs: dict = {}
        self.seen: set = set()
        self.pool = None
        self.failed = False
        self.replays = 0

    def _fwd(self, x: torch.Tensor) -> torch.Tensor:
        if self.amp_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
                return self.model(x)
        return self.model(x)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.failed or not x.is_cuda or self.model.training or torch.is_grad_enabled():
            return self._fwd(x)
        key = (tuple(x.shape), x.dtype, tuple(x.stride()))
        entry = self.graphs.get(key)
        if entry is None:
            fresh = key not in self.seen
            self.seen.add(key)
            if fresh or len(self.graphs) >= self.MAX_SHAPES:
                return self._fwd(x)
            try:
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                # ordinary (non-inference) mode for the capture itself: torch's graph machinery creates bookkeeping tensors
                # (the generator's extra-graph seed / offset) on first use and updates them in place on every later
                # capture — born under inference_mode they would make every later capture outside of it fail
                with torch.inference_mode(False), torch.no_grad():
                    sx = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
                    sx.copy_(x)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with K.capture_journal() as notes:
                        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                            out = self._fwd(sx)
                entry = self.graphs[key] = (g, sx, out, K.journal_guard(notes, x.device))
            except Exception as exc:  # noqa: BLE001
                self.failed = True
                torch.cuda.synchronize()
                warnings.warn(f"hipGraph capture of the eval forward failed ({type(exc).__name__}: {exc}); running eagerly", stacklevel=2)
                return self._fwd(x)
        g, sx, out, guard = entry
        _check_guard(guard, "eval-forward graph")
        sx.copy_(x, non_blocking=True)
        g.replay()
        self.replays += 1
        return out


__all__ = ["GraphedForward", "GraphedTrainStep", "StaleGraphError"]
