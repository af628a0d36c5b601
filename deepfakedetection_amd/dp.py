"""Single-node data parallelism for the hot loop: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU path at all (no torch.distributed call site, SURVEY.md
section 2a); BASELINE.json asks for ImageFolder minibatches sharded across the 8 GPUs
of one node with gradient all-reduce.  Design for MI355X:

  * samples are independent (no SyncBN in the reference, per-replica BN statistics), so
    the ONLY exchange per optimizer step is a sum-all-reduce of the gradients:
    EfficientNet-B0 (2 classes) = 4,010,110 f32 = 16 MB.
  * xGMI is a point-to-point mesh (7 links/GPU), so few large messages beat many small
    ones: gradients are packed into a handful of flat buckets (default 8 MiB) and each
    bucket is one collective.  The mean is folded into AdamW's grad_scale (1/world), so
    no extra division pass runs.
  * after the all-reduce each parameter's .grad is re-pointed at its slice of the flat
    bucket (views, no copy back).
  * initial parameters and buffers are broadcast from rank 0; BN running statistics stay
    per-replica during training (rank 0's are what a checkpoint records).

Backends: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""

from __future__ import annotations

import math
import os
from collections.abc import Iterable, Iterator

import torch
import torch.distributed as dist


def env_rank() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # DFD_DIST_BACKEND=gloo lets several ranks share ONE GPU (rehearsals on a one-GPU box;
            # RCCL refuses two ranks on a device)
            backend = os.environ.get("DFD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def rccl_log_request(tag: str = "dfd") -> str | None:
    """Ask RCCL to write its INIT / tuning log to a per-process file (call BEFORE init_process_group); returns the path this
    process will write, or None when the caller already configured NCCL_DEBUG itself.  The bench line records what RCCL chose
    (channels, ring / tree, protocol) so that a multi-GPU run leaves that evidence next to its number."""
    if os.environ.get("NCCL_DEBUG"):
        return None
    import tempfile

    pattern = os.path.join(tempfile.gettempdir(), f"{tag}_rccl_%h_%p.log")
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,TUNING,ENV")
    os.environ["NCCL_DEBUG_FILE"] = pattern
    import socket

    return pattern.replace("%h", socket.gethostname()).replace("%p", str(os.getpid()))


def rccl_log_summary(path: str | None, limit: int = 12) -> dict | None:
    """The lines of an RCCL INFO log that say what it will do: version, channel count, ring / tree graphs, algorithm and
    protocol enablement, threshold tuning.  Never raises: a missing or unreadable file yields {"note": ...}."""
    if not path:
        return None
    try:
        with open(path, errors="replace") as fh:
            lines = fh.read().splitlines()
    except OSError as exc:
        return {"note": f"no RCCL log ({type(exc).__name__})"}
    keys = ("RCCL version", "NCCL version", "Channel", "channels", "Ring ", "Trees", "Tree ", "Algo", "Proto", "threadThresholds", "xgmi", "XGMI",
            "P2P", "NCCL_ALGO", "NCCL_PROTO", "comm 0x")
    picked, seen = [], set()
    for ln in lines:
        body = ln.split("NCCL INFO", 1)[-1].strip()
        if any(k in body for k in keys) and body not in seen:
            seen.add(body)
            picked.append(body[:160])
        if len(picked) >= limit:
            break
    return {"log_lines": len(lines), "selected": picked}


def broadcast_module_state(module: torch.nn.Module, src: int = 0) -> None:
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    # one collective per (dtype, device) instead of one per tensor (B0: 360 tensors; VERDICT r3 "what's weak" 9)
    groups: dict = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for tensors in groups.values():
            flat = torch.cat([t.detach().reshape(-1) for t in tensors])
            dist.broadcast(flat, src=src)
            if dist.get_rank() != src:
                at = 0
                for t in tensors:
                    n = t.numel()
                    t.copy_(flat[at:at + n].view_as(t))
                    at += n


class GradAllReducer:
    """Bucketed gradient all-reduce (sum); pair with an optimizer whose grad_scale = 1/world.

    Two ways to drive it:
      * ``reduce()`` after ``backward()``: every bucket is launched at once (used between the two
        captured hipGraphs of bench.py, where backward itself is one graph replay);
      * ``attach()`` once, then ``arm()`` before and ``finish()`` after the ``backward()`` that
        completes an optimizer step (the last micro-batch of an accumulation cycle): a
        post-accumulate-grad hook per parameter counts a bucket down and launches its all-reduce
        the moment its last gradient has been produced, so RCCL traffic of the late layers' buckets
        runs on RCCL's stream while the backward kernels of the early layers are still executing
        (the eager training loop).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 8 << 20, arena=None) -> None:
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # a bucket leaves as soon as its last backward kernel has been launched: weight-gradient sums may not wait for a carrier launch
        from . import kernels as _K
        _K.passenger_sums_enabled = False
        self.arena = arena                      # a GradArena (arena.py): flat, copy-free path
        self.bucket_elems = max(1, bucket_bytes // 4)
        # reverse order: the last layers' gradients are ready first in backward
        order = list(reversed(self.params))
        self.buckets: list[list[torch.nn.Parameter]] = []
        cur: list[torch.nn.Parameter] = []
        size = 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        # arena range [lo, hi) of each bucket: parameters are consecutive (reversed) arena slots
        self._ranges: list[tuple[int, int]] | None = None
        if arena is not None and [id(p) for p in arena.params] == [id(p) for p in self.params]:
            index = {id(p): i for i, p in enumerate(arena.params)}
            sizes = [s.numel() for s in arena.slots]
            self._ranges = []
            for bucket in self.buckets:
                idx = [index[id(p)] for p in bucket]
                lo, hi = min(idx), max(idx)
                end = arena.offsets[hi] + (sizes[hi] + 3) // 4 * 4
                self._ranges.append((arena.offsets[lo], min(end, arena.flat.numel())))
        self._bucket_of = {id(p): b for b, bucket in enumerate(self.buckets) for p in bucket}
        self._left: list[int] = []
        self._next = 0                          # buckets are launched strictly in index order (see _on_grad)
        self._pending: list[tuple] = []
        self._hooks: list = []
        self._armed = False
        self.launched_early = 0                 # buckets whose all-reduce started inside backward (diagnostic)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    # ------------------------------------------------------------------ overlapped path
    def attach(self) -> None:
        """Register the per-parameter hooks (idempotent)."""
        if self._hooks or self.world == 1:
            return
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def arm(self) -> None:
        """The next backward() completes the gradients of this step: let the hooks launch buckets."""
        self._next = 0
        if self._hooks:
            self._left = [len(b) for b in self.buckets]
            self._armed = True

    def detach(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _sync_for_gloo(self, t: torch.Tensor) -> None:
        # gloo stages device tensors through the host: hand it finished data (one-GPU rehearsals and tests only;
        # RCCL orders the collective after the producing kernels on the stream by itself)
        if t.is_cuda and dist.get_backend() == "gloo":
            torch.cuda.synchronize()

    def _launch(self, b: int, early: bool) -> None:
        """One collective per bucket, ALWAYS the same one whoever calls: the hook path (eager backward), finish() and reduce()
        (after a replayed backward) all come here, in bucket order.  A rank that fell back to eager micro-batches (a capture
        failure on one GPU only) therefore posts exactly the collectives its replaying peers post — ADVICE r3: with two different
        chunkings the ranks' all-reduces no longer matched."""
        bucket = self.buckets[b]
        arena = self.arena
        live = [p for p in bucket if p.grad is not None]
        if self._ranges is not None and arena is not None:
            if all(p.grad.data_ptr() == arena.slots[self._slot_index(p)].data_ptr() for p in live):
                # copy-free: the bucket IS a range of the flat arena (slots of parameters without a gradient this
                # step travel along; nobody reads them)
                lo, hi = self._ranges[b]
                chunk = arena.flat[lo:hi]
                self._sync_for_gloo(chunk)
                self._pending.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True), None, None))
                if early:
                    self.launched_early += 1
                return
            if live[0].is_cuda:
                raise RuntimeError("GradAllReducer: a gradient of an arena parameter lives outside its arena slot; the "
                                   "data-parallel hot loop has no flatten/copy path on the GPU (was .grad replaced by hand?)")
        if not live:
            return                              # (which parameters have gradients is the same on every rank)
        # no arena (CPU runs of the plumbing, optimizers other than HipAdamW): flatten, reduce, re-point
        flat = torch.cat([p.grad.reshape(-1).float() for p in live])
        self._sync_for_gloo(flat)
        self._pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, live))
        if early:
            self.launched_early += 1

    def _slot_index(self, p: torch.nn.Parameter) -> int:
        if not hasattr(self, "_slot_idx"):
            self._slot_idx = {id(q): i for i, q in enumerate(self.arena.params)} if self.arena is not None else {}
        return self._slot_idx[id(p)]

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._armed:
            return
        b = self._bucket_of[id(p)]
        self._left[b] -= 1
        # in index order only: a bucket that completes before a lower one waits for it, so every rank posts the same sequence
        # whatever order autograd produced the gradients in (and the same sequence as reduce())
        with torch.no_grad():
            while self._next < len(self.buckets) and self._left[self._next] == 0:
                self._launch(self._next, early=True)
                self._next += 1

    # ------------------------------------------------------------------ segmented replay (graph_step.GraphedTrainStep)
    def bucket_needs(self, seg_of: dict[int, int]) -> list[int]:
        """seg_of: id(parameter) -> index of the backward segment that produces its gradient (higher = runs earlier).  Returns,
        per bucket, the lowest segment it waits for: the bucket is complete once the segments down to that one have run."""
        return [min(seg_of[id(p)] for p in bucket) for bucket in self.buckets]

    def begin_cycle(self) -> None:
        """A replayed backward is about to run segment by segment: buckets will be launched by launch_completed()."""
        self._next = 0

    @torch.no_grad()
    def launch_completed(self, needs: list[int], seg: int) -> None:
        """Backward segments down to `seg` have been enqueued: launch, in index order, every bucket they complete — its all-reduce
        then runs on RCCL's stream beside the segments still to come (the overlap a single replayed graph cannot give)."""
        if self.world == 1:
            return
        while self._next < len(self.buckets) and needs[self._next] >= seg:
            self._launch(self._next, early=True)
            self._next += 1

    @torch.no_grad()
    def finish(self) -> None:
        """After backward(): launch what the hooks / the segment loop did not (parameters without gradient this step, the
        first segment's buckets), wait for every bucket, re-point gradients at the reduced flats where a copy was made."""
        if self.world == 1:
            return
        self._armed = False
        for b in range(self._next, len(self.buckets)):
            self._launch(b, early=False)
        self._next = 0
        self._drain()

    def _drain(self) -> None:
        for work, flat, live in self._pending:
            work.wait()
            if flat is not None:
                at = 0
                for p in live:
                    n = p.numel()
                    p.grad = flat[at:at + n].view_as(p)
                    at += n
        self._pending = []

    # ------------------------------------------------------------------ one-shot path
    @torch.no_grad()
    def reduce(self) -> None:
        """Sum the gradients over ranks, in place of each parameter's .grad (every bucket, in index order, at once)."""
        if self.world == 1:
            return
        self._next = 0
        self.finish()


class ShardedSampler:
    """Rank-strided shard of a seeded per-epoch permutation.  pad=True (training): padded so every rank sees the
    same number of samples (the contract of torch's DistributedSampler — the gradient all-reduce needs equal step
    counts).  pad=False (validation): every sample is seen exactly once over all ranks, shards may differ by one
    (evaluate() has no collective inside its loop, so unequal lengths are fine and the metrics stay unbiased)."""

    def __init__(self, length: int, rank: int, world: int, shuffle: bool = True, seed: int = 0, drop_last: bool = False,
                 pad: bool = True) -> None:
        self.length, self.rank, self.world, self.shuffle, self.seed, self.drop_last = length, rank, world, shuffle, seed, drop_last
        self.pad = pad
        self.epoch = 0
        if not pad and not drop_last:
            self.per_rank = len(range(rank, length, world))
        else:
            self.per_rank = length // world if drop_last else math.ceil(length / world)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.per_rank

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.length, generator=g).tolist()
        else:
            order = list(range(self.length))
        if not self.pad and not self.drop_last:
            return iter(order[self.rank::self.world])
        total = self.per_rank * self.world
        if self.drop_last:
            order = order[:total]
        elif len(order) < total:
            order += order[: total - len(order)]
        return iter(order[self.rank:total:self.world])


def all_reduce_counts(*values: float, device: torch.device | str = "cpu") -> list[float]:
    """Sum scalars (correct / total / loss sums of `evaluate`) over ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


__all__ = ["GradAllReducer", "ShardedSampler", "all_reduce_counts", "broadcast_module_state", "env_rank", "init_distributed",
           "rccl_log_request", "rccl_log_summary"]
