"""Torch-tensor front end of the C ABI (include/dfd_hip.h).

Every function here takes NHWC tensors ([N, H, W, C] contiguous, f32 or bf16) that
live on a HIP device, allocates outputs with torch (device memory is torch's job),
and enqueues exactly the kernels of libdfd_hip.so on torch's current stream.  There
is no fallback: a CPU tensor or a missing library raises.
"""

from __future__ import annotations

import contextlib
import functools
import ctypes
import threading
import weakref
from dataclasses import dataclass

import os

import torch

from . import _lib
from ._lib import (
    ACT_NONE, ACT_SILU, BF16, F32, MAX_PARTIALS, PRO_AFFINE2, PRO_BN_ACT, PRO_BN_ACT_GATE, PRO_NONE,
    DwShape, Prologue, StemShape, check,
)

_scratch: dict[tuple[int, int, str], torch.Tensor] = {}


def _L():
    return _lib.load()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- optional weight-gradient side stream (DFD_SIDE_STREAM=1; OFF by default).  Inside one block's
# backward the weight-gradient kernels depend on the data-gradient chain but nothing in that chain
# depends on them, so they can run on a second HIP stream — a parallel branch of the captured hipGraph —
# joined before the block's backward returns (every tensor the side kernels read is then still
# referenced: no record_stream; the shared wgrad scratch is never used by two blocks at once).
# Measured on MI355X: 16.99 ms/step with it vs 16.48 without — every large kernel here is a persistent
# grid sized for the whole chip, so two of them at once just take turns; kept for experiments only.
_side_streams: dict[int, torch.cuda.Stream] = {}
_side_enabled = os.environ.get("DFD_SIDE_STREAM", "0") == "1"
_side_max_rows = int(os.environ.get("DFD_SIDE_MAXROWS", "0"))      # experiments: only layers with at most this many rows


class side_stream:
    """`with side_stream(rows): ...` enqueues the body on the side stream, ordered after everything already
    enqueued on the current stream; pair with `join_side()` before the results are consumed."""

    def __init__(self, rows: int = 0) -> None:
        self.rows = rows

    def __enter__(self):
        self._ctx = None
        if not _side_enabled or _profile_sink is not None:
            return self
        if _side_max_rows and self.rows > _side_max_rows:
            return self
        cur = torch.cuda.current_stream()
        side = _side_streams.get(cur.device_index)
        if side is None:
            side = _side_streams[cur.device_index] = torch.cuda.Stream(device=cur.device)
        side.wait_stream(cur)
        self._ctx = torch.cuda.stream(side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def join_side() -> None:
    cur = torch.cuda.current_stream()
    side = _side_streams.get(cur.device_index)
    if side is not None:
        cur.wait_stream(side)


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


# ---- capture journal (graph_step.GraphedTrainStep / GraphedForward).  A hipGraph records raw addresses.  While a capture
# is open every tensor whose address goes to the library is noted here (weak reference to the tensor — or to its base
# when it is a view — plus the address).  After the capture the owner keeps the entries that (a) are still alive, i.e.
# are owned by something outside the captured body (parameters, buffers, caches, the optimizer's state and tables, the
# Philox state), and (b) do not sit in the graph's private memory pool; before every replay it asserts that each of
# them still lives at the recorded address (`stale_entries`).  A cache that replaced or dropped a recorded tensor
# (dtype switch, regrown buffer, rebuilt table) then raises instead of letting the replay read or write freed memory.
_journal: dict | None = None


def journal_note(t) -> None:
    """Record `t` (tensor, or an iterable of tensors / None) in the open capture journal; no-op when none is open."""
    j = _journal
    if j is None or t is None:
        return
    if not isinstance(t, torch.Tensor):
        for u in t:
            journal_note(u)
        return
    if not t.is_cuda:
        return
    base = t._base if t._base is not None else t
    key = id(base)
    if key not in j:
        try:
            j[key] = (weakref.ref(base), base.data_ptr(), base.numel() * base.element_size(), tuple(base.shape), base.dtype)
        except TypeError:
            pass


@contextlib.contextmanager
def capture_journal():
    """`with capture_journal() as notes:` around a stream capture; `notes` is filled while the body runs."""
    global _journal
    if _journal is not None:
        raise RuntimeError("nested capture journals")
    notes: dict = {}
    _journal = notes
    try:
        yield notes
    finally:
        _journal = None


def _pool_ranges(device) -> list[tuple[int, int]]:
    """Address ranges of the caching allocator's private (graph) pools on `device`."""
    out = []
    idx = torch.device(device).index
    for seg in torch.cuda.memory_snapshot():
        if seg.get("device") == (idx if idx is not None else torch.cuda.current_device()) and tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
            out.append((seg["address"], seg["address"] + seg["total_size"]))
    return out


def journal_guard(notes: dict, device) -> list[tuple]:
    """The journal entries a replay depends on: still referenced after the capture, outside the graph pools."""
    pools = _pool_ranges(device)
    keep = []
    for ref, ptr, nbytes, shape, dtype in notes.values():
        if ref() is None or nbytes == 0:
            continue
        if any(lo <= ptr < hi for lo, hi in pools):
            continue
        keep.append((ref, ptr, nbytes, shape, dtype))
    return keep


def stale_entries(guard: list[tuple]) -> list[str]:
    """Descriptions of the guarded tensors that are gone or have moved (empty list: the graph may replay)."""
    bad = []
    for ref, ptr, nbytes, shape, dtype in guard:
        t = ref()
        if t is None:
            bad.append(f"{dtype} {shape} at {ptr:#x} was freed")
        elif t.data_ptr() != ptr or t.numel() * t.element_size() != nbytes:
            bad.append(f"{dtype} {shape} moved {ptr:#x} -> {t.data_ptr():#x}")
    return bad


def _p(t: torch.Tensor | None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("dfd kernels need tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"dfd kernels need contiguous tensors, got shape {tuple(t.shape)} strides {t.stride()}")
    if _journal is not None:
        journal_note(t)
    return t.data_ptr()


def _chk_nhwc(t: torch.Tensor) -> None:
    if t.dim() != 4 or not t.is_contiguous():
        raise ValueError("expected a contiguous [N, H, W, C] tensor")


def scratch(device: torch.device, name: str, nbytes: int) -> torch.Tensor:
    """A grow-only f32 scratch buffer per (device, stream, purpose): reuse is ordered by the stream it belongs to,
    so two models / threads on different streams never share a buffer (the C ABI itself is re-entrant).

    While the stream is being captured into a hipGraph the buffer is a fresh allocation instead: it comes from the
    graph's private pool, which lives exactly as long as the graph.  A cached buffer would be recorded by address and
    could later be replaced (grown) and freed by code outside the graph while replays still write to it."""
    need = (nbytes + 3) // 4
    if torch.cuda.is_current_stream_capturing():
        buf = torch.empty(max(need, 1), dtype=torch.float32, device=device)
        if _sum_batch.open:
            _sum_batch.keep.append(buf)                     # the recorded sums read it at sum_batch's exit
        return buf
    if _sum_batch.open:
        # every weight gradient of an open batch keeps its own workspace until the batch is summed; with the sums riding along as
        # passengers of later launches (sum_batch) that is up to two batches later: three generations of names
        name = f"{name}#{_sum_batch.gen}#{_sum_batch.count}"
        _sum_batch.count += 1
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, name)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1), dtype=torch.float32, device=device)
        _scratch[key] = buf
    return buf


class _SumBatchState(threading.local):
    """Per host thread, like the library's own batch state (dfd_sum_batch_begin is thread-local)."""

    def __init__(self) -> None:
        self.open, self.count, self.keep, self.gen, self.temp_dest = False, 0, [], 0, False


_sum_batch = _SumBatchState()
_DEFER_SUMS = os.environ.get("DFD_DEFER_SUMS", "1") != "0"      # A/B switch: 0 = every weight gradient sums its slab at once


@contextlib.contextmanager
def sum_batch():
    """Weight gradients computed inside the block leave their final partial-row summation to the block's exit, where
    one pair of launches adds them all (dfd_sum_batch_begin / _end): the returned gradient tensors are valid only after
    the block.  Nothing inside may read them, and nothing else that sums partial rows may run inside.  No-op when
    nested or when the side stream is enabled (the sums would be launched on the wrong stream)."""
    if _sum_batch.open or _side_enabled or not _DEFER_SUMS:
        yield
        return
    check(_L().dfd_sum_batch_begin(), "dfd_sum_batch_begin")
    _sum_batch.open, _sum_batch.count, _sum_batch.temp_dest = True, 0, False
    try:
        yield
    finally:
        _sum_batch.open = False
        # (only when every gradient of the batch went straight into its arena slot, which autograd adopts as .grad without reading it)
        if PASSENGER_SUMS and passenger_sums_enabled and not _sum_batch.temp_dest and _register_backward_flush():
            # the sums are read by the optimizer only: instead of two launches at the end of every block they ride along as extra
            # workgroups of the next two act_bn_bwd launches (one per block) and whatever is left is launched at the end of the
            # backward pass (csrc/dfd_dwconv.hip, dfd_sum_batch_end_deferred)
            rc = _L().dfd_sum_batch_end_deferred()
            with _passenger_lock:
                _passenger_keep.extend(_sum_batch.keep)     # (captured runs: the slabs live until the flush)
            _sum_batch.gen = (_sum_batch.gen + 1) % 3
        else:
            rc = _L().dfd_sum_batch_end()
        _sum_batch.keep.clear()
        check(rc, "dfd_sum_batch_end")


# ---- sums as passengers (see sum_batch).  DFD_PASSENGER_SUMS=0: two launches at the end of every block (A/B switch);
# passenger_sums_enabled is cleared by dp.GradAllReducer: a gradient bucket may be handed to RCCL as soon as its last backward kernel
# has been launched, which a sum still waiting for its carrier would not be part of.
PASSENGER_SUMS = os.environ.get("DFD_PASSENGER_SUMS", "1") != "0"
passenger_sums_enabled = True
_passenger_lock = threading.Lock()
_passenger_keep: list = []
_passenger_flush_registered = False


def flush_passengers() -> None:
    """Launch every sum that still waits for a carrier on the current stream (the end of a backward pass; also safe to call at any
    time on the stream the backward ran on)."""
    global _passenger_flush_registered
    with _passenger_lock:
        _passenger_flush_registered = False
        keep = list(_passenger_keep)
        _passenger_keep.clear()
    check(_L().dfd_sum_passengers_flush(_stream()), "dfd_sum_passengers_flush")
    del keep


def _register_backward_flush() -> bool:
    """flush_passengers as an end-of-backward callback of the autograd engine, once per backward pass; False outside of one (the caller
    then sums at once)."""
    global _passenger_flush_registered
    with _passenger_lock:
        if _passenger_flush_registered:
            return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(flush_passengers)
    except RuntimeError:
        return False
    # the first batch of this backward pass: nothing may be waiting (the previous pass flushed); after a pass that died in an exception
    # something is — drop it, its workspaces may be gone
    check(_L().dfd_sum_passengers_discard(), "dfd_sum_passengers_discard")
    with _passenger_lock:
        _passenger_keep.clear()
        _passenger_flush_registered = True
    return True


def batched_sums(fn):
    """Decorator for a block's `backward`: runs it inside sum_batch()."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with sum_batch():
            return fn(*args, **kwargs)
    return wrapper


def _nbytes(t: torch.Tensor | None) -> int:
    return 0 if t is None else t.numel() * t.element_size()


def _dst(out: torch.Tensor | None, shape, device) -> torch.Tensor:
    """A caller-provided destination (gradient-arena slot) or a fresh f32 tensor."""
    if out is None:
        # a fresh tensor is handed to autograd, which adds it to (or clones it into) .grad as soon as the Function returns: the open
        # batch's sums may then not wait for a carrier launch (sum_batch)
        _sum_batch.temp_dest = True
        return torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f"bad gradient destination {tuple(out.shape)} for {tuple(shape)}")
    return out


def partials_buf(device: torch.device, C: int) -> torch.Tensor:
    # + 40 rows: room for the second stage of dfd_sum_rows when a slab is summed in place
    return scratch(device, "partials", (MAX_PARTIALS + 40) * 2 * C * 4)


@dataclass
class BNParams:
    """The tensors of one BatchNorm2d plus its hyper-parameters."""

    weight: torch.Tensor
    bias: torch.Tensor
    running_mean: torch.Tensor
    running_var: torch.Tensor
    momentum: float
    eps: float
    conv_bias: torch.Tensor | None = None      # bias of the producing convolution (timm ConvNorm), folded into the BN
    ls: torch.Tensor | None = None             # LayerScale gamma applied to the BN output, folded into (scale, shift)


# ------------------------------------------------------------------ BatchNorm
def bn_finalize(partials: torch.Tensor, nparts: int, count: int, bn: BNParams, update_running: bool = True) -> torch.Tensor:
    C = bn.weight.numel()
    state = torch.empty((4, C), dtype=torch.float32, device=partials.device)
    rm = bn.running_mean if update_running else None
    rv = bn.running_var if update_running else None
    check(_L().dfd_bn_finalize_ex(_p(partials), nparts, C, float(count), _p(bn.weight), _p(bn.bias), _p(bn.conv_bias),
                                  _p(bn.ls), _p(rm), _p(rv), bn.momentum, bn.eps, _p(state), _stream()), "dfd_bn_finalize_ex")
    return state


def bn_eval_coeffs(bn: BNParams) -> torch.Tensor:
    C = bn.weight.numel()
    state = torch.empty((4, C), dtype=torch.float32, device=bn.weight.device)
    check(_L().dfd_bn_eval_coeffs_ex(_p(bn.weight), _p(bn.bias), _p(bn.conv_bias), _p(bn.ls), _p(bn.running_mean),
                                     _p(bn.running_var), bn.eps, C, _p(state), _stream()), "dfd_bn_eval_coeffs_ex")
    return state


def bn_bwd_finalize_ex(partials: torch.Tensor, nparts: int, count: int, gamma: torch.Tensor, beta: torch.Tensor | None,
                       ls: torch.Tensor | None, state: torch.Tensor, train: bool, want_bn: bool = True, want_ls: bool = False,
                       want_bias: bool = False, outs=(None, None, None, None)):
    """BatchNorm backward coefficients with LayerScale / convolution-bias gradients.
    Returns (coef, dgamma, dbeta, dls, dbias); outs: optional gradient-arena slots in that order."""
    C = gamma.numel()
    dev = gamma.device
    coef = torch.empty((3, C), dtype=torch.float32, device=dev)
    dgamma = _dst(outs[0], (C,), dev) if want_bn else None
    dbeta = _dst(outs[1], (C,), dev) if want_bn else None
    dls = _dst(outs[2], (C,), dev) if (want_ls and ls is not None) else None
    dbias = _dst(outs[3], (C,), dev) if want_bias else None
    check(_L().dfd_bn_bwd_finalize_ex(_p(partials), nparts, C, float(count), _p(gamma), _p(beta), _p(ls), _p(state), int(train),
                                      _p(dgamma), _p(dbeta), _p(dls), _p(dbias), 0, _p(coef), _stream()), "dfd_bn_bwd_finalize_ex")
    return coef, dgamma, dbeta, dls, dbias


def bn_bwd_finalize(partials: torch.Tensor, nparts: int, count: int, gamma: torch.Tensor, state: torch.Tensor,
                    train: bool, want_param_grads: bool = True, out_dgamma: torch.Tensor | None = None,
                    out_dbeta: torch.Tensor | None = None):
    """out_*: optional destinations (gradient-arena slots) written instead of fresh tensors."""
    C = gamma.numel()
    coef = torch.empty((3, C), dtype=torch.float32, device=gamma.device)
    dgamma = _dst(out_dgamma, (C,), gamma.device) if want_param_grads else None
    dbeta = _dst(out_dbeta, (C,), gamma.device) if want_param_grads else None
    check(_L().dfd_bn_bwd_finalize(_p(partials), nparts, C, float(count), _p(gamma), _p(state), int(train),
                                   _p(dgamma), _p(dbeta), 0, _p(coef), _stream()), "dfd_bn_bwd_finalize")
    return coef, dgamma, dbeta


def bn_act_apply(y: torch.Tensor, state: torch.Tensor, act: int, residual: torch.Tensor | None = None,
                 row_scale: torch.Tensor | None = None) -> torch.Tensor:
    _chk_nhwc(y)
    N, H, W, C = y.shape
    out = torch.empty_like(y)
    check(_L().dfd_bn_act_apply(_dt(y), _p(y), _p(state), act, _p(residual), _p(row_scale), _p(out), N, H * W, C,
                                _stream()), "dfd_bn_act_apply", str(tuple(y.shape)))
    return out


def bn_bwd_reduce(g: torch.Tensor, y: torch.Tensor, state: torch.Tensor, row_scale: torch.Tensor | None = None):
    _chk_nhwc(y)
    N, H, W, C = y.shape
    parts = partials_buf(y.device, C)
    n = ctypes.c_int(0)
    check(_L().dfd_bn_bwd_reduce(_dt(y), _p(g), _p(y), _p(state), _p(row_scale), N, H * W, C, _p(parts), MAX_PARTIALS,
                                 ctypes.byref(n), _stream()), "dfd_bn_bwd_reduce", str(tuple(y.shape)))
    return parts, n.value


_UNSUPPORTED = -2


def gemm_bias_act(x: torch.Tensor, w_nk, state: torch.Tensor, act: int, residual: torch.Tensor | None = None,
                  row_scale: torch.Tensor | None = None, want_raw: bool = False):
    """Linear layer without statistics in one kernel: act(scale * (x w^T) + shift) [* row_scale] [+ residual] -> (out, raw y or
    None), bit-identical to pwconv + bn_act_apply; None when the shape is not the fused kernel's (the caller runs the pair)."""
    if x.dtype != torch.bfloat16 or isinstance(w_nk, MxWeight) or not _FUSE_LINEAR:
        return None
    _chk_nhwc(x)
    N_, H, W, Kd = x.shape
    M, Nout = N_ * H * W, w_nk.shape[0]
    out = torch.empty((N_, H, W, Nout), dtype=x.dtype, device=x.device)
    raw = torch.empty_like(out) if want_raw else None
    rc = _L().dfd_gemm_bias_act(_dt(x), _p(x), _p(w_nk), M, Kd, Nout, _p(state), act, _p(residual), _p(row_scale), H * W,
                                _p(raw), _p(out), _stream())
    if rc == _UNSUPPORTED:
        return None
    check(rc, "dfd_gemm_bias_act", f"{tuple(x.shape)} -> {Nout}")
    return out, raw


_FUSE_LINEAR = os.environ.get("DFD_FUSE_LINEAR", "1") != "0"      # A/B switch


def bias_grad(g: torch.Tensor, row_scale: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Sum of g [N,H,W,C] over its rows (times row_scale[n]) -> f32 [C]: the bias gradient of a Linear layer.  Inside
    sum_batch() the final summation is deferred to the block's batch like a weight gradient's (valid after the block)."""
    _chk_nhwc(g)
    N, H, W, C = g.shape
    nbytes = _L().dfd_bias_grad_ws(N, H * W, C)
    ws = scratch(g.device, "bias_ws", nbytes)
    db = _dst(out, (C,), g.device)
    check(_L().dfd_bias_grad(_dt(g), _p(g), _p(row_scale), N, H * W, C, _p(db), 0, _p(ws), ws.numel() * 4, _stream()),
          "dfd_bias_grad", str(tuple(g.shape)))
    return db


def act_bn_bwd(D: torch.Tensor | None, y: torch.Tensor, gate: torch.Tensor | None, dpool: torch.Tensor | None,
               state: torch.Tensor, act: int, se_job: tuple | None = None):
    """se_job: what se_bwd(defer_wgrad=True) returned — the block's squeeze-excite FC weight gradients then ride along as extra
    workgroups of this launch (dfd_act_bn_bwd_se) instead of being a launch of their own."""
    _chk_nhwc(y)
    N, H, W, C = y.shape
    dz = torch.empty_like(y)
    parts = partials_buf(y.device, C)
    n = ctypes.c_int(0)
    if se_job is not None:
        pooled, ws, R, dw1, db1, dw2, db2 = se_job
        check(_L().dfd_act_bn_bwd_se(_dt(y), _p(D), _p(y), _p(gate), _p(dpool), _p(state), act, _p(dz), N, H * W, C, _p(parts),
                                     MAX_PARTIALS, ctypes.byref(n), _p(pooled), _p(ws), R, _p(dw1), _p(db1), _p(dw2), _p(db2), 0,
                                     _stream()), "dfd_act_bn_bwd_se", str(tuple(y.shape)))
        return dz, parts, n.value
    check(_L().dfd_act_bn_bwd(_dt(y), _p(D), _p(y), _p(gate), _p(dpool), _p(state), act, _p(dz), N, H * W, C, _p(parts),
                              MAX_PARTIALS, ctypes.byref(n), _stream()), "dfd_act_bn_bwd", str(tuple(y.shape)))
    return dz, parts, n.value




def _pool_workspace(y: torch.Tensor, N: int, HW: int, C: int) -> torch.Tensor | None:
    """The pooling workspace (partial vectors of the H*W splits) as a tensor the caller keeps alive across the launch;
    None when the shape needs none."""
    nbytes = int(_L().dfd_pool_ws(_dt(y), N, HW, C))
    if nbytes == 0:
        return None
    return scratch(y.device, "pool_ws", nbytes)


def pool_act(y: torch.Tensor, state: torch.Tensor, act: int) -> torch.Tensor:
    _chk_nhwc(y)
    N, H, W, C = y.shape
    pooled = torch.empty((N, C), dtype=torch.float32, device=y.device)
    ws = _pool_workspace(y, N, H * W, C)
    check(_L().dfd_pool_act(_dt(y), _p(y), _p(state), act, _p(pooled), N, H * W, C, _p(ws), _nbytes(ws), _stream()), "dfd_pool_act")
    return pooled


def pool_bwd_reduce(D: torch.Tensor, y: torch.Tensor, state: torch.Tensor, act: int) -> torch.Tensor:
    _chk_nhwc(y)
    N, H, W, C = y.shape
    dgate = torch.empty((N, C), dtype=torch.float32, device=y.device)
    ws = _pool_workspace(y, N, H * W, C)
    check(_L().dfd_pool_bwd_reduce(_dt(y), _p(D), _p(y), _p(state), act, _p(dgate), N, H * W, C, _p(ws), _nbytes(ws), _stream()),
          "dfd_pool_bwd_reduce")
    return dgate


def scale_rows(x: torch.Tensor, row_scale: torch.Tensor) -> torch.Tensor:
    _chk_nhwc(x)
    N, H, W, C = x.shape
    out = torch.empty_like(x)
    check(_L().dfd_scale_rows(_dt(x), _p(x), _p(row_scale), _p(out), N, H * W, C, _stream()), "dfd_scale_rows")
    return out


# ------------------------------------------------------------------ input tail
def resize_crop_u8(flat_u8: torch.Tensor, jobs_u8: torch.Tensor, n: int, oh: int, ow: int, max_shrink: int) -> torch.Tensor:
    """Decoded uint8 RGB images packed back to back (`flat_u8`) + n 56-byte dfd_resize_job descriptors (`jobs_u8`, uint8 view)
    -> uint8 [n, oh, ow, 3]: Pillow-exact bilinear resize of each image's box, cropped to the output window."""
    if flat_u8.dtype != torch.uint8 or jobs_u8.dtype != torch.uint8 or jobs_u8.numel() != 56 * n:
        raise ValueError("resize_crop_u8: expected uint8 pixel bytes and n 56-byte job descriptors")
    out = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=flat_u8.device)
    check(_L().dfd_resize_crop_u8(_p(flat_u8), _p(jobs_u8), _p(out), n, oh, ow, int(max_shrink), _stream()), "dfd_resize_crop_u8",
          f"n={n} out={oh}x{ow} shrink<={max_shrink}")
    return out


def augment_u8(src_u8: torch.Tensor, jobs_i32: torch.Tensor) -> torch.Tensor:
    """uint8 [N, H, W, 3] on the device + one 16-int job per picture (data.GpuInputTail.sample_augment) -> the rotated and
    colour-jittered uint8 batch, byte-exact with the PIL transforms of data.py (csrc/dfd_augment.hip)."""
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3 or not src_u8.is_contiguous():
        raise ValueError("expected a contiguous uint8 [N, H, W, 3] tensor")
    N, H, W, _ = src_u8.shape
    if jobs_i32.dtype != torch.int32 or jobs_i32.numel() != 16 * N or not jobs_i32.is_contiguous():
        raise ValueError("augment_u8: expected N jobs of 16 int32 each")
    out = torch.empty_like(src_u8)
    check(_L().dfd_augment_u8(_p(src_u8), _p(jobs_i32), _p(out), N, H, W, _stream()), "dfd_augment_u8", f"{tuple(src_u8.shape)}")
    return out


def image_prep(src_u8: torch.Tensor, mean, std, flip: torch.Tensor | None, erase: torch.Tensor | None) -> torch.Tensor:
    """uint8 [N, H, W, 3] on the device -> f32, returned as an [N, 3, H, W] channels_last view of the
    NHWC result (zero-copy: exactly what HipEfficientNet.forward turns back into NHWC)."""
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3 or not src_u8.is_contiguous():
        raise ValueError("expected a contiguous uint8 [N, H, W, 3] tensor")
    N, H, W, _ = src_u8.shape
    dst = torch.empty((N, H, W, 3), dtype=torch.float32, device=src_u8.device)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    if flip is not None and (flip.dtype != torch.uint8 or flip.numel() != N):
        raise ValueError("flip must be uint8 [N]")
    if erase is not None and (erase.dtype != torch.int32 or erase.numel() != 4 * N):
        raise ValueError("erase must be int32 [N, 4]")
    check(_L().dfd_image_prep(_p(src_u8), _p(dst), N, H, W, m3, s3, _p(flip), _p(erase), _stream()), "dfd_image_prep")
    return dst.permute(0, 3, 1, 2)


# ------------------------------------------------------------------ squeeze-excite
def se_fc_fwd(pooled: torch.Tensor, w1, b1, w2, b2, act: int, w2t: torch.Tensor | None = None):
    """w2t given: the [R, C] copy of w2 is already up to date (DerivedWeights) and the transpose is skipped."""
    N, C = pooled.shape
    R = w1.shape[0]
    hpre = torch.empty((N, R), dtype=torch.float32, device=pooled.device)
    gate = torch.empty((N, C), dtype=torch.float32, device=pooled.device)
    ready = w2t is not None
    if not ready:
        w2t = torch.empty((R, C), dtype=torch.float32, device=pooled.device)
    check(_L().dfd_se_fc_fwd(_p(pooled), _p(w1), _p(b1), None if ready else _p(w2), _p(b2), N, C, R, act, _p(hpre), _p(gate),
                             _p(w2t), _stream()), "dfd_se_fc_fwd", f"C={C} R={R}")
    return hpre, gate, w2t


def se_fc_bwd(dgate, gate, hpre, pooled, w1, w2t, act: int, want_param_grads: bool = True, outs=(None, None, None, None)):
    """w2t: the [R, C] transpose returned by se_fc_fwd."""
    N, C = pooled.shape
    R = w1.shape[0]
    dev = pooled.device
    dpooled = torch.empty((N, C), dtype=torch.float32, device=dev)
    ws = scratch(dev, "se_ws", (N * C + 2 * N * R) * 4)
    if want_param_grads:
        dw1 = _dst(outs[0], (R, C), dev)
        db1 = _dst(outs[1], (R,), dev)
        dw2 = _dst(outs[2], (C, R), dev)
        db2 = _dst(outs[3], (C,), dev)
    else:
        dw1 = db1 = dw2 = db2 = None
    check(_L().dfd_se_fc_bwd(_p(dgate), _p(gate), _p(hpre), _p(pooled), _p(w1), _p(w2t), N, C, R, act, _p(dpooled),
                             _p(dw1), _p(db1), _p(dw2), _p(db2), 0, _p(ws), _stream()), "dfd_se_fc_bwd")
    return dpooled, dw1, db1, dw2, db2


def se_fwd(y: torch.Tensor, state: torch.Tensor, act_in: int, w1, b1, w2, b2, act: int, w2t: torch.Tensor | None = None):
    """pool_act + se_fc_fwd in two launches (dfd_se_fwd): returns pooled, hpre, gate, w2t."""
    _chk_nhwc(y)
    N, H, W, C = y.shape
    R = w1.shape[0]
    dev = y.device
    pooled = torch.empty((N, C), dtype=torch.float32, device=dev)
    hpre = torch.empty((N, R), dtype=torch.float32, device=dev)
    gate = torch.empty((N, C), dtype=torch.float32, device=dev)
    ready = w2t is not None
    if not ready:
        w2t = torch.empty((R, C), dtype=torch.float32, device=dev)
    ws = _pool_workspace(y, N, H * W, C)
    check(_L().dfd_se_fwd(_dt(y), _p(y), _p(state), act_in, N, H * W, C, _p(w1), _p(b1), None if ready else _p(w2), _p(b2), R,
                          act, _p(pooled), _p(hpre), _p(gate), _p(w2t), _p(ws), _nbytes(ws), _stream()), "dfd_se_fwd", f"C={C} R={R}")
    return pooled, hpre, gate, w2t


def se_bwd(D, y, state, act_in: int, gate, hpre, pooled, w1, w2t, act: int, want_param_grads: bool = True,
           outs=(None, None, None, None), defer_wgrad: bool = False):
    """pool_bwd_reduce + se_fc_bwd in three launches (dfd_se_bwd): returns dpooled, dw1, db1, dw2, db2.
    defer_wgrad (with want_param_grads): the third launch — the FC weight gradients, which only the optimizer reads — is left out
    and a sixth value is returned, the job for act_bn_bwd(se_job=...), whose launch then carries it; dw1 .. db2 are valid after
    that call.  The caller must make it its NEXT launch (the job points into this call's scratch workspace)."""
    _chk_nhwc(y)
    N, H, W, C = y.shape
    R = w1.shape[0]
    dev = y.device
    dpooled = torch.empty((N, C), dtype=torch.float32, device=dev)
    dgate = scratch(dev, "se_dgate", N * C * 4)
    ws = scratch(dev, "se_ws", (N * C + 2 * N * R) * 4)
    if want_param_grads:
        dw1 = _dst(outs[0], (R, C), dev)
        db1 = _dst(outs[1], (R,), dev)
        dw2 = _dst(outs[2], (C, R), dev)
        db2 = _dst(outs[3], (C,), dev)
    else:
        dw1 = db1 = dw2 = db2 = None
    pws = _pool_workspace(y, N, H * W, C)
    defer = defer_wgrad and want_param_grads
    check(_L().dfd_se_bwd(_dt(y), _p(D), _p(y), _p(state), act_in, N, H * W, C, _p(gate), _p(hpre), _p(pooled), _p(w1), _p(w2t),
                          R, act, _p(dgate), _p(dpooled), None if defer else _p(dw1), None if defer else _p(db1),
                          None if defer else _p(dw2), None if defer else _p(db2), 0, _p(pws), _nbytes(pws), _p(ws),
                          _stream()), "dfd_se_bwd")
    if defer_wgrad:
        return dpooled, dw1, db1, dw2, db2, ((pooled, ws, R, dw1, db1, dw2, db2) if defer else None)
    return dpooled, dw1, db1, dw2, db2


# ------------------------------------------------------------------ depthwise
def _dw_shape(x_shape, Ho: int, Wo: int, k: int, stride: int, pad_top: int, pad_left: int) -> DwShape:
    N, H, W, C = x_shape
    return DwShape(N, H, W, C, Ho, Wo, k, stride, pad_top, pad_left)


def dwconv_fwd(x: torch.Tensor, in_state: torch.Tensor | None, in_act: int, w: torch.Tensor, k: int, stride: int,
               pad_top: int, pad_left: int, Ho: int, Wo: int, stats: bool = True):
    """y = dwconv(act(bn(x))); w is torch's [C,1,k,k] f32 weight. Returns (y, partials, nparts)."""
    _chk_nhwc(x)
    N, H, W, C = x.shape
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    shp = _dw_shape(x.shape, Ho, Wo, k, stride, pad_top, pad_left)
    parts = partials_buf(x.device, C) if stats else None
    n = ctypes.c_int(0)
    check(_L().dfd_dwconv_fwd(_dt(x), _p(x), _p(in_state), in_act, _p(w), _p(y), ctypes.byref(shp), _p(parts),
                              MAX_PARTIALS, ctypes.byref(n), _stream()), "dfd_dwconv_fwd", f"{tuple(x.shape)} k{k}s{stride}")
    return y, parts, n.value


def dwconv_bwd_data(dz: torch.Tensor, y: torch.Tensor | None, coef: torch.Tensor | None, w: torch.Tensor,
                    xin: torch.Tensor | None, in_state: torch.Tensor | None, in_act: int, in_shape, k: int, stride: int,
                    pad_top: int, pad_left: int):
    """Returns (dzin, partials, nparts); xin=None: plain input gradient, no statistics."""
    N, H, W, C = in_shape
    Ho, Wo = dz.shape[1], dz.shape[2]
    dzin = torch.empty((N, H, W, C), dtype=dz.dtype, device=dz.device)
    shp = _dw_shape(in_shape, Ho, Wo, k, stride, pad_top, pad_left)
    parts = partials_buf(dz.device, C) if xin is not None else None
    n = ctypes.c_int(0)
    check(_L().dfd_dwconv_bwd_data(_dt(dz), _p(dz), _p(y), _p(coef), _p(w), _p(xin), _p(in_state), in_act, _p(dzin),
                                   ctypes.byref(shp), _p(parts), MAX_PARTIALS, ctypes.byref(n), _stream()),
          "dfd_dwconv_bwd_data", f"{tuple(in_shape)} k{k}s{stride}")
    return dzin, parts, n.value


def dwconv_bwd_weight(dz: torch.Tensor, y: torch.Tensor | None, coef: torch.Tensor | None, xin: torch.Tensor,
                      in_state: torch.Tensor | None, in_act: int, k: int, stride: int, pad_top: int, pad_left: int,
                      out: torch.Tensor | None = None) -> torch.Tensor:
    N, H, W, C = xin.shape
    Ho, Wo = dz.shape[1], dz.shape[2]
    shp = _dw_shape(xin.shape, Ho, Wo, k, stride, pad_top, pad_left)
    nbytes = _L().dfd_dwconv_bwd_weight_ws(ctypes.byref(shp))
    ws = scratch(dz.device, "wgrad_ws", nbytes)
    dw = _dst(out, (C, 1, k, k), dz.device)
    check(_L().dfd_dwconv_bwd_weight(_dt(dz), _p(dz), _p(y), _p(coef), _p(xin), _p(in_state), in_act, _p(dw),
                                     ctypes.byref(shp), 0, _p(ws), ws.numel() * 4, _stream()), "dfd_dwconv_bwd_weight")
    return dw


# measured on MI355X: the fused kernel takes 176 us where the two separate kernels take 73 + 91 us on
# EfficientFormerV2-S1 (19.6 vs 19.3 ms/step) and is neutral on EfficientNet-B0: the two phases are issue-bound, not bandwidth-
# bound, so sharing the staging buys less than the larger working set costs.  Kept (tested) behind DFD_FUSE_DW_BWD=1.
_FUSE_DW_BWD = os.environ.get("DFD_FUSE_DW_BWD", "0") == "1"


def dwconv_bwd_fused_ok(k: int, stride: int) -> bool:
    """The one-kernel backward (csrc/dfd_dwbwdf.hip) covers 3x3 stride-1 layers."""
    return _FUSE_DW_BWD and k == 3 and stride == 1


def dwconv_bwd_fused(dz: torch.Tensor, y: torch.Tensor, coef: torch.Tensor, w: torch.Tensor, xin: torch.Tensor,
                     in_state: torch.Tensor, in_act: int, k: int, stride: int, pad_top: int, pad_left: int,
                     out_w: torch.Tensor | None = None):
    """dwconv_bwd_data (with its epilogue) and dwconv_bwd_weight (with its prologue) from ONE staging of (dz, y, xin).
    Returns (dzin, partials, nparts, dw)."""
    N, H, W, C = xin.shape
    Ho, Wo = dz.shape[1], dz.shape[2]
    shp = _dw_shape(xin.shape, Ho, Wo, k, stride, pad_top, pad_left)
    dzin = torch.empty((N, H, W, C), dtype=dz.dtype, device=dz.device)
    parts = partials_buf(dz.device, C)
    nbytes = _L().dfd_dwconv_bwd_weight_ws(ctypes.byref(shp))
    ws = scratch(dz.device, "wgrad_ws", nbytes)
    dw = _dst(out_w, (C, 1, k, k), dz.device)
    n = ctypes.c_int(0)
    check(_L().dfd_dwconv_bwd_fused(_dt(dz), _p(dz), _p(y), _p(coef), _p(w), _p(xin), _p(in_state), in_act, _p(dzin), _p(dw),
                                    ctypes.byref(shp), _p(parts), MAX_PARTIALS, ctypes.byref(n), 0, _p(ws), ws.numel() * 4, _stream()),
          "dfd_dwconv_bwd_fused", f"{tuple(xin.shape)} k{k}s{stride}")
    return dzin, parts, n.value, dw


# ------------------------------------------------------------------ pointwise
def _pro(mode: int = PRO_NONE, act: int = ACT_NONE, HW: int = 1, a2=None, coef=None, gate=None) -> Prologue:
    pr = Prologue(mode, act, HW, 0, _p(a2), _p(coef), _p(gate))
    pr._keep = (a2, coef, gate)       # the struct holds raw pointers: keep the tensors alive with it
    pr._nbytes = sum(t.numel() * t.element_size() for t in pr._keep if t is not None)
    return pr


def pro_bn_act(state: torch.Tensor, act: int) -> Prologue:
    return _pro(PRO_BN_ACT, act, 1, None, state, None)


def pro_bn_act_gate(state: torch.Tensor, act: int, gate: torch.Tensor, HW: int) -> Prologue:
    return _pro(PRO_BN_ACT_GATE, act, HW, None, state, gate)


_identity_states: dict = {}


def identity_state(device: torch.device, C: int) -> torch.Tensor:
    """float[4][C] BatchNorm state of the identity map (scale 1, shift 0, mean 0, rstd 1): turns the BN_ACT_GATE prologue into
    a gate-only one for operands that were stored activated."""
    key = (device.type, device.index, C)
    st = _identity_states.get(key)
    if st is None:
        with torch.inference_mode(False):               # a plain tensor: the cache outlives the caller's inference_mode block
            st = torch.zeros((4, C), dtype=torch.float32, device=device)
            st[0].fill_(1.0)
            st[3].fill_(1.0)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _identity_states[key] = st          # a tensor born inside a capture belongs to that graph's pool: not cached
    return st


def pro_affine2(a2: torch.Tensor, coef: torch.Tensor) -> Prologue:
    return _pro(PRO_AFFINE2, ACT_NONE, 1, a2, coef, None)


def prep_weights(w: torch.Tensor, dtype: torch.dtype, want_nk: bool = True, want_kn: bool = True):
    """f32 master [Nout, K(,1,1)] -> (w_nk, w_kn) in the activation dtype."""
    Nn, K = w.shape[0], w.shape[1]
    w_nk = torch.empty((Nn, K), dtype=dtype, device=w.device) if want_nk else None
    w_kn = torch.empty((K, Nn), dtype=dtype, device=w.device) if want_kn else None
    code = BF16 if dtype == torch.bfloat16 else F32
    check(_L().dfd_pw_prep_weights(code, _p(w), _p(w_nk), _p(w_kn), Nn, K, _stream()), "dfd_pw_prep_weights")
    return w_nk, w_kn


class DerivedWeights:
    """Every per-step derived weight of a network, refreshed by ONE batched launch per forward pass:
    the activation-dtype copies [N][K] / [K][N] of the 1x1 convolution weights and the [R][C] copies of the
    squeeze-excite expand weights.  `items`: (f32 source, want_nk, want_kn, is_se) in a fixed order; the
    destination buffers are allocated once and handed out by index."""

    def __init__(self, items: list[tuple[torch.Tensor, bool, bool, bool]], dtype: torch.dtype) -> None:
        from ._lib import PrepJob

        self.dtype = dtype
        self.sources = [src for src, _, _, _ in items]
        self.ptrs = [src.data_ptr() for src in self.sources]
        self.out: list[tuple[torch.Tensor | None, torch.Tensor | None]] = []
        jobs = (PrepJob * len(items))()
        for i, (src, want_nk, want_kn, is_se) in enumerate(items):
            n, kdim = src.shape[0], src.shape[1]
            dt = torch.float32 if is_se else dtype
            nk = torch.empty((n, kdim), dtype=dt, device=src.device) if want_nk else None
            kn = torch.empty((kdim, n), dtype=dt, device=src.device) if want_kn else None
            self.out.append((nk, kn))
            jobs[i] = PrepJob(src.data_ptr(), nk.data_ptr() if nk is not None else None,
                              kn.data_ptr() if kn is not None else None, n, kdim, BF16 if dt == torch.bfloat16 else F32, 0)
        self._jobs = jobs

    def valid_for(self, sources: list[torch.Tensor], dtype: torch.dtype) -> bool:
        return dtype == self.dtype and len(sources) == len(self.ptrs) and all(
            s.data_ptr() == p for s, p in zip(sources, self.ptrs))

    def refresh(self) -> None:
        if _journal is not None:                # the job table carries raw addresses: sources and destinations
            journal_note(self.sources)
            journal_note([t for pair in self.out for t in pair])
        check(_L().dfd_prep_weights_multi(self._jobs, len(self._jobs), _stream()), "dfd_prep_weights_multi")


# ---- MX fp8 weights (include/dfd_hip.h "MX fp8"): FasterViT's Linear layers with `fp8_weights` on ------------------------
class MxWeight:
    """One Linear weight in OCP MX fp8: q [N][K] e4m3fn bytes + scale [N][K/32] e8m0 bytes.  kernels.pwconv() takes it in
    place of the bf16 [N][K] copy and then runs quantise-activations + dfd_mx_gemm."""

    __slots__ = ("q", "scale", "N", "K")

    def __init__(self, q: torch.Tensor, scale: torch.Tensor) -> None:
        self.q, self.scale = q, scale
        self.N, self.K = q.shape

    @property
    def shape(self):
        return (self.N, self.K)

    @property
    def dtype(self):
        return torch.uint8


class MxWeights:
    """Every fp8 weight of a network, re-quantised from the f32 masters by ONE batched launch per forward pass
    (dfd_mx_quant_weights_multi) together with the dequantised [K][N] copies the bf16 backward multiplies by."""

    def __init__(self, sources: list[torch.Tensor], dtype: torch.dtype) -> None:
        from ._lib import MxJob

        self.dtype = dtype
        self.sources = list(sources)
        self.ptrs = [s.data_ptr() for s in self.sources]
        self.out: list[tuple[MxWeight, torch.Tensor]] = []
        jobs = (MxJob * len(self.sources))()
        for i, src in enumerate(self.sources):
            n, kdim = src.shape
            if kdim % 128:
                raise ValueError(f"MX fp8 weights need K % 128 == 0, got {tuple(src.shape)}")
            q = torch.empty((n, kdim), dtype=torch.uint8, device=src.device)
            sc = torch.empty((n, kdim // 32), dtype=torch.uint8, device=src.device)
            kn = torch.empty((kdim, n), dtype=dtype, device=src.device)
            self.out.append((MxWeight(q, sc), kn))
            jobs[i] = MxJob(src.data_ptr(), q.data_ptr(), sc.data_ptr(), kn.data_ptr(), n, kdim, _code(dtype), 0)
        self._jobs = jobs

    def valid_for(self, sources: list[torch.Tensor], dtype: torch.dtype) -> bool:
        return dtype == self.dtype and len(sources) == len(self.ptrs) and all(s.data_ptr() == p for s, p in zip(sources, self.ptrs))

    def refresh(self) -> None:
        if _journal is not None:
            journal_note(self.sources)
            journal_note([t for w, kn in self.out for t in (w.q, w.scale, kn)])
        check(_L().dfd_mx_quant_weights_multi(self._jobs, len(self._jobs), _stream()), "dfd_mx_quant_weights_multi")


def mx_quant_weight(w: torch.Tensor, dtype: torch.dtype = torch.bfloat16):
    """(MxWeight, dequantised [K][N] copy) of one f32 [N][K] weight."""
    mw = MxWeights([w], dtype)
    mw.refresh()
    return mw.out[0]


def mx_quant_rows(a: torch.Tensor, pro: Prologue | None = None):
    """a [..., K] (bf16 / f32) -> (q uint8 [M][K], scale uint8 [M][K/32]) after the optional BN+activation prologue."""
    Kd = a.shape[-1]
    M = a.numel() // Kd
    q = torch.empty((M, Kd), dtype=torch.uint8, device=a.device)
    sc = torch.empty((M, Kd // 32), dtype=torch.uint8, device=a.device)
    check(_L().dfd_mx_quant_rows(_dt(a), _p(a), ctypes.byref(pro) if pro is not None else None, _p(q), _p(sc), M, Kd, _stream()),
          "dfd_mx_quant_rows", f"M={M} K={Kd}")
    return q, sc


def mx_gemm(aq: torch.Tensor, ascale: torch.Tensor, w: MxWeight, out_dtype: torch.dtype, out_shape=None) -> torch.Tensor:
    M, Kd = aq.shape
    if Kd != w.K:
        raise ValueError(f"mx_gemm: K mismatch {Kd} vs {w.K}")
    out = torch.empty(out_shape if out_shape is not None else (M, w.N), dtype=out_dtype, device=aq.device)
    check(_L().dfd_mx_gemm(_p(aq), _p(ascale), _p(w.q), _p(w.scale), _code(out_dtype), _p(out), M, Kd, w.N, _stream()),
          "dfd_mx_gemm", f"M={M} K={Kd} N={w.N}")
    return out


class BNEvalBatch:
    """The eval-mode BatchNorm coefficient requests of one forward pass of a network (kernels.bn_eval_coeffs: real
    BatchNorms in eval mode, and the identity statistics that carry a Linear layer's bias / LayerScale), recorded in call
    order on the first pass and from then on computed by ONE batched launch at the start of the pass and handed out in
    the same order.  Every hand-out is checked against the recorded pointers; any mismatch falls back to the per-layer
    kernel and re-records.  Use through functions.bn_eval_batch()."""

    def __init__(self) -> None:
        self.keys: list[tuple] | None = None
        self.states: list[torch.Tensor] = []
        self._jobs = None
        self._flat = None
        self._rec: list[tuple[tuple, BNParams]] = []
        self.pos = 0
        self.ok = True

    @staticmethod
    def _key(p: "BNParams") -> tuple:
        return (_p(p.weight), _p(p.bias), _p(p.conv_bias), _p(p.ls), _p(p.running_mean), _p(p.running_var), float(p.eps),
                int(p.running_mean.numel()))

    def begin(self) -> None:
        self.pos, self.ok, self._rec = 0, True, []
        if self.keys is not None:
            journal_note(self._flat)
            check(_L().dfd_bn_eval_coeffs_multi(self._jobs, len(self._jobs), _stream()), "dfd_bn_eval_coeffs_multi")

    def request(self, p: "BNParams") -> torch.Tensor:
        key = self._key(p)
        if self.keys is None:
            self._rec.append((key, p))
            return bn_eval_coeffs(p)
        if self.ok and self.pos < len(self.keys) and self.keys[self.pos] == key:
            st = self.states[self.pos]
            self.pos += 1
            return st
        self.ok = False
        return bn_eval_coeffs(p)

    def end(self) -> None:
        if self.keys is None:
            if self._rec:
                self._build()
        elif not self.ok or self.pos != len(self.keys):
            self.keys, self.states, self._jobs, self._flat = None, [], None, None      # the pass changed: record again
        self._rec = []

    def _build(self) -> None:
        from ._lib import BnEvalJob

        recs = self._rec
        dev = recs[0][1].running_mean.device
        with torch.inference_mode(False):
            self._flat = torch.empty(sum(4 * k[-1] for k, _ in recs), dtype=torch.float32, device=dev)
        jobs, at, self.states = (BnEvalJob * len(recs))(), 0, []
        for i, (key, _) in enumerate(recs):
            C = key[-1]
            st = self._flat[at:at + 4 * C].view(4, C)
            at += 4 * C
            self.states.append(st)
            jobs[i] = BnEvalJob(key[0], key[1], key[2], key[3], key[4], key[5], st.data_ptr(), key[6], C)
        self._jobs, self.keys = jobs, [k for k, _ in recs]


class EvalBNStates:
    """The eval-mode coefficient blocks [4][C] of every BatchNorm of a network, refreshed by one batched launch per
    forward pass (dfd_bn_eval_coeffs_multi) instead of one small kernel per layer.  `fresh` is raised by the owning
    network for the duration of its own forward pass: a layer run on its own never sees a stale block."""

    def __init__(self, bns: list) -> None:
        from ._lib import BnEvalJob

        self.ptrs = [(bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr()) for bn in bns]
        dev = bns[0].running_mean.device
        total = sum(4 * bn.running_mean.numel() for bn in bns)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.states, jobs, at = [], (BnEvalJob * len(bns))(), 0
        for i, bn in enumerate(bns):
            C = bn.running_mean.numel()
            st = self.flat[at:at + 4 * C].view(4, C)
            at += 4 * C
            self.states.append(st)
            jobs[i] = BnEvalJob(bn.weight.data_ptr(), bn.bias.data_ptr(), None, None, bn.running_mean.data_ptr(),
                                bn.running_var.data_ptr(), st.data_ptr(), float(bn.eps), C)
        self._jobs = jobs
        self._tensors = [t for bn in bns for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
        self.fresh = False

    def valid_for(self, bns: list) -> bool:
        return len(bns) == len(self.ptrs) and all(
            (bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr()) == p
            for bn, p in zip(bns, self.ptrs))

    def refresh(self) -> None:
        if _journal is not None:
            journal_note(self.flat)
            journal_note(self._tensors)
        check(_L().dfd_bn_eval_coeffs_multi(self._jobs, len(self._jobs), _stream()), "dfd_bn_eval_coeffs_multi")


def pwconv(a: torch.Tensor, pro: Prologue | None, w_nk: torch.Tensor, residual: torch.Tensor | None = None,
           stats: bool = False):
    """out[..., Nout] = P(a)[..., K] @ w_nk[Nout, K]^T (+ residual). Returns (out, partials, nparts).
    w_nk may be an MxWeight (fp8 weights): activations are then quantised to MX fp8 behind the prologue and the product
    runs on the block-scaled fp8 MFMA (no residual / statistics in that form)."""
    if isinstance(w_nk, MxWeight):
        if residual is not None or stats:
            raise ValueError("the MX fp8 GEMM has no residual / statistics epilogue")
        aq, asc = mx_quant_rows(a, pro)
        return mx_gemm(aq, asc, w_nk, a.dtype, (*a.shape[:-1], w_nk.N)), None, 0
    K = a.shape[-1]
    M = a.numel() // K
    Nout = w_nk.shape[0]
    out = torch.empty((*a.shape[:-1], Nout), dtype=a.dtype, device=a.device)
    parts = partials_buf(a.device, Nout) if stats else None
    n = ctypes.c_int(0)
    check(_L().dfd_pwconv_fwd(_dt(a), _p(a), ctypes.byref(pro) if pro is not None else None, _p(w_nk), _p(out),
                              _p(residual), M, K, Nout, _p(parts), MAX_PARTIALS, ctypes.byref(n), _stream()),
          "dfd_pwconv_fwd", f"M={M} K={K} N={Nout}")
    return out, parts, n.value


# ---- eval / inference form of the MBConv block (include/dfd_hip.h "eval / inference form") ---------------------
def pwconv_eval(a: torch.Tensor, w_nk: torch.Tensor, out_state: torch.Tensor, out_act: int) -> torch.Tensor:
    """out = act(scale * (a @ w_nk^T) + shift) with this layer's own BatchNorm coefficients, stored activated."""
    K = a.shape[-1]
    M = a.numel() // K
    Nout = w_nk.shape[0]
    out = torch.empty((*a.shape[:-1], Nout), dtype=a.dtype, device=a.device)
    check(_L().dfd_pwconv_fwd_eval(_dt(a), _p(a), _p(w_nk), _p(out_state), out_act, _p(out), M, K, Nout, _stream()),
          "dfd_pwconv_fwd_eval", f"M={M} K={K} N={Nout}")
    return out


def dwconv_eval(x: torch.Tensor, w: torch.Tensor, out_state: torch.Tensor, out_act: int, k: int, stride: int, pad_top: int,
                pad_left: int, Ho: int, Wo: int):
    """y = act(bn(dwconv(x))) stored activated + per-(tile, image) channel sums of y. Returns (y, pool_parts, ntiles)."""
    _chk_nhwc(x)
    N, H, W, C = x.shape
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    shp = _dw_shape(x.shape, Ho, Wo, k, stride, pad_top, pad_left)
    tiles = _L().dfd_dwconv_fwd_eval_tiles(_dt(x), ctypes.byref(shp))
    if tiles < 1:
        raise ValueError(f"dfd_dwconv_fwd_eval: unsupported shape {tuple(x.shape)} k{k}s{stride}")
    parts = torch.empty((tiles, N, C), dtype=torch.float32, device=x.device)
    n = ctypes.c_int(0)
    check(_L().dfd_dwconv_fwd_eval(_dt(x), _p(x), _p(w), _p(out_state), out_act, _p(y), ctypes.byref(shp), _p(parts),
                                   ctypes.byref(n), _stream()), "dfd_dwconv_fwd_eval", f"{tuple(x.shape)} k{k}s{stride}")
    assert n.value == tiles
    return y, parts, tiles


def se_fwd_parts(parts: torch.Tensor, HW: int, w1, b1, w2, b2, act: int, w2t: torch.Tensor | None = None):
    """The squeeze-excite branch from the producer's per-tile channel sums (no pooling pass): returns pooled, gate, w2t."""
    tiles, N, C = parts.shape
    R = w1.shape[0]
    dev = parts.device
    pooled = torch.empty((N, C), dtype=torch.float32, device=dev)
    hpre = torch.empty((N, R), dtype=torch.float32, device=dev)
    gate = torch.empty((N, C), dtype=torch.float32, device=dev)
    ready = w2t is not None
    if not ready:
        w2t = torch.empty((R, C), dtype=torch.float32, device=dev)
    check(_L().dfd_se_fwd_parts(_p(parts), tiles, N, HW, C, _p(w1), _p(b1), None if ready else _p(w2), _p(b2), R, act,
                                _p(pooled), _p(hpre), _p(gate), _p(w2t), _stream()), "dfd_se_fwd_parts", f"C={C} R={R}")
    return pooled, gate, w2t


def pwconv_wgrad(p: torch.Tensor, pro_p: Prologue | None, q: torch.Tensor, pro_q: Prologue | None,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """dw[Ni, Nj] = sum_m P(p)[m, i] * Q(q)[m, j]."""
    Ni, Nj = p.shape[-1], q.shape[-1]
    M = p.numel() // Ni
    nbytes = _L().dfd_pwconv_wgrad_ws(M, Ni, Nj)
    ws = scratch(p.device, "wgrad_ws", nbytes)
    dw = _dst(None if out is None else out.view(Ni, Nj), (Ni, Nj), p.device)
    check(_L().dfd_pwconv_wgrad(_dt(p), _p(p), ctypes.byref(pro_p) if pro_p is not None else None, Ni, _p(q),
                                ctypes.byref(pro_q) if pro_q is not None else None, Nj, M, _p(dw), 0, _p(ws),
                                ws.numel() * 4, _stream()), "dfd_pwconv_wgrad", f"M={M} Ni={Ni} Nj={Nj}")
    return dw


_FUSE_EXPAND_BWD = os.environ.get("DFD_FUSE_EXPAND_BWD", "1") != "0"      # A/B switch
# the 192- / 240-wide instances need one workgroup per CU (a wave's accumulators are 144-180 registers) and measured a hair SLOWER
# than the two kernels they replace: EfficientNet-B0 13.38 vs 13.33 ms/step, EfficientFormerV2-S1 18.01 vs 17.94 — off, kept for A/B
_FUSE_EXPAND_WIDE = os.environ.get("DFD_FUSE_EXPAND_WIDE", "0") == "1"


def pwconv_bwd_fused_ok(dz: torch.Tensor, x: torch.Tensor) -> bool:
    """Shapes dfd_pwconv_bwd_fused serves (include/dfd_hip.h).  Asked BEFORE the caller takes the weight gradient's arena slot:
    a slot handed to a call that then declines would be lost to the fallback (its gradient would land outside the arena)."""
    Cm, Cin = dz.shape[-1], x.shape[-1]
    M = dz.numel() // Cm
    if not (_FUSE_EXPAND_BWD and dz.dtype == torch.bfloat16 and M >= 2048 * 32 * 3 and Cin <= Cm and Cin % 8 == 0 and Cm % 8 == 0):
        return False
    if Cin <= 32 and (Cm <= 128 or (Cm <= 144 and Cm % 16 == 0)):
        return True
    return _FUSE_EXPAND_WIDE and Cin <= 48 and 144 < Cm <= 240 and Cm % 16 == 0


def pwconv_bwd_fused(dz: torch.Tensor, y: torch.Tensor, coef: torch.Tensor, x: torch.Tensor, w_kn: torch.Tensor,
                     residual: torch.Tensor | None, out_w: torch.Tensor | None = None):
    """The expand layer's data AND weight gradient from one pass over (dz, y) (csrc/dfd_pwtnw.hip, DG): returns (dx, dw [Cm, Cin])
    or None when the shape is not the fused kernel's (the caller runs pwconv + pwconv_wgrad; see pwconv_bwd_fused_ok)."""
    if not pwconv_bwd_fused_ok(dz, x):
        return None
    Cm, Cin = dz.shape[-1], x.shape[-1]
    M = dz.numel() // Cm
    nbytes = _L().dfd_pwconv_wgrad_ws(M, Cm, Cin)
    ws = scratch(dz.device, "wgrad_ws", nbytes)
    dx = torch.empty_like(x)
    dw = _dst(None if out_w is None else out_w.view(Cm, Cin), (Cm, Cin), dz.device)
    rc = _L().dfd_pwconv_bwd_fused(_dt(dz), _p(dz), _p(y), _p(coef), _p(x), _p(w_kn), _p(residual), M, Cm, Cin, _p(dx), _p(dw), 0,
                                   _p(ws), ws.numel() * 4, _stream())
    check(rc, "dfd_pwconv_bwd_fused", f"M={M} Cm={Cm} Cin={Cin}")       # (declining here would have cost the caller its arena slot)
    return dx, dw


# ------------------------------------------------------------------ stem
def _stem_shape(x_shape, Cout: int, Ho: int, Wo: int, k: int, stride: int, pt: int, pl: int) -> StemShape:
    N, H, W, _ = x_shape
    return StemShape(N, H, W, Cout, Ho, Wo, k, stride, pt, pl)


def stem_conv_fwd(x: torch.Tensor, w: torch.Tensor, out_dtype: torch.dtype, stride: int, pad_top: int, pad_left: int,
                  Ho: int, Wo: int, stats: bool = True):
    """x: [N,H,W,3] f32; w: [Cout,3,k,k] f32."""
    _chk_nhwc(x)
    if x.dtype != torch.float32 or x.shape[3] != 3:
        raise ValueError("stem input must be f32 [N,H,W,3]")
    Cout, k = w.shape[0], w.shape[2]
    y = torch.empty((x.shape[0], Ho, Wo, Cout), dtype=out_dtype, device=x.device)
    shp = _stem_shape(x.shape, Cout, Ho, Wo, k, stride, pad_top, pad_left)
    parts = partials_buf(x.device, Cout) if stats else None
    n = ctypes.c_int(0)
    check(_L().dfd_stem_conv_fwd(_dt(y), _p(x), _p(w), _p(y), ctypes.byref(shp), _p(parts), MAX_PARTIALS,
                                 ctypes.byref(n), _stream()), "dfd_stem_conv_fwd")
    return y, parts, n.value


def stem_conv_wgrad(x: torch.Tensor, dz: torch.Tensor, y: torch.Tensor | None, coef: torch.Tensor | None, k: int,
                    stride: int, pad_top: int, pad_left: int, out: torch.Tensor | None = None) -> torch.Tensor:
    Cout, Ho, Wo = dz.shape[3], dz.shape[1], dz.shape[2]
    shp = _stem_shape(x.shape, Cout, Ho, Wo, k, stride, pad_top, pad_left)
    nbytes = _L().dfd_stem_conv_wgrad_ws(ctypes.byref(shp))
    ws = scratch(dz.device, "wgrad_ws", nbytes)
    dw = _dst(out, (Cout, 3, k, k), dz.device)
    check(_L().dfd_stem_conv_wgrad(_dt(dz), _p(x), _p(dz), _p(y), _p(coef), _p(dw), ctypes.byref(shp), 0, _p(ws),
                                   ws.numel() * 4, _stream()), "dfd_stem_conv_wgrad")
    return dw


# ------------------------------------------------------------------ head / loss / optimizer
def dropout(x: torch.Tensor, u: torch.Tensor, p: float) -> torch.Tensor:
    out = torch.empty_like(x)
    check(_L().dfd_dropout(_p(x), _p(u), p, _p(out), x.numel(), _stream()), "dfd_dropout")
    return out


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None) -> torch.Tensor:
    N, K = x.shape
    J = w.shape[0]
    out = torch.empty((N, J), dtype=torch.float32, device=x.device)
    check(_L().dfd_linear_fwd(_p(x), _p(w), _p(b), _p(out), N, K, J, _stream()), "dfd_linear_fwd")
    return out


def linear_bwd(dout: torch.Tensor, x: torch.Tensor, w: torch.Tensor, need_dx: bool, need_dw: bool, has_bias: bool,
               out_dw: torch.Tensor | None = None, out_db: torch.Tensor | None = None):
    N, K = x.shape
    J = w.shape[0]
    dev = x.device
    dx = torch.empty((N, K), dtype=torch.float32, device=dev) if need_dx else None
    dw = _dst(out_dw, (J, K), dev) if need_dw else None
    db = _dst(out_db, (J,), dev) if (need_dw and has_bias) else None
    check(_L().dfd_linear_bwd(_p(dout), _p(x), _p(w), _p(dx), _p(dw), _p(db), N, K, J, 0, _stream()), "dfd_linear_bwd")
    return dx, dw, db


def ce_loss(logits: torch.Tensor, targets: torch.Tensor, label_smoothing: float, grad_scale: float = 1.0,
            want_grad: bool = True):
    N, J = logits.shape
    dev = logits.device
    row_loss = torch.empty(N, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits) if want_grad else None
    check(_L().dfd_ce_loss(_p(logits), _p(targets), N, J, label_smoothing, grad_scale, _p(row_loss), _p(loss),
                           _p(dlogits), _stream()), "dfd_ce_loss")
    return loss, dlogits


def softmax_argmax(logits: torch.Tensor, want_probs: bool = True):
    N, J = logits.shape
    probs = torch.empty_like(logits) if want_probs else None
    preds = torch.empty(N, dtype=torch.int64, device=logits.device)
    check(_L().dfd_softmax_argmax(_p(logits), N, J, _p(probs), _p(preds), _stream()), "dfd_softmax_argmax")
    return probs, preds


def adamw_step(table: torch.Tensor, hp: torch.Tensor) -> None:
    check(_L().dfd_adamw_step(_p(table), table.shape[0], _p(hp), _stream()), "dfd_adamw_step")


# ------------------------------------------------------------------ token-mixer set (ABI 110)
def _rows_c(t: torch.Tensor) -> tuple[int, int]:
    C = t.shape[-1]
    return t.numel() // C, C


def affine2_apply(dz: torch.Tensor, y: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    rows, C = _rows_c(y)
    out = torch.empty_like(y)
    check(_L().dfd_affine2_apply(_dt(y), _p(dz), _p(y), _p(coef), _p(out), rows, C, _stream()), "dfd_affine2_apply")
    return out


def bn_add_act(y: torch.Tensor, state: torch.Tensor | None, other: torch.Tensor | None, act: int) -> torch.Tensor:
    rows, C = _rows_c(y)
    out = torch.empty_like(y)
    check(_L().dfd_bn_add_act(_dt(y), _p(y), _p(state), _p(other), act, _p(out), rows, C, _stream()), "dfd_bn_add_act")
    return out


def bn_add_act_bwd(g: torch.Tensor, y: torch.Tensor, state: torch.Tensor | None, other: torch.Tensor | None, act: int,
                   stats: bool = True):
    rows, C = _rows_c(y)
    d = torch.empty_like(y)
    parts = partials_buf(y.device, C) if (stats and state is not None) else None
    n = ctypes.c_int(0)
    check(_L().dfd_bn_add_act_bwd(_dt(y), _p(g), _p(y), _p(state), _p(other), act, _p(d), rows, C, _p(parts), MAX_PARTIALS,
                                  ctypes.byref(n), _stream()), "dfd_bn_add_act_bwd")
    return d, parts, n.value


def channel_stats(x: torch.Tensor):
    rows, C = _rows_c(x)
    parts = partials_buf(x.device, C)
    n = ctypes.c_int(0)
    check(_L().dfd_channel_stats(_dt(x), _p(x), rows, C, _p(parts), MAX_PARTIALS, ctypes.byref(n), _stream()), "dfd_channel_stats")
    return parts, n.value


def sum_rows(partials: torch.Tensor, P: int, L: int, out: torch.Tensor, accumulate: bool = False, deferred: bool = False) -> torch.Tensor:
    """out[i] = sum_p partials[p][i] in a fixed order; `partials` (flat f32) must hold P + ceil(P/32) rows of L floats
    (the second reduction stage is written behind the slab).  More than 1024 rows are summed in groups of 1024, last
    group first, so that a group's second stage only overwrites rows that have already been consumed.
    deferred: `out` is a gradient-arena slot nobody reads before the end of the enclosing block and `partials` comes from
    scratch(): inside sum_batch() the sum joins the block's batched sums (same order, same bits) instead of two launches here."""
    if partials.numel() < (P + (min(P, 1024) + 31) // 32) * L:
        raise ValueError("sum_rows: partial slab too small for the two-stage reduction")
    if deferred and P <= 1024 and _sum_batch.open:
        check(_L().dfd_sum_rows_deferred(_p(partials), P, L, _p(out), int(accumulate), _stream()), "dfd_sum_rows_deferred")
        return out
    if P <= 1024:
        check(_L().dfd_sum_rows(_p(partials), P, L, _p(out), int(accumulate), _stream()), "dfd_sum_rows")
        return out
    G = (P + 1023) // 1024
    tmp = torch.empty((G + (G + 31) // 32 + 1) * L, dtype=torch.float32, device=partials.device)
    for gi in reversed(range(G)):
        rows = min(1024, P - gi * 1024)
        check(_L().dfd_sum_rows(partials.data_ptr() + gi * 1024 * L * 4, rows, L, tmp.data_ptr() + gi * L * 4, 0, _stream()), "dfd_sum_rows")
    return sum_rows(tmp, G, L, out, accumulate)


def up2_act_fwd(s: torch.Tensor, act: int) -> torch.Tensor:
    _chk_nhwc(s)
    N, h, w, C = s.shape
    out = torch.empty((N, 2 * h, 2 * w, C), dtype=s.dtype, device=s.device)
    check(_L().dfd_up2_act_fwd(_dt(s), _p(s), act, _p(out), N, h, w, C, _stream()), "dfd_up2_act_fwd")
    return out


def up2_act_bwd(g: torch.Tensor, s: torch.Tensor, act: int) -> torch.Tensor:
    N, h, w, C = s.shape
    ds = torch.empty_like(s)
    ws = torch.empty_like(g) if act != ACT_NONE else None
    check(_L().dfd_up2_act_bwd(_dt(s), _p(g), _p(s), act, _p(ds), N, h, w, C, _p(ws), _stream()), "dfd_up2_act_bwd")
    return ds


def subsample_add(a: torch.Tensor, bias: torch.Tensor | None, x: torch.Tensor, stride: int) -> torch.Tensor:
    N, H, W, C = x.shape
    out = torch.empty_like(a)
    check(_L().dfd_subsample_add(_dt(x), _p(a), _p(bias), _p(x), _p(out), N, H, W, stride, C, _stream()), "dfd_subsample_add")
    return out


def subsample_add_bwd(g: torch.Tensor, dx: torch.Tensor, stride: int) -> torch.Tensor:
    """dx[:, ::stride, ::stride] += g, in place."""
    N, H, W, C = dx.shape
    check(_L().dfd_subsample_add_bwd(_dt(dx), _p(g), _p(dx), N, H, W, stride, C, _stream()), "dfd_subsample_add_bwd")
    return dx


def _code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def bgemm(A: torch.Tensor, sa: tuple, B: torch.Tensor, sb: tuple, C: torch.Tensor, sc: tuple, nb: int, nh: int, M: int, N: int,
          Kd: int, alpha: float = 1.0, bias: torch.Tensor | None = None, round_a: bool = False, round_b: bool = False) -> None:
    """C[b,h,m,n] = alpha * sum_k A[b,h,m,k] B[b,h,k,n] (+ bias[h,m,n]); s* = (sb, sh, sr, sc) element strides."""
    from ._lib import Mat

    ma, mb, mc = Mat(*sa), Mat(*sb), Mat(*sc)
    check(_L().dfd_bgemm(_code(A.dtype), A.data_ptr(), ctypes.byref(ma), _code(B.dtype), B.data_ptr(), ctypes.byref(mb),
                         _code(C.dtype), C.data_ptr(), ctypes.byref(mc), _p(bias), alpha, nb, nh, M, N, Kd, int(round_a),
                         int(round_b), _stream()), "dfd_bgemm", f"nb={nb} nh={nh} M={M} N={N} K={Kd}")


def attn_mfma_supported(dtype: torch.dtype, Nq: int, Nk: int, dk: int, dv: int) -> bool:
    """The one-wave-per-(image, head, 64-token block) attention products (csrc/dfd_attn.hip, dfd_attn_scores / dfd_attn_apply): bf16
    activations, at most 256 tokens on either side, head dimensions that are multiples of 8 up to 128.  DFD_ATTN_MFMA=0 keeps
    dfd_bgemm (A/B)."""
    if os.environ.get("DFD_ATTN_MFMA", "1") == "0":
        return False
    return dtype == torch.bfloat16 and 1 <= Nq <= 256 and 1 <= Nk <= 256 and all(d % 8 == 0 and 8 <= d <= 128 for d in (dk, dv))


def attn_scores(x: torch.Tensor, y: torch.Tensor, H: int, alpha: float = 1.0, bias: torch.Tensor | None = None) -> torch.Tensor:
    """x [B, .., H*D] (Tx tokens per image), y [B, .., H*D] (Ty tokens) bf16 -> f32 [B, H, Tx, Ty] = alpha * x_h y_h^T (+ bias[H, Tx*Ty])."""
    B = x.shape[0]
    D = x.shape[-1] // H
    Tx, Ty = x.numel() // (B * H * D), y.numel() // (B * H * D)
    out = torch.empty((B, H, Tx, Ty), dtype=torch.float32, device=x.device)
    check(_L().dfd_attn_scores(_p(x), _p(y), _p(out), _p(bias), float(alpha), B, H, Tx, Ty, D, _stream()), "dfd_attn_scores",
          f"B={B} H={H} Tx={Tx} Ty={Ty} D={D}")
    return out


def attn_apply(f: torch.Tensor, x: torch.Tensor, out_shape, H: int, alpha: float = 1.0, transpose: bool = False) -> torch.Tensor:
    """f f32 [B, H, To, Tc] (transpose: [B, H, Tc, To]), x [B, .., H*D] with Tc tokens per image -> bf16 tensor of `out_shape`
    ([B, .., H*D], To tokens): out_h = alpha * f_h x_h (or f_h^T x_h)."""
    B = x.shape[0]
    D = x.shape[-1] // H
    Tc = x.numel() // (B * H * D)
    To = f.shape[3] if transpose else f.shape[2]
    out = torch.empty(tuple(out_shape), dtype=x.dtype, device=x.device)
    check(_L().dfd_attn_apply(_p(f), int(transpose), _p(x), _p(out), float(alpha), B, H, To, Tc, D, _stream()), "dfd_attn_apply",
          f"B={B} H={H} To={To} Tc={Tc} D={D}")
    return out


def wattn_supported(dtype: torch.dtype, T: int, hd: int) -> bool:
    """The fused MFMA window attention (csrc/dfd_attn.hip) covers bf16, head_dim 32, at most 64 tokens per window."""
    return dtype == torch.bfloat16 and hd == 32 and 1 <= T <= 64


def wattn_fwd(qkv: torch.Tensor, bias: torch.Tensor | None, H: int, scale: float, want_lse: bool = True):
    """qkv [n, T, 1, 3*H*32] bf16 (q | k | v per token), bias f32 [H, T, T] or None ->
    (o [n, T, 1, H*32] bf16, L [n, H, T] f32 = row max + log(row sum) of scale*q k^T + bias)."""
    n, T = qkv.shape[0], qkv.shape[1]
    C = qkv.shape[-1] // 3
    out = torch.empty((n, T, 1, C), dtype=qkv.dtype, device=qkv.device)
    L = torch.empty((n, H, T), dtype=torch.float32, device=qkv.device) if want_lse else None
    check(_L().dfd_wattn_fwd(_p(qkv), _p(bias), _p(out), _p(L), n, T, H, C // H, float(scale), _stream()), "dfd_wattn_fwd",
          f"n={n} T={T} H={H}")
    return out, L


def wattn_bwd(qkv: torch.Tensor, dout: torch.Tensor, L: torch.Tensor, bias: torch.Tensor | None, H: int, scale: float,
              need_bias: bool):
    """-> (dqkv like qkv, dbias f32 [H, T, T] | None): gradients of wattn_fwd's inputs for the gradient `dout` of o."""
    n, T = qkv.shape[0], qkv.shape[1]
    C = qkv.shape[-1] // 3
    dqkv = torch.empty_like(qkv)
    parts = None
    rows = int(_L().dfd_wattn_parts(n))
    if need_bias:
        parts = torch.empty((rows + (min(rows, 1024) + 31) // 32 + 1) * H * T * T, dtype=torch.float32, device=qkv.device)
    check(_L().dfd_wattn_bwd(_p(qkv), _p(dout), _p(L), _p(bias), _p(dqkv), _p(parts), n, T, H, C // H, float(scale), _stream()),
          "dfd_wattn_bwd", f"n={n} T={T} H={H}")
    dbias = None
    if need_bias:
        dbias = torch.empty((H, T, T), dtype=torch.float32, device=qkv.device)
        sum_rows(parts, rows, H * T * T, dbias.view(-1))
    return dqkv, dbias


def attn_softmax_fwd(S: torch.Tensor, th: tuple | None):
    """S [B,H,Nq,Nk] f32 -> (P, T2); th = (w1 [H,H], b1 [H], w2 [H,H], b2 [H]) or None (then T2 is P)."""
    B, H, Nq, Nk = S.shape
    P = torch.empty_like(S)
    if th is None:
        check(_L().dfd_attn_softmax_fwd(_p(S), None, None, None, None, _p(P), None, B, H, Nq, Nk, _stream()), "dfd_attn_softmax_fwd")
        return P, P
    T2 = torch.empty_like(S)
    w1, b1, w2, b2 = th
    check(_L().dfd_attn_softmax_fwd(_p(S), _p(w1), _p(b1), _p(w2), _p(b2), _p(P), _p(T2), B, H, Nq, Nk, _stream()),
          "dfd_attn_softmax_fwd")
    return P, T2


def attn_softmax_bwd(dT2: torch.Tensor, P: torch.Tensor, th: tuple | None):
    """-> (dT1 | None, dS)"""
    B, H, Nq, Nk = P.shape
    dS = torch.empty_like(P)
    if th is None:
        check(_L().dfd_attn_softmax_bwd(_p(dT2), _p(P), None, None, None, _p(dS), B, H, Nq, Nk, _stream()), "dfd_attn_softmax_bwd")
        return None, dS
    dT1 = torch.empty_like(P)
    check(_L().dfd_attn_softmax_bwd(_p(dT2), _p(P), _p(th[0]), _p(th[2]), _p(dT1), _p(dS), B, H, Nq, Nk, _stream()),
          "dfd_attn_softmax_bwd")
    return dT1, dS


def bias_gather(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    H, T = table.shape
    L = idx.numel()
    full = torch.empty((H, L), dtype=torch.float32, device=table.device)
    check(_L().dfd_bias_gather(_p(table), _p(idx), _p(full), H, T, L, _stream()), "dfd_bias_gather")
    return full


def bias_scatter(dfull: torch.Tensor, idx: torch.Tensor, T: int, out: torch.Tensor | None = None) -> torch.Tensor:
    H = dfull.shape[0]
    L = idx.numel()
    dtable = _dst(out, (H, T), dfull.device)
    check(_L().dfd_bias_scatter(_p(dfull), _p(idx), _p(dtable), H, T, L, 0, _stream()), "dfd_bias_scatter")
    return dtable


def im2col(x: torch.Tensor, in_state: torch.Tensor | None, in_act: int, k: int, stride: int, pad: int, Ho: int, Wo: int) -> torch.Tensor:
    _chk_nhwc(x)
    N, H, W, C = x.shape
    col = torch.empty((N, Ho, Wo, k * k * C), dtype=x.dtype, device=x.device)
    shp = _dw_shape(x.shape, Ho, Wo, k, stride, pad, pad)
    check(_L().dfd_im2col(_dt(x), _p(x), _p(in_state), in_act, _p(col), ctypes.byref(shp), _stream()), "dfd_im2col")
    return col


def conv_fwd(x: torch.Tensor, in_state: torch.Tensor | None, in_act: int, w_nk: torch.Tensor, k: int, stride: int, pad: int,
             Ho: int, Wo: int, stats: bool = False):
    """Dense k x k convolution of act(bn(x)) (or x) with w_nk [Cout, k*k*C] as an implicit GEMM: (y, partials, nparts)."""
    _chk_nhwc(x)
    N, H, W, C = x.shape
    Cout = w_nk.shape[0]
    y = torch.empty((N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    parts = partials_buf(x.device, Cout) if stats else None
    n = ctypes.c_int(0)
    shp = _dw_shape(x.shape, Ho, Wo, k, stride, pad, pad)
    check(_L().dfd_conv_fwd(_dt(x), _p(x), ctypes.byref(shp), _p(in_state), in_act, _p(w_nk), Cout, _p(y), _p(parts),
                            MAX_PARTIALS, ctypes.byref(n), _stream()), "dfd_conv_fwd", f"{tuple(x.shape)} k{k}s{stride} -> {Cout}")
    return y, parts, n.value


def conv_wgrad(p: torch.Tensor, pro_p: Prologue | None, x: torch.Tensor, in_state: torch.Tensor | None, in_act: int, k: int,
               stride: int, pad: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """dw[Cout, k*k*C] (GEMM column order) = sum_m P(p)[m, co] * im2col(act(bn(x)))[m, :], the im2col operand gathered."""
    _chk_nhwc(x)
    Cout = p.shape[-1]
    _, Ho, Wo, _ = p.shape
    C = x.shape[-1]
    shp = _dw_shape(x.shape, Ho, Wo, k, stride, pad, pad)
    nbytes = int(_L().dfd_conv_wgrad_ws(ctypes.byref(shp), Cout))
    ws = scratch(p.device, "wgrad_ws", nbytes)
    dw = _dst(out, (Cout, k * k * C), p.device)
    check(_L().dfd_conv_wgrad(_dt(p), _p(p), ctypes.byref(pro_p) if pro_p is not None else None, Cout, _p(x), ctypes.byref(shp),
                              _p(in_state), in_act, _p(dw), 0, _p(ws), ws.numel() * 4, _stream()), "dfd_conv_wgrad",
          f"{tuple(x.shape)} k{k}s{stride} Cout={Cout}")
    return dw


def col2im(dcol: torch.Tensor, in_shape, k: int, stride: int, pad: int) -> torch.Tensor:
    N, H, W, C = in_shape
    Ho, Wo = dcol.shape[1], dcol.shape[2]
    dx = torch.empty((N, H, W, C), dtype=dcol.dtype, device=dcol.device)
    shp = _dw_shape(in_shape, Ho, Wo, k, stride, pad, pad)
    check(_L().dfd_col2im(_dt(dcol), _p(dcol), _p(dx), ctypes.byref(shp), _stream()), "dfd_col2im")
    return dx


def conv_weight_to_gemm(w: torch.Tensor) -> torch.Tensor:
    """torch [O, I, k, k] f32 -> [O, k*k*I] f32 with column (kh*k + kw)*I + i."""
    O, I, k, _ = w.shape
    out = torch.empty((O, k * k * I), dtype=torch.float32, device=w.device)
    check(_L().dfd_conv_weight_perm(_p(w), _p(out), O, I, k, 1, 0, _stream()), "dfd_conv_weight_perm")
    return out


def conv_weight_to_dgrad_gemm(w: torch.Tensor) -> torch.Tensor:
    """torch [O, I, k, k] f32 -> [I, k*k*O] f32, taps flipped: the forward-convolution weight that carries the output
    gradient of a stride-1 convolution back to its input (dfd_conv_fwd on the gradient tensor)."""
    O, I, k, _ = w.shape
    out = torch.empty((I, k * k * O), dtype=torch.float32, device=w.device)
    check(_L().dfd_conv_weight_perm(_p(w), _p(out), O, I, k, 2, 0, _stream()), "dfd_conv_weight_perm")
    return out


def conv_wgrad_from_gemm(dw_gemm: torch.Tensor, shape, out: torch.Tensor | None = None) -> torch.Tensor:
    O, I, k, _ = shape
    dw = _dst(out, (O, I, k, k), dw_gemm.device)
    check(_L().dfd_conv_weight_perm(_p(dw_gemm), _p(dw), O, I, k, 0, 0, _stream()), "dfd_conv_weight_perm")
    return dw


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, want_stats: bool = True):
    rows, C = _rows_c(x)
    y = torch.empty_like(x)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if want_stats else None
    check(_L().dfd_layernorm_fwd(_dt(x), _p(x), _p(gamma), _p(beta), eps, _p(y), _p(stats), rows, C, _stream()), "dfd_layernorm_fwd")
    return y, stats


def layernorm_bwd(g: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, stats: torch.Tensor, residual: torch.Tensor | None = None,
                  out2: torch.Tensor | None = None):
    """-> (dx [+ residual], dgamma, dbeta).  out2: an f32 [2, C] destination for (dgamma, dbeta) that nobody reads before the
    end of the enclosing block (two adjacent gradient-arena slots): the final sum of the partial rows then goes straight
    there and may be left to the block's batched sums (dfd_sum_rows_deferred) — no temporary, no copy kernels."""
    rows, C = _rows_c(x)
    dx = torch.empty_like(x)
    parts = partials_buf(x.device, C)
    n = ctypes.c_int(0)
    check(_L().dfd_layernorm_bwd(_dt(x), _p(g), _p(x), _p(gamma), _p(stats), _p(residual), _p(dx), _p(parts), MAX_PARTIALS,
                                 ctypes.byref(n), rows, C, _stream()), "dfd_layernorm_bwd")
    if out2 is not None:
        if tuple(out2.shape) != (2, C) or out2.dtype != torch.float32 or not out2.is_contiguous():
            raise ValueError("layernorm_bwd: out2 must be a contiguous f32 [2, C] tensor")
        check(_L().dfd_sum_rows_deferred(_p(parts), n.value, 2 * C, _p(out2), 0, _stream()), "dfd_sum_rows_deferred")
        return dx, out2[0], out2[1]
    both = torch.empty((2, C), dtype=torch.float32, device=x.device)
    sum_rows(parts, n.value, 2 * C, both)
    return dx, both[0], both[1]


def copy_rows(src: torch.Tensor, sidx: torch.Tensor | None, dst: torch.Tensor, didx: torch.Tensor | None, n: int) -> torch.Tensor:
    """dst[didx[r]] = src[sidx[r]] for r < n over [rows, C] matrices (int32 index tensors; None = identity)."""
    C = src.shape[-1]
    if dst.shape[-1] != C or src.dtype != dst.dtype:
        raise ValueError("copy_rows: source and destination rows differ")
    check(_L().dfd_copy_rows(_dt(src), _p(src), _p(sidx), _p(dst), _p(didx), n, C, _stream()), "dfd_copy_rows")
    return dst


def add_rowtable(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """x [.., T, C] + table [T, C] (f32), broadcast over the leading dimensions."""
    T, C = table.shape
    out = torch.empty_like(x)
    check(_L().dfd_add_rowtable(_dt(x), _p(x), _p(table), _p(out), x.numel() // C, T, C, _stream()), "dfd_add_rowtable")
    return out


def rowtable_grad(g: torch.Tensor, T: int) -> torch.Tensor:
    C = g.shape[-1]
    dtable = torch.empty((T, C), dtype=torch.float32, device=g.device)
    nbytes = int(_L().dfd_rowtable_grad_ws(T, C))
    ws = scratch(g.device, "rowtable_ws", nbytes)
    check(_L().dfd_rowtable_grad(_dt(g), _p(g), _p(dtable), g.numel() // C, T, C, 0, _p(ws), nbytes, _stream()),
          "dfd_rowtable_grad")
    return dtable


def avgpool_fwd(x: torch.Tensor, k: int, stride: int) -> torch.Tensor:
    _chk_nhwc(x)
    N, H, W, C = x.shape
    out = torch.empty((N, (H - k) // stride + 1, (W - k) // stride + 1, C), dtype=x.dtype, device=x.device)
    check(_L().dfd_avgpool_fwd(_dt(x), _p(x), _p(out), N, H, W, k, stride, C, _stream()), "dfd_avgpool_fwd")
    return out


def avgpool_bwd(g: torch.Tensor, in_shape, k: int, stride: int) -> torch.Tensor:
    N, H, W, C = in_shape
    dx = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
    check(_L().dfd_avgpool_bwd(_dt(g), _p(g), _p(dx), N, H, W, k, stride, C, _stream()), "dfd_avgpool_bwd")
    return dx


def relpos_bias_fwd(table: torch.Tensor, idx: torch.Tensor, n_local: int, n_global: int) -> torch.Tensor:
    """table [T, H] f32 -> [H, S, S] with S = n_local + n_global."""
    H = table.shape[1]
    S = n_local + n_global
    full = torch.empty((H, S, S), dtype=torch.float32, device=table.device)
    check(_L().dfd_relpos_bias_fwd(_p(table), _p(idx), _p(full), H, n_local, n_global, _stream()), "dfd_relpos_bias_fwd")
    return full


def relpos_bias_bwd(dfull: torch.Tensor, table: torch.Tensor, idx: torch.Tensor, n_local: int, n_global: int) -> torch.Tensor:
    T, H = table.shape
    dtable = torch.empty_like(table)
    check(_L().dfd_relpos_bias_bwd(_p(dfull), _p(table), _p(idx), _p(dtable), H, T, n_local, n_global, _stream()), "dfd_relpos_bias_bwd")
    return dtable


def axpby(x: torch.Tensor, y: torch.Tensor | None, a: float = 1.0, b: float = 1.0, a_dev: torch.Tensor | None = None,
          out: torch.Tensor | None = None) -> torch.Tensor:
    """(a * a_dev[0]) * x + b * y on f32 tensors (into `out`, e.g. a gradient-arena slot, when given)."""
    out = torch.empty_like(x) if out is None else out
    check(_L().dfd_axpby(_p(x), _p(y), a, b, _p(a_dev), _p(out), x.numel(), _stream()), "dfd_axpby")
    return out


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    check(_L().dfd_add(_dt(a), _p(a), _p(b), _p(out), a.numel(), _stream()), "dfd_add")
    return out


class DeviceRng:
    """Philox state {seed, offset} in device memory: every random tensor of a forward pass is a pure function of
    (seed, offset, stream id, index), and `tick()` advances the offset with a kernel — so a captured hipGraph
    draws fresh numbers on every replay (a host-side offset would be frozen into the graph)."""

    def __init__(self, device: torch.device, seed: int | None = None) -> None:
        with torch.inference_mode(False):
            self.state = torch.zeros(2, dtype=torch.int64, device=device)
        self.reseed(seed)
        self._counter_key: tuple = ()
        self._counter_table: torch.Tensor | None = None

    @staticmethod
    def mix_seed(seed: int, rank: int) -> int:
        """The Philox key of data-parallel rank `rank`: ranks see different samples and must not share dropout /
        drop-connect / DropPath masks (a shared SEED would otherwise give every rank the same stream)."""
        return (seed ^ (rank * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF

    def reseed(self, seed: int | None = None, rank: int | None = None) -> None:
        """(seed, offset) <- (mix(seed, rank), 0), in place (a captured hipGraph keeps reading the same two words).
        Defaults: torch.initial_seed() and $RANK.  Call after torch.manual_seed() to re-key an existing network."""
        seed = torch.initial_seed() if seed is None else seed
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        host = torch.tensor([self.mix_seed(seed, rank), 0], dtype=torch.int64)
        with torch.inference_mode(False):
            self.state.copy_(host)

    def uniform(self, n: int, stream_id: int) -> torch.Tensor:
        out = torch.empty(n, dtype=torch.float32, device=self.state.device)
        check(_L().dfd_rand(_p(self.state), stream_id, 0.0, _p(out), n, _stream()), "dfd_rand")
        return out

    def drop_path_scale(self, n: int, keep: float, stream_id: int) -> torch.Tensor:
        out = torch.empty(n, dtype=torch.float32, device=self.state.device)
        check(_L().dfd_rand(_p(self.state), stream_id, keep, _p(out), n, _stream()), "dfd_rand")
        return out

    def tick(self, counters: list[torch.Tensor]) -> None:
        """One launch: every int64 counter += 1 (BatchNorm.num_batches_tracked) and the Philox offset += 1."""
        key = tuple(c.data_ptr() for c in counters)
        if key != self._counter_key:
            self._counter_key = key
            with torch.inference_mode(False):
                self._counter_table = torch.tensor(list(key), dtype=torch.int64).to(self.state.device) if key else None
        journal_note(counters)
        check(_L().dfd_step_tick(_p(self._counter_table), len(key), _p(self.state), _stream()), "dfd_step_tick")


# ------------------------------------------------------------------ measurement hook
# bench.py installs a list here; each front-end call then appends
# (name, algorithmic_bytes, flops, start_event, end_event).  Algorithmic bytes = every
# tensor the call must read or write once (activations, weights, coefficients), i.e.
# the DESIGN.md per-kernel figure, not counter traffic.
_profile_sink: list | None = None
_TIMED = ("bn_act_apply", "bn_bwd_reduce", "act_bn_bwd", "pool_act", "pool_bwd_reduce", "scale_rows", "dwconv_fwd",
          "dwconv_bwd_data", "dwconv_bwd_weight", "dwconv_bwd_fused", "pwconv", "pwconv_wgrad", "pwconv_bwd_fused", "stem_conv_fwd", "stem_conv_wgrad",
          "se_fc_fwd", "se_fc_bwd", "linear_fwd", "linear_bwd", "ce_loss", "adamw_step", "prep_weights", "bn_finalize",
          "bn_bwd_finalize", "bn_bwd_finalize_ex", "dropout", "bgemm", "attn_scores", "attn_apply", "attn_softmax_fwd", "attn_softmax_bwd", "im2col", "col2im",
          "wattn_fwd", "wattn_bwd", "mx_quant_rows", "mx_gemm",
          "bn_add_act", "bn_add_act_bwd", "affine2_apply", "up2_act_fwd", "up2_act_bwd", "layernorm_fwd", "layernorm_bwd",
          "channel_stats", "subsample_add")


def set_profile_sink(sink: list | None) -> None:
    global _profile_sink
    _profile_sink = sink


def _tensor_bytes(obj) -> int:
    if isinstance(obj, torch.Tensor):
        if obj.dim() <= 1 and obj.numel() >= MAX_PARTIALS:      # scratch / partial slabs
            return 0
        return obj.numel() * obj.element_size()
    if isinstance(obj, Prologue):
        return getattr(obj, "_nbytes", 0)
    if isinstance(obj, (tuple, list)):
        return sum(_tensor_bytes(o) for o in obj)
    return 0


def _flops(name: str, args) -> float:
    if name == "pwconv":
        a, w = args[0], args[2]
        return 2.0 * (a.numel() // a.shape[-1]) * a.shape[-1] * w.shape[0]
    if name == "pwconv_wgrad":
        p, q = args[0], args[2]
        return 2.0 * (p.numel() // p.shape[-1]) * p.shape[-1] * q.shape[-1]
    if name == "pwconv_bwd_fused":                           # data gradient + weight gradient
        dz, x = args[0], args[3]
        return 4.0 * (dz.numel() // dz.shape[-1]) * dz.shape[-1] * x.shape[-1]
    return 0.0


def _bytes_8d(name: str, args, fallback: int) -> int:
    """Algorithmic bytes of SURVEY.md section 8(d) — every activation tensor the UNFUSED op must move once, without
    the operands the engine's own fusions add (the second tensor of an affine2 prologue, residuals, statistics):
      1x1 conv / dgrad   M*(K + Nout)*es + K*Nout*es          1x1 wgrad   M*(Ni + Nj)*es + Ni*Nj*4
      depthwise fwd      N*C*(Hin*Win + Hout*Wout)*es + k*k*C*4
      depthwise dgrad    N*C*(Hout*Wout + Hin*Win [+ Hin*Win when the activation derivative is fused])*es
      depthwise wgrad    N*C*(Hout*Wout + Hin*Win)*es
    Everything else: the tensors of the call (as `bytes incl. fusion operands`)."""
    try:
        if name == "pwconv":
            a, w = args[0], args[2]
            K_, es = a.shape[-1], a.element_size()
            M = a.numel() // K_
            return (M * K_ + M * w.shape[0]) * es + w.numel() * es
        if name == "pwconv_wgrad":
            p, q = args[0], args[2]
            M = p.numel() // p.shape[-1]
            return M * (p.shape[-1] + q.shape[-1]) * p.element_size() + p.shape[-1] * q.shape[-1] * 4
        if name == "pwconv_bwd_fused":
            # the TWO ops it replaces, each with its own 8(d) bytes (the kernel moves the shared operands once: that is the point)
            dz, x = args[0], args[3]
            Cm, Cin, es = dz.shape[-1], x.shape[-1], dz.element_size()
            M = dz.numel() // Cm
            return 2 * M * (Cm + Cin) * es + Cm * Cin * (es + 4)
        if name == "dwconv_fwd":
            x, k, Ho, Wo = args[0], args[4], args[8], args[9]
            N, H, W, C = x.shape
            return N * C * (H * W + Ho * Wo) * x.element_size() + k * k * C * 4
        if name == "dwconv_bwd_data":
            dz, xin, in_shape = args[0], args[4], args[7]
            N, H, W, C = in_shape
            return N * C * (dz.shape[1] * dz.shape[2] + H * W * (2 if xin is not None else 1)) * dz.element_size()
        if name == "dwconv_bwd_weight":
            dz, xin = args[0], args[3]
            N, H, W, C = xin.shape
            return N * C * (dz.shape[1] * dz.shape[2] + H * W) * dz.element_size()
    except (IndexError, AttributeError, TypeError):
        pass
    return fallback


def _timed(name: str, fn):
    def inner(*args, **kwargs):
        sink = _profile_sink
        if sink is None:
            return fn(*args, **kwargs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kwargs)
        e1.record()
        nbytes = _tensor_bytes(args) + _tensor_bytes(list(kwargs.values())) + _tensor_bytes(out)
        sink.append((name, nbytes, _flops(name, args), e0, e1, _bytes_8d(name, args, nbytes)))
        return out

    inner.__name__ = name
    inner.__doc__ = fn.__doc__
    return inner


for _name in _TIMED:
    globals()[_name] = _timed(_name, globals()[_name])

__all__ = [name for name in dir() if not name.startswith("_")]
