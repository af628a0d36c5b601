"""Detector for `autograd.Function`s that keep their own output alive (ctx -> output -> grad_fn -> ctx).

Such a cycle is invisible to reference counting; the node and the AccumulateGrad nodes behind it then survive the iteration
that made them, and a stale AccumulateGrad node on the legacy stream made a later hipStreamEndCapture crash the host
(VERDICT r3 "faults" 12).  The convention "save None, not out" is pinned here: every output tensor
of every custom Function that runs inside `track()` is remembered by weak reference; after backward and after the caller
has dropped its own references, with the cycle collector DISABLED, each of them must be dead.
"""

from __future__ import annotations

import contextlib
import gc
import inspect
import weakref

import torch


def custom_functions(*modules) -> list[type]:
    found = []
    for mod in modules:
        for _, cls in inspect.getmembers(mod, inspect.isclass):
            if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls.__module__ == mod.__name__:
                found.append(cls)
    return found


@contextlib.contextmanager
def track(classes):
    """yields a list of (class name, weakref to an output tensor) filled while the block runs"""
    refs: list[tuple[str, weakref.ref]] = []

    def wrap(cls, orig):
        def apply(*args, **kwargs):
            out = orig(*args, **kwargs)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor):
                    refs.append((cls.__name__, weakref.ref(t)))
            return out

        return staticmethod(apply)

    for cls in classes:
        cls.apply = wrap(cls, cls.apply)
    try:
        yield refs
    finally:
        for cls in classes:
            del cls.apply               # back to the inherited classmethod


def survivors(refs) -> list[str]:
    return sorted({name for name, r in refs if r() is not None})


@contextlib.contextmanager
def no_cycle_collector():
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
