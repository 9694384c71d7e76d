"""Tracing hooks: roctx ranges per Euler step / UNet forward / UNet block / transformer and residual
block, and per-step device timers.  Off unless ``LKGD_ROCTX=1`` (ranges) or ``LKGD_STEP_TIMERS=1`` (timers) is set: the
hot loop then pays one attribute test per hook.

    LKGD_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -d out -- python3 bench.py --steps 1 --no-cpu-baseline

groups the ~500 launches of a forward under ``down_blocks.0/resnets.0/spatial``-style ranges.  The reference's analogue
is the wall-clock / peak-memory print of CogVideo-main/tools/parallel_inference/parallel_inference_xdit.py:72-104.
A forward replayed from a recorded launch list (lkgd_amd/replay.py, sharded ranks) carries one range per replay."""
from __future__ import annotations

import contextlib
import ctypes
import functools
import os
from typing import List, Optional

ROCTX = os.environ.get("LKGD_ROCTX", "0") == "1"
STEP_TIMERS = os.environ.get("LKGD_STEP_TIMERS", "0") == "1"

_lib = None


def _roctx():
    global _lib
    if _lib is None:
        err = None
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                     "/opt/rocm/lib/libroctx64.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError as e:
                err = e
        if _lib is None:
            raise RuntimeError(f"LKGD_ROCTX=1 but no roctx library could be loaded: {err}")
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePop.restype = ctypes.c_int
    return _lib


def push(name: str) -> None:
    if ROCTX:
        _roctx().roctxRangePushA(name.encode())


def pop() -> None:
    if ROCTX:
        _roctx().roctxRangePop()


@contextlib.contextmanager
def range_(name: str):
    if not ROCTX:
        yield
        return
    push(name)
    try:
        yield
    finally:
        pop()


def traced(label: str):
    """decorator of a block's ``run``: a roctx range named after ``label`` (and the module's `_trace_name`, the dotted
    path `prepare()` stamps on every block) around the launches it enqueues"""
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(self, *a, **k):
            if not ROCTX:
                return fn(self, *a, **k)
            push(f"{getattr(self, '_trace_name', type(self).__name__)}:{label}")
            try:
                return fn(self, *a, **k)
            finally:
                pop()
        return wrapper
    return deco


class StepTimers:
    """device time of every Euler step of one denoise() call: an event pair per step on the launch stream, read back after the
    loop (no synchronisation inside it); ``ms`` is filled by ``finish()``"""

    def __init__(self):
        self.events: List[tuple] = []
        self.ms: Optional[List[float]] = None

    def start(self):
        import torch
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        return s

    def stop(self, s):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append((s, e))

    def finish(self) -> List[float]:
        import torch
        torch.cuda.synchronize()
        self.ms = [s.elapsed_time(e) for s, e in self.events]
        return self.ms
