"""Host-side record / replay of a kernel sequence.

A UNet forward is ~1000 C-ABI launches; building each one in Python (module walk, descriptor structs, tensor
allocation) costs ~14 ms per forward on the host.  On one GPU that is hidden behind ~115 ms of device work, but a rank of an
8-GPU frame-sharded run has ~16 ms of device work per forward and becomes host-bound.  The forward is shape-static across
the Euler steps, so the sequence is recorded once - every C-ABI call with its argument tuple, every collective as a
closure over its (kept-alive) tensors - and replayed as a flat loop of ctypes calls.  Unlike a HIP graph this needs no
capture support from the collective library: RCCL / gloo calls are replayed as ordinary torch.distributed calls in
between the kernel launches.

Contract of a recorded region
  * its inputs that change between replays are updated IN PLACE by the caller (input tokens, timestep buffer); nothing in it
    depends on host-side values that change;
  * device data is moved or computed ONLY by C-ABI launches (recorded) or inside ``step(f)`` closures (re-run by every replay).
    A torch op on device tensors outside those two runs while recording only: its result is a step-invariant CONSTANT of the plan
    (index tables, masks, packed weights built on first use).  ``strict()`` / LKGD_REPLAY_STRICT=1 turns the ops that break this
    contract - an in-place write to a device tensor, a new device tensor computed from device inputs - into errors unless they
    stand inside ``step`` or ``invariant()``.

Why ONLY scratch goes to the pool: a constant made by a torch op in the middle of the region must not sit in memory that an
EARLIER launch of the plan used as scratch - the eager pass is past that launch, a replay runs it again and would overwrite the
constant (seen on the first try: the `torch.arange` feeding a later transformer's frame-position table landed in a block an
earlier transformer's table computation had used and freed).  So the region routes exactly the ``torch.empty``-class calls to
the arena and everything else to the ordinary allocator, where the plan keeps it alive.

Working set (round 6; the reference's knobs for the same problem are ``enable_forward_chunking`` /
``decode_chunk_size``, models/unet_spatio_temporal_condition_controlnet.py:329-356, pipeline_stable_video_diffusion_trans.py:
267-275).  Until round 5 the plan kept every tensor the region created alive (~55 GB for the headline clip: each of the ~1000
launches has its own output buffer).  Now the region allocates from a private pool of the caching allocator (``Arena``): scratch
buffers (``torch.empty`` and friends: whatever the launches fill) are freed when the forward drops them and their blocks are
reused by LATER allocations of the same forward, exactly as in an eager run - so the recorded pointers alias the way the eager
forward's did, which stream order makes valid for every replay too - and since the pool is private nothing outside the plan can
be handed one of those blocks while the plan lives.  The plan keeps alive only what no launch regenerates: the constants above,
its result, and the tensors its ``step`` closures hold.  Footprint = the eager peak of one forward.
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import Callable, List, Optional

import torch
from torch.overrides import TorchFunctionMode

from . import _lib as _lib_module
from ._lib import ERRORS, LkgdHipError

_tls = threading.local()

#: torch calls whose result is scratch: uninitialised memory that a later launch (or step) of the region fills
_SCRATCH = {torch.empty, torch.empty_like, torch.empty_strided, torch.Tensor.new_empty, torch.Tensor.new_empty_strided}

_INPLACE_DUNDER = {"__setitem__", "__iadd__", "__isub__", "__imul__", "__itruediv__", "__ifloordiv__", "__iand__", "__ior__"}

#: False | True (raise) | "log" (append (op, where) to VIOLATIONS): LKGD_REPLAY_STRICT = 0 | 1 | log
STRICT = {"0": False, "1": True, "log": "log"}.get(os.environ.get("LKGD_REPLAY_STRICT", "0"), False)
VIOLATIONS: list = []


def _violation(msg: str, name: str) -> None:
    if STRICT == "log":
        import traceback
        where = [f"{fr.filename.rsplit('/', 1)[-1]}:{fr.lineno}" for fr in traceback.extract_stack()[:-3]
                 if "/lkgd_amd/" in fr.filename and not fr.filename.endswith("replay.py")][-3:]
        VIOLATIONS.append((name, " < ".join(reversed(where))))
        return
    raise ReplayContractError(msg)


class ReplayContractError(LkgdHipError):
    pass


@contextlib.contextmanager
def strict(on: bool = True):
    """inside: recorded regions raise ReplayContractError on data movement that a replay would not redo (tests run under it)"""
    global STRICT
    old, STRICT = STRICT, on
    try:
        yield
    finally:
        STRICT = old


@contextlib.contextmanager
def invariant():
    """declares that the torch ops inside compute step-invariant values from step-invariant inputs (weights, tables): their
    results become constants of the plan being recorded"""
    d = getattr(_tls, "allow", 0)
    _tls.allow = d + 1
    try:
        yield
    finally:
        _tls.allow = d


def step(f: Callable[[], None]) -> None:
    """run a host-side step (collective, torch copy) now; while this thread records, also make it part of the plan so every
    replay runs it again at this point of the launch list"""
    from . import ops
    d = getattr(_tls, "allow", 0)
    _tls.allow = d + 1
    try:
        f()
    finally:
        _tls.allow = d
    plan = ops.current_plan()
    if plan is not None:
        plan.python(f)


class Arena:
    """a private pool of the caching allocator that recorded regions allocate from.  Owned by whoever re-records with the same
    shapes (a pipeline keeps one across its calls, so the second clip's forward finds the first one's blocks cached in the pool
    instead of asking the driver for them again).  ``reserved_bytes()``: device memory the pool holds."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise LkgdHipError("replay.Arena needs the GPU")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.pool = torch.cuda.MemPool()
        #: a plan recorded from this arena is alive: its scratch blocks alias each other the way ITS forward's allocation order
        #: made valid - a second region allocating from the same pool meanwhile would be handed blocks the first plan's replays
        #: still write.  One live plan per arena (ArenaSet hands out idle ones).
        self.busy = False

    @contextlib.contextmanager
    def allocating(self):
        """this thread's allocations come from the pool inside (what torch.cuda.use_mem_pool does: its three calls, without the
        generator machinery - a recorded forward enters here once per scratch allocation)"""
        begin = getattr(torch._C, "_cuda_beginAllocateCurrentThreadToPool", None)
        if begin is None:                                  # another torch build: the public context manager
            with torch.cuda.use_mem_pool(self.pool, self.device):
                yield
            return
        begin(self.device.index, self.pool.id)
        try:
            yield
        finally:
            torch._C._cuda_endAllocateToPool(self.device.index, self.pool.id)
            torch._C._cuda_releasePool(self.device.index, self.pool.id)      # the reference `begin` took; the MemPool keeps its own

    def reserved_bytes(self) -> int:
        pid = tuple(self.pool.id)
        return sum(seg["total_size"] for seg in torch.cuda.memory_snapshot()
                   if tuple(seg.get("segment_pool_id", (0, 0))) == pid and seg.get("device", self.device.index) == self.device.index)

    def allocated_bytes(self) -> int:
        pid = tuple(self.pool.id)
        return sum(seg["allocated_size"] for seg in torch.cuda.memory_snapshot()
                   if tuple(seg.get("segment_pool_id", (0, 0))) == pid and seg.get("device", self.device.index) == self.device.index)


class ArenaSet:
    """the arenas of one owner (a pipeline): ``take()`` returns an idle one - the same one call after call for a single-threaded
    caller - or a new one while others carry live plans (host threads driving the same pipeline object)"""

    def __init__(self):
        self._arenas: List[Arena] = []
        self._lock = threading.Lock()
        self._key = None

    def take(self, device=None, key=None) -> Arena:
        """``key``: what the recorded region's allocation pattern depends on (geometry, variant).  A pool caches blocks of the
        sizes ITS forwards asked for; when the key changes the idle pools are dropped first, so that a caller who walks through
        many resolutions does not accumulate one working set per resolution"""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        with self._lock:
            if key is not None and key != self._key:
                self._arenas = [a for a in self._arenas if a.busy]
                self._key = key
            for a in self._arenas:
                if not a.busy and a.device == dev:
                    a.busy = True
                    return a
            a = Arena(dev)
            a.busy = True
            self._arenas.append(a)
            return a

    def reserved_bytes(self) -> int:
        return sum(a.reserved_bytes() for a in self._arenas)

    def clear(self) -> None:
        """give the pools' cached blocks back to the driver (idle arenas only)"""
        with self._lock:
            self._arenas = [a for a in self._arenas if a.busy]


def _tensors(x, out: list) -> list:
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _tensors(v, out)
    elif isinstance(x, dict):
        for v in x.values():
            _tensors(v, out)
    return out


def _storage(t: torch.Tensor) -> int:
    try:
        return t.untyped_storage().data_ptr()
    except Exception:                      # meta / fake tensors
        return 0


class _Region(TorchFunctionMode):
    """what torch calls mean inside a recorded region.  ``keep_all`` (no arena: the CPU tensors of the gloo tests) keeps every
    result alive, as rounds 2-5 did; with an arena only the results no launch regenerates are kept (module docstring)."""

    def __init__(self, keep: list, keep_all: bool, arena: Optional[Arena] = None):
        super().__init__()
        self.keep, self.keep_all, self.arena = keep, keep_all, arena

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", str(func))
        allowed = getattr(_tls, "allow", 0) > 0
        inplace = name in _INPLACE_DUNDER or (name.endswith("_") and not name.startswith("__")) or kwargs.get("out") is not None
        if inplace and STRICT and not allowed:
            # an in-place write (the target is the first argument or out=): checked BEFORE it runs, so a refused op has no effect
            tgt = kwargs.get("out") if kwargs.get("out") is not None else (args[0] if args else None)
            if isinstance(tgt, torch.Tensor) and tgt.is_cuda:
                _violation(f"in-place torch op '{name}' writes a device tensor inside a recorded region outside replay.step: a "
                           "replay would not redo it", name)
        if self.arena is not None and func in _SCRATCH:
            with self.arena.allocating():
                out = func(*args, **kwargs)
        else:
            out = func(*args, **kwargs)
        if self.keep_all:
            self.keep.append(out)
            if not STRICT:
                return out
        if func in _SCRATCH or inplace:
            return out
        outs = _tensors(out, [])
        if not outs:
            return out
        ins = _tensors(args, _tensors(kwargs, []))
        in_store = {_storage(t) for t in ins}
        fresh = [t for t in outs if _storage(t) not in in_store]
        if fresh:
            # a new tensor that no launch will refill: a constant of the plan
            if not self.keep_all:
                self.keep.extend(fresh)
            if STRICT and not allowed and any(t.is_cuda for t in ins) and any(t.is_cuda for t in fresh):
                _violation(f"torch op '{name}' computes a new device tensor from device inputs inside a recorded region: a replay "
                           "would not redo it (use a C-ABI launch or replay.step, or declare the inputs step-invariant with "
                           "replay.invariant())", name)
        return out


class _RecordingLib:
    """stands in for the ctypes library while recording: calls go through and are appended to the plan"""

    def __init__(self, real, plan: "Plan"):
        self._real, self._plan = real, plan

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        plan = self._plan

        def call(*args):
            rc = fn(*args)
            if name in ("lkgd_groupnorm_chunks", "lkgd_gemm_colstats_block"):     # pure host helpers, return a count
                return rc
            flop = None
            if name == "lkgd_gemm_f16":
                d = args[0]._obj
                flop = 2.0 * d.M * d.N * (72 if d.mode == 3 else d.K)
            # the fused kernels of the GEMM family, with the FLOP formulas ops.ff_fused / ops.tattn_block / ops.ln_qkv use for
            # the eager path's events (bench.py's roofline line counts the same launches replayed or not)
            elif name in ("lkgd_ff_fused_c320", "lkgd_ff_fused_c640"):
                c = 320 if name.endswith("c320") else 640
                flop = 2.0 * args[2] * (8 * c * c + 4 * c * c)
            elif name in ("lkgd_ln_qkv_c320", "lkgd_ln_qkv_c640"):
                c = 320 if name.endswith("c320") else 640
                flop = 2.0 * args[2] * 3 * c * c
            elif name in ("lkgd_tattn_block_c320", "lkgd_tattn_block_c640"):
                c = 320 if name.endswith("c320") else 640
                rows = args[13] * args[14] * args[15]
                flop = 2.0 * rows * (3 * c * c + c * c) + 4.0 * rows * 16 * c
            plan.calls.append((fn, args[:-1], name, flop))   # the last argument of every launch is the stream
            return rc
        return call


class Plan:
    def __init__(self):
        self.calls: List[tuple] = []
        self.keep: list = []
        self.result = None
        self.lib = None          # recording stand-in for the ctypes library (set by `record`)
        self.arena: Optional[Arena] = None

    def release(self) -> None:
        """drop the launch list and every tensor it keeps alive NOW (the plan and its recording stand-in reference each other,
        so without this they wait for the cycle collector); an arena keeps the freed blocks cached for its next plan"""
        self.calls.clear()
        self.keep.clear()
        self.result = None
        self.lib = None
        if self.arena is not None:
            self.arena.busy = False
        self.arena = None

    def python(self, f: Callable[[], None]) -> None:
        """a host-side step (collective, torch copy) to redo at this point of every replay"""
        self.calls.append((None, f, "py", None))

    def run(self, gemm_events=None):
        from . import trace
        with trace.range_("replayed_launch_list"):       # roctx range (LKGD_ROCTX=1): the per-block ranges exist in eager runs only
            return self._run(gemm_events)

    def _run(self, gemm_events=None):
        stream = None
        for fn, args, name, flop in self.calls:
            if fn is None:
                args()
                continue
            if stream is None:
                stream = torch.cuda.current_stream().cuda_stream
            if flop is not None and gemm_events is not None:
                s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_ev.record()
                rc = fn(*args, stream)
                e_ev.record()
                gemm_events.append((s_ev, e_ev, flop))
            else:
                rc = fn(*args, stream)
            if rc:
                raise LkgdHipError(f"{name} failed on replay: {ERRORS.get(rc, rc)}")
        return self.result


class record:
    """``with record(arena) as plan: plan.result = f(...)`` - f runs for real and is recorded.  ``arena`` (a replay.Arena): the
    region allocates from its private pool and the plan holds only what replays do not regenerate; None: every tensor the
    region creates is kept alive (CPU tensors, or a caller that wants no pool)."""

    def __init__(self, arena: Optional[Arena] = None, keep_all: Optional[bool] = None):
        self.arena = arena
        self.keep_all = (arena is None) if keep_all is None else keep_all

    def __enter__(self) -> Plan:
        from . import ops
        if ops.PLAN is not None:
            raise LkgdHipError("nested recording")
        self.plan = Plan()
        self.plan.arena = self.arena
        if self.arena is not None:
            self.arena.busy = True
        self.plan.lib = _RecordingLib(_lib_module.lib(), self.plan)
        ops.set_plan(self.plan)
        self.stack = contextlib.ExitStack()
        try:
            self.stack.enter_context(_Region(self.plan.keep, self.keep_all, self.arena))
        except BaseException:
            ops.set_plan(None)
            self.stack.close()
            self.plan.release()
            raise
        return self.plan

    def __exit__(self, *exc):
        from . import ops
        try:
            self.stack.__exit__(*exc)
        finally:
            ops.set_plan(None)
        return False
