"""Host-side record / replay of a kernel sequence.

A UNet forward is ~1000 C-ABI launches; building each one in Python (module walk, descriptor structs, tensor
allocation) costs ~14 ms per forward on the host.  On one GPU that is hidden behind ~115 ms of device work, but a rank of an
8-GPU frame-sharded run has ~16 ms of device work per forward and becomes host-bound.  The forward is shape-static across
the Euler steps, so the sequence is recorded once - every C-ABI call with its argument tuple, every collective as a
closure over its (kept-alive) tensors - and replayed as a flat loop of ctypes calls.  Unlike a HIP graph this needs no
capture support from the collective library: RCCL / gloo calls are replayed as ordinary torch.distributed calls in
between the kernel launches.

Contract of a recorded region: every tensor it creates is kept alive by the plan (pointers stay valid), its inputs
that change between replays are updated IN PLACE by the caller (input tokens, timestep buffer), nothing in it depends on
host-side values that change.
"""
from __future__ import annotations

from typing import Callable, List

import torch
from torch.overrides import TorchFunctionMode

from . import _lib as _lib_module
from ._lib import ERRORS, LkgdHipError


class _KeepAll(TorchFunctionMode):
    """keeps every tensor any torch call returns inside the recorded region alive"""

    def __init__(self, keep: list):
        super().__init__()
        self.keep = keep

    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        self.keep.append(out)
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
            elif name == "lkgd_ff_fused_c320":
                flop = 2.0 * args[2] * (2560 * 320 + 320 * 1280)
            elif name in ("lkgd_ln_qkv_c320", "lkgd_ln_qkv_c640"):
                c = 320 if name.endswith("c320") else 640
                flop = 2.0 * args[2] * 3 * c * c
            elif name == "lkgd_tattn_block_c320":
                rows = args[13] * args[14] * args[15]
                flop = 2.0 * rows * (960 * 320 + 320 * 320) + 4.0 * rows * 16 * 320
            plan.calls.append((fn, args[:-1], name, flop))   # the last argument of every launch is the stream
            return rc
        return call


class Plan:
    def __init__(self):
        self.calls: List[tuple] = []
        self.keep: list = []
        self.result = None
        self.lib = None          # recording stand-in for the ctypes library (set by `record`)

    def release(self) -> None:
        """drop the launch list and every tensor it keeps alive NOW (the plan and its recording stand-in reference each other,
        so without this the tens of GB of activations a full-size forward's plan holds wait for the cycle collector)"""
        self.calls.clear()
        self.keep.clear()
        self.result = None
        self.lib = None

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
    """``with record() as plan: plan.result = f(...)`` - f runs for real and is recorded"""

    def __enter__(self) -> Plan:
        from . import ops
        if ops.PLAN is not None:
            raise LkgdHipError("nested recording")
        self.plan = Plan()
        self.plan.lib = _RecordingLib(_lib_module.lib(), self.plan)
        ops.set_plan(self.plan)
        self.mode = _KeepAll(self.plan.keep)
        self.mode.__enter__()
        return self.plan

    def __exit__(self, *exc):
        from . import ops
        self.mode.__exit__(*exc)
        ops.set_plan(None)
        return False
