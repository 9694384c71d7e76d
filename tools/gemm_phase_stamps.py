#!/usr/bin/env python3
"""Diagnostic: where do the streaming GEMM kernel's cycles go?  Uses the stamped build (make -C lkgd_amd/csrc dbg):
per workgroup, shader cycles (s_memtime) spent in [ring wait + barrier], [stage issue], [ds_read + MFMA], [epilogue]."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lkgd_amd import _lib

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "liblkgd_hip_dbg.so")
from lkgd_amd import ops   # noqa: E402

L = _lib.lib()
L.lkgd_debug_set_gemm_variant(int(os.environ.get("VARIANT", "3")))
DEV = "cuda:0"


def run(name, M, N, K, res=False, geglu=0):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
    bias = torch.zeros(N, device=DEV)
    r = torch.zeros(M, N, device=DEV, dtype=torch.float16) if res else None
    for _ in range(3):
        ops.gemm(a, w, out, M=M, N=N, K=K, bias=bias, res1=r, geglu=geglu)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 8))()
    assert L.lkgd_debug_read_stamps(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).reshape(256, 8)
    wait, stage, comp, epi, tot, steps = (t[:, i].mean().item() for i in range(6))
    nk = K // 64 if os.environ.get("VARIANT", "3") != "5" else 1
    print(f"{name:28s} steps/blk {steps:6.0f} tiles/blk {steps/nk:5.1f} | per K-step: wait {wait/steps:6.0f} stage {stage/steps:5.0f} "
          f"mfma {comp/steps:6.0f} | epilogue/tile {epi/(steps/nk):7.0f} | total {tot:9.0f} cyc "
          f"= wait {100*wait/tot:4.1f}% stage {100*stage/tot:4.1f}% mfma {100*comp/tot:4.1f}% epi {100*epi/tot:4.1f}%")


if __name__ == "__main__":
    run("proj L0 320x320 +res", 258048, 320, 320, res=True)
    run("proj L0 320x320", 258048, 320, 320)
    run("geglu L0 2560x320", 258048, 2560, 320, geglu=32)
    run("ffout L0 320x1280 +res", 258048, 320, 1280, res=True)
    run("geglu L1 5120x640", 64512, 5120, 640, geglu=32)
    run("ffout L2 1280x5120 +res", 16128, 1280, 5120, res=True)
