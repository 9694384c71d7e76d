#!/usr/bin/env python3
"""Feed-forward of the 72x128 / 36x64 levels as GEGLU GEMM + FF-out GEMM over the whole token matrix vs over row chunks
(GEGLU of chunk i, FF-out of chunk i, ...): does the [rows, 4C] intermediate of a chunk survive in the 256 MB Infinity Cache
between the two launches?  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lkgd_amd import ops

DEV = "cuda:0"


def warm(seconds=3.0):
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()


def main():
    warm()
    for name, M, C in (("L0", 28 * 72 * 128, 320), ("L1", 28 * 36 * 64, 640), ("L2", 28 * 18 * 32, 1280)):
        z = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16) * 0.1   # noqa: E731
        x, res = z(M, C), z(M, C)
        w1, b1 = z(8 * C, C), torch.zeros(8 * C, device=DEV)
        w2, b2 = z(C, 4 * C), torch.zeros(C, device=DEV)
        hid = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
        out = torch.empty(M, C, device=DEV, dtype=torch.float16)

        def ff(nchunks):
            step = (M // nchunks + 255) // 256 * 256
            for r0 in range(0, M, step):
                r1 = min(M, r0 + step)
                ops.gemm(x[r0:r1], w1, hid[r0:r1], M=r1 - r0, N=8 * C, K=C, bias=b1, geglu=80)
                ops.gemm(hid[r0:r1], w2, out[r0:r1], M=r1 - r0, N=C, K=4 * C, bias=b2, res1=res[r0:r1])

        best = {}
        for _ in range(3):
            for n in (1, 2, 4, 8, 16):
                ff(n); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5):
                    ff(n)
                e.record(); torch.cuda.synchronize()
                best[n] = min(best.get(n, 1e9), s.elapsed_time(e) / 5)
        print(f"{name} FF {M} x {C}: " + "  ".join(f"{n} chunk(s) {t:6.3f} ms" for n, t in best.items()), flush=True)


main()
