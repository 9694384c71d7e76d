#!/usr/bin/env python3
"""Does the placement of the operands in HBM change a GEMM's time?  (geglu L0 varied 9.7 -> 11.6 ms between runs of
tools/gemm_shapes_bench.py that differ only in what was allocated before.)  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lkgd_amd import ops
from lkgd_amd.packing import geglu_half

DEV = "cuda:0"
M, N, K = 258048, 2560, 320


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def geglu(tag):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N // 2, device=DEV, dtype=torch.float16)
    bias = torch.zeros(N, device=DEV)
    ms = timeit(lambda: ops.gemm(a, w, out, M=M, N=N, K=K, bias=bias, geglu=geglu_half(N)))
    print(f"{tag:40s} {ms:7.3f} ms  {2.0*M*N*K/ms/1e9:7.1f} TF/s   a@{a.data_ptr():#x} out@{out.data_ptr():#x} "
          f"reserved {torch.cuda.memory_reserved()/2**20:.0f} MiB", flush=True)


geglu("fresh process")
geglu("again")
res1 = torch.zeros(M, 320, device=DEV, dtype=torch.float16)
res2 = torch.zeros(M, 960, device=DEV, dtype=torch.float16)
geglu("with 165+495 MB live")
del res1, res2
geglu("after freeing them (cached blocks)")
torch.cuda.empty_cache()
geglu("after empty_cache")
junk = [torch.empty(int(s * 2**20), device=DEV, dtype=torch.uint8) for s in (3, 77, 130, 513, 1200, 64, 900)]
del junk[1::2]
geglu("fragmented")
del junk
torch.cuda.empty_cache()
geglu("after empty_cache 2")
# out-of-place offsets inside one big block
big = torch.empty(4 * 2**30, device=DEV, dtype=torch.uint8)
for off in (0, 4096, 2**20, 2**21 + 2**16):
    a = big[off:off + M * K * 2].view(torch.float16).view(M, K); a.normal_(0, 0.1)
    o0 = (off + M * K * 2 + 2**21) // 256 * 256
    out = big[o0:o0 + M * (N // 2) * 2].view(torch.float16).view(M, N // 2)
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    bias = torch.zeros(N, device=DEV)
    ms = timeit(lambda: ops.gemm(a, w, out, M=M, N=N, K=K, bias=bias, geglu=geglu_half(N)))
    print(f"big block offset {off:#x}: {ms:7.3f} ms", flush=True)
