#!/usr/bin/env python3
"""Where a wave of the resident-weight GEMM spends its cycles (diagnostic build tools/micro/libresw_STAMPS.so from
tools/micro/resw_knobs.sh STAMPS [+ other knobs]): K-loop incl. the wait for its token fragments / epilogue arithmetic /
row stores, averaged per 32-row block."""
import ctypes as C
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import torch
from lkgd_amd import _lib

libs = sorted(glob.glob(os.path.join(HERE, "libresw_STAMPS*.so")))
_lib.LIB_PATH = sys.argv[1] if len(sys.argv) > 1 else libs[0]
from lkgd_amd import ops   # noqa: E402
L = _lib.lib()
L.lkgd_debug_set_gemm_variant(6)
DEV = "cuda:0"
M0 = 28 * 72 * 128
print(os.path.basename(_lib.LIB_PATH))
for (name, M, N, K, geglu, res) in (("L0 geglu", M0, 2560, 320, True, False), ("L0 qkv", M0, 960, 320, False, False),
                                    ("L0 proj", M0, 320, 320, False, True)):
    a = torch.randn(M, K, device=DEV, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=DEV, dtype=torch.float16) * 0.1
    out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
    r = torch.randn_like(out) if res else None
    b = torch.zeros(N, device=DEV) if (geglu or res) else None
    fn = lambda: ops.gemm(a, w, out, M=M, N=N, K=K, bias=b, geglu=80 if geglu else 0, res1=r)   # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
    rc = L.lkgd_debug_resw_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    st = buf.reshape(256, 8, 4).astype(np.float64)
    n = st[..., 3].sum()
    k, ep, sto = st[..., 0].sum() / n, st[..., 1].sum() / n, st[..., 2].sum() / n
    print(f"  {name:10s} {s.elapsed_time(e):6.3f} ms  per block (s_memtime ticks = 100 MHz x ...): K-loop {k:8.0f}  epilogue {ep:8.0f}  stores {sto:8.0f}"
          f"  total {k + ep + sto:8.0f}   blocks/wave {n / (st[..., 3] > 0).sum():.1f}", flush=True)
