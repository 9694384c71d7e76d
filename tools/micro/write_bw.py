#!/usr/bin/env python3
"""Plain HBM write / copy bandwidth of the box (torch fill_ / copy_), for the store-bound side of the short-K GEMM roofline."""
import torch
DEV = "cuda:0"
for mb in (165, 660, 2640):
    x = torch.empty(mb * 1024 * 1024 // 2, device=DEV, dtype=torch.float16)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("fill ", lambda: x.fill_(1.0), x.numel() * 2), ("copy ", lambda: y.copy_(x), x.numel() * 4)):
        best = 1e9
        for _ in range(3):
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        print(f"{name} {mb:5d} MiB: {best:7.3f} ms  {nbytes / best / 1e9:6.2f} TB/s")
