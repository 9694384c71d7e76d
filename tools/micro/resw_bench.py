#!/usr/bin/env python3
"""The K = 320 projections of the 72x128 level under the 256x320, row-panel and resident-weight programs, interleaved in
one process (tools/gemm_shapes_bench.py machinery).  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemm_shapes_bench as gsb   # noqa: E402

gsb.warm()
M = 28 * 72 * 128
cases = [("L0 proj/out 320x320 (bias+res)", "lin", dict(M=M, N=320, K=320)),
         ("L0 proj_in 320x320 (bias)", "lin", dict(M=M, N=320, K=320, nores=True)),
         ("rank-of-2 proj/out", "lin", dict(M=M // 2, N=320, K=320)),
         ("rank-of-4 proj/out", "lin", dict(M=M // 4, N=320, K=320)),
         ("rank-of-8 proj/out", "lin", dict(M=4 * 72 * 128, N=320, K=320)),
         ("rank-of-4 geglu", "geglu", dict(M=M // 4, N=2560, K=320)),
         ("L0 qkv 960x320 (bare)", "lin", dict(M=M, N=960, K=320, plain=True)),
         ("L0 geglu 2560x320", "geglu", dict(M=M, N=2560, K=320)),
         ("rank-of-2 qkv 960x320", "lin", dict(M=M // 2, N=960, K=320, plain=True)),
         ("rank-of-2 geglu", "geglu", dict(M=M // 2, N=2560, K=320)),
         ("rank-of-8 geglu", "geglu", dict(M=4 * 72 * 128, N=2560, K=320))]
variants = [0, 4, 5, 6]
names = {0: "auto", 4: "wide", 5: "rowp", 6: "resw"}
print(f"{'shape':34s} " + " ".join(f"{names[v]:>8s}" for v in variants) + "   (ms per launch)")
for name, kind, d in cases:
    flop, best = gsb.run(kind, d, iters=10, variants=variants)
    print(f"{name:34s} " + " ".join(f"{best[v]:8.3f}" for v in variants) + "   " +
          " ".join(f"{flop / best[v] / 1e9:6.0f}" for v in variants) + " TF/s", flush=True)
