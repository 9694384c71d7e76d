#!/usr/bin/env python3
"""Every GEMM shape of the C2 forward (tools/gemm_shapes_bench.py, automatic dispatch) under two builds of the library in
interleaved child processes: tools/micro/libhead.so (tools/micro/build_head_lib.sh) vs the in-tree build.  KINDS=conv,tconv,
lin,geglu restricts the shapes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import gemm_shapes_bench as g
g.warm(2.0)
kinds = os.environ.get("KINDS", "conv,tconv,lin,geglu").split(",")
tot = 0.0
for name, cnt, kind, d in g.shapes():
    if kind not in kinds: continue
    best = 1e9
    for _ in range(3):
        flop, ms = g.run(kind, d, iters=8)
        best = min(best, ms)
    tot += best * cnt
    print(f"{name:30s} {cnt:3d} {best:7.3f} ms {flop / best / 1e9:7.0f} TF/s", flush=True)
print(f"TOTAL {tot:.2f} ms")
''' % (REPO, REPO)
libs = [("head", os.path.join(HERE, "libhead.so")), ("tree", os.path.join(REPO, "lkgd_amd", "liblkgd_hip.so"))]
if os.environ.get("LIBS"):        # LIBS=tag=path,tag=path ... (paths relative to the repo root)
    libs = [(t.split("=")[0], os.path.join(REPO, t.split("=")[1])) for t in os.environ["LIBS"].split(",")]
res = {}
for tag, lib in libs:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LKGD_HIP_LIB=lib), capture_output=True, text=True)
    res[tag] = [l for l in r.stdout.splitlines() if " ms" in l]
    if r.returncode:
        print(r.stderr[-2000:])
print(f"{'shape':30s} cnt " + " | ".join(f"{t:>8s}: ms   TF/s  " for t, _ in libs))
for rows in zip(*[res[t] for t, _ in libs]):
    print(rows[0] + "".join("   |" + b[34:] for b in rows[1:]))
