#!/usr/bin/env python3
"""Per-shape throughput of the GEMM / implicit-conv kernel on the distinct shapes of one C2 UNet forward
(SURVEY.md App. E).  Prints TFLOP/s per shape and the FLOP-weighted total.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lkgd_amd import ops

DEV = "cuda:0"
# SHARD="B,F" (env): the shapes of ONE rank of a sharded run, e.g. "1,14" (rank of 2), "1,7" (of 4), "1,4" (of 8)
B, F = (int(v) for v in os.environ.get("SHARD", "2,14").split(","))
N_IMG = B * F
LEVELS = [(72, 128, 320), (36, 64, 640), (18, 32, 1280), (9, 16, 1280)]


def shapes():
    out = []   # (name, count, kind, dict)
    conv = {
        0: [(320, 320, 7), (640, 320, 2), (960, 320, 1)],
        1: [(320, 640, 1), (640, 640, 6), (960, 640, 1), (1280, 640, 1), (1920, 640, 1)],
        2: [(640, 1280, 1), (1280, 1280, 7), (1920, 1280, 1), (2560, 1280, 2)],
        3: [(1280, 1280, 11), (2560, 1280, 3)],
    }
    for lv, lst in conv.items():
        H, W, _ = LEVELS[lv]
        for cin, cout, cnt in lst:
            out.append((f"conv3x3 L{lv} {cin}->{cout}", cnt, "conv", dict(H=H, W=W, cin=cin, cout=cout, stride=1, ups=0)))
    out.append(("conv3x3 up L1->L0 640", 1, "conv", dict(H=36, W=64, cin=640, cout=640, stride=1, ups=1)))
    out.append(("conv3x3 up L2->L1 1280", 1, "conv", dict(H=18, W=32, cin=1280, cout=1280, stride=1, ups=1)))
    out.append(("conv3x3 up L3->L2 1280", 1, "conv", dict(H=9, W=16, cin=1280, cout=1280, stride=1, ups=1)))
    out.append(("conv3x3 s2 L0 320", 1, "conv", dict(H=72, W=128, cin=320, cout=320, stride=2, ups=0)))
    out.append(("conv3x3 s2 L1 640", 1, "conv", dict(H=36, W=64, cin=640, cout=640, stride=2, ups=0)))
    out.append(("conv3x3 s2 L2 1280", 1, "conv", dict(H=18, W=32, cin=1280, cout=1280, stride=2, ups=0)))
    for lv, cnt in ((0, 10), (1, 10), (2, 10), (3, 14)):
        H, W, C = LEVELS[lv]
        out.append((f"tconv L{lv} {C}", cnt, "tconv", dict(H=H, W=W, c=C)))
    for lv, cnt in ((0, 5), (1, 5), (2, 5), (3, 1)):
        H, W, C = LEVELS[lv]
        M = N_IMG * H * W
        out.append((f"lin L{lv} proj/out {C}x{C}", cnt * 4, "lin", dict(M=M, N=C, K=C)))
        out.append((f"lin L{lv} qkv {3*C}x{C}", cnt * 2, "lin", dict(M=M, N=3 * C, K=C, plain=True)))   # no bias, no residual
        out.append((f"lin L{lv} geglu {8*C}x{C}", cnt * 3, "geglu", dict(M=M, N=8 * C, K=C)))
        out.append((f"lin L{lv} ffout {C}x{4*C}", cnt * 3, "lin", dict(M=M, N=C, K=4 * C)))
    return out


def run(kind, d, iters=5, variants=None):
    z = lambda *s: torch.randn(*s, device=DEV, dtype=torch.float16) * 0.1   # noqa: E731
    if kind == "conv":
        H, W, cin, cout = d["H"], d["W"], d["cin"], d["cout"]
        Ho, Wo = ((H << d["ups"]) - 1) // d["stride"] + 1, ((W << d["ups"]) - 1) // d["stride"] + 1
        a, w = z(N_IMG * H * W, cin), z(cout, 9 * cin)
        M = N_IMG * Ho * Wo
        out = torch.empty(M, cout, device=DEV, dtype=torch.float16)
        bias = torch.zeros(cout, device=DEV)
        fn = lambda: ops.gemm(a, w, out, M=M, N=cout, K=9 * cin, bias=bias, mode=ops.A_CONV3X3, Cin=cin,   # noqa
                              conv=(Ho, Wo, H, W, d["stride"], d["ups"]))
        flop = 2.0 * M * cout * 9 * cin
    elif kind == "tconv":
        C, HW = d["c"], d["H"] * d["W"]
        M = B * F * HW
        a, w = z(M, C), z(C, 3 * C)
        out = torch.empty(M, C, device=DEV, dtype=torch.float16)
        bias = torch.zeros(C, device=DEV)
        fn = lambda: ops.gemm(a, w, out, M=M, N=C, K=3 * C, bias=bias, mode=ops.A_TCONV3, Cin=C, tconv=(F, HW))  # noqa
        flop = 2.0 * M * C * 3 * C
    else:
        M, N, K = d["M"], d["N"], d["K"]
        a, w = z(M, K), z(N, K)
        geglu = kind == "geglu"
        out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
        bias = None if d.get("plain") else torch.zeros(N, device=DEV)
        res = None if (geglu or d.get("plain") or d.get("nores") or os.environ.get('NORES')) else torch.zeros(M, N, device=DEV, dtype=torch.float16)
        from lkgd_amd.packing import geglu_half
        gw = [geglu_half(N, K)]       # interleave width; the forced 256x320 variant runs its own 80-wide packing
        fn = lambda: ops.gemm(a, w, out, M=M, N=N, K=K, bias=bias, geglu=gw[0] if geglu else 0, res1=res)   # noqa
        flop = 2.0 * M * N * K
    def once():
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters
    if variants is None:
        return flop, once()
    # A/B inside one process, interleaved, best of 3 rounds: the clocks ramp over the first seconds of a process and
    # comparisons across processes are confounded by that
    from lkgd_amd import _lib
    best = {}
    for _ in range(3):
        for v in variants:
            _lib.lib().lkgd_debug_set_gemm_variant(v)
            if kind == "geglu":
                gw[0] = 80 if v in (4, 6) else (geglu_half(N, K) if v == 0 else 32)
            try:
                t = once()
            except Exception:          # variant not applicable to this shape
                t = float("inf")
            best[v] = min(best.get(v, float("inf")), t)
    _lib.lib().lkgd_debug_set_gemm_variant(0)
    return flop, best


def warm(seconds=3.0):
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()


def main_ldsout():
    """A/B of the 256x320 kernel's output path (direct 8-byte stores vs whole row segments through LDS) on every shape the
    automatic dispatch sends there, interleaved in one process"""
    from lkgd_amd import _lib
    warm()
    L = _lib.lib()
    tot = [0.0, 0.0]
    tot_f = 0.0
    print(f"{'shape':34s} {'cnt':>4s}   direct  via-LDS   (ms per launch)")
    for name, cnt, kind, d in shapes():
        best = [float("inf")] * 2
        for _ in range(3):
            for on in (0, 1):
                L.lkgd_debug_set_wide_lds_out(on)
                flop, ms = run(kind, d)
                best[on] = min(best[on], ms)
        L.lkgd_debug_set_wide_lds_out(-1)
        tot_f += flop * cnt
        for on in (0, 1):
            tot[on] += best[on] * cnt
        print(f"{name:34s} {cnt:4d}  {best[0]:7.3f}  {best[1]:7.3f}   {flop / best[0] / 1e9:6.0f} -> {flop / best[1] / 1e9:6.0f} TF/s", flush=True)
    print(f"{'TOTAL ms (x count)':34s}       {tot[0]:7.2f}  {tot[1]:7.2f}   {tot_f / tot[0] / 1e9:6.0f} -> {tot_f / tot[1] / 1e9:6.0f} TF/s")


def main_ab():
    warm()
    variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,3,4,5,6,7").split(",")]
    names = {0: "auto", 1: "t128", 2: "t256", 3: "strm", 4: "wide", 5: "rowp", 6: "resw", 7: "mid"}
    print(f"{'shape':34s} {'cnt':>4s} " + " ".join(f"{names[v]:>8s}" for v in variants) + "   (ms per launch; * = best)")
    tot = {v: 0.0 for v in variants}
    tot_best = tot_f = 0.0
    for name, cnt, kind, d in shapes():
        flop, best = run(kind, d, variants=variants)
        b = min(best.values())
        tot_best += b * cnt
        tot_f += flop * cnt
        for v in variants:
            tot[v] += best[v] * cnt
        print(f"{name:34s} {cnt:4d} " + " ".join(f"{best[v]:7.3f}{'*' if best[v] == b else ' '}" for v in variants)
              + f"   auto {flop / best[0] / 1e9:6.0f} TF/s, best {flop / b / 1e9:6.0f}", flush=True)
    print(f"{'TOTAL ms (x count)':34s}      " + " ".join(f"{tot[v]:8.2f}" for v in variants)
          + f"   best-of {tot_best:.2f} ms = {tot_f / tot_best / 1e9:.0f} TF/s; auto {tot_f / tot[0] / 1e9:.0f} TF/s")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ab":
        return main_ab()
    if len(sys.argv) > 1 and sys.argv[1] == "ldsout":
        return main_ldsout()
    warm()
    if len(sys.argv) > 1:
        from lkgd_amd import _lib
        _lib.lib().lkgd_debug_set_gemm_variant(int(sys.argv[1]))
        print("forced GEMM variant", sys.argv[1])
    tot_f = tot_ms = 0.0
    rows = []
    for name, cnt, kind, d in shapes():
        flop, ms = run(kind, d)
        rows.append((name, cnt, flop * cnt / 1e12, ms * cnt, flop / ms / 1e9))
        tot_f += flop * cnt
        tot_ms += ms * cnt
    print(f"{'shape':34s} {'cnt':>4s} {'TFLOP':>8s} {'ms':>8s} {'TF/s':>8s}")
    for r in rows:
        print(f"{r[0]:34s} {r[1]:4d} {r[2]:8.3f} {r[3]:8.2f} {r[4]:8.1f}")
    print(f"{'TOTAL':34s} {'':4s} {tot_f/1e12:8.2f} {tot_ms:8.2f} {tot_f/tot_ms/1e9:8.1f}")


if __name__ == "__main__":
    main()
