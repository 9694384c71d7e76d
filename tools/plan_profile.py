#!/usr/bin/env python3
"""Per-CALL device time of a recorded UNet forward (lkgd_amd/replay.py) for the full configs[1] forward and for one rank's
slice of a 2 / 4 / 8-GPU run, side by side: every C-ABI call of the plan is bracketed by HIP events on the launch stream,
median over REPS replays.  The slice's call list is the full forward's with fewer rows, so call i of the slice is set
against call i of the full forward x (slice frame-images / 28): the table says WHICH launches keep a rank from its fair
share.  Usage: python tools/plan_profile.py [frames_per_rank ...]   (default 4 7; negative = one CFG half)"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from lkgd_amd import ops, replay

REPS = int(os.environ.get("REPS", "7"))
from lkgd_amd import _lib
for env, fn in (("LKGD_SPLITK", "lkgd_debug_set_gemm_splitk"), ("LKGD_GN_STATS_KB", "lkgd_debug_set_gn_stats_kb"),
                ("LKGD_GN_APPLY_KB", "lkgd_debug_set_gn_apply_kb"), ("LKGD_GEMM_VARIANT", "lkgd_debug_set_gemm_variant")):
    if env in os.environ:          # A/B knobs of the C library (debug entry points, not part of the reference-facing ABI)
        getattr(_lib.lib(), fn)(int(os.environ[env]))
dev = torch.device("cuda", 0)
unet = B.build_unet(dev, False)


def describe(name, args, images=1):
    if name == "lkgd_gemm_f16":
        d = args[0]._obj
        if images > 1:
            tag = {0: "lin", 1: "c3x3", 2: "tconv", 3: "c3x3c8"}.get(d.mode, str(d.mode))
            extra = ("+geglu" if d.geglu else "") + ("+res" if d.res1 else "") + ("+res2" if d.res2 else "")
            return f"gemm {tag} M/img={d.M // images} N={d.N} K={d.K}{extra}", d.M
        tag = {0: "lin", 1: "c3x3", 2: "tconv", 3: "c3x3c8"}.get(d.mode, str(d.mode))
        extra = ("+geglu" if d.geglu else "") + ("+res" if d.res1 else "") + ("+res2" if d.res2 else "")
        return f"gemm {tag} M={d.M} N={d.N} K={d.K}{extra}", d.M
    ints = [a for a in args if isinstance(a, int) and 0 < a < (1 << 31)]
    return name.replace("lkgd_", "") + " " + ",".join(str(i) for i in ints[:6]), 0


def profile(cfgb, frames, h=72, w=128):
    lat0, img, emb, ids = B.synthetic_inputs(dev, frames, h, w)
    tok = ops.prepare_unet_input(lat0.half(), img, 2, 700.0)
    if cfgb == 1:
        tok, emb, ids = tok[:tok.shape[0] // 2].contiguous(), emb[:1].contiguous(), ids[:1].contiguous()
    t_dev = torch.ones(cfgb, dtype=torch.float32, device=dev)
    unet.forward_tokens(tok, cfgb, frames, h, w, 1.0, emb, ids)
    with replay.record() as plan:
        plan.result, _ = unet.forward_tokens(tok, cfgb, frames, h, w, t_dev, emb.half().contiguous(), ids.float().contiguous())
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    times = []
    for _ in range(REPS):
        evs = []
        for fn, args, name, flop in plan.calls:
            if fn is None:
                args()
                continue
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args, stream)
            e.record()
            assert rc == 0, name
            evs.append((s, e))
        torch.cuda.synchronize()
        times.append([s.elapsed_time(e) for s, e in evs])
    med = torch.tensor(times).median(0).values.tolist()
    # wall of an un-instrumented replay
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        plan.run()
    e.record()
    torch.cuda.synchronize()
    calls = [(name, args, flop) for fn, args, name, flop in plan.calls if fn is not None]
    rows = []
    for (name, args, flop), t in zip(calls, med):
        desc, M = describe(name, args)
        rows.append((name, desc, flop or 0.0, t, describe(name, args, cfgb * frames)[0] if name == "lkgd_gemm_f16" else name))
    return rows, s.elapsed_time(e) / 3


def family(name, desc):
    if name == "lkgd_gemm_f16":
        return "gemm " + desc.split()[1]
    return name.replace("lkgd_", "")


cases = [int(a) for a in sys.argv[1:] if a != "full"] or [-4, -7]
full, wall_full = profile(2, 14)
print(f"# full forward: {len(full)} calls, sum of per-call times {sum(r[3] for r in full):.2f} ms, replay wall {wall_full:.2f} ms")
if "full" in sys.argv[1:]:          # the full forward's own table: time and TFLOP/s per GEMM shape, time per other entry point
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for name, desc, flop, t, key in full:
        a = agg[desc if name == "lkgd_gemm_f16" else name]
        a[0] += 1; a[1] += t; a[2] += flop
    print(f"{'call':64s} {'n':>3s} {'ms':>8s} {'us/call':>8s} {'TFLOP/s':>8s}")
    for k, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:64s} {n:3d} {t:8.3f} {t/n*1e3:8.1f} {fl/t/1e9 if fl else 0:8.0f}")
    cases = [] if sys.argv[1:] == ["full"] else cases
for c in cases:
    cfgb, fr = (1, -c) if c < 0 else (2, c)
    share = cfgb * fr / 28.0
    rows, wall = profile(cfgb, fr)
    print(f"\n## slice: {cfgb} CFG entr{'y' if cfgb == 1 else 'ies'} x {fr} frames = {cfgb*fr} of 28 frame-images; {len(rows)} calls, "
          f"sum {sum(r[3] for r in rows):.2f} ms, replay wall {wall:.2f} ms, fair share {wall_full*share:.2f} ms")
    if len(rows) != len(full):
        print("# call lists differ in length; per-family totals only")
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        f = fam[family(r[0], r[1])]
        f[0] += 1; f[1] += r[3]
    for r in full:
        fam[family(r[0], r[1])][2] += r[3] * share
    print(f"{'family':28s} {'calls':>5s} {'slice ms':>9s} {'share ms':>9s} {'excess':>8s}")
    for k, (n, t, sh) in sorted(fam.items(), key=lambda kv: -(kv[1][1] - kv[1][2])):
        print(f"{k:28s} {n:5d} {t:9.3f} {sh:9.3f} {t-sh:8.3f}")
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0])
    for r in rows:
        a = agg[r[4]]
        a[0] += 1; a[1] += r[3]
    for f in full:
        a = agg[f[4]]
        a[2] += f[3] * share; a[3] += 1
    print(f"\n{'call (rows per frame-image)':60s} {'n':>3s} {'nfull':>5s} {'slice ms':>9s} {'share ms':>9s} {'excess':>8s}  {'us/call':>8s}")
    for k, (n, t, sh, nf) in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:80]:
        print(f"{k:60s} {n:3d} {nf:5d} {t:9.3f} {sh:9.3f} {t-sh:8.3f}  {t/max(n,1)*1e3:8.1f}")
    other = defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r[0] != "lkgd_gemm_f16":
            o = other[r[1]]
            o[0] += 1; o[1] += r[3]
    print(f"\n{'other calls (name + leading integer arguments)':70s} {'n':>3s} {'ms':>8s} {'us/call':>8s}")
    for k, (n, t) in sorted(other.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{k:70s} {n:3d} {t:8.3f} {t/n*1e3:8.1f}")
