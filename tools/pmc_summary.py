#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md 'HBM': counters are in KiB; FETCH_SIZE reports HALF the bytes of wide coalesced
reads -> doubled; WRITE_SIZE is exact for 16-byte streaming stores)."""
import csv
import sys
from collections import defaultdict


def kernels_sha16():
    """the fingerprint bench.py recomputes (bench.kernels_sha16): which kernel sources these passes ran"""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernels_sha16()


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0, 0.0])     # name -> [dispatches, counter sum, ns]
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            a = agg[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg


def main(fetch_csv, write_csv, json_out=None):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    print(f"{'kernel':44s} {'launches':>8s} {'fetch MB/launch (x2 corrected)':>30s} {'write MB/launch':>16s} {'GB/s (profiled pass)':>22s}")
    rows = []
    for k in fe:
        n, fsum, ns = fe[k]
        wsum = wr.get(k, [0, 0.0, 0.0])[1]
        fetch_b = fsum * 1024 * 2
        write_b = wsum * 1024
        rows.append((ns, k, n, fetch_b / n / 1e6, write_b / n / 1e6, (fetch_b + write_b) / ns))
    for ns, k, n, f, w, bw in sorted(rows, reverse=True)[:24]:
        print(f"{k[:44]:44s} {n:8d} {f:30.2f} {w:16.2f} {bw:22.1f}")


    if json_out:
        import json
        fam = {k: v for k, v in fe.items() if "lkgd_gemm" in k or "ff_fused" in k or "tattn_block" in k or "ln_qkv" in k}      # what ops.GEMM_EVENTS times
        launches = sum(v[0] for v in fam.values())
        total = sum(v[1] * 1024 * 2 for v in fam.values()) + sum(wr.get(k, [0, 0.0, 0.0])[1] * 1024 for k in fam)
        doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 0 "
                         "--inference-steps 2`, gfx950 corrections per MI355X_MICROARCH.md (KiB units, FETCH_SIZE x2)",
               "kernels": {k.split("(")[0]: {"launches": v[0], "fetch_mb": round(v[1] * 2048 / v[0] / 1e6, 2),
                                             "write_mb": round(wr.get(k, [0, 0.0, 0.0])[1] * 1024 / v[0] / 1e6, 2)}
                           for k, v in fam.items()},
               "gemm_family_bytes_per_launch": int(total / max(launches, 1)),
               "kernels_sha16": kernels_sha16()}
        with open(json_out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
