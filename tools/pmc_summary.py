#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md 'HBM': counters are in KiB; FETCH_SIZE reports HALF the bytes of wide coalesced
reads -> doubled; WRITE_SIZE is exact for 16-byte streaming stores)."""
import csv
import sys
from collections import defaultdict


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


def main(fetch_csv, write_csv):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    print(f"{'kernel':44s} {'launches':>8s} {'fetch MB/launch (x2 corrected)':>30s} {'write MB/launch':>16s} {'GB/s (profiled pass)':>22s}")
    rows = []
    for k in fe:
        n, fsum, ns = fe[k]
        wsum = wr.get(k, [0, 0.0, 0.0])[1]
        fetch_b = fsum * 1024 * 2
        write_b = wsum * 1024
        rows.append((ns, k, n, fetch_b / n / 1e6, write_b / n / 1e6, (fetch_b + write_b) / ns))
    for ns, k, n, f, w, bw in sorted(rows, reverse=True)[:12]:
        print(f"{k[:44]:44s} {n:8d} {f:30.2f} {w:16.2f} {bw:22.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
