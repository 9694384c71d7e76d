#!/bin/bash
# matrix-pipe / VALU occupancy counters of the one-launch temporal attention at the 72x128 level (two passes).  GPU box, repo root.
out=$PWD/gpurun_out/tblock_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PROBE_SHAPE=2,14,9216
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p1 -o a -- python3 $R/tools/micro/tblock_probe.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/p2 -o b -- python3 $R/tools/micro/tblock_probe.py > $out/p2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/tblock_pmc/p*/*counter_collection.csv")):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "tattn_block" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in agg: print(f"{k:32s} {agg[k]/n[k]:16.0f}  (avg over {n[k]} launches)")
PY
rm -rf $out/p1 $out/p2
