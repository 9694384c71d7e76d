#!/bin/bash
# SQ counters of tattn_front_kernel (fused temporal front, 72x128 level): matrix-pipe busy, VALU issue, waits, LDS activity.
# Two rocprofv3 --pmc passes.  Run on the GPU box from the repo root; prints per-launch averages.
out=$PWD/gpurun_out/tfront_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p1 -o a -- python3 $R/tools/micro/tfront_one.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $out/p2 -o b -- python3 $R/tools/micro/tfront_one.py > $out/p2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
dur = []
for f in sorted(glob.glob("gpurun_out/tfront_pmc/p*/*counter_collection.csv")):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "tattn_front" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k in agg: print(f"{k:32s} {agg[k]/n[k]:16.0f}  (avg over {n[k]} launches)")
print(f"kernel duration under the counters: {sum(dur)/len(dur)/1e3:.1f} us")
PY
rm -rf $out/p1 $out/p2
