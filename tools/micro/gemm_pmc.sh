#!/bin/bash
# SQ counters of the 256x320 GEMM kernel on three shapes (deep-K 3x3 conv, short-K GEGLU, FF-out): matrix-pipe busy, VALU issue,
# waits, LDS activity / conflicts.  Two rocprofv3 --pmc passes per shape.  Run on the GPU box from the repo root.
out=$PWD/gpurun_out/gemm_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASES=${*:-conv geglu ffout}
for c in $CASES; do
  export GEMM_CASE=$c
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/${c}_1 -o a -- python3 $R/tools/micro/gemm_one.py > $out/${c}_1.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $out/${c}_2 -o b -- python3 $R/tools/micro/gemm_one.py > $out/${c}_2.log 2>&1
done
cd $R
CASES="$CASES" python3 - <<'PY'
import csv, glob, collections
import os
for c in os.environ.get("CASES", "conv geglu ffout").split():
    print(f"# {c}")
    for f in sorted(glob.glob(f"gpurun_out/gemm_pmc/{c}_*/*counter_collection.csv")):
        agg = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "lkgd_gemm_" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in agg: print(f"{k:32s} {agg[k]/n[k]:16.0f}  (avg over {n[k]} launches)")
PY
rm -rf $out/*_1 $out/*_2
