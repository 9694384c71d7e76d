#!/bin/bash
# LDS-side counters of the spatial attention kernel (S = 9216): is the kernel bound by LDS bandwidth / bank conflicts?
out=$PWD/gpurun_out/attn_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/l1 -o a -- python3 $R/tools/micro/attn_one.py > $out/l1.log 2>&1
rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/l2 -o b -- python3 $R/tools/micro/attn_one.py > $out/l2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/attn_pmc/l*/*counter_collection.csv")):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "attn_spatial" in r["Kernel_Name"] or "attn_pipe" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in agg: print(f"{k:32s} {agg[k]/n[k]:16.0f}  (avg over {n[k]} launches)")
PY
tail -3 $out/l1.log $out/l2.log | grep -i "error\|invalid\|not" | head
rm -rf $out/l1 $out/l2
