#!/bin/bash
# FETCH_SIZE of one 3x3 conv (36x64, 640 -> 640, tools/micro/gemm_one.py) under the tap-major K order (tools/micro/libold_convorder.so)
# and the kx-inner order (in-tree build).  Run on the GPU box from the repo root.
out=$PWD/gpurun_out/conv_fetch; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GEMM_CASE=conv
for v in old new; do
  if [ $v = old ]; then export LKGD_HIP_LIB=$R/tools/micro/libold_convorder.so; else unset LKGD_HIP_LIB; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$v -o a -- python3 $R/tools/micro/gemm_one.py > $out/$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob
for v in ("old", "new"):
    for f in glob.glob(f"gpurun_out/conv_fetch/{v}/*counter_collection.csv"):
        tot = n = 0; ns = 0.0
        for r in csv.DictReader(open(f)):
            if "lkgd_gemm_wide" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                tot += float(r["Counter_Value"]); n += 1; ns += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        print(f"{v}: FETCH_SIZE x2 corrected {tot * 2048 / n / 1e6:8.1f} MB per launch over {n} launches, {ns / n / 1e3:7.1f} us per launch (profiled)")
PY
rm -rf $out/old $out/new
