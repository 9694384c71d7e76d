#!/bin/bash
# HBM-side traffic of every kernel of the bench workload: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# `bench.py --steps 1 --warmup 0 --inference-steps 2`, summarised by tools/pmc_summary.py (gfx950 corrections of
# MI355X_MICROARCH.md: KiB units, FETCH_SIZE x2).  Run on the GPU box from the repo root; writes gpurun_out/<tag>_pmc_hbm_traffic.txt (tag = $1, default r03)
# and gpurun_out/pmc_traffic.json (copy both to profiles/).
TAG=${1:-r03}
out=$PWD/gpurun_out/pmc_${TAG}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 1 --warmup 0 --inference-steps 2 --no-cpu-baseline --no-vae --no-kernel-events"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -o f -- $B > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -o w -- $B > $out/w.log 2>&1
cd $R
f=$(find $out/f -name "*counter_collection.csv" | head -1); w=$(find $out/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f $w gpurun_out/pmc_traffic.json > gpurun_out/${TAG}_pmc_hbm_traffic.txt
cat gpurun_out/${TAG}_pmc_hbm_traffic.txt
rm -rf $out/f $out/w
