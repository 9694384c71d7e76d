#!/bin/bash
# On the GPU box (gpurun): kernel-trace stats of the bench command + the two PMC passes for HBM-side traffic.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 0 --inference-steps 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python3 $R/bench.py --steps 1 --warmup 0 --inference-steps 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $out/write.log
cd $R
F=$(find $out/fetch -name "*counter_collection.csv" | head -1); W=$(find $out/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $out/pmc_traffic.json > $out/pmc_hbm_traffic.txt
S=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $S $out/kernel_stats.csv
# the raw per-dispatch CSVs are large: keep only the summaries
find $out -name "*.csv" | head -20; rm -rf $out/fetch $out/write $out/trace
ls -la $out; head -12 $out/kernel_stats.csv; cat $out/pmc_hbm_traffic.txt | head -8; tail -1 $out/bench_under_rocprof.json
