TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_bench_under_rocprof.err
cd $R
ls gpurun_out/prof_${TAG}/* | head
f=$(ls gpurun_out/prof_${TAG}/*kernel_stats.csv gpurun_out/prof_${TAG}/*/*kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/${TAG}_kernel_stats_1gpu.csv
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -delete
head -12 gpurun_out/${TAG}_kernel_stats_1gpu.csv | cut -c1-150
tail -c 600 gpurun_out/${TAG}_bench_under_rocprof.json
