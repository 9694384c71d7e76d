#!/bin/bash
# round-6 evidence, part 2: the default bench line (with the CPU-baseline leg), CogVideoX, one rank's slices
set -o pipefail
mkdir -p gpurun_out/r06
O=gpurun_out/r06
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err; tail -c 1500 $O/bench_1gpu.json
python bench.py --cogvideox --inference-steps 50 --steps 1 --warmup 1 > $O/bench_cogvideox.json 2> $O/bench_cogvideox.err; cut -c1-300 $O/bench_cogvideox.json
python tools/host_overhead.py > $O/host_overhead_rank_slices.txt 2>&1; tail -8 $O/host_overhead_rank_slices.txt
echo done
