#!/bin/bash
# same-box A/B of bench.py: two settings of an environment variable, interleaved REPS times (boxes differ by up to 7 %, so
# only pairs taken on one box compare).  Usage: bash tools/micro/bench_pair.sh VAR valueA valueB [REPS]
VAR=$1; A=$2; B=$3; REPS=${4:-2}
for r in $(seq $REPS); do
  for v in "$A" "$B"; do
    echo -n "$VAR=$v : "
    env $VAR=$v python3 bench.py --no-cpu-baseline --no-vae --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/clip', 'gemm', d['roofline']['achieved'])"
  done
done
