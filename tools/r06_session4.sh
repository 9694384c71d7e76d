#!/bin/bash
# round-6 evidence, part 3: same-box pair of the round-5 tree (tools/micro/base_tree, built from a3a0a2e) against this tree;
# matrix-pipe counters of the one-launch temporal attention
mkdir -p gpurun_out/r06
O=gpurun_out/r06
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/clip', 'gemm', d['roofline']['achieved'], d.get('memory'))"; }
for r in 1 2; do
  echo -n "round-5 tree : "; (cd tools/micro/base_tree && python3 bench.py --no-cpu-baseline --no-vae --steps 3 --warmup 1 2>/dev/null | line)
  echo -n "round-6 tree : "; python3 bench.py --no-cpu-baseline --no-vae --steps 3 --warmup 1 2>/dev/null | line
done | tee $O/bench_pair_r05_r06.txt
bash tools/micro/tblock_pmc.sh > $O/pmc_tblock.txt 2>&1; cat $O/pmc_tblock.txt | tail -16
echo done
