#!/bin/bash
# round-6 evidence, part 1: the whole GPU suite, smoke, kernel-trace stats and PMC traffic of the bench command
set -o pipefail
mkdir -p gpurun_out/r06
O=gpurun_out/r06
python -m pytest tests -m gpu -x -q > $O/gputest_full.txt 2>&1; rc=$?; tail -3 $O/gputest_full.txt
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash tools/prof_kernel_stats.sh r06 > $O/prof_kernel_stats.log 2>&1; tail -2 $O/prof_kernel_stats.log | cut -c1-300
bash tools/prof_pmc_traffic.sh r06 > $O/prof_pmc.log 2>&1; head -8 gpurun_out/r06_pmc_hbm_traffic.txt | cut -c1-200
echo done
