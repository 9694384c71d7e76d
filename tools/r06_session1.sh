#!/bin/bash
# round-6 measurement session 1 (one GPU box): variant bench lines with the memory keys, the fill-doubling sensitivity of the fused
# feed-forward (dmapad), per-call profile of the full forward and of a rank's slices.  Output under gpurun_out/r06/.
set -o pipefail
mkdir -p gpurun_out/r06
O=gpurun_out/r06
echo "== bench stock"; python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_stock.json 2> $O/bench_stock.err && tail -c 600 $O/bench_stock.json
for v in lk joint controlnet; do
  echo "== bench $v"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae --$v > $O/bench_$v.json 2> $O/bench_$v.err && python - <<PY
import json; d=json.load(open("$O/bench_$v.json")); print(d["value"], d["memory"])
PY
done
echo "== bench joint+lk"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae --lk --joint > $O/bench_lkjoint.json 2> $O/bench_lkjoint.err && python -c "import json; d=json.load(open('$O/bench_lkjoint.json')); print(d['value'], d['memory'])"
echo "== ff dmapad"; KNOBS="dmapad valupad" bash tools/micro/ff_knobs.sh > $O/ff_dmapad.txt 2>&1; tail -8 $O/ff_dmapad.txt
echo "== plan profile"; python tools/plan_profile.py full -4 -7 > $O/plan_profile.txt 2>&1; head -3 $O/plan_profile.txt
echo done
