#!/bin/bash
# A/B of the schedule options of tools/gen_attn_asm.py (ATTN_GEN_OPT), all parity-correct builds, same box, interleaved twice.
# Run from the repo root on the GPU box:  OPTS="none dot2 w2 dot2+w2" bash tools/micro/attn_pipe_opts.sh
set -e
OPTS=${OPTS:-none dot2 w2 dot2+w2}
cd lkgd_amd/csrc
for k in $OPTS; do
  ATTN_GEN_OPT=$k python3 ../../tools/gen_attn_asm.py > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -c attn_spatial_pipe.hip -o /tmp/pipeopt_$k.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v attn_spatial_pipe.o) /tmp/pipeopt_$k.o -o /tmp/libpipeopt_$k.so
done
python3 ../../tools/gen_attn_asm.py > /dev/null
cd ../..
echo "== compiler-scheduled kernel"; ATTN_PIPE=1 ATTN_ONLY=1 python3 tools/attn_bench.py 2>&1 | grep "S= 9216\|S= 2304"
for rep in 1 2; do
  for k in $OPTS; do
    echo "== $k"; LKGD_HIP_LIB=/tmp/libpipeopt_$k.so PROBE_S=640 python3 tools/micro/attn_pipe_probe.py 2>&1 | grep "max err"
    LKGD_HIP_LIB=/tmp/libpipeopt_$k.so ATTN_PIPE=2 ATTN_ONLY=1 python3 tools/attn_bench.py 2>&1 | grep "S= 9216\|S= 2304"
  done
done
