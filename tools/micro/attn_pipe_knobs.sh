#!/bin/bash
# Builds the software-pipelined attention program with one ingredient removed per build (results are WRONG by construction:
# timing only) and times the 72x128-level shape on each.  Run from the repo root on the GPU box:  bash tools/micro/attn_pipe_knobs.sh
set -e
cd lkgd_amd/csrc
for k in ${KNOBS:-nolds novalu noexp nomfma nobar nolds+novalu}; do
  ATTN_GEN_KNOB=$k python3 ../../tools/gen_attn_asm.py > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -c attn_spatial_pipe.hip -o /tmp/pipe_$k.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v attn_spatial_pipe.o) /tmp/pipe_$k.o -o /tmp/libpipe_$k.so
done
python3 ../../tools/gen_attn_asm.py > /dev/null
cd ../..
echo "== product"; ATTN_PIPE=2 ATTN_ONLY=1 python3 tools/attn_bench.py 2>&1 | grep "S= 9216\|S= 2304"
for k in ${KNOBS:-nolds novalu noexp nomfma nobar nolds+novalu}; do
  echo "== $k"; LKGD_HIP_LIB=/tmp/libpipe_$k.so ATTN_PIPE=2 ATTN_ONLY=1 python3 tools/attn_bench.py 2>&1 | grep "S= 9216\|S= 2304"
done
