#!/bin/bash
# correctness diagnostics: option builds of the generated loop, many launches with fresh data each (tools/micro/attn_pipe_probe2.py)
set -e
OPTS=${OPTS:-w2 w2+sleep w2+bar2}
cd lkgd_amd/csrc
for k in $OPTS; do
  ATTN_GEN_OPT=$k python3 ../../tools/gen_attn_asm.py > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -c attn_spatial_pipe.hip -o /tmp/pipeopt_$k.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v attn_spatial_pipe.o) /tmp/pipeopt_$k.o -o /tmp/libpipeopt_$k.so
done
python3 ../../tools/gen_attn_asm.py > /dev/null
cd ../..
for k in $OPTS; do
  echo "== $k"; LKGD_HIP_LIB=/tmp/libpipeopt_$k.so PROBE_SS=${PROBE_SS:-640,1024,384,256,128} python3 tools/micro/attn_pipe_probe2.py 2>&1 | grep "S="
done
