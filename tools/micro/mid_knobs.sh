#!/bin/bash
# knob builds of the four-stage 128x128 GEMM program (gemm.hip, MID_X_*): one ingredient removed per library
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
OBJS=$(ls *.o | grep -v '^gemm.o$')
for k in BASE NOMFMA NODMA NOLDS NOBAR; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm -DMID_X_$k -c gemm.hip -o /tmp/gemm_$k.o
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gemm_$k.o $OBJS -o ../../tools/micro/liblkgd_mid_$k.so
done
