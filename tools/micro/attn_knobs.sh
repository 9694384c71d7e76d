#!/bin/bash
# Timing-experiment variants of the spatial attention kernel as tools/micro/libatt_<knob>.so (results are WRONG, only the time
# is of interest): one ingredient removed per build.  Run from the repo root, then on the GPU box: python tools/micro/attn_lib.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
make -s
OBJS=""
for s in gemm gemm_stream gemm_wide gemm_rowpanel gemm_resw norm attn_temporal attn_tfront attn_cross attn_dense elementwise fsm conv_small image_ops vae_ops; do OBJS="$OBJS $s.o"; done
rm -f ../../tools/micro/libatt_*.so
for knob in "$@"; do
  flags=""; for k in ${knob//+/ }; do case $k in KVB*) flags="$flags -DATT_KVB16=${k#KVB}";; NST*) flags="$flags -DATT_NST=${k#NST}";; *) flags="$flags -DATT_X_$k";; esac; done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $flags -c attn_spatial.hip -o /tmp/attn_$knob.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/attn_$knob.o -o ../../tools/micro/libatt_$knob.so
done
ls ../../tools/micro/libatt_*.so
