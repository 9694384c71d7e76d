#!/bin/bash
# Timing-experiment variants of the 256x320 GEMM as tools/micro/libwide_<knob>.so (results may be wrong, only the time is
# of interest):  NOSTORE = epilogue computed but not stored;  STAGGER=n = workgroups start (c mod 4) * n * 4 us apart.
# Run from the repo root, then on the GPU box:  python tools/micro/wide_knobs.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
make -s
OBJS=""
for s in gemm gemm_stream gemm_rowpanel gemm_resw norm attn_spatial attn_temporal attn_tfront attn_cross attn_dense elementwise fsm conv_small image_ops vae_ops; do OBJS="$OBJS $s.o"; done
KERNEL=gemm_wide
for knob in BASE "$@"; do
  tag=${knob//=/}; tag=${tag//+/_}
  defs=""; for k in ${knob//+/ }; do defs="$defs -DWIDE_X_$k"; done       # A+B=2: several knobs in one build
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $defs -c $KERNEL.hip -o /tmp/${KERNEL}_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/${KERNEL}_$tag.o -o ../../tools/micro/libwide_$tag.so
done
ls -la ../../tools/micro/libwide_*.so
