#!/bin/bash
# Timing-experiment variants of the 256x320 GEMM as tools/micro/libwide_<knob>.so (results may be wrong, only the time is
# of interest):  NOSTORE = epilogue computed but not stored;  STAGGER=n = workgroups start (c mod 4) * n * 4 us apart.
# Run from the repo root, then on the GPU box:  python tools/micro/wide_knobs.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
make -s
OBJS=""
for s in gemm gemm_stream gemm_rowpanel gemm_pp norm attn_spatial attn_temporal elementwise fsm conv_small image_ops; do OBJS="$OBJS $s.o"; done
for knob in BASE NOSTORE NOSTAGE NOMFMA "$@"; do
  tag=${knob/=/}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DWIDE_X_$knob -c gemm_wide.hip -o /tmp/gemm_wide_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_wide_$tag.o -o ../../tools/micro/libwide_$tag.so
done
ls -la ../../tools/micro/libwide_*.so
