#!/bin/bash
# Timing experiment: the GEGLU epilogues (256x320 and resident-weight programs) without their GELU -> tools/micro/libgelu_NONE.so
# (results wrong).  Then on the GPU box:
#   LIBS=base=lkgd_amd/liblkgd_hip.so,none=tools/micro/libgelu_NONE.so KINDS=geglu python tools/micro/lib_ab_shapes.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
make -s
OBJS=""
for s in gemm gemm_stream gemm_rowpanel norm attn_spatial attn_temporal attn_tfront attn_cross attn_dense elementwise fsm conv_small image_ops vae_ops; do OBJS="$OBJS $s.o"; done
for knob in NONE; do
  for f in gemm_wide gemm_resw; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DLKGD_X_GELU_$knob -c $f.hip -o /tmp/${f}_$knob.o
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_wide_$knob.o /tmp/gemm_resw_$knob.o -o ../../tools/micro/libgelu_$knob.so
done
ls ../../tools/micro/libgelu_*.so
