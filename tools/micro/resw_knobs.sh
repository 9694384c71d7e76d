#!/bin/bash
# Timing-experiment variants of the resident-weight GEMM as tools/micro/libresw_<knob>.so (results are wrong, only the
# time is of interest): NOSTORE, NOMFMA, NOREAD (weight-fragment LDS reads), NOLOADX (token-fragment loads), SMALLA (token
# rows from an L2-resident window), COALX (token loads as 64 contiguous bytes per four lanes).
# Run from the repo root, then on the GPU box:  python tools/micro/resw_knobs.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
make -s
OBJS=""
for s in gemm gemm_stream gemm_wide gemm_rowpanel norm attn_spatial attn_temporal attn_tfront attn_cross attn_dense elementwise fsm conv_small image_ops vae_ops; do OBJS="$OBJS $s.o"; done
KERNEL=gemm_resw
for knob in BASE "$@"; do
  tag=${knob//=/}; tag=${tag//+/_}
  defs=""; for k in ${knob//+/ }; do defs="$defs -DRESW_X_$k"; done       # A+B: several knobs in one build
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $defs -c $KERNEL.hip -o /tmp/${KERNEL}_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/${KERNEL}_$tag.o -o ../../tools/micro/libresw_$tag.so
done
ls -la ../../tools/micro/libresw_*.so
