#!/bin/bash
# Timing-experiment variants of the fused temporal front as tools/micro/libtf_<knob>.so (results are wrong): NOEPI (chunk loops
# only), NOMFMA, NOREAD (weight-fragment LDS reads), NODMA (no weight stream).  Then: python tools/micro/tfront_knobs.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
make -s
OBJS=""
for s in gemm gemm_stream gemm_wide gemm_rowpanel gemm_resw norm attn_spatial attn_temporal attn_cross attn_dense elementwise fsm conv_small image_ops vae_ops; do OBJS="$OBJS $s.o"; done
for knob in BASE "$@"; do
  tag=${knob//+/_}
  defs=""; for k in ${knob//+/ }; do defs="$defs -DTF_X_$k"; done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm $defs -c attn_tfront.hip -o /tmp/attn_tfront_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/attn_tfront_$tag.o -o ../../tools/micro/libtf_$tag.so
done
ls ../../tools/micro/libtf_*.so
