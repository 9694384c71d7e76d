#!/bin/bash
# Build timing-experiment variants of the ping-pong GEMM (one ingredient of its K-tile body dropped each; results are
# wrong, only the time is of interest) as tools/micro/libpp_<knob>.so.  Run from the repo root, then on the GPU box:
#   python tools/micro/pp_knobs.py
set -e
cd "$(dirname "$0")/../../lkgd_amd/csrc"
SRCS="gemm.hip gemm_stream.hip gemm_wide.hip gemm_rowpanel.hip norm.hip attn_spatial.hip attn_temporal.hip elementwise.hip fsm.hip"
OBJS=""
for s in $SRCS; do OBJS="$OBJS ${s%.hip}.o"; done
for knob in BASE NOSTAGE NOREAD NOBAR NOMFMA "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DPP_X_$knob -c gemm_pp.hip -o /tmp/gemm_pp_$knob.o
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_pp_$knob.o -o ../../tools/micro/libpp_$knob.so
done
ls -la ../../tools/micro/*.so
