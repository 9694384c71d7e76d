#!/bin/bash
# Builds the fused LayerNorm + QKV projection with one ingredient removed per build (results WRONG by construction: timing only) and times
# the 72x128-level shape on each.  Run from the repo root on the GPU box:  bash tools/micro/ln_qkv_knobs.sh
set -e
KNOBS=${KNOBS:-novalu nolds nobar nodma novalu+nolds novalu+nolds+nobar nomfma}
cd lkgd_amd/csrc
for k in $KNOBS; do
  DEFS=""; GK=$k      # "gen+knobs", "cxx:A+B", or "gen+knobs/A+B" (generator knobs / C++ knobs)
  case $k in
    cxx:*) DEFS=$(echo ${k#cxx:} | sed 's/+/ -DQK_X_/g; s/^/-DQK_X_/'); GK="";;
    */*) DEFS=$(echo ${k#*/} | sed 's/+/ -DQK_X_/g; s/^/-DQK_X_/'); GK=${k%/*};;
  esac
  k=$(echo $k | tr '/:' '__')
  QKV_GEN_KNOB=$GK python3 ../../tools/gen_qkv_asm.py > /dev/null
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -mllvm -amdgpu-spill-vgpr-to-agpr=0 $DEFS -c qkv_fused.hip -o /tmp/qk_$k.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v qkv_fused.o) /tmp/qk_$k.o -o /tmp/libqk_$k.so
done
python3 ../../tools/gen_qkv_asm.py > /dev/null
cd ../..
echo "== product"; PROBE_T=258048 python3 tools/micro/ln_qkv_probe.py 2>&1 | grep "one launch"
for k in $KNOBS; do
  echo "== $k"; k=$(echo $k | tr '/:' '__'); LKGD_HIP_LIB=/tmp/libqk_$k.so PROBE_T=258048 python3 tools/micro/ln_qkv_probe.py 2>&1 | grep "one launch" | tail -1
done
