#!/bin/bash
# Builds the library of the last commit (HEAD) as tools/micro/libhead.so: the "before" side of same-box A/B runs
# (LKGD_HIP_LIB=tools/micro/libhead.so) while the work tree holds the "after" side.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
rm -rf /tmp/headcsrc && mkdir -p /tmp/headcsrc
(cd $R && git archive HEAD lkgd_amd/csrc include tools/gen_wide_asm.py tools/gen_resw_asm.py) | tar -x -C /tmp/headcsrc
(cd /tmp/headcsrc/lkgd_amd/csrc && make -s -j8)
cp /tmp/headcsrc/lkgd_amd/liblkgd_hip.so $R/tools/micro/libhead.so
ls -la $R/tools/micro/libhead.so
