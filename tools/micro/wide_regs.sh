#!/bin/bash
# register / spill report of the wide GEMM kernels (no GPU needed)
cd "$(dirname "$0")/../../lkgd_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm "$@" -c gemm_wide.hip -o /tmp/gemm_wide_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep "Function Name\|VGPRs Spill\| VGPRs:\|error\|ScratchSize" | sed 's/.*remark: //; s/\[-R.*//' | paste - - - - | awk '{print $3, $5, $9, $NF}'
