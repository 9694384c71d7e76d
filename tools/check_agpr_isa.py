#!/usr/bin/env python3
"""Build-time check of the kernels whose generated main loops keep state in accumulation registers ACROSS asm statements
(ff_fused.hip: Y^T, z^T; attn_tblock.hip; qkv_fused.hip; attn_spatial_pipe.hip: Q, O).  The compiler is not told about
that state, so it must never place a value of its own in an accumulation register: what protects the kernels is
`-mllvm -amdgpu-spill-vgpr-to-agpr=0` plus register pressure kept low by hand.  This script reads the device ISA of one
source (`hipcc -S --cuda-device-only`) and fails unless every instantiation of the kernel

  * allocates exactly the accumulation registers its generated loop names (NumAgprs),
  * has no scratch (no spills),
  * has no compiler-emitted instruction outside the #ASMSTART / #ASMEND blocks that touches an accumulation register.

Called by lkgd_amd/csrc/Makefile after compiling each of those objects (so a compiler upgrade or a build with other flags
fails loudly instead of corrupting results) and by tests/test_host_cpu.py.
usage: check_agpr_isa.py file.s kernel_name NumAgprs [NumAgprs ...]"""
import re
import sys


def check(text: str, kernel: str, want) -> list:
    """returns the list of problems (empty = fine)"""
    problems = []
    if kernel not in text:
        return [f"kernel {kernel} not in the ISA"]
    counts = [int(v) for v in re.findall(r"; NumAgprs: (\d+)", text)]
    scratch = [int(v) for v in re.findall(r"; ScratchSize: (\d+)", text)]
    if not counts or sorted(set(counts)) != sorted(set(want)):
        problems.append(f"NumAgprs {sorted(set(counts))}, expected {sorted(set(want))}")
    if any(scratch):
        problems.append(f"scratch in use: {scratch}")
    inside = False
    for line in text.splitlines():
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and not line.lstrip().startswith((";", ".")) and re.search(r"\ba(\d+|\[\d+:\d+\])", line.split(";")[0]):
            problems.append("compiler instruction on an accumulation register: " + line.strip())
    return problems


if __name__ == "__main__":
    if len(sys.argv) < 4:
        sys.exit(__doc__)
    probs = check(open(sys.argv[1]).read(), sys.argv[2], [int(a) for a in sys.argv[3:]])
    if probs:
        print(f"check_agpr_isa: {sys.argv[1]} ({sys.argv[2]}): " + "; ".join(probs[:10]), file=sys.stderr)
        sys.exit(1)
    print(f"check_agpr_isa: {sys.argv[2]} ok")
