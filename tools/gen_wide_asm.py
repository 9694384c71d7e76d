#!/usr/bin/env python3
r"""Generates the inline-asm K-tile body of lkgd_amd/csrc/gemm_wide.hip (macros WIDE_KTILE_ASM_FIRST / _NEXT in
lkgd_amd/csrc/gemm_wide_ktile.inc): 80 x v_mfma_f32_16x16x32_f16 with the fragment reads software-pipelined one weight
fragment ahead and the next K-tile's nine LDS-DMA loads issued one per weight-fragment step of K-step 0.
Run from the repo root:  python tools/gen_wide_asm.py

Operands of the asm statement (see gemm_wide.hip::wide_ktile):
  %0-%3   x0[j]  token fragments of K-step 0          %4-%7  x1[j]  token fragments of K-step 1
  %8,%9   wf[2]  weight fragment double buffer
  %10 xa0, %11 xa1   LDS address of token fragment 0 for K-step 0 / 1;  fragment j at + j*2048
  %12 wa0, %13 wa1   LDS address of weight fragment 0 for K-step 0 / 1; fragment i at + i*2048
  %14-%17 pA[4]  global sources of this thread's four A rows of the NEXT K-tile (64-bit VGPR pairs)
  %18-%22 oB[5]  byte offsets of its five weight rows from %23 = weights + K offset of the next K-tile (SGPR pair)
  %24     m_a    LDS destination (other stage, A part) + this wave's 1 KiB slice
acc[i][j] (weight fragment i < 10, token fragment j < 4) = a[(4i+j)*4 .. +3].
"""
import os

NL = r"\n\t"


def q(text):
    return '"' + text + NL + '"'


def rd(text):
    # WIDE_RD / WIDE_MM / WIDE_BAR / WIDE_LD are identity macros in the product build (timing knobs: tools/micro/wide_knobs.sh)
    return "WIDE_RD(" + q(text) + ")"


def acc(i, j):
    b = (4 * i + j) * 4
    return "a[%d:%d]" % (b, b + 3)


def body(first, part):
    # operand numbers: statement A uses the full list of the module docstring; statement B has its own short list
    #   B: %0 wf[1] (scratch), %1 wf[0] (in: K-step 1's first weight fragment), %2-%5 x1[j], %6 wa1
    X = (lambda j: j) if part == 0 else None
    if part == 1:
        xop, wop, waop = (lambda j: 2 + j), (lambda b: 1 - b), 6
    else:
        xop, wop, waop = None, (lambda b: 8 + b), None
    """part 0: barrier, K-step 0 (fragment reads, the nine DMA loads, 40 MFMAs) + the reads of K-step 1's first fragments;
    part 1: K-step 1 (40 MFMAs).  The C++ between the two statements prepares the next K-tile's sources while K-step 0's
    MFMAs drain."""
    lines = ["WIDE_WAIT_TOP", "WIDE_BAR(" + q("s_barrier") + ")"] if part == 0 else [q("s_waitcnt lgkmcnt(0)")]
    # K-step 0 fragments first: they are in flight while the DMA loads below issue
    if part == 0:
        for j in range(4):
            lines.append(rd("ds_read_b128 %%%d, %%10 offset:%d" % (j, j * 2048)))
        lines.append(rd("ds_read_b128 %8, %12"))
    # the next K-tile's 9 LDS-DMA loads (4 A rows, 5 weight rows), one per weight-fragment step of K-step 0
    stage = []
    for i in range(4):
        m0 = q("s_mov_b32 m0, %24") if i == 0 else q("s_add_u32 m0, m0, 8192")
        stage.append(m0 + " " + q("s_nop 0") + " WIDE_LD(" + q("global_load_lds_dwordx4 %%%d, off" % (14 + i)) + ")")
    for i in range(5):
        m0 = q("s_add_u32 m0, %24, 32768") if i == 0 else q("s_add_u32 m0, m0, 8192")
        stage.append(m0 + " " + q("s_nop 0") + " WIDE_LD(" + q("global_load_lds_dwordx4 %%%d, %%23" % (18 + i)) + ")")
    for ks in (part,):
        xbase = 0 if part == 0 else 2
        wa = 12 if part == 0 else waop
        for i in range(10):
            cur = wop((ks * 10 + i) & 1)
            nxt = wop((ks * 10 + i + 1) & 1)
            if i < 9:
                lines.append(rd("ds_read_b128 %%%d, %%%d offset:%d" % (nxt, wa, (i + 1) * 2048)))
                if ks == 0:
                    lines.append(stage[i])
                lines.append(q("s_waitcnt lgkmcnt(1)"))
            elif ks == 0:
                for j in range(4):
                    lines.append(rd("ds_read_b128 %%%d, %%11 offset:%d" % (4 + j, j * 2048)))
                lines.append(rd("ds_read_b128 %%%d, %%13" % nxt))
                lines.append(q("s_waitcnt lgkmcnt(5)"))
            else:
                lines.append(q("s_waitcnt lgkmcnt(0)"))
            for j in range(4):
                c = "0" if (first and ks == 0) else acc(i, j)
                lines.append("WIDE_MM(" + q("v_mfma_f32_16x16x32_f16 %s, %%%d, %%%d, %s" % (acc(i, j), cur, xbase + j, c)) + ")")
    return " \\\n  ".join(lines)


def duo_body(first):
    """K-tile body of gemm_duo.hip (128 x 320 tiles, two workgroups of 4 waves per CU, BK = 32: ONE K-step per body).
    Operands: %0-%3 x[j] token fragments, %4,%5 wf[2] weight fragment double buffer, %6 xa / %7 wa LDS address of token /
    weight fragment 0 (fragment n at + n*1024: a fragment is 16 rows x 64 bytes, contiguous), %8,%9 pA[2] global sources of
    this lane's two A chunks of the NEXT K-tile, %10-%14 oB[5] byte offsets of its five weight chunks from %15 (SGPR pair),
    %16 m_a LDS destination (other stage) + this wave's first fragment.  Wave w stages A fragments w, w+4 and weight
    fragments w, w+4, .., w+16: the destinations are 4 KiB apart."""
    lines = [q("s_waitcnt lgkmcnt(0)"), q("s_barrier")]
    for j in range(4):
        lines.append(q("ds_read_b128 %%%d, %%6 offset:%d" % (j, j * 1024)))
    lines.append(q("ds_read_b128 %4, %7"))
    stage = []
    for i in range(2):
        m0 = q("s_mov_b32 m0, %16") if i == 0 else q("s_add_u32 m0, m0, 4096")
        stage.append(m0 + " " + q("s_nop 0") + " " + q("global_load_lds_dwordx4 %%%d, off" % (8 + i)))
    for i in range(5):
        m0 = q("s_add_u32 m0, %16, 8192") if i == 0 else q("s_add_u32 m0, m0, 4096")
        stage.append(m0 + " " + q("s_nop 0") + " " + q("global_load_lds_dwordx4 %%%d, %%15" % (10 + i)))
    for i in range(10):
        cur, nxt = 4 + (i & 1), 4 + ((i + 1) & 1)
        if i < 9:
            lines.append(q("ds_read_b128 %%%d, %%7 offset:%d" % (nxt, (i + 1) * 1024)))
            if i < len(stage):
                lines.append(stage[i])
            lines.append(q("s_waitcnt lgkmcnt(1)"))
        else:
            lines.append(q("s_waitcnt lgkmcnt(0)"))
        for j in range(4):
            c = "0" if first else acc(i, j)
            lines.append(q("v_mfma_f32_16x16x32_f16 %s, %%%d, %%%d, %s" % (acc(i, j), cur, j, c)))
    return " \\\n  ".join(lines)


root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lkgd_amd", "csrc")
with open(os.path.join(root, "gemm_duo_ktile.inc"), "w") as f:
    f.write("// GENERATED by tools/gen_wide_asm.py - do not edit.  K-tile body of gemm_duo.hip (see that script for operands).\n")
    f.write("#define DUO_KTILE_ASM_FIRST \\\n  " + duo_body(True) + "\n\n")
    f.write("#define DUO_KTILE_ASM_NEXT \\\n  " + duo_body(False) + "\n\n")
    f.write("#define DUO_AGPR_CLOBBERS " + ", ".join('"a%d"' % i for i in range(160)) + "\n")
print("wrote gemm_duo_ktile.inc")

out = os.path.join(root, "gemm_wide_ktile.inc")
with open(out, "w") as f:
    f.write("// GENERATED by tools/gen_wide_asm.py - do not edit.  K-tile body of gemm_wide.hip (see that script for operands).\n")
    f.write("#define WIDE_KTILE_ASM_FIRST_A \\\n  " + body(True, 0) + "\n\n")
    f.write("#define WIDE_KTILE_ASM_NEXT_A \\\n  " + body(False, 0) + "\n\n")
    f.write("#define WIDE_KTILE_ASM_B \\\n  " + body(False, 1) + "\n\n")
    f.write("#define WIDE_AGPR_CLOBBERS " + ", ".join('"a%d"' % i for i in range(160)) + "\n")
print("wrote", out)
