#!/usr/bin/env python3
r"""Generates the inline-asm K-tile body of lkgd_amd/csrc/gemm_wide4.hip (lkgd_amd/csrc/gemm_wide4_ktile.inc): the 256x320
tile on FOUR waves of 128 tokens x 160 channels (one wave per SIMD, 512 registers each).

Per K-step (32 k) a wave issues 8 token-fragment reads + 10 weight-fragment reads for 80 x v_mfma_f32_16x16x32_f16
(0.225 ds_read_b128 per MFMA; the eight-wave kernel needs 0.35).  Everything the body keeps across statements lives in
registers the compiler never sees (the kernel is compiled with amdgpu_num_vgpr(VC)):
  accumulators  acc[i][j], weight fragment i < 10, token fragment j < 8:
                i < 8  -> a[(8i + j)*4 .. +3]        (256 AGPRs)
                i >= 8 -> v[192 + ((i-8)*8 + j)*4 .. +3]   (64 VGPRs)
  token fragments  XA[j] = v[128 + 4j ..], XB[j] = v[160 + 4j ..]   (K-step 0 / K-step 1 of a K-tile)
  weight fragments W[b] = v[116 + 4b ..], b < 3  (ring: fragment g = 10*kstep + i lives in W[g % 3], read two steps ahead)
Statement A = barrier + K-step 0 + the eight A-row LDS-DMA loads of the NEXT K-tile + the prefetch of K-step 1's token
fragments; statement B = K-step 1 + the ten weight-row loads.  Operands (inputs only):
  A: %0 xa0  %1 xa1  %2 wa0  %3 wa1   LDS byte addresses of fragment 0 (tokens / weights, K-step 0 / 1), fragment n at + n*2048
     %4-%11 pA[8]  global sources of this thread's eight A rows (64-bit VGPR pairs)
     %12 m_a  LDS destination of the other stage's A part + this wave's 1 KiB slice (row block n at + n*4096)
  B: %0 wa1  %1-%10 oB[10] byte offsets of its ten weight rows from %11 = weights + K offset of the next K-tile (SGPR pair)
     %12 m_a
The counted s_waitcnt lgkmcnt values come from a model of the in-order LDS return queue (class Q below).
Run from the repo root:  python tools/gen_wide4_asm.py
"""
import os

NL = r"\n\t"
VC = 116                      # VGPRs left to the compiler
# The 18 LDS-DMA loads of the next K-tile go out ONE per weight-fragment step: the eight A-row loads in K-step 0 (statement
# A), the ten weight-row loads in K-step 1 (statement B).  Measured on 32768x2560x5120 (tools/micro/lib_ab.py): two per step
# in K-step 0 0.705 ms, three 0.728, six 0.765, all eighteen behind the barrier 0.797, one per step 0.655 - an LDS-DMA
# instruction holds the issue port for tens of cycles, and the loads have latency slack to spare.

def q(text):
    return '"' + text + NL + '"'


def acc(i, j):
    if i < 8:
        b = (8 * i + j) * 4
        return "a[%d:%d]" % (b, b + 3)
    b = 192 + ((i - 8) * 8 + j) * 4
    return "v[%d:%d]" % (b, b + 3)


def xa(j):
    return "v[%d:%d]" % (128 + 4 * j, 131 + 4 * j)


def xb(j):
    return "v[%d:%d]" % (160 + 4 * j, 163 + 4 * j)


def wreg(g):
    b = 116 + 4 * (g % 3)
    return "v[%d:%d]" % (b, b + 3)


# 1: the K-tile boundary (vmcnt wait, workgroup barrier, the first nine fragment reads of the next K-tile) sits in front of the
# LAST weight-fragment step's eight MFMAs instead of between two K-tiles: with one wave per SIMD nothing else covers the
# barrier skew and the first reads' latency.  Statement B then takes %13 wa0 / %14 xa0 of the OTHER stage; a one-off prologue
# statement (WIDE4_KTILE_ASM_PRO: %0 xa0, %1 wa0) does the same boundary before the first K-tile.
TAIL = int(os.environ.get("WIDE4_TAIL", "0"))      # measured neutral (0.705 vs 0.706-0.713 ms on 32768x2560x5120): off


class Q:
    """in-order LDS read queue: lgkmcnt(N) guarantees everything but the N youngest reads has returned"""

    def __init__(self, lines):
        self.lines, self.issued, self.done = lines, [], 0     # done = number of oldest reads known complete

    def read(self, key, text):
        self.lines.append("WIDE4_RD(" + q(text) + ")")
        self.issued.append(key)

    def need(self, *keys):
        last = max(self.issued.index(k) for k in keys)
        if last + 1 > self.done:
            self.lines.append(q("s_waitcnt lgkmcnt(%d)" % (len(self.issued) - 1 - last)))
            self.done = last + 1


def body(first, part):
    lines = []
    qq = Q(lines)
    if part == 0:
        if TAIL:
            lines += ["WIDE4_WAIT_TOP"]             # also retires the reads the previous statement's tail issued
            qq.issued = [("w", 0)] + [("xa", j) for j in range(8)]
            qq.done = len(qq.issued)
        else:
            lines += ["WIDE4_WAIT_TOP", "WIDE4_BAR(" + q("s_barrier") + ")"]
            # first operands first: MFMA (0,0) can start after two reads
            qq.read(("w", 0), "ds_read_b128 %s, %%2" % wreg(0))
            for j in range(8):
                qq.read(("xa", j), "ds_read_b128 %s, %%0 offset:%d" % (xa(j), j * 2048))
        qq.read(("w", 1), "ds_read_b128 %s, %%2 offset:2048" % wreg(1))
        stage = []
        for n in range(8):
            m0 = q("s_mov_b32 m0, %12") if n == 0 else q("s_add_u32 m0, m0, 4096")
            stage.append(m0 + " " + q("s_nop 0") + " WIDE4_LD(" + q("global_load_lds_dwordx4 %%%d, off" % (4 + n)) + ")")
        for i in range(10):
            g = i
            if g + 2 < 10:
                qq.read(("w", g + 2), "ds_read_b128 %s, %%2 offset:%d" % (wreg(g + 2), (g + 2) * 2048))
            else:                                   # K-step 1's first two weight fragments
                qq.read(("w", g + 2), "ds_read_b128 %s, %%3 offset:%d" % (wreg(g + 2), (g + 2 - 10) * 2048))
            if i < 8:
                qq.read(("xb", i), "ds_read_b128 %s, %%1 offset:%d" % (xb(i), i * 2048))
            if i < 8:
                lines.append(stage[i])
            for j in range(8):
                qq.need(("w", g), ("xa", j))
                c = "0" if first else acc(i, j)
                lines.append("WIDE4_MM(" + q("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (acc(i, j), wreg(g), xa(j), c)) + ")")
    else:
        lines.append(q("s_waitcnt lgkmcnt(0)"))     # K-step 1's token fragments and first two weight fragments are in
        qq.issued = [("w", 10), ("w", 11)] + [("xb", j) for j in range(8)]
        qq.done = len(qq.issued)
        for i in range(10):
            g = 10 + i
            if i + 2 < 10:
                qq.read(("w", g + 2), "ds_read_b128 %s, %%0 offset:%d" % (wreg(g + 2), (i + 2) * 2048))
            loads = ([0, 1] if i == 0 else [i + 1] if i < 9 else []) if TAIL else [i]
            for n in loads:
                m0 = q("s_add_u32 m0, %12, 32768") if n == 0 else q("s_add_u32 m0, m0, 4096")
                lines.append(m0 + " " + q("s_nop 0") + " WIDE4_LD(" + q("global_load_lds_dwordx4 %%%d, %%11" % (1 + n)) + ")")
            if TAIL and i == 9:
                qq.need(("w", g))
                lines.append(q("s_waitcnt vmcnt(0)"))
                lines.append("WIDE4_BAR(" + q("s_barrier") + ")")
                qq.read(("w0n",), "ds_read_b128 %s, %%13" % wreg(0))
                for j in range(8):
                    qq.read(("xan", j), "ds_read_b128 %s, %%14 offset:%d" % (xa(j), j * 2048))
            for j in range(8):
                qq.need(("w", g), ("xb", j))
                lines.append("WIDE4_MM(" + q("v_mfma_f32_16x16x32_f16 %s, %s, %s, %s" % (acc(i, j), wreg(g), xb(j), acc(i, j))) + ")")
    return " \\\n  ".join(lines)


out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lkgd_amd", "csrc", "gemm_wide4_ktile.inc")
with open(out, "w") as f:
    f.write("// GENERATED by tools/gen_wide4_asm.py - do not edit.  K-tile body of gemm_wide4.hip (see that script for the register map).\n")
    f.write("#define WIDE4_KTILE_ASM_FIRST_A \\\n  " + body(True, 0) + "\n\n")
    f.write("#define WIDE4_KTILE_ASM_NEXT_A \\\n  " + body(False, 0) + "\n\n")
    f.write("#define WIDE4_KTILE_ASM_B \\\n  " + body(False, 1) + "\n\n")
    pro = [q("s_waitcnt vmcnt(0)"), q("s_barrier"), q("ds_read_b128 %s, %%1" % wreg(0))] + \
          [q("ds_read_b128 %s, %%0 offset:%d" % (xa(j), j * 2048)) for j in range(8)]
    f.write("#define WIDE4_KTILE_ASM_PRO \\\n  " + " \\\n  ".join(pro) + "\n\n")
    f.write("#define WIDE4_TAIL %d\n" % TAIL)
    f.write("#define WIDE4_VC %d\n" % VC)
    f.write("#define WIDE4_CLOBBERS " + ", ".join('"a%d"' % i for i in range(256)) + ", " +
            ", ".join('"v%d"' % i for i in range(VC, 256)) + "\n")
print("wrote", out)
