#!/usr/bin/env python3
r"""Generates lkgd_amd/csrc/attn_spatial_pipe.inc: the software-pipelined main loop of attn_spatial_pipe.hip as ONE inline-asm
statement.  Run from the repo root:  python tools/gen_attn_asm.py   (add --stats for the per-phase issue-cost table).

Why generated asm: the compiler-scheduled kernel (attn_spatial.hip) runs QK^T, softmax and P.V of a 64-key tile one after the
other inside a wave and leaves the overlap to the four waves of a SIMD - 52 % of the vector-issue bound (profiles/
r02_pmc_attn_spatial.txt).  Here every MFMA issue slot of a wave carries ~30 cycles of independent softmax work of the OTHER
query tile of the same wave, so one wave alone keeps the SIMD's vector issue port busy.

A wave owns TWO 32-query tiles A, B.  A "unit" u = (tile, 64-key sub-tile); units are visited A0 B0 A1 B1 ...  Phase n:
    VALU : softmax(u_n)          32 v_exp_f32 + 32 adds (row sum) + 16 cvt_pk (P -> fp16 B operand) on tile T = n & 1
    MFMA : QK^T(u_{n+1})         2 bias k-steps (A = e0, B = -m: the reference maximum is subtracted by the matrix pipe) + 8
           P.V (u_{n-1})         8                                                     ... both on tile T' = the other tile
    The reference maximum m of a tile is checked AFTER the exponentials, on the unit's row-sum partial (free: the adds are
    there anyway; 16 v_max3 per unit in front of the exponentials cost 0.5 ms of 2.9 at S = 9216, profiles/
    r04_attn_pipe_knobs.txt): while every lane's partial stays <= 1024 no probability exceeds 2^10 (fp16 holds 2^16) and
    nothing happens.  Otherwise (rare: a score jumped by > ~7 over the reference) an out-of-line block recomputes the unit
    from LDS - QK^T again, true maximum, new reference, exponentials, sums, conversions - and rescales O and l.
K / V^T fragments stream through a ring of four 4-register slots, read four MFMAs ahead (counted lgkmcnt).
Stages of 128 keys (two sub-tiles) live in a three-buffer LDS ring filled by LDS-DMA one stage ahead; ONE barrier per stage.

    PRO : wait stage 0, barrier, issue stage 1;  P0: QK_A(0), first reference of A;  P1: sm_A(0) | QK_B(0), first reference of B
    LOOP(s): beta(2s)   : sm_B(2s)   | QK_A(2s+1) [K s, sub 1]   PV_A(2s)   [V s, sub 0]
             alpha(2s+1): sm_A(2s+1) | QK_B(2s+1) [K s, sub 1]   PV_B(2s)   [V s, sub 0]
             last stage -> TAIL;  wait own DMA, barrier (stage s+1 visible), issue stage s+2, K addresses -> stage s+1
             beta(2s+1) : sm_B(2s+1) | QK_A(2s+2) [K s+1, sub 0] PV_A(2s+1) [V s, sub 1]
             alpha(2s+2): sm_A(2s+2) | QK_B(2s+2) [K s+1, sub 0] PV_B(2s+1) [V s, sub 1];  V addresses -> stage s+1
    TAIL: sm_B(last) | PV_A(last);  PV_B(last)

Register plan.  Arch VGPRs v[VB ...] are named here and listed as clobbers (the compiler keeps its operands below VB):
    S_A, S_B   32 each   score accumulators (MFMA destinations, exponentials in place)
    PF_A, PF_B 16 each   packed fp16 probabilities = B operands of P.V
    RING       16        K / V^T fragment ring
    L 2 x 2 partial row sums, MB 2 reference maxima, MX 2 partial tile maxima, T0..T3 scratch of the rare path
AGPRs (invisible to the compiler): a[0:31] O_A, a[32:63] O_B, a[64:79] Q_A, a[80:95] Q_B (pre-scaled, written by the kernel
before the statement), a[96:99] bias A operand (1.0 in k-slot 0), a[100:103] / a[104:107] bias B operands (-m_A / -m_B).

The generator keeps an instruction list with register reads / writes and checks it (check()): MFMA result -> any other use
>= 20 wait states, VALU write -> MFMA read >= 3, transcendental -> consumer >= 1, every LDS fragment waited for by a counted
lgkmcnt before its MFMA, ring slots holding the fragment the MFMA expects.
"""
import os
import sys

NL = r"\n\t"
VB = 24
S_ = {"A": VB, "B": VB + 32}
PF = {"A": VB + 64, "B": VB + 80}
RING = VB + 96
L_ = {"A": VB + 112, "B": VB + 113}          # running row sums
PS0, PS1 = VB + 114, VB + 115                 # the current unit's row-sum partials (even / odd scores: adjacent adds independent)
MB = {"A": VB + 116, "B": VB + 117}
MX, MX2 = VB + 118, VB + 119                  # two partial tile maxima (one tile at a time is between its max and its compare)
T0, T1, T2, T3 = VB + 120, VB + 121, VB + 122, VB + 123
VEND = VB + 124
OACC = {"A": 0, "B": 32}
QF = {"A": 64, "B": 80}
ABIAS = 96
BB = {"A": 100, "B": 104}
AEND = 108
# named SGPRs (clobbered): DMA source pointers, counters
SK, SV, SIT, SDST, SKPOS, SVPOS, STMP, STMP2, SP1, SKD = 60, 62, 64, 65, 66, 67, 68, 69, 70, 72   # SP1 = s[70:71]
SMASK = {"A": 73, "B": 74}     # "mask" variant: keys m of the tile's next 32-key tiles with m + 32 f < SMASK are duplicates
STMP3 = 75
SEND = 76 if "mask" in os.environ.get("ATTN_GEN_OPT", "w2").split("+") else 73
STAGE = 32768
NBUF = 3
OPT = os.environ.get("ATTN_GEN_OPT", "w2").split("+")   # schedule options (A/B: tools/micro/attn_pipe_opts.sh)
PSUM_LIMIT = "0x44800000"   # 1024.0
# "mask": the variant for S % 128 != 0 (written to attn_spatial_pipe_masked.inc).  The LAST 128-key stage is loaded from keys
# [S - 128, S): it overlaps the stage before it by dup = 128 - S % 128 keys, which are masked - a second k-slot of the bias
# k-step carries 1.0 on the A side for exactly those key rows and -30000 on the B side - so nothing is read beyond S.
MASKED = "mask" in OPT


def v(n):
    return "v%d" % n


def vr(a, n):
    return "v[%d:%d]" % (a, a + n - 1)


def ar(a, n):
    return "a[%d:%d]" % (a, a + n - 1)


class Ins:
    __slots__ = ("text", "kind", "rd", "wr", "meta")

    def __init__(self, text, kind, rd=(), wr=(), **meta):
        self.text, self.kind, self.rd, self.wr, self.meta = text, kind, tuple(rd), tuple(wr), meta


def R(base, n, f="v"):
    return [(f, base + i) for i in range(n)]


class Gen:
    def __init__(self):
        self.ins = []
        self.lbl = 0

    def e(self, text, kind, rd=(), wr=(), **meta):
        self.ins.append(Ins(text, kind, rd, wr, **meta))

    def label(self, name):
        self.e(name + "_%=:", "label", name=name)

    def nop(self, n):
        self.e("s_nop %d" % n, "nop", n=n)

    def ckpt(self):
        """bisection aid (ATTN_GEN_STOP=k python tools/gen_attn_asm.py): leave the statement at the k-th checkpoint"""
        self.nck = getattr(self, "nck", 0) + 1
        if os.environ.get("ATTN_GEN_STOP") and int(os.environ["ATTN_GEN_STOP"]) == self.nck:
            self.e("s_branch END_%=", "branch", target="END")

    # ---- fragment reads ------------------------------------------------------------------------------------------------
    def read_frag(self, frag, slot):
        reg = RING + 4 * slot
        if frag[0] == "K":
            _, sub, f, ks = frag
            self.e("ds_read_b128 %s, %%[ka%d] offset:%d" % (vr(reg, 4), ks, sub * 8192 + f * 4096), "lds", wr=R(reg, 4),
                   frag=frag, nops=1)
        else:
            _, sub, f, ss, df = frag
            off = 16384 + sub * 8192 + (32 * f + 16 * ss) * 128
            self.e("ds_read_b64_tr_b16 %s, %%[va%d] offset:%d" % (vr(reg, 2), df, off), "lds", wr=R(reg, 2), frag=frag, nops=1)
            self.e("ds_read_b64_tr_b16 %s, %%[va%d] offset:%d" % (vr(reg + 2, 2), df, off + 1024), "lds", wr=R(reg + 2, 2),
                   frag=frag, nops=1)

    def bias_mask(self, t, f, extra):
        """mask variant: a96 <- (1.0 in k-slot 0 | 1.0 in k-slot 1 where key row m of tile f is a duplicate: m + 32 f < SMASK)"""
        e = self.e
        if f or extra:
            e("s_sub_i32 s%d, s%d, %d" % (STMP3, SMASK[t], 32 * f - extra), "salu")
        src = STMP3 if (f or extra) else SMASK[t]
        e("v_mov_b32_e32 %s, 0x3c00" % v(MX), "valu", wr=[("v", MX)])
        e("v_mov_b32_e32 %s, 0x3c003c00" % v(MX2), "valu", wr=[("v", MX2)])
        e("v_cmp_gt_i32_e32 vcc, s%d, %%[vm]" % src, "valu", wr=[("vcc", 0)])
        e("s_nop 1", "nop", n=1)       # (the previous bias MFMA has read a96 by now)
        e("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v(T0), v(MX), v(MX2)), "valu", rd=[("vcc", 0), ("v", MX), ("v", MX2)], wr=[("v", T0)])
        e("v_and_b32_e32 %s, %%[hm32], %s" % (v(T0), v(T0)), "valu", rd=[("v", T0)], wr=[("v", T0)])     # k-group 0 lanes only
        e("v_accvgpr_write_b32 a%d, %s" % (ABIAS, v(T0)), "valu", rd=[("v", T0)], wr=[("a", ABIAS)])
        e("s_nop 1", "nop", n=1)

    # ---- one phase -----------------------------------------------------------------------------------------------------
    def phase(self, name, sm, mm, do_qk, do_pv, ksub, vsub, prefetched, nextfrags):
        """sm: tile whose softmax runs (or None); mm: tile of the MFMAs; prefetched: the first four fragments are already in
        flight; nextfrags: fragments of the following phase to read behind this phase's last four fragment MFMAs (or None)"""
        self.e("; ---- phase %s: softmax %s | mfma tile %s qk=%d pv=%d" % (name, sm, mm, do_qk, do_pv), "comment")
        mf = []            # (text, frag or None, rd, wr)
        sm_ = S_[mm] if mm else 0
        if do_qk:
            for f in range(2):
                mf.append(("v_mfma_f32_32x32x16_f16 %s, %s, %s, 0" % (vr(sm_ + 16 * f, 16), ar(ABIAS, 4), ar(BB[mm], 4)), None,
                           R(ABIAS, 4, "a") + R(BB[mm], 4, "a"), R(sm_ + 16 * f, 16)))
            for ks in range(4):
                for f in range(2):
                    mf.append(("QK", ("K", ksub, f, ks), f, ks))
        if do_pv:
            for f in range(2):
                for ss in range(2):
                    for df in range(2):
                        mf.append(("PV", ("V", vsub, f, ss, df), f, ss, df))
        frags = [m[1] for m in mf if m[1] is not None]
        nfr = len(frags)
        assert nfr % 4 == 0
        # ---- filler queue
        fill = []
        if sm:
            s0, p0 = S_[sm], PF[sm]
            def E(r):
                return ("E", "v_exp_f32_e32 %s, %s" % (v(s0 + r), v(s0 + r)), [("v", s0 + r)], [("v", s0 + r)], 8)
            def C(q):
                return ("C", "v_cvt_pk_f16_f32 %s, %s, %s" % (v(p0 + q), v(s0 + 2 * q), v(s0 + 2 * q + 1)),
                        [("v", s0 + 2 * q), ("v", s0 + 2 * q + 1)], [("v", p0 + q)], 4)
            def A2(q):
                if "pkadd" in OPT and q:      # experiment: one packed add per pair (v_pk_add_f32 beside MFMAs)
                    return [("A", "v_pk_add_f32 %s, %s, %s" % (vr(PS0, 2), vr(PS0, 2), vr(s0 + 2 * q, 2)),
                             R(PS0, 2) + R(s0 + 2 * q, 2), R(PS0, 2), 5)]
                out = []
                for r in (2 * q, 2 * q + 1):
                    ps = PS0 + (r & 1)
                    if q == 0:
                        out.append(("A", "v_add_f32_e32 %s, 0, %s" % (v(ps), v(s0 + r)), [("v", s0 + r)], [("v", ps)], 4))
                    else:
                        out.append(("A", "v_add_f32_e32 %s, %s, %s" % (v(ps), v(ps), v(s0 + r)), [("v", ps), ("v", s0 + r)], [("v", ps)], 4))
                return out
            if "pkadd" in OPT:                # the pair's sum and conversion run one pair behind its exponentials
                for q in range(16):
                    fill += [E(2 * q), E(2 * q + 1)]
                    if q:
                        fill += A2(q - 1) + [C(q - 1)]
                fill += A2(15) + [C(15)]
            else:
                for q in range(16):
                    fill += [E(2 * q), E(2 * q + 1)] + A2(q) + [C(q)]
        ngaps = max(len(mf), 1)
        fi = 0
        if not prefetched:
            for j in range(min(4, nfr)):
                self.read_frag(frags[j], j % 4)
        frn = 0
        for g in range(ngaps):
            if mf:
                m = mf[g]
                if m[1] is not None:
                    slot = frn % 4
                    reg = RING + 4 * slot
                    if "w2" not in OPT or frn % 2 == 0:     # w2: one counted wait per TWO fragment MFMAs
                        self.e("WAITFRAG", "waitfrag", frag=frags[frn + 1] if "w2" in OPT else m[1])
                    if m[0] == "QK":
                        _, fr, f, ks = m
                        d = sm_ + 16 * f
                        self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), vr(reg, 4), ar(QF[mm] + 4 * ks, 4), vr(d, 16)),
                               "mfma", rd=R(reg, 4) + R(QF[mm] + 4 * ks, 4, "a") + R(d, 16), wr=R(d, 16), frag=fr, acc=True)
                    else:
                        _, fr, f, ss, df = m
                        d = OACC[mm] + 16 * df
                        b = PF[mm] + 8 * f + 4 * ss
                        self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (ar(d, 16), vr(reg, 4), vr(b, 4), ar(d, 16)),
                               "mfma", rd=R(reg, 4) + R(b, 4) + R(d, 16, "a"), wr=R(d, 16, "a"), frag=fr, acc=True)
                    nxt = frn + 4
                    if nxt < nfr:
                        self.read_frag(frags[nxt], slot)
                    elif nextfrags is not None and nxt - nfr < len(nextfrags):
                        self.read_frag(nextfrags[nxt - nfr], slot)
                    frn += 1
                else:
                    if MASKED:
                        self.bias_mask(mm, g, 0)
                    self.e(m[0], "mfma", rd=m[2], wr=m[3], frag=None, acc=False)
            # fillers of this gap: an equal share of what is left
            lastgap = g == ngaps - 1
            budget = sum(x[4] for x in fill[fi:]) / float(ngaps - g)
            used = 0.0
            while fi < len(fill):
                it = fill[fi]
                if not lastgap and used + it[4] > budget + 2.0:
                    break
                self.e(it[1], "trans" if it[0] == "E" else "valu", rd=it[2], wr=it[3], cost=it[4])
                used += it[4]
                fi += 1
        assert fi == len(fill)

    # ---- reference maximum of a tile: prologue (first reference) and the rare recompute of a unit --------------------------
    def maxima(self, t):
        """per-lane maximum of the 32 scores of tile t -> MX (two chains, then joined)"""
        sb = S_[t]
        def m3(d, a, b, c):
            self.e("v_max3_f32 %s, %s, %s, %s" % (v(d), v(a), v(b), v(c)), "valu", rd=[("v", x) for x in (a, b, c)], wr=[("v", d)])
        ca = [(MX, sb, sb + 1, sb + 2)] + [(MX, MX, sb + r, sb + r + 1) for r in range(3, 15, 2)]
        cb = [(MX2, sb + 15, sb + 16, sb + 17)] + [(MX2, MX2, sb + r, sb + r + 1) for r in range(18, 30, 2)]
        for x, y in zip(ca, cb):
            m3(*x)
            m3(*y)
        m3(MX, MX, sb + 30, sb + 31)
        self.e("v_max_f32_e32 %s, %s, %s" % (v(MX), v(MX), v(MX2)), "valu", rd=[("v", MX), ("v", MX2)], wr=[("v", MX)])

    def newref(self, t, first):
        """MX (scores relative to the current reference) -> new reference m (fp16-exact), T2 = delta, T3 = alpha = 2^-delta,
        bias operand; the scores are brought to the new reference"""
        s0, mb = S_[t], MB[t]
        e = self.e
        e("ds_bpermute_b32 %s, %%[xora], %s" % (v(T0), v(MX)), "lds", rd=[("v", MX)], wr=[("v", T0)], frag=("X",), nops=1)
        e("s_waitcnt lgkmcnt(0)", "waitall")                       # (the two lane halves hold disjoint keys of the same query)
        e("v_max_f32_e32 %s, %s, %s" % (v(T0), v(T0), v(MX)), "valu", rd=[("v", T0), ("v", MX)], wr=[("v", T0)])
        e("v_add_f32_e32 %s, %s, %s" % (v(T0), v(T0), v(mb)), "valu", rd=[("v", T0), ("v", mb)], wr=[("v", T0)])
        if not first:                                              # the reference never moves down after the first tile
            e("v_max_f32_e32 %s, %s, %s" % (v(T0), v(T0), v(mb)), "valu", rd=[("v", T0), ("v", mb)], wr=[("v", T0)])
        e("v_max_f32_e32 %s, 0xc76a6000, %s" % (v(T0), v(T0)), "valu", rd=[("v", T0)], wr=[("v", T0)])     # -60000
        e("v_min_f32_e32 %s, 0x476a6000, %s" % (v(T0), v(T0)), "valu", rd=[("v", T0)], wr=[("v", T0)])     # +60000
        e("v_cvt_f16_f32_e32 %s, %s" % (v(T1), v(T0)), "valu", rd=[("v", T0)], wr=[("v", T1)])
        e("v_cvt_f32_f16_e32 %s, %s" % (v(T1), v(T1)), "valu", rd=[("v", T1)], wr=[("v", T1)])             # m_new, fp16-exact
        e("v_sub_f32_e32 %s, %s, %s" % (v(T2), v(T1), v(mb)), "valu", rd=[("v", T1), ("v", mb)], wr=[("v", T2)])   # delta
        if not first:
            e("v_exp_f32_e64 %s, -%s" % (v(T3), v(T2)), "trans", rd=[("v", T2)], wr=[("v", T3)])           # alpha
        e("v_mov_b32_e32 %s, %s" % (v(mb), v(T1)), "valu", rd=[("v", T1)], wr=[("v", mb)])
        e("v_cvt_f16_f32_e64 %s, -%s" % (v(T0), v(T1)), "valu", rd=[("v", T1)], wr=[("v", T0)])
        e("v_and_b32_e32 %s, %%[hmask], %s" % (v(T0), v(T0)), "valu", rd=[("v", T0)], wr=[("v", T0)])
        if MASKED:
            e("v_and_b32_e32 %s, 0xf7530000, %%[hm32]" % v(MX2), "valu", wr=[("v", MX2)])          # -30000 in k-slot 1
            e("v_or_b32_e32 %s, %s, %s" % (v(T0), v(MX2), v(T0)), "valu", rd=[("v", T0), ("v", MX2)], wr=[("v", T0)])
        e("v_accvgpr_write_b32 a%d, %s" % (BB[t], v(T0)), "valu", rd=[("v", T0)], wr=[("a", BB[t])])
        for r in range(32):
            e("v_sub_f32_e32 %s, %s, %s" % (v(s0 + r), v(s0 + r), v(T2)), "valu", rd=[("v", s0 + r), ("v", T2)], wr=[("v", s0 + r)])

    def firstref(self, t):
        """prologue: the scores of the tile's first unit have just been computed against m = 0"""
        self.nop(15)
        self.nop(15)
        self.maxima(t)
        self.newref(t, True)
        self.nop(3)

    def redo(self, t, tag, back, ksub):
        """out of line: the unit whose softmax has just run on tile t overflowed the window above its reference.  Its
        scores are computed again from the K sub-tile still in LDS (fragments through the tile's P registers), the
        reference moves to the true maximum, O and l follow, and the unit's exponentials / sums / conversions run again."""
        s0, p0, l0 = S_[t], PF[t], L_[t]
        e = self.e
        if back:          # the K addresses already point one stage ahead of this unit's
            ka = [v(T0 + ks) for ks in range(4)]
            for ks in range(4):
                e("v_subrev_u32_e32 %s, s%d, %%[ka%d]" % (ka[ks], SKD, ks), "valu", wr=[("v", T0 + ks)])
        else:
            ka = ["%%[ka%d]" % ks for ks in range(4)]
        for half in range(2):
            if half:
                self.nop(7)
            for i, (ks, f) in enumerate(((2 * half, 0), (2 * half, 1), (2 * half + 1, 0), (2 * half + 1, 1))):
                e("ds_read_b128 %s, %s offset:%d" % (vr(p0 + 4 * i, 4), ka[ks], ksub * 8192 + f * 4096), "lds",
                  rd=[("v", T0 + ks)] if back else [], wr=R(p0 + 4 * i, 4), frag=("X",), nops=1)
            if not half:
                for f in range(2):
                    if MASKED:
                        self.bias_mask(t, f, 0)
                        self.nop(1)
                    e("v_mfma_f32_32x32x16_f16 %s, %s, %s, 0" % (vr(s0 + 16 * f, 16), ar(ABIAS, 4), ar(BB[t], 4)), "mfma",
                      rd=R(ABIAS, 4, "a") + R(BB[t], 4, "a"), wr=R(s0 + 16 * f, 16), frag=None, acc=False)
            e("s_waitcnt lgkmcnt(0)", "waitall")
            for i, (ks, f) in enumerate(((2 * half, 0), (2 * half, 1), (2 * half + 1, 0), (2 * half + 1, 1))):
                d = s0 + 16 * f
                e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), vr(p0 + 4 * i, 4), ar(QF[t] + 4 * ks, 4), vr(d, 16)), "mfma",
                  rd=R(p0 + 4 * i, 4) + R(QF[t] + 4 * ks, 4, "a") + R(d, 16), wr=R(d, 16), frag=None, acc=True)
        self.nop(15)
        self.nop(15)
        self.maxima(t)
        self.newref(t, False)
        for r in range(32):
            e("v_exp_f32_e32 %s, %s" % (v(s0 + r), v(s0 + r)), "trans", rd=[("v", s0 + r)], wr=[("v", s0 + r)])
        for r in range(32):
            ps = PS0 + (r & 1)
            if r < 2:
                e("v_add_f32_e32 %s, 0, %s" % (v(ps), v(s0 + r)), "valu", rd=[("v", s0 + r)], wr=[("v", ps)])
            else:
                e("v_add_f32_e32 %s, %s, %s" % (v(ps), v(ps), v(s0 + r)), "valu", rd=[("v", ps), ("v", s0 + r)], wr=[("v", ps)])
        for q in range(16):
            e("v_cvt_pk_f16_f32 %s, %s, %s" % (v(p0 + q), v(s0 + 2 * q), v(s0 + 2 * q + 1)), "valu",
              rd=[("v", s0 + 2 * q), ("v", s0 + 2 * q + 1)], wr=[("v", p0 + q)])
        e("v_add_f32_e32 %s, %s, %s" % (v(PS0), v(PS0), v(PS1)), "valu", rd=[("v", PS0), ("v", PS1)], wr=[("v", PS0)])
        e("v_mul_f32_e32 %s, %s, %s" % (v(l0), v(l0), v(T3)), "valu", rd=[("v", l0), ("v", T3)], wr=[("v", l0)])
        for r in range(32):
            a = OACC[t] + r
            e("v_accvgpr_read_b32 %s, a%d" % (v(T0), a), "valu", rd=[("a", a)], wr=[("v", T0)])
            e("v_mul_f32_e32 %s, %s, %s" % (v(T0), v(T0), v(T3)), "valu", rd=[("v", T0), ("v", T3)], wr=[("v", T0)])
            e("v_accvgpr_write_b32 a%d, %s" % (a, v(T0)), "valu", rd=[("v", T0)], wr=[("a", a)])
        self.nop(3)
        e("s_branch CONT%s_%%=" % tag, "branch", target="CONT" + tag)

    def boundary(self, t, tag, back, ksub, slow):
        """end of the softmax of a unit of tile t: the unit's row-sum partial decides (NaN included) whether the unit is redone"""
        e = self.e
        e("v_add_f32_e32 %s, %s, %s" % (v(PS0), v(PS0), v(PS1)), "valu", rd=[("v", PS0), ("v", PS1)], wr=[("v", PS0)])
        if tag is not None:
            e("v_cmp_nge_f32_e32 vcc, %s, %s" % (PSUM_LIMIT, v(PS0)), "valu", rd=[("v", PS0)], wr=[("vcc", 0)])
            e("s_cbranch_vccnz REDO%s_%%=" % tag, "branch", target="REDO" + tag)
            self.label("CONT" + tag)
            slow.append((t, tag, back, ksub))
        e("v_add_f32_e32 %s, %s, %s" % (v(L_[t]), v(L_[t]), v(PS0)), "valu", rd=[("v", L_[t]), ("v", PS0)], wr=[("v", L_[t])])

    # ---- DMA of one 128-key stage (this wave's four 1-KiB pieces) --------------------------------------------------------
    def dma_stage(self, nxt_is_last):
        """nxt_is_last (mask variant): SALU text that sets SCC = "the stage AFTER the one issued here is the last one" """
        s = self.e
        s("s_mov_b32 m0, s%d" % SDST, "salu")
        s("s_add_u32 s%d, s%d, %%[kp1]" % (SP1, SK), "salu")
        s("global_load_lds_dwordx4 %%[vok], s[%d:%d]" % (SK, SK + 1), "vmem")
        s("s_addc_u32 s%d, s%d, 0" % (SP1 + 1, SK + 1), "salu")
        s("s_add_u32 m0, s%d, 8192" % SDST, "salu")
        s("s_nop 0", "nop", n=0)
        s("global_load_lds_dwordx4 %%[vok], s[%d:%d]" % (SP1, SP1 + 1), "vmem")
        s("s_add_u32 m0, s%d, 16384" % SDST, "salu")
        s("s_add_u32 s%d, s%d, %%[vp1]" % (SP1, SV), "salu")
        s("global_load_lds_dwordx4 %%[vov], s[%d:%d]" % (SV, SV + 1), "vmem")
        s("s_addc_u32 s%d, s%d, 0" % (SP1 + 1, SV + 1), "salu")
        s("s_add_u32 m0, s%d, 24576" % SDST, "salu")
        s("s_nop 0", "nop", n=0)
        s("global_load_lds_dwordx4 %%[vov], s[%d:%d]" % (SP1, SP1 + 1), "vmem")
        # advance the sources to the next stage, rotate the destination buffer
        if MASKED:                    # the last stage starts at key S - 128, not at a multiple of 128
            for t in nxt_is_last:
                s(t, "salu")
            s("s_cselect_b32 s%d, %%[klast], %%[kstr]" % STMP, "salu")
            s("s_cselect_b32 s%d, %%[vlast], %%[vstr]" % STMP3, "salu")
            s("s_add_u32 s%d, s%d, s%d" % (SK, SK, STMP), "salu")
            s("s_addc_u32 s%d, s%d, 0" % (SK + 1, SK + 1), "salu")
            s("s_add_u32 s%d, s%d, s%d" % (SV, SV, STMP3), "salu")
            s("s_addc_u32 s%d, s%d, 0" % (SV + 1, SV + 1), "salu")
        else:
            s("s_add_u32 s%d, s%d, %%[kstr]" % (SK, SK), "salu")
            s("s_addc_u32 s%d, s%d, 0" % (SK + 1, SK + 1), "salu")
            s("s_add_u32 s%d, s%d, %%[vstr]" % (SV, SV), "salu")
            s("s_addc_u32 s%d, s%d, 0" % (SV + 1, SV + 1), "salu")
        s("s_add_u32 s%d, s%d, %d" % (SDST, SDST, STAGE), "salu")
        s("s_sub_u32 s%d, s%d, %d" % (STMP, SDST, NBUF * STAGE), "salu")
        s("s_cmp_ge_u32 s%d, %%[ldsend]" % SDST, "salu")
        s("s_cselect_b32 s%d, s%d, s%d" % (SDST, STMP, SDST), "salu")

    def advance(self, pos, names, keep):
        """fragment addresses -> next LDS buffer of the ring (pos = byte offset of the buffer they point into); keep: an
        SGPR that remembers the step (the recompute path of a unit one stage back subtracts it again)"""
        s = self.e
        s("s_add_u32 s%d, s%d, %d" % (pos, pos, STAGE), "salu")
        s("s_cmp_eq_u32 s%d, %d" % (pos, NBUF * STAGE), "salu")
        s("s_mov_b32 s%d, %d" % (STMP, STAGE), "salu")
        s("s_cselect_b32 s%d, 0x%x, s%d" % (STMP, (-(NBUF - 1) * STAGE) & 0xffffffff, STMP), "salu")
        s("s_cselect_b32 s%d, 0, s%d" % (pos, pos), "salu")
        if keep is not None:
            s("s_mov_b32 s%d, s%d" % (keep, STMP), "salu")
        for nm in names:
            s("v_add_u32_e32 %%[%s], s%d, %%[%s]" % (nm, STMP, nm), "valu", rd=[("op", nm)], wr=[("op", nm)])

    # ---- the whole statement ---------------------------------------------------------------------------------------------
    def build(self):
        s = self.e
        slow = []
        self.ckpt()                                   # 1: nothing executed
        K = lambda sub: [("K", sub, f, ks) for ks in range(2) for f in range(2)]     # first four fragments of a QK^T
        V = lambda sub: [("V", sub, 0, ss, df) for ss in range(2) for df in range(2)]
        # ---- init
        s("s_mov_b32 s%d, %%[klo]" % SK, "salu")
        s("s_mov_b32 s%d, %%[khi]" % (SK + 1), "salu")
        s("s_mov_b32 s%d, %%[vlo]" % SV, "salu")
        s("s_mov_b32 s%d, %%[vhi]" % (SV + 1), "salu")
        s("s_mov_b32 s%d, %%[dst1]" % SDST, "salu")
        s("s_mov_b32 s%d, 0" % SIT, "salu")
        s("s_mov_b32 s%d, 0" % SKPOS, "salu")
        s("s_mov_b32 s%d, 0" % SVPOS, "salu")
        for a in range(64):
            s("v_accvgpr_write_b32 a%d, 0" % a, "valu", wr=[("a", a)])
        for a in list(range(ABIAS + 1, ABIAS + 4)) + list(range(BB["A"], BB["A"] + 4)) + list(range(BB["B"], BB["B"] + 4)):
            s("v_accvgpr_write_b32 a%d, 0" % a, "valu", wr=[("a", a)])
        s("v_and_b32_e32 %s, 0x3c00, %%[hmask]" % v(T0), "valu", wr=[("v", T0)])
        s("v_accvgpr_write_b32 a%d, %s" % (ABIAS, v(T0)), "valu", rd=[("v", T0)], wr=[("a", ABIAS)])
        for t in "AB":
            s("v_mov_b32_e32 %s, 0" % v(L_[t]), "valu", wr=[("v", L_[t])])
            s("v_mov_b32_e32 %s, 0" % v(MB[t]), "valu", wr=[("v", MB[t])])
        s("s_mov_b32 s%d, 0" % SKD, "salu")
        if MASKED:                                    # (at least two stages: stage 0 is never the last one)
            s("s_mov_b32 s%d, 0" % SMASK["A"], "salu")
            s("s_mov_b32 s%d, 0" % SMASK["B"], "salu")
        self.ckpt()                                   # 2: after the register initialisation
        # ---- stage 0 has been issued by the kernel; publish it, start stage 1
        s("s_waitcnt vmcnt(0)", "waitvm")
        s("s_barrier", "barrier")
        self.ckpt()                                   # 3: after the first barrier
        s("s_cmp_lt_u32 1, %[nst]", "salu")
        s("s_cbranch_scc0 NOST1_%=", "branch", target="NOST1")
        self.dma_stage(["s_cmp_eq_u32 %[nst], 3"])    # stage 1 issued here; stage 2 is the last one iff nst == 3
        self.label("NOST1")
        self.ckpt()                                   # 4: after the DMA of stage 1
        # ---- P0: QK_A(0), first reference of A.  P1 = alpha(0): sm_A(0) | QK_B(0), first reference of B
        self.phase("P0", None, "A", True, False, 0, 0, False, K(0))
        self.ckpt()                                   # 5: after P0
        self.firstref("A")
        self.ckpt()                                   # 6: after the first reference of A
        self.phase("P1", "A", "B", True, False, 0, 0, True, K(1))
        self.boundary("A", None, False, 0, slow)      # (scores <= 0 after the first reference: no check)
        self.firstref("B")
        self.ckpt()                                   # 7: after P1 and the first reference of B
        # ---- loop over stages.  boundary(tile, tag, back, ksub): where the K sub-tile of the unit just exponentiated lives
        self.label("LOOP")
        self.ckpt()                                   # 8
        if MASKED:                                    # second unit of a stage: 64 keys further
            s("s_sub_i32 s%d, s%d, 64" % (SMASK["A"], SMASK["A"]), "salu")
        self.phase("beta(2s)", "B", "A", True, True, 1, 0, True, K(1))
        self.ckpt()                                   # 9
        self.boundary("B", "0", False, 0, slow)
        self.ckpt()                                   # 10
        if MASKED:
            s("s_sub_i32 s%d, s%d, 64" % (SMASK["B"], SMASK["B"]), "salu")
        self.phase("alpha(2s+1)", "A", "B", True, True, 1, 0, True, None)
        self.boundary("A", "1", False, 1, slow)
        self.ckpt()                                   # 11
        s("s_add_u32 s%d, s%d, 1" % (STMP2, SIT), "salu")
        s("s_cmp_eq_u32 s%d, %%[nst]" % STMP2, "salu")
        s("s_cbranch_scc1 TAIL_%=", "branch", target="TAIL")
        s("s_waitcnt vmcnt(0)", "waitvm")
        s("s_barrier", "barrier")
        s("s_add_u32 s%d, s%d, 2" % (STMP2, SIT), "salu")
        s("s_cmp_lt_u32 s%d, %%[nst]" % STMP2, "salu")
        s("s_cbranch_scc0 NODMA_%=", "branch", target="NODMA")
        self.dma_stage(["s_add_u32 s%d, s%d, 4" % (STMP2, SIT), "s_cmp_eq_u32 s%d, %%[nst]" % STMP2])   # stage s+2 issued; s+3 last?
        self.label("NODMA")
        self.advance(SKPOS, ["ka0", "ka1", "ka2", "ka3"], SKD)
        if MASKED:                                    # first unit of stage s+1: the last stage starts with dup duplicates
            s("s_add_u32 s%d, s%d, 2" % (STMP2, SIT), "salu")
            s("s_cmp_eq_u32 s%d, %%[nst]" % STMP2, "salu")
            s("s_cselect_b32 s%d, %%[dup], 0" % SMASK["A"], "salu")
        self.phase("beta(2s+1)", "B", "A", True, True, 0, 1, False, K(0))
        self.boundary("B", "2", True, 1, slow)        # B(2s+1): K of stage s, the addresses are at stage s+1
        if MASKED:
            s("s_mov_b32 s%d, s%d" % (SMASK["B"], SMASK["A"]), "salu")
        self.phase("alpha(2s+2)", "A", "B", True, True, 0, 1, True, K(1))
        self.boundary("A", "3", False, 0, slow)
        self.advance(SVPOS, ["va0", "va1"], None)
        s("s_add_u32 s%d, s%d, 1" % (SIT, SIT), "salu")
        s("s_branch LOOP_%=", "branch", target="LOOP")
        # ---- tail
        self.label("TAIL")
        self.ckpt()                                   # 12
        self.phase("tail beta", "B", "A", False, True, 0, 1, False, V(1))
        self.boundary("B", "4", False, 1, slow)
        self.ckpt()                                   # 13
        self.phase("tail pv_B", None, "B", False, True, 0, 1, True, None)
        self.ckpt()                                   # 14
        self.nop(15)
        self.nop(15)
        s("v_mov_b32_e32 %%[la], %s" % v(L_["A"]), "valu", rd=R(L_["A"], 1))
        s("v_mov_b32_e32 %%[lb], %s" % v(L_["B"]), "valu", rd=R(L_["B"], 1))
        s("s_branch END_%=", "branch", target="END")
        # ---- out-of-line: units whose scores left the window above their reference
        for t, tag, back, ksub in slow:
            self.label("REDO" + tag)
            self.redo(t, tag, back, ksub)
        self.label("END")

    # ---- resolve the WAITFRAG markers into counted lgkmcnt waits ----------------------------------------------------------
    def resolve_waits(self):
        self.loop_entry_fifo = [("K", 1, f, ks) for ks in range(2) for f in range(2)]
        """Walks the list in emission order (the fall-through path: prologue, loop body, tail).  Branch targets inherit a
        conservative state: a wait for N outstanding is still correct when fewer are outstanding."""
        out = []
        fifo = []          # fragments of outstanding LDS ops, oldest first
        for i in self.ins:
            if i.kind == "lds":
                fifo.append(i.meta["frag"])
                out.append(i)
            elif i.kind == "waitall":
                fifo = []
                out.append(i)
            elif i.kind == "waitfrag":
                fr = i.meta["frag"]
                idx = [k for k, f in enumerate(fifo) if f == fr]
                if idx:
                    keep = len(fifo) - 1 - idx[-1]
                    assert keep <= 15
                    out.append(Ins("s_waitcnt lgkmcnt(%d)" % keep, "waitlgkm", n=keep))
                    fifo = fifo[idx[-1] + 1:]
                # else: already waited for
            elif i.kind == "label" and i.meta["name"] == "LOOP":
                # entered from the prologue (everything landed) and from the back edge (the next phase's first four
                # fragments in flight): count for the back edge, which is also correct for the other entry
                assert all(f in self.loop_entry_fifo for f in fifo)
                fifo = list(self.loop_entry_fifo)
                out.append(i)
            elif i.kind == "branch" and i.meta["target"] == "LOOP":
                assert fifo == self.loop_entry_fifo, (fifo, self.loop_entry_fifo)
                out.append(i)
            elif i.kind == "branch" and i.meta["target"] == "TAIL":
                self.tail_fifo = list(fifo)
                out.append(i)
            elif i.kind == "label" and i.meta["name"] == "TAIL":
                fifo = list(self.tail_fifo)          # the path that reaches TAIL
                out.append(i)
            else:
                out.append(i)
        self.ins = out

    # ---- checks --------------------------------------------------------------------------------------------------------
    def check(self):
        """Hazard distances on the straight-line order, with the loop body walked twice (the second walk starts from the
        state the first leaves).  Distances are counted in issued instructions (s_nop n = n + 1)."""
        def ws(i):
            return i.meta["n"] + 1 if i.kind == "nop" else (0 if i.kind in ("label", "comment") else 1)

        seq = [i for i in self.ins]
        a = next(k for k, i in enumerate(seq) if i.kind == "label" and i.meta["name"] == "LOOP")
        b = next(k for k, i in enumerate(seq) if i.kind == "branch" and i.meta["target"] == "LOOP")
        walk = seq[:b] + seq[a:b] + seq[b:]
        last_mfma_wr, last_valu_wr, last_trans_wr = {}, {}, {}
        pos = 0
        nerr = 0
        for i in walk:
            if i.kind in ("label", "comment"):
                continue
            for r in i.rd + i.wr:
                if r in last_mfma_wr:
                    same_chain = i.kind == "mfma" and i.meta.get("acc") and r in i.wr
                    d = pos - last_mfma_wr[r]
                    if not same_chain and d < 20:
                        print("HAZARD mfma->use %s dist %d: %s" % (r, d, i.text))
                        nerr += 1
            if i.kind == "mfma":
                for r in i.rd:
                    if r in last_valu_wr and pos - last_valu_wr[r] < 3:
                        print("HAZARD valu->mfma %s: %s" % (r, i.text))
                        nerr += 1
            if i.kind in ("valu", "trans"):
                for r in i.rd:
                    if r in last_trans_wr and pos - last_trans_wr[r] < 2:
                        print("HAZARD trans->valu %s: %s" % (r, i.text))
                        nerr += 1
            for r in i.wr:
                last_mfma_wr.pop(r, None)
                last_valu_wr.pop(r, None)
                last_trans_wr.pop(r, None)
                if i.kind == "mfma":
                    last_mfma_wr[r] = pos
                elif i.kind == "trans":
                    last_trans_wr[r] = pos
                    last_valu_wr[r] = pos
                elif i.kind == "valu":
                    last_valu_wr[r] = pos
            pos += ws(i)
        # no named register is read before the statement has written it (Q fragments a[64:95] come from the kernel)
        written = set(("a", i) for i in range(QF["A"], QF["B"] + 16))
        for i in self.ins:
            if i.kind in ("label", "comment"):
                continue
            for r in i.rd:
                if r[0] in ("v", "a") and r not in written:
                    print("UNINITIALISED %s read by: %s" % (r, i.text))
                    nerr += 1
                    written.add(r)
            written.update(i.wr)
        # ring contents: the fragment an MFMA consumes is the one last read into its slot, and it has been waited for
        slotfrag, pending = {}, []
        for i in walk:
            if i.kind == "lds" and i.meta["frag"][0] != "X":
                for r in i.wr:
                    slotfrag[r] = i.meta["frag"]
                pending.append(i.meta["frag"])
            elif i.kind == "waitlgkm":
                n = i.meta["n"]
                pending = pending[len(pending) - n:] if n else []
            elif i.kind == "waitall":
                pending = []
            elif i.kind == "mfma" and i.meta.get("frag") is not None:
                regs = [r for r in i.rd if r[0] == "v" and RING <= r[1] < RING + 16]
                assert len(regs) == 4
                for r in regs:
                    if slotfrag.get(r) != i.meta["frag"]:
                        print("RING slot %s holds %s, MFMA expects %s" % (r, slotfrag.get(r), i.meta["frag"]))
                        nerr += 1
                if i.meta["frag"] in pending:
                    print("RING fragment not waited for: %s" % (i.meta["frag"],))
                    nerr += 1
        assert nerr == 0, "%d hazards" % nerr

    def text(self):
        lines = []
        knob = os.environ.get("ATTN_GEN_KNOB", "").split("+")    # timing experiments only (tools/micro/attn_pipe_knobs.sh)
        for i in self.ins:
            if i.kind == "comment":
                continue
            if knob != [""] and i.kind == "branch" and i.meta["target"].startswith("REDO"):
                continue          # every knob build: the reference never moves (removed ingredients leave garbage maxima)
            if "nolds" in knob and (i.kind == "waitlgkm" or (i.kind == "lds" and i.meta["frag"][0] != "X")):
                continue
            if "nokread" in knob and i.kind == "lds" and i.meta["frag"][0] == "K":
                continue
            if "novread" in knob and i.kind == "lds" and i.meta["frag"][0] == "V":
                continue
            if "nowait" in knob and i.kind == "waitlgkm":
                continue
            if "novalu" in knob and i.kind in ("valu", "trans") and "cost" in i.meta:
                continue
            if "noexp" in knob and i.kind == "trans" and "cost" in i.meta:
                continue
            for cls, key in (("noadd", "v_add_f32"), ("nocvt", "v_cvt_pk")):
                if cls in knob and "cost" in i.meta and i.text.startswith(key):
                    i = None
                    break
            if i is None:
                continue
            if "nomfma" in knob and i.kind == "mfma":
                continue
            if "nobar" in knob and i.kind in ("barrier", "waitvm", "vmem"):
                continue
            lines.append('"' + i.text + NL + '"')
        return " \\\n  ".join(lines)

    def stats(self):
        cost = {"mfma": 8, "trans": 8, "valu": 4}
        cur, tot, n = None, {}, {}
        for i in self.ins:
            if i.kind == "comment":
                cur = i.text
                tot[cur] = 0
                n[cur] = {}
            elif cur is not None:
                tot[cur] += cost.get(i.kind, 0)
                n[cur][i.kind] = n[cur].get(i.kind, 0) + 1
        for k in tot:
            print("%-70s vector-issue cycles %4d  %s" % (k, tot[k], n[k]))
        print("instructions:", sum(1 for i in self.ins if i.kind not in ("comment", "label")))


def main():
    g = Gen()
    g.build()
    g.resolve_waits()
    g.check()
    if "--stats" in sys.argv:
        g.stats()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lkgd_amd", "csrc",
                       "attn_spatial_pipe_masked.inc" if MASKED else "attn_spatial_pipe.inc")
    P = "ATTN_PIPEM" if MASKED else "ATTN_PIPE"
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_attn_asm.py - do not edit.  Main loop of attn_spatial_pipe.hip (register plan and\n"
                "// schedule: see that script).\n")
        if not MASKED:
            f.write("#define ATTN_PIPE_VB %d\n#define ATTN_PIPE_VEND %d\n#define ATTN_PIPE_AEND %d\n" % (VB, VEND, AEND))
            f.write("#define ATTN_PIPE_QF_A %d\n#define ATTN_PIPE_QF_B %d\n#define ATTN_PIPE_O_A %d\n#define ATTN_PIPE_O_B %d\n"
                    % (QF["A"], QF["B"], OACC["A"], OACC["B"]))
        f.write("#define %s_ASM \\\n  %s\n\n" % (P, g.text()))
        clob = ['"v%d"' % i for i in range(VB, VEND)] + ['"a%d"' % i for i in range(AEND)] + ['"s%d"' % i for i in range(SK, SEND)]
        f.write("#define %s_CLOBBERS " % P + ", ".join(clob) + ', "vcc", "scc", "m0", "memory"\n')
    print("wrote", out)


if __name__ == "__main__":
    main()
