#!/usr/bin/env python3
r"""Generates lkgd_amd/csrc/ff_fused_loop.inc: the main loop of ff_fused.hip (LayerNorm + GEGLU feed-forward + FF-out of the
72x128 level, C = 320, in ONE kernel) as one inline-asm statement per 128-token panel.
Run from the repo root:  python tools/gen_ff_asm.py   (--stats for the per-chunk issue-cost table)

The computation has the shape of the attention program (tools/gen_attn_asm.py), with hidden channels in the role of keys:
    H^T = W1 . z^T      "swapped" product: A = W1 rows (fragments from LDS), B = z^T = the wave's 32 LayerNorm-ed token rows held
                        as MFMA operands in a[160:239] (20 k-steps of 16 input channels); a lane owns ONE token column
    g   = hidden * gelu(gate)   in registers (exact erf, Abramowitz-Stegun 7.1.26, the form of csrc/common.h::gelu_erf2)
    Y^T += W2 . g^T     the fp16 g IS the B operand; Y^T (320 output channels x 32 tokens) stays in a[0:159] for the whole panel
so the [T, 1280] intermediate (660 MB written and read back, 15 x per forward) never exists.

One wave = 32 tokens, one workgroup = 4 waves (one per SIMD, 512 registers) = a 128-token panel; the two weight matrices stream
L2 -> LDS once per panel in CHUNKS through a ring of six 24-KiB slots, three chunks ahead (LDS-DMA, counted vmcnt), one barrier
per chunk placed four MFMAs before the chunk it opens, so that fragment reads run ahead across chunk borders.  A unit u = 64
hidden channels = two 32-channel tiles f = 0, 1.  Chunk order of the stream (packing.pack_ff_fused writes it in this order):
    unit u:  c0 W1 hidden f0 (u) | c1 W1 gate f0 (u) | c2 W2 f1 (u-1) | c3 W1 hidden f1 (u) | c4 W1 gate f1 (u) | c5 W2 f0 (u)
(unit 0 has no c2; the stream ends with W2 f1 (19): 120 chunks).  W1 chunks: a bias fragment (b_hi, b_lo in k-slots 0, 1
against ones: the projection bias comes out of the matrix pipe, exact to 2^-22) + 20 k-step fragments = 21 MFMAs; W2 chunks:
2 k-steps x 10 output tiles = 20 MFMAs.  The GEGLU of tile f0 (16 values per lane, 15 VALU instructions each) rides in the
MFMA gaps of c2..c4, that of f1 in c5, c0', c1' of the next unit: 124 MFMAs (3968 matrix-pipe cycles) against ~3200 cycles of
vector issue per unit - matrix-bound, unlike the attention program.

Register plan (named, clobbered): arch v[24:87] the four H tiles (hidden f0, gate f0, hidden f1, gate f1), v[88:103] g (two
tiles x 8), v[104:119] fragment ring, v[120:127] GELU temporaries (two elements in flight); a[0:159] Y^T, a[160:239] z^T,
a[240:243] the ones operand of the bias k-step.  The generator checks what gen_attn_asm.py checks.
"""
import os
import sys

NL = r"\n\t"
VB = 24
HT = {("h", 0): VB, ("g", 0): VB + 16, ("h", 1): VB + 32, ("g", 1): VB + 48}
G_ = {0: VB + 64, 1: VB + 72}
NRING = 8                 # fragment ring slots (4 registers each), read NRING MFMAs ahead
RING = VB + 80
TMP = RING + 4 * NRING      # 8 temporaries: two elements in flight x (t, e/q, z, m)
CA4 = TMP + 8               # the polynomial's constant term in a register (v_fma takes one scalar operand)
VEND = TMP + 10
YACC, ZF, ONESB, AEND = 0, 160, 240, 244
NKS = 20
W1_FR, W2_FR = NKS + 1, 20
W1_BYTES, W2_BYTES = W1_FR * 1024, W2_FR * 1024
SLOT = 24576
NSLOT = 6
WAITN = 4                 # fragment MFMAs per counted lgkmcnt wait
NUNIT = 20
# named SGPRs (clobbered inside the statement only; everything that lives across statements is an operand)
SC = {"c1": 60, "c2": 61, "a1": 62, "a2": 63, "a3": 64, "a4": 65, "a5": 66}      # GELU constants
SUNIT = 67
SP = 68                # s[68:69]: the weight stream pointer (carried across statements through operands)


def v(n):
    return "v%d" % n


def vr(a, n):
    return "v[%d:%d]" % (a, a + n - 1)


def ar(a, n):
    return "a[%d:%d]" % (a, a + n - 1)


def R(base, n, f="v"):
    return [(f, base + i) for i in range(n)]


class Ins:
    __slots__ = ("text", "kind", "rd", "wr", "meta")

    def __init__(self, text, kind, rd=(), wr=(), **meta):
        self.text, self.kind, self.rd, self.wr, self.meta = text, kind, tuple(rd), tuple(wr), meta


def slot_addr(slot):
    """(address operand, immediate) of byte 0 of an LDS ring slot: three base registers cover the 16-bit offset field"""
    return "%%[fa%d]" % (slot // 2), (slot % 2) * SLOT


class Gen:
    def __init__(self):
        self.ins = []

    def e(self, text, kind, rd=(), wr=(), **meta):
        self.ins.append(Ins(text, kind, rd, wr, **meta))

    def label(self, name):
        self.e(name + "_%=:", "label", name=name)

    def nop(self, n):
        self.e("s_nop %d" % n, "nop", n=n)

    # ---- GELU / GEGLU filler stream of one tile ---------------------------------------------------------------------------
    def geglu_items(self, f):
        """g[r] = hidden[r] * gelu(gate[r]) for the 16 accumulator registers of tile f, two elements in flight; then the
        8 conversions into the B operand.  Returns a list of (text, kind, rd, wr)."""
        hh, hg, g0 = HT[("h", f)], HT[("g", f)], G_[f]
        items = []

        def chain(r, t):
            x, hid = hg + r, hh + r
            T, E, Z, M = TMP + 4 * t, TMP + 4 * t + 1, TMP + 4 * t + 2, TMP + 4 * t + 3
            s = lambda k: "s%d" % SC[k]
            return [
                ("v_fma_f32 %s, |%s|, %s, 1.0" % (v(T), v(x), s("c1")), "valu", [("v", x)], [("v", T)]),
                ("v_mul_f32_e32 %s, %s, %s" % (v(Z), v(x), v(x)), "valu", [("v", x)], [("v", Z)]),
                ("v_rcp_f32_e32 %s, %s" % (v(T), v(T)), "trans", [("v", T)], [("v", T)]),
                ("v_mul_f32_e32 %s, %s, %s" % (v(Z), s("c2"), v(Z)), "valu", [("v", Z)], [("v", Z)]),
                ("v_max_f32_e32 %s, 0, %s" % (v(M), v(x)), "valu", [("v", x)], [("v", M)]),
                ("v_exp_f32_e64 %s, -%s" % (v(E), v(Z)), "trans", [("v", Z)], [("v", E)]),
                ("v_fma_f32 %s, %s, %s, %s" % (v(Z), v(T), s("a5"), v(CA4)), "valu", [("v", T), ("v", CA4)], [("v", Z)]),
                ("v_fma_f32 %s, %s, %s, %s" % (v(Z), v(Z), v(T), s("a3")), "valu", [("v", T), ("v", Z)], [("v", Z)]),
                ("v_fma_f32 %s, %s, %s, %s" % (v(Z), v(Z), v(T), s("a2")), "valu", [("v", T), ("v", Z)], [("v", Z)]),
                ("v_fma_f32 %s, %s, %s, %s" % (v(Z), v(Z), v(T), s("a1")), "valu", [("v", T), ("v", Z)], [("v", Z)]),
                ("v_mul_f32_e32 %s, %s, %s" % (v(Z), v(Z), v(T)), "valu", [("v", T), ("v", Z)], [("v", Z)]),
                ("v_mul_f32_e32 %s, %s, %s" % (v(Z), v(Z), v(E)), "valu", [("v", E), ("v", Z)], [("v", Z)]),
                ("v_fma_f32 %s, -%s, |%s|, %s" % (v(M), v(Z), v(x), v(M)), "valu", [("v", Z), ("v", x), ("v", M)], [("v", M)]),
                ("v_mul_f32_e32 %s, %s, %s" % (v(hid), v(hid), v(M)), "valu", [("v", hid), ("v", M)], [("v", hid)]),
            ]

        for r in range(0, 16, 2):
            a, b = chain(r, 0), chain(r + 1, 1)
            for x, y in zip(a, b):
                items += [x, y]
            items.append(("v_cvt_pk_f16_f32 %s, %s, %s" % (v(g0 + r // 2), v(hh + r), v(hh + r + 1)), "valu",
                          [("v", hh + r), ("v", hh + r + 1)], [("v", g0 + r // 2)]))
        return items

    # ---- fragment reads ------------------------------------------------------------------------------------------------
    def read_frag(self, slot, frag_i, ring_slot, tag):
        reg = RING + 4 * ring_slot
        base, imm = slot_addr(slot)
        self.e("ds_read_b128 %s, %s offset:%d" % (vr(reg, 4), base, imm + frag_i * 1024), "lds", wr=R(reg, 4), frag=tag)

    # ---- DMA of one chunk (this wave's pieces) ------------------------------------------------------------------------------
    def dma_items(self, slot, kind, adv):
        """instructions that copy a chunk ("w1": 21 KiB, "w2": 20 KiB) from the stream pointer s[SP:SP+1] into ring slot
        `slot` and move the pointer behind it (adv = bytes, or "wrap": back to the stream start).  Pieces w + 4j (j < 5)
        through the five offset registers; the 21st KiB of a W1 chunk (the bias fragment) by every wave (same bytes, same
        place).  An s_nop separates every M0 write from its LDS-DMA."""
        it = []          # pairs (M0 write, LDS-DMA): emitted around an MFMA, which is the wait state between the two
        for j in range(5):
            it.append(("s_add_u32 m0, %%[ldsw], %d" % (slot * SLOT + j * 4096),
                       "global_load_lds_dwordx4 %%[vo%d], s[%d:%d]" % (j, SP, SP + 1)))
        if kind == "w1":
            it.append(("s_add_u32 m0, %%[lds0], %d" % (slot * SLOT + 20480),
                       "global_load_lds_dwordx4 %%[vob], s[%d:%d]" % (SP, SP + 1)))
        if adv == "wrap":
            it.append(("s_mov_b32 s%d, %%[sp0lo]" % SP, "s_mov_b32 s%d, %%[sp0hi]" % (SP + 1)))
        else:
            it.append(("s_add_u32 s%d, s%d, %d" % (SP, SP, adv), "s_addc_u32 s%d, s%d, 0" % (SP + 1, SP + 1)))
        return it

    def dma_first(self):
        """first half of the next pending DMA pair (an M0 write or the pointer's low word): goes in FRONT of an MFMA"""
        if self.pending_dma:
            self.e(self.pending_dma[0][0], "salu")
            self.dma_half = True

    def dma_second(self):
        """second half (the LDS-DMA itself / the pointer's high word): behind that MFMA"""
        if getattr(self, "dma_half", False):
            t = self.pending_dma.pop(0)[1]
            self.e(t, "vmem" if t.startswith("global_load") else "salu")
            self.dma_half = False

    def emit_dma_all(self):
        while self.pending_dma:
            self.dma_first()
            self.nop(0)
            self.dma_second()

    # ---- one chunk ----------------------------------------------------------------------------------------------------------
    def chunk(self, cname, ctype, f, slot, fillers, nxt, dma, wait_n, first_y=False, fill_from=0, flush=False):
        """ctype "h" / "g": W1 tile f (bias fragment + 20 k-steps -> H tile); "w2": W2 half f (2 k-steps x 10 tiles -> Y^T).
        nxt = (slot, cname) of the FOLLOWING chunk or None: its barrier sits four MFMAs before this chunk's end and its first
        four fragments are read behind this chunk's last four MFMAs.  dma = (slot, kind, advance) of the chunk issued behind
        that barrier (three chunks ahead of the one it opens).  wait_n: own DMA operations that may stay in flight at that
        barrier.  fillers: {"q": VALU items, "gaps": gaps left in their window}, consumed from gap fill_from of this chunk on;
        flush: the window ends with this chunk."""
        self.e("; ---- chunk %s" % cname, "comment")
        nfr = W1_FR if ctype in ("h", "g") else W2_FR
        tags = [(cname, i) for i in range(nfr)]
        for i in range(nfr):
            if i == nfr - NRING and nxt is not None:
                # ---- the next chunk becomes visible: own pieces landed, everybody's published
                assert not self.pending_dma, "the previous chunk's DMA is still being issued"
                self.e("s_waitcnt vmcnt(%d)" % wait_n, "waitvm")
                self.e("s_barrier", "barrier")
                if dma is not None:
                    self.pending_dma = self.dma_items(*dma)
            self.dma_first()
            if i % WAITN == 0:            # one counted wait per WAITN fragment MFMAs
                self.e("WAITFRAG", "waitfrag", frag=tags[min(i + WAITN - 1, nfr - 1)])
            rs = self.ringpos % NRING     # ring slots rotate over ALL fragment MFMAs (a W1 chunk has 21)
            self.ringpos += 1
            reg = RING + 4 * rs
            if ctype in ("h", "g"):
                d = HT[(ctype, f)]
                if i == 0:          # bias fragment x ones: H = b (C = 0)
                    self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, 0" % (vr(d, 16), vr(reg, 4), ar(ONESB, 4)), "mfma",
                           rd=R(reg, 4) + R(ONESB, 4, "a"), wr=R(d, 16), frag=tags[i], acc=False)
                else:
                    ks = i - 1
                    self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), vr(reg, 4), ar(ZF + 4 * ks, 4), vr(d, 16)), "mfma",
                           rd=R(reg, 4) + R(ZF + 4 * ks, 4, "a") + R(d, 16), wr=R(d, 16), frag=tags[i], acc=True)
            else:
                ss, ti = divmod(i, 10)
                d = YACC + 16 * ti
                b = G_[f] + 4 * ss
                c = "0" if (first_y and ss == 0) else ar(d, 16)
                self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (ar(d, 16), vr(reg, 4), vr(b, 4), c), "mfma",
                       rd=R(reg, 4) + R(b, 4) + ([] if c == "0" else R(d, 16, "a")), wr=R(d, 16, "a"), frag=tags[i],
                       acc=c != "0")
            # fragment NRING ahead (this chunk's, or the next chunk's first ones)
            j = i + NRING
            if j < nfr:
                self.read_frag(slot, j, rs, tags[j])
            elif nxt is not None:
                self.read_frag(nxt[0], j - nfr, rs, (nxt[1], j - nfr))
            self.dma_second()
            # VALU fillers: an equal share of what is left for their window
            if fillers is not None and i >= fill_from:
                q, left = fillers["q"], fillers["gaps"]
                n = len(q) if (left <= 1 or (flush and i == nfr - 1)) else -(-len(q) // left)
                for _ in range(min(n, len(q))):
                    t, k, rd, wr = q.pop(0)
                    self.e(t, k, rd=rd, wr=wr, cost=8 if k == "trans" else 4)
                fillers["gaps"] = left - 1
        if flush and fillers is not None:
            assert not fillers["q"]

    # ---- the statement of one panel ----------------------------------------------------------------------------------------
    def build(self):
        import struct
        e = self.e
        self.pending_dma = []
        fbits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
        consts = {"c1": 0.3275911 * 0.70710678118654752440, "c2": 0.5 * 1.4426950408889634, "a1": 0.5 * 0.254829592,
                  "a2": 0.5 * -0.284496736, "a3": 0.5 * 1.421413741, "a4": 0.5 * -1.453152027, "a5": 0.5 * 1.061405429}
        for k, val in consts.items():
            e("s_mov_b32 s%d, 0x%08x" % (SC[k], fbits(val)), "salu")
        e("s_mov_b32 s%d, %%[splo]" % SP, "salu")
        e("s_mov_b32 s%d, %%[sphi]" % (SP + 1), "salu")
        e("v_mov_b32_e32 %s, s%d" % (v(CA4), SC["a4"]), "valu", wr=[("v", CA4)])
        # ones operand of the bias k-step: k-slots 0, 1 of the h = 0 lanes
        e("v_and_b32_e32 %s, 0x3c003c00, %%[hmask]" % v(TMP), "valu", wr=[("v", TMP)])
        e("v_accvgpr_write_b32 a%d, %s" % (ONESB, v(TMP)), "valu", rd=[("v", TMP)], wr=[("a", ONESB)])
        for i in range(1, 4):
            e("v_accvgpr_write_b32 a%d, 0" % (ONESB + i), "valu", wr=[("a", ONESB + i)])
        # chunks 0..2 of this panel were issued a panel ago (or by the kernel before the first panel); everything landed
        e("s_waitcnt vmcnt(0)", "waitvm")
        e("s_barrier", "barrier")
        W1P, W2P = 6, 5
        isw1 = lambda c: c[1] in ("h", "g")
        pieces = lambda c: W1P if isw1(c) else W2P
        size = lambda c: W1_BYTES if isw1(c) else W2_BYTES
        LOOPC = [("c0", "h", 0), ("c1", "g", 0), ("c2", "w2", 1), ("c3", "h", 1), ("c4", "g", 1), ("c5", "w2", 0)]
        FIRSTC = [c for c in LOOPC if c[0] != "c2"]
        # how GEGLU f1 splits between c5 and the next unit's c0, c1 (the same in every unit: simulated once)
        total_f1 = len(self.geglu_items(1))
        win_f1 = (W2_FR - 3) + 2 * W1_FR
        left, gaps, used_c5 = total_f1, win_f1, 0
        for _ in range(W2_FR - 3):
            n = -(-left // gaps)
            left, gaps, used_c5 = left - n, gaps - 1, used_c5 + n
        self.f1_in_c5 = used_c5

        # ---- statement start: the DMA of chunk 3 (unit 0's c4) and the first fragments of chunk 0
        self.pending_dma = self.dma_items(3, "w1", W1_BYTES)
        self.emit_dma_all()
        self.ringpos = 0
        for j in range(NRING):
            self.read_frag(0, j, j, ("c0", j))

        def unit(u_kind):
            first, last = u_kind == "first", u_kind == "last"
            seq = list(FIRSTC if first else LOOPC) + ([("tail", "w2", 1)] if last else [])
            follow = list(FIRSTC) if last else list(LOOPC)
            base_n = 0 if first else 5
            allc = seq + follow
            allslots = [(base_n + i) % NSLOT for i in range(len(allc))]
            if first:
                fill_prev = None
                fill_f0 = {"q": self.geglu_items(0), "gaps": (W1_FR - 3) + W1_FR}                  # c3 (from gap 3), c4
            else:
                fill_prev = {"q": self.geglu_items(1)[self.f1_in_c5:], "gaps": 2 * W1_FR}        # rest of GEGLU f1 (u-1): c0, c1
                fill_f0 = {"q": self.geglu_items(0), "gaps": (W2_FR - 3) + 2 * W1_FR}             # c2 (from gap 3), c3, c4
            fill_f1 = {"q": self.geglu_items(1), "gaps": win_f1}
            for i, c in enumerate(seq):
                cname, ctype, f = c
                if cname == "tail":
                    self.chunk(cname, ctype, f, allslots[i], None, None, None, 0)
                    continue
                nxt = (allslots[i + 1], allc[i + 1][0])
                pre = allc[i + 4]
                wraps = last and i + 4 == len(seq) - 1              # `pre` is the stream's last chunk (the tail)
                dma = (allslots[i + 4], "w1" if isw1(pre) else "w2", "wrap" if wraps else size(pre))
                wait_n = pieces(allc[i + 2]) + pieces(allc[i + 3])
                fl, ff, fsh = None, 0, False
                if cname in ("c0", "c1") and not first:
                    fl, fsh = fill_prev, cname == "c1"
                elif cname in ("c2", "c3", "c4"):
                    fl, fsh = fill_f0, cname == "c4"
                    ff = 3 if cname == ("c3" if first else "c2") else 0
                elif cname == "c5":
                    fl, ff = fill_f1, 3
                self.chunk(cname, ctype, f, allslots[i], fl, nxt, dma, wait_n, first_y=(first and cname == "c5"), fill_from=ff,
                           flush=fsh)
                if cname == "c5":
                    assert len(self.geglu_items(1)) - len(fill_f1["q"]) == self.f1_in_c5, "GEGLU f1 split differs"
                    if last:            # the rest of GEGLU f1 (19) has no MFMAs left to hide behind
                        for t, k, rd, wr in fill_f1["q"]:
                            self.e(t, k, rd=rd, wr=wr)
                        self.nop(2)

        unit("first")
        # (a unit has 124 fragment MFMAs: with an eight-slot ring the slot rotation repeats every TWO units)
        assert (NUNIT - 2) % 2 == 0
        e("s_mov_b32 s%d, %d" % (SUNIT, (NUNIT - 2) // 2), "salu")
        self.label("LOOP")
        unit("loop")
        unit("loop")
        e("s_sub_u32 s%d, s%d, 1" % (SUNIT, SUNIT), "salu")
        e("s_cmp_lg_u32 s%d, 0" % SUNIT, "salu")
        e("s_cbranch_scc1 LOOP_%=", "branch", target="LOOP")
        unit("last")
        self.emit_dma_all()
        self.nop(15)
        self.nop(15)
        e("s_mov_b32 %%[splo], s%d" % SP, "salu")
        e("s_mov_b32 %%[sphi], s%d" % (SP + 1), "salu")

    # ---- counted lgkmcnt waits ----------------------------------------------------------------------------------------------
    def resolve_waits(self):
        out, fifo = [], []
        loop_fifo = None
        for i in self.ins:
            if i.kind == "lds":
                fifo.append(i.meta["frag"])
                out.append(i)
            elif i.kind == "waitfrag":
                fr = i.meta["frag"]
                idx = [k for k, f in enumerate(fifo) if f == fr]
                assert idx, ("fragment never read", fr)
                keep = len(fifo) - 1 - idx[-1]
                out.append(Ins("s_waitcnt lgkmcnt(%d)" % keep, "waitlgkm", n=keep))
                fifo = fifo[idx[-1] + 1:]
            elif i.kind == "label" and i.meta["name"] == "LOOP":
                loop_fifo = list(fifo)
                out.append(i)
            elif i.kind == "branch" and i.meta["target"] == "LOOP":
                assert fifo == loop_fifo, (fifo, loop_fifo)
                out.append(i)
            else:
                out.append(i)
        self.ins = out

    # ---- checks (see gen_attn_asm.py) -----------------------------------------------------------------------------------------
    def check(self):
        def ws(i):
            return i.meta["n"] + 1 if i.kind == "nop" else (0 if i.kind in ("label", "comment") else (8 if i.kind == "mfma" else 1))

        seq = self.ins
        a = next(k for k, i in enumerate(seq) if i.kind == "label" and i.meta["name"] == "LOOP")
        b = next(k for k, i in enumerate(seq) if i.kind == "branch" and i.meta["target"] == "LOOP")
        walk = seq[:b] + seq[a:b] + seq[b:]
        last_mfma_wr, last_valu_wr, last_trans_wr = {}, {}, {}
        pos = nerr = 0
        for i in walk:
            if i.kind in ("label", "comment"):
                continue
            for r in i.rd + i.wr:
                if r in last_mfma_wr:
                    same_chain = i.kind == "mfma" and i.meta.get("acc") and r in i.wr
                    if not same_chain and pos - last_mfma_wr[r] < 20:
                        print("HAZARD mfma->use %s dist %d: %s" % (r, pos - last_mfma_wr[r], i.text))
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
        written = set(("a", i) for i in range(ZF, ZF + 80))
        for i in walk:
            if i.kind in ("label", "comment"):
                continue
            for r in i.rd:
                if r[0] in ("v", "a") and r not in written and not (r[0] == "a" and r[1] < 160):
                    print("UNINITIALISED %s read by: %s" % (r, i.text))
                    nerr += 1
                    written.add(r)
            written.update(i.wr)
        slotfrag, pending = {}, []
        for i in walk:
            if i.kind == "lds":
                for r in i.wr:
                    slotfrag[r] = i.meta["frag"]
                pending.append(i.meta["frag"])
            elif i.kind == "waitlgkm":
                n = i.meta["n"]
                pending = pending[len(pending) - n:] if n else []
            elif i.kind == "mfma":
                regs = [r for r in i.rd if r[0] == "v" and RING <= r[1] < RING + 4 * NRING]
                assert len(regs) == 4
                want = i.meta["frag"]
                for r in regs:
                    got = slotfrag.get(r)
                    if got != want:
                        print("RING slot %s holds %s, MFMA expects %s" % (r, got, want))
                        nerr += 1
                if want in pending:
                    print("RING fragment not waited for: %s" % (want,))
                    nerr += 1
        assert nerr == 0, "%d problems" % nerr

    def text(self):
        knob = os.environ.get("FF_GEN_KNOB", "").split("+")      # timing experiments only (tools/micro/ff_knobs.sh): results WRONG
        keep = []
        for i in self.ins:
            if i.kind == "comment":
                continue
            if "nolds" in knob and i.kind in ("lds", "waitlgkm"):
                continue
            if "novalu" in knob and i.kind in ("valu", "trans") and "cost" in i.meta:
                continue
            if "nomfma" in knob and i.kind == "mfma":
                continue
            if "nodma" in knob and i.kind == "vmem":
                continue
            if "nobar" in knob and i.kind in ("barrier", "waitvm", "vmem"):
                continue
            if "dmapad" in knob and i.kind == "waitvm":          # (twice the operations in flight: the counted waits double)
                n = int(i.text.split("(")[1].rstrip(")"))
                i = Ins("s_waitcnt vmcnt(%d)" % (2 * n), "waitvm")
            keep.append(i)
            # "dmapad" (round 6): every LDS-DMA issued twice (same bytes to the same place) - the L2 -> LDS fill, its issue slots and
            # the LDS write traffic of a form whose chunks serve two waves instead of four (the C = 640 wave-pair form)
            if "dmapad" in knob and i.kind == "vmem":
                keep.append(i)
            # sensitivity knobs that leave the RESULT (and so the data the MFMAs see, and the clock) unchanged: every GEGLU
            # instruction followed by a dead move / every fragment read issued twice
            if "valupad" in knob and i.kind in ("valu", "trans") and "cost" in i.meta:
                keep.append(Ins("v_mov_b32_e32 v%d, v%d" % (VEND - 1, VEND - 1), "valu"))
            if "ldspad" in knob and i.kind == "lds":
                keep.append(i)
        return " \\\n  ".join('"' + i.text + NL + '"' for i in keep)

    def stats(self):
        cost = {"mfma": 8, "trans": 8, "valu": 4, "salu": 4, "lds": 4, "vmem": 4, "waitlgkm": 4, "waitvm": 4, "barrier": 4}
        cur, tot, n = None, {}, {}
        for i in self.ins:
            if i.kind == "comment":
                cur = i.text
                tot[cur], n[cur] = 0, {}
            elif cur is not None:
                tot[cur] += cost.get(i.kind, 0)
                n[cur][i.kind] = n[cur].get(i.kind, 0) + 1
        for k in tot:
            print("%-28s issue cycles %5d (matrix pipe %4d)  %s" % (k, tot[k], 32 * n[k].get("mfma", 0), n[k]))
        print("instructions:", sum(1 for i in self.ins if i.kind not in ("comment", "label")))


def main():
    g = Gen()
    g.build()
    g.resolve_waits()
    g.check()
    if "--stats" in sys.argv:
        g.stats()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lkgd_amd", "csrc", "ff_fused_loop.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_ff_asm.py - do not edit.  Panel loop of ff_fused.hip (plan: see that script).\n")
        f.write("#define FF_VB %d\n#define FF_VEND %d\n#define FF_AEND %d\n#define FF_YACC %d\n#define FF_ZF %d\n" % (VB, VEND, AEND, YACC, ZF))
        f.write("#define FF_W1_BYTES %d\n#define FF_W2_BYTES %d\n#define FF_SLOT %d\n#define FF_NSLOT %d\n" % (W1_BYTES, W2_BYTES, SLOT, NSLOT))
        f.write("#define FF_PANEL_ASM \\\n  %s\n\n" % g.text())
        clob = ['"v%d"' % i for i in range(VB, VEND)] + ['"a%d"' % i for i in range(AEND)] + ['"s%d"' % i for i in range(60, 70)]
        f.write("#define FF_CLOBBERS " + ", ".join(clob) + ', "vcc", "scc", "m0", "memory"\n')
    print("wrote", out)


if __name__ == "__main__":
    main()
