#!/usr/bin/env python3
r"""Generates lkgd_amd/csrc/attn_tblock_loop.inc: the panel statement of attn_tblock.hip - the attention half of a temporal
transformer block at the 72x128 level (C = 320, 5 heads of 64, F <= 16 frames) in ONE kernel:
    out = to_out(attention_over_frames(to_q | to_k | to_v (LayerNorm(x)))) + x       (patch/patch.py:610, :660-661)
Run from the repo root:  python tools/gen_tblock_asm.py   (--stats for the per-chunk issue-cost table)

Same skeleton as the fused feed-forward (tools/gen_ff_asm.py): one wave = 32 token rows = TWO pixels x 16 frame slots, whose
LayerNorm-ed rows sit in a[160:239] as MFMA operands for the whole panel; one workgroup = 4 waves (one per SIMD) = 8 pixels; the
weights stream L2 -> LDS in 20/21-KiB CHUNKS through a ring of five slots, three chunks ahead (LDS-DMA, counted vmcnt), one
barrier per chunk, fragment reads eight MFMAs ahead across chunk borders.  Per head h (stream order; packing.pack_tblock):
    q0 q1 , k0 k1   "swapped" products  Q^T, K^T [64 d][32 tokens] = W . z^T   (a lane owns a token, registers run over d);
                    two 32-row tiles each, bias fragment (b_hi, b_lo against ones) + 20 k-steps = 21 MFMAs per chunk;
                    the softmax scale log2(e)/8 multiplies the fp32 scores, not the weights
    v0 v1           "direct" product V [32 tokens][64 d] = z . Wv^T: the same token registers as the A operand, the weight
                    fragment as B (a lane owns a head channel, registers run over the tokens)
    o0 o1 (h-1)     Y^T [320][32 tokens] += Wo[:, head h-1] . O^T: 2 k-steps x 10 output tiles = 20 MFMAs per chunk,
                    between the q and the k chunks of head h
(head 0 has no o chunks; the stream ends with o0 o1 of heads 3 and 4: 40 chunks per panel - see stream()).  The attention itself rides in the gaps:
    S^T [key][query] = K . Q^T + MASK   4 MFMAs on the PACKED projection accumulators (they are already operands: a lane of K
                    is a key row, a lane of Q a query column, the d order is the same permutation on both sides); MASK (the C
                    operand: -30000 for the other pixel's keys and for frame slots >= F) makes the 32x32 product block-diagonal
    softmax         a lane holds half of a query's keys: max and sum meet the other half through ds_bpermute (lane ^ 32)
    O^T [d][query]  = V^T . P            4 MFMAs: A = packed V (lane = d, k-slots = tokens), B = packed P (same key order)
and the packed O^T is the B operand of the out-projection.  Nothing of q, k, v, the scores or the head outputs reaches memory.

Register plan (named, clobbered): v[24:55] Q tiles (packed in place into v[24:39]; S^T in v[40:55]), v[56:87] K tiles (packed
into v[56:71]; later the O^T accumulators), v[88:119] V tiles (packed in place into v[88:103]), v[120:135] packed O^T,
v[136:143] P, v[144:159] MASK, v[160:191] fragment ring, v[192:199] temporaries; a[0:159] Y^T, a[160:239] z^T, a[240:243] ones.
The generator checks MFMA -> use distances, VALU -> MFMA and trans -> VALU wait states, ring contents, counted lgkmcnt waits,
reads of uninitialised registers, and that no chunk overwrites a register a queued instruction still has to read.
"""
import os
import sys

NL = r"\n\t"
VB = 24
QA, KA, VA = VB, VB + 32, VB + 64
SACC = QA + 16            # S^T accumulators: the upper half of the Q tiles (free once q is packed into the lower half)
OACC = KA                 # O^T accumulators: the K tiles (dead once S^T is computed)
OPK = VB + 96             # packed O^T (16): the B operand of the out-projection chunks
PREG = VB + 112           # packed probabilities (8)
MASK = VB + 120           # 16
NRING = 8
RING = VB + 136
TMP = RING + 4 * NRING    # M, M2, T, L, L2, INV
VEND = TMP + 8
YACC, ZF, ONESB, AEND = 0, 160, 240, 244
NKS = 20
W1_FR, W2_FR = NKS + 1, 20
W1_BYTES, W2_BYTES = W1_FR * 1024, W2_FR * 1024
SLOT = 24576
NSLOT = 5
WAITN = 4
HEADS = 5
RATE = 3                  # queued VALU instructions per MFMA gap
SP = 68                   # s[68:69]: the weight stream pointer
SCL = 70                  # log2(e) / 8: the softmax scale, applied to the fp32 scores (weights and biases stay as packed)
NEG = "0xc6ea6000"        # -30000.0f


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
    return "%%[fa%d]" % (slot // 2), (slot % 2) * SLOT


def stream():
    """the 40 chunks of a panel: (name, type, head, tile).  The out-projection of head h - 1 sits between the q and the k
    chunks of head h (its operand, the packed O^T of head h - 1, is finished under q0 q1 of head h); the last head keeps the
    previous head's out-projection behind its v chunks, as cover for its own softmax."""
    def C(t, h):
        return [("%s%d.%d" % (t, f, h), t, h, f) for f in (0, 1)]
    cs = []
    for h in range(HEADS):
        cs += C("q", h)
        if 0 < h < HEADS - 1:
            cs += C("o", h - 1)
        cs += C("k", h) + C("v", h)
    cs += C("o", HEADS - 2) + C("o", HEADS - 1)
    return cs


class Gen:
    def __init__(self):
        self.ins = []
        self.queue = []          # instructions waiting for MFMA gaps: Ins, ("GATE", ring-MFMA position) or ("GAP",)
        self.mpos = 0            # ring MFMAs emitted so far
        self.ringpos = 0
        self.pending_dma = []
        self.dma_half = False
        self.xid = 0
        self.chunk_end = {}      # chunk name -> ring-MFMA count at its end

    def e(self, text, kind, rd=(), wr=(), **meta):
        self.ins.append(Ins(text, kind, rd, wr, **meta))

    def nop(self, n):
        self.e("s_nop %d" % n, "nop", n=n)

    def q(self, text, kind, rd=(), wr=(), **meta):
        self.queue.append(Ins(text, kind, rd, wr, **meta))

    # ---- the queued work of a head ----------------------------------------------------------------------------------------
    def q_pack(self, src, dst, head, what):
        """two 16-register accumulator tiles at src -> 16 packed registers at dst (dst may be src: in-place, ascending)"""
        for t in (0, 1):
            for r in range(0, 16, 2):
                a, b, d = src + 16 * t + r, src + 16 * t + r + 1, dst + 8 * t + r // 2
                self.q("v_cvt_pk_f16_f32 %s, %s, %s" % (v(d), v(a), v(b)), "valu", rd=[("v", a), ("v", b)], wr=[("v", d)],
                       head=head, what=what)

    def q_scores(self, h):
        """S^T = K . Q^T + MASK: four k-steps on the packed accumulators, one MFMA per gap"""
        for s in range(4):
            c = vr(MASK, 16) if s == 0 else vr(SACC, 16)
            self.q("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(SACC, 16), vr(KA + 4 * s, 4), vr(QA + 4 * s, 4), c), "mfma",
                   rd=R(KA + 4 * s, 4) + R(QA + 4 * s, 4) + (R(MASK, 16) if s == 0 else R(SACC, 16)), wr=R(SACC, 16),
                   acc=s > 0, inline=True, head=h, what="S")
            self.queue.append(("GAP",))

    def q_softmax(self, h):
        M, M2, T, L, L2, INV = (TMP + i for i in range(6))
        S = SACC
        q = self.q

        def m3(d, a, b, c):
            q("v_max3_f32 %s, %s, %s, %s" % (v(d), v(a), v(b), v(c)), "valu", rd=[("v", x) for x in (a, b, c)], wr=[("v", d)], head=h)

        ca = [(M, S, S + 1, S + 2), (M, M, S + 3, S + 4), (M, M, S + 5, S + 6), (M, M, S + 7, S + 15)]
        cb = [(M2, S + 8, S + 9, S + 10), (M2, M2, S + 11, S + 12), (M2, M2, S + 13, S + 14)]
        for i in range(4):
            m3(*ca[i])
            if i < 3:
                m3(*cb[i])
        q("v_max_f32_e32 %s, %s, %s" % (v(M), v(M), v(M2)), "valu", rd=[("v", M), ("v", M2)], wr=[("v", M)], head=h)

        def other_half(dst, src):
            self.xid += 1
            tag = ("X", self.xid)
            q("ds_bpermute_b32 %s, %%[xora], %s" % (v(dst), v(src)), "lds", rd=[("v", src)], wr=[("v", dst)], frag=tag, head=h)
            q("WAITFRAG", "waitfrag", frag=tag, head=h)

        other_half(T, M)
        q("v_max_f32_e32 %s, %s, %s" % (v(M), v(M), v(T)), "valu", rd=[("v", M), ("v", T)], wr=[("v", M)], head=h)
        q("v_mul_f32_e32 %s, s%d, %s" % (v(M), SCL, v(M)), "valu", rd=[("v", M)], wr=[("v", M)], head=h)
        for r in range(16):      # c s - c m: exponent of 2 (masked slots: c (-30000 - m) -> 2^x = 0)
            q("v_fma_f32 %s, %s, s%d, -%s" % (v(S + r), v(S + r), SCL, v(M)), "valu", rd=[("v", S + r), ("v", M)], wr=[("v", S + r)], head=h)
        for r in range(16):
            q("v_exp_f32_e32 %s, %s" % (v(S + r), v(S + r)), "trans", rd=[("v", S + r)], wr=[("v", S + r)], head=h)
        # row sum: two chains, joined
        q("v_add_f32_e32 %s, %s, %s" % (v(L), v(S), v(S + 1)), "valu", rd=[("v", S), ("v", S + 1)], wr=[("v", L)], head=h)
        q("v_add_f32_e32 %s, %s, %s" % (v(L2), v(S + 8), v(S + 9)), "valu", rd=[("v", S + 8), ("v", S + 9)], wr=[("v", L2)], head=h)
        for r in range(2, 8):
            q("v_add_f32_e32 %s, %s, %s" % (v(L), v(L), v(S + r)), "valu", rd=[("v", L), ("v", S + r)], wr=[("v", L)], head=h)
            q("v_add_f32_e32 %s, %s, %s" % (v(L2), v(L2), v(S + 8 + r)), "valu", rd=[("v", L2), ("v", S + 8 + r)], wr=[("v", L2)], head=h)
        q("v_add_f32_e32 %s, %s, %s" % (v(L), v(L), v(L2)), "valu", rd=[("v", L), ("v", L2)], wr=[("v", L)], head=h)
        other_half(T, L)
        q("v_add_f32_e32 %s, %s, %s" % (v(L), v(L), v(T)), "valu", rd=[("v", L), ("v", T)], wr=[("v", L)], head=h)
        q("v_rcp_f32_e32 %s, %s" % (v(INV), v(L)), "trans", rd=[("v", L)], wr=[("v", INV)], head=h)
        q("v_nop", "valu", head=h)
        for r in range(16):
            q("v_mul_f32_e32 %s, %s, %s" % (v(S + r), v(S + r), v(INV)), "valu", rd=[("v", S + r), ("v", INV)], wr=[("v", S + r)], head=h)
        for r in range(0, 16, 2):
            q("v_cvt_pk_f16_f32 %s, %s, %s" % (v(PREG + r // 2), v(S + r), v(S + r + 1)), "valu", rd=[("v", S + r), ("v", S + r + 1)],
              wr=[("v", PREG + r // 2)], head=h)

    def q_heads_out(self, h):
        """O^T = V^T . P: two d tiles x two k-steps (k-step s = the tokens of pixel s) into the V accumulators"""
        for t in (0, 1):
            for s in (0, 1):
                d = OACC + 16 * t
                c = "0" if s == 0 else vr(d, 16)
                self.q("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), vr(VA + 8 * t + 4 * s, 4), vr(PREG + 4 * s, 4), c), "mfma",
                       rd=R(VA + 8 * t + 4 * s, 4) + R(PREG + 4 * s, 4) + ([] if s == 0 else R(d, 16)), wr=R(d, 16), acc=s > 0,
                       inline=True, head=h, what="O")
                self.queue.append(("GAP",))

    def dispense(self):
        """queued instructions for one MFMA gap"""
        n = 0
        while self.queue and n < RATE:
            it = self.queue[0]
            if isinstance(it, tuple):
                if it[0] == "GAP":
                    self.queue.pop(0)
                    break
                if it[0] == "GATE3":          # three ring MFMAs behind the inline MFMA that was dispensed last
                    self.queue[0] = it = ("GATE", self.mpos + 3)
                if it[0] == "AFTER":          # three ring MFMAs behind the last MFMA of a chunk (its operands are free then)
                    if it[1] not in self.chunk_end:
                        break
                    self.queue[0] = it = ("GATE", self.chunk_end[it[1]] + 3)
                if self.mpos < it[1]:
                    break
                self.queue.pop(0)
                continue
            if it.kind == "mfma" and n:
                break
            self.queue.pop(0)
            self.ins.append(it)
            n += 1

    def gate(self, after):
        self.queue.append(("GATE", self.mpos + after))

    # ---- fragment reads / DMA (as gen_ff_asm.py) ------------------------------------------------------------------------
    def read_frag(self, slot, frag_i, ring_slot, tag):
        reg = RING + 4 * ring_slot
        base, imm = slot_addr(slot)
        self.e("ds_read_b128 %s, %s offset:%d" % (vr(reg, 4), base, imm + frag_i * 1024), "lds", wr=R(reg, 4), frag=tag)

    def dma_items(self, slot, w1, wrap):
        it = []
        for j in range(5):
            it.append(("s_add_u32 m0, %%[ldsw], %d" % (slot * SLOT + j * 4096),
                       "global_load_lds_dwordx4 %%[vo%d], s[%d:%d]" % (j, SP, SP + 1)))
        if w1:
            it.append(("s_add_u32 m0, %%[lds0], %d" % (slot * SLOT + 20480),
                       "global_load_lds_dwordx4 %%[vob], s[%d:%d]" % (SP, SP + 1)))
        if wrap:
            it.append(("s_mov_b32 s%d, %%[sp0lo]" % SP, "s_mov_b32 s%d, %%[sp0hi]" % (SP + 1)))
        else:
            it.append(("s_add_u32 s%d, s%d, %d" % (SP, SP, W1_BYTES if w1 else W2_BYTES), "s_addc_u32 s%d, s%d, 0" % (SP + 1, SP + 1)))
        return it

    def dma_first(self):
        if self.pending_dma:
            self.e(self.pending_dma[0][0], "salu")
            self.dma_half = True

    def dma_second(self):
        if self.dma_half:
            t = self.pending_dma.pop(0)[1]
            self.e(t, "vmem" if t.startswith("global_load") else "salu")
            self.dma_half = False

    def emit_dma_all(self):
        while self.pending_dma:
            self.dma_first()
            self.nop(0)
            self.dma_second()

    # ---- one chunk ------------------------------------------------------------------------------------------------------
    def chunk(self, n, cs):
        name, ctype, h, f = cs[n]
        self.e("; ---- chunk %s" % name, "comment")
        w1 = ctype != "o"
        nfr = W1_FR if w1 else W2_FR
        slot = n % NSLOT
        nxt = cs[n + 1] if n + 1 < len(cs) else None
        tags = [(name, i) for i in range(nfr)]
        # what this chunk writes / reads must not be pending in the queue
        acc = {"q": QA, "k": KA, "v": VA}.get(ctype)
        wrs = set(R(acc + 16 * f, 16)) if w1 else set()
        rds = set() if w1 else set(R(OPK + 8 * f, 8))
        names = [c[0] for c in cs]
        for it in self.queue:
            if isinstance(it, tuple) and it[0] == "AFTER" and names.index(it[1]) >= n:
                break                 # (what follows waits for a chunk that is not behind us: it cannot meet this one)
            if isinstance(it, Ins):
                assert not (wrs & set(it.rd)) and not (wrs & set(it.wr)), ("chunk %s overwrites registers of a queued instruction" % name, it.text)
                assert not (rds & set(it.wr)), ("chunk %s reads what a queued instruction has yet to write" % name, it.text)
        isw1 = lambda c: c[1] != "o"
        pieces = lambda c: 6 if isw1(c) else 5
        for i in range(nfr):
            if i == nfr - NRING and nxt is not None:
                assert not self.pending_dma, "the previous chunk's DMA is still being issued"
                wait_n = pieces(cs[(n + 2) % len(cs)]) + pieces(cs[(n + 3) % len(cs)])
                self.e("s_waitcnt vmcnt(%d)" % wait_n, "waitvm")
                self.e("s_barrier", "barrier")
                pre = (n + 4) % len(cs)
                self.pending_dma = self.dma_items(pre % NSLOT, isw1(cs[pre]), wrap=(pre == len(cs) - 1))
            self.dma_first()
            if i % WAITN == 0:
                self.e("WAITFRAG", "waitfrag", frag=tags[min(i + WAITN - 1, nfr - 1)])
            rs = self.ringpos % NRING
            self.ringpos += 1
            reg = RING + 4 * rs
            if ctype in ("q", "k"):
                d = acc + 16 * f
                if i == 0:
                    self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, 0" % (vr(d, 16), vr(reg, 4), ar(ONESB, 4)), "mfma",
                           rd=R(reg, 4) + R(ONESB, 4, "a"), wr=R(d, 16), frag=tags[i], acc=False)
                else:
                    ks = i - 1
                    self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), vr(reg, 4), ar(ZF + 4 * ks, 4), vr(d, 16)), "mfma",
                           rd=R(reg, 4) + R(ZF + 4 * ks, 4, "a") + R(d, 16), wr=R(d, 16), frag=tags[i], acc=True)
            elif ctype == "v":         # direct: the token rows are the A operand, the weight fragment the B operand
                d = acc + 16 * f
                if i == 0:
                    self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, 0" % (vr(d, 16), ar(ONESB, 4), vr(reg, 4)), "mfma",
                           rd=R(reg, 4) + R(ONESB, 4, "a"), wr=R(d, 16), frag=tags[i], acc=False)
                else:
                    ks = i - 1
                    self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), ar(ZF + 4 * ks, 4), vr(reg, 4), vr(d, 16)), "mfma",
                           rd=R(reg, 4) + R(ZF + 4 * ks, 4, "a") + R(d, 16), wr=R(d, 16), frag=tags[i], acc=True)
            else:
                ss, ti = divmod(i, 10)
                d = YACC + 16 * ti
                b = OPK + 8 * f + 4 * ss
                first = h == 0 and f == 0 and ss == 0
                c = "0" if first else ar(d, 16)
                self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (ar(d, 16), vr(reg, 4), vr(b, 4), c), "mfma",
                       rd=R(reg, 4) + R(b, 4) + ([] if first else R(d, 16, "a")), wr=R(d, 16, "a"), frag=tags[i], acc=not first)
            self.mpos += 1
            j = i + NRING
            if j < nfr:
                self.read_frag(slot, j, rs, tags[j])
            elif nxt is not None:
                self.read_frag((n + 1) % NSLOT, j - nfr, rs, (nxt[0], j - nfr))
            self.dma_second()
            self.dispense()
        self.chunk_end[name] = self.mpos
        # ---- what becomes possible once this chunk's accumulators are complete (three gaps behind its last MFMA)
        if ctype == "q" and f == 1:
            self.gate(3)
            self.q_pack(QA, QA, h, "q")
        elif ctype == "k" and f == 1:
            self.gate(3)
            self.q_pack(KA, KA, h, "k")
            self.queue.append(("GAP",))
            self.q_scores(h)
            self.queue.append(("GATE3",))
            self.q_softmax(h)
        elif ctype == "v" and f == 1:
            self.gate(3)
            self.q_pack(VA, VA, h, "v")
            self.queue.append(("GAP",))
            self.q_heads_out(h)
            self.queue.append(("GATE3",))
            if h:                # the one packed-O^T buffer is free when head h - 1's out-projection has read it
                self.queue.append(("AFTER", "o1.%d" % (h - 1)))
            self.q_pack(OACC, OPK, h, "O")

    # ---- the statement of one panel ---------------------------------------------------------------------------------------
    def build(self):
        e = self.e
        cs = stream()
        assert len(cs) % NSLOT == 0
        e("s_mov_b32 s%d, %%[splo]" % SP, "salu")
        e("s_mov_b32 s%d, %%[sphi]" % (SP + 1), "salu")
        import struct
        e("s_mov_b32 s%d, 0x%08x" % (SCL, struct.unpack("<I", struct.pack("<f", 0.125 * 1.4426950408889634))[0]), "salu")
        # ones operand of the bias k-steps: k-slots 0, 1 of the h = 0 lanes
        e("v_and_b32_e32 %s, 0x3c003c00, %%[hmask]" % v(TMP), "valu", wr=[("v", TMP)])
        e("v_accvgpr_write_b32 a%d, %s" % (ONESB, v(TMP)), "valu", rd=[("v", TMP)], wr=[("a", ONESB)])
        for i in range(1, 4):
            e("v_accvgpr_write_b32 a%d, 0" % (ONESB + i), "valu", wr=[("a", ONESB + i)])
        # MASK: accumulator register r of a lane (query n, half hh) is key row (r & 3) + 8 (r >> 2) + 4 hh: the lane's own
        # pixel is rows 16 p .. 16 p + 15 (p = n >> 4), frame slot 8 (j & 1) + 4 hh + i of it must be < F.
        #   %[pm0] / %[pm1]: 0 where the lane's pixel is 0 / 1, -30000 elsewhere;  %[flim] = F - 4 hh
        for r in range(16):
            j, i = r >> 2, r & 3
            e("v_cmp_lt_i32_e32 vcc, %d, %%[flim]" % (8 * (j & 1) + i), "valu")
            e("v_mov_b32_e32 %s, %s" % (v(MASK + r), NEG), "valu", wr=[("v", MASK + r)])
            e("v_cndmask_b32_e32 %s, %s, %%[pm%d], vcc" % (v(MASK + r), v(MASK + r), j >> 1), "valu", rd=[("v", MASK + r)], wr=[("v", MASK + r)])
        e("s_waitcnt vmcnt(0)", "waitvm")
        e("s_barrier", "barrier")
        self.pending_dma = self.dma_items(3 % NSLOT, True, wrap=False)
        self.emit_dma_all()
        for j in range(NRING):
            self.read_frag(0, j, j, (cs[0][0], j))
        for n in range(len(cs)):
            late = [it for it in self.queue if isinstance(it, Ins) and it.meta.get("head") == cs[n][2]]
            if cs[n][1] == "o" and late:            # the packed head outputs this chunk multiplies must be complete
                # (only the last head's sixteen conversions come here: nothing is left to hide them behind)
                assert cs[n][2] == HEADS - 1 and cs[n][3] == 0 and all(isinstance(it, tuple) or it.meta.get("what") == "O" for it in self.queue)
                self.nop(7)                     # the out-projection MFMAs just issued still read the buffer
                self.nop(7)
                for it in self.queue:
                    if isinstance(it, Ins):
                        self.ins.append(it)
                self.queue = []
                self.nop(1)
            self.chunk(n, cs)
        assert not self.queue, "queued work left over"
        self.emit_dma_all()
        self.nop(15)
        self.nop(15)
        e("s_mov_b32 %%[splo], s%d" % SP, "salu")
        e("s_mov_b32 %%[sphi], s%d" % (SP + 1), "salu")

    # ---- counted lgkmcnt waits ------------------------------------------------------------------------------------------------
    def resolve_waits(self):
        out, fifo, retired = [], [], set()
        for i in self.ins:
            if i.kind == "lds":
                fifo.append(i.meta["frag"])
                out.append(i)
            elif i.kind == "waitfrag":
                fr = i.meta["frag"]
                idx = [k for k, f in enumerate(fifo) if f == fr]
                if not idx:
                    assert fr in retired, ("fragment never read", fr)
                    continue
                keep = len(fifo) - 1 - idx[-1]
                assert keep <= 15
                out.append(Ins("s_waitcnt lgkmcnt(%d)" % keep, "waitlgkm", n=keep))
                retired.update(fifo[:idx[-1] + 1])
                fifo = fifo[idx[-1] + 1:]
            else:
                out.append(i)
        self.ins = out

    # ---- checks ------------------------------------------------------------------------------------------------------------------
    def check(self):
        def ws(i):
            return i.meta["n"] + 1 if i.kind == "nop" else (0 if i.kind in ("label", "comment") else (8 if i.kind == "mfma" else 1))

        walk = self.ins
        last_mfma_wr, last_valu_wr, last_trans_wr = {}, {}, {}
        pos = nerr = 0
        for i in walk:
            if i.kind in ("label", "comment"):
                continue
            for r in i.rd + i.wr:
                if r in last_mfma_wr:
                    same_chain = i.kind == "mfma" and i.meta.get("acc") and r in i.wr and r in i.rd
                    if not same_chain and pos - last_mfma_wr[r] < 20:
                        print("HAZARD mfma->use %s dist %d: %s" % (r, pos - last_mfma_wr[r], i.text))
                        nerr += 1
            if i.kind == "mfma":
                for r in i.rd:
                    if r in last_valu_wr and pos - last_valu_wr[r] < 3:
                        print("HAZARD valu->mfma %s: %s" % (r, i.text))
                        nerr += 1
            if i.kind in ("valu", "trans", "lds"):
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
                if r[0] in ("v", "a") and r not in written:
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
            elif i.kind == "mfma" and not i.meta.get("inline"):
                regs = [r for r in i.rd if r[0] == "v" and RING <= r[1] < RING + 4 * NRING]
                assert len(regs) == 4
                want = i.meta["frag"]
                for r in regs:
                    if slotfrag.get(r) != want:
                        print("RING slot %s holds %s, MFMA expects %s" % (r, slotfrag.get(r), want))
                        nerr += 1
                if want in pending:
                    print("RING fragment not waited for: %s" % (want,))
                    nerr += 1
            elif i.kind in ("valu", "trans"):
                for r in i.rd:
                    if slotfrag.get(r) in pending and slotfrag.get(r, ("",))[0] == "X":
                        print("exchange result not waited for: %s" % i.text)
                        nerr += 1
        assert nerr == 0, "%d problems" % nerr

    def text(self):
        knob = os.environ.get("TB_GEN_KNOB", "").split("+")      # timing experiments only: results WRONG
        keep = []
        for i in self.ins:
            if i.kind == "comment":
                continue
            if "nolds" in knob and i.kind in ("lds", "waitlgkm"):
                continue
            if "novalu" in knob and i.kind in ("valu", "trans") and "head" in i.meta:
                continue
            if "nomfma" in knob and i.kind == "mfma":
                continue
            if "nobar" in knob and i.kind in ("barrier", "waitvm", "vmem"):
                continue
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
            print("%-22s issue cycles %5d (matrix pipe %4d)  %s" % (k, tot[k], 32 * n[k].get("mfma", 0), n[k]))
        print("instructions:", sum(1 for i in self.ins if i.kind not in ("comment", "label")),
              " MFMAs:", sum(1 for i in self.ins if i.kind == "mfma"))


def main():
    g = Gen()
    g.build()
    g.resolve_waits()
    g.check()
    if "--stats" in sys.argv:
        g.stats()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lkgd_amd", "csrc", "attn_tblock_loop.inc")
    if "-o" in sys.argv:
        out = sys.argv[sys.argv.index("-o") + 1]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_tblock_asm.py - do not edit.  Panel statement of attn_tblock.hip (plan: see that script).\n")
        f.write("#define TB_VB %d\n#define TB_VEND %d\n#define TB_AEND %d\n#define TB_YACC %d\n#define TB_ZF %d\n" % (VB, VEND, AEND, YACC, ZF))
        f.write("#define TB_W1_BYTES %d\n#define TB_W2_BYTES %d\n#define TB_SLOT %d\n#define TB_NSLOT %d\n#define TB_NCHUNK %d\n" %
                (W1_BYTES, W2_BYTES, SLOT, NSLOT, len(stream())))
        f.write("#define TB_PANEL_ASM \\\n  %s\n\n" % g.text())
        clob = ['"v%d"' % i for i in range(VB, VEND)] + ['"a%d"' % i for i in range(AEND)] + ['"s%d"' % i for i in range(SP, SCL + 1)]
        f.write("#define TB_CLOBBERS " + ", ".join(clob) + ', "vcc", "scc", "m0", "memory"\n')
    print("wrote", out)


if __name__ == "__main__":
    main()
