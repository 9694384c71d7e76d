#!/usr/bin/env python3
r"""Generates lkgd_amd/csrc/qkv_fused_loop.inc: the panel statement of qkv_fused.hip - LayerNorm + a 320 -> 960 projection
(to_q | to_k | to_v of the spatial transformer block at the 72x128 level, patch/patch.py:416,440-445) in ONE kernel, the
rows read once and the normalised copy never written.
Run from the repo root:  python tools/gen_qkv_asm.py   (--stats for the per-chunk issue-cost table)

The skeleton of the fused feed-forward / one-launch temporal attention (tools/gen_ff_asm.py, gen_tblock_asm.py): one wave = 32
token rows, LayerNorm-ed in registers and parked as MFMA B operands in a[0:79]; one workgroup = 4 waves (one per SIMD) = a
128-token panel; the 960 x 320 weights stream L2 -> LDS as 30 chunks of 21 KiB (a bias fragment + 20 k-step fragments = one
32-row output tile) through a ring of five slots, three chunks ahead (LDS-DMA, counted vmcnt), one barrier per chunk, fragment
reads eight MFMAs ahead across chunk borders.  There is no second product here: tile n's accumulators (H^T = W . z^T: a lane owns
a token, registers run over the tile's 32 output channels) are converted and STORED - two 16-byte global stores per lane after a half-wave exchange - in the
MFMA gaps of tile n + 1, from the other of two accumulator sets.  The stores count in vmcnt like the DMA pieces: every counted
wait is computed from the statement's own issue log (the statement is straight-line), not from a fixed formula.

Register plan (named, clobbered): v[24:55] two accumulator tiles, v[56:71] two packed tiles (8 registers each), v[72:103]
fragment ring; a[0:79] z^T, a[80:83] the ones operand of the bias k-step.  Checks as in gen_tblock_asm.py.
"""
import os
import sys

NL = r"\n\t"
VB = 24
ACC = [VB, VB + 16]
PK = [VB + 32, VB + 40]
NRING = 8
RING = VB + 48
TMP = RING + 4 * NRING
VEND = TMP + 2
C_IN = int(os.environ.get("QKV_GEN_C", "320"))      # 320 (30 tiles, the 72x128 level) or 640 (60 tiles, the 36x64 level)
assert C_IN in (320, 640)
NKS = C_IN // 16
PARTS = NKS // 20         # chunks per tile: a chunk is the bias fragment + 20 k-steps, or 20 further k-steps (<= 21 KiB)
ZF, ONESB, AEND = 0, 4 * NKS, 4 * NKS + 4
W1_FR = 21
W1_BYTES = W1_FR * 1024
W2_BYTES = 20 * 1024
SLOT = 24576
AHEAD = int(os.environ.get("QKV_GEN_AHEAD", "3"))      # chunks in flight ahead of the one that becomes visible
NSLOT = AHEAD + 2
WAITN = 4
NTILE = 3 * C_IN // 32    # 3 C output channels
NCH = NTILE * PARTS
RATE = 2
SP = 68                   # s[68:69]: the weight stream pointer


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


class Gen:
    def __init__(self):
        self.ins = []
        self.queue = []          # Ins or ("GATE", ring-MFMA position)
        self.mpos = 0
        self.ringpos = 0
        self.pending_dma = []
        self.dma_half = False
        self.vmlog = []          # issue order of vector-memory operations: ("dma", chunk) / ("st", tile)

    def e(self, text, kind, rd=(), wr=(), **meta):
        self.ins.append(Ins(text, kind, rd, wr, **meta))
        if kind == "vmem":
            self.vmlog.append(meta["vm"])

    def nop(self, n):
        self.e("s_nop %d" % n, "nop", n=n)

    def vm_after(self, chunk):
        """vector-memory operations issued behind the last DMA piece of `chunk` (all of them, if its pieces are older than this
        statement): what a counted wait for that chunk may leave in flight"""
        last = max((k for k, t in enumerate(self.vmlog) if t == ("dma", chunk)), default=-1)
        return len(self.vmlog) - 1 - last

    # ---- the queued work of a finished tile: conversions and stores ------------------------------------------------------
    def q_store(self, n):
        a, p = ACC[n & 1], PK[n & 1]
        for r in range(0, 16, 2):
            self.queue.append(Ins("v_cvt_pk_f16_f32 %s, %s, %s" % (v(p + r // 2), v(a + r), v(a + r + 1)), "valu",
                                  rd=[("v", a + r), ("v", a + r + 1)], wr=[("v", p + r // 2)], tile=n))
        # accumulator registers 4 g .. 4 g + 3 of lane (token, hh) are channels 32 n + 8 g + 4 hh + 0..3: packed, group g is the
        # register pair p + 2 g.  One half-wave exchange per dword of a group PAIR (v_permlane32_swap: lanes 32..63 of the first
        # operand <-> lanes 0..31 of the second) leaves lanes 0..31 with channels 16 j + 0..7 and lanes 32..63 with 16 j + 8..15
        # in four consecutive registers: one 16-byte store per pair (cdna_hip_programming.md T21; the row pointer of the upper
        # half-wave is 16 bytes further).  Its operands must have been written >= 2 wait states before (checked below).
        for j in range(2):
            for d in range(2):
                a, b = p + 4 * j + d, p + 4 * j + 2 + d
                self.queue.append(Ins("v_permlane32_swap_b32 %s, %s" % (v(a), v(b)), "swap", rd=[("v", a), ("v", b)],
                                      wr=[("v", a), ("v", b)], tile=n))
        for j in range(2):
            self.queue.append(Ins("global_store_dwordx4 %%[orow], %s, off offset:%d" % (vr(p + 4 * j, 4), 64 * n + 32 * j), "vmem",
                                  rd=R(p + 4 * j, 4), vm=("st", n), tile=n))

    def dispense(self):
        k = 0
        while self.queue and k < RATE:
            it = self.queue[0]
            if isinstance(it, tuple):
                if self.mpos < it[1]:
                    break
                self.queue.pop(0)
                continue
            self.queue.pop(0)
            self.ins.append(it)
            if it.kind == "vmem":
                self.vmlog.append(it.meta["vm"])
            k += 1

    # ---- fragment reads / DMA -----------------------------------------------------------------------------------------------
    def read_frag(self, slot, frag_i, ring_slot, tag):
        reg = RING + 4 * ring_slot
        base, imm = slot_addr(slot)
        self.e("ds_read_b128 %s, %s offset:%d" % (vr(reg, 4), base, imm + frag_i * 1024), "lds", wr=R(reg, 4), frag=tag)

    def dma_items(self, chunk, wrap):
        slot = chunk % NSLOT
        first = chunk % PARTS == 0           # the chunk that opens a tile carries the bias fragment (a 21st KiB)
        it = []
        for j in range(5):
            it.append(("s_add_u32 m0, %%[ldsw], %d" % (slot * SLOT + j * 4096),
                       "global_load_lds_dwordx4 %%[vo%d], s[%d:%d]" % (j, SP, SP + 1)))
        if first:
            it.append(("s_add_u32 m0, %%[lds0], %d" % (slot * SLOT + 20480),
                       "global_load_lds_dwordx4 %%[vob], s[%d:%d]" % (SP, SP + 1)))
        if wrap:
            it.append(("s_mov_b32 s%d, %%[sp0lo]" % SP, "s_mov_b32 s%d, %%[sp0hi]" % (SP + 1)))
        else:
            it.append(("s_add_u32 s%d, s%d, %d" % (SP, SP, W1_BYTES if first else W2_BYTES), "s_addc_u32 s%d, s%d, 0" % (SP + 1, SP + 1)))
        self.dma_chunk = chunk
        return it

    def dma_first(self):
        if self.pending_dma:
            self.e(self.pending_dma[0][0], "salu")
            self.dma_half = True

    def dma_second(self):
        if self.dma_half:
            t = self.pending_dma.pop(0)[1]
            if t.startswith("global_load"):
                self.e(t, "vmem", vm=("dma", self.dma_chunk))
            else:
                self.e(t, "salu")
            self.dma_half = False

    def emit_dma_all(self):
        while self.pending_dma:
            self.dma_first()
            self.nop(0)
            self.dma_second()

    # ---- one chunk = (part of) one 32-channel output tile ---------------------------------------------------------------------
    def chunk(self, c):
        n, part = divmod(c, PARTS)
        self.e("; ---- tile %d part %d" % (n, part), "comment")
        nfr, slot = (W1_FR if part == 0 else 20), c % NSLOT
        last = c == NCH - 1
        tags = [(c, i) for i in range(nfr)]
        d = ACC[n & 1]
        if part == 0:
            for it in self.queue:       # the accumulator set of tile n - 2 must have been converted
                if isinstance(it, Ins):
                    assert not (set(R(d, 16)) & set(it.rd)), ("tile %d overwrites accumulators still to be converted" % n, it.text)
        for i in range(nfr):
            if i == nfr - NRING and not last:
                assert not self.pending_dma, "the previous chunk's DMA is still being issued"
                # chunk c + 1 becomes visible: own pieces landed (whatever was issued behind them may stay in flight)
                wait_n = self.vm_after(c + 1)
                assert wait_n <= 63, wait_n
                self.e("s_waitcnt vmcnt(%d)" % wait_n, "waitvm")
                self.e("s_barrier", "barrier")
                pre = c + 1 + AHEAD              # the next panel's first chunks from the last chunks on
                self.pending_dma = self.dma_items(pre, wrap=(pre % NCH == NCH - 1))
            self.dma_first()
            if i % WAITN == 0:
                self.e("WAITFRAG", "waitfrag", frag=tags[min(i + WAITN - 1, nfr - 1)])
            rs = self.ringpos % NRING
            self.ringpos += 1
            reg = RING + 4 * rs
            if part == 0 and i == 0:
                self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, 0" % (vr(d, 16), vr(reg, 4), ar(ONESB, 4)), "mfma",
                       rd=R(reg, 4) + R(ONESB, 4, "a"), wr=R(d, 16), frag=tags[i], acc=False)
            else:
                ks = 20 * part + (i - 1 if part == 0 else i)
                self.e("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (vr(d, 16), vr(reg, 4), ar(ZF + 4 * ks, 4), vr(d, 16)), "mfma",
                       rd=R(reg, 4) + R(ZF + 4 * ks, 4, "a") + R(d, 16), wr=R(d, 16), frag=tags[i], acc=True)
            self.mpos += 1
            j = i + NRING
            if j < nfr:
                self.read_frag(slot, j, rs, tags[j])
            elif not last:
                self.read_frag((c + 1) % NSLOT, j - nfr, rs, (c + 1, j - nfr))
            self.dma_second()
            self.dispense()
        if part == PARTS - 1:
            self.queue.append(("GATE", self.mpos + 3))
            self.q_store(n)

    # ---- the statement of one panel ---------------------------------------------------------------------------------------
    def build(self):
        e = self.e
        assert NCH % NSLOT == 0, (NCH, NSLOT)
        e("s_mov_b32 s%d, %%[splo]" % SP, "salu")
        e("s_mov_b32 s%d, %%[sphi]" % (SP + 1), "salu")
        e("v_and_b32_e32 %s, 0x3c003c00, %%[hmask]" % v(TMP), "valu", wr=[("v", TMP)])
        e("v_accvgpr_write_b32 a%d, %s" % (ONESB, v(TMP)), "valu", rd=[("v", TMP)], wr=[("a", ONESB)])
        for i in range(1, 4):
            e("v_accvgpr_write_b32 a%d, 0" % (ONESB + i), "valu", wr=[("a", ONESB + i)])
        # chunks 0..AHEAD-1 of this panel were issued a panel ago (or by the kernel, which then waits for everything itself); the
        # previous statement's LAST stores, issued behind them, may still be in flight (vmcnt retires in order): START_VM
        e("START_WAIT", "startwait")
        e("s_barrier", "barrier")
        self.pending_dma = self.dma_items(AHEAD, wrap=False)
        self.emit_dma_all()
        # the NEXT panel's token rows (this lane: 20 x 16 bytes of its row) into the statement's output registers: issued behind
        # the first MFMAs, in registers long before the statement ends (every later counted wait retires them first)
        if C_IN == 320 and "norows" not in os.environ.get("QKV_GEN_KNOB", "").split("+"):   # (640: 160 row registers do not fit)
            for ks in range(20):
                self.queue.append(Ins("global_load_dwordx4 %%[r%d], %%[xrow], off offset:%d" % (ks, 32 * ks), "vmem", vm=("row", ks)))
        for j in range(NRING):
            self.read_frag(0, j, j, (0, j))
        for c in range(NCH):
            self.chunk(c)
        # the last tile's conversions and stores have no MFMAs left to hide behind
        self.emit_dma_all()
        self.nop(7)
        self.nop(7)
        self.nop(7)
        for it in self.queue:
            if isinstance(it, Ins):
                self.ins.append(it)
                if it.kind == "vmem":
                    self.vmlog.append(it.meta["vm"])
        self.queue = []
        self.nop(1)
        start_vm = self.vm_after(NCH + AHEAD - 1)            # this statement's tail = the next statement's head
        for i in self.ins:
            if i.kind == "startwait":
                i.text, i.kind = "s_waitcnt vmcnt(%d)" % start_vm, "waitvm"
        self.start_vm = start_vm
        e("s_mov_b32 %%[splo], s%d" % SP, "salu")
        e("s_mov_b32 %%[sphi], s%d" % (SP + 1), "salu")

    # ---- counted lgkmcnt waits ------------------------------------------------------------------------------------------------
    def resolve_waits(self):
        out, fifo = [], []
        for i in self.ins:
            if i.kind == "lds":
                fifo.append(i.meta["frag"])
                out.append(i)
            elif i.kind == "waitfrag":
                fr = i.meta["frag"]
                idx = [k for k, f in enumerate(fifo) if f == fr]
                assert idx, ("fragment never read", fr)
                keep = len(fifo) - 1 - idx[-1]
                assert keep <= 15
                out.append(Ins("s_waitcnt lgkmcnt(%d)" % keep, "waitlgkm", n=keep))
                fifo = fifo[idx[-1] + 1:]
            else:
                out.append(i)
        self.ins = out

    # ---- checks ------------------------------------------------------------------------------------------------------------------
    def check(self):
        def ws(i):
            return i.meta["n"] + 1 if i.kind == "nop" else (0 if i.kind in ("label", "comment") else (8 if i.kind == "mfma" else 1))

        walk = self.ins
        last_mfma_wr, last_valu_wr, store_rd = {}, {}, {}
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
            if i.kind == "swap":          # VALU write -> v_permlane32_swap read: 2 wait states (LLVM gfx950 hazard rule)
                for r in i.rd:
                    if r in last_valu_wr and pos - last_valu_wr[r] < 3:
                        print("HAZARD valu->permlane swap %s: %s" % (r, i.text))
                        nerr += 1
            if i.kind == "vmem":          # a store's data registers: written >= 2 wait states before, not rewritten for 2 after
                for r in i.rd:
                    if r in last_valu_wr and pos - last_valu_wr[r] < 2:
                        print("HAZARD valu->store data %s: %s" % (r, i.text))
                        nerr += 1
                    store_rd[r] = pos
            for r in i.wr:
                if r in store_rd and pos - store_rd[r] < 3:
                    print("HAZARD store data rewritten %s: %s" % (r, i.text))
                    nerr += 1
                last_mfma_wr.pop(r, None)
                last_valu_wr.pop(r, None)
                if i.kind == "mfma":
                    last_mfma_wr[r] = pos
                elif i.kind in ("valu", "trans", "swap"):
                    last_valu_wr[r] = pos
            pos += ws(i)
        written = set(("a", i) for i in range(ZF, ZF + 4 * NKS))
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
            elif i.kind == "mfma":
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
        # the token rows are retired by a counted wait: some DMA piece issued behind them is waited for
        rows = [k for k, t in enumerate(self.vmlog) if t[0] == "row"]
        if rows:
            assert any(t[0] == "dma" and t[1] < NCH and k > rows[-1] for k, t in enumerate(self.vmlog)), "row loads never retired"
        # every tile stored exactly once, four pieces
        st = [t for t in self.vmlog if t[0] == "st"]
        assert sorted(st) == sorted([("st", n) for n in range(NTILE) for _ in range(2)]), "stores"
        assert nerr == 0, "%d problems" % nerr

    def text(self):
        knob = os.environ.get("QKV_GEN_KNOB", "").split("+")      # timing experiments only: results WRONG
        keep = []
        for i in self.ins:
            if i.kind == "comment":
                continue
            if "nostore" in knob and i.kind == "vmem" and i.meta["vm"][0] == "st":
                continue
            if "nomfma" in knob and i.kind == "mfma":
                continue
            keep.append(i)
        return " \\\n  ".join('"' + i.text + NL + '"' for i in keep)

    def stats(self):
        print("instructions:", sum(1 for i in self.ins if i.kind not in ("comment", "label")),
              " MFMAs:", sum(1 for i in self.ins if i.kind == "mfma"), " stores:", sum(1 for t in self.vmlog if t[0] == "st"),
              " max counted vmcnt:", max(int(i.text.split("(")[1][:-1]) for i in self.ins if i.kind == "waitvm"))


def main():
    g = Gen()
    g.build()
    g.resolve_waits()
    g.check()
    if "--stats" in sys.argv:
        g.stats()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lkgd_amd", "csrc",
                       "qkv_fused_loop.inc" if C_IN == 320 else "qkv%d_fused_loop.inc" % C_IN)
    P = "QK" if C_IN == 320 else "QK%d" % (C_IN // 100)
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_qkv_asm.py - do not edit.  Panel statement of qkv_fused.hip (plan: see that script).\n")
        f.write("#define %s_VB %d\n#define %s_VEND %d\n#define %s_AEND %d\n#define %s_ZF %d\n#define %s_PARTS %d\n" % (P, VB, P, VEND, P, AEND, P, ZF, P, PARTS))
        f.write("#define %s_W1_BYTES %d\n#define %s_W2_BYTES %d\n#define %s_SLOT %d\n#define %s_NSLOT %d\n#define %s_NTILE %d\n#define %s_AHEAD %d\n" % (P, W1_BYTES, P, W2_BYTES, P, SLOT, P, NSLOT, P, NTILE, P, AHEAD))
        f.write("#define %s_PANEL_ASM \\\n  %s\n\n" % (P, g.text()))
        clob = ['"v%d"' % i for i in range(VB, VEND)] + ['"a%d"' % i for i in range(AEND)] + ['"s%d"' % i for i in range(SP, SP + 2)]
        f.write("#define %s_CLOBBERS " % P + ", ".join(clob) + ', "vcc", "scc", "m0", "memory"\n')
    print("wrote", out)


if __name__ == "__main__":
    main()
