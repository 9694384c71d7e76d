"""Weight re-layout for the gfx950 kernels (done once at load time, on whatever device the weights live on).

All GEMM-shaped weights become fp16 [N][K] with K contiguous and K ordered the way the kernel's A-operand gather walks
it (include/lkgd_hip.h section 1)."""
from __future__ import annotations

import torch


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> fp16 [N, K]"""
    return w.reshape(w.shape[0], -1).to(torch.float16).contiguous()


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [Cout, Cin, 3, 3] -> fp16 [Cout, 9*Cin], k = ((ky*(Cin/64) + c//64)*3 + kx)*64 + c%64: for one kernel
    row and one 64-channel chunk the three horizontal taps are consecutive K-tiles (they read the same source cache lines
    shifted by one pixel - lkgd_amd/csrc/gemm_common.h::conv_k_decode).  Cin must be a multiple of 64."""
    co, ci = w.shape[:2]
    assert ci % 64 == 0, "3x3 implicit GEMM needs Cin % 64 == 0 (conv_in uses pack_conv3x3_c8)"
    t = w.reshape(co, ci // 64, 64, 3, 3).permute(0, 3, 1, 4, 2)      # [co, ky, chunk, kx, 64]
    return t.reshape(co, 9 * ci).to(torch.float16).contiguous()


def pack_conv3x3_c8(w: torch.Tensor) -> torch.Tensor:
    """conv_in weight [Cout, 8, 3, 3] -> fp16 [Cout, 128]: 9 taps x 8 channels, zero padded to K = 128"""
    co, ci = w.shape[:2]
    assert ci == 8, "LKGD_A_CONV3X3_C8 is the conv_in path (8 input channels)"
    out = torch.zeros(co, 128, dtype=torch.float16, device=w.device)
    out[:, :72] = w.permute(0, 2, 3, 1).reshape(co, 72).to(torch.float16)
    return out


def pack_tconv3(w: torch.Tensor) -> torch.Tensor:
    """Conv3d (3,1,1) weight [Cout, Cin, 3, 1, 1] -> fp16 [Cout, 3*Cin], k = kt*Cin + c"""
    co, ci = w.shape[:2]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(co, 3 * ci).to(torch.float16).contiguous()


def pack_tfront(wqkv: torch.Tensor, heads: int) -> torch.Tensor:
    """fused projection [3C, C] (rows q | k | v, head-major) -> the MFMA-fragment stream of lkgd_tattn_front:
    [head][q,k,v][fragment i of 16 rows][K-step ks of 32][lane = 16*lq + row][8 halfs at k = 32 ks + 8 lq]"""
    c3, c = wqkv.shape
    assert c3 == 3 * c and c == heads * 64 and c % 32 == 0
    t = wqkv.to(torch.float16).reshape(3, heads, 4, 16, c // 32, 4, 8)      # [which, h, i, row, ks, lq, e]
    return t.permute(1, 0, 2, 4, 5, 3, 6).contiguous().reshape(-1)


def pack_tblock(wqkv: torch.Tensor, bqkv, wo: torch.Tensor, heads: int = 5) -> torch.Tensor:
    """Chunk stream of lkgd_tattn_block_c320 (lkgd_amd/csrc/attn_tblock.hip, tools/gen_tblock_asm.py): the fused projection
    wqkv [960, 320] (rows q | k | v, head-major, LayerNorm affine folded in) with bias bqkv [960] (or None), and the
    out-projection wo [320, 320].  Returns fp16 [n] = 40 chunks in the order of tools/gen_tblock_asm.py::stream(): per head h
        q0 q1 (h) | o0 o1 (h - 1; heads 1..3 only) | k0 k1 v0 v1 (h);  the stream ends with o0 o1 of heads 3 and 4.
    q / k / v chunk (tile f = 32 of the head's 64 channels) = 21 MFMA fragments of 64 lanes x 8 halfs (lane = 32 hh + row):
    fragment 0 carries the bias as (b_hi, b_lo) in k-slots 0, 1 of the hh = 0 lanes, fragments 1..20 W[base + row][16 ks + 8 hh
    + e] - q and k use them as the A operand (rows = head channels), v as the B operand (columns = head channels): same bytes.
    o chunk (tile f of head h) = 20 fragments (k-step ss, output tile ti): Wo[32 ti + row][64 h + 32 f + 16 ss + (e & 3) +
    8 (e >> 2) + 4 hh] - the k-slot order in which the attention's accumulators hold the head channels."""
    assert heads == 5 and wqkv.shape == (960, 320) and wo.shape == (320, 320)
    dev = wqkv.device
    w = wqkv.to(torch.float16).reshape(3, heads, 2, 32, 20, 2, 8)           # [which, h, f, row, ks, hh, e]
    b = (bqkv if bqkv is not None else torch.zeros(960, device=dev)).to(torch.float32).reshape(3, heads, 2, 32)
    b_hi = b.to(torch.float16)
    b_lo = (b - b_hi.to(torch.float32)).to(torch.float16)
    h_, e_ = torch.arange(2, device=dev)[:, None], torch.arange(8, device=dev)[None, :]
    kidx = (e_ & 3) + 8 * (e_ >> 2) + 4 * h_                                 # [hh, e] -> position inside a 16-channel k-step
    woh = wo.to(torch.float16).reshape(10, 32, heads, 2, 2, 16)              # [ti, row, h, f, ss, k16]

    def proj_chunk(t, h, f):
        bias = torch.zeros(2, 32, 8, dtype=torch.float16, device=dev)
        bias[0, :, 0], bias[0, :, 1] = b_hi[t, h, f], b_lo[t, h, f]
        body = w[t, h, f].permute(1, 2, 0, 3)                                # [ks, hh, row, e]
        return torch.cat([bias.reshape(1, -1), body.reshape(20, -1)]).reshape(-1)

    def out_chunk(h, f):
        blk = woh[:, :, h, f][..., kidx]                                     # [ti, row, ss, hh, e]
        return blk.permute(2, 0, 3, 1, 4).reshape(-1)                        # [ss, ti, hh, row, e]

    chunks = []
    for h in range(heads):
        chunks += [proj_chunk(0, h, 0), proj_chunk(0, h, 1)]
        if 0 < h < heads - 1:
            chunks += [out_chunk(h - 1, 0), out_chunk(h - 1, 1)]
        chunks += [proj_chunk(t, h, f) for t in (1, 2) for f in (0, 1)]
    chunks += [out_chunk(h, f) for h in (heads - 2, heads - 1) for f in (0, 1)]
    out = torch.cat(chunks).contiguous()
    assert out.numel() * 2 == 30 * 21504 + 10 * 20480
    return out


def pack_ln_proj(w: torch.Tensor, b) -> torch.Tensor:
    """Chunk stream of lkgd_ln_qkv_c320 / _c640 (lkgd_amd/csrc/qkv_fused.hip, tools/gen_qkv_asm.py): w [3C, C] (LayerNorm affine
    folded in), b [3C] or None, C = 320 or 640.  Returns fp16 [n]: per 32 output rows (a tile) C / 320 chunks - the first one 21 MFMA
    A fragments of 64 lanes x 8 halfs (lane = 32 hh + row): fragment 0 the bias as (b_hi, b_lo) in k-slots 0, 1 of the hh = 0 lanes,
    then k-steps 0..19: w[32 t + row][16 ks + 8 hh + e]; a second chunk (C = 640) k-steps 20..39."""
    n_out, c = w.shape
    assert c in (320, 640) and n_out == 3 * c
    dev = w.device
    tiles, nks = n_out // 32, c // 16
    wh = w.to(torch.float16).reshape(tiles, 32, nks, 2, 8)                  # [tile, row, ks, hh, e]
    bf = (b if b is not None else torch.zeros(n_out, device=dev)).to(torch.float32).reshape(tiles, 32)
    b_hi = bf.to(torch.float16)
    b_lo = (bf - b_hi.to(torch.float32)).to(torch.float16)
    bias = torch.zeros(tiles, 2, 32, 8, dtype=torch.float16, device=dev)
    bias[:, 0, :, 0], bias[:, 0, :, 1] = b_hi, b_lo
    body = wh.permute(0, 2, 3, 1, 4).reshape(tiles, nks, -1)                # [tile, ks, (hh, row, e)]
    out = torch.cat([bias.reshape(tiles, 1, -1), body], dim=1).reshape(-1).contiguous()    # bias | ks 0..nks-1 per tile
    assert out.numel() * 2 == tiles * (21504 + (nks // 20 - 1) * 20480)
    return out


def geglu_perm(inner: int, half: int = 32, device=None) -> torch.Tensor:
    """row permutation of the GEGLU projection [2*inner, K]: every 2*half packed rows = `half` hidden rows followed by
    their `half` gate rows (inner + ..), so that the output columns one wave owns hold both factors of its GEGLU
    outputs.  half = 80 for the 256x320-tile kernel (wave = 160 columns), 32 for the 128-wide-tile kernels."""
    assert inner % half == 0
    t = torch.arange(inner // half, device=device)[:, None] * half + torch.arange(half, device=device)[None, :]
    return torch.cat([t, t + inner], dim=1).reshape(-1)


def geglu_half(n_packed_rows: int, k: int = 0) -> int:
    """interleave width the kernels want for a GEGLU projection with 2*inner = n_packed_rows rows and K = k inputs"""
    # the 256x320 kernel, 80 hidden | 80 gate per wave, when the rows tile by 320 (measured: profiles/r01_gemm_shapes_ab*.txt;
    # since its bias strip comes through LDS it also leads at K = 320); otherwise 32 | 32 for the other kernels
    if k >= 320 and n_packed_rows % 320 == 0:
        return 80
    return 32


def pack_geglu(w: torch.Tensor, b: torch.Tensor, half: int = None):
    inner = w.shape[0] // 2
    if half is None:
        half = geglu_half(w.shape[0], w.shape[1])
    perm = geglu_perm(inner, half, w.device)
    return w[perm].to(torch.float16).contiguous(), b[perm].to(torch.float32).contiguous(), half


def pack_ff_fused(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """Chunk stream of lkgd_ff_fused_c320 (lkgd_amd/csrc/ff_fused.hip, tools/gen_ff_asm.py) for a GEGLU feed-forward
    320 -> 2 x 1280 -> 320.  w1 [2560, 320] / b1 [2560] fp32: the GEGLU projection in diffusers' row order (hidden rows
    0..1279, gate rows 1280..2559) with the LayerNorm affine already folded in; w2 [320, 1280].  Returns fp16 [n] = 120 chunks:
        unit u = 64 hidden channels, tile f = 32 of them:  c0 W1 hidden f0 | c1 W1 gate f0 | c2 W2 f1 of unit u-1 (u > 0) |
        c3 W1 hidden f1 | c4 W1 gate f1 | c5 W2 f0; after unit 19 the stream ends with W2 f1 of unit 19.
    W1 chunk = 21 MFMA A fragments of 64 lanes x 8 halfs (lane = 32 h + row): fragment 0 carries the bias as (b_hi, b_lo) in
    k-slots 0, 1 of the h = 0 lanes (multiplied by ones in the kernel: b_hi + b_lo restores fp32 to 2^-22), fragments 1..20
    W1[base + row][16 ks + 8 h + e].  W2 chunk = 20 fragments (k-step ss, output tile ti): W2[32 ti + row][64 u + 32 f + 16 ss +
    (e & 3) + 8 (e >> 2) + 4 h] - the k-slot order in which the first product's accumulators hold the hidden channels."""
    assert w1.shape == (2560, 320) and b1.shape == (2560,) and w2.shape == (320, 1280)
    dev = w1.device
    w1h = w1.to(torch.float16).reshape(2, 20, 2, 32, 20, 2, 8)            # [type, u, f, row, ks, h, e]
    b1f = b1.to(torch.float32).reshape(2, 20, 2, 32)
    b_hi = b1f.to(torch.float16)
    b_lo = (b1f - b_hi.to(torch.float32)).to(torch.float16)
    h_, e_ = torch.arange(2, device=dev)[:, None], torch.arange(8, device=dev)[None, :]
    kidx = (e_ & 3) + 8 * (e_ >> 2) + 4 * h_                               # [h, e] -> position inside a 16-channel k-step
    w2h = w2.to(torch.float16).reshape(10, 32, 20, 2, 2, 16)               # [ti, row, u, f, ss, k16]

    def w1_chunk(t, u, f):
        bias = torch.zeros(2, 32, 8, dtype=torch.float16, device=dev)
        bias[0, :, 0], bias[0, :, 1] = b_hi[t, u, f], b_lo[t, u, f]
        body = w1h[t, u, f].permute(1, 2, 0, 3)                             # [ks, h, row, e]
        return torch.cat([bias.reshape(1, -1), body.reshape(20, -1)]).reshape(-1)

    def w2_chunk(u, f):
        blk = w2h[:, :, u, f][..., kidx]                                    # [ti, row, ss, h, e]
        return blk.permute(2, 0, 3, 1, 4).reshape(-1)                       # [ss, ti, h, row, e]

    chunks = []
    for u in range(20):
        chunks += [w1_chunk(0, u, 0), w1_chunk(1, u, 0)]
        if u:
            chunks.append(w2_chunk(u - 1, 1))
        chunks += [w1_chunk(0, u, 1), w1_chunk(1, u, 1), w2_chunk(u, 0)]
    chunks.append(w2_chunk(19, 1))
    out = torch.cat(chunks).contiguous()
    assert out.numel() * 2 == 80 * 21504 + 40 * 20480
    return out
