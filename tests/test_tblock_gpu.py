"""lkgd_tattn_block_c320 (lkgd_amd/csrc/attn_tblock.hip): LayerNorm + Q|K|V + attention over the frames + out-projection +
residual of a temporal transformer block in one launch, against the same chain in fp32 (patch/patch.py:610, :660-661)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    wqkv = torch.randn(960, 320, generator=g) / 320 ** 0.5
    bqkv = 0.3 * torch.randn(960, generator=g)
    wo = torch.randn(320, 320, generator=g) / 320 ** 0.5
    bo = 0.3 * torch.randn(320, generator=g)
    return wqkv, bqkv, wo, bo


def _ref(x, wqkv, bqkv, wo, bo, B, Fr, HW, rowbias=None, idx=None):
    """x [B*Fr*HW, 320] (row = (b*Fr + f)*HW + pixel), fp32 math on the fp16-rounded weights"""
    xf = x.float()
    z = F.layer_norm(xf, (320,)).half().float()        # the normalised rows are fp16 matrix operands (as in the unfused chain)
    qkv = z @ wqkv.half().float().T + bqkv
    q, k, v = (t.half().float().reshape(B, Fr, HW, 5, 64).permute(0, 2, 3, 1, 4) for t in qkv.split(320, dim=1))   # [B, HW, 5, Fr, 64]
    o = F.scaled_dot_product_attention(q, k, v)                                  # over the frames of a pixel
    o = o.permute(0, 3, 1, 2, 4).reshape(B * Fr * HW, 320)
    out = o @ wo.half().float().T + bo + xf
    if rowbias is not None:
        out = out + rowbias.float()[idx]
    return out


def _close(got, ref, what):
    err = (got.float().cpu() - ref).abs().max().item()
    rel = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    assert err < 2.5e-2 and rel < 2e-3, (what, err, rel)


@pytest.mark.parametrize("B,Fr,HW", [(1, 14, 16), (2, 14, 24), (1, 3, 8), (1, 16, 8), (2, 5, 13), (1, 1, 40), (2, 14, 1152)])
def test_tblock_vs_fp32(B, Fr, HW):
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_tblock
    wqkv, bqkv, wo, bo = _weights(B * 100 + Fr)
    g = torch.Generator().manual_seed(HW)
    T = B * Fr * HW
    x = (torch.randn(T, 320, generator=g) * 1.5 + 0.3).half()
    ws = pack_tblock(wqkv, bqkv, wo).to(DEV)
    out = torch.full((T, 320), float("nan"), dtype=torch.float16, device=DEV)
    ops.tattn_block(x.to(DEV), ws, bo.to(DEV), out, B, Fr, HW)
    assert torch.isfinite(out.float()).all()
    _close(out, _ref(x, wqkv, bqkv, wo, bo, B, Fr, HW), f"tblock {B}x{Fr}x{HW}")
    again = torch.empty_like(out)
    ops.tattn_block(x.to(DEV), ws, bo.to(DEV), again, B, Fr, HW)
    assert torch.equal(out, again)


def test_tblock_row_bias_and_sharp_scores():
    """the folded one-token cross-attention as a row-indexed table; keys with very different scores (softmax close to one-hot)"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_tblock
    wqkv, bqkv, wo, bo = _weights(5)
    wqkv[:640] *= 6.0                                   # |scores| up to a few hundred
    g = torch.Generator().manual_seed(6)
    B, Fr, HW = 2, 14, 40
    T = B * Fr * HW
    x = torch.randn(T, 320, generator=g).half()
    table = torch.randn(4, 320, generator=g).half()
    rmap = (Fr * HW, HW, HW, 4, 3)                     # ((row / d1) * m1 + row % d2 + c0) % md
    rows = torch.arange(T)
    idx = ((rows // rmap[0]) * rmap[1] + rows % rmap[2] + rmap[4]) % rmap[3]
    ws = pack_tblock(wqkv, bqkv, wo).to(DEV)
    out = torch.empty(T, 320, dtype=torch.float16, device=DEV)
    ops.tattn_block(x.to(DEV), ws, bo.to(DEV), out, B, Fr, HW, rowbias=table.to(DEV), rowmap=rmap)
    _close(out, _ref(x, wqkv, bqkv, wo, bo, B, Fr, HW, rowbias=table, idx=idx), "row bias form")


def test_tblock_equals_front_plus_out_projection_and_same_rows_same_bits():
    """against the two-launch form it replaces (fused front + resident-weight out-projection), at the 72x128 level; and the two
    CFG halves (same rows, 1152 panels apart) agree bitwise"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_linear, pack_tblock, pack_tfront
    wqkv, bqkv, wo, bo = _weights(9)
    g = torch.Generator().manual_seed(10)
    B, Fr, HW = 2, 14, 9216
    x0 = torch.randn(Fr * HW, 320, generator=g).half().to(DEV)
    x = torch.cat([x0, x0])
    T = x.shape[0]
    ws = pack_tblock(wqkv, bqkv, wo).to(DEV)
    out = torch.empty_like(x)
    ops.tattn_block(x, ws, bo.to(DEV), out, B, Fr, HW)
    assert torch.equal(out[:Fr * HW], out[Fr * HW:])
    att = torch.empty_like(x)
    ops.tattn_front(x, pack_tfront(wqkv.half().to(DEV), 5), bqkv.to(DEV), att, B, Fr, HW, 5)
    chain = torch.empty_like(x)
    ops.gemm(att, pack_linear(wo).to(DEV), chain, M=T, N=320, K=320, bias=bo.to(DEV), res1=x)
    assert (out.float() - chain.float()).abs().max().item() < 2e-2
    assert ((out.float() - chain.float()).norm() / chain.float().norm()).item() < 1.5e-3
