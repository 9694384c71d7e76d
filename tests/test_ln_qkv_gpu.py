"""lkgd_ln_qkv_c320 (lkgd_amd/csrc/qkv_fused.hip): LayerNorm + the fused to_q | to_k | to_v projection of the 72x128 level in one
launch, against fp32 `layer_norm -> linear` (patch/patch.py:416, :440-445) and against the row-panel GEMM with the LayerNorm fold."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _weights(seed, C=320):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(3 * C, C, generator=g) / C ** 0.5
    b = 0.3 * torch.randn(3 * C, generator=g)
    return w, b


def _ref(x, w, b):
    z = F.layer_norm(x.float(), (x.shape[1],)).half().float()          # the normalised rows are fp16 matrix operands
    return z @ w.half().float().T + b


@pytest.mark.parametrize("C", [320, 640])
@pytest.mark.parametrize("T", [128, 32, 1000, 128 * 7 + 5, 128 * 300])
def test_ln_qkv_vs_fp32(T, C):
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_ln_proj
    w, b = _weights(T, C)
    g = torch.Generator().manual_seed(T + 1)
    x = (torch.randn(T, C, generator=g) * 1.5 + 0.3).half()
    ws = pack_ln_proj(w, b).to(DEV)
    out = torch.full((T, 3 * C), float("nan"), dtype=torch.float16, device=DEV)
    ops.ln_qkv(x.to(DEV), ws, out)
    ref = _ref(x, w, b)
    err = (out.float().cpu() - ref).abs().max().item()
    rel = ((out.float().cpu() - ref).norm() / ref.norm()).item()
    assert err < 2e-2 and rel < 1e-3, (T, C, err, rel)
    again = torch.empty_like(out)
    ops.ln_qkv(x.to(DEV), ws, again)
    assert torch.equal(out, again)


def test_ln_qkv_into_a_wider_buffer_and_rows_with_large_means():
    """ldo > 960 (the projection lands in columns of a wider buffer); rows with |mean| >> sigma"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_ln_proj
    w, b = _weights(3)
    g = torch.Generator().manual_seed(4)
    T = 700
    x = torch.randn(T, 320, generator=g)
    x[::7] += 40.0
    x = x.half()
    ws = pack_ln_proj(w, b).to(DEV)
    big = torch.full((T, 1024), 7.0, dtype=torch.float16, device=DEV)
    ops.ln_qkv(x.to(DEV), ws, big[:, :960])
    assert (big[:, 960:] == 7.0).all()
    ref = _ref(x, w, b)
    assert (big[:, :960].float().cpu() - ref).abs().max().item() < 2e-2


def test_ln_qkv_same_row_same_bits_and_equals_the_folded_gemm():
    """the two CFG halves (same rows, 1008 panels apart) agree bitwise; the row-panel GEMM with the LayerNorm fold it replaces
    gives the same projection to fp16 rounding"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_linear, pack_ln_proj
    w, b = _weights(9)
    g = torch.Generator().manual_seed(10)
    Th = 14 * 9216
    x0 = (torch.randn(Th, 320, generator=g) * 1.2).half().to(DEV)
    x = torch.cat([x0, x0])
    ws = pack_ln_proj(w, b).to(DEV)
    out = torch.empty(2 * Th, 960, dtype=torch.float16, device=DEV)
    ops.ln_qkv(x, ws, out)
    assert torch.equal(out[:Th], out[Th:])
    wp = pack_linear(w).to(DEV)
    cs = wp.float().sum(dim=1).contiguous()
    chain = torch.empty_like(out)
    ops.gemm(x, wp, chain, M=2 * Th, N=960, K=320, bias=b.to(DEV), ln=(cs, 1e-5))
    assert (out.float() - chain.float()).abs().max().item() < 2e-2
    assert ((out.float() - chain.float()).norm() / chain.float().norm()).item() < 1e-3


def test_ln_qkv_640_same_row_same_bits_and_equals_layernorm_plus_gemm():
    """the 36x64 level: the two CFG halves agree bitwise; LayerNorm + the 256x320 GEMM it replaces gives the same projection"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_linear, pack_ln_proj
    w, b = _weights(19, 640)
    g = torch.Generator().manual_seed(20)
    Th = 14 * 2304
    x0 = (torch.randn(Th, 640, generator=g) * 1.2).half().to(DEV)
    x = torch.cat([x0, x0])
    ws = pack_ln_proj(w, b).to(DEV)
    out = torch.empty(2 * Th, 1920, dtype=torch.float16, device=DEV)
    ops.ln_qkv(x, ws, out)
    assert torch.equal(out[:Th], out[Th:])
    ln = ops.layernorm(x, None, None, 1e-5)
    chain = torch.empty_like(out)
    ops.gemm(ln, pack_linear(w).to(DEV), chain, M=2 * Th, N=1920, K=640, bias=b.to(DEV))
    assert (out.float() - chain.float()).abs().max().item() < 2e-2
    assert ((out.float() - chain.float()).norm() / chain.float().norm()).item() < 1e-3
