"""The one-launch feed-forward of the 72x128 level (lkgd_ff_fused_c320: LayerNorm + GEGLU + FF-out + residuals, generated
main loop) against fp32 `layer_norm -> linear -> hidden * gelu(gate) -> linear` (+ residual / row bias / AlphaBlender form;
patch/patch.py:551-580, :599-608, :670-680; diffusers FeedForward / GEGLU [EXT]) and against the three launches it replaces.
Tolerance: fp16 outputs, 4e-3 of the output scale + 2e-3 absolute, relative L2 < 3e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, what):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err <= 4e-3 * scale + 2e-3, f"{what}: max abs err {err:.4g} vs scale {scale:.4g}"
    rel = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    assert rel < 3e-3, f"{what}: relative L2 {rel:.4g}"


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(2560, 320, generator=g) / 320 ** 0.5
    b1 = 0.3 * torch.randn(2560, generator=g)
    w2 = torch.randn(320, 1280, generator=g) / 1280 ** 0.5
    b2 = 0.3 * torch.randn(320, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(320, generator=g), 0.2 * torch.randn(320, generator=g)
    return w1, b1, w2, b2, gamma, beta


def _ref(x, w1, b1, w2, b2, gamma, beta, pe=None, frames=1, hw=1, s_acc=1.0, res2=None, r2=0.0):
    xf = x.float()
    if pe is not None:
        idx = (torch.arange(x.shape[0]) // hw) % frames
        xf = xf + pe.float()[idx]
    z = F.layer_norm(xf, (320,), gamma, beta, 1e-5)
    hg = z @ w1.half().float().T + b1
    hid, gate = hg[:, :1280], hg[:, 1280:]
    y = (hid * F.gelu(gate)) @ w2.half().float().T + b2
    out = s_acc * (y + xf)
    if res2 is not None:
        out = out + r2 * res2.float()
    return out


def _pack(w1, b1, w2, gamma, beta):
    from lkgd_amd.packing import pack_ff_fused
    w1f = w1.half().float() * gamma[None, :]           # LayerNorm affine folded into the projection (exact algebra, fp32)
    b1f = w1.half().float() @ beta + b1
    return pack_ff_fused(w1f, b1f, w2).to(DEV)


@pytest.mark.parametrize("T", [128, 32, 1000, 128 * 7 + 5, 128 * 300])
def test_ff_fused_vs_fp32(T):
    from lkgd_amd import ops
    w1, b1, w2, b2, gamma, beta = _weights(T)
    g = torch.Generator().manual_seed(T + 1)
    x = (torch.randn(T, 320, generator=g) * 1.5 + 0.3).half()
    ws = _pack(w1, b1, w2, gamma, beta)
    out = torch.full((T, 320), float("nan"), dtype=torch.float16, device=DEV)
    ops.ff_fused(x.to(DEV), ws, b2.to(DEV), out)
    # the reference multiplies with the FOLDED fp16 weights' values: compare against the unfolded math in fp32
    ref = _ref(x, w1 * 1.0, b1, w2, b2, gamma, beta)
    # (folding rounds W1 * gamma to fp16 once more: inside the tolerance)
    _close(out, ref, f"ff fused T={T}")
    again = torch.empty_like(out)
    ops.ff_fused(x.to(DEV), ws, b2.to(DEV), again)
    assert torch.equal(out, again)


def test_ff_fused_rowbias_and_blend_forms():
    """ff_in: x' = x + pe[frame(row)] inside the LayerNorm and the residual; temporal ff: (1 - a) (FF + x) + a res2"""
    from lkgd_amd import ops
    w1, b1, w2, b2, gamma, beta = _weights(7)
    g = torch.Generator().manual_seed(8)
    Fr, HW = 3, 50
    T = 2 * Fr * HW
    x = torch.randn(T, 320, generator=g).half()
    pe = torch.randn(Fr, 320, generator=g).half()
    res2 = torch.randn(T, 320, generator=g).half()
    ws = _pack(w1, b1, w2, gamma, beta)
    out = torch.empty(T, 320, dtype=torch.float16, device=DEV)
    ops.ff_fused(x.to(DEV), ws, b2.to(DEV), out, rowbias=pe.to(DEV), rowmap=ops.rowmap_div_mod(HW, Fr))
    _close(out, _ref(x, w1, b1, w2, b2, gamma, beta, pe=pe, frames=Fr, hw=HW), "ff_in form")
    ops.ff_fused(x.to(DEV), ws, b2.to(DEV), out, s_acc=0.3, res2=res2.to(DEV), r2=0.7)
    _close(out, _ref(x, w1, b1, w2, b2, gamma, beta, s_acc=0.3, res2=res2, r2=0.7), "AlphaBlender form")


def test_ff_fused_equals_the_three_launch_chain():
    """the same feed-forward through layernorm + GEGLU GEMM + FF-out GEMM (what every other level runs)"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_geglu, pack_linear
    w1, b1, w2, b2, gamma, beta = _weights(11)
    g = torch.Generator().manual_seed(12)
    T = 128 * 40
    x = torch.randn(T, 320, generator=g).half().to(DEV)
    ws = _pack(w1, b1, w2, gamma, beta)
    fused = torch.empty(T, 320, dtype=torch.float16, device=DEV)
    ops.ff_fused(x, ws, b2.to(DEV), fused)
    w1f = (w1.half().float() * gamma[None, :])
    b1f = w1.half().float() @ beta + b1
    wp, bp, half = pack_geglu(w1f.to(DEV), b1f.to(DEV))
    ln = ops.layernorm(x, None, None, 1e-5)
    mid = torch.empty(T, 1280, dtype=torch.float16, device=DEV)
    ops.gemm(ln, wp, mid, M=T, N=2560, K=320, bias=bp, geglu=half)
    chain = torch.empty(T, 320, dtype=torch.float16, device=DEV)
    ops.gemm(mid, pack_linear(w2).to(DEV), chain, M=T, N=320, K=1280, bias=b2.to(DEV), res1=x)
    assert (fused.float() - chain.float()).abs().max().item() < 1.5e-2
    assert ((fused.float() - chain.float()).norm() / chain.float().norm()).item() < 2e-3


def test_ff_fused_same_row_same_bits_wherever_it_sits():
    """A token row gives the same output bits in whichever panel, wave and workgroup turn it is processed (the first panel of
    a workgroup is normalised by other code than the later ones): the two CFG halves of a batch - the same rows 1008 panels
    apart at the 72x128 level - must agree bitwise for test_full_size_loop_properties' guidance identity to hold."""
    from lkgd_amd import ops
    w1, b1, w2, b2, gamma, beta = _weights(21)
    g = torch.Generator().manual_seed(22)
    Fr, HW = 14, 9216
    Th = Fr * HW                                   # 1008 panels per half; a launch has 256 workgroups
    x0 = (torch.randn(Th, 320, generator=g) * 1.5).half().to(DEV)
    r0 = torch.randn(Th, 320, generator=g).half().to(DEV)
    x, res2 = torch.cat([x0, x0]), torch.cat([r0, r0])
    pe = (0.5 * torch.randn(Fr, 320, generator=g)).half().to(DEV)
    ws = _pack(w1, b1, w2, gamma, beta)
    out = torch.empty_like(x)
    for what, kw in (("plain", {}), ("row bias", dict(rowbias=pe, rowmap=ops.rowmap_div_mod(HW, Fr))),
                     ("blend", dict(s_acc=0.3, res2=res2, r2=0.7))):
        out.fill_(float("nan"))
        ops.ff_fused(x, ws, b2.to(DEV), out, **kw)
        assert torch.equal(out[:Th], out[Th:]), what
