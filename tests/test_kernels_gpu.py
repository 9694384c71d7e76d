"""Per-kernel parity: every C-ABI entry point against a plain PyTorch fp32 reference of the same op (CPU), on seeded
inputs.  fp16 tolerance is stated per test.  Runs only on the MI355X box (-m gpu)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _h(x):
    return x.to(torch.float16)


def _close(got, ref, rtol=4e-3, what=""):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err <= rtol * scale + 2e-3, f"{what}: max abs err {err:.4g} vs scale {scale:.4g}"
    rel = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    assert rel < 3e-3, f"{what}: relative L2 {rel:.4g}"


@pytest.fixture(scope="module", params=["tile128", "tile256", "stream", "wide", "rowpanel", "resw", "mid"])
def ops(request):
    """every GEMM/conv test runs against ALL kernel variants (128x128 two-stage, 256x128 three-stage ring, persistent
    streaming kernel with register epilogue, 256x320, row-panel, resident-weight, 128x128 on the four-stage ring; a forced variant falls back to the 128x128
    program on shapes it does not cover)"""
    from lkgd_amd import _lib, ops
    _lib.lib().lkgd_debug_set_gemm_variant({"tile128": 1, "tile256": 2, "stream": 3, "wide": 4, "rowpanel": 5, "resw": 6, "mid": 7}[request.param])
    yield ops
    _lib.lib().lkgd_debug_set_gemm_variant(0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 320, 320), (1000, 64, 1280), (2, 1280, 320), (4032, 960, 640)])
def test_gemm_plain_bias_residual(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    res = _h(torch.randn(M, N, generator=g))
    ref = a.float() @ w.float().T + bias + 0.5 * res.float()
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=bias.to(DEV), res1=res.to(DEV), r1=0.5)
    _close(out, ref, what="gemm plain")


def test_gemm_many_tiles_short_k(ops):
    """> 256 output tiles with K = 64/128 (epilogue every 1-2 K-steps), ragged M and N: stresses the streaming ring"""
    g = torch.Generator().manual_seed(77)
    for M, N, K in ((256 * 70 + 37, 320, 64), (256 * 41 + 200, 704, 128), (256 * 300, 128, 192),
                    (256 * 33 + 5, 960, 320)):
        a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
        bias = torch.randn(N, generator=g)
        res = _h(torch.randn(M, N, generator=g))
        ref = a.float() @ w.float().T + bias + res.float()
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=bias.to(DEV), res1=res.to(DEV))
        _close(out, ref, what=f"gemm many tiles {M}x{N}x{K}")
        out2 = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out2, M=M, N=N, K=K)       # no epilogue loads at all
        _close(out2, a.float() @ w.float().T, what="gemm many tiles, bare")


def test_gemm_rowpanel_shapes(ops):
    """the short-K projection shapes (K = 64..320, several panels per workgroup, ragged M/N, residual + row bias + blend)"""
    g = torch.Generator().manual_seed(55)
    for M, N, K in ((256 * 300 + 77, 320, 320), (256 * 9 + 1, 960, 320), (4100, 200, 256), (70000, 64, 64)):
        a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
        bias = torch.randn(N, generator=g)
        r1, r2 = _h(torch.randn(M, N, generator=g)), _h(torch.randn(M, N, generator=g))
        table = _h(torch.randn(7, N, generator=g))
        idx = (torch.arange(M) // 100) % 7
        ref = 0.7 * (a.float() @ w.float().T + bias + table.float()[idx]) + 0.7 * r1.float() + 0.3 * r2.float()
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=bias.to(DEV), rowbias=table.to(DEV),
                 rowmap=ops.rowmap_div_mod(100, 7), s_acc=0.7, res1=r1.to(DEV), r1=0.7, res2=r2.to(DEV), r2=0.3)
        _close(out, ref, what=f"rowpanel-class gemm {M}x{N}x{K}")


def test_gemm_two_source_and_blend(ops):
    g = torch.Generator().manual_seed(5)
    M, N, K0, K1 = 520, 192, 128, 192
    a0, a1 = _h(torch.randn(M, K0, generator=g)), _h(torch.randn(M, K1, generator=g))
    w = _h(torch.randn(N, K0 + K1, generator=g) / 18)
    bias = torch.randn(N, generator=g)
    r1, r2 = _h(torch.randn(M, N, generator=g)), _h(torch.randn(M, N, generator=g))
    alpha = 0.3
    ref = (1 - alpha) * (torch.cat([a0, a1], 1).float() @ w.float().T + bias) + (1 - alpha) * r1.float() + alpha * r2.float()
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a0.to(DEV), w.to(DEV), out, M=M, N=N, K=K0 + K1, a1=a1.to(DEV), csplit=K0, bias=bias.to(DEV),
             s_acc=1 - alpha, res1=r1.to(DEV), r1=1 - alpha, res2=r2.to(DEV), r2=alpha)
    _close(out, ref, what="gemm 2-source blend")


def test_gemm_rowbias_maps(ops):
    g = torch.Generator().manual_seed(6)
    B, Fr, HW, C = 2, 3, 20, 64
    M = B * Fr * HW
    a, w = _h(torch.randn(M, C, generator=g)), _h(torch.randn(C, C, generator=g) / 8)
    rows = torch.arange(M)
    for name, table_rows, rowmap, idx in [
        ("per-frame-image", B * Fr, ops.rowmap_div(HW), rows // HW),
        ("frame-pos", Fr, ops.rowmap_div_mod(HW, Fr), (rows // HW) % Fr),
        ("per-batch", B, ops.rowmap_div(Fr * HW), rows // (Fr * HW)),
        ("interleaved-0.27", B, (Fr * HW, HW, HW, B), ((rows // (Fr * HW)) * HW + rows % HW) % B),
    ]:
        table = _h(torch.randn(table_rows, C, generator=g))
        ref = a.float() @ w.float().T + table.float()[idx]
        out = torch.empty(M, C, dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=C, K=C, rowbias=table.to(DEV), rowmap=rowmap)
        _close(out, ref, what=name)


def test_gemm_rowbias_long_periods(ops):
    """row maps that are constant over >= 256 rows (time embedding per clip, frame position at HW >= 256): the 256x320
    kernel serves them from two LDS strips per tile - tiles that straddle a boundary, ragged last tile, N = 320 / 640,
    table rows reached through the modulo, a table that is a column slice of a wider one"""
    g = torch.Generator().manual_seed(16)
    for M, N, K, d1, md, c0 in [(700, 320, 64, 300, 1 << 30, 0), (1500, 640, 128, 257, 3, 0), (1024, 320, 64, 256, 2, 1),
                                (900, 320, 320, 4096, 1 << 30, 0)]:
        a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
        rows = torch.arange(M)
        idx = (rows // d1 + c0) % md
        wide = _h(torch.randn(int(idx.max()) + 1, N + 64, generator=g))
        table = wide[:, 32:32 + N]
        bias = torch.randn(N, generator=g)
        ref = a.float() @ w.float().T + bias + table.float()[idx]
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=bias.to(DEV), rowbias=wide.to(DEV)[:, 32:32 + N],
                 rowmap=(d1, 1, 1, md, c0))
        _close(out, ref, what=f"rowbias period {d1} mod {md}")


def test_gemm_geglu(ops):
    from lkgd_amd.packing import pack_geglu
    g = torch.Generator().manual_seed(7)
    M, C = 260, 128
    a = _h(torch.randn(M, C, generator=g))
    w = torch.randn(8 * C, C, generator=g) / C ** 0.5
    b = torch.randn(8 * C, generator=g) * 0.1
    wh = _h(w)
    y = a.float() @ wh.float().T + b
    hid, gate = y.chunk(2, dim=-1)
    ref = hid * F.gelu(gate)
    wp, bp, half = pack_geglu(w, b)
    assert half == 32
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), wp.to(DEV), out, M=M, N=8 * C, K=C, bias=bp.to(DEV), geglu=half)
    _close(out, ref, what="geglu")
    # C = 320: 8C = 2560 is a multiple of 320 -> interleave 80, always served by the 256x320-tile kernel
    M, C = 700, 320
    a = _h(torch.randn(M, C, generator=g))
    w = torch.randn(8 * C, C, generator=g) / C ** 0.5
    b = torch.randn(8 * C, generator=g) * 0.1
    y = a.float() @ _h(w).float().T + b
    hid, gate = y.chunk(2, dim=-1)
    wp32, bp32, _ = pack_geglu(w, b, half=32)
    M2 = 256 * 20 + 9
    a2 = _h(torch.randn(M2, C, generator=g))
    y2 = a2.float() @ _h(w).float().T + b
    h2, g2 = y2.chunk(2, dim=-1)
    out2 = torch.empty(M2, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(a2.to(DEV), wp32.to(DEV), out2, M=M2, N=8 * C, K=C, bias=bp32.to(DEV), geglu=32)
    _close(out2, h2 * F.gelu(g2), what="geglu K=320 (row-panel class)")
    wp, bp, half = pack_geglu(w, b, half=80)
    out = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), wp.to(DEV), out, M=M, N=8 * C, K=C, bias=bp.to(DEV), geglu=half)
    _close(out, hid * F.gelu(gate), what="geglu wide")


def _tokens(x):  # [N,C,H,W] -> [N*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _untokens(t, N, H, W):
    return t.reshape(N, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("stride,ups", [(1, 0), (2, 0), (1, 1)])
def test_conv3x3(ops, stride, ups):
    from lkgd_amd.packing import pack_conv3x3
    g = torch.Generator().manual_seed(10 + stride + ups)
    N, Cin, Cout, H, W = 3, 64, 128, 10, 12
    x = _h(torch.randn(N, Cin, H, W, generator=g))
    w = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    Ho, Wo = ref.shape[-2:]
    out = torch.empty(N * Ho * Wo, Cout, dtype=torch.float16, device=DEV)
    ops.gemm(_tokens(x).to(DEV), pack_conv3x3(w).to(DEV), out, M=N * Ho * Wo, N=Cout, K=9 * Cin, bias=b.to(DEV),
             mode=ops.A_CONV3X3, Cin=Cin, conv=(Ho, Wo, H, W, stride, ups))
    _close(_untokens(out.cpu(), N, Ho, Wo), ref, what=f"conv3x3 s{stride} u{ups}")


def test_conv3x3_concat_temb_residual(ops):
    from lkgd_amd.packing import pack_conv3x3
    g = torch.Generator().manual_seed(14)
    N, C0, C1, Cout, H, W = 4, 64, 128, 64, 8, 8
    x0, x1 = _h(torch.randn(N, C0, H, W, generator=g)), _h(torch.randn(N, C1, H, W, generator=g))
    w = _h(torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 40)
    b = torch.randn(Cout, generator=g)
    temb = _h(torch.randn(N, Cout, generator=g))
    ref = F.conv2d(torch.cat([x0, x1], 1).float(), w.float(), b, padding=1) + temb.float()[:, :, None, None]
    out = torch.empty(N * H * W, Cout, dtype=torch.float16, device=DEV)
    ops.gemm(_tokens(x0).to(DEV), pack_conv3x3(w).to(DEV), out, M=N * H * W, N=Cout, K=9 * (C0 + C1),
             a1=_tokens(x1).to(DEV), csplit=C0, bias=b.to(DEV), mode=ops.A_CONV3X3, Cin=C0 + C1,
             conv=(H, W, H, W, 1, 0), rowbias=temb.to(DEV), rowmap=ops.rowmap_div(H * W))
    _close(_untokens(out.cpu(), N, H, W), ref, what="conv3x3 concat+temb")


def test_conv_in_c8_and_conv_out(ops):
    from lkgd_amd.packing import pack_conv3x3, pack_conv3x3_c8
    g = torch.Generator().manual_seed(15)
    N, H, W = 3, 9, 16
    x = _h(torch.randn(N, 8, H, W, generator=g))
    w = _h(torch.randn(64, 8, 3, 3, generator=g) / 72 ** 0.5)
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    out = torch.empty(N * H * W, 64, dtype=torch.float16, device=DEV)
    ops.gemm(_tokens(x).to(DEV), pack_conv3x3_c8(w).to(DEV), out, M=N * H * W, N=64, K=128, bias=b.to(DEV),
             mode=ops.A_CONV3X3_C8, Cin=8, conv=(H, W, H, W, 1, 0))
    _close(_untokens(out.cpu(), N, H, W), ref, what="conv_in")
    x = _h(torch.randn(N, 64, H, W, generator=g))
    w = _h(torch.randn(4, 64, 3, 3, generator=g) / 24)
    b = torch.randn(4, generator=g)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    out = torch.empty(N * H * W, 4, dtype=torch.float16, device=DEV)
    ops.gemm(_tokens(x).to(DEV), pack_conv3x3(w).to(DEV), out, M=N * H * W, N=4, K=9 * 64, bias=b.to(DEV),
             mode=ops.A_CONV3X3, Cin=64, conv=(H, W, H, W, 1, 0))
    _close(_untokens(out.cpu(), N, H, W), ref, what="conv_out")


def test_temporal_conv(ops):
    from lkgd_amd.packing import pack_tconv3
    g = torch.Generator().manual_seed(16)
    B, Fr, C, H, W = 2, 5, 64, 6, 7
    x5 = _h(torch.randn(B, C, Fr, H, W, generator=g))
    w = _h(torch.randn(C, C, 3, 1, 1, generator=g) / (3 * C) ** 0.5)
    b = torch.randn(C, generator=g)
    ref5 = F.conv3d(x5.float(), w.float(), b, padding=(1, 0, 0))
    tok = x5.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()
    out = torch.empty(B * Fr * H * W, C, dtype=torch.float16, device=DEV)
    ops.gemm(tok.to(DEV), pack_tconv3(w).to(DEV), out, M=B * Fr * H * W, N=C, K=3 * C, bias=b.to(DEV),
             mode=ops.A_TCONV3, Cin=C, tconv=(Fr, H * W))
    got5 = out.cpu().reshape(B, Fr, H, W, C).permute(0, 4, 1, 2, 3)
    _close(got5, ref5, what="temporal conv")


@pytest.mark.parametrize("C0,C1,rows,ns", [(64, 0, 100, 3), (320, 0, 576, 2), (128, 192, 77, 2), (2560, 0, 144, 1),
                                           (640, 1280, 64, 2)])
def test_groupnorm_silu(ops, C0, C1, rows, ns):
    g = torch.Generator().manual_seed(C0 + C1 + rows)
    C = C0 + C1
    x = _h(torch.randn(ns * rows, C, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.silu(F.group_norm(x.float().reshape(ns, rows, C).permute(0, 2, 1), 32, gamma, beta, 1e-5))
    ref = ref.permute(0, 2, 1).reshape(ns * rows, C)
    xd = x.to(DEV)
    x0, x1 = (xd[:, :C0], xd[:, C0:]) if C1 else (xd, None)
    out = ops.groupnorm_silu(x0, x1, ns, rows, gamma.to(DEV), beta.to(DEV), 1e-5)
    _close(out, ref, what="groupnorm+silu")


@pytest.mark.parametrize("C0,C1,rows,ns", [(320, 0, 9216, 4), (1280, 0, 576, 4), (1280, 1280, 144, 4), (640, 0, 4 * 2304, 1),
                                           (320, 0, 14 * 9216, 2), (128, 192, 77, 2)])
def test_groupnorm_one_call_equals_three_launches_bitwise(C0, C1, rows, ns):
    """lkgd_groupnorm_silu against lkgd_groupnorm_stats + lkgd_groupnorm_apply on the maps of a frame-sharded rank (chunks cut
    down to 8 KiB so that small maps still fill the CUs) and of the full forward: the same bits, within tolerance of F.group_norm"""
    from lkgd_amd import _lib, ops
    g = torch.Generator().manual_seed(C0 + C1 + rows)
    C = C0 + C1
    x = _h(torch.randn(ns * rows, C, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    xd = x.to(DEV)
    x0, x1 = (xd[:, :C0], xd[:, C0:]) if C1 else (xd, None)
    _lib.lib().lkgd_debug_set_gn_small(0)          # (small samples take ONE launch by default: the test below)
    try:
        one = ops.groupnorm_silu(x0, x1, ns, rows, gamma, beta, 1e-5)
    finally:
        _lib.lib().lkgd_debug_set_gn_small(1)
    stats = ops.groupnorm_stats(x0, x1, ns, rows, 1e-5)
    three = ops.groupnorm_apply(x0, x1, ns, rows, stats, gamma, beta, True, torch.empty_like(one))
    assert torch.equal(one, three)
    sub = slice(0, min(ns * rows, 20000))
    ref = F.silu(F.group_norm(x.float().reshape(ns, rows, C).permute(0, 2, 1), 32, gamma.cpu(), beta.cpu(), 1e-5))
    _close(one[sub], ref.permute(0, 2, 1).reshape(ns * rows, C)[sub], what="one-call groupnorm")


@pytest.mark.parametrize("C0,C1,rows,ns,silu", [(1280, 0, 576, 4, True), (1280, 640, 384, 3, True), (640, 0, 1152, 4, True),
                                                (320, 0, 144, 2, False), (1280, 1280, 144, 28, True), (640, 320, 100, 2, True),
                                                (128, 192, 77, 2, True)])
def test_groupnorm_one_launch_for_small_samples(C0, C1, rows, ns, silu):
    """gn_small_kernel (one workgroup per (sample, group), the group's values held in LDS): every access width (C/32 = 40 / 80:
    16 bytes, 20 / 60: 8, 10 / 30: 4), groups that straddle the two sources of a concatenated input, groups of 45 KiB - against the three launches (same arithmetic, another summation order of the statistics) and F.group_norm;
    its (mean, rstd) output; a strided destination"""
    from lkgd_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(C0 + 3 * C1 + rows)
    C = C0 + C1
    x = _h(torch.randn(ns * rows, C, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    xd = x.to(DEV)
    x0, x1 = (xd[:, :C0], xd[:, C0:]) if C1 else (xd, None)
    wide = torch.full((ns * rows, C + 64), -7.0, dtype=torch.float16, device=DEV)
    L.lkgd_debug_set_gn_small_limits(1 << 40)      # (the size rule would send the larger tensors to the three launches)
    try:
        one = ops.groupnorm_silu(x0, x1, ns, rows, gamma, beta, 1e-5, silu=silu, out=wide[:, 32:32 + C])
    finally:
        L.lkgd_debug_set_gn_small_limits(10 * 1024 * 1024 + 512 * 1024)
    L.lkgd_debug_set_gn_small(0)
    try:
        three = ops.groupnorm_silu(x0, x1, ns, rows, gamma, beta, 1e-5, silu=silu)
    finally:
        L.lkgd_debug_set_gn_small(1)
    assert (wide[:, :32] == -7).all() and (wide[:, 32 + C:] == -7).all()
    d = (one.float() - three.float()).abs()
    assert d.max().item() <= 4e-3 * three.float().abs().max().item() + 2e-3 and (d > 0).float().mean().item() < 0.05
    ref = F.group_norm(x.float().reshape(ns, rows, C).permute(0, 2, 1), 32, gamma.cpu(), beta.cpu(), 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1).reshape(ns * rows, C)
    _close(one, ref, what="one-launch groupnorm")


def test_groupnorm_apply_segments_equals_apply_bitwise():
    """one launch over a table of row segments (a frame-sharded rank: own frames of two entries + four boundary frames of other
    sizes) = lkgd_groupnorm_apply segment by segment, bit for bit"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(5)
    for C, HW, F in ((1280, 144, 3), (320, 2304, 2), (640, 100, 4)):
        x = _h(torch.randn(2 * F * HW, C, generator=g) * 1.5 + 0.2).to(DEV)
        halo = _h(torch.randn(4 * HW, C, generator=g)).to(DEV)
        gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        stats = ops.groupnorm_stats(x, None, 2, F * HW, 1e-5)
        out = torch.zeros(2 * (F + 2) * HW, C, dtype=torch.float16, device=DEV)
        want = torch.zeros_like(out)
        segs = []
        for b in range(2):
            blk = (F + 2) * HW
            trio = [(x[b * F * HW:(b + 1) * F * HW], slice(b * blk + HW, (b + 1) * blk - HW)),
                    (halo[2 * b * HW:(2 * b + 1) * HW], slice(b * blk, b * blk + HW)),
                    (halo[(2 * b + 1) * HW:(2 * b + 2) * HW], slice((b + 1) * blk - HW, (b + 1) * blk))]
            for src, dst in trio:
                segs.append((src, out[dst], b))
                ops.groupnorm_apply(src, None, 1, src.shape[0], stats[b:b + 1], gamma, beta, True, want[dst])
        ops.groupnorm_apply_segments(segs, stats, gamma, beta, True)
        assert torch.equal(out, want) and out.float().abs().sum().item() > 0


def test_layernorm_with_rowbias(ops):
    g = torch.Generator().manual_seed(20)
    for C in (64, 320, 640, 1280):
        T, Fr, HW = 2 * 3 * 10, 3, 10
        x = _h(torch.randn(T, C, generator=g) * 3 + 1)
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
        out = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5)
        _close(out, ref, what=f"layernorm C={C}")
        emb = _h(torch.randn(Fr, C, generator=g))
        idx = (torch.arange(T) // HW) % Fr
        ref = F.layer_norm((x + emb[idx]).float(), (C,), gamma, beta, 1e-5)
        out = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, rowbias=emb.to(DEV),
                            rowmap=ops.rowmap_div_mod(HW, Fr))
        _close(out, ref, what=f"layernorm+emb C={C}")


@pytest.mark.parametrize("S,heads,nb", [(16, 1, 2), (64, 2, 3), (144, 2, 2), (576, 3, 2), (1024, 2, 2), (2304, 1, 1)])
def test_attn_spatial(ops, S, heads, nb):
    g = torch.Generator().manual_seed(S + heads)
    C = heads * 64
    qkv = _h(torch.randn(nb * S, 3 * C, generator=g))
    q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(nb, S, heads, 64).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(nb * S, C)
    d = qkv.to(DEV)
    out = torch.empty(nb * S, C, dtype=torch.float16, device=DEV)
    ops.attn_spatial(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], out, nb, S, heads)
    _close(out, ref, what="attn spatial")
    if nb >= 2:   # joint attention: K/V of the partner batch entry (patch/patch.py:466-468)
        perm = torch.arange(nb).flip(0)
        ref = F.scaled_dot_product_attention(q, k[perm], v[perm]).transpose(1, 2).reshape(nb * S, C)
        ops.attn_spatial(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], out, nb, S, heads,
                         kv_batch_map=perm.to(torch.int32).to(DEV))
        _close(out, ref, what="attn spatial kv-map")


def test_attn_spatial_large_scores(ops):
    # forces the online-softmax rescale path: one key dominates late in the sequence
    g = torch.Generator().manual_seed(99)
    S, C = 320, 64
    qkv = torch.randn(S, 3 * C, generator=g)
    qkv[200, C:2 * C] = qkv[5, :C] * 6.0   # key 200 aligned with query 5
    qkv = _h(qkv)
    q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(1, S, 1, 64).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(S, C)
    d = qkv.to(DEV)
    out = torch.empty(S, C, dtype=torch.float16, device=DEV)
    ops.attn_spatial(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], out, 1, S, 1)
    _close(out, ref, what="attn spatial rescale")


@pytest.mark.parametrize("B,Fr,S,heads", [(2, 14, 40, 2), (1, 4, 17, 1), (2, 25, 9, 1), (2, 3, 64, 5)])
def test_attn_temporal(ops, B, Fr, S, heads):
    g = torch.Generator().manual_seed(B + Fr + S)
    C = heads * 64
    qkv = _h(torch.randn(B * Fr * S, 3 * C, generator=g))

    def split(i):
        x = qkv[:, i * C:(i + 1) * C].float().reshape(B, Fr, S, heads, 64)
        return x.permute(0, 2, 3, 1, 4).reshape(B * S, heads, Fr, 64)
    ref = F.scaled_dot_product_attention(split(0), split(1), split(2))
    ref = ref.reshape(B, S, heads, Fr, 64).permute(0, 3, 1, 2, 4).reshape(B * Fr * S, C)
    d = qkv.to(DEV)
    out = torch.empty(B * Fr * S, C, dtype=torch.float16, device=DEV)
    ops.attn_temporal(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], out, B, Fr, S, heads)
    _close(out, ref, what="attn temporal")


def test_loop_glue_against_oracle_scheduler(ops):
    from oracle.scheduler import EulerDiscreteOracle
    g = torch.Generator().manual_seed(30)
    B, Fr, H, W = 1, 4, 6, 5
    sch = EulerDiscreteOracle()
    sch.set_timesteps(5)
    lat = _h(torch.randn(B, Fr, 4, H, W, generator=g) * float(sch.init_noise_sigma))
    img = _h(torch.randn(2 * B, Fr, 4, H, W, generator=g))
    t = sch.timesteps[1]
    sch._step_index = 1
    sigma, sigma_next = float(sch.sigmas[1]), float(sch.sigmas[2])
    # reference semantics in fp16 tensors (pipeline :549-553)
    x = torch.cat([lat] * 2)
    x = (x / ((sch.sigmas[1] ** 2 + 1) ** 0.5)).to(torch.float16)
    x = torch.cat([x, img], dim=2)
    ref_tok = x.permute(0, 1, 3, 4, 2).reshape(-1, 8)
    tok = ops.prepare_unet_input(lat.to(DEV), img.to(DEV), 2, sigma)
    assert torch.equal(tok.cpu(), ref_tok), "prepare_unet_input must be bit-exact"
    noise = _h(torch.randn(2 * B, Fr, 4, H, W, generator=g))
    gs = torch.linspace(1.0, 3.0, Fr)
    u, c = noise.chunk(2)
    n = u + gs.to(torch.float16)[None, :, None, None, None] * (c - u)
    ref = sch.step(n, t, lat)
    latd = lat.to(DEV).clone()
    ops.cfg_euler_step(noise.permute(0, 1, 3, 4, 2).reshape(-1, 4).contiguous().to(DEV), latd, gs.to(DEV), 2, sigma,
                       sigma_next)
    err = (latd.cpu().float() - ref.float()).abs().max().item()
    assert err <= 2 ** -10 * ref.abs().max().item() + 1e-3, err   # <= 1 fp16 ulp of the largest latent
    back = ops.tokens_to_nchw(tok, 2 * B * Fr, 8, H, W)
    assert torch.equal(back.cpu(), x.reshape(2 * B * Fr, 8, H, W))
    assert torch.equal(ops.nchw_to_tokens(back).cpu(), ref_tok)


def test_embedding_helpers(ops):
    from oracle.blocks import Timesteps
    t = torch.tensor([1.6378, -1.5537, 6.0, 127.0, 0.02])
    ref = Timesteps(320, True, 0)(t)
    got = ops.timestep_embedding(t.to(DEV), 320)
    assert (got.cpu().float() - ref).abs().max() < 2e-3
    x = _h(torch.randn(1003))
    assert (ops.silu(x.to(DEV)[:1000]).cpu().float() - F.silu(x[:1000].float())).abs().max() < 2e-3


def test_errors_are_loud(ops):
    from lkgd_amd import LkgdHipError
    a = torch.zeros(128, 100, dtype=torch.float16, device=DEV)   # K not a multiple of 64
    w = torch.zeros(128, 100, dtype=torch.float16, device=DEV)
    out = torch.zeros(128, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(LkgdHipError):
        ops.gemm(a, w, out, M=128, N=128, K=100)
    with pytest.raises(LkgdHipError):
        ops.gemm(a.cpu(), w, out, M=128, N=128, K=64)


# ------------------------------------------------------------------------------------------------ FSM row kernel (a15)
@pytest.mark.parametrize("C_", [64, 320, 1280])
def test_fsm_rows_copy_and_scatter_mean(C_):
    """lkgd_fsm_rows: copy / combine mode is bit-exact; scatter-mean == sequential fp32 scatter_add / (count + 1e-6)"""
    from lkgd_amd import ops
    from lkgd_amd.patch_FSM import _csr
    g = torch.Generator().manual_seed(C_)
    pairs, HW, P = 3, 40, 90
    hA = torch.randn(2 * pairs * HW, C_, generator=g).half()
    res = torch.randn(2 * pairs * HW, C_, generator=g).half()
    bias = torch.randn(2, C_ + 8, generator=g).half()[:, :C_]          # strided bias rows; entry -> row entry // 3
    d = lambda t: t.to(DEV)
    # de-interleave copy of the odd entries
    out = torch.zeros(pairs * HW, C_, dtype=torch.float16, device=DEV)
    ops.fsm_rows(d(hA), out, pairs=pairs, HW=HW, C_=C_, a_rows=(2 * HW, HW), o_rows=(HW, 0))
    assert torch.equal(out.cpu(), hA.reshape(pairs, 2, HW, C_)[:, 1].reshape(-1, C_))
    # combine: out[even] = a + res[even] + bias
    a = torch.randn(pairs * HW, 2 * C_, generator=g).half()
    out2 = torch.zeros(2 * pairs * HW, C_, dtype=torch.float16, device=DEV)
    bdev = d(torch.randn(2, C_ + 8, generator=g).half())[:, :C_]
    ops.fsm_rows(d(a)[:, C_:], out2, pairs=pairs, HW=HW, C_=C_, a_rows=(HW, 0), o_rows=(2 * HW, 0), res=d(res),
                 r_rows=(2 * HW, 0), bias=bdev, bias_map=(2, 0, 3))
    want = (a[:, C_:].float().reshape(pairs, HW, C_) + res.float().reshape(pairs, 2, HW, C_)[:, 0]
            + bdev.cpu().float()[(2 * torch.arange(pairs)) // 3][:, None]).half()
    got = out2.cpu().reshape(pairs, 2, HW, C_)
    assert torch.equal(got[:, 0], want)
    assert torch.equal(got[:, 1], torch.zeros_like(got[:, 1]))          # odd entries untouched
    # scatter-mean with collisions, empty cells and invisible points
    tgt = torch.randint(0, HW // 2, (pairs, P), generator=g)             # upper half of the cells stays empty
    src = torch.randint(0, HW, (pairs, P), generator=g)
    vis = (torch.rand(pairs, P, generator=g) > 0.3).float()
    off, pt = _csr(d(tgt), pairs, HW)
    out3 = torch.full((pairs * HW, C_), 7.0, dtype=torch.float16, device=DEV)
    ops.fsm_rows(d(hA), out3, pairs=pairs, HW=HW, C_=C_, a_rows=(2 * HW, HW), o_rows=(HW, 0),
                 csr=(off, pt, d(src.reshape(-1).int()), d(vis.reshape(-1))))
    feats = hA.float().reshape(pairs, 2, HW, C_)[:, 1]
    canvas = torch.zeros(pairs, HW, C_)
    cnt = torch.zeros(pairs, HW, 1)
    for p_ in range(pairs):
        for k in range(P):                                               # sequential order of torch.scatter_add on CPU
            if vis[p_, k] != 0:
                canvas[p_, tgt[p_, k]] += feats[p_, src[p_, k]]
            cnt[p_, tgt[p_, k]] += vis[p_, k]
    want3 = (canvas / (cnt + 1e-6)).half().reshape(-1, C_)
    assert torch.equal(out3.cpu(), want3)
    assert float(out3.cpu().reshape(pairs, HW, C_)[:, HW // 2:].abs().max()) == 0.0


def test_fsm_rows_rejects_bad_descriptors():
    from lkgd_amd import ops
    from lkgd_amd._lib import LkgdHipError
    a = torch.zeros(16, 64, dtype=torch.float16, device=DEV)
    with pytest.raises(LkgdHipError):
        ops.fsm_rows(a[:, :60], a, pairs=1, HW=8, C_=60, a_rows=(8, 0), o_rows=(8, 0))       # C % 8
    with pytest.raises(LkgdHipError):
        ops.fsm_rows(a, a, pairs=0, HW=8, C_=64, a_rows=(8, 0), o_rows=(8, 0))
    off = torch.zeros(9, dtype=torch.int32, device=DEV)
    with pytest.raises(LkgdHipError):
        ops.fsm_rows(a, a, pairs=1, HW=8, C_=64, a_rows=(8, 0), o_rows=(8, 0),
                     csr=(off, off[:4], off[:3], torch.zeros(4, device=DEV)))


def test_gemm_split_k_few_rows():
    """few-row problems (a frame-sharded rank's 9x16 level: 576 rows) cut K into slices with fp32 partial tiles in the
    caller's workspace and a deterministic second pass; same result as the unsplit kernel up to fp32 summation order"""
    from lkgd_amd import _lib, ops
    from lkgd_amd.packing import pack_conv3x3, pack_geglu
    L = _lib.lib()
    g = torch.Generator().manual_seed(21)
    # 3x3 conv, two sources, time-embedding row bias, residual: M = 4 * 12 * 12 = 576, K = 9 * 256
    N, C0, C1, Cout, H, W = 4, 128, 128, 192, 12, 12
    x0, x1 = _h(torch.randn(N, C0, H, W, generator=g)), _h(torch.randn(N, C1, H, W, generator=g))
    w = _h(torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 48)
    b = torch.randn(Cout, generator=g)
    temb = _h(torch.randn(N, Cout, generator=g))
    res = _h(torch.randn(N * H * W, Cout, generator=g))
    ref = F.conv2d(torch.cat([x0, x1], 1).float(), w.float(), b, padding=1) + temb.float()[:, :, None, None]
    ref = 0.75 * ref + 0.5 * _untokens(res.float(), N, H, W)
    outs = []
    for on in (1, 0):
        L.lkgd_debug_set_gemm_splitk(on)
        out = torch.empty(N * H * W, Cout, dtype=torch.float16, device=DEV)
        ops.gemm(_tokens(x0).to(DEV), pack_conv3x3(w).to(DEV), out, M=N * H * W, N=Cout, K=9 * (C0 + C1),
                 a1=_tokens(x1).to(DEV), csplit=C0, bias=b.to(DEV), mode=ops.A_CONV3X3, Cin=C0 + C1,
                 conv=(H, W, H, W, 1, 0), rowbias=temb.to(DEV), rowmap=ops.rowmap_div(H * W), res1=res.to(DEV),
                 s_acc=0.75, r1=0.5)
        outs.append(out.float().cpu())
    L.lkgd_debug_set_gemm_splitk(1)
    _close(_untokens(outs[0], N, H, W), ref, what="split-K conv3x3")
    assert (outs[0] - outs[1]).abs().max().item() <= 4e-3 * ref.abs().max().item()
    # run-to-run determinism of the two-pass reduction
    out2 = torch.empty(N * H * W, Cout, dtype=torch.float16, device=DEV)
    ops.gemm(_tokens(x0).to(DEV), pack_conv3x3(w).to(DEV), out2, M=N * H * W, N=Cout, K=9 * (C0 + C1),
             a1=_tokens(x1).to(DEV), csplit=C0, bias=b.to(DEV), mode=ops.A_CONV3X3, Cin=C0 + C1,
             conv=(H, W, H, W, 1, 0), rowbias=temb.to(DEV), rowmap=ops.rowmap_div(H * W), res1=res.to(DEV),
             s_acc=0.75, r1=0.5)
    assert torch.equal(out2.float().cpu(), outs[0])
    # GEGLU (32 | 32 interleave) with ragged rows: M = 301, K = 1280
    M, C = 301, 1280
    a = _h(torch.randn(M, C, generator=g))
    wg = torch.randn(512, C, generator=g) / C ** 0.5
    bg = torch.randn(512, generator=g) * 0.1
    y = a.float() @ _h(wg).float().T + bg
    hid, gate = y.chunk(2, dim=-1)
    wp, bp, half = pack_geglu(wg, bg, half=32)
    out = torch.empty(M, 256, dtype=torch.float16, device=DEV)
    ops.gemm(a.to(DEV), wp.to(DEV), out, M=M, N=512, K=C, bias=bp.to(DEV), geglu=half)
    _close(out, hid * F.gelu(gate), what="split-K geglu")


def test_gemm_four_stage_ring_k_slices():
    """the 128x128 program on the four-stage ring (few-row problems, round 5) over forced K slices: slices that do not divide
    K evenly, a slice shorter than the ring, ragged rows; against fp32 and against its own unsplit run, bitwise repeatable"""
    from lkgd_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(77)
    L.lkgd_debug_set_gemm_variant(7)
    try:
        for M, N, K, forced in ((576, 1280, 1280, 3), (301, 320, 704, 5), (2304, 640, 1920, 2), (130, 128, 64 * 11, 4)):
            a = _h(torch.randn(M, K, generator=g))
            w = _h(torch.randn(N, K, generator=g) / K ** 0.5)
            b = torch.randn(N, generator=g)
            res = _h(torch.randn(M, N, generator=g))
            ref = a.float() @ w.float().T + b + 0.5 * res.float()
            outs = []
            for f in (forced, 0, forced):
                L.lkgd_debug_set_mid_model(0.0, 0.0, 0.0, 0.0, f)
                L.lkgd_debug_set_gemm_splitk(1 if f else 0)
                out = torch.empty(M, N, dtype=torch.float16, device=DEV)
                ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=b.to(DEV), res1=res.to(DEV), r1=0.5)
                outs.append(out.cpu())
            _close(outs[0], ref, what=f"four-stage ring, {forced} K slices, {M}x{N}x{K}")
            _close(outs[1], ref, what=f"four-stage ring, unsplit, {M}x{N}x{K}")
            assert torch.equal(outs[0], outs[2])
    finally:
        L.lkgd_debug_set_mid_model(0.0, 0.0, 0.0, 0.0, 0)
        L.lkgd_debug_set_gemm_splitk(1)
        L.lkgd_debug_set_gemm_variant(0)


def test_gemm_split_k_on_256x320_tiles():
    """few-row, deep-K problems on the 256x320 kernel: equal K slices as virtual tiles + the shared reduce pass (the 18x32 /
    9x16 levels of the model at 4032 / 2304 rows).  Same result as the unsplit kernel up to fp32 summation order; a slice may
    start in the middle of a (tap, source) segment of the implicit convolution.  The slice count is forced
    (lkgd_debug_set_wide_ksplit) so that the case does not depend on the dispatcher's fill rule, which has its own checks below."""
    from lkgd_amd import _lib, ops
    from lkgd_amd.packing import pack_conv3x3
    L = _lib.lib()
    g = torch.Generator().manual_seed(23)
    # plain: 12000 x 640 x 2048 -> 94 tiles, two slices of 16 K-tiles (ops.gemm hands a workspace to problems below 12288 rows)
    M, N, K = 12000, 640, 2048
    a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    res = _h(torch.randn(M, N, generator=g))
    rows = torch.arange(0, M, 53)
    ref = a[rows].float() @ w.float().T + b + 0.5 * res[rows].float()
    outs = []
    L.lkgd_debug_set_gemm_variant(4)
    for ks in (2, 0):
        L.lkgd_debug_set_wide_ksplit(ks)
        L.lkgd_debug_set_gemm_splitk(1 if ks else 0)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=b.to(DEV), res1=res.to(DEV), r1=0.5)
        outs.append(out.cpu())
    L.lkgd_debug_set_gemm_splitk(1)
    _close(outs[0][rows], ref, what="256x320 split-K plain")
    assert (outs[0].float() - outs[1].float()).abs().max().item() <= 4e-3 * ref.abs().max().item()
    assert not torch.equal(outs[0], outs[1])        # another fp32 summation order: the sliced path really ran
    # 3x3 conv, two sources (the slice boundary falls inside a tap): 3 images of 64x64, 128 + 128 channels -> K = 2304
    Nimg, C0, C1, Cout, H, W = 3, 128, 128, 320, 64, 64
    x0, x1 = _h(torch.randn(Nimg, C0, H, W, generator=g)), _h(torch.randn(Nimg, C1, H, W, generator=g))
    wc = _h(torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 48)
    bc = torch.randn(Cout, generator=g)
    refc = F.conv2d(torch.cat([x0, x1], 1).float(), wc.float(), bc, padding=1)
    outs = []
    for ks in (2, 0):
        L.lkgd_debug_set_wide_ksplit(ks)
        L.lkgd_debug_set_gemm_splitk(1 if ks else 0)
        out = torch.empty(Nimg * H * W, Cout, dtype=torch.float16, device=DEV)
        ops.gemm(_tokens(x0).to(DEV), pack_conv3x3(wc).to(DEV), out, M=Nimg * H * W, N=Cout, K=9 * (C0 + C1),
                 a1=_tokens(x1).to(DEV), csplit=C0, bias=bc.to(DEV), mode=ops.A_CONV3X3, Cin=C0 + C1,
                 conv=(H, W, H, W, 1, 0))
        outs.append(out.cpu())
    L.lkgd_debug_set_gemm_splitk(1)
    L.lkgd_debug_set_wide_ksplit(0)
    L.lkgd_debug_set_gemm_variant(0)
    _close(_untokens(outs[0], Nimg, H, W), refc, what="256x320 split-K conv3x3")
    assert (outs[0].float() - outs[1].float()).abs().max().item() <= 4e-3 * refc.abs().max().item()
    # the dispatcher's own choices (fill rule of wide_split): 4032 x 1280 x 3840 (64 tiles -> 4 slices), 8064 rows (128 tiles
    # -> 2 slices), 2304 rows (36 tiles -> 6 slices of 10 K-tiles): against fp32 on sampled rows
    for M, N, K in ((4032, 1280, 3840), (8064, 1280, 3840), (2304, 1280, 3840)):
        a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
        b = torch.randn(N, generator=g)
        rows = torch.arange(0, M, 37)
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=b.to(DEV))
        _close(out.cpu()[rows], a[rows].float() @ w.float().T + b, what=f"auto split {M}x{N}x{K}")


@pytest.mark.parametrize("tile_rows", [256, 192])
def test_gemm_wide_rows_through_lds_match_direct_stores(tile_rows):
    """(both tile-row forms of the program: 256 rows, and 192 = wave tile 48 rows)
    the 256x320 kernel's two output paths (8-byte stores straight from the accumulator layout; whole 320-byte row
    segments staged through LDS - the default for plain linears without GEGLU): bit-identical, on ragged M, a column-slice
    output (ldc > N), residual + row bias; rows past M and columns outside the slice stay untouched.  GEGLU and the
    convolutions keep the direct path under either setting."""
    from lkgd_amd import _lib, ops
    from lkgd_amd.packing import pack_conv3x3, pack_geglu
    L = _lib.lib()
    g = torch.Generator().manual_seed(29)

    def both(fn, shape, cols=None):
        outs = []
        for on in (0, 1):
            L.lkgd_debug_set_wide_lds_out(on)
            buf = torch.full(shape, -7.0, dtype=torch.float16, device=DEV)
            fn(buf if cols is None else buf[:, cols[0]:cols[1]])
            outs.append(buf.cpu())
        L.lkgd_debug_set_wide_lds_out(-1)
        assert torch.equal(outs[0], outs[1])
        return outs[1]

    L.lkgd_debug_set_gemm_variant(4)
    L.lkgd_debug_set_wide_tile_m(tile_rows)
    try:
        # plain, ragged M, residual + row bias, output = columns [64, 64+640) of a 768-wide buffer
        M, N, K = 256 * 5 + 77, 640, 320
        a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
        b = torch.randn(N, generator=g)
        res = _h(torch.randn(M, N, generator=g))
        table = _h(torch.randn(5, N, generator=g))
        idx = (torch.arange(M) // 300) % 5
        out = both(lambda o: ops.gemm(a.to(DEV), w.to(DEV), o, M=M, N=N, K=K, bias=b.to(DEV), res1=res.to(DEV), r1=0.5,
                                      rowbias=table.to(DEV), rowmap=(300, 1, 1, 5, 0)), (M + 3, 768), cols=(64, 64 + N))
        _close(out[:M, 64:64 + N], a.float() @ w.float().T + b + table.float()[idx] + 0.5 * res.float(), what="wide via LDS")
        assert (out[M:] == -7).all() and (out[:, :64] == -7).all() and (out[:, 64 + N:] == -7).all()
        # GEGLU at C = 320 (interleave 80): output 1280 columns
        M, C = 256 * 3 + 130, 320
        a = _h(torch.randn(M, C, generator=g))
        w = torch.randn(8 * C, C, generator=g) / C ** 0.5
        b = torch.randn(8 * C, generator=g) * 0.1
        wp, bp, half = pack_geglu(w, b, half=80)
        out = both(lambda o: ops.gemm(a.to(DEV), wp.to(DEV), o, M=M, N=8 * C, K=C, bias=bp.to(DEV), geglu=half), (M + 1, 4 * C))
        hid, gate = (a.float() @ _h(w).float().T + b).chunk(2, dim=-1)
        _close(out[:M], hid * F.gelu(gate), what="wide geglu via LDS")
        assert (out[M:] == -7).all()
        # 3x3 conv 64 -> 320 on 3 images of 40x24 (2880 rows: ragged last tile), temporal conv 320 -> 320
        Nimg, Cin, Cout, H, W = 3, 64, 320, 40, 24
        x = _h(torch.randn(Nimg, Cin, H, W, generator=g))
        wc = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / 24)
        bc = torch.randn(Cout, generator=g)
        out = both(lambda o: ops.gemm(_tokens(x).to(DEV), pack_conv3x3(wc).to(DEV), o, M=Nimg * H * W, N=Cout, K=9 * Cin,
                                      bias=bc.to(DEV), mode=ops.A_CONV3X3, Cin=Cin, conv=(H, W, H, W, 1, 0)),
                   (Nimg * H * W, Cout))
        _close(_untokens(out, Nimg, H, W), F.conv2d(x.float(), wc.float(), bc, padding=1), what="wide conv via LDS")
        from lkgd_amd.packing import pack_tconv3
        Bc, Fr, C, Ht, Wt = 1, 5, 320, 23, 10
        x5 = _h(torch.randn(Bc, C, Fr, Ht, Wt, generator=g))
        wt = _h(torch.randn(C, C, 3, 1, 1, generator=g) / (3 * C) ** 0.5)
        bt = torch.randn(C, generator=g)
        tok = x5.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()
        out = both(lambda o: ops.gemm(tok.to(DEV), pack_tconv3(wt).to(DEV), o, M=Bc * Fr * Ht * Wt, N=C, K=3 * C, bias=bt.to(DEV),
                                      mode=ops.A_TCONV3, Cin=C, tconv=(Fr, Ht * Wt)), (Bc * Fr * Ht * Wt, C))
        reft = F.conv3d(x5.float(), wt.float(), bt, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(-1, C)
        _close(out, reft, what="wide tconv via LDS")
    finally:
        L.lkgd_debug_set_gemm_variant(0)
        L.lkgd_debug_set_wide_lds_out(-1)
        L.lkgd_debug_set_wide_tile_m(0)


@pytest.mark.parametrize("tile_rows", [256, 192])
def test_gemm_wide_256_column_tiles(tile_rows):
    """(with 256- and 192-row tiles)
    the 256x256 form of the 256x320 program (wave tile 64 x 128: channel counts 256 / 512 / 768 - the VAE decoder's
    widths): plain linear with bias + residual + both row-bias paths on ragged M and a column-slice output, 3x3 conv
    (stride 1, upsampled input), temporal conv; both output paths bit-identical; the GroupNorm sums of its epilogue"""
    from lkgd_amd import _lib, ops
    from lkgd_amd.packing import pack_conv3x3, pack_tconv3
    L = _lib.lib()
    assert L.lkgd_gemm_wide_tile_n(256) == 256 and L.lkgd_gemm_wide_tile_n(512) == 256 and L.lkgd_gemm_wide_tile_n(768) == 256
    assert L.lkgd_gemm_wide_tile_n(640) == 320 and L.lkgd_gemm_wide_tile_n(3072) == 320 and L.lkgd_gemm_wide_tile_n(128) == 320
    g = torch.Generator().manual_seed(256)

    def both(fn, shape, cols=None):
        outs = []
        for on in (0, 1):
            L.lkgd_debug_set_wide_lds_out(on)
            buf = torch.full(shape, -7.0, dtype=torch.float16, device=DEV)
            fn(buf if cols is None else buf[:, cols[0]:cols[1]])
            outs.append(buf.cpu())
        L.lkgd_debug_set_wide_lds_out(-1)
        assert torch.equal(outs[0], outs[1])
        return outs[1]

    L.lkgd_debug_set_gemm_variant(4)
    L.lkgd_debug_set_wide_tile_m(tile_rows)
    try:
        for M, N, K, d1 in ((256 * 5 + 77, 512, 320, 300), (256 * 9, 256, 1024, 40), (700, 768, 128, 256)):
            a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
            b = torch.randn(N, generator=g)
            res = _h(torch.randn(M, N, generator=g))
            table = _h(torch.randn(5, N, generator=g))
            idx = (torch.arange(M) // d1) % 5
            out = both(lambda o: ops.gemm(a.to(DEV), w.to(DEV), o, M=M, N=N, K=K, bias=b.to(DEV), res1=res.to(DEV), r1=0.5,
                                          rowbias=table.to(DEV), rowmap=(d1, 1, 1, 5, 0)), (M + 3, N + 128), cols=(64, 64 + N))
            _close(out[:M, 64:64 + N], a.float() @ w.float().T + b + table.float()[idx] + 0.5 * res.float(),
                   what=f"256-column tiles {M}x{N}x{K}")
            assert (out[M:] == -7).all() and (out[:, :64] == -7).all() and (out[:, 64 + N:] == -7).all()
        # 3x3 conv 128 -> 256 on 3 images of 40x24 (ragged last tile) and 64 -> 512 on a 2x-upsampled 20x12 input
        for Cin, Cout, H, W, ups in ((128, 256, 40, 24, 0), (64, 512, 20, 12, 1)):
            Nimg = 3
            x = _h(torch.randn(Nimg, Cin, H, W, generator=g))
            wc = _h(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)
            bc = torch.randn(Cout, generator=g)
            Ho, Wo = H << ups, W << ups
            out = both(lambda o: ops.gemm(_tokens(x).to(DEV), pack_conv3x3(wc).to(DEV), o, M=Nimg * Ho * Wo, N=Cout, K=9 * Cin,
                                          bias=bc.to(DEV), mode=ops.A_CONV3X3, Cin=Cin, conv=(Ho, Wo, H, W, 1, ups)),
                       (Nimg * Ho * Wo, Cout))
            xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
            _close(_untokens(out, Nimg, Ho, Wo), F.conv2d(xin, wc.float(), bc, padding=1), what=f"256-column conv {Cin}->{Cout}")
        Bc, Fr, C, Ht, Wt = 1, 5, 256, 23, 10
        x5 = _h(torch.randn(Bc, C, Fr, Ht, Wt, generator=g))
        wt = _h(torch.randn(C, C, 3, 1, 1, generator=g) / (3 * C) ** 0.5)
        bt = torch.randn(C, generator=g)
        tok = x5.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()
        out = both(lambda o: ops.gemm(tok.to(DEV), pack_tconv3(wt).to(DEV), o, M=Bc * Fr * Ht * Wt, N=C, K=3 * C, bias=bt.to(DEV),
                                      mode=ops.A_TCONV3, Cin=C, tconv=(Fr, Ht * Wt)), (Bc * Fr * Ht * Wt, C))
        reft = F.conv3d(x5.float(), wt.float(), bt, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(-1, C)
        _close(out, reft, what="256-column tconv")
        # GroupNorm sums from the epilogue: conv 128 -> 512 on 4 images of 16x32 (two tiles per image), against the read pass
        Nimg, H, W = 4, 24, 32                  # 768 tokens per image: three 256-row tiles, four 192-row tiles
        T = Nimg * H * W
        x = _h(torch.randn(T, 128, generator=g)).to(DEV)
        w = torch.randn(512, 128, 3, 3, generator=g) / (9 * 128) ** 0.5
        out = torch.empty(T, 512, dtype=torch.float16, device=DEV)
        ops.gemm(x, pack_conv3x3(w).to(DEV), out, M=T, N=512, K=9 * 128, bias=torch.randn(512, generator=g).to(DEV),
                 mode=ops.A_CONV3X3, Cin=128, conv=(H, W, H, W, 1, 0), colstats=H * W)
        assert out._lkgd_colstats[1] == tile_rows
        got = ops.groupnorm_stats(out, None, Nimg, H * W, 1e-6)
        ops.COLSTATS = False
        try:
            two_pass = ops.groupnorm_stats(out, None, Nimg, H * W, 1e-6)
        finally:
            ops.COLSTATS = True
        assert torch.allclose(got.cpu(), two_pass.cpu(), rtol=2e-5, atol=2e-6)
        ref = F.conv2d(_untokens(x.cpu(), Nimg, H, W).float(), _h(w).float(), None, padding=1)
        assert out.shape == (T, 512) and torch.isfinite(out).all() and ref.shape[1] == 512
    finally:
        L.lkgd_debug_set_gemm_variant(0)
        L.lkgd_debug_set_wide_lds_out(-1)
        L.lkgd_debug_set_wide_tile_m(0)
    # the dispatcher's own choice between the two widths where both divide N (a CFG-parallel rank's 18x32 level: 8064 rows,
    # N = 1280 / 3840 -> 256-column tiles; 16 128 rows -> 320): same result under either forced width
    for M, N, K in ((8064, 1280, 1280), (8064, 3840, 1280), (16128, 1280, 1280)):
        a, w = _h(torch.randn(M, K, generator=g)).to(DEV), _h(torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
        b, res = torch.randn(N, generator=g).to(DEV), _h(torch.randn(M, N, generator=g)).to(DEV)
        outs = []
        for wn in (0, 256, 320):
            L.lkgd_debug_set_wide_tile_n(wn)
            o = torch.empty(M, N, dtype=torch.float16, device=DEV)
            ops.gemm(a, w, o, M=M, N=N, K=K, bias=b, res1=res)
            outs.append(o)
        L.lkgd_debug_set_wide_tile_n(0)
        _close(outs[0], (a.float() @ w.float().T + b + res.float()).cpu(), what=f"auto width {M}x{N}x{K}")
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])      # same K order per output element


def test_attn_cross_short_contexts():
    """lkgd_attn_cross: every row against the Lk keys of the context its row map selects (block-constant and interleaved
    maps, a table that starts at a later context, ragged T) vs fp32 softmax attention"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(31)
    for T, heads, NC, Lk, rowmap, ld in [(700, 2, 3, 5, (250, 1, 1, 1 << 30), 128), (513, 3, 2, 77, (1 << 20, 0, 2, 2), 200),
                                         (300, 1, 4, 1, (60, 7, 1, 4, 2), 64), (40, 5, 2, 128, (16, 1, 1, 2), 320)]:
        C = heads * 64
        q = _h(torch.randn(T, ld, generator=g))
        k, v = _h(torch.randn(NC * Lk, ld, generator=g)), _h(torch.randn(NC * Lk, ld, generator=g))
        d1, m1, d2, md = rowmap[:4]
        c0 = rowmap[4] if len(rowmap) > 4 else 0
        rows = torch.arange(T)
        idx = ((rows // d1) * m1 + rows % d2 + c0) % md
        kk = k[:, :C].float().reshape(NC, Lk, heads, 64)[idx]           # [T, Lk, heads, 64]
        vv = v[:, :C].float().reshape(NC, Lk, heads, 64)[idx]
        sc = torch.einsum("thd,tjhd->thj", q[:, :C].float().reshape(T, heads, 64), kk) * 0.125
        ref = torch.einsum("thj,tjhd->thd", sc.softmax(-1), vv).reshape(T, C)
        out = torch.full((T, ld), -3.0, dtype=torch.float16, device=DEV)
        ops.attn_cross(q.to(DEV)[:, :C], k.to(DEV)[:, :C], v.to(DEV)[:, :C], out[:, :C], heads, NC, Lk, rowmap)
        _close(out[:, :C], ref, what=f"attn_cross T={T} Lk={Lk}")
        assert (out[:, C:] == -3).all()
    with pytest.raises(Exception):          # a row map that reaches a context the table does not hold
        ops.attn_cross(q.to(DEV)[:, :C], k.to(DEV)[:, :C], v.to(DEV)[:, :C], out[:, :C], heads, 2, 128, (8, 1, 1, 1 << 30))


def test_attn_spatial_fewer_queries_than_keys(ops):
    """lkgd_attn_spatial_qk: Sq query rows against S key rows per batch entry (a frame-sharded DiT rank)"""
    g = torch.Generator().manual_seed(9)
    nb, heads, S, Sq = 2, 3, 200, 72
    C = heads * 64
    q = torch.randn(nb * Sq, C, generator=g).half().to(DEV)
    k = torch.randn(nb * S, C, generator=g).half().to(DEV)
    v = torch.randn(nb * S, C, generator=g).half().to(DEV)
    out = torch.empty(nb * Sq, C, dtype=torch.float16, device=DEV)
    ops.attn_spatial(q, k, v, out, nb, S, heads, Sq=Sq)
    qf, kf, vf = (t.float().reshape(nb, -1, heads, 64).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(nb * Sq, C)
    assert (out.float() - ref).abs().max() < 4e-3


def test_gemm_resident_weight_kernel():
    """lkgd_gemm_resw_kernel (K <= 320, N % 160 == 0: a 160-channel weight slab resident in LDS, waves independent): every
    epilogue source combination, ragged M (partial 32-row block, fewer blocks than waves, one row), 1 / 2 / 6 slabs,
    K = 64 .. 320, GEGLU with the 80 | 80 interleave - against fp32, and bitwise against the 256x320 kernel where both apply"""
    from lkgd_amd import _lib, ops
    from lkgd_amd.packing import pack_geglu
    L = _lib.lib()
    g = torch.Generator().manual_seed(606)
    try:
        for M, N, K in ((32 * 700 + 13, 320, 320), (1, 160, 64), (250, 960, 320), (32 * 40, 320, 128), (9000, 160, 256),
                        (32 * 257 + 31, 640, 192)):
            a, w = _h(torch.randn(M, K, generator=g)), _h(torch.randn(N, K, generator=g) / K ** 0.5)
            bias = torch.randn(N, generator=g)
            r1, r2 = _h(torch.randn(M, N, generator=g)), _h(torch.randn(M, N, generator=g))
            table = _h(torch.randn(7, N, generator=g))
            idx = (torch.arange(M) // 100) % 7
            acc = a.float() @ w.float().T
            ad, wd = a.to(DEV), w.to(DEV)
            for src in range(8):
                kw, ref = {}, acc + bias
                if src & 1:
                    kw.update(rowbias=table.to(DEV), rowmap=ops.rowmap_div_mod(100, 7))
                    ref = ref + table.float()[idx]
                ref = 0.7 * ref
                if src & 2:
                    kw.update(res1=r1.to(DEV), r1=0.6)
                    ref = ref + 0.6 * r1.float()
                if src & 4:
                    kw.update(res2=r2.to(DEV), r2=0.3)
                    ref = ref + 0.3 * r2.float()
                out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
                L.lkgd_debug_set_gemm_variant(6)
                ops.gemm(ad, wd, out, M=M, N=N, K=K, bias=bias.to(DEV), s_acc=0.7, **kw)
                _close(out, ref, what=f"resw {M}x{N}x{K} src {src}")
                if N % 320 == 0 and K >= 64:
                    out4 = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
                    L.lkgd_debug_set_gemm_variant(4)
                    ops.gemm(ad, wd, out4, M=M, N=N, K=K, bias=bias.to(DEV), s_acc=0.7, **kw)
                    assert torch.equal(out, out4), f"resw vs 256x320 {M}x{N}x{K} src {src}"
            out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
            L.lkgd_debug_set_gemm_variant(6)
            ops.gemm(ad, wd, out, M=M, N=N, K=K)                     # no bias, no sources (QKV)
            _close(out, acc, what=f"resw bare {M}x{N}x{K}")
        for M, C in ((700, 320), (32 * 300 + 5, 320), (40, 64)):
            a = _h(torch.randn(M, C, generator=g))
            w = torch.randn(8 * C, C, generator=g) / C ** 0.5
            b = torch.randn(8 * C, generator=g) * 0.1
            y = a.float() @ _h(w).float().T + b
            hid, gate = y.chunk(2, dim=-1)
            if (4 * C) % 80:
                continue
            wp, bp, half = pack_geglu(w, b, half=80)
            out = torch.full((M, 4 * C), float("nan"), dtype=torch.float16, device=DEV)
            L.lkgd_debug_set_gemm_variant(6)
            ops.gemm(a.to(DEV), wp.to(DEV), out, M=M, N=8 * C, K=C, bias=bp.to(DEV), geglu=half)
            _close(out, hid * F.gelu(gate), what=f"resw geglu {M}x{C}")
            out4 = torch.full((M, 4 * C), float("nan"), dtype=torch.float16, device=DEV)
            L.lkgd_debug_set_gemm_variant(4)
            ops.gemm(a.to(DEV), wp.to(DEV), out4, M=M, N=8 * C, K=C, bias=bp.to(DEV), geglu=half)
            assert torch.equal(out, out4), f"resw geglu vs 256x320 {M}x{C}"
    finally:
        L.lkgd_debug_set_gemm_variant(0)


def test_groupnorm_statistics_from_gemm_epilogues():
    """lkgd_gemm_desc.colstats: the producing GEMM leaves per-(row block, channel) sums of its rounded outputs; the
    GroupNorm statistics taken from them equal the statistics of the separate read pass over the same fp16 tensor
    (same values, another summation order) - 3x3 conv, temporal conv and a plain linear on the 256x320 program (256-row
    blocks, ragged last tile), the resident-weight program (32-row blocks), a two-source input whose groups straddle the
    sources, the 5-D temporal form and the raw-sums form of the frame-sharded path"""
    from lkgd_amd import _lib, ops
    from lkgd_amd.packing import pack_conv3x3, pack_tconv3
    g = torch.Generator().manual_seed(909)
    Nimg, H, W = 4, 16, 32                       # 512 tokens per image: two 256-row tiles
    T = Nimg * H * W

    def stats_ref(x, nsamples, rows):
        xs = x.float().reshape(nsamples, rows, 32, -1)
        m = xs.mean(dim=(1, 3))
        v = xs.var(dim=(1, 3), unbiased=False)
        return torch.stack([m, 1.0 / torch.sqrt(v + 1e-5)], dim=-1)

    def check(out, x1=None, nsamples=Nimg, rows=H * W):
        assert getattr(out, "_lkgd_colstats", None) is not None, "the GEMM did not attach column sums"
        got = ops.groupnorm_stats(out, x1, nsamples, rows, 1e-5)
        ops.COLSTATS = False
        try:
            two_pass = ops.groupnorm_stats(out, x1, nsamples, rows, 1e-5)
        finally:
            ops.COLSTATS = True
        full = out if x1 is None else torch.cat([out, x1], dim=1)
        ref = stats_ref(full.cpu(), nsamples, rows)
        assert torch.allclose(got.cpu(), two_pass.cpu(), rtol=2e-5, atol=2e-6)
        assert torch.allclose(got.cpu(), ref, rtol=1e-3, atol=1e-4)
        sums = ops.groupnorm_sums(out, x1, nsamples, rows)
        cnt = rows * full.shape[1] // 32
        assert torch.allclose(sums.cpu()[..., 0] / cnt, ref[..., 0], rtol=1e-3, atol=1e-4)
        if x1 is None:       # an in-place update of the tensor voids the attached sums: the statistics follow the NEW values
            keep = out.clone()
            out.mul_(0.5)
            half = ops.groupnorm_stats(out, None, nsamples, rows, 1e-5)
            assert torch.allclose(half.cpu(), stats_ref(out.cpu(), nsamples, rows), rtol=1e-3, atol=1e-4)
            out.copy_(keep)

    # (small shapes: the 256x320 program is forced - the automatic dispatch would give 16 tiles to the 128x128 program,
    # which attaches nothing, as the last case checks)
    _lib.lib().lkgd_debug_set_gemm_variant(4)
    # 3x3 conv 320 -> 640 with time-embedding row bias (ResnetBlock conv1 -> norm2)
    x = _h(torch.randn(T, 320, generator=g)).to(DEV)
    w = torch.randn(640, 320, 3, 3, generator=g) / (9 * 320) ** 0.5
    temb = _h(torch.randn(Nimg, 640, generator=g)).to(DEV)
    out = torch.empty(T, 640, dtype=torch.float16, device=DEV)
    ops.gemm(x, pack_conv3x3(w).to(DEV), out, M=T, N=640, K=9 * 320, bias=torch.randn(640, generator=g).to(DEV),
             mode=ops.A_CONV3X3, Cin=320, conv=(H, W, H, W, 1, 0), rowbias=temb, rowmap=ops.rowmap_div(H * W), colstats=H * W)
    check(out)
    # a second tensor (320 channels) as the other source of a concatenated input: 960 channels, 30 per group
    out2 = torch.empty(T, 320, dtype=torch.float16, device=DEV)
    w2 = torch.randn(320, 320, 3, 3, generator=g) / (9 * 320) ** 0.5
    ops.gemm(x, pack_conv3x3(w2).to(DEV), out2, M=T, N=320, K=9 * 320, mode=ops.A_CONV3X3, Cin=320, conv=(H, W, H, W, 1, 0),
             res1=x, colstats=H * W)
    check(out, out2)
    # temporal conv over F = 4 frames, statistics across the frames of a clip (5-D GroupNorm: one sample = F * HW rows)
    wt = torch.randn(320, 320, 3, 1, 1, generator=g) / (3 * 320) ** 0.5
    out3 = torch.empty(T, 320, dtype=torch.float16, device=DEV)
    ops.gemm(x, pack_tconv3(wt).to(DEV), out3, M=T, N=320, K=3 * 320, mode=ops.A_TCONV3, Cin=320, tconv=(2, H * W),
             s_acc=0.5, res1=x, colstats=2 * H * W)
    check(out3, nsamples=2, rows=2 * H * W)
    _lib.lib().lkgd_debug_set_gemm_variant(0)
    # plain linear with residual at K = 320 and 72k rows: the resident-weight program (32-row blocks); 640 rows per sample
    M = 112 * 640
    a = _h(torch.randn(M, 320, generator=g)).to(DEV)
    wl = _h(torch.randn(320, 320, generator=g) / 320 ** 0.5).to(DEV)
    res = _h(torch.randn(M, 320, generator=g)).to(DEV)
    out4 = torch.empty(M, 320, dtype=torch.float16, device=DEV)
    ops.gemm(a, wl, out4, M=M, N=320, K=320, bias=torch.randn(320, generator=g).to(DEV), res1=res, colstats=640)
    assert out4._lkgd_colstats[1] == 32
    check(out4, nsamples=112, rows=640)
    # plain linear at K = 640 (256x320 program, rows through LDS), last tile ragged; samples of 256 rows
    M = 256 * 13 + 0
    a = _h(torch.randn(M, 640, generator=g)).to(DEV)
    wl = _h(torch.randn(640, 640, generator=g) / 640 ** 0.5).to(DEV)
    out5 = torch.empty(M, 640, dtype=torch.float16, device=DEV)
    _lib.lib().lkgd_debug_set_gemm_variant(4)
    try:
        ops.gemm(a, wl, out5, M=M, N=640, K=640, res1=a, colstats=256)
    finally:
        _lib.lib().lkgd_debug_set_gemm_variant(0)
    assert out5._lkgd_colstats[1] == 256
    check(out5, nsamples=13, rows=256)
    # a program without column sums (128x128 tiles: N = 64): nothing attached, the read pass runs
    out6 = torch.empty(512, 64, dtype=torch.float16, device=DEV)
    ops.gemm(a[:512], wl[:64], out6, M=512, N=64, K=640, colstats=256)
    assert getattr(out6, "_lkgd_colstats", None) is None


@pytest.mark.parametrize("B,Fr,HW", [(2, 14, 48), (1, 16, 16), (3, 5, 1024), (2, 1, 32)])
def test_temporal_attention_front_fused(B, Fr, HW):
    """lkgd_tattn_front: LayerNorm + Q|K|V projection + attention over the frames of every (pixel, head) in one kernel, against
    fp32 (F.layer_norm -> F.linear -> SDPA over the frame axis) and against the three-launch HIP form; several panels per
    workgroup at HW = 1024, frame counts 1 / 5 / 14 / 16 (masked keys, padded rows)"""
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_linear, pack_tfront
    g = torch.Generator().manual_seed(100 * B + Fr)
    C, heads = 320, 5
    T = B * Fr * HW
    x = _h(torch.randn(T, C, generator=g) * 1.7 + 0.4)
    w = torch.randn(3 * C, C, generator=g) / C ** 0.5
    b = torch.randn(3 * C, generator=g) * 0.2
    wh = _h(w)
    xn = F.layer_norm(x.float(), (C,), None, None, 1e-5)
    qkv = xn @ wh.float().T + b
    q, k, v = (t.reshape(B, Fr, HW, heads, 64).permute(0, 2, 3, 1, 4) for t in qkv.chunk(3, dim=-1))     # [B, HW, h, F, 64]
    ref = F.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(T, C)
    out = torch.full((T, C), float("nan"), dtype=torch.float16, device=DEV)
    ops.tattn_front(x.to(DEV), pack_tfront(wh, heads).to(DEV), b.to(DEV), out, B, Fr, HW, heads)
    _close(out, ref, what=f"fused temporal front B={B} F={Fr} HW={HW}")
    # the unfused HIP form of the same block front
    xd = x.to(DEV)
    ln = ops.layernorm(xd, None, None, 1e-5)
    qkv_d = torch.empty(T, 3 * C, dtype=torch.float16, device=DEV)
    ops.gemm(ln, pack_linear(wh).to(DEV), qkv_d, M=T, N=3 * C, K=C, bias=b.to(DEV))
    att = torch.empty(T, C, dtype=torch.float16, device=DEV)
    ops.attn_temporal(qkv_d[:, :C], qkv_d[:, C:2 * C], qkv_d[:, 2 * C:], att, B, Fr, HW, heads)
    assert ((out.float() - att.float()).norm() / att.float().norm()).item() < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,res", [(40000, 960, 320, False), (33000, 320, 320, True), (36864, 576, 192, False), (4100, 768, 256, False)])
def test_gemm_layernorm_fold(M, N, K, res):
    """lkgd_gemm_desc.ln_colsum: LayerNorm of the A rows folded into the row-panel GEMM (patch/patch.py:416 norm1 -> to_q/k/v) vs
    fp32 layer_norm -> linear, and vs the two-launch HIP form; rows with |mean| >> sigma included"""
    from lkgd_amd import ops
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device=DEV) * 1.3
    x[::7] += 6.0                                     # large row means: the fold subtracts mean * colsum in fp32
    x[5::11] *= 0.05
    x = x.half()
    gamma, beta = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.2
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV) * 0.1
    wf = (w * gamma[None, :]).half().contiguous()      # gamma folded into the weights, beta into the bias
    bf = (b + w @ beta).contiguous()
    cs = wf.float().sum(dim=1).contiguous()
    r = (torch.randn(M, N, device=DEV) * 0.5).half() if res else None
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(x, wf, out, M=M, N=N, K=K, bias=bf, res1=r, ln=(cs, 1e-5))
    want = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.t() + b
    if res:
        want = want + r.float()
    err = (out.float() - want).abs().max().item()
    rel = ((out.float() - want).norm() / want.norm()).item()
    assert rel < 2e-3 and err < 3e-2, (rel, err)
    two = torch.empty_like(out)
    ops.gemm(ops.layernorm(x, None, None, 1e-5), wf, two, M=M, N=N, K=K, bias=bf, res1=r)
    rel2 = ((out.float() - two.float()).norm() / two.float().norm()).item()
    assert rel2 < 2e-3, rel2
    # shapes outside the row-panel program are refused, not silently run without the LayerNorm
    from lkgd_amd._lib import LkgdHipError
    xw = torch.zeros(4096, 640, dtype=torch.float16, device=DEV)
    with pytest.raises(LkgdHipError):
        ops.gemm(xw, torch.zeros(640, 640, dtype=torch.float16, device=DEV), torch.empty(4096, 640, dtype=torch.float16, device=DEV),
                 M=4096, N=640, K=640, ln=(torch.zeros(640, device=DEV), 1e-5))


@pytest.mark.parametrize("fl,HW,C,px", [(4, 144, 320, (48, 32, 32, 32)), (3, 9216, 320, (2304, 2304, 2304, 2304)), (1, 7, 8, (3, 2, 2)),
                                         (2, 576, 1280, (576,))])
def test_shard_rows_pack_and_unpack_equal_the_strided_copies(ops, fl, HW, C, px):
    """lkgd_shard_rows: the send buffer of frames_to_pixels (rows grouped by destination pixel shard) and its inverse, against
    the k strided copies of lkgd_amd/dist.py"""
    g = torch.Generator().manual_seed(fl * 1000 + HW)
    local = torch.randn(fl, HW, C, generator=g).half().to(DEV)
    want = torch.cat([local[:, sum(px[:r]):sum(px[:r + 1]), :].reshape(-1, C) for r in range(len(px))])
    send = torch.full((fl * HW, C), float("nan"), dtype=torch.float16, device=DEV)
    ops.shard_rows(local, send, fl, HW, C, px, True)
    assert torch.equal(send, want)
    back = torch.full((fl, HW, C), float("nan"), dtype=torch.float16, device=DEV)
    ops.shard_rows(send, back, fl, HW, C, px, False)
    assert torch.equal(back, local)
