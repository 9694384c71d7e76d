"""The software-pipelined spatial-attention program (lkgd_amd/csrc/attn_spatial_pipe.hip, generated main loop) against
F.scaled_dot_product_attention in fp32 (`patch/patch.py:503-508` -> AttnProcessor2_0 [EXT]) and against the compiler-scheduled
kernel it replaces at S % 128 == 0.  Tolerance: fp16 outputs, 4e-3 of the output scale + 2e-3 absolute, relative L2 < 3e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    from lkgd_amd import _lib
    lib = _lib.lib()
    yield lib
    lib.lkgd_debug_set_attn_pipe(0)


def _ref(q, k, v, nb, heads, kvperm=None):
    qf, kf, vf = (t.float().cpu().reshape(nb, -1, heads, 64).transpose(1, 2) for t in (q, k, v))
    if kvperm is not None:
        kf, vf = kf[kvperm], vf[kvperm]
    o = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2)
    return o.reshape(-1, heads * 64)


def _close(got, ref, what):
    got = got.float().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err <= 4e-3 * scale + 2e-3, f"{what}: max abs err {err:.4g} vs scale {scale:.4g}"
    rel = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    assert rel < 3e-3, f"{what}: relative L2 {rel:.4g}"


# S = 128: one stage (prologue -> tail); 256 / 384: the loop once / twice, every LDS buffer of the ring; 640: the ring wraps;
# Sq: a partial last workgroup (rows >= Sq masked), more than one workgroup per (batch, head), fewer queries than keys
@pytest.mark.parametrize("S,Sq,heads,nb", [(128, 128, 1, 1), (256, 256, 2, 2), (384, 384, 1, 3), (640, 640, 3, 2),
                                           (1024, 1024, 2, 2), (2304, 2304, 2, 1), (512, 72, 2, 2), (1152, 1100, 1, 2)])
def test_pipe_vs_fp32(L, S, Sq, heads, nb):
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(S + Sq + heads)
    C = heads * 64
    q = torch.randn(nb * Sq, C, generator=g).half().to(DEV)
    k = torch.randn(nb * S, C, generator=g).half().to(DEV)
    v = torch.randn(nb * S, C, generator=g).half().to(DEV)
    out = torch.full((nb * Sq, C), float("nan"), dtype=torch.float16, device=DEV)
    L.lkgd_debug_set_attn_pipe(2)
    ops.attn_spatial(q, k, v, out, nb, S, heads, Sq=Sq)
    _close(out, _ref(q, k, v, nb, heads), f"pipe S={S} Sq={Sq}")
    # the kernel it replaces, same inputs: both within fp16 rounding of each other
    old = torch.empty_like(out)
    L.lkgd_debug_set_attn_pipe(1)
    ops.attn_spatial(q, k, v, old, nb, S, heads, Sq=Sq)
    assert (out.float() - old.float()).abs().max().item() < 4e-3
    if nb >= 2:          # joint attention: K / V of the partner batch entry (patch/patch.py:466-468)
        perm = torch.arange(nb).flip(0)
        L.lkgd_debug_set_attn_pipe(2)
        ops.attn_spatial(q, k, v, out, nb, S, heads, kv_batch_map=perm.to(torch.int32).to(DEV), Sq=Sq)
        _close(out, _ref(q, k, v, nb, heads, perm), f"pipe kv-map S={S}")


def test_pipe_strided_qkv_layout(L):
    """the UNet's layout: Q | K | V are column blocks of one [T, 3C] projection output (ld = 3C)"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(5)
    nb, heads, S = 2, 5, 768
    C = heads * 64
    qkv = torch.randn(nb * S, 3 * C, generator=g).half().to(DEV)
    out = torch.empty(nb * S, C, dtype=torch.float16, device=DEV)
    L.lkgd_debug_set_attn_pipe(2)
    ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb, S, heads)
    _close(out, _ref(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], nb, heads), "pipe strided")


@pytest.mark.parametrize("where", [5, 70, 130, 200, 260, 330, 400, 460, 511])
def test_pipe_reference_moves(L, where):
    """one key dominates queries of tile A and of tile B of a wave, in every 64-key sub-tile of a four-stage sequence: every
    instance of the out-of-line block that recomputes a unit (one per phase of the loop body + the tail; the sub-tile one
    stage behind the fragment addresses included) moves the reference and rescales O and l exactly once"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(99 + where)
    S, C = 512, 64
    q, k, v = (torch.randn(S, C, generator=g) for _ in range(3))
    k[where] = q[3] * 6.0 + q[40] * 5.0                 # queries of tile A / tile B of wave 0
    k[(where + 77) % S] = q[100] * 7.0                  # tile B of wave 1
    k[(where + 130) % S] = q[250] * 4.0 + q[200] * 4.0  # tile B / tile A of wave 3
    q, k, v = q.half().to(DEV), k.half().to(DEV), v.half().to(DEV)
    out = torch.empty(S, C, dtype=torch.float16, device=DEV)
    L.lkgd_debug_set_attn_pipe(2)
    ops.attn_spatial(q, k, v, out, 1, S, 1)
    _close(out, _ref(q, k, v, 1, 1), f"pipe rescale key {where}")


def test_pipe_very_negative_and_very_large_scores(L):
    """all scores of a query far below zero (the first reference must follow them down: no underflow to l = 0), and scores
    growing tile after tile (the reference moves at every tile)"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(7)
    S, C = 512, 64
    q, k, v = (torch.randn(S, C, generator=g) for _ in range(3))
    k = k * 0.3
    q[17] = -k.mean(0) * 400.0                          # every score of query 17 strongly negative on average
    ramp = torch.linspace(0.2, 6.0, S)[:, None]
    k[:, :8] = q[200, :8][None, :] * ramp               # scores of query 200 grow with the key index
    q, k, v = q.half().to(DEV), k.half().to(DEV), v.half().to(DEV)
    out = torch.empty(S, C, dtype=torch.float16, device=DEV)
    L.lkgd_debug_set_attn_pipe(2)
    ops.attn_spatial(q, k, v, out, 1, S, 1)
    assert torch.isfinite(out).all()
    _close(out, _ref(q, k, v, 1, 1), "pipe extreme scores")


def test_pipe_is_deterministic_and_default_at_unet_levels(L):
    """bitwise repeatable; and the dispatch rule sends the 72x128-level shape to it (same bits as the forced run)"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(11)
    nb, heads, S = 1, 1, 9216
    C = heads * 64
    qkv = torch.randn(nb * S, 3 * C, generator=g).half().to(DEV)
    outs = []
    for mode in (2, 2, 0):
        L.lkgd_debug_set_attn_pipe(mode)
        o = torch.empty(nb * S, C, dtype=torch.float16, device=DEV)
        ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, nb, S, heads)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


# S % 128 != 0: the masked form of the statement - the last stage is keys [S - 128, S), its overlap with the stage before it masked.
# 129 / 255: two stages that overlap almost entirely / by one key; 200, 300: the overlap inside the first / second unit of the last
# stage; 1000 and 2160 (= 16 x 128 + 112, CogVideoX's remainder): many stages; Sq != S; K / V are followed by NaN rows - a read
# beyond the S keys of the last batch entry would show
@pytest.mark.parametrize("S,Sq,heads,nb", [(129, 129, 1, 1), (255, 255, 2, 1), (200, 200, 1, 2), (300, 300, 2, 2), (448, 448, 1, 1),
                                           (1000, 1000, 2, 2), (2160, 2160, 3, 1), (777, 100, 2, 2)])
def test_pipe_ragged_keys_vs_fp32(L, S, Sq, heads, nb):
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(3 * S + Sq + heads)
    C = heads * 64
    q = torch.randn(nb * Sq, C, generator=g).half().to(DEV)
    kbig = torch.full((nb * S + 128, C), float("nan"), dtype=torch.float16, device=DEV)
    vbig = torch.full((nb * S + 128, C), float("nan"), dtype=torch.float16, device=DEV)
    kbig[:nb * S] = torch.randn(nb * S, C, generator=g).half().to(DEV)
    vbig[:nb * S] = torch.randn(nb * S, C, generator=g).half().to(DEV)
    k, v = kbig[:nb * S], vbig[:nb * S]
    out = torch.full((nb * Sq, C), float("nan"), dtype=torch.float16, device=DEV)
    try:
        L.lkgd_debug_set_attn_pipe(2)
        ops.attn_spatial(q, k, v, out, nb, S, heads, Sq=Sq)
        _close(out, _ref(q, k, v, nb, heads), f"masked pipe S={S} Sq={Sq}")
        again = torch.empty_like(out)
        ops.attn_spatial(q, k, v, again, nb, S, heads, Sq=Sq)
        assert torch.equal(out, again)
        old = torch.empty_like(out)
        L.lkgd_debug_set_attn_pipe(1)
        ops.attn_spatial(q, k, v, old, nb, S, heads, Sq=Sq)
        assert (out.float() - old.float()).abs().max().item() < 4e-3
        if nb >= 2:
            perm = torch.arange(nb).flip(0)
            L.lkgd_debug_set_attn_pipe(2)
            ops.attn_spatial(q, k, v, out, nb, S, heads, kv_batch_map=perm.to(torch.int32).to(DEV), Sq=Sq)
            _close(out, _ref(q, k, v, nb, heads, perm), f"masked pipe kv-map S={S}")
    finally:
        L.lkgd_debug_set_attn_pipe(0)


def test_pipe_ragged_keys_with_a_moving_reference(L):
    """large scores in the LAST (overlapping) stage: its units are redone out of line with the same duplicate masks"""
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(11)
    S, C = 700, 64
    q = torch.randn(S, C, generator=g).half()
    k = torch.randn(S, C, generator=g).half()
    v = torch.randn(S, C, generator=g).half()
    k[690] = 9.0 * q[5]                     # one key far above the running reference, inside the last stage
    k[600] = 7.0 * q[300]                   # ... and one among the masked duplicates' stage neighbours
    q, k, v = q.to(DEV), k.to(DEV), v.to(DEV)
    out = torch.empty(S, C, dtype=torch.float16, device=DEV)
    try:
        L.lkgd_debug_set_attn_pipe(2)
        ops.attn_spatial(q, k, v, out, 1, S, 1)
        _close(out, _ref(q, k, v, 1, 1), "masked pipe, reference moves in the last stage")
    finally:
        L.lkgd_debug_set_attn_pipe(0)
