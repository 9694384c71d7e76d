"""CPU-side checks of the product's host logic (no kernels run): the C-ABI library loads and exports every symbol the
header declares, the scheduler host tables equal the reference's KAT, state-dict compatibility with the diffusers
names, weight packing layouts, patch API bookkeeping, loud failure without a GPU."""
import ctypes
import json
import os
import sys
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from lkgd_amd import _lib
    hdr = open(os.path.join(REPO, "include", "lkgd_hip.h")).read()
    declared = set(re.findall(r"\b(lkgd_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("lkgd_gemm_desc")
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert not any(s.startswith("lkgd_debug") for s in declared)          # the product header carries no knob
    dbg = open(os.path.join(REPO, "include", "lkgd_hip_debug.h")).read()
    dbg_declared = set(re.findall(r"^void (lkgd_debug_[a-z0-9_]+)\s*\(", dbg, re.M))
    assert dbg_declared == set(_lib.DEBUG_SYMBOLS), dbg_declared ^ set(_lib.DEBUG_SYMBOLS)
    lib = _lib.lib()                       # loads liblkgd_hip.so (links libamdhip64; no GPU needed to load)
    for s in declared | dbg_declared:
        assert hasattr(lib, s), s
    assert lib.lkgd_version().decode().startswith("lkgd_hip")
    # 31 int32/float fields, padded to 8; workspace + size; colstats; ln_colsum + ln_eps + pad
    assert ctypes.sizeof(_lib.GemmDesc) == 9 * 8 + 31 * 4 + 4 + 16 + 8 + 16


def test_debug_knobs_are_per_thread():
    """include/lkgd_hip_debug.h: a knob set by one host thread does not reach another thread's launches ("no global
    state; safe to call from multiple host threads").  lkgd_gemm_wide_tile_n is a host-side introspection of the tile rule that
    honours the forced width, and lkgd_groupnorm_chunks honours the chunk-size knobs: both are read from two threads at once."""
    import threading
    from lkgd_amd import _lib
    lib = _lib.lib()
    assert lib.lkgd_gemm_wide_tile_n(1280) == 320
    base_chunks = lib.lkgd_groupnorm_chunks(14 * 9216, 320)
    seen, gate_a, gate_b = {}, threading.Event(), threading.Event()

    def thread_a():
        lib.lkgd_debug_set_wide_tile_n(256)
        lib.lkgd_debug_set_gn_apply_kb(64)
        seen["a"] = (lib.lkgd_gemm_wide_tile_n(1280), lib.lkgd_groupnorm_chunks(14 * 9216, 320))
        gate_a.set()
        gate_b.wait(10)                                   # stay alive, knob set, while thread b looks
        seen["a2"] = lib.lkgd_gemm_wide_tile_n(1280)
        lib.lkgd_debug_set_wide_tile_n(0)

    def thread_b():
        gate_a.wait(10)
        seen["b"] = (lib.lkgd_gemm_wide_tile_n(1280), lib.lkgd_groupnorm_chunks(14 * 9216, 320))
        lib.lkgd_debug_set_wide_tile_n(320)
        gate_b.set()

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start(); ta.join(); tb.join()
    assert seen["a"][0] == 256 and seen["a2"] == 256 and seen["a"][1] != base_chunks
    assert seen["b"] == (320, base_chunks)
    assert lib.lkgd_gemm_wide_tile_n(1280) == 320 and lib.lkgd_groupnorm_chunks(14 * 9216, 320) == base_chunks   # this thread: untouched


def test_gemm_desc_validation_without_gpu():
    """argument errors are reported before anything touches the device"""
    from lkgd_amd import _lib
    d = _lib.GemmDesc()
    assert _lib.lib().lkgd_gemm_f16(ctypes.byref(d), None) == -1      # LKGD_E_NULL
    # small maps are cut down to 8-KiB chunks (12 rows of 320 channels, 8 rows of 1280); large ones keep the 32-KiB apply chunks
    assert _lib.lib().lkgd_groupnorm_chunks(129, 320) == 11 and _lib.lib().lkgd_groupnorm_chunks(576, 1280) == 72
    assert _lib.lib().lkgd_groupnorm_chunks(14 * 9216, 320) == (14 * 9216 + 50) // 51


def test_scheduler_tables_equal_reference_kat():
    from lkgd_amd.scheduler import EulerDiscreteScheduler
    with open(os.path.join(REPO, "tests", "golden", "scheduler_kat.json")) as f:
        kat = json.load(f)
    for c in kat["cases"]:
        s = EulerDiscreteScheduler(**kat["config"])
        s.set_timesteps(c["n"])
        assert torch.equal(s.sigmas, torch.tensor(c["sigmas"]))
        assert torch.equal(s.timesteps, torch.tensor(c["timesteps"]))
        assert float(s.init_noise_sigma) == c["init_noise_sigma"]
        assert s.order == 1 and s.step_index is None
    s = EulerDiscreteScheduler.from_svd_config()
    s.set_timesteps(25)
    with pytest.raises(ValueError):
        s.step(torch.zeros(1), 3, torch.zeros(1))          # integer timestep guard (:458-469)


def test_state_dict_names_match_diffusers_tree():
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    for ocls, pcls in ((ou.UNetSpatioTemporalConditionControlNetModel, pu.UNetSpatioTemporalConditionControlNetModel),
                       (ou.UNetSpatioTemporalConditionModel, pu.UNetSpatioTemporalConditionModel)):
        with torch.device("meta"):
            o, p = ocls(ou.SVD_CONFIG), pcls(pu.UNetConfig())
        so, sp = o.state_dict(), p.state_dict()
        assert set(so) == set(sp)
        assert all(so[k].shape == sp[k].shape for k in so)
    assert sum(v.numel() for v in sp.values()) == 1_524_623_082 + 726_796
    names = {type(m).__name__ for m in p.modules()}
    assert {"BasicTransformerBlock", "TemporalBasicTransformerBlock"} <= names   # what patch looks for
    assert p.config.in_channels == 8 and p.config.num_frames == 14 and p.add_embedding.linear_1.in_features == 768
    blk = p.down_blocks[0].attentions[0].transformer_blocks[0]
    for a in ("attn1", "attn2", "norm1", "norm2", "norm3", "ff", "norm_type", "pos_embed", "only_cross_attention",
              "_chunk_size", "_chunk_dim"):
        assert hasattr(blk, a), a
    assert blk.attn1.out_dim == 320


def test_packing_layouts():
    from lkgd_amd import packing as pk
    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 128, 3, 3, generator=g)
    p = pk.pack_conv3x3(w)
    assert p.shape == (6, 9 * 128) and p.dtype == torch.float16
    # k = ((ky * (Cin/64) + c // 64) * 3 + kx) * 64 + c % 64   (include/lkgd_hip.h section 1)
    for (ky, kx, c) in ((1, 2, 3), (0, 0, 64), (2, 1, 127)):
        assert torch.equal(p[2, ((ky * 2 + c // 64) * 3 + kx) * 64 + c % 64], w[2, c, ky, kx].half())
    w8 = torch.randn(5, 8, 3, 3, generator=g)
    p = pk.pack_conv3x3_c8(w8)
    assert p.shape == (5, 128) and torch.equal(p[:, 72:], torch.zeros(5, 56, dtype=torch.float16))
    assert torch.equal(p[1, 4 * 8 + 7], w8[1, 7, 1, 1].half())
    wt = torch.randn(6, 4, 3, 1, 1, generator=g)
    p = pk.pack_tconv3(wt)
    assert torch.equal(p[3, 2 * 4 + 1], wt[3, 1, 2, 0, 0].half())
    perm = pk.geglu_perm(128, 32)
    assert perm[:32].tolist() == list(range(32)) and perm[32:64].tolist() == list(range(128, 160))
    assert perm[64:96].tolist() == list(range(32, 64)) and sorted(perm.tolist()) == list(range(256))
    perm = pk.geglu_perm(1280, 80)
    assert perm[:80].tolist() == list(range(80)) and perm[80:160].tolist() == list(range(1280, 1360))
    assert pk.geglu_half(2560) == 32 and pk.geglu_half(512) == 32


def test_patch_api_bookkeeping():
    from lkgd_amd import patch
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    m = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    patch.apply_patch(m, flip=True, with_spatial_block=True, with_temporal_block=False)
    sp = [b for b in m.modules() if type(b).__name__ == "BasicTransformerBlock"]
    tp = [b for b in m.modules() if type(b).__name__ == "TemporalBasicTransformerBlock"]
    assert all(b.enable_joint_attention for b in sp) and not any(b.enable_joint_attention for b in tp)
    assert m._tome_info["args"]["flip"] is True
    patch.initialize_joint_layers(m)
    assert all(hasattr(b, "attn1n") and float(b.conv1n.weight.abs().sum()) == 0.0 for b in sp)
    assert any(k.endswith("attn1n.to_q.weight") for k in m.state_dict())      # A.10 hook names
    patch.set_joint_attention_mask(m, [0, 1, 0, 1])
    patch.set_joint_scale(m, 0.25)
    assert all(b.joint_scale == 0.25 for b in sp)
    patch.set_joint_attention(m, False, name_filter="down_blocks")
    assert not m.down_blocks[0].attentions[0].transformer_blocks[0].enable_joint_attention
    assert m.mid_block.attentions[0].transformer_blocks[0].enable_joint_attention
    patch.update_patch(m, foo=3)
    assert patch.collect_from_patch(m, "foo")
    patch.remove_patch(m)
    assert not any(b.enable_joint_attention for b in sp)
    # partner permutation for masks [0,1,0,1], B=4, F=3 (patch.py:466-475)
    patch.apply_patch(m, flip=False)
    patch.set_joint_attention_mask(m, [0, 1, 0, 1])
    ctx = pu.Ctx(4, 3, 2, 2, torch.device("cpu"))
    m._joint_maps(ctx)
    assert ctx.spatial_partner.tolist() == [3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8]
    assert ctx.temporal_partner.tolist() == [1, 0, 3, 2]
    m._tome_info["args"]["flip"] = True
    m._joint_maps(ctx)
    assert ctx.spatial_partner.tolist() == [5, 4, 3, 2, 1, 0, 11, 10, 9, 8, 7, 6]
    # the LoRA-mask entry points work on lkgd_amd.lora.Linear wrappers (tests/test_lora.py); without wrappers they are no-ops
    assert patch.hack_lora_forward(m) is m and patch.set_patch_lora_mask(m, "xy_lora", [1, 0, 1, 0]) is m
    assert m.lora_mask["xy_lora"].tolist() == [True, False, True, False]
    from lkgd_amd import patch_FSM
    with pytest.raises(AttributeError):
        patch_FSM.initialize_joint_lora(m, "a", "b")       # commented out in the reference's ToMeBlock (patch_FSM.py:107-120)


def test_forward_fails_loudly_without_gpu():
    from lkgd_amd import LkgdHipError
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    m = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    with pytest.raises(LkgdHipError):
        m(torch.zeros(1, 2, 8, 8, 8), 1.0, torch.zeros(1, 1, 1024), added_time_ids=torch.zeros(1, 3))
    with pytest.raises(LkgdHipError):
        pu.Attention(64, None, 4, 16)          # head_dim != 64 is refused at construction


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import lkgd_amd, lkgd_amd.unet, lkgd_amd.pipeline, lkgd_amd.patch, "
            "lkgd_amd.scheduler, lkgd_amd.lk_fuse, lkgd_amd.dist; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % REPO)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_fsm_track_tables_match_reference_indexing():
    """host side of the FSM hook (patch_FSM.py:381-403): CSR inversion of the tracks == what scatter_add would visit"""
    from types import SimpleNamespace
    from lkgd_amd import patch_FSM
    from lkgd_amd._lib import LkgdHipError
    g = torch.Generator().manual_seed(3)
    pairs, P, fh, fw = 3, 50, 5, 7
    src = torch.stack([torch.randint(0, 2 * fw, (pairs, P), generator=g),
                       torch.randint(0, 2 * fh, (pairs, P), generator=g)], -1).float()
    dst = torch.stack([torch.randint(-4, 2 * fw + 4, (pairs, P), generator=g),
                       torch.randint(-4, 2 * fh + 4, (pairs, P), generator=g)], -1).float()
    vis = (torch.rand(pairs, P, generator=g) > 0.3).float()
    blk = SimpleNamespace(_tome_info={"fsm_tables": {}}, track=(src, dst, vis), track_res=(2 * fh, 2 * fw))
    ctx = SimpleNamespace(H=fh, W=fw, HW=fh * fw, N=2 * pairs, B=1, F=2 * pairs, b0=0, f0=0, B_total=1, F_total=2 * pairs,
                          device=torch.device("cpu"))
    (off, pt, gi, v), (boff, bpt, bgi, _) = patch_FSM.track_tables(blk, ctx)
    # a sharded rank (round 6): entries [b0, b0 + B) x frames [f0, f0 + F) of a 2-entry x 6-frame call take their pairs' rows
    if pairs == 3:
        blk6 = SimpleNamespace(_tome_info={"fsm_tables": {}}, track=(torch.cat([src, src]), torch.cat([dst, dst]),
                                                                     torch.cat([vis, 1.0 - vis])), track_res=(2 * fh, 2 * fw))
        c2 = SimpleNamespace(H=fh, W=fw, HW=fh * fw, N=2, B=1, F=2, b0=1, f0=4, B_total=2, F_total=6, device=torch.device("cpu"))
        (_, _, gi2, v2), _ = patch_FSM.track_tables(blk6, c2)          # global pair (1 * 6 + 4) / 2 = 5 = the second copy's pair 2
        assert torch.equal(gi2, gi.reshape(pairs, -1)[2].reshape(-1)) and torch.equal(v2, 1.0 - v.reshape(pairs, -1)[2].reshape(-1))
    assert off.dtype == torch.int32 and off.numel() == pairs * fh * fw + 1 and int(off[-1]) == pairs * P
    sidx = (src / 2).long()
    sidx = sidx[..., 0] + sidx[..., 1] * fw
    didx = (dst / 2).long()
    didx = didx[..., 0].clamp(0, fw - 1) + didx[..., 1].clamp(0, fh - 1) * fw
    for table, tgt, gat, o_, p_ in ((0, sidx, didx, off, pt), (1, didx, sidx, boff, bpt)):
        g_ = gi if table == 0 else bgi
        for pair in range(pairs):
            for cell in range(fh * fw):
                want = [pair * P + k for k in range(P) if int(tgt[pair, k]) == cell]     # increasing point order
                r = pair * fh * fw + cell
                assert p_[int(o_[r]):int(o_[r + 1])].tolist() == want
        assert g_.tolist() == gat.reshape(-1).tolist()
    assert torch.equal(v, vis.reshape(-1))
    assert patch_FSM.track_tables(blk, ctx) is blk._tome_info["fsm_tables"][(fh, fw, 2 * pairs, 0, 0, 1, 2 * pairs)]   # cached
    bad = SimpleNamespace(_tome_info={"fsm_tables": {}}, track=(src - 100.0, dst, vis), track_res=(2 * fh, 2 * fw))
    with pytest.raises(LkgdHipError):
        patch_FSM.track_tables(bad, ctx)
    with pytest.raises(LkgdHipError):
        patch_FSM.track_tables(SimpleNamespace(_tome_info={}, track=None, track_res=None), ctx)


def test_lk_fuse_cache_never_serves_a_recycled_address(monkeypatch):
    """the one-entry cache of the latent-knowledge fuse owns its key tensors: 20 freshly allocated, same-shape,
    different-content embeddings (the allocator hands addresses back) all get their own result; the same objects hit.  (The
    fuse itself is a HIP launch since round 5 - tests/test_unet_gpu.py -; here a host stand-in computes a value that depends on
    every input, the cache logic is what is under test.)"""
    from lkgd_amd import lk_fuse as lf
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    m = pu.UNetSpatioTemporalConditionModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    with pytest.raises(lf.LkgdHipError):
        lf.lk_fuse(m, torch.zeros(2, 1, 1024), torch.zeros(1, 1, 1000), torch.zeros(1, 1, 1000))     # no CPU path
    calls = []

    def stand_in(unet, e, d, f):
        calls.append(1)
        return (e * 2.0 + d.sum() - f.sum()).half()
    monkeypatch.setattr(lf, "lk_fuse", stand_in)
    g = torch.Generator().manual_seed(5)
    d, f = torch.randn(1, 1, 1000, generator=g), torch.randn(1, 1, 1000, generator=g)
    for i in range(20):
        e = torch.randn(2, 1, 1024, generator=g)           # fresh tensor, very likely a recycled address
        got = lf.lk_fuse_cached(m, e, d, f)
        assert torch.equal(got, stand_in(m, e, d, f)), i
        n = len(calls)
        assert lf.lk_fuse_cached(m, e, d, f) is got and len(calls) == n          # same objects, unchanged: a hit
        del e, got
    e = torch.randn(2, 1, 1024, generator=g)
    a = lf.lk_fuse_cached(m, e, d, f)
    e.mul_(2.0)                                             # in-place change bumps the version: recompute
    assert not torch.equal(lf.lk_fuse_cached(m, e, d, f), a)


def test_controlnet_condition_cache_owns_its_key():
    from lkgd_amd import controlnet as pc
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    c = pc.ControlNetSDVModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    calls = []

    class _Emb:
        def run_until_out(self, cond):
            calls.append(float(cond.sum()))
            return torch.full((2 * 4 * 8 * 8, 4), float(cond.sum())), 8, 8
    object.__setattr__(c, "controlnet_cond_embedding", _Emb())
    c.__dict__["_modules"].pop("controlnet_cond_embedding", None)
    for i in range(6):
        cond = torch.full((2, 4, 3, 64, 64), float(i))      # fresh same-shape tensor per "call"
        t = c._cond_tokens(cond, 2, 4, 8, 8)
        assert float(t[0, 0]) == float(cond.sum())
        assert c._cond_tokens(cond, 2, 4, 8, 8) is t
        del cond, t
    assert len(calls) == 6


def test_resident_weight_kernel_has_no_register_spills():
    """lkgd_amd/csrc/gemm_resw.hip issues and awaits its epilogue loads by hand (inline asm, counted vmcnt): spill code
    between a load and its wait would save a register whose data has not arrived.  Every instantiation the launcher can
    pick must therefore be allocated without scratch spills inside its 96-register budget (hipcc cross-compiles here)."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    src = os.path.join(REPO, "lkgd_amd", "csrc", "gemm_resw.hip")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    kernels = [(n, s, c) for n, s, c in zip(names, spills, scratch) if "lkgd_gemm_resw_kernel" in n]
    assert len(kernels) >= 30, len(kernels)
    bad = [k for k in kernels if k[1] or k[2]]
    assert not bad, bad


@pytest.mark.parametrize("src,kernel,agprs", [("ff_fused.hip", "ff_fused_kernel", 244), ("attn_spatial_pipe.hip", "attn_pipe_kernel", 108),
                                              ("attn_tblock.hip", "tattn_block_kernel", 244), ("qkv_fused.hip", "ln_qkv_kernel", (84, 164))])
def test_generated_loop_kernels_keep_their_accumulation_registers(src, kernel, agprs, tmp_path):
    """ff_fused.hip / attn_spatial_pipe.hip keep state in accumulation registers ACROSS their asm statements (Y^T and z^T; Q and
    O), which the compiler knows nothing about: it must never place values of its own there.  Checked on the ISA: every
    instantiation allocates exactly the accumulation registers the generated loop names, has no scratch spills, and no
    compiler-generated instruction (outside the #ASMSTART / #ASMEND blocks) touches an accumulation register."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = tmp_path / "k.o"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-mllvm",
                        "-amdgpu-spill-vgpr-to-agpr=0", "--save-temps=obj", "-c", os.path.join(REPO, "lkgd_amd", "csrc", src),
                        "-o", str(out)], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    isa = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert len(isa) == 1
    text = open(tmp_path / isa[0]).read()
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from check_agpr_isa import check          # the same check lkgd_amd/csrc/Makefile runs after compiling these objects
    problems = check(text, kernel, agprs if isinstance(agprs, tuple) else (agprs,))
    assert not problems, problems[:10]
    assert check(text.replace("#ASMEND", "#ASMEND\n\tv_accvgpr_write_b32 a3, v1", 1), kernel,
                 agprs if isinstance(agprs, tuple) else (agprs,)), "the check must notice a compiler write to an a-register"


def test_clip_module_takes_transformers_state_dict_names(tmp_path):
    """lkgd_amd.clip keeps transformers' parameter names (image_encoder/ checkpoints load as they are), round-trips through
    save_pretrained / from_pretrained, ignores the persisted position_ids of older checkpoints, and refuses to run off the GPU"""
    import json
    import pytest
    import torch
    from safetensors.torch import load_file, save_file
    from lkgd_amd._lib import LkgdHipError
    from lkgd_amd.clip import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, patch_size=14,
                           image_size=28, projection_dim=64)
    m = CLIPVisionModelWithProjection(cfg)
    names = set(m.state_dict())
    want = {"vision_model.embeddings.class_embedding", "vision_model.embeddings.patch_embedding.weight",
            "vision_model.embeddings.position_embedding.weight", "vision_model.pre_layrnorm.weight", "vision_model.pre_layrnorm.bias",
            "vision_model.post_layernorm.weight", "vision_model.post_layernorm.bias", "visual_projection.weight"}
    for i in range(2):
        for lin in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "mlp.fc1", "mlp.fc2",
                    "layer_norm1", "layer_norm2"):
            want |= {f"vision_model.encoder.layers.{i}.{lin}.weight", f"vision_model.encoder.layers.{i}.{lin}.bias"}
    assert names == want
    try:
        import transformers
        c = transformers.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                          patch_size=14, image_size=28, projection_dim=64, hidden_act="gelu")
        assert set(transformers.CLIPVisionModelWithProjection(c).state_dict()) == names
    except ImportError:
        pass
    d = tmp_path / "image_encoder"
    m.save_pretrained(str(d))
    sd = load_file(str(d / "model.safetensors"))
    sd["vision_model.embeddings.position_ids"] = torch.arange(5)[None]
    save_file(sd, str(d / "model.safetensors"))
    assert json.load(open(d / "config.json"))["hidden_size"] == 128
    m2 = CLIPVisionModelWithProjection.from_pretrained(str(d), torch_dtype=torch.float32)
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v)
    with pytest.raises(LkgdHipError):
        m2(torch.zeros(1, 3, 28, 28))
    with pytest.raises(LkgdHipError):
        CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=128, num_attention_heads=2, hidden_act="relu"))
