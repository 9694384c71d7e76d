"""The SVD VAE (AutoencoderKLTemporalDecoder) on the HIP path against the fp32 oracle restatement
(oracle/vae.py; [EXT] diffusers 0.27.2 - PARITY UNPINNED, nothing under /root/reference holds VAE code or fixtures), the
two VAE kernels and the encoder's asymmetric convolution against plain PyTorch fp32, `from_pretrained` / `save_pretrained`
round trips and the `run_inference_svd` harness end to end on a tiny random pipeline directory."""
import os

import pytest
import torch

DEV = "cuda:0"


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _pair(cfg_o, seed):
    from lkgd_amd import vae as pv
    from oracle import vae as ov
    o = ov.init_weights_(ov.AutoencoderKLTemporalDecoder(cfg_o), seed)
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.half().float())
    m = pv.AutoencoderKLTemporalDecoder(pv.VAEConfig(**{k: v for k, v in cfg_o.__dict__.items()}))
    m.load_state_dict(o.state_dict())
    return o, m


def test_vae_parameter_tree_matches_diffusers_names_and_count():
    """structural gate: the SVD VAE is 97 742 847 parameters (34 163 592 encoder / 63 579 183 decoder / 72 quant_conv); the
    HIP module and the oracle share diffusers' parameter names, so a real `vae/` checkpoint loads into both"""
    from lkgd_amd import vae as pv
    from oracle import vae as ov
    with torch.device("meta"):
        m, o = pv.AutoencoderKLTemporalDecoder(), ov.AutoencoderKLTemporalDecoder(ov.SVD_VAE_CONFIG)
    sm, so = {k: tuple(v.shape) for k, v in m.state_dict().items()}, {k: tuple(v.shape) for k, v in o.state_dict().items()}
    assert sm == so
    assert sum(p.numel() for p in m.parameters()) == 97742847
    assert sum(p.numel() for p in m.encoder.parameters()) == 34163592
    for k in ("encoder.down_blocks.0.resnets.1.conv2.weight", "encoder.down_blocks.1.resnets.0.conv_shortcut.weight",
              "encoder.down_blocks.2.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.to_out.0.bias",
              "encoder.mid_block.attentions.0.group_norm.weight", "quant_conv.weight",
              "decoder.mid_block.resnets.1.temporal_res_block.conv2.weight", "decoder.mid_block.resnets.0.time_mixer.mix_factor",
              "decoder.up_blocks.3.resnets.2.spatial_res_block.norm1.bias", "decoder.up_blocks.0.upsamplers.0.conv.bias",
              "decoder.up_blocks.2.resnets.0.spatial_res_block.conv_shortcut.weight", "decoder.time_conv_out.weight"):
        assert k in sm, k
    assert not any(k.startswith("decoder.up_blocks.3.upsamplers") or k.startswith("post_quant") for k in sm)


def test_from_pretrained_round_trips_and_pipeline_directory(tmp_path):
    """diffusers directory layout written and read back without diffusers: UNet / VAE / scheduler / pipeline"""
    from lkgd_amd import run_inference_svd as rs
    from lkgd_amd import unet as pu
    from lkgd_amd import vae as pv
    from lkgd_amd.pipeline import StableVideoDiffusionPipeline
    from lkgd_amd.scheduler import EulerDiscreteScheduler
    d = rs.make_tiny_pipeline_dir(str(tmp_path / "svd"))
    assert sorted(os.listdir(d)) == ["feature_extractor", "image_encoder", "scheduler", "unet", "vae"]
    u = pu.UNetSpatioTemporalConditionControlNetModel.from_pretrained(d, subfolder="unet", torch_dtype=torch.float16)
    assert u.config.block_out_channels == (64, 128, 128, 128) and u.dtype == torch.float16
    u.save_pretrained(str(tmp_path / "again"), variant="fp16")
    u2 = pu.UNetSpatioTemporalConditionControlNetModel.from_pretrained(str(tmp_path / "again"), variant="fp16")
    assert all(torch.equal(a, b) for a, b in zip(u.state_dict().values(), u2.state_dict().values()))
    v = pv.AutoencoderKLTemporalDecoder.from_pretrained(d, subfolder="vae")
    assert v.config.force_upcast is True and v.config.scaling_factor == 0.18215 and v.config.layers_per_block == 1
    s = EulerDiscreteScheduler.from_pretrained(d, subfolder="scheduler")
    assert s.config.prediction_type == "v_prediction" and s.config.sigma_max == 700.0 and s.config.use_karras_sigmas
    pipe = StableVideoDiffusionPipeline.from_pretrained(d, torch_dtype=torch.float16, low_cpu_mem_usage=False,
                                                        device_map=None, device=None)
    assert pipe.vae is not None and pipe.image_encoder is not None and pipe.feature_extractor is not None
    assert pipe.vae_scale_factor == 8
    with pytest.raises(OSError):
        pu.UNetSpatioTemporalConditionControlNetModel.from_pretrained(str(tmp_path / "nowhere"))
    # a checkpoint with a missing / unexpected key is an error, not a silent partial load
    from safetensors.torch import load_file, save_file
    f = os.path.join(d, "unet", "diffusion_pytorch_model.safetensors")
    sd = load_file(f)
    sd.pop("conv_in.bias")
    save_file(sd, f)
    with pytest.raises(RuntimeError):
        pu.UNetSpatioTemporalConditionControlNetModel.from_pretrained(d, subfolder="unet")


def test_harness_helpers(tmp_path):
    """run_inference_svd.py:183-207 (LoRA file copied into matching UNet parameters by name) and utils/util.py:791-859"""
    from PIL import Image
    from safetensors.torch import save_file
    from lkgd_amd import run_inference_svd as rs
    from lkgd_amd import unet as pu
    from oracle import unet as ou
    u = pu.UNetSpatioTemporalConditionControlNetModel(pu.UNetConfig(**ou.TINY_CONFIG.__dict__))
    k = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    new = torch.full_like(u.state_dict()[k], 0.25)
    save_file({"unet." + k: new, "unet.not_a_parameter.lora_A.weight": torch.zeros(2, 2)}, str(tmp_path / "l.safetensors"))
    assert rs.update_unet_from_lora_file(u, str(tmp_path / "l.safetensors")) == [k]
    assert torch.equal(u.state_dict()[k], new)
    vids = [[Image.new("RGB", (16, 12), (i * 40, 0, 0)) for i in range(3)], [torch.rand(3, 12, 16) for _ in range(4)]]
    out = rs.save_gifs_side_by_side(vids, str(tmp_path / "gifs"))
    g = Image.open(out)
    assert g.n_frames == 3 and g.size == (16, 24) and os.listdir(str(tmp_path / "gifs")) == [os.path.basename(out)]


@pytest.mark.gpu
def test_vae_kernels_vs_torch():
    """softmax rows, time_conv_out, and the pad_off (F.pad(0,1,0,1) + stride-2, padding-0) convolution vs fp32 PyTorch"""
    import torch.nn.functional as F
    from lkgd_amd import ops
    from lkgd_amd.packing import pack_conv3x3
    g = torch.Generator().manual_seed(1)
    for rows, cols in ((64, 96), (300, 9216), (5, 16384)):
        x = (3.0 * torch.randn(rows, cols, generator=g)).half().to(DEV)
        y = ops.softmax_rows(x.clone())
        ref = torch.softmax(x.float(), dim=1)
        assert (y.float() - ref).abs().max() < 2e-3 and abs(float(y.float().sum(1).mean()) - 1.0) < 2e-3
    xw = torch.empty(40, 128, dtype=torch.float16, device=DEV).normal_()        # strided view: ld 128, 96 columns
    y = ops.softmax_rows(xw[:, :96], torch.empty(40, 96, dtype=torch.float16, device=DEV))
    assert (y.float() - torch.softmax(xw[:, :96].float(), 1)).abs().max() < 2e-3
    # time_conv_out: Conv3d (3,1,1) on 3 channels, B = 2 chunks of F = 5 frames
    B, Fr, H, W = 2, 5, 6, 10
    tok = torch.randn(B * Fr * H * W, 8, generator=g).half()
    w, b = torch.randn(3, 3, 3, 1, 1, generator=g), torch.randn(3, generator=g)
    x5 = tok[:, :3].float().reshape(B, Fr, H, W, 3).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(x5, w, b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(B * Fr, 3, H, W)
    for dt in (torch.float32, torch.float16):
        out = ops.time_conv_out(tok.to(DEV), w.reshape(3, 3, 3).contiguous().to(DEV), b.to(DEV), B, Fr, H, W, dtype=dt)
        assert out.dtype == dt and _rel(out, ref) < 2e-3
    # asymmetric downsample
    N, C_, H, W = 2, 64, 10, 14
    x = torch.randn(N, C_, H, W, generator=g).half()
    wt, bs = (torch.randn(C_, C_, 3, 3, generator=g) / 24).half(), torch.randn(C_, generator=g)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), wt.float(), bs, stride=2)
    Ho, Wo = ref.shape[2:]
    out = torch.empty(N * Ho * Wo, C_, dtype=torch.float16, device=DEV)
    ops.gemm(ops.nchw_to_tokens(x.to(DEV)), pack_conv3x3(wt).to(DEV), out, M=N * Ho * Wo, N=C_, K=9 * C_, bias=bs.to(DEV),
             mode=ops.A_CONV3X3, Cin=C_, conv=(Ho, Wo, H, W, 2, 0, 1))
    assert _rel(ops.tokens_to_nchw(out, N, C_, Ho, Wo), ref) < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(64, 64), (64, 96)])
def test_vae_tiny_encode_decode_vs_oracle(hw):
    from oracle import vae as ov
    o, m = _pair(ov.TINY_VAE_CONFIG, 11)
    m = m.half().to(DEV)
    g = torch.Generator().manual_seed(12)
    H, W = hw
    x = (torch.rand(2, 3, H, W, generator=g) * 2 - 1).half().float()
    with torch.no_grad():
        ref = o.encode(x).latent_dist.mode()
    got = m.encode(x.to(DEV)).latent_dist.mode()
    assert got.shape == ref.shape == (2, 4, H // 8, W // 8) and got.dtype == torch.float32
    assert _rel(got, ref) < 1e-2, _rel(got, ref)
    z = torch.randn(2 * 3, 4, H // 8, W // 8, generator=g).half().float()       # two chunks of 3 frames
    with torch.no_grad():
        ref = o.decode(z, num_frames=3).sample
    got = m.decode(z.half().to(DEV), num_frames=3).sample
    assert got.shape == ref.shape == (6, 3, H, W) and got.dtype == torch.float16
    r = _rel(got, ref)
    assert r < 1e-2 and (got.float().cpu() - ref).abs().max() < 5e-2 * max(1.0, float(ref.abs().max())), r
    # the temporal statistics depend on the chunking (decode_chunk_size), so must the HIP path
    with torch.no_grad():
        ref1 = o.decode(z, num_frames=1).sample
    assert _rel(m.decode(z.half().to(DEV), num_frames=1).sample, ref1) < 1e-2 and _rel(ref1, ref) > 1e-2


@pytest.mark.gpu
def test_vae_real_width_decode_and_encode_vs_oracle():
    """the SVD VAE's channel widths (128, 256, 512, 512; one 512-dim attention head) on a small image: 2 frames of 96x128
    pixels (latent 12x16: S = 192, not a multiple of 64 -> the padded PV product)"""
    from oracle import vae as ov
    o, m = _pair(ov.SVD_VAE_CONFIG, 21)
    m = m.half().to(DEV)
    g = torch.Generator().manual_seed(22)
    z = torch.randn(2, 4, 12, 16, generator=g).half().float()
    with torch.no_grad():
        ref = o.decode(z, num_frames=2).sample
    got = m.decode(z.half().to(DEV), num_frames=2).sample
    r = _rel(got, ref)
    print(f"\nreal-width VAE decode vs oracle: rel L2 {r:.3e}")
    assert got.shape == (2, 3, 96, 128) and r < 1e-2
    x = (torch.rand(1, 3, 96, 128, generator=g) * 2 - 1).half().float()
    with torch.no_grad():
        ref = o.encode(x).latent_dist.mode()
    r = _rel(m.encode(x.to(DEV)).latent_dist.mode(), ref)
    print(f"real-width VAE encode vs oracle: rel L2 {r:.3e}")
    assert r < 1e-2


@pytest.mark.gpu
def test_run_inference_svd_harness_end_to_end(tmp_path):
    """from_pretrained -> PIL image in -> CLIP (transformers) + VAE encode (HIP) -> Euler loop (HIP) -> temporal VAE decode
    (HIP) -> PIL frames -> GIF, through the harness with the reference's knobs"""
    from PIL import Image
    from lkgd_amd import run_inference_svd as rs
    out = str(tmp_path / "out")
    rs.main(["--random_init_tiny", "--output_dir", out, "--num_frames", "4", "--num_inference_steps", "2",
             "--decode_chunk_size", "2", "--seed", "7"])
    gifs = [f for f in os.listdir(os.path.join(out, "validation_images")) if f.endswith(".gif")]
    assert len(gifs) == 1
    g = Image.open(os.path.join(out, "validation_images", gifs[0]))
    assert g.n_frames == 4 and g.size == (64, 64)


def test_flow_latent_constants_and_round_trip():
    """utils/optical_flow.py:62-77"""
    from lkgd_amd import optical_flow as of
    assert of.FLOW_LATENT_MEAN == 0.5020191669464111 and of.FLOW_LATENT_STD == 1.2818458080291748
    x = torch.randn(2, 3, 4)
    torch.testing.assert_close(of.optical_flow_latent_unnormalize(of.optical_flow_latent_normalize(x)), x)
    torch.testing.assert_close(of.optical_flow_latent_normalize(x, scale=4.0),
                               ((x * 4.0 - of.FLOW_LATENT_MEAN) / of.FLOW_LATENT_STD) / 4.0)
    assert of.optical_flow_latent_unnormalize(torch.zeros(1, dtype=torch.float16)).dtype == torch.float16


@pytest.mark.gpu
def test_flow_pipeline_unnormalises_before_decode_and_ignores_the_condition(tmp_path):
    """pipeline_stable_video_diffusion_controlnet_flow.py as the reference runs it (configs[3]): the vanilla loop, the
    condition argument unused, flow latents un-normalised before the temporal VAE decode (:641)"""
    from lkgd_amd import optical_flow as of
    from lkgd_amd import run_inference_svd as rs
    from lkgd_amd.pipeline import StableVideoDiffusionPipeline, StableVideoDiffusionPipelineControlNetFlow
    d = rs.make_tiny_pipeline_dir(str(tmp_path / "svd"))
    base = StableVideoDiffusionPipeline.from_pretrained(d)
    flow = StableVideoDiffusionPipelineControlNetFlow.from_pretrained(d)
    import numpy as np
    from PIL import Image
    image = Image.fromarray(np.random.RandomState(3).randint(0, 256, (64, 64, 3), dtype=np.uint8))   # PIL: the CLIP branch resizes
    kw = dict(height=64, width=64, num_frames=4, num_inference_steps=2)
    cond = torch.rand(4, 3, 64, 64)
    lat = base(image, output_type="latent", generator=torch.Generator().manual_seed(4), **kw).frames
    lat_f = flow(image, cond, output_type="latent", generator=torch.Generator().manual_seed(4), **kw).frames
    assert torch.equal(lat, lat_f)                                   # same loop, condition unused
    got = flow(image, cond, output_type="pt", generator=torch.Generator().manual_seed(4), **kw).frames
    z = of.optical_flow_latent_unnormalize(lat).flatten(0, 1) / base.vae.config.scaling_factor
    ref = base.vae.decode(z, num_frames=4).sample.float()
    ref = ((ref / 2 + 0.5).clamp(0, 1)).reshape(1, 4, 3, 64, 64)
    assert got.shape == ref.shape and (got.float().cpu() - ref.cpu()).abs().max() < 2e-3
    plain = base(image, output_type="pt", generator=torch.Generator().manual_seed(4), **kw).frames
    assert (plain.float() - got.float()).abs().max() > 1e-2           # the un-normalisation changes the decoded frames
