"""Clip-level image pre-processing: the anti-aliased resize in front of CLIP and the PIL / numpy
conversions around the loop.  CPU part: oracle vs outputs of the reference's own `_resize_with_antialiasing`
(tests/golden/image_ops.safetensors), host conversions.  GPU part: the HIP kernels vs the same goldens, through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from image_cases import IMAGE_CASES, image_case_input   # noqa: E402


@pytest.fixture(scope="module")
def golden(golden_dir):
    return load_file(os.path.join(golden_dir, "image_ops.safetensors"))


def test_oracle_resize_vs_reference_golden(golden):
    from oracle import image_ops as oi
    for i, (name, shape, size, sub) in enumerate(IMAGE_CASES):
        y = oi.resize_with_antialiasing(image_case_input(shape, 900 + i), size)
        assert y.shape[-2:] == size
        torch.testing.assert_close(y[..., ::sub, ::sub], golden[name], rtol=1e-6, atol=1e-6)
        assert abs(y.mean().item() - golden[name + "_mean"].item()) < 1e-6
    # geometry the reference derives for the SVD conditioning frame: factors (2.571, 4.571) -> sigma, odd tap counts
    sig, ks = oi.blur_geometry((576, 1024), (224, 224))
    assert ks == (3, 7) and abs(sig[0] - 0.7857142857) < 1e-9 and abs(sig[1] - 1.7857142857) < 1e-9
    assert abs(oi.gaussian_taps(7, sig[1]).sum().item() - 1.0) < 1e-6


def test_image_processor_conversions():
    import PIL.Image
    from lkgd_amd.image_processor import VaeImageProcessor, tensor2vid
    p = VaeImageProcessor(vae_scale_factor=8)
    rng = np.random.RandomState(0)
    arr = rng.randint(0, 256, size=(37, 53, 3), dtype=np.uint8)
    img = PIL.Image.fromarray(arr)
    x = p.numpy_to_pt(p.pil_to_numpy(img))
    assert x.shape == (1, 3, 37, 53) and x.dtype == torch.float32
    assert torch.equal((x * 255).round().to(torch.uint8)[0].permute(1, 2, 0), torch.from_numpy(arr))
    # preprocess: PIL is resized to (height, width) rounded down to the VAE factor and normalised to [-1, 1]
    y = p.preprocess(img, height=32, width=48)
    assert y.shape == (1, 3, 32, 48) and -1.0 <= y.min() and y.max() <= 1.0 and y.min() < 0
    y = p.preprocess(img)
    assert y.shape == (1, 3, 32, 48)
    # tensors in [0, 1] are normalised, tensors that already hold negatives are passed through (with the reference's warning)
    t = torch.rand(2, 3, 16, 24)
    torch.testing.assert_close(p.preprocess(t, 16, 24), 2 * t - 1)
    with pytest.warns(FutureWarning):
        torch.testing.assert_close(p.preprocess(2 * t - 1, 16, 24), 2 * t - 1)
    # postprocess / tensor2vid: [-1,1] video [B,C,F,H,W] -> frames
    vid = torch.rand(2, 3, 4, 8, 8) * 2 - 1
    out_np = tensor2vid(vid, p, "np")
    assert out_np.shape == (2, 4, 8, 8, 3) and out_np.min() >= 0 and out_np.max() <= 1
    out_pt = tensor2vid(vid, p, "pt")
    torch.testing.assert_close(out_pt, (vid / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4))
    out_pil = tensor2vid(vid, p, "pil")
    assert len(out_pil) == 2 and len(out_pil[0]) == 4 and out_pil[0][0].size == (8, 8)
    back = p.pil_to_numpy(out_pil[1])
    assert np.abs(back - out_np[1]).max() <= 0.5 / 255 + 1e-6
    with pytest.raises(ValueError):
        tensor2vid(vid, p, "jpeg")


@pytest.mark.gpu
def test_hip_resize_vs_reference_golden(golden):
    from lkgd_amd.image_ops import resize_with_antialiasing
    for i, (name, shape, size, sub) in enumerate(IMAGE_CASES):
        x = image_case_input(shape, 900 + i)
        y = resize_with_antialiasing(x.cuda(), size)
        assert y.is_cuda and y.dtype == torch.float32 and tuple(y.shape[-2:]) == size
        # fp32 on both sides; tolerance = summation-order / fma noise on values in [-1, 1] (measured max 7e-6)
        torch.testing.assert_close(y.cpu()[..., ::sub, ::sub], golden[name], rtol=1e-4, atol=2e-5)
        assert abs(y.mean().item() - golden[name + "_mean"].item()) < 1e-6
        # a CPU tensor comes back on the CPU (the reference calls it on CPU tensors): same values
        y2 = resize_with_antialiasing(x, size)
        assert not y2.is_cuda and torch.equal(y2, y.cpu())


@pytest.mark.gpu
def test_hip_image_kernels_vs_torch():
    """the two kernels on their own: reflect-padded 1-D filter (even / odd tap counts, both axes), bicubic align_corners"""
    import torch.nn.functional as F
    from lkgd_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 19, 23, generator=g)
    for k in (1, 3, 4, 7):
        taps = torch.rand(k, generator=g)
        front, rear = (k - 1) // 2, k - 1 - (k - 1) // 2
        ref_w = F.conv2d(F.pad(x, (front, rear, 0, 0), mode="reflect").reshape(6, 1, 19, -1), taps.reshape(1, 1, 1, k))
        ref_h = F.conv2d(F.pad(x, (0, 0, front, rear), mode="reflect").reshape(6, 1, -1, 23), taps.reshape(1, 1, k, 1))
        torch.testing.assert_close(ops.conv1d_reflect(x.cuda(), taps.cuda(), 1).cpu(), ref_w.reshape(x.shape), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(ops.conv1d_reflect(x.cuda(), taps.cuda(), 0).cpu(), ref_h.reshape(x.shape), rtol=1e-5, atol=1e-5)
    for size in ((7, 5), (19, 23), (40, 31), (1, 1)):
        ref = F.interpolate(x, size=size, mode="bicubic", align_corners=True)
        torch.testing.assert_close(ops.resize_bicubic_ac(x.cuda(), *size).cpu(), ref, rtol=1e-5, atol=1e-5)
    from lkgd_amd._lib import LkgdHipError
    with pytest.raises(LkgdHipError):
        ops.conv1d_reflect(x.cuda(), torch.rand(60).cuda(), 1)      # reflect padding wider than the image
