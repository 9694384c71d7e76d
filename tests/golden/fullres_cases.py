"""Seeded inputs of the full-resolution / 25-step fixtures: the generator script (make_goldens.py) and the tests rebuild
inputs, tracks and hook weights from here, so the fixtures hold expected outputs (and checksums) only."""
import torch

FULLRES_SEED = 131        # seeds the INPUTS of the full-resolution fixtures (the stock UNet's weights are the c1 fixtures': seed 31)
FULLRES_LK_SEED = 141     # LKGD UNet weights of the full-resolution LK + FSM-hook forward
LOOP25_SEED = 151         # tiny-width 25-step loop
F14_SEED = 161            # inputs of the 14-frame real-width loop fixtures (round 3)
#: latent geometry of BASELINE.json configs[1] with 2 of the 14 frames: CFG batch 2 x 2 frames x 72 x 128 (S = 9216)
FULLRES_GEOM = dict(B=2, F=2, H=72, W=128)


FULLRES_F14_SEED = 171    # inputs of the ONE-forward fixture at the headline geometry itself: CFG 2 x 14 frames x 72 x 128 (round 4)


HEADLINE_SEED = 181       # inputs of the 25-step loop fixture at the headline geometry: 1 clip x 14 frames x 576 x 1024 px (round 5)


def headline_inputs():
    """(image [1,3,576,1024] in [0,1], initial latents [1,14,4,72,128] rounded to fp16) of `loop25_headline.safetensors`"""
    g = torch.Generator().manual_seed(HEADLINE_SEED)
    image = torch.rand(1, 3, 576, 1024, generator=g)
    lat0 = torch.randn(1, 14, 4, 72, 128, generator=g).half().float()
    return image, lat0


def fullres_inputs(seed=FULLRES_SEED + 1, lk=False, frames=None):
    g = torch.Generator().manual_seed(seed)
    B, F, H, W = (FULLRES_GEOM[k] for k in "BFHW")
    F = frames or F
    d = {"sample": torch.randn(B, F, 8, H, W, generator=g).half().float(), "t": torch.tensor(0.875),
         "enc": torch.randn(B, 1, 1024, generator=g).half().float(), "ids": torch.tensor([[6.0, 127.0, 0.02]] * B)}
    if lk:
        d["domain"] = torch.randn(1, 1, 1000, generator=g)
        d["flow"] = torch.randn(1, 1, 1000, generator=g)
    return d


def fullres_tracks(seed=FULLRES_LK_SEED + 2, points=2048):
    """(src, dst, visibility) point tracks on a 144 x 256 grid (2x the 72 x 128 latent) for the B*F/2 = 2 (src, dst)
    image pairs of the batch; dst = src + a bounded displacement and may leave the image (the hook clamps it)"""
    g = torch.Generator().manual_seed(seed)
    B, F, H, W = (FULLRES_GEOM[k] for k in "BFHW")
    pairs = B * F // 2
    res = (2 * H, 2 * W)
    src = torch.stack([torch.randint(0, res[1], (pairs, points), generator=g),
                       torch.randint(0, res[0], (pairs, points), generator=g)], -1).float()
    dst = (src + torch.randint(-9, 10, (pairs, points, 2), generator=g)).float()
    vis = (torch.rand(pairs, points, generator=g) > 0.2).float()
    return (src, dst, vis), res


def seed_conv_fuse_(blocks, seed=FULLRES_LK_SEED + 3):
    """non-zero weights for the (zero-initialised) `conv_fuse` of every hooked block, in module order, rounded to fp16"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for b in blocks:
            w, bias = b.conv_fuse.weight, b.conv_fuse.bias
            w.copy_((torch.randn(w.shape, generator=g) / w[0].numel() ** 0.5).half().float())
            bias.copy_((0.1 * torch.randn(bias.shape, generator=g)).half().float())
