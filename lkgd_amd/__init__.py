"""lkgd_amd - MI355X-native (gfx950) implementation of the SVD / LKGD denoising hot path.

Host side mirrors the reference's Python interface (UNet.forward, scheduler, pipeline.__call__, patch API); every
tensor op of the per-step UNet forward runs in the hand-written HIP kernels of ``lkgd_amd/csrc`` through the C-ABI in
``include/lkgd_hip.h``.  There is no CPU or eager-PyTorch fallback.
"""
from ._lib import LkgdHipError, lib  # noqa: F401

__all__ = ["LkgdHipError", "lib"]
