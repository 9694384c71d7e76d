"""Data-format conversions either side of the loop (PIL / numpy <-> tensors): the subset of diffusers'
`VaeImageProcessor` **[EXT diffusers 0.27.2 `image_processor.py`, not vendored by the reference]** that the reference's
pipeline calls - `pil_to_numpy` / `numpy_to_pt` (pipeline_stable_video_diffusion_trans.py:168-169), `preprocess(image,
height, width)` (:435) and, through `tensor2vid` (:79-98), `postprocess(video, output_type)`.  Host-side numpy / PIL code:
nothing here touches the GPU.  Restated from the published behaviour of the pinned version (parity unpinned: diffusers is
not installable in the build container); defaults are the class defaults the reference relies on (do_resize, lanczos for
PIL, do_normalize to [-1, 1])."""
from __future__ import annotations

import warnings
from typing import List, Optional, Union

import numpy as np
import torch

try:                                    # PIL is optional: tensor / numpy inputs work without it
    import PIL.Image
    _LANCZOS = getattr(getattr(PIL.Image, "Resampling", PIL.Image), "LANCZOS")
except Exception:                       # pragma: no cover
    PIL = None


class VaeImageProcessor:
    def __init__(self, do_resize: bool = True, vae_scale_factor: int = 8, resample: str = "lanczos",
                 do_normalize: bool = True):
        if resample != "lanczos":
            raise NotImplementedError("only the default PIL resampling (lanczos) is implemented")
        self.config = type("Config", (), dict(do_resize=do_resize, vae_scale_factor=vae_scale_factor,
                                              resample=resample, do_normalize=do_normalize))()

    # ---- elementary conversions ---------------------------------------------------------------------------------
    @staticmethod
    def numpy_to_pil(images: np.ndarray) -> list:
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        if images.shape[-1] == 1:
            return [PIL.Image.fromarray(im.squeeze(), mode="L") for im in images]
        return [PIL.Image.fromarray(im) for im in images]

    @staticmethod
    def pil_to_numpy(images) -> np.ndarray:
        if not isinstance(images, list):
            images = [images]
        return np.stack([np.array(im).astype(np.float32) / 255.0 for im in images], axis=0)

    @staticmethod
    def numpy_to_pt(images: np.ndarray) -> torch.Tensor:
        if images.ndim == 3:
            images = images[..., None]
        return torch.from_numpy(images.transpose(0, 3, 1, 2))

    @staticmethod
    def pt_to_numpy(images: torch.Tensor) -> np.ndarray:
        return images.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def normalize(images):
        return 2.0 * images - 1.0

    @staticmethod
    def denormalize(images):
        return (images / 2 + 0.5).clamp(0, 1)

    # ---- preprocess / postprocess -------------------------------------------------------------------------------
    def get_default_height_width(self, image, height: Optional[int] = None, width: Optional[int] = None):
        if height is None:
            height = image.height if PIL is not None and isinstance(image, PIL.Image.Image) else (
                image.shape[2] if isinstance(image, torch.Tensor) else image.shape[1])
        if width is None:
            width = image.width if PIL is not None and isinstance(image, PIL.Image.Image) else (
                image.shape[3] if isinstance(image, torch.Tensor) else image.shape[2])
        f = self.config.vae_scale_factor
        return height - height % f, width - width % f

    def resize(self, image, height: int, width: int):
        if PIL is not None and isinstance(image, PIL.Image.Image):
            return image.resize((width, height), resample=_LANCZOS)
        if isinstance(image, torch.Tensor):
            return torch.nn.functional.interpolate(image, size=(height, width))
        pt = torch.nn.functional.interpolate(self.numpy_to_pt(image), size=(height, width))
        return self.pt_to_numpy(pt)

    def preprocess(self, image, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        is_pil = PIL is not None and isinstance(image, PIL.Image.Image)
        if is_pil or isinstance(image, (np.ndarray, torch.Tensor)):
            image = [image]
        if not isinstance(image, list) or not image:
            raise ValueError("image must be a PIL image, numpy array, torch tensor or a list of them")
        if PIL is not None and isinstance(image[0], PIL.Image.Image):
            if self.config.do_resize:
                height, width = self.get_default_height_width(image[0], height, width)
                image = [self.resize(i, height, width) for i in image]
            image = self.numpy_to_pt(self.pil_to_numpy(image))
        elif isinstance(image[0], np.ndarray):
            image = np.concatenate(image, axis=0) if image[0].ndim == 4 else np.stack(image, axis=0)
            image = self.numpy_to_pt(image)
            height, width = self.get_default_height_width(image, height, width)
            if self.config.do_resize:
                image = self.resize(image, height, width)
        elif isinstance(image[0], torch.Tensor):
            image = torch.cat(image, axis=0) if image[0].ndim == 4 else torch.stack(image, axis=0)
            height, width = self.get_default_height_width(image, height, width)
            if self.config.do_resize:
                image = self.resize(image, height, width)
        else:
            raise ValueError(f"unsupported image type {type(image[0])}")
        do_normalize = self.config.do_normalize
        if do_normalize and image.min() < 0:
            warnings.warn("Passing `image` as torch tensor with value range in [-1,1] is deprecated. The expected value "
                          "range for image tensor is [0,1] when passing as pytorch tensor or numpy Array.", FutureWarning)
            do_normalize = False
        return self.normalize(image) if do_normalize else image

    def postprocess(self, image: torch.Tensor, output_type: str = "pil", do_denormalize: Optional[List[bool]] = None):
        if not isinstance(image, torch.Tensor):
            raise ValueError("postprocess expects a torch tensor")
        if output_type not in ("latent", "pt", "np", "pil"):
            output_type = "np"
        if output_type == "latent":
            return image
        if do_denormalize is None:
            do_denormalize = [self.config.do_normalize] * image.shape[0]
        image = torch.stack([self.denormalize(image[i]) if do_denormalize[i] else image[i] for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        image = self.pt_to_numpy(image)
        if output_type == "np":
            return image
        return self.numpy_to_pil(image)


def tensor2vid(video: torch.Tensor, processor: VaeImageProcessor, output_type: str = "np"):
    """reference pipeline_stable_video_diffusion_trans.py:79-98: [B,C,F,H,W] -> per-clip lists of frames"""
    outputs = []
    for b in range(video.shape[0]):
        outputs.append(processor.postprocess(video[b].permute(1, 0, 2, 3), output_type))
    if output_type == "np":
        return np.stack(outputs)
    if output_type == "pt":
        return torch.stack(outputs)
    if output_type != "pil":
        raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil']")
    return outputs
