"""Seeded inputs of the image-op fixtures (tests/golden/image_ops.safetensors): the generator script and the tests both
rebuild the inputs from here, so only the expected outputs are stored."""
import torch

#: (fixture key, input shape, output size, stored subsampling step)
IMAGE_CASES = (("down_64x96_to_24", (1, 3, 64, 96), (24, 24), 1), ("up_50x70_to_96", (2, 3, 50, 70), (96, 96), 1),
               ("down_144x256_to_56", (1, 3, 144, 256), (56, 56), 1), ("svd_576x1024_to_224", (1, 3, 576, 1024), (224, 224), 4))


def image_case_input(shape, seed):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * 2.0 - 1.0
