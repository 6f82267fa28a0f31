"""Image normalization constants of the encoders on this path (reference: encoders/image_normalizations.py)."""
from dataclasses import dataclass

import torch


@dataclass
class ImageNormalization:
    mean: torch.Tensor
    std: torch.Tensor


_IMAGENET = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
_TABLE = {
    "dummy": ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0]),
    "identity": ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0]),
    "croco": _IMAGENET,
    "dust3r": ([0.5, 0.5, 0.5], [0.5, 0.5, 0.5]),
    "dinov2": _IMAGENET,
}
IMAGE_NORMALIZATION_DICT = {k: ImageNormalization(mean=torch.tensor(m), std=torch.tensor(s)) for k, (m, s) in _TABLE.items()}
