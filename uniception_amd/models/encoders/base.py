"""Encoder base classes and dataclasses (reference: encoders/base.py:14-152)."""
from dataclasses import dataclass
from typing import Optional

import torch.nn as nn
from torch import Tensor


@dataclass
class EncoderInput:
    "Data class for Encoder Input"
    data_norm_type: str


@dataclass
class EncoderOutput:
    "Data class for Encoder Output"
    pass


@dataclass
class EncoderGlobalRepInput:
    data: Tensor  # [batch, channel]


@dataclass
class EncoderGlobalRepOutput:
    features: Tensor  # [batch, enc_embed_dim]


class UniCeptionEncoderBase(nn.Module):
    "Encoder Base Class"

    def __init__(self, name: str, data_norm_type: str, size: Optional[str] = None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.name: str = name
        self.size: Optional[str] = size
        self.data_norm_type: str = data_norm_type

    def forward(self, encoder_input: EncoderInput) -> EncoderOutput:
        raise NotImplementedError

    def _check_data_normalization_type(self, data_norm_type: str):
        assert (
            data_norm_type == self.data_norm_type
        ), f"Input normalization type {data_norm_type} does not match the encoder's normalization type {self.data_norm_type}."


@dataclass
class ViTEncoderInput(EncoderInput):
    image: Tensor  # [batch, channel, height, width]


@dataclass
class ViTEncoderNonImageInput:
    data: Tensor  # [batch, channel, height, width]


@dataclass
class ViTEncoderOutput(EncoderOutput):
    features: Tensor  # [batch, enc_embed_dim, feat_height, feat_width]
    registers: Optional[Tensor] = None


class UniCeptionViTEncoderBase(UniCeptionEncoderBase):
    "Vision Transformer Encoder Base Class"

    def __init__(self, patch_size: int, gradient_checkpointing: bool = False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.patch_size = patch_size
        self.gradient_checkpointing = gradient_checkpointing

    def wrap_module_with_gradient_checkpointing(self, module: nn.Module):
        "Re-compute `module`'s forward in the backward pass (encoders/base.py:139-152); see models/utils/checkpointing.py."
        from ..utils.checkpointing import wrap_module_with_gradient_checkpointing
        return wrap_module_with_gradient_checkpointing(module)
