"""Prediction-head dataclasses and base classes used on the pointmap path (reference: prediction_heads/base.py:14-170)."""
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch.nn as nn
from torch import Tensor


@dataclass
class PredictionHeadInput:
    last_feature: Tensor  # [batch, feat_dim, feat_height, feat_width]


@dataclass
class PredictionHeadLayeredInput:
    list_features: List[Tensor]
    target_output_shape: Tuple[int, int]


@dataclass
class PredictionHeadTokenInput:
    last_feature: Tensor  # [batch, feat_dim, num_tokens]


@dataclass
class PixelTaskOutput:
    "Dense BCHW output at the input image resolution."
    decoded_channels: Tensor


@dataclass
class SummaryTaskOutput:
    decoded_channels: Tensor  # [batch, channels]


@dataclass
class AdaptorInput:
    adaptor_feature: Tensor  # [batch, sliced_channels, height, width]
    output_shape_hw: Tuple[int, int]


@dataclass
class AdaptorOutput:
    value: Tensor


@dataclass
class PredictionHeadOutput:
    adaptor_output: Dict[str, AdaptorOutput]


@dataclass
class RegressionAdaptorOutput:
    value: Tensor


@dataclass
class RegressionWithConfidenceAdaptorOutput:
    value: Tensor
    confidence: Tensor


@dataclass
class MaskAdaptorOutput:
    logits: Tensor
    mask: Tensor


@dataclass
class Covariance2DAdaptorOutput:
    covariance: Tensor
    log_det: Tensor
    inv_covariance: Tensor
    log_representation: Tensor


@dataclass
class RegressionWithMaskAdaptorOutput:
    value: Tensor
    logits: Tensor
    mask: Tensor


@dataclass
class RegressionWithConfidenceAndMaskAdaptorOutput:
    value: Tensor
    confidence: Tensor
    logits: Tensor
    mask: Tensor


class UniCeptionPredictionHeadBase(nn.Module):
    def __init__(self, name: str, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.name: str = name

    def forward(self, head_input: PredictionHeadInput) -> PredictionHeadOutput:
        raise NotImplementedError


class UniCeptionAdaptorBase(nn.Module):
    def __init__(self, name: str, required_channels: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.name: str = name
        self.required_channels: int = required_channels

    def forward(self, adaptor_input: AdaptorInput) -> AdaptorOutput:
        raise NotImplementedError
