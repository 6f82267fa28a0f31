"""Info-sharing base classes and dataclasses (reference: info_sharing/base.py:14-97)."""
from dataclasses import dataclass
from typing import List, Optional

import torch.nn as nn
from torch import Tensor


@dataclass
class InfoSharingInput:
    pass


@dataclass
class InfoSharingOutput:
    pass


class UniCeptionInfoSharingBase(nn.Module):
    "Information Sharing Base Class"

    def __init__(self, name: str, size: Optional[str] = None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.name: str = name
        self.size: Optional[str] = size

    def forward(self, model_input: InfoSharingInput) -> InfoSharingOutput:
        raise NotImplementedError

    def wrap_module_with_gradient_checkpointing(self, module: nn.Module):
        "Re-compute `module`'s forward in the backward pass (info_sharing/base.py:59-72); see models/utils/checkpointing.py."
        from ..utils.checkpointing import wrap_module_with_gradient_checkpointing
        return wrap_module_with_gradient_checkpointing(module)


@dataclass
class MultiViewTransformerInput(InfoSharingInput):
    features: List[Tensor]  # per view [batch, input_embed_dim, feat_height, feat_width]
    additional_input_tokens: Optional[Tensor] = None
    additional_input_tokens_per_view: Optional[List[Tensor]] = None


@dataclass
class MultiViewTransformerOutput(InfoSharingOutput):
    features: List[Tensor]  # per view [batch, transformer_embed_dim, feat_height, feat_width]
    additional_token_features: Optional[Tensor] = None
    additional_token_features_per_view: Optional[List[Tensor]] = None
