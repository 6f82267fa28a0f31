"""Info-sharing registry (reference: info_sharing/__init__.py:23-37): the cross-attention transformer of the DUSt3R path and the
global / alternating self-attention transformers built from the same kernels.  (The differential cross-attention variant is
outside the path.)"""
from .alternating_attention_transformer import (MultiViewAlternatingAttentionTransformer,
                                                MultiViewAlternatingAttentionTransformerIFR)
from .base import MultiViewTransformerInput, MultiViewTransformerOutput, UniCeptionInfoSharingBase  # noqa: F401
from .cross_attention_transformer import MultiViewCrossAttentionTransformer, MultiViewCrossAttentionTransformerIFR
from .global_attention_transformer import MultiViewGlobalAttentionTransformer, MultiViewGlobalAttentionTransformerIFR

INFO_SHARING_CLASSES = {
    "cross_attention": (MultiViewCrossAttentionTransformer, MultiViewCrossAttentionTransformerIFR),
    "alternating_attention": (MultiViewAlternatingAttentionTransformer, MultiViewAlternatingAttentionTransformerIFR),
    "global_attention": (MultiViewGlobalAttentionTransformer, MultiViewGlobalAttentionTransformerIFR),
}

__all__ = ["INFO_SHARING_CLASSES", "MultiViewTransformerInput"]
