"""Info-sharing registry, restricted to the cross-attention transformer of the DUSt3R path
(reference: info_sharing/__init__.py:23-37)."""
from .base import MultiViewTransformerInput, MultiViewTransformerOutput, UniCeptionInfoSharingBase  # noqa: F401
from .cross_attention_transformer import MultiViewCrossAttentionTransformer, MultiViewCrossAttentionTransformerIFR

INFO_SHARING_CLASSES = {
    "cross_attention": (MultiViewCrossAttentionTransformer, MultiViewCrossAttentionTransformerIFR),
}

__all__ = ["INFO_SHARING_CLASSES", "MultiViewTransformerInput"]
