"""Multi-view alternating-attention transformer (reference: info_sharing/alternating_attention_transformer.py:22-990): even
depths attend over all views' tokens (global), odd depths inside each view (frame level).  Shares the token-stream core of
global_attention_transformer.py: on the row-major [B*V*T, dim] token matrix a frame-level block is the same memory read as B*V
sequences of T tokens, so alternating costs no data movement."""
from functools import partial
from typing import Callable, List, Optional, Type, Union

import torch.nn as nn

from ..utils.intermediate_feature_return import IntermediateFeatureReturner
from ..utils.transformer_blocks import Mlp
from .base import MultiViewTransformerInput
from .global_attention_transformer import _CTOR_DOC, _MultiViewSelfAttentionCore


class MultiViewAlternatingAttentionTransformer(_MultiViewSelfAttentionCore):
    "UniCeption Multi-View Alternating-Attention Transformer: global attention at even depths, per-view attention at odd depths."
    __doc__ += "\n" + _CTOR_DOC

    def __init__(self, name: str, input_embed_dim: int, distinguish_ref_and_non_ref_views: bool = True,
                 use_pe_for_non_reference_views: bool = False, max_num_views_for_pe: int = 1000,
                 use_rand_idx_pe_for_non_reference_views: bool = True, size: Optional[str] = None, depth: int = 12,
                 dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: Type[nn.Module] = nn.GELU,
                 norm_layer: Union[Type[nn.Module], Callable[..., nn.Module]] = partial(nn.LayerNorm, eps=1e-6),
                 mlp_layer: Type[nn.Module] = Mlp, custom_positional_encoding: Optional[Callable] = None,
                 use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4,
                 pretrained_checkpoint_path: Optional[str] = None, gradient_checkpointing: bool = False, *args, **kwargs):
        super().__init__(name=name, size=size, *args, **kwargs)
        self._build(input_embed_dim, distinguish_ref_and_non_ref_views, use_pe_for_non_reference_views, max_num_views_for_pe,
                    use_rand_idx_pe_for_non_reference_views, depth, dim, num_heads, mlp_ratio, qkv_bias, qk_norm, proj_drop,
                    attn_drop, init_values, drop_path, act_layer, norm_layer, mlp_layer, custom_positional_encoding,
                    use_scalable_softmax, use_entropy_scaling, base_token_count_for_entropy_scaling,
                    entropy_scaling_growth_factor, pretrained_checkpoint_path, gradient_checkpointing, "alternating-attention")

    def _frame_level(self, depth_idx: int) -> bool:
        return depth_idx % 2 == 1      # alternating_attention_transformer.py:398-403


class MultiViewAlternatingAttentionTransformerIFR(MultiViewAlternatingAttentionTransformer, IntermediateFeatureReturner):
    "Same transformer, also returning the features after the blocks in `indices` (alternating_attention_transformer.py:502-990)."

    def __init__(self, name: str, input_embed_dim: int, distinguish_ref_and_non_ref_views: bool = True,
                 use_pe_for_non_reference_views: bool = False, max_num_views_for_pe: int = 1000,
                 use_rand_idx_pe_for_non_reference_views: bool = True, size: Optional[str] = None, depth: int = 12,
                 dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6),
                 mlp_layer: nn.Module = Mlp, custom_positional_encoding: Callable = None, use_scalable_softmax: bool = False,
                 use_entropy_scaling: bool = False, base_token_count_for_entropy_scaling: int = 444,
                 entropy_scaling_growth_factor: float = 1.4, pretrained_checkpoint_path: str = None,
                 indices: Optional[Union[int, List[int]]] = None, norm_intermediate: bool = True,
                 intermediates_only: bool = False, gradient_checkpointing: bool = False, *args, **kwargs):
        MultiViewAlternatingAttentionTransformer.__init__(
            self, name=name, input_embed_dim=input_embed_dim, distinguish_ref_and_non_ref_views=distinguish_ref_and_non_ref_views,
            use_pe_for_non_reference_views=use_pe_for_non_reference_views, max_num_views_for_pe=max_num_views_for_pe,
            use_rand_idx_pe_for_non_reference_views=use_rand_idx_pe_for_non_reference_views, size=size, depth=depth, dim=dim,
            num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_norm=qk_norm, proj_drop=proj_drop,
            attn_drop=attn_drop, init_values=init_values, drop_path=drop_path, act_layer=act_layer, norm_layer=norm_layer,
            mlp_layer=mlp_layer, custom_positional_encoding=custom_positional_encoding, use_scalable_softmax=use_scalable_softmax,
            use_entropy_scaling=use_entropy_scaling, base_token_count_for_entropy_scaling=base_token_count_for_entropy_scaling,
            entropy_scaling_growth_factor=entropy_scaling_growth_factor, pretrained_checkpoint_path=pretrained_checkpoint_path,
            gradient_checkpointing=gradient_checkpointing, *args, **kwargs)
        IntermediateFeatureReturner.__init__(self, indices=indices, norm_intermediate=norm_intermediate,
                                             intermediates_only=intermediates_only)

    def forward(self, model_input: MultiViewTransformerInput):
        return self._forward_ifr(model_input)
