"""Multi-view global-attention transformer (reference: info_sharing/global_attention_transformer.py:25-898) and the shared
token-stream core of the global / alternating transformers, on the HIP kernels.

All views' tokens form ONE sequence per batch element ([B, V*T (+G), dim], T = h*w + per-view extra tokens, G = global
extra tokens); every depth is a SelfAttentionBlock (utils/transformer_blocks.py) over that sequence.  The alternating variant
(alternating_attention_transformer.py) runs the odd depths per view instead — for the row-major [B*V*T, dim] token matrix
that is the same memory read as B*V sequences of T tokens, so switching between global and frame attention costs nothing.
BCHW inputs that are channels-last views enter without a transpose; BCHW outputs are channels-last views of the token matrix.
"""
from functools import partial
from typing import Callable, List, Optional, Type, Union

import numpy as np
import torch
import torch.nn as nn

from ... import autograd, engine, ops
from ..libs.croco.pos_embed import RoPE2D
from ..utils.intermediate_feature_return import IntermediateFeatureReturner, feature_take_indices
from ..utils.positional_encoding import PositionGetter
from ..utils.transformer_blocks import Mlp, SelfAttentionBlock
from .base import MultiViewTransformerInput, MultiViewTransformerOutput, UniCeptionInfoSharingBase


def sinusoid_encoding_table(n_position: int, d_hid: int, base: float) -> torch.Tensor:
    "Sinusoid position table [n_position, d_hid] (global_attention_transformer.py:202-212), fp32 from a float64 evaluation."
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(float(base), 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table)


class _MultiViewSelfAttentionCore(UniCeptionInfoSharingBase):
    "Construction and token-stream forward shared by the global and the alternating transformer."

    def _build(self, input_embed_dim, distinguish_ref_and_non_ref_views, use_pe_for_non_reference_views, max_num_views_for_pe,
               use_rand_idx_pe_for_non_reference_views, depth, dim, num_heads, mlp_ratio, qkv_bias, qk_norm, proj_drop, attn_drop,
               init_values, drop_path, act_layer, norm_layer, mlp_layer, custom_positional_encoding, use_scalable_softmax,
               use_entropy_scaling, base_token_count_for_entropy_scaling, entropy_scaling_growth_factor,
               pretrained_checkpoint_path, gradient_checkpointing, what):
        self.input_embed_dim = input_embed_dim
        self.distinguish_ref_and_non_ref_views = distinguish_ref_and_non_ref_views
        self.use_pe_for_non_reference_views = use_pe_for_non_reference_views
        self.max_num_views_for_pe = max_num_views_for_pe
        self.use_rand_idx_pe_for_non_reference_views = use_rand_idx_pe_for_non_reference_views
        self.depth, self.dim, self.num_heads, self.mlp_ratio = depth, dim, num_heads, mlp_ratio
        self.qkv_bias, self.qk_norm, self.proj_drop, self.attn_drop = qkv_bias, qk_norm, proj_drop, attn_drop
        self.init_values, self.drop_path = init_values, drop_path
        self.act_layer, self.norm_layer, self.mlp_layer = act_layer, norm_layer, mlp_layer
        self.custom_positional_encoding = custom_positional_encoding
        self.use_scalable_softmax, self.use_entropy_scaling = use_scalable_softmax, use_entropy_scaling
        self.base_token_count_for_entropy_scaling = base_token_count_for_entropy_scaling
        self.entropy_scaling_growth_factor = entropy_scaling_growth_factor
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        self.gradient_checkpointing = gradient_checkpointing
        self.proj_embed = nn.Linear(input_embed_dim, dim, bias=True) if input_embed_dim != dim else nn.Identity()
        if isinstance(self.custom_positional_encoding, str):
            if self.custom_positional_encoding != "rope":
                raise ValueError(f"Unknown custom positional encoding: {self.custom_positional_encoding}")
            self.rope = RoPE2D(freq=100.0, F0=1.0)
            self.custom_positional_encoding = self.rope
        self.self_attention_blocks = nn.ModuleList([
            SelfAttentionBlock(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_norm=qk_norm,
                               proj_drop=proj_drop, attn_drop=attn_drop, init_values=init_values, drop_path=drop_path,
                               act_layer=act_layer, norm_layer=norm_layer, mlp_layer=mlp_layer,
                               custom_positional_encoding=self.custom_positional_encoding,
                               use_scalable_softmax=use_scalable_softmax, use_entropy_scaling=use_entropy_scaling,
                               base_token_count_for_entropy_scaling=base_token_count_for_entropy_scaling,
                               entropy_scaling_growth_factor=entropy_scaling_growth_factor)
            for _ in range(depth)])
        self.norm = norm_layer(dim)
        if self.custom_positional_encoding is not None:
            self.position_getter = PositionGetter()
        if distinguish_ref_and_non_ref_views:
            n = max_num_views_for_pe if use_pe_for_non_reference_views else 1
            self.register_buffer("view_pos_table", sinusoid_encoding_table(n, dim, 10000))
        self.initialize_weights()
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained multi-view {what} transformer weights from {pretrained_checkpoint_path} ...")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))
        if self.gradient_checkpointing:       # (global_attention_transformer.py:196-198, alternating_attention_transformer.py:178-180)
            for i, block in enumerate(self.self_attention_blocks):
                self.self_attention_blocks[i] = self.wrap_module_with_gradient_checkpointing(block)

    _get_sinusoid_encoding_table = staticmethod(sinusoid_encoding_table)

    def initialize_weights(self):
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _frame_level(self, depth_idx: int) -> bool:
        "True when block `depth_idx` attends inside each view only (alternating transformer); the global one never does."
        return False

    # ---- token-stream core ------------------------------------------------------------------------------------
    def _run(self, model_input: MultiViewTransformerInput, take_indices, norm_intermediate):
        feats = model_input.features
        if self.distinguish_ref_and_non_ref_views and self.use_pe_for_non_reference_views:
            assert len(feats) <= self.max_num_views_for_pe, f"Expected less than {self.max_num_views_for_pe} views, got {len(feats)}"
        assert all(f.shape[1] == self.input_embed_dim for f in feats), f"All views must have input dimension {self.input_embed_dim}"
        assert all(f.ndim == 4 for f in feats), "All views must have 4 dimensions (N, C, H, W)"
        V = len(feats)
        B, _, h, w = feats[0].shape
        hw = h * w
        dt = engine.compute_dtype()
        per_view, glob = model_input.additional_input_tokens_per_view, model_input.additional_input_tokens
        Tp = 0
        if per_view is not None:
            assert len(per_view) == V, f"Number of additional token tensors ({len(per_view)}) must match number of views ({V})"
            assert all(t.ndim == 3 for t in per_view), "Additional tokens per view must have 3 dimensions (N, C, T)"
            assert all(t.shape[1] == self.input_embed_dim for t in per_view), f"Additional tokens per view must have input dimension {self.input_embed_dim}"
            assert all(t.shape[0] == B for t in per_view), "Batch size mismatch for additional tokens per view"
            Tp = per_view[0].shape[2]
        G = 0
        if glob is not None:
            assert glob.ndim == 3, "Additional tokens must have 3 dimensions (N, C, T)"
            assert glob.shape[1] == self.input_embed_dim, f"Additional tokens must have input dimension {self.input_embed_dim}"
            assert glob.shape[0] == B, "Batch size mismatch for additional tokens"
            G = glob.shape[2]
        if self.custom_positional_encoding is not None and (per_view is not None or glob is not None):
            raise ValueError("Custom positional encoding is not supported when additional_input_tokens or "
                             "additional_input_tokens_per_view are provided. Please set custom_positional_encoding=None "
                             "or remove additional tokens from the input.")
        # training (gradients requested through the features, the extra tokens or the parameters): the blocks switch to their
        # HIP forward + backward sub-layers themselves; here only the input projection needs the differentiable form
        train = autograd.grad_needed(*feats, *(per_view or ()), glob, self.norm.weight, *self.proj_embed.parameters())
        T = hw + Tp
        L = V * T + G
        in_dt = torch.float32 if isinstance(self.proj_embed, nn.Identity) else dt
        # token matrix [B, L, Cin]: rows (view, token) per batch element, global extra tokens last (layout hops only)
        parts = []
        for v, f in enumerate(feats):
            parts.append(engine.bchw_to_nhwc(f, in_dt).reshape(B, hw, self.input_embed_dim))
            if per_view is not None:
                parts.append(per_view[v].permute(0, 2, 1).to(in_dt))
        if glob is not None:
            parts.append(glob.permute(0, 2, 1).to(in_dt))
        tok = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        x2d = tok.reshape(B * L, self.input_embed_dim)
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        if not isinstance(self.proj_embed, nn.Identity):
            if train:
                x2d = autograd.linear(x2d, self.proj_embed.weight, self.proj_embed.bias, self.proj_embed, dt, torch.float32)
            else:
                wpe, bpe = engine.lin_weights(self.proj_embed, dt)
                x2d = ops.gemm(x2d, wpe, bpe, out_dtype=torch.float32)
        elif x2d.data_ptr() == feats[0].data_ptr():
            x2d = x2d.clone()      # the view encoding below is added in place: never into the caller's features
        pos = None
        if self.custom_positional_encoding is not None:
            p1 = self.position_getter(B, h, w, x2d.device)
            pos = torch.cat([p1] * V, dim=1) if V > 1 else p1            # [B, V*hw, 2]
        if self.distinguish_ref_and_non_ref_views:
            idx = [0]
            if self.use_pe_for_non_reference_views and V > 1:
                if self.use_rand_idx_pe_for_non_reference_views:   # same generator call as the reference (:378-380)
                    idx += torch.randint(low=1, high=self.max_num_views_for_pe, size=(V - 1,)).tolist()
                else:
                    idx += list(range(1, V))
            pe = self.view_pos_table[idx].float().contiguous()      # [1 or V, dim]: reference view only, or every view
            # (in place through the raw kernel, also under autograd: d(x + const)/dx = 1 and no Function saved x2d's values)
            ops.add_view_pe_(x2d.detach().view(B, L, self.dim), pe, T)
        if not train and engine.stream_dtype(dt, self.dim) == torch.bfloat16:
            x2d = ops.convert(x2d, torch.bfloat16)      # bf16 residual stream (engine.stream_dtype): one cast, then 4 bytes per element per sub-layer
        taken = []
        for d, blk in enumerate(self.self_attention_blocks):
            if not self._frame_level(d):
                x2d = blk.forward_tokens(x2d, B, L, pos, dt)
            elif G == 0:
                # frame attention: the same rows read as B*V sequences of T tokens
                x2d = blk.forward_tokens(x2d, B * V, T, None if pos is None else pos.reshape(B * V, T, 2), dt)
            else:
                # global extra tokens sit out the frame-level blocks (alternating_attention_transformer.py:404-446)
                x3 = x2d.view(B, L, self.dim)
                xv = blk.forward_tokens(x3[:, :V * T].reshape(B * V * T, self.dim), B * V, T, None, dt)
                x2d = torch.cat([xv.view(B, V * T, self.dim), x3[:, V * T:]], dim=1).reshape(B * L, self.dim)
            if d in take_indices:
                taken.append(engine.layernorm(x2d, self.norm, torch.float32, twin=True) if norm_intermediate else x2d)

        def out(t2d):
            t4 = t2d.view(B, L, self.dim)[:, :V * T].reshape(B, V, T, self.dim)
            views = [t4[:, v, :hw].reshape(B, h, w, self.dim).permute(0, 3, 1, 2) for v in range(V)]
            pv = None if per_view is None else [t4[:, v, hw:].permute(0, 2, 1).contiguous() for v in range(V)]
            gl = None if glob is None else t2d.view(B, L, self.dim)[:, V * T:].permute(0, 2, 1).contiguous()
            return MultiViewTransformerOutput(features=views, additional_token_features=gl, additional_token_features_per_view=pv)

        return x2d, [out(t) for t in taken], out

    def forward(self, model_input: MultiViewTransformerInput) -> MultiViewTransformerOutput:
        x2d, _, out = self._run(model_input, (), False)
        return out(engine.layernorm(x2d, self.norm, torch.float32, twin=True))

    def _forward_ifr(self, model_input: MultiViewTransformerInput):
        take_indices, _ = feature_take_indices(self.depth, self.indices)
        x2d, inter, out = self._run(model_input, take_indices, self.norm_intermediate)
        if self.intermediates_only:
            return inter
        return out(engine.layernorm(x2d, self.norm, torch.float32, twin=True)), inter


_CTOR_DOC = """Same constructor as the reference class (all arguments recorded as attributes); dropout / DropPath > 0 in
training, qk_norm, latent attention and gradient checkpointing raise instead of silently diverging."""


class MultiViewGlobalAttentionTransformer(_MultiViewSelfAttentionCore):
    "UniCeption Multi-View Global-Attention Transformer: every block attends over the tokens of all views at once."
    __doc__ += "\n" + _CTOR_DOC

    def __init__(self, name: str, input_embed_dim: int, distinguish_ref_and_non_ref_views: bool = True,
                 use_pe_for_non_reference_views: bool = True, max_num_views_for_pe: int = 1000,
                 use_rand_idx_pe_for_non_reference_views: bool = True, size: Optional[str] = None, depth: int = 12,
                 dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: Type[nn.Module] = nn.GELU,
                 norm_layer: Union[Type[nn.Module], Callable[..., nn.Module]] = partial(nn.LayerNorm, eps=1e-6),
                 mlp_layer: Type[nn.Module] = Mlp, custom_positional_encoding: Optional[Union[str, Callable]] = None,
                 use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4,
                 pretrained_checkpoint_path: Optional[str] = None, gradient_checkpointing: bool = False, *args, **kwargs):
        super().__init__(name=name, size=size, *args, **kwargs)
        self._build(input_embed_dim, distinguish_ref_and_non_ref_views, use_pe_for_non_reference_views, max_num_views_for_pe,
                    use_rand_idx_pe_for_non_reference_views, depth, dim, num_heads, mlp_ratio, qkv_bias, qk_norm, proj_drop,
                    attn_drop, init_values, drop_path, act_layer, norm_layer, mlp_layer, custom_positional_encoding,
                    use_scalable_softmax, use_entropy_scaling, base_token_count_for_entropy_scaling,
                    entropy_scaling_growth_factor, pretrained_checkpoint_path, gradient_checkpointing, "global-attention")


class MultiViewGlobalAttentionTransformerIFR(MultiViewGlobalAttentionTransformer, IntermediateFeatureReturner):
    "Same transformer, also returning the features after the blocks in `indices` (global_attention_transformer.py:462-898)."

    def __init__(self, name: str, input_embed_dim: int, distinguish_ref_and_non_ref_views: bool = True,
                 use_pe_for_non_reference_views: bool = True, max_num_views_for_pe: int = 1000,
                 use_rand_idx_pe_for_non_reference_views: bool = True, size: Optional[str] = None, depth: int = 12,
                 dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6),
                 mlp_layer: nn.Module = Mlp, custom_positional_encoding: Callable = None, use_scalable_softmax: bool = False,
                 use_entropy_scaling: bool = False, base_token_count_for_entropy_scaling: int = 444,
                 entropy_scaling_growth_factor: float = 1.4, pretrained_checkpoint_path: str = None,
                 indices: Optional[Union[int, List[int]]] = None, norm_intermediate: bool = True,
                 intermediates_only: bool = False, gradient_checkpointing: bool = False, *args, **kwargs):
        MultiViewGlobalAttentionTransformer.__init__(
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
