"""Multi-view cross-attention transformer (reference: info_sharing/cross_attention_transformer.py:22-505).

Per depth every view's block reads the *previous* depth's tokens of all other views (Jacobi update, :470).
Token streams stay [B*N, dim] fp32 between blocks; BCHW inputs that are channels-last views (what CroCoEncoder
returns) enter without a copy, BCHW outputs are channels-last views of the token matrices.
"""
from copy import deepcopy
from functools import partial
from typing import Callable, List, Optional, Type, Union

import torch
import torch.nn as nn

from ... import autograd, engine, ops
from ..utils.intermediate_feature_return import IntermediateFeatureReturner, feature_take_indices
from ..utils.positional_encoding import PositionGetter
from ..utils.transformer_blocks import CrossAttentionBlock, Mlp
from .base import MultiViewTransformerInput, MultiViewTransformerOutput, UniCeptionInfoSharingBase


class MultiViewCrossAttentionTransformer(UniCeptionInfoSharingBase):
    "Cross-attention transformer with one branch of blocks per view."

    def __init__(self, name: str, input_embed_dim: int, num_views: int, size: Optional[str] = None, depth: int = 12,
                 dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: Type[nn.Module] = nn.GELU,
                 norm_layer: Union[Type[nn.Module], Callable[..., nn.Module]] = partial(nn.LayerNorm, eps=1e-6),
                 mlp_layer: Type[nn.Module] = Mlp, custom_positional_encoding: Optional[Callable] = None,
                 norm_cross_tokens: bool = True, use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4,
                 pretrained_checkpoint_path: Optional[str] = None, gradient_checkpointing: bool = False, *args, **kwargs):
        super().__init__(name=name, size=size, *args, **kwargs)
        self.input_embed_dim = input_embed_dim
        self.num_views = num_views
        self.depth = depth
        self.dim = dim
        self.num_heads = num_heads
        self.mlp_ratio = mlp_ratio
        self.qkv_bias = qkv_bias
        self.qk_norm = qk_norm
        self.proj_drop = proj_drop
        self.attn_drop = attn_drop
        self.init_values = init_values
        self.drop_path = drop_path
        self.act_layer = act_layer
        self.norm_layer = norm_layer
        self.mlp_layer = mlp_layer
        self.custom_positional_encoding = custom_positional_encoding
        self.norm_cross_tokens = norm_cross_tokens
        self.use_scalable_softmax = use_scalable_softmax
        self.use_entropy_scaling = use_entropy_scaling
        self.base_token_count_for_entropy_scaling = base_token_count_for_entropy_scaling
        self.entropy_scaling_growth_factor = entropy_scaling_growth_factor
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        self.gradient_checkpointing = gradient_checkpointing

        self.proj_embed = nn.Linear(input_embed_dim, dim, bias=True) if input_embed_dim != dim else nn.Identity()
        branch = nn.ModuleList([
            CrossAttentionBlock(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_norm=qk_norm,
                                proj_drop=proj_drop, attn_drop=attn_drop, init_values=init_values, drop_path=drop_path,
                                act_layer=act_layer, norm_layer=norm_layer, mlp_layer=mlp_layer,
                                custom_positional_encoding=custom_positional_encoding, norm_cross_tokens=norm_cross_tokens,
                                use_scalable_softmax=use_scalable_softmax, use_entropy_scaling=use_entropy_scaling,
                                base_token_count_for_entropy_scaling=base_token_count_for_entropy_scaling,
                                entropy_scaling_growth_factor=entropy_scaling_growth_factor)
            for _ in range(depth)])
        # per-view weights: every further view starts as a copy of view 0's branch (:146-150)
        self.multi_view_branches = nn.ModuleList([branch])
        for _ in range(1, num_views):
            self.multi_view_branches.append(deepcopy(branch))
        self.norm = norm_layer(dim)
        if custom_positional_encoding is not None:
            self.position_getter = PositionGetter()
        self.initialize_weights()
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained multi-view cross-attention transformer weights from {pretrained_checkpoint_path} ...")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))
        if self.gradient_checkpointing:
            # what the reference means to do at :163-165 (it raises AttributeError there: self.cross_attention_blocks is never assigned)
            for branch in self.multi_view_branches:
                for i, block in enumerate(branch):
                    branch[i] = self.wrap_module_with_gradient_checkpointing(block)

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

    # ---- shared token-stream core ---------------------------------------------------------------
    def _check_input(self, model_input):
        feats = model_input.features
        assert len(feats) == self.num_views, f"Expected {self.num_views} views, got {len(feats)}"
        assert all(f.shape[1] == self.input_embed_dim for f in feats), f"All views must have input dimension {self.input_embed_dim}"
        assert all(f.ndim == 4 for f in feats), "All views must have 4 dimensions (N, C, H, W)"

    def _run(self, model_input, take_indices, norm_intermediate):
        feats = model_input.features
        B, _, h, w = feats[0].shape
        N = h * w
        dt = engine.compute_dtype()
        V = self.num_views
        # NCHW -> NLC (free for channels-last views), then proj_embed
        xs = []
        train = autograd.grad_needed(*feats, self.norm.weight)
        for f in feats:
            if train:
                # differentiable layout hop (a view for the channels-last features the encoder returns)
                x2d = f.float().permute(0, 2, 3, 1).reshape(B * N, self.input_embed_dim)
                if not isinstance(self.proj_embed, nn.Identity):
                    x2d = autograd.linear(x2d, self.proj_embed.weight, self.proj_embed.bias, self.proj_embed, dt,
                                          engine.stream_dtype(dt, self.dim, self.input_embed_dim))
                xs.append(x2d)
                continue
            nlc = engine.bchw_to_nhwc(f, torch.float32 if isinstance(self.proj_embed, nn.Identity) else dt)
            x2d = nlc.reshape(B * N, self.input_embed_dim)
            if not isinstance(self.proj_embed, nn.Identity):
                wpe, bpe = engine.lin_weights(self.proj_embed, dt)
                x2d = ops.gemm(x2d, wpe, bpe, out_dtype=engine.stream_dtype(dt, self.dim, self.input_embed_dim),
                               emit_ln=engine.fold_ok(dt, self.dim, self.input_embed_dim))
            xs.append(x2d)
        if self.custom_positional_encoding is not None:
            pos = [self.position_getter(B, h, w, f.device) for f in feats]
        else:
            pos = [None] * V
        if engine.is_native_rope(self.custom_positional_encoding) and dt == torch.bfloat16:
            # shared caches are filled on the main stream before any branch work is forked to a side stream
            ops.rope_table(feats[0].device, engine.ROPE_TABLE_NPOS, self.custom_positional_encoding.base, self.custom_positional_encoding.F0)
        taken = []
        for d in range(self.depth):
            if V == 2:   # the two branches of a depth level are independent (Jacobi update): two streams when they are small
                b0, b1 = self.multi_view_branches[0][d], self.multi_view_branches[1][d]
                x0, x1 = xs
                engine.finalize_ln(x0, b0.norm1)   # both branches read both streams' row statistics
                engine.finalize_ln(x1, b1.norm1)
                xs = list(engine.run_branches(
                    lambda: b0.forward_tokens(x0, x1, B, N, N, pos[0], pos[1], dt),
                    lambda: b1.forward_tokens(x1, x0, B, N, N, pos[1], pos[0], dt), B * N, inputs1=(x0, x1),
                    warm_key=("dec", B, N, str(dt), torch.is_grad_enabled()), owner=self, disjoint_params=True))
                if d in take_indices:
                    taken.append([engine.layernorm(x, self.norm, torch.float32, twin=True) if norm_intermediate else x for x in xs])
                continue
            new = []
            for v in range(V):
                others = [u for u in range(V) if u != v]
                if len(others) == 1:
                    y2d, ypos, Ny = xs[others[0]], pos[others[0]], N
                else:  # K/V = all other views' tokens, concatenated per batch element (:246-256)
                    y2d = torch.cat([xs[u].view(B, N, -1) for u in others], dim=1).reshape(B * N * len(others), -1)
                    ypos = torch.cat([pos[u] for u in others], dim=1) if pos[v] is not None else None
                    Ny = N * len(others)
                new.append(self.multi_view_branches[v][d].forward_tokens(xs[v], y2d, B, N, Ny, pos[v], ypos, dt))
            xs = new
            if d in take_indices:
                taken.append([engine.layernorm(x, self.norm, torch.float32, twin=True) if norm_intermediate else x for x in xs])

        def out(ts):
            return MultiViewTransformerOutput(features=[engine.nlc_as_bchw(t, B, h, w) for t in ts])

        return xs, [out(t) for t in taken], out, B, h, w

    def forward(self, model_input: MultiViewTransformerInput) -> MultiViewTransformerOutput:
        self._check_input(model_input)
        xs, _, out, *_ = self._run(model_input, (), False)
        return out([engine.layernorm(x, self.norm, torch.float32, twin=True) for x in xs])


class MultiViewCrossAttentionTransformerIFR(MultiViewCrossAttentionTransformer, IntermediateFeatureReturner):
    "Same transformer, also returning the features after the blocks in `indices` (:278-505)."

    def __init__(self, name: str, input_embed_dim: int, num_views: int, size: Optional[str] = None, depth: int = 12,
                 dim: int = 768, num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6),
                 mlp_layer: nn.Module = Mlp, custom_positional_encoding: Callable = None, norm_cross_tokens: bool = True,
                 use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4,
                 pretrained_checkpoint_path: str = None, indices: Optional[Union[int, List[int]]] = None,
                 norm_intermediate: bool = True, intermediates_only: bool = False, gradient_checkpointing: bool = False,
                 *args, **kwargs):
        MultiViewCrossAttentionTransformer.__init__(
            self, name=name, input_embed_dim=input_embed_dim, num_views=num_views, size=size, depth=depth, dim=dim,
            num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_norm=qk_norm, proj_drop=proj_drop,
            attn_drop=attn_drop, init_values=init_values, drop_path=drop_path, act_layer=act_layer, norm_layer=norm_layer,
            mlp_layer=mlp_layer, custom_positional_encoding=custom_positional_encoding, norm_cross_tokens=norm_cross_tokens,
            use_scalable_softmax=use_scalable_softmax, use_entropy_scaling=use_entropy_scaling,
            base_token_count_for_entropy_scaling=base_token_count_for_entropy_scaling,
            entropy_scaling_growth_factor=entropy_scaling_growth_factor, pretrained_checkpoint_path=pretrained_checkpoint_path,
            gradient_checkpointing=gradient_checkpointing, *args, **kwargs)
        IntermediateFeatureReturner.__init__(self, indices=indices, norm_intermediate=norm_intermediate,
                                             intermediates_only=intermediates_only)

    def forward(self, model_input: MultiViewTransformerInput):
        self._check_input(model_input)
        take_indices, _ = feature_take_indices(self.depth, self.indices)
        xs, inter, out, *_ = self._run(model_input, take_indices, self.norm_intermediate)
        if self.intermediates_only:
            return inter
        return out([engine.layernorm(x, self.norm, torch.float32, twin=True) for x in xs]), inter
