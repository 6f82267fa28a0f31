"""CroCo / DUSt3R ViT encoder (reference: encoders/croco.py:18-327) on the HIP token-stream pipeline.

forward: patch gather + GEMM -> `enc_depth` fused blocks -> final LayerNorm.  The returned BCHW features are a
channels-last *view* of the [B*N, D] token matrix (the reference materialises a contiguous NCHW copy,
croco.py:177-180); values are identical and the decoder consumes the view without a transpose.
"""
from functools import partial
from typing import Callable, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import engine
from ..libs.croco.blocks import Block, _as_2d
from ..libs.croco.patch_embed import get_patch_embed
from ..libs.croco.pos_embed import RoPE2D
from ..utils.intermediate_feature_return import IntermediateFeatureReturner, feature_take_indices
from .base import UniCeptionViTEncoderBase, ViTEncoderInput, ViTEncoderOutput


class CroCoEncoder(UniCeptionViTEncoderBase):
    "UniCeption CroCov2 Encoder"

    def __init__(self, name: str, data_norm_type: str, patch_embed_cls: str = "PatchEmbedDust3R",
                 img_size: Union[int, Tuple[int, int]] = (224, 224), patch_size: int = 16, enc_embed_dim: int = 1024,
                 enc_depth: int = 24, enc_num_heads: int = 16, mlp_ratio: int = 4,
                 norm_layer: Callable = partial(nn.LayerNorm, eps=1e-6), pos_embed: str = "RoPE100",
                 pretrained_checkpoint_path: str = None, override_checkpoint_attributes: bool = False, *args, **kwargs):
        super().__init__(name=name, data_norm_type=data_norm_type, patch_size=patch_size, *args, **kwargs)
        self.patch_embed_cls = patch_embed_cls
        self.img_size = img_size
        self.enc_embed_dim = enc_embed_dim
        self.enc_depth = enc_depth
        self.enc_num_heads = enc_num_heads
        self.mlp_ratio = mlp_ratio
        self.norm_layer = norm_layer
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        self.override_checkpoint_attributes = override_checkpoint_attributes

        self.pos_embed = pos_embed
        if pos_embed.startswith("RoPE"):  # e.g. RoPE100
            self.enc_pos_embed = None
            self.dec_pos_embed = None
            self.rope = RoPE2D(freq=float(pos_embed[len("RoPE"):]))
        else:
            raise NotImplementedError("Unknown pos_embed " + pos_embed)

        self._set_patch_embed(img_size, patch_size, enc_embed_dim)
        self._set_encoder(enc_depth, enc_embed_dim, enc_num_heads, mlp_ratio, norm_layer, self.rope)
        self.initialize_weights()

        if pretrained_checkpoint_path:
            print(f"Loading pretrained CroCo checkpoint from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))
            if not override_checkpoint_attributes:
                assert data_norm_type == ckpt["data_norm_type"], (
                    f"Data normalization type {data_norm_type} does not match the checkpoint {ckpt['data_norm_type']}.")
                assert patch_embed_cls == ckpt["patch_embed_cls"], (
                    f"Patch embedding class {patch_embed_cls} does not match the checkpoint {ckpt['patch_embed_cls']}.")

    def _set_patch_embed(self, img_size=224, patch_size=16, enc_embed_dim=768):
        self.patch_embed = get_patch_embed(self.patch_embed_cls, img_size, patch_size, enc_embed_dim)

    def _set_encoder(self, enc_depth, enc_embed_dim, enc_num_heads, mlp_ratio, norm_layer, rope):
        self.enc_blocks = nn.ModuleList(
            [Block(enc_embed_dim, enc_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer, rope=rope)
             for _ in range(enc_depth)])
        self.enc_norm = norm_layer(enc_embed_dim)

    def initialize_weights(self):
        self.patch_embed._init_weights()
        self.apply(self._init_weights)

    def _init_weights(self, m):
        # xavier-uniform linears, unit LayerNorms (croco.py:135-145)
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---- token-stream core shared by both forward variants -------------------------------------
    def _embed(self, encoder_input: ViTEncoderInput):
        self._check_data_normalization_type(encoder_input.data_norm_type)
        image = encoder_input.image
        B, _, H, W = image.shape
        if hasattr(encoder_input, "true_shape"):
            true_shape = encoder_input.true_shape
        else:
            true_shape = torch.tensor([H, W])[None].repeat(B, 1)
        tokens, pos = self.patch_embed(image, true_shape=true_shape)
        return tokens, pos, B, H // self.patch_size, W // self.patch_size

    def _tokens_to_output(self, x2d, B, h, w):
        return ViTEncoderOutput(features=engine.nlc_as_bchw(x2d, B, h, w))

    def forward(self, encoder_input: ViTEncoderInput) -> ViTEncoderOutput:
        tokens, pos, B, h, w = self._embed(encoder_input)
        dt = engine.compute_dtype()
        N = tokens.shape[1]
        x2d = _as_2d(tokens)
        for blk in self.enc_blocks:
            x2d = blk.forward_tokens(x2d, B, N, pos, dt)
        x2d = engine.layernorm(x2d, self.enc_norm, torch.float32, twin=True)
        return self._tokens_to_output(x2d, B, h, w)


class CroCoIntermediateFeatureReturner(CroCoEncoder, IntermediateFeatureReturner):
    "Intermediate Feature Returner for UniCeption CroCo Encoder (croco.py:185-327)"

    def __init__(self, name: str, data_norm_type: str, patch_embed_cls: str = "PatchEmbedDust3R",
                 img_size: Union[int, Tuple[int, int]] = (224, 224), patch_size: int = 16, enc_embed_dim: int = 1024,
                 enc_depth: int = 24, enc_num_heads: int = 16, mlp_ratio: int = 4,
                 norm_layer: Callable = partial(nn.LayerNorm, eps=1e-6), pos_embed: str = "RoPE100",
                 pretrained_checkpoint_path: str = None, indices: Optional[Union[int, List[int]]] = None,
                 norm_intermediate: bool = True, stop_early: bool = False, intermediates_only: bool = True, *args, **kwargs):
        CroCoEncoder.__init__(self, name=name, data_norm_type=data_norm_type, patch_embed_cls=patch_embed_cls,
                              img_size=img_size, patch_size=patch_size, enc_embed_dim=enc_embed_dim, enc_depth=enc_depth,
                              enc_num_heads=enc_num_heads, mlp_ratio=mlp_ratio, norm_layer=norm_layer, pos_embed=pos_embed,
                              pretrained_checkpoint_path=pretrained_checkpoint_path, *args, **kwargs)
        IntermediateFeatureReturner.__init__(self, indices=indices, norm_intermediate=norm_intermediate,
                                             stop_early=stop_early, intermediates_only=intermediates_only)

    def forward(self, encoder_input: ViTEncoderInput):
        tokens, pos, B, h, w = self._embed(encoder_input)
        dt = engine.compute_dtype()
        N = tokens.shape[1]
        take_indices, max_index = feature_take_indices(len(self.enc_blocks), self.indices)
        blocks = self.enc_blocks if not self.stop_early else self.enc_blocks[: max_index + 1]
        x2d = _as_2d(tokens)
        inter = []
        for i, blk in enumerate(blocks):
            x2d = blk.forward_tokens(x2d, B, N, pos, dt)
            if i in take_indices:
                inter.append(engine.layernorm(x2d, self.enc_norm, torch.float32, twin=True) if self.norm_intermediate else x2d)
        inter = [self._tokens_to_output(t, B, h, w) for t in inter]
        if self.intermediates_only:
            return inter
        final = self._tokens_to_output(engine.layernorm(x2d, self.enc_norm, torch.float32, twin=True), B, h, w)
        return final, inter
