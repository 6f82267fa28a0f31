"""DINOv2 encoder (reference wrapper: encoders/dinov2.py:15-327) on the HIP kernels — BASELINE config 4.

The reference obtains the network itself from `torch.hub.load("facebookresearch/dinov2", "dinov2_vit{s,b,l}14[_reg]")`
(encoders/dinov2.py:91-102): third-party code that is NOT vendored under the reference tree and cannot be fetched here.
`DinoVisionTransformerParams` below therefore restates the PUBLISHED DINOv2 ViT architecture (ViT-S/B/L-14: patch 14,
cls token, learned position embedding stored for a 37x37 grid and bicubically resized to other grids, optional 4 register
tokens, pre-LN blocks with LayerScale, erf-GELU MLP, final LayerNorm eps 1e-6) with the hub checkpoints' parameter names so
their `state_dict`s load unchanged.  Pinned, at the native 37x37 grid, to an independent implementation of the same network
(HuggingFace transformers' Dinov2Model / Dinov2WithRegistersModel: tests/golden/dinov2_hf.npz); PARITY UNPINNED for resized
position embeddings (other grids), where the GPU tests check this module against the oracle's restatement of the hub code only.

Kernel mapping: patch gather + GEMM (K = 588: fp32 GEMM, 0.1 % of the FLOPs) -> uc_assemble_tokens (cls + pos, registers,
patches + pos) -> per block: LayerNorm -> QKV GEMM with the VT epilogue (no RoPE) -> flash attention -> proj GEMM with
LayerScale folded into its weights + residual -> LayerNorm -> fc1 GEMM + GELU -> fc2 GEMM (LayerScale folded) + residual
-> final LayerNorm -> uc_token_slice (patch tokens / cls+registers).
Training (gradients requested through any parameter): the blocks run as HIP forward + backward sub-layers (autograd.py; LayerScale
folded forward, unfolded gradients), the patch embedding as autograd.patch_embed; the token assembly (cls / registers / position
embedding, incl. its bicubic resize) and the output split are a handful of torch ops on small tensors, so that cls_token,
register_tokens and pos_embed receive their gradients from autograd.
"""
import math
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import autograd, engine, ops
from ..utils.intermediate_feature_return import IntermediateFeatureReturner, feature_take_indices
from .base import UniCeptionViTEncoderBase, ViTEncoderInput, ViTEncoderOutput

_SIZES = {"small": (384, 12, 6), "base": (768, 12, 12), "large": (1024, 24, 16), "giant": (1536, 40, 24)}   # embed dim, depth, heads
_SWIGLU = {"giant"}      # vit_giant2: ffn_layer="swiglufused" (the hub's SwiGLUFFNFused), every other size the GELU MLP


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=True)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim, bias=True)


class _SwiGLUFFN(nn.Module):
    """The hub's SwiGLUFFNFused under its parameter names: x1, x2 = w12(x).chunk(2); w3(silu(x1) * x2), hidden width
    (int(4 dim * 2 / 3) + 7) // 8 * 8 (giant: 4096)."""

    def __init__(self, dim, hidden):
        super().__init__()
        hidden = (int(hidden * 2 / 3) + 7) // 8 * 8
        self.w12 = nn.Linear(dim, 2 * hidden, bias=True)
        self.w3 = nn.Linear(hidden, dim, bias=True)


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Block(nn.Module):
    "x += ls1(attn(norm1(x))); x += ls2(mlp(norm2(x)))  — parameter container; the fused pipeline is in forward_tokens."

    def __init__(self, dim, num_heads, mlp_ratio=4.0, swiglu=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _SwiGLUFFN(dim, int(dim * mlp_ratio)) if swiglu else _Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LayerScale(dim)

    def forward_tokens(self, x2d, B, N, dt):
        if autograd.grad_needed(x2d, *self.parameters()):
            x2d = autograd.self_attn_sublayer(x2d, self.norm1, self.attn.qkv, self.attn.proj, B, N, self.attn.num_heads, None, None,
                                              self.attn.scale, dt, gamma=self.ls1.gamma)
            if isinstance(self.mlp, _SwiGLUFFN):
                return autograd.swiglu_sublayer(x2d, self.norm2, self.mlp.w12, self.mlp.w3, dt, gamma=self.ls2.gamma)
            return autograd.mlp_sublayer(x2d, self.norm2, self.mlp.fc1, self.mlp.fc2, "gelu", dt, gamma=self.ls2.gamma)
        # (the same pipeline as utils/transformer_blocks.SelfAttentionBlock: LayerNorms folded into the QKV / fc1 GEMMs once a producer
        # GEMM has left the row statistics — from the first proj on —, LayerScale folded into the proj / fc2 weights)
        h, fold = engine.ln_operand(x2d, self.norm1, dt)
        x2d = engine.self_attention(h, B, N, self.attn.qkv, self.attn.proj, self.attn.num_heads, None, None, self.attn.scale,
                                    x2d, x2d.dtype, proj_wb=engine.layerscale_lin_weights(self.attn.proj, self.ls1.gamma, dt),
                                    fold=fold, emit_ln=True)
        h, fold = engine.ln_operand(x2d, self.norm2, dt)
        if isinstance(self.mlp, _SwiGLUFFN):
            return engine.mlp_swiglu(h, self.mlp.w12, self.mlp.w3, x2d, x2d.dtype,
                                     w3_wb=engine.layerscale_lin_weights(self.mlp.w3, self.ls2.gamma, dt), fold=fold, emit_ln=True)
        return engine.mlp(h, self.mlp.fc1, self.mlp.fc2, "gelu", x2d, x2d.dtype,
                          fc2_wb=engine.layerscale_lin_weights(self.mlp.fc2, self.ls2.gamma, dt), fold=fold, emit_ln=True)


class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch_size, stride=patch_size)


class DinoVisionTransformerParams(nn.Module):
    "Parameters of a DINOv2 ViT under the hub checkpoints' names (`model.*` inside DINOv2Encoder)."

    def __init__(self, size: str, patch_size: int, num_register_tokens: int, pretrain_grid: int = 37):
        super().__init__()
        dim, depth, heads = _SIZES[size]
        self.embed_dim, self.patch_size, self.num_register_tokens = dim, patch_size, num_register_tokens
        # hub settings: *_reg models interpolate with antialias and no offset, the others with the 0.1 offset kludge
        self.interpolate_antialias = num_register_tokens > 0
        self.interpolate_offset = 0.0 if num_register_tokens > 0 else 0.1
        self.patch_embed = _PatchEmbed(patch_size, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + pretrain_grid * pretrain_grid, dim))
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, dim)) if num_register_tokens else None
        self.blocks = nn.ModuleList([_Block(dim, heads, swiglu=size in _SWIGLU) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        if self.register_tokens is not None:
            nn.init.normal_(self.register_tokens, std=1e-6)

    def _resized_pos_embed(self, pe: torch.Tensor, h0: int, w0: int) -> torch.Tensor:
        "[1+h0*w0, D]: class position + the patch grid resized bicubically (published `interpolate_pos_encoding`); differentiable."
        N = pe.shape[1] - 1
        M = int(math.sqrt(N))
        assert N == M * M
        if (h0, w0) == (M, M):
            return pe[0]
        grid = pe[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2)
        if self.interpolate_offset:
            kw = dict(scale_factor=(float(h0 + self.interpolate_offset) / M, float(w0 + self.interpolate_offset) / M))
        else:
            kw = dict(size=(h0, w0))
        grid = F.interpolate(grid, mode="bicubic", antialias=self.interpolate_antialias, **kw)
        assert tuple(grid.shape[-2:]) == (h0, w0)
        return torch.cat([pe[0, :1], grid.permute(0, 2, 3, 1).reshape(h0 * w0, -1)], 0)

    def interpolated_pos_embed(self, h0: int, w0: int) -> torch.Tensor:
        "Inference: the resized table, cached per grid and per version of pos_embed."
        return engine.prepared(self, ("pos", h0, w0), (self.pos_embed,),
                               lambda: self._resized_pos_embed(self.pos_embed.detach().float(), h0, w0).contiguous())


class DINOv2Encoder(UniCeptionViTEncoderBase):
    "UniCeption DINOv2 Encoder (encoders/dinov2.py:15-216)"

    def __init__(self, name: str, data_norm_type: str = "dinov2", patch_size: int = 14, size: str = "large",
                 with_registers: bool = False, norm_returned_features: bool = True, pretrained_checkpoint_path: str = None,
                 torch_hub_force_reload: bool = False, torch_hub_pretrained: bool = True, gradient_checkpointing: bool = False,
                 keep_first_n_layers: Optional[int] = None, use_pytorch_sdpa=True, disable_torch_compile_for_pe=False,
                 *args, **kwargs):
        name = name if not with_registers else f"{name}_reg"
        super().__init__(name=name, data_norm_type=data_norm_type, patch_size=patch_size,
                         gradient_checkpointing=gradient_checkpointing, *args, **kwargs)
        if size not in _SIZES:
            raise engine.UcHipError(f"unknown DINOv2 size '{size}'; supported: {sorted(_SIZES)}")
        self.version = size
        self.with_registers = with_registers
        self.norm_returned_features = norm_returned_features
        self.enc_embed_dim = _SIZES[size][0]
        # there is no torch.hub here: the architecture is built directly; weights come from pretrained_checkpoint_path
        # (a {"model": state_dict} file with the hub names under "model.") or stay at their initialisation
        self.model = DinoVisionTransformerParams(size, patch_size, 4 if with_registers else 0)
        if not norm_returned_features:
            self.model.norm = nn.Identity()
        if keep_first_n_layers is not None:
            self.model.blocks = nn.ModuleList(self.model.blocks[:keep_first_n_layers])
        if pretrained_checkpoint_path:
            print(f"Loading custom pretrained DINOv2 checkpoint from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))
        if self.gradient_checkpointing:       # (encoders/dinov2.py:125-127)
            for i in range(len(self.model.blocks)):
                self.model.blocks[i] = self.wrap_module_with_gradient_checkpointing(self.model.blocks[i])

    # ---- token-stream core ---------------------------------------------------------------------
    def _check(self, encoder_input):
        self._check_data_normalization_type(encoder_input.data_norm_type)
        assert isinstance(encoder_input.image, torch.Tensor), "Input must be a torch.Tensor"
        assert encoder_input.image.ndim == 4, "Input must be of shape (B, C, H, W)"
        B, C, H, W = encoder_input.image.shape
        assert C == 3, "Input must have 3 channels"
        assert H % self.patch_size == 0 and W % self.patch_size == 0, \
            f"Input shape must be divisible by patch size: {self.patch_size}"
        return B, H // self.patch_size, W // self.patch_size

    def _train(self, image) -> bool:
        return autograd.grad_needed(image, *self.model.parameters())

    def _tokens(self, image, h0, w0):
        m = self.model
        P = self.patch_size
        dt = engine.compute_dtype()
        pdt = dt if (3 * P * P) % 8 == 0 else torch.float32
        img = image.float().contiguous() if (image.dtype != torch.float32 or not image.is_contiguous()) else image
        if self._train(image):
            B = image.shape[0]
            tok = autograd.patch_embed(img, m.patch_embed.proj, P, pdt).view(B, h0 * w0, -1)
            pe = m._resized_pos_embed(m.pos_embed.float(), h0, w0)
            parts = [(m.cls_token.float() + pe[:1]).expand(B, -1, -1)]
            if m.register_tokens is not None:          # registers are inserted AFTER the position embedding was added: none for them
                parts.append(m.register_tokens.float().expand(B, -1, -1))
            parts.append(tok + pe[1:])
            return torch.cat(parts, dim=1).contiguous(), dt
        K = 3 * P * P
        if dt == torch.bfloat16 and K % 64 != 0:
            # bf16 mode, patch 14: K = 588 is not a multiple of 8 — the plain route is the fp32 GEMM (39 TFLOP/s: 2.7 ms of a 140-ms step at
            # 32 pairs of 518 x 518).  Patches gathered in bf16 (the CroCo patch embed's policy in this mode) and K zero-padded to 640:
            # the MFMA kernel, < 0.3 ms
            kpad = (K + 63) // 64 * 64
            cols = ops.patch_gather(img, P, torch.bfloat16)
            colsp = torch.zeros((cols.shape[0], kpad), dtype=torch.bfloat16, device=cols.device)
            colsp[:, :K] = cols
            w, b = engine.patch_weights_padded(m.patch_embed.proj, torch.bfloat16, kpad)
            tok = ops.gemm(colsp, w, b, out_dtype=torch.float32).view(image.shape[0], h0 * w0, -1)
        else:
            cols = ops.patch_gather(img, P, pdt)
            w, b = engine.patch_weights(m.patch_embed.proj, pdt)
            tok = ops.gemm(cols, w, b, out_dtype=torch.float32).view(image.shape[0], h0 * w0, -1)
        cls = engine.prepared(m, "cls", (m.cls_token,), lambda: m.cls_token.detach().float().reshape(-1).contiguous())
        reg = None
        if m.register_tokens is not None:
            reg = engine.prepared(m, "reg", (m.register_tokens,), lambda: m.register_tokens.detach().float()[0].contiguous())
        x = ops.assemble_tokens(tok, cls, reg, m.interpolated_pos_embed(h0, w0))
        if engine.stream_dtype(dt, self.enc_embed_dim) == torch.bfloat16:      # bf16 residual stream (engine.stream_dtype): one cast here,
            x = ops.convert(x, torch.bfloat16)                                  # 4 instead of 10 bytes per element in every sub-layer after it
        return x, dt

    def _final_norm(self, x2d):
        if isinstance(self.model.norm, nn.Identity):
            return x2d if (x2d.dtype == torch.float32 or x2d.requires_grad) else ops.convert(x2d, torch.float32)
        return engine.layernorm(x2d, self.model.norm, torch.float32)

    def _split(self, xn, B, h0, w0):
        "normed [B,Nt,D] -> (features BCHW view, registers [B,D,1+R])"
        R = self.model.num_register_tokens
        if xn.requires_grad:      # training: plain (differentiable) slices
            return xn[:, 1 + R:].reshape(B, h0, w0, -1).permute(0, 3, 1, 2), xn[:, :1 + R].permute(0, 2, 1).contiguous()
        patches = ops.token_slice(xn, 1 + R, h0 * w0)
        extra = ops.token_slice(xn, 0, 1 + R)
        return engine.nlc_as_bchw(patches.view(B * h0 * w0, -1), B, h0, w0), extra.permute(0, 2, 1).contiguous()

    def forward(self, encoder_input: ViTEncoderInput) -> ViTEncoderOutput:
        B, h0, w0 = self._check(encoder_input)
        x, dt = self._tokens(encoder_input.image, h0, w0)
        Nt, D = x.shape[1], x.shape[2]
        x2d = x.view(B * Nt, D)
        for blk in self.model.blocks:
            x2d = blk.forward_tokens(x2d, B, Nt, dt)
        feats, regs = self._split(self._final_norm(x2d).view(B, Nt, D), B, h0, w0)
        return ViTEncoderOutput(features=feats, registers=regs)


class DINOv2IntermediateFeatureReturner(DINOv2Encoder, IntermediateFeatureReturner):
    "Intermediate Feature Returner for UniCeption DINOv2 Encoder (encoders/dinov2.py:219-327)"

    def __init__(self, name: str, data_norm_type: str = "dinov2", patch_size: int = 14, size: str = "large",
                 with_registers: bool = False, pretrained_checkpoint_path: str = None, torch_hub_force_reload: bool = False,
                 gradient_checkpointing: bool = False, keep_first_n_layers: Optional[int] = None, use_pytorch_sdpa=True,
                 disable_torch_compile_for_pe=False, indices: Optional[Union[int, List[int]]] = 1, norm_intermediate: bool = True,
                 *args, **kwargs):
        DINOv2Encoder.__init__(self, name=name, data_norm_type=data_norm_type, patch_size=patch_size, size=size,
                               with_registers=with_registers, pretrained_checkpoint_path=pretrained_checkpoint_path,
                               torch_hub_force_reload=torch_hub_force_reload, gradient_checkpointing=gradient_checkpointing,
                               keep_first_n_layers=keep_first_n_layers, use_pytorch_sdpa=use_pytorch_sdpa,
                               disable_torch_compile_for_pe=disable_torch_compile_for_pe, *args, **kwargs)
        IntermediateFeatureReturner.__init__(self, indices=indices, norm_intermediate=norm_intermediate)

    def forward(self, encoder_input: ViTEncoderInput) -> List[ViTEncoderOutput]:
        """Published `get_intermediate_layers(x, n=indices, reshape=True, norm=..., return_class_token=True)`: the outputs of
        the selected blocks, optionally through the final norm, as (patch map, class token) pairs."""
        B, h0, w0 = self._check(encoder_input)
        if self.indices is None:
            self.indices = range(len(self.model.blocks))
        take, _ = feature_take_indices(len(self.model.blocks), self.indices)
        x, dt = self._tokens(encoder_input.image, h0, w0)
        Nt, D = x.shape[1], x.shape[2]
        R = self.model.num_register_tokens
        x2d = x.view(B * Nt, D)
        outs = []
        for i, blk in enumerate(self.model.blocks):
            x2d = blk.forward_tokens(x2d, B, Nt, dt)
            if i in take:
                xn = self._final_norm(x2d) if self.norm_intermediate else x2d
                if xn.dtype != torch.float32 and not xn.requires_grad:
                    xn = ops.convert(xn, torch.float32)
                xn = xn.view(B, Nt, D)
                if xn.requires_grad:
                    outs.append(ViTEncoderOutput(features=xn[:, 1 + R:].reshape(B, h0, w0, D).permute(0, 3, 1, 2),
                                                 registers=xn[:, :1].permute(0, 2, 1).contiguous()))
                    continue
                patches = ops.token_slice(xn, 1 + R, h0 * w0)
                cls = ops.token_slice(xn, 0, 1)
                outs.append(ViTEncoderOutput(features=engine.nlc_as_bchw(patches.view(B * h0 * w0, D), B, h0, w0),
                                             registers=cls.permute(0, 2, 1).contiguous()))
        return outs
