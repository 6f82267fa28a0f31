"""Decoder transformer blocks (reference: utils/transformer_blocks.py:38-89, 136-412, 517-647) on the HIP kernels.

CrossAttentionBlock.forward(x, y, xpos, ypos):
    x += proj(SDPA(rope(q), rope(k), v))              q,k,v = qkv(norm1(x))
    x += proj(SDPA(rope(projq(norm2(x)), xpos), rope(projk(norm_y(y)), ypos), projv(norm_y(y))))
    x += fc2(gelu(fc1(norm3(x))))
Each line is: LayerNorm kernel -> GEMM(s) with fused bias/RoPE/VT epilogues -> flash attention ->
GEMM with fused bias+residual.  K and V of the other view share one GEMM (weights concatenated once).
"""
import math
from typing import Callable, Optional

import torch
import torch.nn as nn

from ... import autograd, engine
from ..libs.croco.blocks import DropPath, Mlp, _as_2d, _check_attn_drop, _check_no_dropout, _drop_path_rate, to_2tuple  # noqa: F401
from .config import use_fused_attn


def _softmax_scale_multiplier(module, n_tokens: int) -> float:
    """The reference's optional q-scalings are scalar multipliers of q, i.e. of the softmax scale
    (transformer_blocks.py:231-241, 360-370)."""
    m = 1.0
    if module.use_scalable_softmax:
        m *= math.log(n_tokens)
    if module.use_entropy_scaling:
        m *= math.sqrt(module.entropy_scaling_growth_factor * math.log(n_tokens)
                       / math.log(module.base_token_count_for_entropy_scaling))
    return m


class Attention(nn.Module):
    "Self-Attention Layer"

    def __init__(self, dim: int, latent_attn_dim: Optional[int] = None, num_heads: int = 8, qkv_bias: bool = False,
                 qk_norm: bool = False, attn_drop: float = 0.0, proj_drop: float = 0.0, norm_layer: nn.Module = nn.LayerNorm,
                 custom_positional_encoding: Callable = None, use_scalable_softmax: bool = False,
                 use_entropy_scaling: bool = False, base_token_count_for_entropy_scaling: int = 444,
                 entropy_scaling_growth_factor: float = 1.4):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        # latent attention (utils/transformer_blocks.py:178-199; round 6): q / k / v live in `latent_attn_dim` channels, proj maps back
        if latent_attn_dim is not None:
            assert latent_attn_dim % num_heads == 0, "latent_attn_dim should be divisible by num_heads"
            self.latent_attn_dim = latent_attn_dim
            self.latent_attn = True
        else:
            self.latent_attn = False
        self.num_heads = num_heads
        self.head_dim = dim // num_heads if not self.latent_attn else latent_attn_dim // num_heads
        self.scale = self.head_dim**-0.5
        self.fused_attn = use_fused_attn()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias) if not self.latent_attn else nn.Linear(dim, latent_attn_dim * 3, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim) if not self.latent_attn else nn.Linear(latent_attn_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.custom_positional_encoding = custom_positional_encoding
        self.use_scalable_softmax = use_scalable_softmax
        self.use_entropy_scaling = use_entropy_scaling
        self.base_token_count_for_entropy_scaling = base_token_count_for_entropy_scaling
        self.entropy_scaling_growth_factor = entropy_scaling_growth_factor

    def _run(self, h2d, B, N, xpos, residual, out_dtype, fold=None, emit_ln=False, proj_wb=None):
        _check_no_dropout(self, self.attn_drop.p, self.proj_drop.p)
        if self.custom_positional_encoding is not None:
            assert xpos is not None, "Positions of tokens (xpos) are a required input when using custom positional encoding"
        scale = self.scale * _softmax_scale_multiplier(self, N)
        return engine.self_attention(h2d, B, N, self.qkv, self.proj, self.num_heads, self.custom_positional_encoding, xpos,
                                     scale, residual, out_dtype, proj_wb=proj_wb, fold=fold, emit_ln=emit_ln, q_norm=self.q_norm,
                                     k_norm=self.k_norm)

    def forward(self, x: torch.Tensor, xpos: torch.Tensor = None) -> torch.Tensor:
        engine.require_inference(x, self.qkv.weight)
        B, N, C = x.shape
        dt = engine.compute_dtype()
        x2 = _as_2d(x)
        h = x2 if x2.dtype == dt else engine.ops.convert(x2, dt)
        return self._run(h, B, N, xpos, None, dt).view(B, N, C)


class CrossAttention(nn.Module):
    "Cross-Attention Layer"

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, qk_norm: bool = False, attn_drop: float = 0.0,
                 proj_drop: float = 0.0, norm_layer: nn.Module = nn.LayerNorm, custom_positional_encoding: Callable = None,
                 use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim**-0.5
        self.fused_attn = use_fused_attn()
        self.projq = nn.Linear(dim, dim, bias=qkv_bias)
        self.projk = nn.Linear(dim, dim, bias=qkv_bias)
        self.projv = nn.Linear(dim, dim, bias=qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.custom_positional_encoding = custom_positional_encoding
        self.use_scalable_softmax = use_scalable_softmax
        self.use_entropy_scaling = use_entropy_scaling
        self.base_token_count_for_entropy_scaling = base_token_count_for_entropy_scaling
        self.entropy_scaling_growth_factor = entropy_scaling_growth_factor

    def _run(self, hq2d, hkv2d, B, Nq, Nk, qpos, kpos, residual, out_dtype, fold_q=None, fold_kv=None, emit_ln=False, hv2d=None,
             proj_wb=None):
        _check_no_dropout(self, self.attn_drop.p, self.proj_drop.p)
        if self.custom_positional_encoding is not None:
            assert qpos is not None, "Positions of queries (qpos) are a required input when using custom positional encoding"
            assert kpos is not None, "Positions of keys (kpos) are a required input when using custom positional encoding"
        scale = self.scale * _softmax_scale_multiplier(self, Nq)
        return engine.cross_attention(hq2d, hkv2d, B, Nq, Nk, self.projq, self.projk, self.projv, self.proj, self.num_heads,
                                      self.custom_positional_encoding, qpos, kpos, scale, residual, out_dtype,
                                      fold_q=fold_q, fold_kv=fold_kv, emit_ln=emit_ln, q_norm=self.q_norm, k_norm=self.k_norm, hv2d=hv2d,
                                      proj_wb=proj_wb)

    def forward(self, query, key, value, qpos=None, kpos=None):
        engine.require_inference(query, key, value, self.projq.weight)
        B, Nq, C = query.shape
        Nk = key.shape[1]
        dt = engine.compute_dtype()
        q2, k2 = _as_2d(query), _as_2d(key)
        hq = q2 if q2.dtype == dt else engine.ops.convert(q2, dt)
        hk = k2 if k2.dtype == dt else engine.ops.convert(k2, dt)
        hv = None
        if value is not key:      # (utils/transformer_blocks.py:341-348: projv runs on its own tokens)
            assert value.shape[1] == Nk, "key and value must have the same number of tokens"
            v2 = _as_2d(value)
            hv = v2 if v2.dtype == dt else engine.ops.convert(v2, dt)
        return self._run(hq, hk, B, Nq, Nk, qpos, kpos, None, dt, hv2d=hv).view(B, Nq, C)


class LayerScale(nn.Module):
    """Per-channel scale; parameter container only.  The blocks fold it into the preceding linear's weights
    (gamma * (x W^T + b) = x (gamma W)^T + gamma b: no kernel work); in training the folded weight's gradient is unfolded into
    d W, d b and d gamma (autograd._unfold_layerscale).  SelfAttentionBlock since round 2, CrossAttentionBlock since round 6."""

    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise engine.UcHipError("LayerScale is folded into the block's proj / fc2 weights; it is not callable on its own in the HIP path")


class SelfAttentionBlock(nn.Module):
    """x += ls1(attn(norm1(x))); x += ls2(mlp(norm2(x)))  (reference: utils/transformer_blocks.py:415-514) — the block of the
    global / alternating multi-view transformers.  Same fused pipeline as the CroCo encoder block: LayerNorm folded into the
    QKV / fc1 GEMMs, RoPE + V-transpose in the QKV epilogue, residual adds in the proj / fc2 epilogues."""

    def __init__(self, dim: int, num_heads: int, latent_attn_dim: Optional[int] = None, mlp_ratio: float = 4.0,
                 qkv_bias: bool = False, qk_norm: bool = False, proj_drop: float = 0.0, attn_drop: float = 0.0,
                 init_values: Optional[float] = None, drop_path: float = 0.0, act_layer: nn.Module = nn.GELU,
                 norm_layer: nn.Module = nn.LayerNorm, mlp_layer: nn.Module = Mlp, custom_positional_encoding: Callable = None,
                 use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, latent_attn_dim=latent_attn_dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_norm=qk_norm,
                              attn_drop=attn_drop, proj_drop=proj_drop, norm_layer=norm_layer,
                              custom_positional_encoding=custom_positional_encoding, use_scalable_softmax=use_scalable_softmax,
                              use_entropy_scaling=use_entropy_scaling,
                              base_token_count_for_entropy_scaling=base_token_count_for_entropy_scaling,
                              entropy_scaling_growth_factor=entropy_scaling_growth_factor)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.custom_positional_encoding = custom_positional_encoding

    def forward_tokens(self, x2d, B, N, xpos, dt):
        """[B*N, C] fp32 residual stream in, new residual stream out; attention spans the N tokens of each of the B sequences."""
        if not isinstance(self.mlp, Mlp):
            raise engine.UcHipError("only the standard Mlp layer has a fused HIP pipeline")
        if self.custom_positional_encoding is not None:
            assert xpos is not None, "Positions of tokens (xpos) are a required input when using custom positional encoding"
        sa = self.attn
        if autograd.grad_needed(x2d, *self.parameters()):   # HIP forward + HIP backward sub-layers
            ad = autograd.attn_dropout(self.training, sa.attn_drop.p)       # (round 6: inside the attention kernels, forward and backward)
            C = x2d.shape[1]
            (p1, k1), (p2, k2) = _drop_path_rate(self.drop_path1), _drop_path_rate(self.drop_path2)
            d1 = autograd.make_drops(self.training, x2d.device, B, N, C, p_out=sa.proj_drop.p, p_path=p1, scale_by_keep=k1)
            d2 = autograd.make_drops(self.training, x2d.device, B, N, C, p_out=self.mlp.drop2.p, p_path=p2, hidden=self.mlp.fc1.out_features,
                                     p_mid=self.mlp.drop1.p, scale_by_keep=k2)
            g1 = None if isinstance(self.ls1, nn.Identity) else self.ls1.gamma       # LayerScale: folded weights forward, unfolded gradients
            g2 = None if isinstance(self.ls2, nn.Identity) else self.ls2.gamma
            x2d = autograd.self_attn_sublayer(x2d, self.norm1, sa.qkv, sa.proj, B, N, sa.num_heads, sa.custom_positional_encoding,
                                              xpos, sa.scale * _softmax_scale_multiplier(sa, N), dt, gamma=g1, q_norm=sa.q_norm,
                                              k_norm=sa.k_norm, drops=d1, attn_drop=ad)
            return autograd.mlp_sublayer(x2d, self.norm2, self.mlp.fc1, self.mlp.fc2, engine.act_name(self.mlp.act), dt, gamma=g2, drops=d2)
        _check_no_dropout(self, sa.attn_drop.p, sa.proj_drop.p, self.mlp.drop1.p, self.mlp.drop2.p)
        for dp in (self.drop_path1, self.drop_path2):
            if isinstance(dp, DropPath) and dp.drop_prob > 0 and self.training:
                raise engine.UcHipError("DropPath with drop_prob > 0 in train mode without gradients has no HIP form")
        proj_wb = None if isinstance(self.ls1, nn.Identity) else engine.layerscale_lin_weights(sa.proj, self.ls1.gamma, dt)
        fc2_wb = None if isinstance(self.ls2, nn.Identity) else engine.layerscale_lin_weights(self.mlp.fc2, self.ls2.gamma, dt)
        h, fold = engine.ln_operand(x2d, self.norm1, dt)
        x2d = engine.self_attention(h, B, N, sa.qkv, sa.proj, sa.num_heads, sa.custom_positional_encoding, xpos,
                                    sa.scale * _softmax_scale_multiplier(sa, N), x2d, x2d.dtype, proj_wb=proj_wb, fold=fold, emit_ln=True,
                                    q_norm=sa.q_norm, k_norm=sa.k_norm)
        h, fold = engine.ln_operand(x2d, self.norm2, dt)
        return engine.mlp(h, self.mlp.fc1, self.mlp.fc2, engine.act_name(self.mlp.act), x2d, x2d.dtype, fc2_wb=fc2_wb, fold=fold,
                          emit_ln=True)

    def forward(self, x: torch.Tensor, xpos: torch.Tensor = None) -> torch.Tensor:
        B, N, C = x.shape
        x2 = _as_2d(x)
        if x2.dtype != torch.float32:
            x2 = x2.float()
        return self.forward_tokens(x2, B, N, xpos, engine.compute_dtype()).view(B, N, C)


class CrossAttentionBlock(nn.Module):
    "Cross-Attention Block"

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = False, qk_norm: bool = False,
                 proj_drop: float = 0.0, attn_drop: float = 0.0, init_values: Optional[float] = None, drop_path: float = 0.0,
                 act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = nn.LayerNorm, mlp_layer: nn.Module = Mlp,
                 custom_positional_encoding: Callable = None, norm_cross_tokens: bool = True,
                 use_scalable_softmax: bool = False, use_entropy_scaling: bool = False,
                 base_token_count_for_entropy_scaling: int = 444, entropy_scaling_growth_factor: float = 1.4):
        super().__init__()
        akw = dict(num_heads=num_heads, qkv_bias=qkv_bias, qk_norm=qk_norm, attn_drop=attn_drop, proj_drop=proj_drop,
                   norm_layer=norm_layer, custom_positional_encoding=custom_positional_encoding,
                   use_scalable_softmax=use_scalable_softmax, use_entropy_scaling=use_entropy_scaling,
                   base_token_count_for_entropy_scaling=base_token_count_for_entropy_scaling,
                   entropy_scaling_growth_factor=entropy_scaling_growth_factor)
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, **akw)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm_y = norm_layer(dim) if norm_cross_tokens else nn.Identity()
        self.custom_positional_encoding = custom_positional_encoding
        self.norm2 = norm_layer(dim)
        self.cross_attn = CrossAttention(dim, **akw)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm3 = norm_layer(dim)
        self.mlp = mlp_layer(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=proj_drop)
        self.ls3 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path3 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def _gammas(self):
        "LayerScale parameters of the three sub-layers (None where init_values was None)."
        return tuple(None if isinstance(ls, nn.Identity) else ls.gamma for ls in (self.ls1, self.ls2, self.ls3))

    def _check_supported(self):
        if not isinstance(self.mlp, Mlp):
            raise engine.UcHipError("only the standard Mlp layer has a fused HIP pipeline")

    def forward_tokens(self, x2d, y2d, B, Nx, Ny, xpos, ypos, dt):
        """x2d [B*Nx, C] residual stream, y2d [B*Ny, C] other-view tokens (previous depth) -> new x2d."""
        self._check_supported()
        if self.custom_positional_encoding is not None:
            assert xpos is not None, "Positions of tokens (xpos) are a required input when using custom positional encoding"
            assert ypos is not None, "Positions of cross tokens (ypos) are a required input when using custom positional encoding"
        if autograd.grad_needed(x2d, y2d, *self.parameters()):
            return self._forward_tokens_train(x2d, y2d, B, Nx, Ny, xpos, ypos, dt)
        for dp in (self.drop_path1, self.drop_path2, self.drop_path3):
            if isinstance(dp, DropPath) and dp.drop_prob > 0 and self.training:
                raise engine.UcHipError("DropPath with drop_prob > 0 in train mode without gradients has no HIP form")
        # LayerNorm -> GEMM pairs run fused when a stream carries its producer's bf16 twin + row statistics (engine.ln_operand)
        g1, g2, g3 = self._gammas()        # LayerScale (utils/transformer_blocks.py:584-647): folded into proj / fc2, no kernel work
        wb1 = None if g1 is None else engine.layerscale_lin_weights(self.attn.proj, g1, dt)
        wb2 = None if g2 is None else engine.layerscale_lin_weights(self.cross_attn.proj, g2, dt)
        wb3 = None if g3 is None else engine.layerscale_lin_weights(self.mlp.fc2, g3, dt)
        h, fold = engine.ln_operand(x2d, self.norm1, dt)
        x2d = self.attn._run(h, B, Nx, xpos, x2d, x2d.dtype, fold, True, proj_wb=wb1)
        if isinstance(self.norm_y, nn.Identity):
            yn, fold_y = (y2d if y2d.dtype == dt else engine.ops.convert(y2d, dt)), None
        else:
            yn, fold_y = engine.ln_operand(y2d, self.norm_y, dt)
        h, fold = engine.ln_operand(x2d, self.norm2, dt)
        x2d = self.cross_attn._run(h, yn, B, Nx, Ny, xpos, ypos, x2d, x2d.dtype, fold, fold_y, True, proj_wb=wb2)
        h, fold = engine.ln_operand(x2d, self.norm3, dt)
        return self.mlp._run(h, x2d, x2d.dtype, fold, True, fc2_wb=wb3)

    def _forward_tokens_train(self, x2d, y2d, B, Nx, Ny, xpos, ypos, dt):
        """Same three sub-layers as autograd Functions (HIP forward + HIP backward)."""
        sa, ca = self.attn, self.cross_attn
        ad1 = autograd.attn_dropout(self.training, sa.attn_drop.p)          # (round 6: inside the attention kernels, forward and backward)
        ad2 = autograd.attn_dropout(self.training, ca.attn_drop.p)
        C, dev = x2d.shape[1], x2d.device
        (p1, k1), (p2, k2), (p3, k3) = (_drop_path_rate(m) for m in (self.drop_path1, self.drop_path2, self.drop_path3))
        d1 = autograd.make_drops(self.training, dev, B, Nx, C, p_out=sa.proj_drop.p, p_path=p1, scale_by_keep=k1)
        d2 = autograd.make_drops(self.training, dev, B, Nx, C, p_out=ca.proj_drop.p, p_path=p2, scale_by_keep=k2)
        d3 = autograd.make_drops(self.training, dev, B, Nx, C, p_out=self.mlp.drop2.p, p_path=p3, hidden=self.mlp.fc1.out_features,
                                 p_mid=self.mlp.drop1.p, scale_by_keep=k3)
        rope = self.custom_positional_encoding
        g1, g2, g3 = self._gammas()
        x2d = autograd.self_attn_sublayer(x2d, self.norm1, sa.qkv, sa.proj, B, Nx, sa.num_heads, rope, xpos,
                                          sa.scale * _softmax_scale_multiplier(sa, Nx), dt, gamma=g1, q_norm=sa.q_norm, k_norm=sa.k_norm,
                                          drops=d1, attn_drop=ad1)
        lny = None if isinstance(self.norm_y, nn.Identity) else self.norm_y
        x2d = autograd.cross_attn_sublayer(x2d, y2d, self.norm2, lny, ca, B, Nx, Ny, ca.num_heads, rope, xpos, ypos,
                                           ca.scale * _softmax_scale_multiplier(ca, Nx), dt, gamma=g2, drops=d2, attn_drop=ad2)
        return autograd.mlp_sublayer(x2d, self.norm3, self.mlp.fc1, self.mlp.fc2, engine.act_name(self.mlp.act), dt, gamma=g3, drops=d3)

    def forward(self, x, y, xpos=None, ypos=None):
        B, Nx, C = x.shape
        Ny = y.shape[1]
        out = self.forward_tokens(_as_2d(x), _as_2d(y), B, Nx, Ny, xpos, ypos, engine.compute_dtype())
        return out.view(B, Nx, C)
