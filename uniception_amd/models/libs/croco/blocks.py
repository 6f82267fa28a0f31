"""CroCo encoder blocks (reference: libs/croco/blocks.py:37-161) on the HIP kernel library.

The nn.Linear / nn.LayerNorm children are parameter containers (identical state_dict keys, shapes and init);
`forward` never calls them — it hands their weights to the fused pipelines in uniception_amd.engine:
    LN -> [QKV GEMM + bias + RoPE-2D + V-transpose epilogue] -> flash attention -> [proj GEMM + bias + residual]
    LN -> [fc1 GEMM + bias + erf-GELU] -> [fc2 GEMM + bias + residual]
"""
import collections.abc
from itertools import repeat

import torch
import torch.nn as nn

from .... import autograd, engine
from ...utils.config import use_fused_attn

use_torch_attn = use_fused_attn()


def _ntuple(n):
    def parse(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return x
        return tuple(repeat(x, n))

    return parse


to_2tuple = _ntuple(2)


class DropPath(nn.Module):
    """Stochastic depth; identity in eval / at rate 0, which is all the HIP inference path supports."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        # (in training the blocks apply it inside their sub-layer Functions: autograd.make_drops / uc_mask_scale)
        raise engine.UcHipError("DropPath with drop_prob > 0 is applied by the blocks' training path; called on its own (or without "
                                "gradients in train mode) it has no HIP form")

    def extra_repr(self):
        return f"drop_prob={round(self.drop_prob, 3):0.3f}"


def _check_no_dropout(module, *ps):
    if module.training and any(p > 0.0 for p in ps):
        raise engine.UcHipError("dropout > 0 in training mode is not supported by the HIP path")


def _drop_path_rate(m):
    "(rate, scale_by_keep) of a block's drop_path module (nn.Identity: 0)"
    return (m.drop_prob, m.scale_by_keep) if isinstance(m, DropPath) else (0.0, True)


def _check_attn_drop(module, p):
    "Kept for callers outside the blocks: attention dropout runs inside the sub-layer Functions since round 6 (autograd.attn_dropout)."
    if module.training and p > 0.0 and not torch.is_grad_enabled():
        raise engine.UcHipError("attn_drop > 0 in train mode without gradients has no HIP form (the inference kernels apply no dropout)")


def _as_2d(x):
    B, N, C = x.shape
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise engine.UcHipError(f"token dtype {x.dtype} not supported (fp32 or bf16)")
    return engine.carry_ln(x, x.reshape(B * N, C)) if x.is_contiguous() else x.contiguous().view(B * N, C)


class Mlp(nn.Module):
    """fc1 -> act -> fc2 (blocks.py:64-86)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        bias = to_2tuple(bias)
        drop_probs = to_2tuple(drop)
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias[0])
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop_probs[0])
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias[1])
        self.drop2 = nn.Dropout(drop_probs[1])

    def _run(self, h2d, residual, out_dtype, fold=None, emit_ln=False, fc2_wb=None):
        _check_no_dropout(self, self.drop1.p, self.drop2.p)
        return engine.mlp(h2d, self.fc1, self.fc2, engine.act_name(self.act), residual, out_dtype, fc2_wb=fc2_wb, fold=fold, emit_ln=emit_ln)

    def forward(self, x):
        engine.require_inference(x, self.fc1.weight)
        dt = engine.compute_dtype()
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        h = x2 if x2.dtype == dt else engine.ops.convert(x2.contiguous(), dt)
        return self._run(h.contiguous(), None, dt).view(*shp[:-1], -1)


class Attention(nn.Module):
    """MHSA with RoPE on q,k (blocks.py:89-130)."""

    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, torch_attn=use_torch_attn):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim**-0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.rope = rope
        self.torch_attn = torch_attn
        self.dropout_p = attn_drop

    def _run(self, h2d, B, N, xpos, residual, out_dtype, fold=None, emit_ln=False):
        _check_no_dropout(self, self.dropout_p, self.proj_drop.p)
        return engine.self_attention(h2d, B, N, self.qkv, self.proj, self.num_heads, self.rope, xpos, self.scale,
                                     residual, out_dtype, fold=fold, emit_ln=emit_ln)

    def forward(self, x, xpos):
        engine.require_inference(x, self.qkv.weight)
        B, N, C = x.shape
        dt = engine.compute_dtype()
        x2 = _as_2d(x)
        h = x2 if x2.dtype == dt else engine.ops.convert(x2, dt)
        return self._run(h, B, N, xpos, None, dt).view(B, N, C)


class Block(nn.Module):
    """x += attn(norm1(x)); x += mlp(norm2(x)) (blocks.py:133-161)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, rope=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, rope=rope, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)

    def forward_tokens(self, x2d, B, N, xpos, dt):
        """[B*N, C] residual stream in, new residual stream out (same dtype)."""
        if autograd.grad_needed(x2d, *self.parameters()):
            # dropout (round 5): proj_drop / the Mlp's drops / DropPath as masks through the sub-layer Functions (autograd.make_drops)
            ad = autograd.attn_dropout(self.training, self.attn.dropout_p)      # (round 6: inside the attention kernels, forward and backward)
            C = x2d.shape[1]
            pp, sbk = _drop_path_rate(self.drop_path)
            d1 = autograd.make_drops(self.training, x2d.device, B, N, C, p_out=self.attn.proj_drop.p, p_path=pp, scale_by_keep=sbk)
            d2 = autograd.make_drops(self.training, x2d.device, B, N, C, p_out=self.mlp.drop2.p, p_path=pp, hidden=self.mlp.fc1.out_features,
                                     p_mid=self.mlp.drop1.p, scale_by_keep=sbk)
            x2d = autograd.self_attn_sublayer(x2d, self.norm1, self.attn.qkv, self.attn.proj, B, N, self.attn.num_heads,
                                              self.attn.rope, xpos, self.attn.scale, dt, drops=d1, attn_drop=ad)
            return autograd.mlp_sublayer(x2d, self.norm2, self.mlp.fc1, self.mlp.fc2, engine.act_name(self.mlp.act), dt, drops=d2)
        if isinstance(self.drop_path, DropPath):
            self.drop_path(x2d)  # (train mode without gradients: raises at rate > 0)
        # LayerNorm -> GEMM pairs run fused when the stream carries its producer's bf16 twin + row statistics (engine.ln_operand)
        h, fold = engine.ln_operand(x2d, self.norm1, dt)
        x2d = self.attn._run(h, B, N, xpos, x2d, x2d.dtype, fold, True)
        h, fold = engine.ln_operand(x2d, self.norm2, dt)
        return self.mlp._run(h, x2d, x2d.dtype, fold, True)

    def forward(self, x, xpos):
        B, N, C = x.shape
        return self.forward_tokens(_as_2d(x), B, N, xpos, engine.compute_dtype()).view(B, N, C)
