"""Training path of the DUSt3R hot path: ``torch.autograd.Function``s whose forward AND backward are the HIP kernels.

The reference trains through PyTorch autograd over its modules (it ships no hand-written backward); here autograd only
wires the graph (residual stream, the Jacobi update between the two views, the view ops at the BCHW boundary) while every
sub-layer's backward is one Function calling uc_hip entry points:

    pre-LN sub-layer  x_out = x + f(LN(x)):   backward gets d(x_out) and returns
        dx = LN_bwd(x, gamma, df/dh) + d(x_out)              (residual add fused into uc_layernorm_bwd)
    linear y = h W^T + b:
        dW = dy^T h   -> uc_gemm_tn (both operands row-major, transposing LDS reads, split-K slabs + uc_splitk_reduce —
                         straight into the flat gradient buffer under the Trainer); fp32 mode: uc_transpose2d + fp32 uc_gemm
        db = column sums of dy formed inside uc_gemm_tn; dh = uc_gemm(dy, W^T)   (W^T prepared once per weight version)
    attention: uc_attention_fwd saves LSE; uc_attention_bwd recomputes P tile by tile (dQ kernel + dK/dV kernel)
    RoPE: gradients of the rotated q/k are rotated back in place with the inverse angle (curope2d.py:24-28)

Precision follows the forward: bf16 operands with fp32 accumulation, fp32 residual stream, fp32 weight gradients; in
fp32 verification mode every kernel is the exact-fp32 variant.
"""
import weakref
from typing import Optional

import torch
import torch.nn as nn
from torch.autograd import Function

from . import engine, ops
from ._lib import UcHipError

KPAD = 64  # the weight-gradient GEMMs reduce over tokens: pad that axis to the direct-to-LDS kernel's K granule


def grad_needed(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _c(g: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if g is None else (g if g.is_contiguous() else g.contiguous())


def _tp(x2d: torch.Tensor, dt: torch.dtype, with_copy: bool = False):
    return ops.transpose2d(x2d, dt, pad_to=KPAD, with_copy=with_copy)


def _split_k(I: int, J: int, T: int) -> int:
    """Few 256x256 output tiles, very long reduction: split it so tiles*sk just fills the 256 CUs once."""
    tiles = ((I + 255) // 256) * ((J + 255) // 256)
    return max(1, min(T // 512, 256 // tiles))


def _split_k_x3(I: int, J: int, T: int, dt) -> int:
    """split_k of an fp32 weight-gradient GEMM on the split-operand (bf16x3) route: its output is a few 128 x 128 tiles and its
    reduction runs over every token / pixel (x3) — unsplit, the fp32-class heads' 3x3 weight gradients ran on 9 CUs for 44 ms.
    1 on the exact-fp32 route (the verification kernel has no split form)."""
    if dt != torch.float32 or ops.fp32_matmul_hook() != "bf16x3":
        return 1
    tiles = ((I + 127) // 128) * ((J + 127) // 128)
    return max(1, min(T // 1024, 256 // max(tiles, 1), 64))


def _tn_ok(*ts) -> bool:
    return all(t.dtype == torch.bfloat16 and t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and t.shape[-1] % 8 == 0
               and (t.dim() != 2 or t.stride(0) % 8 == 0) for t in ts)


def _reduce_slabs(ws: torch.Tensor) -> torch.Tensor:
    return ws[0] if ws.shape[0] == 1 else ops.splitk_reduce(ws)


# Gradient sink (switched on by training.Trainer): weight gradients that come out of the TN kernel as split-K slabs are
# reduced STRAIGHT INTO the parameter's existing .grad (a view of the flat gradient buffer) — the reduction kernel does
# autograd's accumulation and the Function returns None for that input.  The parameter's AccumulateGrad node still runs
# once all of its contributions are in (with an undefined gradient) and fires its post-accumulate hooks, so the bucket
# bookkeeping of the trainer needs no extra signal; should a PyTorch build skip the hook for undefined gradients, the
# trainer's finish() reduces the buckets that never completed.
_grad_sink = False
# Two-stream training with SHARED parameters (the encoder's two views on two streams, engine.run_branches(shared_params=True)): the
# sink's reduction is a plain read-modify-write of the parameter's gradient slice, and two streams doing that concurrently would race.
# Functions whose forward ran inside the forked (second) branch therefore return their weight gradients as tensors — autograd's
# AccumulateGrad adds them on the stream the parameter was first used on, behind the first branch's sink writes.  `_sink_fwd_ok` is what
# a Function records at forward time (ctx._uc_sink_ok, see _sink_aware); `_grad_sink_gate` is that record during its backward.
_sink_fwd_ok = [True]
_grad_sink_gate = True


def set_grad_sink(enabled) -> None:
    global _grad_sink
    _grad_sink = bool(enabled)


def _sink_aware(cls):
    "Class decorator of the autograd Functions below: remember at forward time whether this node may sink, apply it in backward."
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kw):
        ctx._uc_sink_ok = _sink_fwd_ok[0]
        # how fp32-operand GEMMs ran in this forward (exact VALU kernel / split-operand bf16x3 MFMA: engine._fp32_matmul).  The backward
        # is called by the autograd engine OUTSIDE the caller's engine.precision(...) scope, where the policy would read "exact": the
        # fp32-class heads' training step ran 86 % of its time in gemm_f32_kernel (round 6, 14x the bf16-head step) — the backward now
        # runs its fp32 GEMMs the way the forward did
        ctx._uc_mm = ops.fp32_matmul_hook()
        return fwd(ctx, *args, **kw)

    def backward(ctx, *grads):
        global _grad_sink_gate
        prev = _grad_sink_gate
        _grad_sink_gate = getattr(ctx, "_uc_sink_ok", True)
        prev_mm, engine._mm_override = engine._mm_override, getattr(ctx, "_uc_mm", None)
        try:
            return bwd(ctx, *grads)
        finally:
            _grad_sink_gate = prev
            engine._mm_override = prev_mm
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _sinkable(param, rows: int, cols: int) -> bool:
    g = param.grad
    return (_grad_sink and _grad_sink_gate and g is not None and g.dtype == torch.float32 and g.is_contiguous()
            and g.numel() == rows * cols and param.requires_grad)


def _ln_grad_targets(ln, g):
    """(dgamma buffer, dbeta buffer, sunk): with the gradient sink on, uc_layernorm_bwd adds (atomically, per block) STRAIGHT INTO the
    LayerNorm parameters' own gradient buffers — views of the trainer's flat buffer — and the Function returns None for them: no zero
    fill, no autograd add kernel per LayerNorm (240 of each per training step of the ViT-L model); otherwise fresh zeroed buffers."""
    w, b = getattr(ln, "weight", None), getattr(ln, "bias", None)
    if w is not None and b is not None and _sinkable(w, w.numel(), 1) and _sinkable(b, b.numel(), 1):
        return w.grad.view(-1), b.grad.view(-1), True
    return torch.zeros_like(g), torch.zeros_like(g), False


def _bias_target(bias_sink, N: int):
    """One contiguous fp32 [N] view covering the gradient buffers of the bias parameter(s) of a linear (two adjacent ones
    for the fused K|V projection), or None when they cannot take direct accumulation."""
    if not _grad_sink or not _grad_sink_gate or not bias_sink or any(p is None or not _sinkable(p, p.numel(), 1) for p in bias_sink):
        return None
    g0 = bias_sink[0].grad
    total, ptr = 0, g0.data_ptr()
    for p in bias_sink:
        if p.grad.data_ptr() != ptr + 4 * total:
            return None          # not adjacent in the flat gradient buffer
        total += p.numel()
    if total != N:
        return None
    return g0.reshape(-1) if len(bias_sink) == 1 else torch.as_strided(g0, (N,), (1,))


def _wgrad(dy2d: torch.Tensor, x2d: torch.Tensor, dt: torch.dtype, bias: bool = False, sink=None, bias_sink=None):
    """(dW [N,K] fp32, db [N] fp32 | None) = (dy^T x, column sums of dy) for row-major dy [M,N], x [M,K].
    sink: optional list of (parameter, row0, row1) covering the rows of dW; when every target has a gradient buffer the
    slabs are reduced into those buffers and dW is returned as None (see _grad_sink).
    bf16: the TN kernel contracts over the slow axis directly (uc_gemm_tn: split-K slabs + uc_splitk_reduce) and forms the
    bias gradient from the dy fragments it already holds; fp32 verification mode: explicit transposes + the exact fp32
    GEMM + uc_colsum."""
    if dt == torch.bfloat16:
        a = dy2d if dy2d.dtype == dt else ops.convert(_c(dy2d), dt)
        b = x2d if x2d.dtype == dt else ops.convert(_c(x2d), dt)
        if _tn_ok(a, b):
            sk = _split_k(a.shape[1], b.shape[1], a.shape[0])
            K = b.shape[1]
            bt = _bias_target(bias_sink, a.shape[1]) if bias else None
            if bt is not None:      # bias gradient added atomically into the bias's own gradient buffer: db is returned None
                ws, db = ops.gemm_tn(a, b, split_k=sk, colsum_into=bt), None
            elif bias:
                ws, cs = ops.gemm_tn(a, b, split_k=sk, colsum=True)
                db = _reduce_slabs(cs.unsqueeze(1)).reshape(-1)
            else:
                ws, db = ops.gemm_tn(a, b, split_k=sk), None
            if sink is not None and all(_sinkable(p, r1 - r0, K) for p, r0, r1 in sink):
                for p, r0, r1 in sink:
                    ops.splitk_reduce(ws[:, r0:r1], out=p.grad.view(r1 - r0, K), accumulate=True)
                return None, db
            return _reduce_slabs(ws), db
    aT, bT = _tp(_c(dy2d), dt), _tp(_c(x2d), dt)
    sk = _split_k_x3(aT.shape[0], bT.shape[0], aT.shape[1], dt)
    dW = ops.gemm(aT, bT, out_dtype=torch.float32, split_k=sk)
    return (dW if sk <= 1 else _reduce_slabs(dW)), (_colsum(_c(dy2d)) if bias else None)


def _wgrad_conv(dz: torch.Tensor, x: torch.Tensor, stride: int, relu_in: bool, bias: bool = False):
    """(dW_gemm [Cout, 9*Cin] with K ordered (ky,kx,c), db | None) of a 3x3/pad-1 conv: dz NHWC [B,Ho,Wo,Cout], x NHWC."""
    Cout = dz.shape[-1]
    dz2 = dz.view(-1, Cout)
    if x.dtype == torch.bfloat16 and _tn_ok(dz2, x):
        # (split so that tiles * sk fills the 256 CUs once; the tile count depends on which kernel the shape takes)
        sk = max(1, min(dz2.shape[0] // 512, 256 // ops.gemm_tn_conv_tiles(Cout, x.shape[1], x.shape[2], x.shape[3], stride)))
        if bias:
            ws, cs = ops.gemm_tn(dz2, x, split_k=sk, conv=(stride, relu_in), colsum=True)
            return _reduce_slabs(ws), _reduce_slabs(cs.unsqueeze(1)).reshape(-1)
        return _reduce_slabs(ops.gemm_tn(dz2, x, split_k=sk, conv=(stride, relu_in))), None
    aT, bT = _tp(dz2, x.dtype), ops.im2col_t(x, stride, relu_in, KPAD)
    sk = _split_k_x3(aT.shape[0], bT.shape[0], aT.shape[1], x.dtype)
    dW = ops.gemm(aT, bT, out_dtype=torch.float32, split_k=sk)
    return (dW if sk <= 1 else _reduce_slabs(dW)), (_colsum(dz2) if bias else None)


# bf16 twins of residual-stream gradients: uc_layernorm_bwd writes a bf16 copy of the dx it produces; the sub-layer that
# receives exactly that tensor as its d(x_out) (autograd hands it over untouched when x_out had a single consumer) takes
# the twin instead of running a conversion pass.  Keyed by storage address, validated by OBJECT IDENTITY through a weak
# reference (a recycled address can never alias a stale entry), a handful of entries at most.
_twins = {}


def _ln_bwd_residual(x2d, g, dh, eps, dg, db, dres, dt):
    if dres is not None and dres.dtype != x2d.dtype:
        dres = ops.convert(dres, x2d.dtype)
    if dt != torch.bfloat16 or x2d.dtype == torch.bfloat16:      # (a bf16 stream's dx IS the bf16 operand of the next backward GEMMs)
        return ops.layernorm_bwd(x2d, g, dh, eps, dg, db, dres=dres)
    dx, twin = ops.layernorm_bwd(x2d, g, dh, eps, dg, db, dres=dres, bf16_twin=True)
    if len(_twins) > 8:
        _twins.clear()
    _twins[dx.data_ptr()] = (weakref.ref(dx), dx._version, twin)
    return dx


def _as_dt(g: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    if g.dtype == dt:
        return g
    hit = _twins.pop(g.data_ptr(), None)
    if hit is not None and dt == torch.bfloat16 and hit[0]() is g and hit[1] == g._version:
        return hit[2]
    return ops.convert(g, dt)


def _colsum(src2d: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(src2d.shape[1], dtype=torch.float32, device=src2d.device)
    ops.colsum_(src2d, out)
    return out


def _w_t(owner, tag, weight2d_sources, build_2d, dt):
    """W^T [K,N] in dt of a [N,K] fp32 weight, cached per weight version."""
    return engine.prepared(owner, (tag, "T", dt), weight2d_sources, lambda: ops.transpose2d(build_2d(), dt))


def lin_weight_t(lin, dt):
    return _w_t(lin, "lin", (lin.weight,), lambda: lin.weight.detach().float().reshape(lin.weight.shape[0], -1).contiguous(), dt)


def kv_weight_t(projk, projv, dt):
    return _w_t(projk, "kv", (projk.weight, projv.weight),
                lambda: torch.cat([projk.weight.detach(), projv.weight.detach()], 0).float().contiguous(), dt)


def _check_rope(rope):
    if rope is not None and not engine.is_native_rope(rope):
        raise UcHipError("training through a foreign positional-encoding callable is not supported; use RoPE2D / cuRoPE2D")


def _rope_inverse_(t4, pos, rope):
    if rope is not None:
        ops.rope_2d_(t4, pos.contiguous(), rope.base, -rope.F0)


def _bwd_rope(rope, qpos, kpos, dt):
    "rope argument of ops.attention_bwd when the inverse rotation can ride in the bf16 backward kernels, else None."
    if rope is None or dt != torch.bfloat16 or qpos is None or kpos is None or qpos.numel() == 0:
        return None
    return (engine._pos2d(qpos), engine._pos2d(kpos), rope.base, rope.F0)


class Drops:
    """The dropout masks of ONE sub-layer call in training (uint8, 1 = keep; made by make_drops from PyTorch's generator, which owns the
    RNG state): `out` = nn.Dropout on the sub-layer's output (proj_drop / the Mlp's drop2), `path` = DropPath on the sub-layer's branch
    (one byte per sample), `mid` = the Mlp's drop1 on the hidden activation.  attn_drop (dropout of the attention probabilities inside
    the flash kernels) is not supported."""
    __slots__ = ("out", "out_scale", "path", "path_rows", "path_scale", "mid", "mid_scale")

    def __init__(self):
        self.out = self.path = self.mid = None
        self.out_scale = self.path_scale = self.mid_scale = 1.0
        self.path_rows = 0

    @property
    def has_out(self):
        return self.out is not None or self.path is not None


def make_drops(training: bool, device, B: int, N: int, C: int, p_out: float = 0.0, p_path: float = 0.0, hidden: int = 0,
               p_mid: float = 0.0, scale_by_keep: bool = True):
    """Drops for one sub-layer over [B * N, C] tokens, or None (eval mode / every rate 0).  Semantics: nn.Dropout (keep with
    probability 1 - p, scale 1 / (1 - p)); timm's DropPath (per sample, scale 1 / keep when scale_by_keep)."""
    if not training or (p_out <= 0.0 and p_path <= 0.0 and p_mid <= 0.0):
        return None
    d = Drops()
    if p_out > 0.0:
        d.out = torch.empty((B * N, C), dtype=torch.uint8, device=device).bernoulli_(1.0 - p_out)
        d.out_scale = 1.0 / (1.0 - p_out) if p_out < 1.0 else 0.0
    if p_path > 0.0:
        keep = 1.0 - p_path
        d.path = torch.empty((B,), dtype=torch.uint8, device=device).bernoulli_(keep)
        d.path_rows = N
        d.path_scale = 1.0 / keep if (scale_by_keep and keep > 0.0) else 1.0
    if p_mid > 0.0:
        d.mid = torch.empty((B * N, hidden), dtype=torch.uint8, device=device).bernoulli_(1.0 - p_mid)
        d.mid_scale = 1.0 / (1.0 - p_mid) if p_mid < 1.0 else 0.0
    return d


def _drops_saved(drops):
    """(mask tensors, spec) of a call's Drops: the masks go through ctx.save_for_backward — visible to checkpoint's saved-tensor hooks
    (dropped with the other activations of a checkpointed block and re-drawn from the restored generator state) and covered by
    autograd's version check; ctx.meta keeps only the scales."""
    if drops is None:
        return (), None
    masks = tuple(m for m in (drops.out, drops.path, drops.mid) if m is not None)
    return masks, (drops.out is not None, drops.out_scale, drops.path is not None, drops.path_rows, drops.path_scale,
                   drops.mid is not None, drops.mid_scale)


def _drops_restore(spec, saved):
    "(Drops or None, the saved tensors without the trailing masks): inverse of _drops_saved in a backward."
    if spec is None:
        return None, tuple(saved)
    n = int(spec[0]) + int(spec[2]) + int(spec[5])
    masks, saved = list(saved[len(saved) - n:]), tuple(saved[:len(saved) - n])
    d = Drops()
    if spec[0]:
        d.out, d.out_scale = masks.pop(0), spec[1]
    if spec[2]:
        d.path, d.path_rows, d.path_scale = masks.pop(0), spec[3], spec[4]
    if spec[5]:
        d.mid, d.mid_scale = masks.pop(0), spec[6]
    return d, saved


def _drop_out(f2d, drops, residual=None, out_dtype=None):
    "residual + dropout(f) (+ DropPath): the output masks of `drops`, the residual add riding in the last pass."
    steps = [(m, r, sc) for m, r, sc in ((drops.out, 0, drops.out_scale), (drops.path, drops.path_rows, drops.path_scale)) if m is not None]
    for i, (m, r, sc) in enumerate(steps):
        last = i == len(steps) - 1
        f2d = ops.mask_scale(f2d, m, r, sc, residual if last else None, out_dtype if last else None)
    return f2d


def _norm_or_none(norm):
    return None if norm is None or isinstance(norm, nn.Identity) else norm


def _qknorm_fwd(view4, norm):
    """qk_norm (utils/transformer_blocks.py:196-197, 229): LayerNorm over head_dim of a [B, N, H, Dh] view of q / k, BEFORE the positional
    encoding -> contiguous [B, N, H, Dh] in the same dtype."""
    if not isinstance(norm, nn.LayerNorm) or norm.weight is None or norm.bias is None:
        raise UcHipError(f"qk_norm with {type(norm).__name__} has no HIP path (nn.LayerNorm with affine parameters only)")
    pre = view4.contiguous()
    Dh = pre.shape[-1]
    return ops.layernorm(pre.view(-1, Dh), norm.weight.detach().float(), norm.bias.detach().float(), norm.eps, pre.dtype).view(pre.shape)


def _qknorm_bwd(view4, norm, dn):
    "(d view4 [contiguous], dgamma, dbeta) of _qknorm_fwd: uc_layernorm_bwd over B N H rows of head_dim (its 64-wide kernel)."
    pre = view4.contiguous()
    Dh = pre.shape[-1]
    g = norm.weight.detach().float()
    dg, db = torch.zeros_like(g), torch.zeros_like(g)
    dpre = ops.layernorm_bwd(pre.view(-1, Dh), g, dn.contiguous().view(-1, Dh), norm.eps, dg, db)
    return dpre.view(pre.shape), dg, db


def _attention_fwd(q, k, v, scale, lse, dropout=None):
    if q.dtype == torch.bfloat16:
        if q.shape[-1] != 64:
            raise UcHipError(f"bf16 attention needs head_dim 64 (got {q.shape[-1]})")
        return ops.attention(q, k, ops.vt_pack(v), scale, v_packed=True, lse=lse, dropout=dropout)
    return ops.attention(q, k, v, scale, lse=lse, dropout=dropout)


def attn_dropout(training: bool, p: float):
    """(p, seed) for a sub-layer's attention dropout, or None (eval mode / rate 0).  The seed comes from PyTorch's CPU generator —
    torch.manual_seed reproduces it, and a checkpointed block's re-computation draws the same one (preserve_rng_state) — and keys the
    counter-based mask the forward and backward kernels evaluate (uc_attention_fwd_drop)."""
    if not training or p <= 0.0:
        return None
    if p >= 1.0:
        raise UcHipError("attn_drop must be below 1")
    return (float(p), int(torch.randint(0, 2 ** 62, (1,)).item()))


# =================================================================================================================
# LayerNorm alone (encoder / decoder final norms, intermediate norms)
# =================================================================================================================
@_sink_aware
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, eps, out_dtype):
        x2d = _c(x2d)
        if x2d.dtype not in (torch.float32, torch.bfloat16):
            raise UcHipError("the training residual stream is fp32 or bf16")
        g, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        ctx.save_for_backward(x2d, g)
        ctx.eps = eps
        return ops.layernorm(x2d, g, b, eps, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        x2d, g = ctx.saved_tensors
        dy = _c(dy)
        dg, db = torch.zeros_like(g), torch.zeros_like(g)
        dx = ops.layernorm_bwd(x2d, g, dy, ctx.eps, dg, db)
        return dx, dg, db, None, None


def layer_norm(x2d, ln, out_dtype):
    return LayerNormFn.apply(x2d, ln.weight, ln.bias, ln.eps, out_dtype)


# =================================================================================================================
# Linear (proj_embed, the linear head's 1x1 conv, patch embedding GEMM)
# =================================================================================================================
@_sink_aware
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, owner, dt, out_dtype):
        x2d = _c(x2d)
        xb = x2d if x2d.dtype == dt else ops.convert(x2d, dt)
        w, b = engine.prepared(owner, ("lin2d", dt), (weight, bias),
                               lambda: (weight.detach().reshape(weight.shape[0], -1).to(dt).contiguous(),
                                        None if bias is None else bias.detach().float().contiguous()))
        ctx.save_for_backward(xb)
        ctx.owner, ctx.dt, ctx.x_dtype, ctx.wshape, ctx.has_bias = owner, dt, x2d.dtype, weight.shape, bias is not None
        ctx.weight = weight
        return ops.gemm(xb, w, b, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dy):
        (xb,) = ctx.saved_tensors
        dt = ctx.dt
        dy = _c(dy)
        need_dx = ctx.needs_input_grad[0]
        dyb = _as_dt(dy, dt)
        dW, db = _wgrad(dyb, xb, dt, ctx.has_bias, sink=[(ctx.weight, 0, ctx.wshape[0])])
        dW = None if dW is None else dW.view(ctx.wshape)
        dx = None
        if need_dx:
            wT = _w_t(ctx.owner, "lin2d", (ctx.weight,),
                      lambda: ctx.weight.detach().float().reshape(ctx.wshape[0], -1).contiguous(), dt)
            dx = ops.gemm(dyb, wT, out_dtype=ctx.x_dtype)
        return dx, dW, db, None, None, None


def linear(x2d, weight, bias, owner, dt, out_dtype):
    return LinearFn.apply(x2d, weight, bias, owner, dt, out_dtype)


@_sink_aware
class PatchEmbedFn(Function):
    """tokens = gather(img) . W^T + b (libs/croco/patch_embed.py:47,69-82).  The image's gradient (round 6; free under the reference's
    autograd) is d cols = d tok . W — one more GEMM — scattered back: patches do not overlap (stride = patch size), so the adjoint of the
    gather is a permutation of [B, gh, gw, c, u, v] into [B, c, gh u, gw v]."""

    @staticmethod
    def forward(ctx, img, weight, bias, owner, P, dt, out_dtype=torch.float32):
        cols = ops.patch_gather(img, P, dt)
        w, b = engine.patch_weights(owner, dt)
        ctx.save_for_backward(cols)
        ctx.dt, ctx.wshape, ctx.has_bias, ctx.weight, ctx.owner, ctx.P, ctx.ishape = dt, weight.shape, bias is not None, weight, owner, P, img.shape
        return ops.gemm(cols, w, b, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dtok):
        (cols,) = ctx.saved_tensors
        dtok = _as_dt(_c(dtok), ctx.dt)
        dimg = None
        if ctx.needs_input_grad[0]:
            owner, P, (B, Cn, H, W) = ctx.owner, ctx.P, ctx.ishape
            wt = _w_t(owner, "pe", (owner.weight,), lambda: owner.weight.detach().float().reshape(owner.weight.shape[0], -1).contiguous(), ctx.dt)
            dcols = ops.gemm(dtok, wt, out_dtype=torch.float32)                       # [B gh gw, (c, u, v)]
            dimg = dcols.view(B, H // P, W // P, Cn, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, Cn, H, W)
        dW, db = (None, None)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = _wgrad(dtok, cols, ctx.dt, ctx.has_bias, sink=[(ctx.weight, 0, ctx.wshape[0])])
            dW = None if dW is None else dW.view(ctx.wshape)
        return dimg, dW, db, None, None, None, None


def patch_embed(img, conv, P, dt, out_dtype=torch.float32):
    return PatchEmbedFn.apply(img, conv.weight, conv.bias, conv, P, dt, out_dtype)


# =================================================================================================================
# Attention alone, and the UNFUSED sub-layers built from it: training through a FOREIGN positional-encoding callable (round 6; free
# under the reference's autograd: custom_positional_encoding is any callable of (tokens [B, H, N, Dh], positions),
# utils/transformer_blocks.py:226-229, 352-356).  The callable runs as ordinary PyTorch code between HIP Functions — LayerNorm, the
# QKV / q / kv linears, attention forward + backward, the output projection — and PyTorch's autograd differentiates it.
# =================================================================================================================
class AttentionFn(Function):
    "o = softmax(scale q k^T) v for [B, N, H, Dh] views; backward = uc_attention_bwd from the saved log-sum-exp."

    @staticmethod
    def forward(ctx, q, k, v, scale, adrop=None):
        q, k, v = (t if t.stride(3) == 1 else t.contiguous() for t in (q, k, v))
        B, Nq, H, _ = q.shape
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        o = _attention_fwd(q, k, v, scale, lse, adrop)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale, ctx.adrop = scale, adrop
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = ops.attention_bwd(q, k, v, o, _c(do).to(q.dtype), lse, ctx.scale, dropout=ctx.adrop)
        return dq, dk, dv, None, None


def attention(q, k, v, scale, attn_drop=None):
    return AttentionFn.apply(q, k, v, scale, attn_drop)


def _foreign_rope(rope, t4, pos):
    "the reference's call: rope(tokens [B, H, N, Dh], positions) (utils/transformer_blocks.py:228-229) on a [B, N, H, Dh] view"
    return rope(t4.transpose(1, 2), pos).transpose(1, 2)


def self_attn_sublayer_unfused(x2d, ln, qkv, proj, B, N, H, rope, pos, scale, dt, drops=None, attn_drop=None):
    """x + proj(SDPA(rope(q), rope(k), v)) with a foreign `rope` callable: HIP Functions around PyTorch-differentiated rope calls."""
    M, C = x2d.shape
    Ca = qkv.out_features // 3
    h = layer_norm(x2d, ln, dt)
    t5 = linear(h, qkv.weight, qkv.bias, qkv, dt, dt).view(B, N, 3, H, Ca // H)
    q, k = _foreign_rope(rope, t5[:, :, 0], pos), _foreign_rope(rope, t5[:, :, 1], pos)
    o = attention(q.to(dt), k.to(dt), t5[:, :, 2], scale, attn_drop)
    f = linear(o.reshape(M, Ca), proj.weight, proj.bias, proj, dt, x2d.dtype)
    if drops is not None and drops.has_out:
        raise UcHipError("dropout of the sub-layer output next to a foreign positional-encoding callable has no HIP path")
    return x2d + f


def cross_attn_sublayer_unfused(x2d, y2d, ln, lny, ca, B, Nq, Nk, H, rope, qpos, kpos, scale, dt, drops=None, attn_drop=None):
    "the cross-attention sub-layer of CrossAttentionBlock with a foreign `rope` callable (see self_attn_sublayer_unfused)"
    Mq, C = x2d.shape
    Dh = C // H
    hq = layer_norm(x2d, ln, dt)
    hy = layer_norm(y2d, lny, dt) if lny is not None else convert(y2d, dt)
    q = linear(hq, ca.projq.weight, ca.projq.bias, ca.projq, dt, dt).view(B, Nq, H, Dh)
    k = linear(hy, ca.projk.weight, ca.projk.bias, ca.projk, dt, dt).view(B, Nk, H, Dh)
    v = linear(hy, ca.projv.weight, ca.projv.bias, ca.projv, dt, dt).view(B, Nk, H, Dh)
    q, k = _foreign_rope(rope, q, qpos), _foreign_rope(rope, k, kpos)
    o = attention(q.to(dt), k.to(dt), v, scale, attn_drop)
    f = linear(o.reshape(Mq, C), ca.proj.weight, ca.proj.bias, ca.proj, dt, x2d.dtype)
    if drops is not None and drops.has_out:
        raise UcHipError("dropout of the sub-layer output next to a foreign positional-encoding callable has no HIP path")
    return x2d + f


# LayerScale behind a sub-layer's output linear (DINOv2 blocks, SelfAttentionBlock(init_values=...)): the forward runs the linear with
# the folded weights W_f = gamma[:,None] W, b_f = gamma b (engine.layerscale_lin_weights — zero kernel work); the backward takes the
# gradients of the FOLDED parameters from the usual weight-gradient GEMM and unfolds them:
#   dW = gamma[:,None] dW_f,  db = gamma db_f,  dgamma = rowsum(dW_f * W) + db_f * b.
def _folded_weight_t(lin, gamma, dt):
    return _w_t(lin, "ls", (lin.weight, gamma),
                lambda: (lin.weight.detach().float() * gamma.detach().float()[:, None]).contiguous(), dt)


def _unfold_layerscale(lin, gamma, dWf, dbf):
    W = lin.weight.detach().float()
    g = gamma.detach().float()
    dgamma = (dWf * W).sum(1)
    db = None
    if dbf is not None:
        dgamma = dgamma + dbf * lin.bias.detach().float()
        db = dbf * g
    return dWf * g[:, None], db, dgamma.to(gamma.dtype)


# =================================================================================================================
# pre-LN sub-layers of the transformer blocks
# =================================================================================================================
@_sink_aware
class SelfAttnSubLayerFn(Function):
    """x + proj(SDPA(rope(q), rope(k), v)),  q,k,v = qkv(LN(x))   (blocks.py:105-125,154-158; transformer_blocks.py:214-260)."""

    @staticmethod
    def forward(ctx, x2d, ln_w, ln_b, w_qkv, b_qkv, w_proj, b_proj, ln, qkv, proj, B, N, H, rope, pos, scale, dt, gamma=None,
                qn_w=None, qn_b=None, kn_w=None, kn_b=None, qn=None, kn=None, drops=None, adrop=None):
        x2d = _c(x2d)
        M, C = x2d.shape
        Ca = qkv.out_features // 3          # width of q / k / v: C, or the layer's latent_attn_dim (utils/transformer_blocks.py:178-199)
        Dh = Ca // H
        _check_rope(rope)
        g, bta = engine.ln_params(ln)
        h = ops.layernorm(x2d, g, bta, ln.eps, dt)
        wq, bq = engine.lin_weights(qkv, dt)
        wp, bp = engine.lin_weights(proj, dt) if gamma is None else engine.layerscale_lin_weights(proj, gamma, dt)
        qkn = None
        if qn is not None or kn is not None:
            # qk_norm: q / k are normalised over head_dim BEFORE the positional encoding — the unfused route (t keeps the RAW q | k | v;
            # the normalised, rotated q / k the attention sees are saved next to it)
            t = ops.gemm(h, wq, bq)
            t5 = t.view(B, N, 3, H, Dh)
            qx = _qknorm_fwd(t5[:, :, 0], qn) if qn is not None else t5[:, :, 0].contiguous()
            kx = _qknorm_fwd(t5[:, :, 1], kn) if kn is not None else t5[:, :, 1].contiguous()
            if rope is not None:
                ops.rope_2d_(qx, pos.contiguous(), rope.base, rope.F0)
                ops.rope_2d_(kx, pos.contiguous(), rope.base, rope.F0)
            qkn = (qx, kx)
        elif dt == torch.bfloat16 and rope is not None:
            if Dh != 64:
                raise UcHipError(f"bf16 attention needs head_dim 64 (got {Dh})")
            t = ops.gemm(h, wq, bq, rope=engine._rope_epilogue(rope, engine._pos2d(pos), 2 * Ca))
            t5 = t.view(B, N, 3, H, Dh)
        else:
            t = ops.gemm(h, wq, bq)
            t5 = t.view(B, N, 3, H, Dh)
            if rope is not None:
                ops.rope_2d_(t5[:, :, 0], pos.contiguous(), rope.base, rope.F0)
                ops.rope_2d_(t5[:, :, 1], pos.contiguous(), rope.base, rope.F0)
        lse = torch.empty((B, H, N), dtype=torch.float32, device=x2d.device)
        o = _attention_fwd(*(qkn if qkn is not None else (t5[:, :, 0], t5[:, :, 1])), t5[:, :, 2], scale, lse, adrop)
        if drops is not None and drops.has_out:
            out = _drop_out(ops.gemm(o.view(M, Ca), wp, bp), drops, x2d, x2d.dtype)
        else:
            out = ops.gemm(o.view(M, Ca), wp, bp, residual=x2d, out_dtype=x2d.dtype)
        # (gamma goes through save_for_backward: autograd's version check then catches an in-place edit between forward and backward)
        masks, dspec = _drops_saved(drops)
        ctx.save_for_backward(x2d, g, h, t, o, lse, pos if pos is not None else torch.empty(0), *(() if qkn is None else qkn),
                              *(() if gamma is None else (gamma,)), *masks)
        ctx.meta = (ln, qkv, proj, B, N, H, rope, scale, dt, b_qkv is not None, b_proj is not None, qn, kn, qkn is not None, dspec)
        ctx.adrop = adrop
        return out

    @staticmethod
    def backward(ctx, dxo):
        ln, qkv, proj, B, N, H, rope, scale, dt, has_bq, has_bp, qn, kn, has_qkn, dspec = ctx.meta
        drops, saved = _drops_restore(dspec, ctx.saved_tensors)
        x2d, g, h, t, o, lse, pos, *rest = saved
        qx, kx = (rest[0], rest[1]) if has_qkn else (None, None)
        rest = rest[2:] if has_qkn else rest
        gamma = rest[0] if rest else None
        M, C = x2d.shape
        Ca = qkv.out_features // 3          # width of q / k / v: C, or the layer's latent_attn_dim (utils/transformer_blocks.py:178-199)
        Dh = Ca // H
        dxo = _c(dxo)
        dyb = _as_dt(dxo, dt)
        if drops is not None and drops.has_out:
            dyb = _drop_out(dyb, drops)
        dgamma = None
        if gamma is None:
            dWp, dbp = _wgrad(dyb, o.view(M, Ca), dt, has_bp, sink=[(proj.weight, 0, C)], bias_sink=[proj.bias])
            do = ops.gemm(dyb, lin_weight_t(proj, dt))
        else:
            dWp, dbp = _wgrad(dyb, o.view(M, Ca), dt, has_bp)
            dWp, dbp, dgamma = _unfold_layerscale(proj, gamma, dWp, dbp)
            do = ops.gemm(dyb, _folded_weight_t(proj, gamma, dt))
        dt3 = torch.empty_like(t)
        d5, t5 = dt3.view(B, N, 3, H, Dh), t.view(B, N, 3, H, Dh)
        fused_rope = _bwd_rope(rope, pos, pos, dt)      # (bf16: the inverse rotation of dq / dk rides in the backward kernels)
        dqn_w = dqn_b = dkn_w = dkn_b = None
        if has_qkn:
            dqx, dkx = torch.empty_like(qx), torch.empty_like(kx)
            ops.attention_bwd(qx, kx, t5[:, :, 2], o, do.view(B, N, H, Dh), lse, scale, out=(dqx, dkx, d5[:, :, 2]), rope=fused_rope,
                              dropout=ctx.adrop)
            if fused_rope is None:
                _rope_inverse_(dqx, pos, rope)
                _rope_inverse_(dkx, pos, rope)
            if qn is not None:
                dqx, dqn_w, dqn_b = _qknorm_bwd(t5[:, :, 0], qn, dqx)
            if kn is not None:
                dkx, dkn_w, dkn_b = _qknorm_bwd(t5[:, :, 1], kn, dkx)
            d5[:, :, 0].copy_(dqx)
            d5[:, :, 1].copy_(dkx)
        else:
            ops.attention_bwd(t5[:, :, 0], t5[:, :, 1], t5[:, :, 2], o, do.view(B, N, H, Dh), lse, scale,
                              out=(d5[:, :, 0], d5[:, :, 1], d5[:, :, 2]), rope=fused_rope, dropout=ctx.adrop)
            if fused_rope is None:
                _rope_inverse_(d5[:, :, 0], pos, rope)
                _rope_inverse_(d5[:, :, 1], pos, rope)
        dWq, dbq = _wgrad(dt3, h, dt, has_bq, sink=[(qkv.weight, 0, 3 * Ca)], bias_sink=[qkv.bias])
        dh = ops.gemm(dt3, lin_weight_t(qkv, dt))
        dg, db, sunk = _ln_grad_targets(ln, g)
        dx = _ln_bwd_residual(x2d, g, dh, ln.eps, dg, db, dxo, dt)
        if sunk:
            dg = db = None
        return (dx, dg, db, dWq, dbq, dWp, dbp) + (None,) * 10 + (dgamma, dqn_w, dqn_b, dkn_w, dkn_b, None, None, None, None)


def self_attn_sublayer(x2d, ln, qkv, proj, B, N, H, rope, pos, scale, dt, gamma=None, q_norm=None, k_norm=None, drops=None, attn_drop=None):
    """gamma: LayerScale on the sub-layer's output (x + gamma * proj(...)), or None.  q_norm / k_norm: the layer's qk_norm modules
    (LayerNorm over head_dim before the positional encoding; nn.Identity / None: off).  attn_drop: (p, seed) from attn_dropout, or None."""
    qn, kn = _norm_or_none(q_norm), _norm_or_none(k_norm)
    if rope is not None and not engine.is_native_rope(rope):      # a foreign positional-encoding callable: the unfused route
        if gamma is not None or qn is not None or kn is not None:
            raise UcHipError("LayerScale / qk_norm next to a foreign positional-encoding callable have no HIP training path")
        return self_attn_sublayer_unfused(x2d, ln, qkv, proj, B, N, H, rope, pos, scale, dt, drops, attn_drop)
    return SelfAttnSubLayerFn.apply(x2d, ln.weight, ln.bias, qkv.weight, qkv.bias, proj.weight, proj.bias, ln, qkv, proj,
                                    B, N, H, rope, pos, scale, dt, gamma, getattr(qn, "weight", None), getattr(qn, "bias", None),
                                    getattr(kn, "weight", None), getattr(kn, "bias", None), qn, kn, drops, attn_drop)


@_sink_aware
class CrossAttnSubLayerFn(Function):
    """x + proj(SDPA(rope(projq(LN2(x)), xpos), rope(projk(LNy(y)), ypos), projv(LNy(y))))  (transformer_blocks.py:329-412,604-647)."""

    @staticmethod
    def forward(ctx, x2d, y2d, ln_w, ln_b, lny_w, lny_b, wq_, bq_, wk_, bk_, wv_, bv_, wp_, bp_, ln, lny, projq, projk, projv, proj,
                B, Nq, Nk, H, rope, qpos, kpos, scale, dt, qn_w=None, qn_b=None, kn_w=None, kn_b=None, qn=None, kn=None, drops=None,
                gamma=None, adrop=None):
        x2d, y2d = _c(x2d), _c(y2d)
        Mq, C = x2d.shape
        Dh = C // H
        _check_rope(rope)
        g, bta = engine.ln_params(ln)
        hq = ops.layernorm(x2d, g, bta, ln.eps, dt)
        if lny is not None:
            gy, by = engine.ln_params(lny)
            hy = ops.layernorm(y2d, gy, by, lny.eps, dt)
        else:
            gy = torch.empty(0, device=x2d.device)
            hy = y2d if y2d.dtype == dt else ops.convert(y2d, dt)
        wq, bq = engine.lin_weights(projq, dt)
        wkv, bkv = engine.kv_weights(projk, projv, dt)
        wp, bp = engine.lin_weights(proj, dt) if gamma is None else engine.layerscale_lin_weights(proj, gamma, dt)
        qkn = None
        if qn is not None or kn is not None:     # qk_norm: the unfused route (see SelfAttnSubLayerFn)
            q = ops.gemm(hq, wq, bq)
            kv = ops.gemm(hy, wkv, bkv)
            q4, k4 = q.view(B, Nq, H, Dh), kv.view(B, Nk, 2, H, Dh)[:, :, 0]
            qx = _qknorm_fwd(q4, qn) if qn is not None else q4.contiguous()
            kx = _qknorm_fwd(k4, kn) if kn is not None else k4.contiguous()
            if rope is not None:
                ops.rope_2d_(qx, qpos.contiguous(), rope.base, rope.F0)
                ops.rope_2d_(kx, kpos.contiguous(), rope.base, rope.F0)
            qkn = (qx, kx)
        elif dt == torch.bfloat16 and rope is not None:
            if Dh != 64:
                raise UcHipError(f"bf16 attention needs head_dim 64 (got {Dh})")
            q = ops.gemm(hq, wq, bq, rope=engine._rope_epilogue(rope, engine._pos2d(qpos), C))
            kv = ops.gemm(hy, wkv, bkv, rope=engine._rope_epilogue(rope, engine._pos2d(kpos), C))
        else:
            q = ops.gemm(hq, wq, bq)
            kv = ops.gemm(hy, wkv, bkv)
            if rope is not None:
                ops.rope_2d_(q.view(B, Nq, H, Dh), qpos.contiguous(), rope.base, rope.F0)
                ops.rope_2d_(kv.view(B, Nk, 2, H, Dh)[:, :, 0], kpos.contiguous(), rope.base, rope.F0)
        kv5 = kv.view(B, Nk, 2, H, Dh)
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=x2d.device)
        o = _attention_fwd(*(qkn if qkn is not None else (q.view(B, Nq, H, Dh), kv5[:, :, 0])), kv5[:, :, 1], scale, lse, adrop)
        ctx.adrop = adrop
        if drops is not None and drops.has_out:
            out = _drop_out(ops.gemm(o.view(Mq, C), wp, bp), drops, x2d, x2d.dtype)
        else:
            out = ops.gemm(o.view(Mq, C), wp, bp, residual=x2d, out_dtype=x2d.dtype)
        e = torch.empty(0)
        masks, dspec = _drops_saved(drops)
        ctx.save_for_backward(x2d, y2d, g, gy, hq, hy, q, kv, o, lse, qpos if qpos is not None else e, kpos if kpos is not None else e,
                              *(() if qkn is None else qkn), *(() if gamma is None else (gamma,)), *masks)
        ctx.meta = (ln, lny, projq, projk, projv, proj, B, Nq, Nk, H, rope, scale, dt,
                    bq_ is not None, bk_ is not None, bv_ is not None, bp_ is not None, qn, kn, dspec, qkn is not None)
        return out

    @staticmethod
    def backward(ctx, dxo):
        ln, lny, projq, projk, projv, proj, B, Nq, Nk, H, rope, scale, dt, has_bq, has_bk, has_bv, has_bp, qn, kn, dspec, has_qkn = ctx.meta
        drops, saved = _drops_restore(dspec, ctx.saved_tensors)
        x2d, y2d, g, gy, hq, hy, q, kv, o, lse, qpos, kpos, *rest = saved
        qkn = rest[:2] if has_qkn else []
        rest = rest[2:] if has_qkn else rest
        gamma = rest[0] if rest else None
        Mq, C = x2d.shape
        Dh = C // H
        dxo = _c(dxo)
        dyb = _as_dt(dxo, dt)
        if drops is not None and drops.has_out:
            dyb = _drop_out(dyb, drops)
        dgamma = None
        if gamma is None:
            dWp, dbp = _wgrad(dyb, o.view(Mq, C), dt, has_bp, sink=[(proj.weight, 0, C)], bias_sink=[proj.bias])
            do = ops.gemm(dyb, lin_weight_t(proj, dt))
        else:       # LayerScale folded into proj: gradient of the folded weight, unfolded into d W, d b, d gamma
            dWp, dbp = _wgrad(dyb, o.view(Mq, C), dt, has_bp)
            dWp, dbp, dgamma = _unfold_layerscale(proj, gamma, dWp, dbp)
            do = ops.gemm(dyb, _folded_weight_t(proj, gamma, dt))
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        kv5, dkv5 = kv.view(B, Nk, 2, H, Dh), dkv.view(B, Nk, 2, H, Dh)
        fused_rope = _bwd_rope(rope, qpos, kpos, dt)
        dqn_w = dqn_b = dkn_w = dkn_b = None
        if qkn:
            qx, kx = qkn
            dqx, dkx = torch.empty_like(qx), torch.empty_like(kx)
            ops.attention_bwd(qx, kx, kv5[:, :, 1], o, do.view(B, Nq, H, Dh), lse, scale, out=(dqx, dkx, dkv5[:, :, 1]), rope=fused_rope,
                              dropout=ctx.adrop)
            if fused_rope is None:
                _rope_inverse_(dqx, qpos, rope)
                _rope_inverse_(dkx, kpos, rope)
            if qn is not None:
                dqx, dqn_w, dqn_b = _qknorm_bwd(q.view(B, Nq, H, Dh), qn, dqx)
            if kn is not None:
                dkx, dkn_w, dkn_b = _qknorm_bwd(kv5[:, :, 0], kn, dkx)
            dq.view(B, Nq, H, Dh).copy_(dqx)
            dkv5[:, :, 0].copy_(dkx)
        else:
            ops.attention_bwd(q.view(B, Nq, H, Dh), kv5[:, :, 0], kv5[:, :, 1], o, do.view(B, Nq, H, Dh), lse, scale,
                              out=(dq.view(B, Nq, H, Dh), dkv5[:, :, 0], dkv5[:, :, 1]), rope=fused_rope, dropout=ctx.adrop)
            if fused_rope is None:
                _rope_inverse_(dq.view(B, Nq, H, Dh), qpos, rope)
                _rope_inverse_(dkv5[:, :, 0], kpos, rope)
        # query side
        dWq, dbq = _wgrad(dq, hq, dt, has_bq, sink=[(projq.weight, 0, C)], bias_sink=[projq.bias])
        dhq = ops.gemm(dq, lin_weight_t(projq, dt))
        dg, db, sunk = _ln_grad_targets(ln, g)
        dx = _ln_bwd_residual(x2d, g, dhq, ln.eps, dg, db, dxo, dt)
        if sunk:
            dg = db = None
        # key/value side (the other view's tokens)
        dWkv, dbkv = _wgrad(dkv, hy, dt, has_bk or has_bv, sink=[(projk.weight, 0, C), (projv.weight, C, 2 * C)],
                            bias_sink=[projk.bias, projv.bias] if (has_bk and has_bv) else None)
        dhy = ops.gemm(dkv, kv_weight_t(projk, projv, dt), out_dtype=dt if lny is not None else y2d.dtype)
        if lny is not None:
            dgy, dby, sunk_y = _ln_grad_targets(lny, gy)
            dy = ops.layernorm_bwd(y2d, gy, dhy, lny.eps, dgy, dby)
            if sunk_y:
                dgy = dby = None
        else:
            dgy = dby = None
            dy = dhy
        dWk, dWv = (None, None) if dWkv is None else (dWkv[:C], dWkv[C:])
        dbk = dbkv[:C] if (has_bk and dbkv is not None) else None
        dbv = dbkv[C:] if (has_bv and dbkv is not None) else None
        return (dx, dy, dg, db, dgy, dby, dWq, dbq, dWk, dbk, dWv, dbv, dWp, dbp) + (None,) * 15 + (dqn_w, dqn_b, dkn_w, dkn_b, None, None, None,
                                                                                                      dgamma, None)


def cross_attn_sublayer(x2d, y2d, ln, lny, ca, B, Nq, Nk, H, rope, qpos, kpos, scale, dt, drops=None, gamma=None, attn_drop=None):
    lw, lb = (lny.weight, lny.bias) if lny is not None else (None, None)
    qn, kn = _norm_or_none(getattr(ca, "q_norm", None)), _norm_or_none(getattr(ca, "k_norm", None))
    if rope is not None and not engine.is_native_rope(rope):      # a foreign positional-encoding callable: the unfused route
        if gamma is not None or qn is not None or kn is not None:
            raise UcHipError("LayerScale / qk_norm next to a foreign positional-encoding callable have no HIP training path")
        return cross_attn_sublayer_unfused(x2d, y2d, ln, lny, ca, B, Nq, Nk, H, rope, qpos, kpos, scale, dt, drops, attn_drop)
    return CrossAttnSubLayerFn.apply(x2d, y2d, ln.weight, ln.bias, lw, lb, ca.projq.weight, ca.projq.bias, ca.projk.weight,
                                     ca.projk.bias, ca.projv.weight, ca.projv.bias, ca.proj.weight, ca.proj.bias, ln, lny,
                                     ca.projq, ca.projk, ca.projv, ca.proj, B, Nq, Nk, H, rope, qpos, kpos, scale, dt,
                                     getattr(qn, "weight", None), getattr(qn, "bias", None), getattr(kn, "weight", None),
                                     getattr(kn, "bias", None), qn, kn, drops, gamma, attn_drop)


@_sink_aware
class MlpSubLayerFn(Function):
    """x + fc2(act(fc1(LN(x))))   (blocks.py:64-86,159; transformer_blocks.py:517-560)."""

    @staticmethod
    def forward(ctx, x2d, ln_w, ln_b, w1_, b1_, w2_, b2_, ln, fc1, fc2, act, dt, gamma=None, drops=None):
        x2d = _c(x2d)
        g, bta = engine.ln_params(ln)
        h = ops.layernorm(x2d, g, bta, ln.eps, dt)
        w1, b1 = engine.lin_weights(fc1, dt)
        w2, b2 = engine.lin_weights(fc2, dt) if gamma is None else engine.layerscale_lin_weights(fc2, gamma, dt)
        u = torch.empty((x2d.shape[0], w1.shape[0]), dtype=dt, device=x2d.device)
        a = ops.gemm(h, w1, b1, act=act, preact_out=u)
        if drops is not None and drops.mid is not None:      # drop1: the hidden activation fc2 (and its weight gradient) sees
            a = ops.mask_scale(a, drops.mid, 0, drops.mid_scale)
        if drops is not None and drops.has_out:
            out = _drop_out(ops.gemm(a, w2, b2), drops, x2d, x2d.dtype)
        else:
            out = ops.gemm(a, w2, b2, residual=x2d, out_dtype=x2d.dtype)
        masks, dspec = _drops_saved(drops)
        ctx.save_for_backward(x2d, g, h, u, a, *(() if gamma is None else (gamma,)), *masks)
        ctx.meta = (ln, fc1, fc2, act, dt, b1_ is not None, b2_ is not None, dspec)
        return out

    @staticmethod
    def backward(ctx, dxo):
        ln, fc1, fc2, act, dt, has_b1, has_b2, dspec = ctx.meta
        drops, saved = _drops_restore(dspec, ctx.saved_tensors)
        x2d, g, h, u, a, *rest = saved
        gamma = rest[0] if rest else None
        dxo = _c(dxo)
        dyb = _as_dt(dxo, dt)
        if drops is not None and drops.has_out:
            dyb = _drop_out(dyb, drops)
        dgamma = None
        if gamma is None:
            dW2, db2 = _wgrad(dyb, a, dt, has_b2, sink=[(fc2.weight, 0, fc2.weight.shape[0])], bias_sink=[fc2.bias])
            w2t = lin_weight_t(fc2, dt)
        else:
            dW2, db2 = _wgrad(dyb, a, dt, has_b2)
            dW2, db2, dgamma = _unfold_layerscale(fc2, gamma, dW2, db2)
            w2t = _folded_weight_t(fc2, gamma, dt)
        if act != "none" and dt == torch.bfloat16 and w2t.shape[1] % 64 == 0:
            du = ops.gemm(dyb, w2t, dact=(u, act))          # act'(u) applied in the data-gradient GEMM's epilogue
        else:
            da = ops.gemm(dyb, w2t)
            du = ops.act_bwd(da, u, act) if act != "none" else da
        if drops is not None and drops.mid is not None:      # (elementwise factors commute: mask * act'(u) * da)
            du = ops.mask_scale(du, drops.mid, 0, drops.mid_scale)
        dW1, db1 = _wgrad(du, h, dt, has_b1, sink=[(fc1.weight, 0, fc1.weight.shape[0])], bias_sink=[fc1.bias])
        dh = ops.gemm(du, lin_weight_t(fc1, dt))
        dg, db, sunk = _ln_grad_targets(ln, g)
        dx = _ln_bwd_residual(x2d, g, dh, ln.eps, dg, db, dxo, dt)
        if sunk:
            dg = db = None
        return (dx, dg, db, dW1, db1, dW2, db2) + (None,) * 5 + (dgamma, None)


def mlp_sublayer(x2d, ln, fc1, fc2, act, dt, gamma=None, drops=None):
    "gamma: LayerScale on the sub-layer's output (x + gamma * fc2(...)), or None.  drops: the call's dropout masks (make_drops) or None."
    return MlpSubLayerFn.apply(x2d, ln.weight, ln.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, ln, fc1, fc2, act, dt, gamma, drops)


@_sink_aware
class SwiGLUSubLayerFn(Function):
    """x + w3(silu(x1) * x2),  x1, x2 = w12(LN(x)).chunk(2)   (DINOv2 giant: the hub's SwiGLUFFNFused behind encoders/dinov2.py:68-84)."""

    @staticmethod
    def forward(ctx, x2d, ln_w, ln_b, w12_, b12_, w3_, b3_, ln, w12, w3, dt, gamma=None):
        x2d = _c(x2d)
        g, bta = engine.ln_params(ln)
        h = ops.layernorm(x2d, g, bta, ln.eps, dt)
        w1, b1 = engine.lin_weights(w12, dt)
        w2, b2 = engine.lin_weights(w3, dt) if gamma is None else engine.layerscale_lin_weights(w3, gamma, dt)
        t = ops.gemm(h, w1, b1)
        a = ops.swiglu(t)
        out = ops.gemm(a, w2, b2, residual=x2d, out_dtype=x2d.dtype)
        ctx.save_for_backward(x2d, g, h, t, a, *(() if gamma is None else (gamma,)))
        ctx.meta = (ln, w12, w3, dt, b12_ is not None, b3_ is not None)
        return out

    @staticmethod
    def backward(ctx, dxo):
        x2d, g, h, t, a, *rest = ctx.saved_tensors
        gamma = rest[0] if rest else None
        ln, w12, w3, dt, has_b1, has_b2 = ctx.meta
        dxo = _c(dxo)
        dyb = _as_dt(dxo, dt)
        dgamma = None
        if gamma is None:
            dW2, db2 = _wgrad(dyb, a, dt, has_b2, sink=[(w3.weight, 0, w3.weight.shape[0])], bias_sink=[w3.bias])
            w2t = lin_weight_t(w3, dt)
        else:
            dW2, db2 = _wgrad(dyb, a, dt, has_b2)
            dW2, db2, dgamma = _unfold_layerscale(w3, gamma, dW2, db2)
            w2t = _folded_weight_t(w3, gamma, dt)
        dtt = ops.swiglu_bwd(ops.gemm(dyb, w2t), t)
        dW1, db1 = _wgrad(dtt, h, dt, has_b1, sink=[(w12.weight, 0, w12.weight.shape[0])], bias_sink=[w12.bias])
        dh = ops.gemm(dtt, lin_weight_t(w12, dt))
        dg, db, sunk = _ln_grad_targets(ln, g)
        dx = _ln_bwd_residual(x2d, g, dh, ln.eps, dg, db, dxo, dt)
        if sunk:
            dg = db = None
        return (dx, dg, db, dW1, db1, dW2, db2) + (None,) * 4 + (dgamma,)


def swiglu_sublayer(x2d, ln, w12, w3, dt, gamma=None):
    "gamma: LayerScale on the sub-layer's output (x + gamma * w3(...)), or None."
    return SwiGLUSubLayerFn.apply(x2d, ln.weight, ln.bias, w12.weight, w12.bias, w3.weight, w3.bias, ln, w12, w3, dt, gamma)


# =================================================================================================================
# heads: pixel shuffle, adaptor, loss
# =================================================================================================================
@_sink_aware
class PixelShuffleFn(Function):
    @staticmethod
    def forward(ctx, rows, B, h, w, P, Cout):
        ctx.P = P
        return ops.pixel_shuffle(_c(rows), B, h, w, P, Cout)

    @staticmethod
    def backward(ctx, dimg):
        return ops.pixel_unshuffle(_c(dimg), ctx.P, torch.float32), None, None, None, None, None


def pixel_shuffle(rows, B, h, w, P, Cout):
    return PixelShuffleFn.apply(rows, B, h, w, P, Cout)


@_sink_aware
class PointmapAdaptorFn(Function):
    @staticmethod
    def forward(ctx, x, vmin, vmax):
        ctx.save_for_backward(x)
        ctx.v = (vmin, vmax)
        pts, conf = ops.pointmap_adaptor(x, vmin, vmax)
        return pts, conf

    @staticmethod
    def backward(ctx, dpts, dconf):
        (x,) = ctx.saved_tensors
        return ops.pointmap_adaptor_bwd(x, _c(dpts), _c(dconf), *ctx.v), None, None


def pointmap_adaptor(x, vmin, vmax):
    return PointmapAdaptorFn.apply(x, vmin, vmax)


@_sink_aware
class AdaptorProgramFn(Function):
    """A channel program of the generic adaptor pass (ops.adaptor_program) and its gradient with respect to the decoded channels
    (what torch autograd computes through the reference's adaptor compositions, prediction_heads/adaptors.py:25-2300)."""

    @staticmethod
    def forward(ctx, x, segs, cout):
        ctx.save_for_backward(x)
        ctx.segs = segs
        return ops.adaptor_program(x, segs, cout)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return ops.adaptor_program_bwd(x, _c(dout), ctx.segs), None, None


def adaptor_program(x, segs, cout):
    return AdaptorProgramFn.apply(x, segs, cout)


@_sink_aware
class ConfLossFn(Function):
    """mean_pix(conf * |pts - gt|) - alpha * mean_pix(log conf): the DUSt3R confidence-weighted regression objective
    (the reference ships no loss; SURVEY §8d names this one).  One kernel computes the loss and both gradients."""

    @staticmethod
    def forward(ctx, pts, conf, gt, alpha):
        pts, conf, gt = _c(pts), _c(conf), _c(gt)
        npix = conf.numel()
        acc = torch.zeros(1, dtype=torch.float32, device=pts.device)
        dpts, dconf = ops.conf_loss(pts, conf, gt, alpha, 1.0 / npix, acc)
        ctx.save_for_backward(dpts, dconf)
        return acc[0] / npix

    @staticmethod
    def backward(ctx, g):
        dpts, dconf = ctx.saved_tensors
        return dpts * g, dconf * g, None, None


def conf_loss(pts, conf, gt, alpha: float = 0.2):
    return ConfLossFn.apply(pts, conf, gt.float(), alpha)


# =================================================================================================================
# DPT head (NHWC maps in the head dtype): 3x3 implicit-GEMM convs, 1x1 convs (LinearFn on the pixel matrix),
# ConvTranspose2d(k=s), bilinear resize, the 4-channel regressor tail, and the dtype hop at the head's entry
# =================================================================================================================
@_sink_aware
class ConvertFn(Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.src = x.dtype
        x = _c(x)
        return x if x.dtype == dt else ops.convert(x, dt)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        return (dy if dy.dtype == ctx.src else ops.convert(dy, ctx.src)), None


def convert(x, dt):
    return ConvertFn.apply(x, dt)


def _conv3x3_rot_weight(conv, dt):
    """GEMM weight of the data gradient: [Cin, 9*Cout] with K ordered (ky', kx', o) = W[o, c, 2-ky', 2-kx']."""
    return engine.prepared(conv, ("c3rot", dt), (conv.weight,),
                           lambda: conv.weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(conv.in_channels, -1).to(dt).contiguous())


@_sink_aware
class Conv3x3Fn(Function):
    """y = [residual +] act(conv3x3(relu?(x)) + b) on NHWC maps (dpt_block.py:17-289, dpt.py:116-178,271-277)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, residual2, conv, relu_in, act, grad_mask_cell=None):
        ctx.grad_mask_cell = grad_mask_cell
        x = _c(x)
        B, H, W, Cin = x.shape
        s = conv.stride[0]
        w, b = engine.conv3x3_weights(conv, x.dtype)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        r1 = None if residual is None else _c(residual).reshape(-1, residual.shape[-1])
        r2 = None if residual2 is None else _c(residual2).reshape(-1, residual2.shape[-1])
        y = ops.gemm(x, w, b, act=act, residual=r1, residual2=r2, relu_a=relu_in, conv=(B, H, W, Cin, s)).view(B, Ho, Wo, -1)
        if act not in (None, "none", "relu"):
            raise UcHipError(f"conv3x3 backward: unsupported fused activation {act}")
        if act == "relu" and residual is not None:
            raise UcHipError("conv3x3 backward: fused ReLU together with a residual is not used by the DPT head")
        ctx.save_for_backward(x, y if act == "relu" else torch.empty(0))
        ctx.meta = (conv, relu_in, act, bias is not None, residual is not None, residual2 is not None, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        conv, relu_in, act, has_b, has_r1, has_r2, s = ctx.meta
        B, H, W, Cin = x.shape
        dt = x.dtype
        dy = _c(dy)
        Cout = dy.shape[-1]
        cell = ctx.grad_mask_cell
        already_masked = cell is not None and cell[0]          # the only consumer's backward applied this ReLU's mask (Conv1x1To4Fn)
        if cell is not None:
            cell[0] = False
        dz = ops.act_bwd(dy, y, "relu") if (act == "relu" and not already_masked) else dy
        dz2 = dz.view(-1, Cout)
        dWg, db = _wgrad_conv(dz, x, s, relu_in, has_b)                          # [Cout, 9*Cin], K ordered (ky,kx,c)
        dW = dWg.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        dx = None
        if ctx.needs_input_grad[0]:
            g = dz if s == 1 else ops.dilate_nhwc(dz, H, W, s)
            if relu_in and dt == torch.bfloat16 and Cout % 64 == 0:
                dx = ops.gemm(g, _conv3x3_rot_weight(conv, dt), conv=(B, H, W, Cout, 1), dact=(x.view(-1, Cin), "relu")).view(B, H, W, Cin)
            else:
                dx = ops.gemm(g, _conv3x3_rot_weight(conv, dt), conv=(B, H, W, Cout, 1)).view(B, H, W, Cin)
                if relu_in:
                    dx = ops.act_bwd(dx, x, "relu")
        return dx, dW, db, (dy if has_r1 else None), (dy if has_r2 else None), None, None, None, None


def conv3x3(x, conv, relu_in=False, act=None, residual=None, residual2=None, grad_mask_cell=None):
    """grad_mask_cell: a one-element list shared with the SOLE consumer of a fused-ReLU output (conv1x1_to4(relu_cell=...)); when that
    consumer's backward has applied the ReLU mask it sets cell[0] and this backward skips its own mask pass."""
    return Conv3x3Fn.apply(x, conv.weight, conv.bias, residual, residual2, conv, relu_in, act, grad_mask_cell)


def conv1x1(x, conv):
    B, H, W, Cin = x.shape
    y = LinearFn.apply(x.reshape(-1, Cin), conv.weight, conv.bias, conv, x.dtype, x.dtype)
    return y.view(B, H, W, -1)


def padded_conv1x1_weights(conv, dt, cpad):
    "Conv2d(k=1) weights with the output channels padded by zero rows to `cpad` (the 8-channel granule of the NHWC kernels)."
    n = conv.out_channels

    def build():
        w = torch.zeros(cpad, conv.in_channels, device=conv.weight.device)
        w[:n] = conv.weight.detach().reshape(n, -1).float()
        b = torch.zeros(cpad, device=conv.weight.device)
        if conv.bias is not None:
            b[:n] = conv.bias.detach().float()
        return w.to(dt).contiguous(), b
    return engine.prepared(conv, ("c1pad", dt), (conv.weight, conv.bias), build)


@_sink_aware
class PaddedConv1x1Fn(Function):
    """A 1x1 convolution to a channel count that is not a multiple of 8 (DPTSegmentationProcessor's class logits, dpt.py:314-381):
    computed on `cpad` output columns (zero weight rows), fp32 output [M, cpad]; gradients of the real rows only."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, conv, dt, cpad):
        x2d = _c(x2d)
        xb = x2d if x2d.dtype == dt else ops.convert(x2d, dt)
        w, b = padded_conv1x1_weights(conv, dt, cpad)
        ctx.save_for_backward(xb)
        ctx.meta = (conv, dt, x2d.dtype, weight.shape, bias is not None, cpad)
        return ops.gemm(xb, w, b, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, dy):
        (xb,) = ctx.saved_tensors
        conv, dt, x_dtype, wshape, has_b, cpad = ctx.meta
        n = wshape[0]
        dyb = _as_dt(_c(dy), dt)
        dW, db = _wgrad(dyb, xb, dt, has_b)
        dx = None
        if ctx.needs_input_grad[0]:
            def padded32():
                w = torch.zeros(cpad, conv.in_channels, device=conv.weight.device)
                w[:n] = conv.weight.detach().reshape(n, -1).float()
                return w
            dx = ops.gemm(dyb, _w_t(conv, "c1pad", (conv.weight,), padded32, dt), out_dtype=x_dtype)
        return dx, dW[:n].reshape(wshape), (db[:n] if has_b else None), None, None, None


def padded_conv1x1(x, conv, dt, cpad):
    B, H, W, Cin = x.shape
    return PaddedConv1x1Fn.apply(x.reshape(-1, Cin), conv.weight, conv.bias, conv, dt, cpad).view(B, H, W, cpad)


@_sink_aware
class ConvTransposeFn(Function):
    """ConvTranspose2d(kernel = stride, no padding) = GEMM to (u,v,o) columns + pixel scatter (dpt.py:116-140)."""

    @staticmethod
    def forward(ctx, x, weight, bias, ct):
        x = _c(x)
        B, H, W, Cin = x.shape
        k = ct.kernel_size[0]
        w, b = engine.convt_weights(ct, x.dtype)
        x2 = x.view(-1, Cin)
        y = ops.gemm(x2, w, b)
        ctx.save_for_backward(x2)
        ctx.meta = (ct, k, (B, H, W, Cin), bias is not None)
        return ops.convt_scatter(y, B, H, W, k, ct.out_channels)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        ct, k, (B, H, W, Cin), has_b = ctx.meta
        dt = x2.dtype
        Cout = ct.out_channels
        dyg = ops.convt_gather(_c(dy), k)                                        # [B*H*W, k*k*Cout]
        dWg, dbg = _wgrad(dyg, x2, dt, has_b)                                    # [k*k*Cout, Cin], [k*k*Cout]
        dW = dWg.view(k, k, Cout, Cin).permute(3, 2, 0, 1)
        db = dbg.view(k * k, Cout).sum(0) if has_b else None
        dx = None
        if ctx.needs_input_grad[0]:
            wT = _w_t(ct, "ct", (ct.weight,),
                      lambda: ct.weight.detach().permute(2, 3, 1, 0).reshape(k * k * Cout, Cin).float().contiguous(), dt)
            dx = ops.gemm(dyg, wT).view(B, H, W, Cin)
        return dx, dW, db, None


def conv_transpose_ks(x, ct):
    return ConvTransposeFn.apply(x, ct.weight, ct.bias, ct)


@_sink_aware
class BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, crop):
        x = _c(x)
        ctx.geom = (x.shape[1], x.shape[2], Ho, Wo)
        return ops.bilinear_nhwc(x, Ho, Wo, crop)

    @staticmethod
    def backward(ctx, dy):
        Hi, Wi, Ho, Wo = ctx.geom
        return ops.bilinear_nhwc_bwd(_c(dy), Hi, Wi, Ho, Wo), None, None, None


def bilinear(x, Ho, Wo, crop=None):
    return BilinearFn.apply(x, Ho, Wo, crop)


@_sink_aware
class Conv1x1To4Fn(Function):
    """Regressor tail: features -> 4 decoded channels, fp32 output (dpt.py:271-277 conv2[2])."""

    @staticmethod
    def forward(ctx, x, weight, bias, w4, b4, relu_cell=None):
        x = _c(x)
        ctx.save_for_backward(x, w4)
        ctx.wshape, ctx.has_b, ctx.relu_cell = weight.shape, bias is not None, relu_cell
        return ops.conv1x1_to4(x, w4, b4)

    @staticmethod
    def backward(ctx, dout):
        x, w4 = ctx.saved_tensors
        dw = torch.zeros_like(w4)
        db = torch.zeros(4, dtype=torch.float32, device=x.device)
        # x straight out of a fused-ReLU conv3x3 (relu_cell from conv3x3(..., grad_mask_cell=...)): that ReLU's backward rides in this
        # kernel, and the cell tells the conv's backward not to mask again
        masked = ctx.relu_cell is not None
        dfeat = ops.conv1x1_to4_bwd(x, w4, _c(dout), dw, db, relu_mask=masked)
        if masked:
            ctx.relu_cell[0] = True
        return dfeat, dw.view(ctx.wshape), (db if ctx.has_b else None), None, None, None


def conv1x1_to4(x, conv, w4, b4, relu_cell=None):
    return Conv1x1To4Fn.apply(x, conv.weight, conv.bias, w4, b4, relu_cell)
