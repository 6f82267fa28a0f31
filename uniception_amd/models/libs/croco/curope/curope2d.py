"""HIP replacement of the reference's `curope` CUDA extension (libs/croco/curope/{curope.cpp,kernels.cu,curope2d.py}).

`rope_2d(tokens[B,N,H,D], positions[B,N,2] int64, base, fwd)` keeps the pybind entry point's contract:
in place, returns None, RuntimeError on rank/shape/device mismatch — and additionally supports bf16, strided
views (q/k slices of a fused qkv buffer) and launches on PyTorch's *current* stream instead of the legacy
default stream.
"""
import torch

from uniception_amd import ops


def rope_2d(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    ops.rope_2d_(tokens, positions, base, fwd)


class cuRoPE2D_func(torch.autograd.Function):
    """In-place forward, inverse rotation of the incoming gradient in backward (curope2d.py:12-28)."""

    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1):
        ctx.save_for_backward(positions)
        ctx.saved_base = base
        ctx.saved_F0 = F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        positions, base, F0 = ctx.saved_tensors[0], ctx.saved_base, ctx.saved_F0
        if grad_res.stride(-1) != 1:
            grad_res = grad_res.contiguous()
        rope_2d(grad_res, positions, base, -F0)
        ctx.mark_dirty(grad_res)
        return grad_res, None, None, None


class cuRoPE2D(torch.nn.Module):
    """tokens [B,H,N,D] (any strides with unit last stride), positions [B,N,2] -> same tensor, rotated in place."""

    _uc_native_rope = True  # lets the fused QKV epilogue take over when this module is the positional encoding

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens, positions):
        if positions.dtype != torch.int64:
            positions = positions.long()
        cuRoPE2D_func.apply(tokens.transpose(1, 2), positions.contiguous(), self.base, self.F0)
        return tokens
