"""RoPE-2D positional embedding (reference: libs/croco/pos_embed.py:100-155).
The reference picks `cuRoPE2D` when its CUDA extension imports and a pure-PyTorch class otherwise; here
the HIP kernel is always the implementation."""
from .curope import cuRoPE2D

RoPE2D = cuRoPE2D

__all__ = ["RoPE2D"]
