"""Fused-attention switch kept for interface compatibility (reference: utils/config.py:13-34).
The HIP path always runs the fused flash-style kernel; the flag only records the caller's choice."""
import os
import warnings

__all__ = ["use_fused_attn", "set_fused_attn"]

_USE_FUSED_ATTN = int(os.environ.get("UNICEPTION_FUSED_ATTN", "1"))


def use_fused_attn() -> bool:
    return _USE_FUSED_ATTN > 0


def set_fused_attn(enable: bool = True):
    global _USE_FUSED_ATTN
    if not enable:
        warnings.warn("uniception_amd always uses its fused HIP attention kernel; set_fused_attn(False) is recorded but has no effect.")
    _USE_FUSED_ATTN = 1 if enable else 0
