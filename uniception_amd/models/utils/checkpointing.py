"""Gradient checkpointing of token-stream blocks (reference: encoders/base.py:139-152, info_sharing/base.py:59-72 — the class-swapping
`_CheckpointingWrapper` around `torch.utils.checkpoint(..., use_reentrant=False)`).

The HIP sub-layer Functions (autograd.py) keep what their backward needs through `ctx.save_for_backward`, so the non-reentrant
checkpoint's saved-tensor hooks see every activation a block holds: inside a wrapped block they are dropped after the forward and the
block's forward is run again — the same kernels on the same inputs, the same bits — when its backward needs them.  Parameter gradients
written straight into a trainer's flat buffer (the gradient sink) and LayerScale's unfolded gradients are produced by that second
backward exactly as without the wrapper.  Outside training (no gradient requested) the wrapper is transparent: the fused inference
pipeline runs, nothing is recorded.

Random masks: a block with a non-zero proj_drop / Mlp drop / DropPath / attn_drop rate draws its masks from PyTorch's generator in
every forward; the re-computation must draw the SAME ones (the original Function ctx applies its own masks to tensors the second
forward produced), so such blocks are checkpointed with `preserve_rng_state=True` — the reference's default.  Blocks without any
random mask skip the generator save / restore (a device round trip per block)."""
import torch
from torch import nn
from torch.utils.checkpoint import checkpoint

from ... import autograd


def has_random_masks(module: nn.Module) -> bool:
    "True when `module` (a block) holds a dropout-like sub-module with a non-zero rate: its training forward consumes the generator."
    for m in module.modules():
        if isinstance(m, nn.Dropout) and m.p > 0.0:
            return True
        if float(getattr(m, "drop_prob", 0.0) or 0.0) > 0.0:          # DropPath (libs/croco/blocks.py, timm-style)
            return True
        if float(getattr(m, "dropout_p", 0.0) or 0.0) > 0.0:          # CroCo Attention's attn_drop rate
            return True
    return False


def wrap_module_with_gradient_checkpointing(module: nn.Module) -> nn.Module:
    "Swap `module`'s class for a subclass whose forward_tokens / forward re-compute in the backward pass.  Returns the module."
    if getattr(module.__class__, "_restore_cls", None) is not None:
        return module           # already wrapped

    class _CheckpointingWrapper(module.__class__):
        _restore_cls = module.__class__

        def _ckpt(self, fn, *args, **kwargs):
            tensors = [a for a in args if isinstance(a, torch.Tensor)]
            if not torch.is_grad_enabled() or not autograd.grad_needed(*tensors, *self.parameters()):
                return fn(*args, **kwargs)
            return checkpoint(fn, *args, use_reentrant=False, preserve_rng_state=self.training and has_random_masks(self), **kwargs)

        def forward_tokens(self, *args, **kwargs):
            return self._ckpt(super().forward_tokens, *args, **kwargs)

    _CheckpointingWrapper.__name__ = f"Checkpointed{module.__class__.__name__}"
    module.__class__ = _CheckpointingWrapper
    return module
