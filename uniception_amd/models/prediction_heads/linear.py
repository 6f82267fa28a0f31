"""Linear pointmap head (reference: prediction_heads/linear.py:15-84): a 1x1 conv (one GEMM on the token matrix)
followed by pixel_shuffle(P) (one scatter kernel)."""
import torch
import torch.nn as nn

from ... import autograd, engine, ops
from .base import PixelTaskOutput, PredictionHeadInput


class LinearFeature(nn.Module):
    "Patch features -> per-pixel features: Conv2d 1x1 (C -> out_dim*P^2) + pixel_shuffle(P)."

    def __init__(self, input_feature_dim: int, output_dim: int, patch_size: int, pretrained_checkpoint_path: str = None,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.input_feature_dim = input_feature_dim
        self.output_dim = output_dim
        self.patch_size = patch_size
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        self.linear = nn.Conv2d(in_channels=input_feature_dim, out_channels=output_dim * (patch_size**2), kernel_size=1,
                                stride=1, padding=0, bias=True)
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained linear dense feature head from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))

    def forward(self, feature_input: PredictionHeadInput):
        x = feature_input.last_feature
        assert x.shape[1] == self.input_feature_dim, f"Input feature dimension mismatch: {x.shape[1]} != {self.input_feature_dim}"
        B, C, h, w = x.shape
        dt = engine.head_dtype()
        if autograd.grad_needed(x, self.linear.weight):
            tok = x.float().permute(0, 2, 3, 1).reshape(B * h * w, C)
            y = autograd.linear(tok, self.linear.weight, self.linear.bias, self.linear, dt, torch.float32)
            return PixelTaskOutput(decoded_channels=autograd.pixel_shuffle(y, B, h, w, self.patch_size, self.output_dim))
        tok = engine.bchw_to_nhwc(x, dt).reshape(B * h * w, C)
        wl, bl = engine.conv1x1_weights(self.linear, dt)
        y = ops.gemm(tok, wl, bl, out_dtype=torch.float32)
        return PixelTaskOutput(decoded_channels=ops.pixel_shuffle(y, B, h, w, self.patch_size, self.output_dim))
