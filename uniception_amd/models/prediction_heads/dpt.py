"""DPT pointmap head (reference: prediction_heads/dpt.py:23-311) as an NHWC implicit-GEMM pipeline.

DPTFeature   : 4 token maps -> [1x1 conv (+ConvT k=s as GEMM+scatter | + 3x3 s2 conv)] -> 3x3 projection to
               feature_dim -> 4 fusion blocks (residual conv units with ReLU-on-load, x2 bilinear, 1x1 conv).
DPTRegressionProcessor : conv3x3 -> bilinear to (H,W) -> conv3x3+ReLU -> conv1x1 (Cin -> 4, fp32 output).
All intermediate maps are NHWC in the head compute dtype; BCHW tensors at the API are channels-last views.
"""
from dataclasses import dataclass
from typing import Iterable, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import autograd, engine, ops
from ..libs.croco.dpt_block import make_fusion_block, make_nonlinearity, make_scratch, pair
from .base import PixelTaskOutput, PredictionHeadLayeredInput


@dataclass
class DPTFeatureInput:
    features_upsampled_8x: Tensor  # [batch, dpt_output_feat_dim, 8*feat_height, 8*feat_width]
    target_output_shape: Tuple[int, int]


class DPTFeature(nn.Module):
    "DPT feature head: list of 4 BCHW token maps -> 8x upsampled fused feature map."

    def __init__(self, patch_size: Union[int, Tuple[int, int]] = 16, main_tasks: Iterable[str] = ("rgb",),
                 hooks: List[int] = [2, 5, 8, 11], input_feature_dims: Optional[Union[int, List[int]]] = 768,
                 layer_dims: List[int] = [96, 192, 384, 768], feature_dim: int = 256, use_bn: bool = False,
                 output_width_ratio=1, pretrained_checkpoint_path: str = None, checkpoint_gradient: bool = False,
                 nonlinearity: str = "relu", *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.patch_size = pair(patch_size)
        self.main_tasks = main_tasks
        self.hooks = hooks
        self.layer_dims = layer_dims
        self.feature_dim = feature_dim
        self.checkpoint_gradient = checkpoint_gradient
        if isinstance(input_feature_dims, int):
            input_feature_dims = 4 * [input_feature_dims]
        else:
            assert isinstance(input_feature_dims, List) and len(input_feature_dims) == 4
        self.input_feature_dims = input_feature_dims

        self.scratch = make_scratch(layer_dims, feature_dim, groups=1, expand=False)
        for i in (1, 2, 3, 4):
            setattr(self.scratch, f"refinenet{i}", make_fusion_block(feature_dim, use_bn, output_width_ratio, nonlinearity=nonlinearity))
        # unused in forward (refinenet4 has no skip input); the reference deletes it for DDP (dpt.py:82-83)
        del self.scratch.refinenet4.resConfUnit1
        if self.input_feature_dims is not None:
            self.init(input_feature_dims=input_feature_dims)
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained DPT dense feature head from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))

    def init(self, input_feature_dims: Union[int, List[int]] = 768):
        "Build the reassemble layers that depend on the token width (dpt.py:94-178)."
        if isinstance(input_feature_dims, int):
            input_feature_dims = 4 * [input_feature_dims]
        self.input_feature_dims = [dt * len(self.main_tasks) for dt in input_feature_dims]
        d, L = self.input_feature_dims, self.layer_dims
        act_postprocess = [
            nn.Sequential(nn.Conv2d(d[0], L[0], kernel_size=1, stride=1, padding=0),
                          nn.ConvTranspose2d(L[0], L[0], kernel_size=4, stride=4, padding=0, bias=True, dilation=1, groups=1)),
            nn.Sequential(nn.Conv2d(d[1], L[1], kernel_size=1, stride=1, padding=0),
                          nn.ConvTranspose2d(L[1], L[1], kernel_size=2, stride=2, padding=0, bias=True, dilation=1, groups=1)),
            nn.Sequential(nn.Conv2d(d[2], L[2], kernel_size=1, stride=1, padding=0)),
            nn.Sequential(nn.Conv2d(d[3], L[3], kernel_size=1, stride=1, padding=0),
                          nn.Conv2d(L[3], L[3], kernel_size=3, stride=2, padding=1)),
        ]
        self.input_process = nn.ModuleList(
            [nn.Sequential(act_, layer_rn_) for act_, layer_rn_ in zip(act_postprocess, self.scratch.layer_rn)])

    def _reassemble(self, idx: int, x: Tensor) -> Tensor:
        act, layer_rn = self.input_process[idx][0], self.input_process[idx][1]
        x = engine.conv1x1(x, act[0])
        if idx in (0, 1):
            x = engine.conv_transpose_ks(x, act[1])
        elif idx == 3:
            x = engine.conv3x3(x, act[1])
        return engine.conv3x3(x, layer_rn)

    def _nhwc(self, layered: List[Tensor]) -> Tensor:
        layers = [self._reassemble(i, x) for i, x in enumerate(layered)]
        s = self.scratch
        path4 = s.refinenet4._nhwc(layers[3], None, crop=(layers[2].shape[1], layers[2].shape[2]))
        path3 = s.refinenet3._nhwc(path4, layers[2])
        path2 = s.refinenet2._nhwc(path3, layers[1])
        return s.refinenet1._nhwc(path2, layers[0])

    def forward(self, dpt_input: PredictionHeadLayeredInput) -> DPTFeatureInput:
        assert self.input_feature_dims is not None, "Need to call init(input_feature_dims) function first"
        feats = dpt_input.list_features
        for hook_idx, hook in enumerate(self.hooks):
            assert feats[hook].shape[1] == self.input_feature_dims[hook_idx], (
                f"Input feature dimension mismatch at hook {hook}. Expected BCHW")
        dt = engine.head_dtype()
        out = self._nhwc([engine.bchw_to_nhwc(feats[hook], dt) for hook in self.hooks])
        return DPTFeatureInput(features_upsampled_8x=out.permute(0, 3, 1, 2), target_output_shape=dpt_input.target_output_shape)


class DPTRegressionProcessor(nn.Module):
    "8x-upsampled DPT features -> regression channels at the target resolution (dpt.py:238-311)."

    def __init__(self, input_feature_dim: int, output_dim: int, hidden_dims: Optional[List[int]] = None,
                 pretrained_checkpoint_path: str = None, checkpoint_gradient: bool = False, nonlinearity: str = "relu",
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        if hidden_dims is None:
            hidden_dims = [input_feature_dim // 2] * 2
        else:
            assert isinstance(hidden_dims, List) and len(hidden_dims) == 2
        self.checkpoint_gradient = checkpoint_gradient
        self.conv1 = nn.Conv2d(input_feature_dim, hidden_dims[0], kernel_size=3, stride=1, padding=1)
        self.conv2 = nn.Sequential(
            nn.Conv2d(hidden_dims[0], hidden_dims[1], kernel_size=3, stride=1, padding=1),
            make_nonlinearity(nonlinearity),
            nn.Conv2d(hidden_dims[1], output_dim, kernel_size=1, stride=1, padding=0),
        )
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained DPT regression processor from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))

    def forward(self, dpt_processor_input: DPTFeatureInput):
        x = dpt_processor_input.features_upsampled_8x
        H, W = dpt_processor_input.target_output_shape
        dt = engine.head_dtype()
        x = engine.bchw_to_nhwc(x, dt)
        x = engine.conv3x3(x, self.conv1)
        x = engine.bilinear(x, H, W)
        last = self.conv2[2]
        fused = engine.conv3x3_tail4(x, self.conv2[0], "relu", last)     # conv2.0 -> ReLU -> conv2.2 in one kernel (inference)
        if fused is not None:
            return PixelTaskOutput(decoded_channels=fused.permute(0, 3, 1, 2))
        to4 = last.out_channels == 4 and last.in_channels <= 256 and last.in_channels % 8 == 0
        # training: the 1x1 tail is the ONLY consumer of this ReLU's output, so the ReLU's backward rides in the tail's backward kernel
        # (one-element cell shared by the two autograd functions: autograd.conv3x3 / conv1x1_to4)
        cell = [False] if (to4 and engine._train(x, last.weight)) else None
        x = engine.conv3x3(x, self.conv2[0], act="relu", grad_mask_cell=cell)
        if to4:
            w = engine.prepared(last, "c1x4", (last.weight, last.bias),
                                lambda: (last.weight.detach().reshape(4, -1).float().contiguous(),
                                         last.bias.detach().float().contiguous() if last.bias is not None
                                         else torch.zeros(4, device=last.weight.device)))
            if engine._train(x, last.weight):
                out = autograd.conv1x1_to4(x, last, w[0], w[1], relu_cell=cell)
            else:
                out = ops.conv1x1_to4(x, w[0], w[1])  # fp32 NHWC [B,H,W,4]
        else:
            wl, bl = engine.conv1x1_weights(last, dt)
            B, Hh, Ww, Cin = x.shape
            if engine._train(x, last.weight):
                out = autograd.linear(x.view(-1, Cin), last.weight, last.bias, last, dt, torch.float32).view(B, Hh, Ww, -1)
            else:
                out = ops.gemm(x.view(-1, Cin), wl, bl, out_dtype=torch.float32).view(B, Hh, Ww, -1)
        return PixelTaskOutput(decoded_channels=out.permute(0, 3, 1, 2))


class DPTSegmentationProcessor(nn.Module):
    """8x-upsampled DPT features -> `output_dim` channels at the target resolution (reference: prediction_heads/dpt.py:314-381):
    conv3x3 (no bias) -> ReLU -> [Dropout, identity in eval] -> conv1x1 -> bilinear (align_corners=True) to the target shape.
    The 1x1 conv and the resize run in fp32 on a channel count padded to the kernels' 8-channel granule."""

    def __init__(self, input_feature_dim: int, output_dim: int, hidden_dim: Optional[int] = None, use_bn: bool = False,
                 pretrained_checkpoint_path: str = None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if hidden_dim is None:
            hidden_dim = input_feature_dim
        self.output_dim = output_dim
        self.conv = nn.Sequential(
            nn.Conv2d(input_feature_dim, hidden_dim, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(hidden_dim) if use_bn else nn.Identity(),      # (dpt.py:346; eval mode: folded into the convolution, round 6)
            nn.ReLU(True),
            nn.Dropout(0.1, False),
            nn.Conv2d(hidden_dim, output_dim, kernel_size=1),
        )
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained DPT segmentation processor from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))

    def forward(self, dpt_processor_input: DPTFeatureInput):
        x = dpt_processor_input.features_upsampled_8x
        H, W = dpt_processor_input.target_output_shape
        if self.training and self.conv[3].p > 0:
            raise engine.UcHipError("Dropout in training mode is not supported by the HIP path (call .eval(), or build with p = 0)")
        dt = engine.head_dtype()
        last = self.conv[4]
        train = engine._train(x, self.conv[0].weight, last.weight)
        bn = self.conv[1] if isinstance(self.conv[1], nn.BatchNorm2d) else None
        x = engine.conv3x3(engine.bchw_to_nhwc(x, dt), self.conv[0], act="relu", bn=bn)
        cpad = (self.output_dim + 7) // 8 * 8       # output channels padded with zero rows to the 8-channel granule of the NHWC kernels
        if train:
            y = autograd.padded_conv1x1(x, last, dt, cpad)
        else:
            wl, bl = autograd.padded_conv1x1_weights(last, dt, cpad)
            B, Hh, Ww, Cin = x.shape
            y = ops.gemm(x.view(-1, Cin), wl, bl, out_dtype=torch.float32).view(B, Hh, Ww, cpad)
        y = engine.bilinear(y, H, W)
        return PixelTaskOutput(decoded_channels=y[..., :self.output_dim].permute(0, 3, 1, 2))


class DPTFeatureDoubleUpsampling(nn.Module):
    """Two-level DPT feature head (reference: prediction_heads/dpt.py:385-573): two BCHW token maps -> [1x1 conv | 1x1 conv + 3x3 s2
    conv] -> 3x3 projections to feature_dim -> refinenet4 (cropped to level-3's size) -> refinenet3 with the skip: a 2x
    upsampled fused map (returned, like the reference, in the `features_upsampled_8x` field)."""

    def __init__(self, patch_size: Union[int, Tuple[int, int]] = 16, main_tasks: Iterable[str] = ("rgb",), hooks: List[int] = [0, 1],
                 input_feature_dims: Optional[Union[int, List[int]]] = 768, layer_dims: List[int] = [384, 768],
                 feature_dim: int = 256, use_bn: bool = False, output_width_ratio=1, pretrained_checkpoint_path: str = None,
                 checkpoint_gradient: bool = False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.patch_size = pair(patch_size)
        self.main_tasks = main_tasks
        self.hooks = hooks
        self.layer_dims = layer_dims
        self.feature_dim = feature_dim
        self.checkpoint_gradient = checkpoint_gradient
        if isinstance(input_feature_dims, int):
            input_feature_dims = 2 * [input_feature_dims]
        else:
            assert isinstance(input_feature_dims, List) and len(input_feature_dims) == 2
        self.input_feature_dims = input_feature_dims
        self.scratch = self.make_scratch_2(layer_dims, feature_dim, groups=1, expand=False)
        self.scratch.refinenet3 = make_fusion_block(feature_dim, use_bn, output_width_ratio)
        self.scratch.refinenet4 = make_fusion_block(feature_dim, use_bn, output_width_ratio)
        del self.scratch.refinenet4.resConfUnit1      # unused (no skip input); deleted like the reference does for DDP
        if self.input_feature_dims is not None:
            self.init(input_feature_dims=input_feature_dims)
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained DPT dense feature head from {pretrained_checkpoint_path}")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))

    def make_scratch_2(self, in_shape, out_shape, groups=1, expand=False):
        scratch = nn.Module()
        o3, o4 = (out_shape * 4, out_shape * 8) if expand else (out_shape, out_shape)
        scratch.layer3_rn = nn.Conv2d(in_shape[0], o3, kernel_size=3, stride=1, padding=1, bias=False, groups=groups)
        scratch.layer4_rn = nn.Conv2d(in_shape[1], o4, kernel_size=3, stride=1, padding=1, bias=False, groups=groups)
        scratch.layer_rn = nn.ModuleList([scratch.layer3_rn, scratch.layer4_rn])
        return scratch

    def init(self, input_feature_dims: Union[int, List[int]] = 768):
        if isinstance(input_feature_dims, int):
            input_feature_dims = 2 * [input_feature_dims]
        self.input_feature_dims = [dt * len(self.main_tasks) for dt in input_feature_dims]
        d, L = self.input_feature_dims, self.layer_dims
        act_postprocess = [
            nn.Sequential(nn.Conv2d(d[0], L[0], kernel_size=1, stride=1, padding=0)),
            nn.Sequential(nn.Conv2d(d[1], L[1], kernel_size=1, stride=1, padding=0),
                          nn.Conv2d(L[1], L[1], kernel_size=3, stride=2, padding=1)),
        ]
        self.input_process = nn.ModuleList(
            [nn.Sequential(act_, layer_rn_) for act_, layer_rn_ in zip(act_postprocess, self.scratch.layer_rn)])

    def forward(self, dpt_input: PredictionHeadLayeredInput) -> DPTFeatureInput:
        assert self.input_feature_dims is not None, "Need to call init(input_feature_dims) function first"
        feats = dpt_input.list_features
        for hook_idx, hook in enumerate(self.hooks):
            assert feats[hook].shape[1] == self.input_feature_dims[hook_idx], (
                f"Input feature dimension mismatch at hook {hook}. Expected BCHW")
        dt = engine.head_dtype()
        xs = [engine.bchw_to_nhwc(feats[hook], dt) for hook in self.hooks]
        l0 = engine.conv3x3(engine.conv1x1(xs[0], self.input_process[0][0][0]), self.input_process[0][1])
        t = engine.conv3x3(engine.conv1x1(xs[1], self.input_process[1][0][0]), self.input_process[1][0][1])
        l1 = engine.conv3x3(t, self.input_process[1][1])
        path4 = self.scratch.refinenet4._nhwc(l1, None, crop=(l0.shape[1], l0.shape[2]))
        out = self.scratch.refinenet3._nhwc(path4, l0)
        return DPTFeatureInput(features_upsampled_8x=out.permute(0, 3, 1, 2), target_output_shape=dpt_input.target_output_shape)
