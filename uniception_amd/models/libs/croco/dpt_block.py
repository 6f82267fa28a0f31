"""DPT building blocks (reference: libs/croco/dpt_block.py:17-289) on NHWC maps and implicit-GEMM convolutions.

Public modules keep the reference's parameters/state_dict keys.  `forward` accepts BCHW-shaped tensors like the
reference; the fused head pipeline calls the `_nhwc` methods directly and never leaves channel-last layout.
"""
import torch.nn as nn

from .... import engine



_COMMUTE_TRAIN = __import__("os").environ.get("UNICEPTION_AMD_DPT_COMMUTE_TRAIN", "0") == "1"

def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def make_scratch(in_shape, out_shape, groups=1, expand=False):
    """Four bias-free 3x3 projections to the fusion width (dpt_block.py:21-80), registered under both
    `layerK_rn` and `layer_rn.(K-1)` like the reference (aliased state_dict entries)."""
    if groups != 1:
        raise engine.UcHipError("grouped convolutions are not supported by the HIP DPT head")
    scratch = nn.Module()
    outs = [out_shape * m for m in ((1, 2, 4, 8) if expand else (1, 1, 1, 1))]
    for i in range(4):
        setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(in_shape[i], outs[i], kernel_size=3, stride=1, padding=1, bias=False, groups=groups))
    scratch.layer_rn = nn.ModuleList([scratch.layer1_rn, scratch.layer2_rn, scratch.layer3_rn, scratch.layer4_rn])
    return scratch


def make_nonlinearity(nonlinearity, dim=None, on_channels=False):
    if nonlinearity == "relu":
        return nn.ReLU(False)
    raise engine.UcHipError(f"nonlinearity '{nonlinearity}' has no fused HIP epilogue (DUSt3R uses 'relu')")


def _to_nhwc(x):
    return engine.bchw_to_nhwc(x, engine.head_dtype())


def _to_bchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


class ResidualConvUnit_custom(nn.Module):
    """x + conv2(act(conv1(act(x)))) — 3x3 convs, activation applied on load inside the implicit GEMM."""

    def __init__(self, features, activation, bn):
        super().__init__()
        self.bn = bn
        self.groups = 1
        # (dpt_block.py:135-150: the convolutions carry a bias exactly when no BatchNorm follows them)
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=not self.bn, groups=1)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=not self.bn, groups=1)
        if self.bn:      # round 6: eval-mode BatchNorm folded into the convolution's weights (engine.conv3x3_bn_weights); training raises
            self.bn1 = nn.BatchNorm2d(features)
            self.bn2 = nn.BatchNorm2d(features)
        self.activation = activation

    def _nhwc(self, x, extra=None):
        """RCU(x) (+ extra), both residual adds fused into conv2's epilogue."""
        if not isinstance(self.activation, nn.ReLU):
            raise engine.UcHipError("only ReLU is fused into the HIP residual conv unit")
        # conv1's output is only ever consumed through the second ReLU, so that ReLU runs in conv1's epilogue and conv2
        # loads plain operands (ReLU-on-load costs ~14 % of an implicit-GEMM conv); x itself is needed un-activated for
        # the residual, so its ReLU stays on the load path of conv1.
        t = engine.conv3x3(x, self.conv1, relu_in=True, act="relu", bn=self.bn1 if self.bn else None)
        return engine.conv3x3(t, self.conv2, relu_in=False, residual=x, residual2=extra, bn=self.bn2 if self.bn else None)

    def forward(self, x):
        return _to_bchw_view(self._nhwc(_to_nhwc(x)))


class FeatureFusionBlock_custom(nn.Module):
    """(path + RCU1(skip)) -> RCU2 -> x2 bilinear (align_corners=True) -> 1x1 conv (dpt_block.py:180-255); computed as
    1x1 conv -> x2 bilinear (the two commute)."""

    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True, width_ratio=1):
        super().__init__()
        if width_ratio != 1:
            raise engine.UcHipError("width_ratio != 1 is not supported by the HIP DPT head")
        if not align_corners:
            raise engine.UcHipError("align_corners=False is not supported by the HIP DPT head")
        self.width_ratio = width_ratio
        self.deconv = deconv
        self.align_corners = align_corners
        self.groups = 1
        self.expand = expand
        out_features = features // 2 if expand else features
        self.out_conv = nn.Conv2d(features, out_features, kernel_size=1, stride=1, padding=0, bias=True, groups=1)
        self.resConfUnit1 = ResidualConvUnit_custom(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit_custom(features, activation, bn)

    def _nhwc(self, path, skip=None, crop=None):
        out = path if skip is None else self.resConfUnit1._nhwc(skip, extra=path)
        out = self.resConfUnit2._nhwc(out)
        B, H, W, _ = out.shape
        # The reference upsamples, then applies the 1x1 convolution (dpt_block.py:251-255).  Both are linear and the bilinear
        # weights of a pixel sum to 1 (bias included), so the two commute exactly in real arithmetic: the 1x1 GEMM runs on the
        # H x W map — a quarter of the rows — and the x2 resize on its output (same channel count, same resize cost).
        # (inference only by default: the training path keeps the reference's order, whose backward the gradient fixtures pin.  Round 6
        #  measured the commuted pair in training — both orders' gradients are exact in isolation (tools/scratch/check_commute_grad.py),
        #  but a 1e-7 rounding difference upstream flips one ReLU of the tiny odd-grid fixture and moves three head weights' gradients by
        #  1e-3 — for 0.5 % of a training step.  UNICEPTION_AMD_DPT_COMMUTE_TRAIN=1 takes the commuted order in training too.)
        if not _COMMUTE_TRAIN and engine._train(out, self.out_conv.weight):
            return engine.conv1x1(engine.bilinear(out, 2 * H, 2 * W, crop), self.out_conv)
        out = engine.conv1x1(out, self.out_conv)
        return engine.bilinear(out, 2 * H, 2 * W, crop)

    def forward(self, *xs):
        path = _to_nhwc(xs[0])
        skip = _to_nhwc(xs[1]) if len(xs) == 2 else None
        return _to_bchw_view(self._nhwc(path, skip))


def make_fusion_block(features, use_bn, width_ratio=1, nonlinearity="relu"):
    return FeatureFusionBlock_custom(features, make_nonlinearity(nonlinearity, features, on_channels=True), deconv=False,
                                     bn=use_bn, expand=False, align_corners=True, width_ratio=width_ratio)
