"""Pointmap + confidence adaptors (reference: prediction_heads/adaptors.py:299-355, 1035-1096, 1189-1230, 1269-1293).

`PointMapWithConfidenceAdaptor` ("exp" pointmap, "exp" confidence — the DUSt3R setting) runs as ONE kernel that
also produces the BHWC layout the factory returns: `value` / `confidence` are BCHW-shaped views of BHWC memory,
so the factory's `.permute(0, 2, 3, 1).contiguous()` is free.  The component adaptors are kept for interface
compatibility and compose the same kernel.
"""
from math import isfinite

import numpy as np
import torch

from ... import autograd, engine, ops
from .base import AdaptorInput, RegressionWithConfidenceAdaptorOutput, UniCeptionAdaptorBase


def _as_f32_map(x):
    if x.dtype != torch.float32:
        x = x.float()
    B, C, H, W = x.shape
    if x.stride(2) != W * x.stride(3):
        x = x.contiguous()
    return x


class PointMapAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, mode: str, vmin: float = -np.inf, vmax: float = np.inf, *args, **kwargs):
        super().__init__(name, required_channels=3, *args, **kwargs)
        self.mode = mode
        self.vmin = vmin
        self.vmax = vmax
        self.no_bounds = (vmin == -float("inf")) and (vmax == float("inf"))

    def forward(self, adaptor_input: AdaptorInput):
        raise engine.UcHipError("PointMapAdaptor runs fused inside PointMapWithConfidenceAdaptor on the HIP path")


class ConfidenceAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, confidence_type: str, vmin: float, vmax: float, *args, **kwargs):
        super().__init__(name, required_channels=1, *args, **kwargs)
        self.confidence_type = confidence_type
        self.vmin = vmin
        self.vmax = vmax
        assert vmin < vmax, "vmin must be less than vmax"
        if confidence_type == "sigmoid":
            assert isfinite(vmin) and isfinite(vmax), "vmin and vmax must be finite for sigmoid confidence"
            assert vmin >= 0

    def forward(self, adaptor_input: AdaptorInput):
        raise engine.UcHipError("ConfidenceAdaptor runs fused inside PointMapWithConfidenceAdaptor on the HIP path")


class ValueWithConfidenceAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, value_adaptor: UniCeptionAdaptorBase, confidence_adaptor: UniCeptionAdaptorBase, *args, **kwargs):
        super().__init__(name, required_channels=value_adaptor.required_channels + confidence_adaptor.required_channels,
                         *args, **kwargs)
        self.value_adaptor = value_adaptor
        self.confidence_adaptor = confidence_adaptor

    def forward(self, adaptor_input: AdaptorInput):
        va, ca = self.value_adaptor, self.confidence_adaptor
        if not (isinstance(va, PointMapAdaptor) and isinstance(ca, ConfidenceAdaptor)):
            raise engine.UcHipError("only PointMapAdaptor + ConfidenceAdaptor have a fused HIP adaptor kernel")
        if va.mode != "exp" or not va.no_bounds or ca.confidence_type != "exp":
            raise engine.UcHipError("the HIP adaptor kernel implements pointmap_mode='exp' without bounds and confidence_type='exp'")
        x = adaptor_input.adaptor_feature
        assert x.shape[1] == 4, "pointmap + confidence needs 4 channels"
        if autograd.grad_needed(x):
            pts, conf = autograd.pointmap_adaptor(_as_f32_map(x), float(ca.vmin), float(ca.vmax))
        else:
            pts, conf = ops.pointmap_adaptor(_as_f32_map(x), float(ca.vmin), float(ca.vmax))
        return RegressionWithConfidenceAdaptorOutput(value=pts.permute(0, 3, 1, 2), confidence=conf.permute(0, 3, 1, 2))


class PointMapWithConfidenceAdaptor(ValueWithConfidenceAdaptor):
    def __init__(self, name: str, pointmap_mode: str, pointmap_vmin: float, pointmap_vmax: float, confidence_type: str,
                 confidence_vmin: float, confidence_vmax: float, *args, **kwargs):
        pointmap_adaptor = PointMapAdaptor(name=f"{name}", mode=pointmap_mode, vmin=pointmap_vmin, vmax=pointmap_vmax)
        confidence_adaptor = ConfidenceAdaptor(name=f"{name}_confidence", confidence_type=confidence_type,
                                               vmin=confidence_vmin, vmax=confidence_vmax)
        super().__init__(name, value_adaptor=pointmap_adaptor, confidence_adaptor=confidence_adaptor, *args, **kwargs)
