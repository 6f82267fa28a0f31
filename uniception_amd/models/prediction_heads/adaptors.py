"""Adaptors of the prediction heads (reference: prediction_heads/adaptors.py:25-2300) on the HIP path.

Every adaptor of the reference splits the decoded channels, applies a small per-pixel transform to each group and concatenates
the results.  Here an adaptor contributes *segments* of a per-pixel channel program (`uc_adaptor_program`, one kernel for the
whole composition, fp32); composite adaptors concatenate the segments of their parts.  Outputs are BCHW-shaped views of one
NHWC buffer, so the factory's `.permute(0, 2, 3, 1).contiguous()` is free.

`PointMapWithConfidenceAdaptor` in the DUSt3R setting ("exp" pointmap without bounds, "exp" confidence) keeps its dedicated
fused kernel (`uc_pointmap_adaptor`) and its HIP backward; every other configuration trains through the gradient of the same
channel program (`uc_adaptor_program_bwd`).

The reference's adaptors take any (B x C x ...) tensor (the pose heads hand over [B, 7], pose_head.py:157-179): inputs that are
not 4-D maps are viewed as [B, C, -1, 1], run through the program and reshaped back.

The named composite classes of the reference differ only in which parts they combine; their constructors take the parts'
parameters as prefixed groups in a fixed order (`pointmap_*`, `ray_origins_*`, `ray_directions_*`, `depth_*`, `scene_flow_*`,
`quaternions_*`, `confidence_*`).  They are generated here from that table with identical parameter names and order.
"""
from math import isfinite
from typing import List, Tuple, Union

import numpy as np
import torch

from ... import autograd, engine, ops
from ..._lib import (UC_AD_CONF_EXP, UC_AD_CONF_SIGMOID, UC_AD_COV2D, UC_AD_DIR, UC_AD_ELEM, UC_AD_FLOW, UC_AD_FLOWCOORD,
                     UC_AD_MASK, UC_AD_NORM, UC_AD_ZEXP, AdaptorSeg)
from .base import (AdaptorInput, AdaptorOutput, Covariance2DAdaptorOutput, MaskAdaptorOutput, RegressionAdaptorOutput,
                   RegressionWithConfidenceAdaptorOutput, RegressionWithConfidenceAndMaskAdaptorOutput,
                   RegressionWithMaskAdaptorOutput, UniCeptionAdaptorBase)

_INF = float("inf")
_MODES = {"linear": 0, "square": 1, "exp": 2}


def _as_f32_map(x):
    if x.dtype != torch.float32:
        x = x.float()
    B, C, H, W = x.shape
    if x.stride(2) != W * x.stride(3):
        x = x.contiguous()
    return x


def _seg(op, c0, n, o0, mode=0, flags=0, p=(0.0, 0.0, 0.0, 0.0), vmin=-_INF, vmax=_INF):
    s = AdaptorSeg()
    s.op, s.mode, s.flags, s.c0, s.n, s.o0 = op, mode, flags, c0, n, o0
    for i, v in enumerate(p):
        s.p[i] = float(v)
    s.vmin, s.vmax = float(vmin), float(vmax)
    return s


def _bchw(out_nhwc, a, b):
    return out_nhwc[..., a:b].permute(0, 3, 1, 2)


def _run(adaptor_input: AdaptorInput, segs):
    x = adaptor_input.adaptor_feature
    cout = max(s.o0 + (2 if s.op == UC_AD_MASK else (7 if s.op == UC_AD_COV2D else s.n)) for s in segs)
    return _program(x, segs, cout)


def _program(x, segs, cout):
    "One pass over the decoded channels; with gradients enabled the pass is recorded with its HIP backward (uc_adaptor_program_bwd)."
    if autograd.grad_needed(x):
        return autograd.adaptor_program(_as_f32_map(x), segs, cout)
    return ops.adaptor_program(_as_f32_map(x), segs, cout)


class _ProgramAdaptor(UniCeptionAdaptorBase):
    """An adaptor whose forward is a list of channel-program segments: input channels [c0, c0 + required_channels) -> output
    channels [o0, o0 + out_channels)."""
    out_channels: int = 0
    _output_cls = RegressionAdaptorOutput

    def segments(self, c0: int, o0: int, shape_hw) -> List[AdaptorSeg]:
        raise NotImplementedError

    def forward(self, adaptor_input: AdaptorInput):
        x = adaptor_input.adaptor_feature
        assert x.shape[1] == self.required_channels, f"{type(self).__name__} needs {self.required_channels} channels, got {x.shape[1]}"
        if x.dim() != 4:      # (B x C ...) that is not a map — the pose heads' [B, 3 | 4 | 7]: a [B, C, n, 1] map, reshaped back
            shp = tuple(x.shape)
            as_map = AdaptorInput(adaptor_feature=x.reshape(shp[0], shp[1], -1, 1), output_shape_hw=adaptor_input.output_shape_hw)
            out = _run(as_map, self.segments(0, 0, adaptor_input.output_shape_hw))
            return self._output_cls(value=_bchw(out, 0, self.out_channels).reshape((shp[0], self.out_channels) + shp[2:]))
        out = _run(adaptor_input, self.segments(0, 0, adaptor_input.output_shape_hw))
        return self._output_cls(value=_bchw(out, 0, self.out_channels))


class _Elementwise(_ProgramAdaptor):
    "linear / square / exp per channel, then clip (Depth, Scale, SceneFlow)."
    _channels = 1

    def __init__(self, name: str, mode: str, vmin: float, vmax: float, *args, **kwargs):
        super().__init__(name, required_channels=self._channels, *args, **kwargs)
        self.mode, self.vmin, self.vmax = mode, vmin, vmax
        self.no_bounds = (vmin == -_INF) and (vmax == _INF)
        self.out_channels = self._channels

    def segments(self, c0, o0, shape_hw):
        if self.mode not in _MODES:
            raise ValueError(f"Invalid mode: {self.mode}")
        return [_seg(UC_AD_ELEM, c0, self._channels, o0, mode=_MODES[self.mode], vmin=self.vmin, vmax=self.vmax)]


class ScaleAdaptor(_Elementwise):
    _output_cls = AdaptorOutput

    def __init__(self, name: str, mode: str, vmin: float = 0, vmax: float = np.inf, *args, **kwargs):
        super().__init__(name, mode, vmin, vmax, *args, **kwargs)


class DepthAdaptor(_Elementwise):
    def __init__(self, name: str, mode: str, vmin: float = 0, vmax: float = np.inf, *args, **kwargs):
        super().__init__(name, mode, vmin, vmax, *args, **kwargs)


class SceneFlowAdaptor(_Elementwise):
    _channels = 3

    def __init__(self, name: str, mode: str, vmin: float = -np.inf, vmax: float = np.inf, *args, **kwargs):
        super().__init__(name, mode, vmin, vmax, *args, **kwargs)


class _Radial(_ProgramAdaptor):
    "direction x f(distance to the origin) for a 3-vector: linear | square | exp (expm1), then clip (PointMap, RayOrigins, CamTranslation)."

    def __init__(self, name: str, mode: str, vmin: float = -np.inf, vmax: float = np.inf, *args, **kwargs):
        super().__init__(name, required_channels=3, *args, **kwargs)
        self.mode, self.vmin, self.vmax = mode, vmin, vmax
        self.no_bounds = (vmin == -_INF) and (vmax == _INF)
        self.out_channels = 3

    def segments(self, c0, o0, shape_hw):
        if self.mode == "linear":
            return [_seg(UC_AD_ELEM, c0, 3, o0, mode=0, vmin=self.vmin, vmax=self.vmax)]
        if self.mode in ("square", "exp"):
            return [_seg(UC_AD_NORM, c0, 3, o0, mode=_MODES[self.mode], vmin=self.vmin, vmax=self.vmax)]
        if self.mode == "z_exp" and isinstance(self, PointMapAdaptor):
            return [_seg(UC_AD_ZEXP, c0, 3, o0, vmin=self.vmin, vmax=self.vmax)]
        raise ValueError(f"Invalid mode: {self.mode}")


class PointMapAdaptor(_Radial):
    pass


class RayOriginsAdaptor(_Radial):
    pass


class CamTranslationAdaptor(_Radial):
    _output_cls = AdaptorOutput


class RayDirectionsAdaptor(_ProgramAdaptor):
    def __init__(self, name: str, mode: str, normalize_to_unit_sphere: bool, normalize_to_unit_image_plane: bool,
                 vmin: float = -np.inf, vmax: float = np.inf, clamp_min_of_z_dir: bool = False, z_dir_min: float = 1, *args, **kwargs):
        super().__init__(name, required_channels=3, *args, **kwargs)
        self.mode = mode
        self.normalize_to_unit_sphere = normalize_to_unit_sphere
        self.normalize_to_unit_image_plane = normalize_to_unit_image_plane
        self.vmin, self.vmax = vmin, vmax
        self.clamp_min_of_z_dir, self.z_dir_min = clamp_min_of_z_dir, z_dir_min
        self.no_bounds = (vmin == -_INF) and (vmax == _INF)
        self.out_channels = 3

    def segments(self, c0, o0, shape_hw):
        if self.mode != "linear":
            raise ValueError(f"Invalid mode: {self.mode}")
        flags = (1 if self.clamp_min_of_z_dir else 0) | (2 if self.normalize_to_unit_sphere else (4 if self.normalize_to_unit_image_plane else 0))
        return [_seg(UC_AD_DIR, c0, 3, o0, flags=flags, p=(self.z_dir_min, 0, 0, 0), vmin=self.vmin, vmax=self.vmax)]


class QuaternionsAdaptor(_ProgramAdaptor):
    "Quaternions (x, y, z, w): clip, optional normalization."
    _output_cls = AdaptorOutput

    def __init__(self, name: str, mode: str, normalize: bool, vmin: float = -np.inf, vmax: float = np.inf, *args, **kwargs):
        super().__init__(name, required_channels=4, *args, **kwargs)
        self.mode, self.normalize, self.vmin, self.vmax = mode, normalize, vmin, vmax
        self.no_bounds = (vmin == -_INF) and (vmax == _INF)
        self.out_channels = 4

    def segments(self, c0, o0, shape_hw):
        if self.mode != "linear":
            raise ValueError(f"Invalid mode: {self.mode}")
        return [_seg(UC_AD_DIR, c0, 4, o0, flags=2 if self.normalize else 0, vmin=self.vmin, vmax=self.vmax)]


class FlowAdaptor(_ProgramAdaptor):
    def __init__(self, name: str, flow_mean: Union[Tuple[float, float], List[float]], flow_std: Union[Tuple[float, float], List[float]],
                 base_shape: Tuple[int, int], scale_strategy: str, output_normalized_coordinate: bool = False, *args, **kwargs):
        super().__init__(name, required_channels=2, *args, **kwargs)
        flow_mean = torch.tensor(list(flow_mean), dtype=torch.float32)
        flow_std = torch.tensor(list(flow_std), dtype=torch.float32)
        assert flow_mean.shape == (2,), f"Flow mean must be a 2D tensor, got {flow_mean.shape}"
        assert flow_std.shape == (2,), f"Flow std must be a 2D tensor, got {flow_std.shape}"
        self.register_buffer("flow_mean", flow_mean.view(1, 2, 1, 1))
        self.register_buffer("flow_std", flow_std.view(1, 2, 1, 1))
        self._mean_std = (tuple(float(v) for v in flow_mean), tuple(float(v) for v in flow_std))   # host copies: no sync per forward
        self.base_shape = list(base_shape)
        self.scale_strategy = scale_strategy
        self.output_normalized_coordinate = output_normalized_coordinate
        self.out_channels = 2

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        super()._load_from_state_dict(state_dict, prefix, *a, **k)
        self._mean_std = (tuple(float(v) for v in self.flow_mean.flatten()), tuple(float(v) for v in self.flow_std.flatten()))

    def _get_xy_scale(self, output_shape):
        if self.scale_strategy == "none":
            return 1.0, 1.0
        if self.scale_strategy == "scale_width":
            return output_shape[1] / self.base_shape[1], output_shape[1] / self.base_shape[1]
        if self.scale_strategy == "scale_height":
            return output_shape[0] / self.base_shape[0], output_shape[0] / self.base_shape[0]
        if self.scale_strategy == "scale_both":
            return output_shape[1] / self.base_shape[1], output_shape[0] / self.base_shape[0]
        raise ValueError(f"Invalid scaling strategy: {self.scale_strategy}")

    def segments(self, c0, o0, shape_hw):
        if self.output_normalized_coordinate:
            return [_seg(UC_AD_FLOWCOORD, c0, 2, o0, p=(shape_hw[1], shape_hw[0], 0, 0))]
        xs, ys = self._get_xy_scale(shape_hw)
        (mx, my), (sx, sy) = self._mean_std
        # reference order of operations: mean * scale and std * scale in fp32, then x * std + mean
        f32 = np.float32
        return [_seg(UC_AD_FLOW, c0, 2, o0, p=(f32(sx) * f32(xs), f32(mx) * f32(xs), f32(sy) * f32(ys), f32(my) * f32(ys)))]


class ConfidenceAdaptor(_ProgramAdaptor):
    def __init__(self, name: str, confidence_type: str, vmin: float, vmax: float, *args, **kwargs):
        super().__init__(name, required_channels=1, *args, **kwargs)
        self.confidence_type, self.vmin, self.vmax = confidence_type, vmin, vmax
        assert vmin < vmax, "vmin must be less than vmax"
        if confidence_type == "sigmoid":
            assert isfinite(vmin) and isfinite(vmax), "vmin and vmax must be finite for sigmoid confidence"
            assert vmin >= 0
        self.out_channels = 1

    def segments(self, c0, o0, shape_hw):
        if self.confidence_type == "exp":
            return [_seg(UC_AD_CONF_EXP, c0, 1, o0, vmin=self.vmin, vmax=self.vmax)]
        if self.confidence_type == "sigmoid":
            return [_seg(UC_AD_CONF_SIGMOID, c0, 1, o0, vmin=self.vmin, vmax=self.vmax)]
        raise engine.UcHipError(f"confidence_type '{self.confidence_type}' is not supported by the HIP adaptor kernel "
                                "(softmax confidence needs a reduction over the image)")


class MaskAdaptor(_ProgramAdaptor):
    def __init__(self, name: str, *args, **kwargs):
        super().__init__(name, required_channels=1, *args, **kwargs)
        self.out_channels = 2       # logits, mask

    def segments(self, c0, o0, shape_hw):
        return [_seg(UC_AD_MASK, c0, 1, o0)]

    def forward(self, adaptor_input: AdaptorInput):
        out = _run(adaptor_input, self.segments(0, 0, adaptor_input.output_shape_hw))
        return MaskAdaptorOutput(logits=_bchw(out, 0, 1), mask=_bchw(out, 1, 2))


class Covariance2DAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, parametrization: str = "exp_tanh", low_confidence_init: bool = False, *args, **kwargs):
        super().__init__(name, required_channels=3, *args, **kwargs)
        self.parametrization = parametrization
        self.low_confidence_init = low_confidence_init

    @staticmethod
    def _decode(x, offset):
        out = _program(x, [_seg(UC_AD_COV2D, 0, 3, 0, p=(offset, 0, 0, 0))], 7)
        return Covariance2DAdaptorOutput(covariance=_bchw(out, 0, 3), log_det=_bchw(out, 3, 4), inv_covariance=_bchw(out, 4, 7), log_representation=x)

    def forward(self, adaptor_input: AdaptorInput):
        if self.parametrization != "exp_tanh":
            raise ValueError(f"Invalid parametrization: {self.parametrization}")
        x = adaptor_input.adaptor_feature
        return self._decode(x, 8.0 if self.low_confidence_init else 0.0)

    @classmethod
    def decode(cls, x: torch.Tensor, representation: str):
        if representation != "exp_tanh":
            raise ValueError(f"Invalid parametrization: {representation}")
        return cls._decode(x, 8.0)


# ---- composites ---------------------------------------------------------------------------------------------------------
class _Concat(_ProgramAdaptor):
    "value = concatenation of the parts' values over consecutive channel groups (the reference's `...Plus...` adaptors)."

    def _init_parts(self, name, parts, *args, **kwargs):
        UniCeptionAdaptorBase.__init__(self, name, required_channels=sum(p.required_channels for _, p in parts), *args, **kwargs)
        for attr, p in parts:
            setattr(self, attr, p)
        self._part_attrs = [a for a, _ in parts]
        self.out_channels = sum(p.out_channels for _, p in parts)

    def segments(self, c0, o0, shape_hw):
        segs = []
        for a in self._part_attrs:
            p = getattr(self, a)
            segs += p.segments(c0, o0, shape_hw)
            c0 += p.required_channels
            o0 += p.out_channels
        return segs


class ValueWithConfidenceAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, value_adaptor: UniCeptionAdaptorBase, confidence_adaptor: UniCeptionAdaptorBase, *args, **kwargs):
        super().__init__(name, required_channels=value_adaptor.required_channels + confidence_adaptor.required_channels, *args, **kwargs)
        self.value_adaptor = value_adaptor
        self.confidence_adaptor = confidence_adaptor

    def _dust3r_fast_path(self):
        va, ca = self.value_adaptor, self.confidence_adaptor
        return (type(va) is PointMapAdaptor and type(ca) is ConfidenceAdaptor and va.mode == "exp" and va.no_bounds
                and ca.confidence_type == "exp")

    def forward(self, adaptor_input: AdaptorInput):
        va, ca = self.value_adaptor, self.confidence_adaptor
        x = adaptor_input.adaptor_feature
        assert x.shape[1] == self.required_channels, f"{type(self).__name__} needs {self.required_channels} channels, got {x.shape[1]}"
        if self._dust3r_fast_path():      # ONE dedicated kernel, with a HIP backward
            if autograd.grad_needed(x):
                pts, conf = autograd.pointmap_adaptor(_as_f32_map(x), float(ca.vmin), float(ca.vmax))
            else:
                pts, conf = ops.pointmap_adaptor(_as_f32_map(x), float(ca.vmin), float(ca.vmax))
            return RegressionWithConfidenceAdaptorOutput(value=pts.permute(0, 3, 1, 2), confidence=conf.permute(0, 3, 1, 2))
        nv = va.out_channels
        out = _run(adaptor_input, va.segments(0, 0, adaptor_input.output_shape_hw)
                   + ca.segments(va.required_channels, nv, adaptor_input.output_shape_hw))
        return RegressionWithConfidenceAdaptorOutput(value=_bchw(out, 0, nv), confidence=_bchw(out, nv, nv + 1))


class ValueWithMaskAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, value_adaptor: UniCeptionAdaptorBase, mask_adaptor: UniCeptionAdaptorBase, *args, **kwargs):
        super().__init__(name, required_channels=value_adaptor.required_channels + mask_adaptor.required_channels, *args, **kwargs)
        self.value_adaptor = value_adaptor
        self.mask_adaptor = mask_adaptor

    def forward(self, adaptor_input: AdaptorInput):
        va, ma = self.value_adaptor, self.mask_adaptor
        nv = va.out_channels
        hw = adaptor_input.output_shape_hw
        out = _run(adaptor_input, va.segments(0, 0, hw) + ma.segments(va.required_channels, nv, hw))
        return RegressionWithMaskAdaptorOutput(value=_bchw(out, 0, nv), logits=_bchw(out, nv, nv + 1), mask=_bchw(out, nv + 1, nv + 2))


class ValueWithConfidenceAndMaskAdaptor(UniCeptionAdaptorBase):
    def __init__(self, name: str, value_adaptor: UniCeptionAdaptorBase, confidence_adaptor: UniCeptionAdaptorBase,
                 mask_adaptor: UniCeptionAdaptorBase, *args, **kwargs):
        super().__init__(name, required_channels=value_adaptor.required_channels + confidence_adaptor.required_channels
                         + mask_adaptor.required_channels, *args, **kwargs)
        self.value_adaptor = value_adaptor
        self.confidence_adaptor = confidence_adaptor
        self.mask_adaptor = mask_adaptor

    def forward(self, adaptor_input: AdaptorInput):
        va, ca, ma = self.value_adaptor, self.confidence_adaptor, self.mask_adaptor
        nv = va.out_channels
        hw = adaptor_input.output_shape_hw
        out = _run(adaptor_input, va.segments(0, 0, hw) + ca.segments(va.required_channels, nv, hw)
                   + ma.segments(va.required_channels + 1, nv + 1, hw))
        return RegressionWithConfidenceAndMaskAdaptorOutput(value=_bchw(out, 0, nv), confidence=_bchw(out, nv, nv + 1),
                                                            logits=_bchw(out, nv + 1, nv + 2), mask=_bchw(out, nv + 2, nv + 3))


class FlowWithConfidenceAdaptor(ValueWithConfidenceAdaptor):
    def __init__(self, name: str, flow_mean, flow_std, base_shape, scale_strategy: str, output_normalized_coordinate: bool,
                 confidence_type: str, vmin: float, vmax: float, *args, **kwargs):
        super().__init__(name, value_adaptor=FlowAdaptor(f"{name}", flow_mean, flow_std, base_shape, scale_strategy, output_normalized_coordinate),
                         confidence_adaptor=ConfidenceAdaptor(f"{name}_confidence", confidence_type, vmin, vmax), *args, **kwargs)


# parameter groups of the named composites (prefix, parameter suffixes, part class, attribute name the reference uses)
_GROUPS = {
    "pointmap": (("mode", "vmin", "vmax"), PointMapAdaptor, "pointmap_adaptor"),
    "ray_origins": (("mode", "vmin", "vmax"), RayOriginsAdaptor, "ray_origins_adaptor"),
    "ray_directions": (("mode", "normalize_to_unit_sphere", "normalize_to_unit_image_plane", "vmin", "vmax", "clamp_min_of_z_dir", "z_dir_min"),
                       RayDirectionsAdaptor, "ray_directions_adaptor"),
    "depth": (("mode", "vmin", "vmax"), DepthAdaptor, "depth_adaptor"),
    "scene_flow": (("mode", "vmin", "vmax"), SceneFlowAdaptor, "scene_flow_adaptor"),
    "cam_trans": (("mode", "vmin", "vmax"), CamTranslationAdaptor, "cam_trans_adaptor"),
    "quaternions": (("mode", "normalize", "vmin", "vmax"), QuaternionsAdaptor, "quaternions_adaptor"),
}
_CONF_PARAMS = ("confidence_type", "confidence_vmin", "confidence_vmax")


def _bind(cls_name, names, args, kwargs):
    if len(args) > len(names):
        raise TypeError(f"{cls_name}() takes {len(names) + 1} positional arguments but {len(args) + 1} were given")
    vals = dict(zip(names, args))
    for n in names[len(args):]:
        if n not in kwargs:
            raise TypeError(f"{cls_name}() missing required argument: '{n}'")
        vals[n] = kwargs.pop(n)
    return vals


def _make_value(cls_name, groups, output_cls):
    names = [f"{g}_{s}" for g in groups for s in _GROUPS[g][0]]

    def __init__(self, name, *args, **kwargs):
        vals = _bind(cls_name, names, args, kwargs)
        parts = [(_GROUPS[g][2], _GROUPS[g][1](name, *[vals[f"{g}_{s}"] for s in _GROUPS[g][0]])) for g in groups]
        self._init_parts(name, parts, **kwargs)

    return type(cls_name, (_Concat,), {"__init__": __init__, "_output_cls": output_cls, "_param_names": tuple(names),
                                       "__doc__": f"value = concat({', '.join(groups)}) — same constructor parameters as the reference's {cls_name}."})


def _value_part(name, groups, vals):
    if groups == ("pointmap",):
        return PointMapAdaptor(f"{name}", *[vals[f"pointmap_{s}"] for s in _GROUPS["pointmap"][0]])
    cls = _VALUE_CLASSES[groups]
    return cls(name, *[vals[n] for n in cls._param_names])


def _make_wrapped(cls_name, groups, conf, mask):
    names = [f"{g}_{s}" for g in groups for s in _GROUPS[g][0]] + (list(_CONF_PARAMS) if conf else [])
    base = ValueWithConfidenceAndMaskAdaptor if (conf and mask) else (ValueWithConfidenceAdaptor if conf else ValueWithMaskAdaptor)

    def __init__(self, name, *args, **kwargs):
        vals = _bind(cls_name, names, args, kwargs)
        kw = {"value_adaptor": _value_part(name, groups, vals)}
        if conf:
            kw["confidence_adaptor"] = ConfidenceAdaptor(f"{name}_confidence", vals["confidence_type"], vals["confidence_vmin"], vals["confidence_vmax"])
        if mask:
            kw["mask_adaptor"] = MaskAdaptor(f"{name}_mask")
        base.__init__(self, name, **kw, **kwargs)

    return type(cls_name, (base,), {"__init__": __init__, "_param_names": tuple(names),
                                    "__doc__": f"Same constructor parameters as the reference's {cls_name}."})


_VALUE_CLASSES = {}
for _n, _g, _o in (("RayDirectionsPlusDepthAdaptor", ("ray_directions", "depth"), RegressionAdaptorOutput),
                   ("RayDirectionsPlusDepthPlusSceneFlowAdaptor", ("ray_directions", "depth", "scene_flow"), RegressionAdaptorOutput),
                   ("CamTranslationPlusQuatsAdaptor", ("cam_trans", "quaternions"), AdaptorOutput),
                   ("RayMapAdaptor", ("ray_origins", "ray_directions"), RegressionAdaptorOutput),
                   ("RayMapPlusDepthAdaptor", ("ray_origins", "ray_directions", "depth"), RegressionAdaptorOutput),
                   ("RayMapPlusDepthPlusQuatsAdaptor", ("ray_origins", "ray_directions", "depth", "quaternions"), RegressionAdaptorOutput),
                   ("PointMapPlusRayDirectionsPlusDepthAdaptor", ("pointmap", "ray_directions", "depth"), RegressionAdaptorOutput)):
    _VALUE_CLASSES[_g] = globals()[_n] = _make_value(_n, _g, _o)

for _stem, _g in (("PointMap", ("pointmap",)),
                  ("PointMapPlusRayDirectionsPlusDepth", ("pointmap", "ray_directions", "depth")),
                  ("RayDirectionsPlusDepth", ("ray_directions", "depth")),
                  ("RayDirectionsPlusDepthPlusSceneFlow", ("ray_directions", "depth", "scene_flow")),
                  ("RayMapPlusDepth", ("ray_origins", "ray_directions", "depth")),
                  ("RayMapPlusDepthPlusQuats", ("ray_origins", "ray_directions", "depth", "quaternions"))):
    for _suffix, _c, _m in (("WithConfidenceAdaptor", True, False), ("WithMaskAdaptor", False, True), ("WithConfidenceAndMaskAdaptor", True, True)):
        globals()[_stem + _suffix] = _make_wrapped(_stem + _suffix, _g, _c, _m)
del _n, _g, _o, _stem, _suffix, _c, _m
