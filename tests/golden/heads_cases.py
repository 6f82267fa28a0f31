"""Cases of the widened prediction-head surface: adaptors, DPTSegmentationProcessor, DPTFeatureDoubleUpsampling —
shared by the golden generator (real reference) and the GPU tests."""
import numpy as np

INF = float("inf")
RD = ("linear", True, False, -INF, INF, True, 0.5)          # ray directions: unit sphere, z clamped from below
RD_PLANE = ("linear", False, True, -2.0, 2.0, False, 1.0)    # unit image plane, clipped
# name -> (class name, positional constructor arguments after `name`, input channels)
ADAPTOR_CASES = {
    "flow_none": ("FlowAdaptor", ((0.3, -0.2), (2.0, 3.0), (8, 16), "none", False), 2),
    "flow_both": ("FlowAdaptor", ((0.3, -0.2), (2.0, 3.0), (8, 16), "scale_both", False), 2),
    "flow_width": ("FlowAdaptor", ((0.3, -0.2), (2.0, 3.0), (8, 16), "scale_width", False), 2),
    "flow_coord": ("FlowAdaptor", ((0.0, 0.0), (1.0, 1.0), (8, 16), "none", True), 2),
    "scale_exp": ("ScaleAdaptor", ("exp", 0, 3.0), 1),
    "depth_square": ("DepthAdaptor", ("square", 0, 2.5), 1),
    "depth_exp": ("DepthAdaptor", ("exp", 0, INF), 1),
    "sceneflow_lin": ("SceneFlowAdaptor", ("linear", -1.0, 1.0), 3),
    "pointmap_square": ("PointMapAdaptor", ("square", -INF, INF), 3),
    "pointmap_exp_clip": ("PointMapAdaptor", ("exp", -1.5, 1.5), 3),
    "pointmap_zexp": ("PointMapAdaptor", ("z_exp", -INF, INF), 3),
    "pointmap_linear": ("PointMapAdaptor", ("linear", -0.5, 0.5), 3),
    "rayorigins_exp": ("RayOriginsAdaptor", ("exp", -INF, INF), 3),
    "raydirs_sphere": ("RayDirectionsAdaptor", RD, 3),
    "raydirs_plane": ("RayDirectionsAdaptor", RD_PLANE, 3),
    "camtrans_square": ("CamTranslationAdaptor", ("square", -INF, INF), 3),
    "quats_norm": ("QuaternionsAdaptor", ("linear", True, -INF, INF), 4),
    "conf_exp": ("ConfidenceAdaptor", ("exp", 1, 20.0), 1),
    "conf_sigmoid": ("ConfidenceAdaptor", ("sigmoid", 0.0, 5.0), 1),
    "mask": ("MaskAdaptor", (), 1),
    "cov2d": ("Covariance2DAdaptor", ("exp_tanh", True), 3),
    "raydirs_depth": ("RayDirectionsPlusDepthAdaptor", RD + ("exp", 0, INF), 4),
    "raymap_depth_quats": ("RayMapPlusDepthPlusQuatsAdaptor", ("exp", -INF, INF) + RD + ("square", 0, INF) + ("linear", True, -INF, INF), 11),
    "camtrans_quats": ("CamTranslationPlusQuatsAdaptor", ("exp", -INF, INF, "linear", True, -INF, INF), 7),
    "flow_conf": ("FlowWithConfidenceAdaptor", ((0.1, 0.2), (1.5, 2.5), (8, 16), "scale_height", False, "exp", 1, INF), 3),
    "pointmap_conf_dust3r": ("PointMapWithConfidenceAdaptor", ("exp", -INF, INF, "exp", 1, INF), 4),
    "pointmap_conf_sigmoid": ("PointMapWithConfidenceAdaptor", ("square", -3.0, 3.0, "sigmoid", 0.0, 1.0), 4),
    "pm_rd_depth_conf": ("PointMapPlusRayDirectionsPlusDepthWithConfidenceAdaptor", ("exp", -INF, INF) + RD + ("exp", 0, INF) + ("exp", 1, INF), 8),
    "raymap_depth_mask": ("RayMapPlusDepthWithMaskAdaptor", ("linear", -INF, INF) + RD_PLANE + ("linear", 0, INF), 8),
    "pointmap_conf_mask": ("PointMapWithConfidenceAndMaskAdaptor", ("exp", -INF, INF, "exp", 1, INF), 5),
    "rd_depth_flow_conf_mask": ("RayDirectionsPlusDepthPlusSceneFlowWithConfidenceAndMaskAdaptor", RD + ("exp", 0, INF) + ("linear", -INF, INF) + ("sigmoid", 0.0, 2.0), 9),
}
# the pose heads hand [B, C] vectors to their adaptors (reference pose_head.py:157-179): name -> case of ADAPTOR_CASES run on a 2-D input
ADAPTOR_CASES_2D = {"camtrans_quats_2d": "camtrans_quats", "camtrans_2d": "camtrans_square", "quats_2d": "quats_norm", "scale_2d": "scale_exp"}
AD_B, AD_H, AD_W = 2, 5, 7
OUT_FIELDS = ("value", "confidence", "logits", "mask", "covariance", "log_det", "inv_covariance")


def adaptor_input(name):
    import torch
    g = torch.Generator().manual_seed(sum(map(ord, name)) + 7)
    return torch.randn(AD_B, ADAPTOR_CASES[name][2], AD_H, AD_W, generator=g)


def adaptor_input_2d(name):
    import torch
    g = torch.Generator().manual_seed(sum(map(ord, name)) + 11)
    return torch.randn(3, ADAPTOR_CASES[ADAPTOR_CASES_2D[name]][2], generator=g)


# DPT extras: (feature dims of the layered inputs, layer dims, feature dim, token grid, target shape, seg classes)
DPT_SEG = dict(input_feature_dim=32, output_dim=5, hidden_dim=16, feat_hw=(24, 40), target=(35, 61), B=2)
DPT_DOUBLE = dict(input_feature_dims=[128, 192], layer_dims=[32, 64], feature_dim=32, grid=(5, 7), B=2)


def adaptor_grad_weight(name, field, shape):
    "Weights of the scalar the gradient goldens differentiate: loss = sum over output fields of (field * weight).sum()."
    import torch
    g = torch.Generator().manual_seed(sum(map(ord, name + "/" + field)) + 1013)
    return torch.randn(*shape, generator=g)


def dpt_grad_weight(name, shape):
    import torch
    g = torch.Generator().manual_seed(sum(map(ord, name)) + 2027)
    return torch.randn(*shape, generator=g)


# use_bn=True (round 6, tests/golden/dpt_bn.npz from make_golden_dpt_bn.py): eval-mode BatchNorm inside the DPT head's residual conv
# units and the segmentation processor
DPT_BN_DOUBLE = dict(input_feature_dims=[128, 192], layer_dims=[32, 64], feature_dim=32, grid=(5, 7), B=2)
DPT_BN_SEG = dict(input_feature_dim=32, output_dim=5, hidden_dim=16, feat_hw=(24, 40), target=(35, 61), B=2)


def bn_buffers_(sd):
    "After the name-keyed filler: make every BatchNorm's running statistics those of a plausible trained layer (positive variances)."
    for k, v in sd.items():
        if k.endswith("running_var"):
            v.copy_(v.abs() * 0.5 + 0.25)
        elif k.endswith("running_mean"):
            v.mul_(0.3)
        elif k.endswith("num_batches_tracked"):
            v.fill_(100)
