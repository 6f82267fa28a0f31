"""Encoder factory, restricted to the encoders on the DUSt3R path (reference: encoders/__init__.py:32-117)."""
from .base import (EncoderGlobalRepInput, EncoderGlobalRepOutput, EncoderInput, EncoderOutput, UniCeptionEncoderBase,  # noqa: F401
                   UniCeptionViTEncoderBase, ViTEncoderInput, ViTEncoderNonImageInput, ViTEncoderOutput)
from .croco import CroCoEncoder, CroCoIntermediateFeatureReturner
from .dinov2 import DINOv2Encoder, DINOv2IntermediateFeatureReturner
from .image_normalizations import IMAGE_NORMALIZATION_DICT  # noqa: F401

ENCODER_CONFIGS = {
    "croco": {"class": CroCoEncoder, "intermediate_feature_returner_class": CroCoIntermediateFeatureReturner,
              "supported_models": ["CroCov2", "DUSt3R", "MASt3R"]},
    "dinov2": {"class": DINOv2Encoder, "intermediate_feature_returner_class": DINOv2IntermediateFeatureReturner,
               "supported_models": ["DINOv2", "DINOv2-Registers", "DINOv2-Depth-Anythingv2"]},
}


def encoder_factory(encoder_str: str, **kwargs):
    "encoder_factory('croco', name=..., data_norm_type=..., ...) -> CroCoEncoder"
    if encoder_str not in ENCODER_CONFIGS:
        raise ValueError(f"Unknown encoder: {encoder_str}. For valid encoder_str options, please use print_available_encoder_models()")
    return ENCODER_CONFIGS[encoder_str]["class"](**kwargs)


def feature_returner_encoder_factory(encoder_str: str, **kwargs):
    if encoder_str not in ENCODER_CONFIGS:
        raise ValueError(f"Unknown encoder: {encoder_str}. For valid encoder_str options, please use print_available_encoder_models()")
    return ENCODER_CONFIGS[encoder_str]["intermediate_feature_returner_class"](**kwargs)


def get_available_encoders():
    return list(ENCODER_CONFIGS.keys())


def print_available_encoder_models():
    for name, cfg in ENCODER_CONFIGS.items():
        print(f"{name}: {', '.join(cfg['supported_models'])}")
