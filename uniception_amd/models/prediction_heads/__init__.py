"""Prediction heads of the pointmap path (reference: prediction_heads/__init__.py)."""
from .base import (AdaptorInput, AdaptorOutput, PixelTaskOutput, PredictionHeadInput, PredictionHeadLayeredInput,  # noqa: F401
                   PredictionHeadOutput, RegressionAdaptorOutput, RegressionWithConfidenceAdaptorOutput,
                   UniCeptionAdaptorBase, UniCeptionPredictionHeadBase)
