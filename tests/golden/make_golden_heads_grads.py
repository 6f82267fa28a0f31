"""Gradient goldens of the adaptors from the REAL reference's autograd (same recipe as make_golden_heads.py):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_heads_grads.py

Writes tests/golden/heads_extra_grads.npz: for every case of tests/golden/heads_cases.py the gradient, with respect to the decoded
channels, of loss = sum over the adaptor's output fields of (field * weight).sum() with the seeded weights of
heads_cases.adaptor_grad_weight; and for DPTSegmentationProcessor (eval mode: Dropout is the identity) and
DPTFeatureDoubleUpsampling the gradients of (output * weight).sum() with respect to the inputs and every parameter (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.heads_cases import (AD_H, AD_W, ADAPTOR_CASES, DPT_DOUBLE, DPT_SEG, OUT_FIELDS, adaptor_grad_weight, adaptor_input,  # noqa: E402
                                      dpt_grad_weight)

from uniception.models.prediction_heads import adaptors as RA  # noqa: E402
from uniception.models.prediction_heads.base import AdaptorInput, PredictionHeadLayeredInput  # noqa: E402
from uniception.models.prediction_heads.dpt import DPTFeatureDoubleUpsampling, DPTFeatureInput, DPTSegmentationProcessor  # noqa: E402


def main():
    store = {}
    for name, (cls, args, _) in ADAPTOR_CASES.items():
        ad = getattr(RA, cls)(name, *args)
        x = adaptor_input(name).requires_grad_(True)
        out = ad(AdaptorInput(adaptor_feature=x, output_shape_hw=(AD_H, AD_W)))
        loss = 0.0
        for f in OUT_FIELDS:
            if hasattr(out, f):
                v = getattr(out, f)
                loss = loss + (v * adaptor_grad_weight(name, f, v.shape)).sum()
        loss.backward()
        store[f"ad/{name}/dx"] = x.grad.numpy()
        store[f"ad/{name}/loss"] = np.float64(loss.item())
        print(f"{name}: loss {loss.item():.6g}  |dx|max {x.grad.abs().max().item():.4g}")
    c = DPT_SEG
    seg = DPTSegmentationProcessor(c["input_feature_dim"], c["output_dim"], hidden_dim=c["hidden_dim"]).eval()
    O.fill_state_dict_(seg.state_dict())
    g = torch.Generator().manual_seed(41)
    x = torch.randn(c["B"], c["input_feature_dim"], *c["feat_hw"], generator=g).requires_grad_(True)
    out = seg(DPTFeatureInput(features_upsampled_8x=x, target_output_shape=c["target"])).decoded_channels
    (out * dpt_grad_weight("dpt_seg", out.shape)).sum().backward()
    store["dpt_seg/dx"] = x.grad.numpy()
    for k, p in seg.named_parameters():
        store[f"dpt_seg/param/{k}"] = p.grad.numpy()
    print("dpt_seg", {k: tuple(v.shape) for k, v in store.items() if k.startswith("dpt_seg/")})
    c = DPT_DOUBLE
    dbl = DPTFeatureDoubleUpsampling(input_feature_dims=c["input_feature_dims"], layer_dims=c["layer_dims"], feature_dim=c["feature_dim"]).eval()
    O.fill_state_dict_(dbl.state_dict())
    g = torch.Generator().manual_seed(42)
    feats = [torch.randn(c["B"], d, *c["grid"], generator=g).requires_grad_(True) for d in c["input_feature_dims"]]
    out = dbl(PredictionHeadLayeredInput(list_features=feats, target_output_shape=(80, 112))).features_upsampled_8x
    (out * dpt_grad_weight("dpt_double", out.shape)).sum().backward()
    for i, f in enumerate(feats):
        store[f"dpt_double/dx{i}"] = f.grad.numpy()
    for k, p in dbl.named_parameters():
        if p.grad is not None:
            store[f"dpt_double/param/{k}"] = p.grad.numpy()
    print("dpt_double", sorted(k for k in store if k.startswith("dpt_double/")))
    np.savez_compressed(os.path.join(HERE, "heads_extra_grads.npz"), **store)
    print("wrote", os.path.join(HERE, "heads_extra_grads.npz"))


if __name__ == "__main__":
    main()
