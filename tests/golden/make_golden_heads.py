"""Golden vectors of the widened prediction-head surface from the REAL reference (same recipe as make_golden.py):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_heads.py

Writes tests/golden/heads_extra.npz: adaptor outputs for every case of tests/golden/heads_cases.py, DPTSegmentationProcessor and
DPTFeatureDoubleUpsampling outputs (weights from the name-keyed filler, inputs from seeds; data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.heads_cases import (AD_H, AD_W, ADAPTOR_CASES, ADAPTOR_CASES_2D, DPT_DOUBLE, DPT_SEG, OUT_FIELDS, adaptor_input,  # noqa: E402
                                      adaptor_input_2d)

from uniception.models.prediction_heads import adaptors as RA  # noqa: E402
from uniception.models.prediction_heads.base import AdaptorInput, PredictionHeadLayeredInput  # noqa: E402
from uniception.models.prediction_heads.dpt import DPTFeatureDoubleUpsampling, DPTFeatureInput, DPTSegmentationProcessor  # noqa: E402


def main():
    store = {}
    for name, (cls, args, _) in ADAPTOR_CASES.items():
        ad = getattr(RA, cls)(name, *args)
        with torch.no_grad():
            out = ad(AdaptorInput(adaptor_feature=adaptor_input(name), output_shape_hw=(AD_H, AD_W)))
        for f in OUT_FIELDS:
            if hasattr(out, f):
                store[f"ad/{name}/{f}"] = getattr(out, f).numpy()
        print(name, [k.split("/")[-1] for k in store if k.startswith(f"ad/{name}/")])
    for name, base in ADAPTOR_CASES_2D.items():      # [B, C] inputs (pose heads)
        cls, args, _ = ADAPTOR_CASES[base]
        with torch.no_grad():
            out = getattr(RA, cls)(name, *args)(AdaptorInput(adaptor_feature=adaptor_input_2d(name), output_shape_hw=(AD_H, AD_W)))
        store[f"ad2d/{name}/value"] = out.value.numpy()
        print(name, store[f"ad2d/{name}/value"].shape)
    c = DPT_SEG
    seg = DPTSegmentationProcessor(c["input_feature_dim"], c["output_dim"], hidden_dim=c["hidden_dim"]).eval()
    O.fill_state_dict_(seg.state_dict())
    g = torch.Generator().manual_seed(41)
    x = torch.randn(c["B"], c["input_feature_dim"], *c["feat_hw"], generator=g)
    with torch.no_grad():
        store["dpt_seg/out"] = seg(DPTFeatureInput(features_upsampled_8x=x, target_output_shape=c["target"])).decoded_channels.numpy()
    c = DPT_DOUBLE
    dbl = DPTFeatureDoubleUpsampling(input_feature_dims=c["input_feature_dims"], layer_dims=c["layer_dims"], feature_dim=c["feature_dim"]).eval()
    O.fill_state_dict_(dbl.state_dict())
    g = torch.Generator().manual_seed(42)
    feats = [torch.randn(c["B"], d, *c["grid"], generator=g) for d in c["input_feature_dims"]]
    with torch.no_grad():
        store["dpt_double/out"] = dbl(PredictionHeadLayeredInput(list_features=feats, target_output_shape=(80, 112))).features_upsampled_8x.numpy()
    print("dpt_seg", store["dpt_seg/out"].shape, "dpt_double", store["dpt_double/out"].shape)
    np.savez_compressed(os.path.join(HERE, "heads_extra.npz"), **store)
    print("wrote", os.path.join(HERE, "heads_extra.npz"))


if __name__ == "__main__":
    main()
