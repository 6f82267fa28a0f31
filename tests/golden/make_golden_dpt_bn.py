"""Golden vectors of the DPT head with `use_bn=True` from the REAL reference (same recipe as make_golden_heads.py), eval mode:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_dpt_bn.py

Writes tests/golden/dpt_bn.npz: outputs of DPTFeatureDoubleUpsampling(use_bn=True) (BatchNorm after both convolutions of every residual
conv unit, libs/croco/dpt_block.py:125-176) and DPTSegmentationProcessor(use_bn=True) (prediction_heads/dpt.py:346); weights from the
name-keyed filler + heads_cases.bn_buffers_, inputs from seeds.  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.heads_cases import DPT_BN_DOUBLE, DPT_BN_SEG, bn_buffers_  # noqa: E402

from uniception.models.prediction_heads.base import PredictionHeadLayeredInput  # noqa: E402
from uniception.models.prediction_heads.dpt import DPTFeatureDoubleUpsampling, DPTFeatureInput, DPTSegmentationProcessor  # noqa: E402


def main():
    store = {}
    c = DPT_BN_SEG
    seg = DPTSegmentationProcessor(c["input_feature_dim"], c["output_dim"], hidden_dim=c["hidden_dim"], use_bn=True).eval()
    O.fill_state_dict_(seg.state_dict())
    bn_buffers_(seg.state_dict())
    g = torch.Generator().manual_seed(51)
    x = torch.randn(c["B"], c["input_feature_dim"], *c["feat_hw"], generator=g)
    with torch.no_grad():
        store["dpt_seg_bn/out"] = seg(DPTFeatureInput(features_upsampled_8x=x, target_output_shape=c["target"])).decoded_channels.numpy()
    c = DPT_BN_DOUBLE
    dbl = DPTFeatureDoubleUpsampling(input_feature_dims=c["input_feature_dims"], layer_dims=c["layer_dims"], feature_dim=c["feature_dim"],
                                     use_bn=True).eval()
    O.fill_state_dict_(dbl.state_dict())
    bn_buffers_(dbl.state_dict())
    assert any(k.endswith("bn1.running_var") for k in dbl.state_dict())
    g = torch.Generator().manual_seed(52)
    feats = [torch.randn(c["B"], d, *c["grid"], generator=g) for d in c["input_feature_dims"]]
    with torch.no_grad():
        store["dpt_double_bn/out"] = dbl(PredictionHeadLayeredInput(list_features=feats, target_output_shape=(80, 112))).features_upsampled_8x.numpy()
    print("dpt_seg_bn", store["dpt_seg_bn/out"].shape, float(np.abs(store["dpt_seg_bn/out"]).mean()),
          "dpt_double_bn", store["dpt_double_bn/out"].shape, float(np.abs(store["dpt_double_bn/out"]).mean()))
    np.savez_compressed(os.path.join(HERE, "dpt_bn.npz"), **store)
    print("wrote dpt_bn.npz")


if __name__ == "__main__":
    main()
