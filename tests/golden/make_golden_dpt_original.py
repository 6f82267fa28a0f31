"""Numeric golden for the DPT half of the checkpoint converter, from the REAL reference (build container only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_dpt_original.py

The reference still ships the module the ORIGINAL DUSt3R checkpoints were trained with — `DPTOutputAdapter`
(uniception/models/libs/croco/dpt_block.py:326-530: `act_postprocess.*`, `scratch.layerK_rn`, `scratch.refinenetK.*`, `head.{0,2,4}`).
This script builds it (regression head, 4 output channels, hooks 0-3) with name-keyed filler weights, runs it on seeded token lists
and writes tests/golden/dpt_original.npz: the weights under their ORIGINAL checkpoint names (`downstream_head1.dpt.<key>`), the input
tokens and the module's output.  tests/test_convert_checkpoint_gpu.py puts exactly these tensors through
uniception_amd/tools/convert_checkpoint.py into `DPTFeature + DPTRegressionProcessor` and must reproduce the output: the key map
`downstream_headK.dpt.* -> dpt_feature_headK.* / dpt_regressor_headK.*` (convert_dust3r_weights_to_uniception.py:20-122) is then
pinned by numbers, not by its own inverse.  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.cases import GAINS  # noqa: E402
from tests.golden.dpt_original_case import DPT_ORIGINAL as C, dpt_original_tokens  # noqa: E402

from uniception.models.libs.croco.dpt_block import DPTOutputAdapter  # noqa: E402


def main():
    torch.manual_seed(0)
    m = DPTOutputAdapter(num_channels=4, stride_level=1, patch_size=C["patch"], hooks=[0, 1, 2, 3], layer_dims=list(C["layer_dims"]),
                         feature_dim=C["feature_dim"], last_dim=C["feature_dim"] // 2, dim_tokens_enc=list(C["token_dims"]),
                         head_type="regression").eval()
    # what the dust3r code base does to this module before training (its checkpoints carry no act_N_postprocess.* entries)
    for i in (1, 2, 3, 4):
        delattr(m, f"act_{i}_postprocess")
    sd = m.state_dict()
    O.fill_state_dict_(sd, gain=1.0, gains={"head.4.weight": GAINS["conv2.2.weight"]})
    tokens = dpt_original_tokens()
    with torch.no_grad():
        out = m(tokens, image_size=C["img"])
    store = {"out": out.numpy()}
    for k, v in m.state_dict().items():
        store["w/downstream_head1.dpt." + k] = v.detach().numpy()
    for i, t in enumerate(tokens):
        store[f"tokens{i}"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "dpt_original.npz"), **store)
    print("out", tuple(out.shape), float(out.abs().mean()), "keys", len(sd))
    print(sorted(k for k in sd)[:12], "...")


if __name__ == "__main__":
    main()
