"""Gradient goldens of the adaptors from the REAL reference's autograd (same recipe as make_golden_heads.py):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_heads_grads.py

Writes tests/golden/heads_extra_grads.npz: for every case of tests/golden/heads_cases.py the gradient, with respect to the decoded
channels, of loss = sum over the adaptor's output fields of (field * weight).sum() with the seeded weights of
heads_cases.adaptor_grad_weight (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.heads_cases import AD_H, AD_W, ADAPTOR_CASES, OUT_FIELDS, adaptor_grad_weight, adaptor_input  # noqa: E402

from uniception.models.prediction_heads import adaptors as RA  # noqa: E402
from uniception.models.prediction_heads.base import AdaptorInput  # noqa: E402


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
    np.savez_compressed(os.path.join(HERE, "heads_extra_grads.npz"), **store)
    print("wrote", os.path.join(HERE, "heads_extra_grads.npz"))


if __name__ == "__main__":
    main()
