"""Config 5 at FULL SIZE from the REAL reference (tests/golden/fullsize.npz held oracle outputs only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_fullsize_ref.py

Runs the reference's own factory model (uniception.models.factory.dust3r.DUSt3R: ViT-L/16 encoder, 12-block decoder, linear head) on the
1024 x 1024 pair of make_golden_fullsize.py — same name-keyed filler weights, same seeded images — on the CPU (a few minutes), asserts
that the oracle values stored in fullsize.npz agree with it on the stored pixel sub-grid (every 16th pixel) to < 2e-5, and writes the
REFERENCE's values next to them as `c5ref_*` (tests/golden/fullsize_ref.npz, data only): tests/test_config5_gpu.py compares the HIP
outputs with the reference's numbers.  (Config 3's DINOv2 encoder is torch.hub code that is absent from /root/reference: it stays pinned
through transformers, make_golden_dinov2_hf.py.)"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.cases import GAINS  # noqa: E402
from tests.golden.make_golden_fullsize import C5_STEP  # noqa: E402

from uniception.models.factory.dust3r import DUSt3R  # noqa: E402


def main():
    t0 = time.time()
    torch.manual_seed(0)
    model = DUSt3R(name="c5", img_size=(1024, 1024), pred_head_type="linear").eval()
    O.fill_state_dict_(model.state_dict(), gains=GAINS)
    img1, img2 = O.make_images(21, 1, 1024, 1024)
    v1 = {"img": img1, "instance": ["0"], "data_norm_type": "dust3r"}
    v2 = {"img": img2, "instance": ["100"], "data_norm_type": "dust3r"}
    with torch.no_grad():
        r1, r2 = model(v1, v2)
    s = C5_STEP
    ref = {"c5ref_pts3d_1": r1["pts3d"][:, ::s, ::s], "c5ref_conf_1": r1["conf"][:, ::s, ::s],
           "c5ref_pts3d_2": r2["pts3d_in_other_view"][:, ::s, ::s], "c5ref_conf_2": r2["conf"][:, ::s, ::s]}
    z = np.load(os.path.join(HERE, "fullsize.npz"))
    worst = 0.0
    for k, v in ref.items():
        o = torch.from_numpy(z[k.replace("c5ref_", "c5_")])
        e = float((o - v).norm() / v.norm())
        worst = max(worst, e)
        print(k, tuple(v.shape), f"oracle (fullsize.npz) vs reference rel-L2 {e:.2e}")
    assert worst < 2e-5, worst
    np.savez_compressed(os.path.join(HERE, "fullsize_ref.npz"), **{k: v.numpy() for k, v in ref.items()})
    print(f"wrote fullsize_ref.npz in {time.time() - t0:.0f} s; oracle == reference to {worst:.2e} at 1024 x 1024")


if __name__ == "__main__":
    main()
