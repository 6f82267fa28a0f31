"""Oracle outputs of the two full-size pipelines (BASELINE configs[3] and [4]) on a pixel sub-grid: tests/golden/fullsize.npz.

    python tests/golden/make_golden_fullsize.py          (CPU, ~5 minutes; needs nothing but the repository)

The GPU tests used to run the oracle at full size themselves (tests/test_config5_gpu.py: ~80 s of a 106-s test for the 1024 x 1024 pair,
tests/test_dinov2_gpu.py: ~40 s for the 518 x 518 DINOv2 pipeline) — a fifth of the GPU suite's wall time spent on the host.  The
oracle's results are deterministic (name-keyed filler weights, seeded images), so they are computed HERE once and kept as data: every
16th (config 4: 1024^2) / 7th (config 3: 518^2) pixel of the four outputs, compared by the tests against the same sub-grid of the HIP
outputs (relative L2 over ~4 k / 5.5 k pixels per output, and max-abs where the gate has one).  The oracle itself stays pinned to the
reference by make_golden*.py (imported reference) and make_golden_dinov2_hf.py (transformers)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.cases import GAINS  # noqa: E402

C5_STEP, C3_STEP = 16, 7


def config5():
    from uniception_amd.models.factory import DUSt3R
    model = DUSt3R(name="c5", img_size=(1024, 1024), pred_head_type="linear").eval()
    O.fill_state_dict_(model.state_dict(), gains=GAINS)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img1, img2 = O.make_images(21, 1, 1024, 1024)
    with torch.no_grad():
        o1, o2 = O.dust3r_forward(sd, img1, img2, head="linear", enc_depth=24, dec_depth=12)
    s = C5_STEP
    return {"c5_pts3d_1": o1["pts3d"][:, ::s, ::s], "c5_conf_1": o1["conf"][:, ::s, ::s],
            "c5_pts3d_2": o2["pts3d_in_other_view"][:, ::s, ::s], "c5_conf_2": o2["conf"][:, ::s, ::s]}


def config3():
    from tests.golden.dinov2_cases import GAINS as GD      # (pos_embed / cls_token / register_tokens gains of the DINOv2 filler)
    from uniception_amd.models.encoders import encoder_factory
    from uniception_amd.models.factory import DUSt3R
    torch.manual_seed(0)
    model = DUSt3R(name="c3", img_size=(518, 518), pred_head_type="dpt")
    model.encoder = encoder_factory("dinov2", name="c3_dinov2", size="large")
    model = model.eval()
    O.fill_state_dict_(model.state_dict(), gains=dict(GD, **GAINS))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(33)
    img1, img2 = torch.randn(1, 3, 518, 518, generator=g), torch.randn(1, 3, 518, 518, generator=g)
    out = {}
    with torch.no_grad():
        enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        feats, _ = O.dinov2_encoder(torch.cat([img1, img2], 0), enc_sd, "model.", num_heads=16)
        final, taken = O.cross_attention_transformer([feats[:1], feats[1:]], sd, "info_sharing.", depth=12, num_heads=12, indices=(5, 8),
                                                     norm_intermediate=False)
        for v in range(2):
            up8 = O.dpt_feature([feats[v:v + 1], taken[0][v], taken[1][v], final[v]], sd, f"dpt_feature_head{v + 1}.")
            pts, conf = O.pointmap_adaptor(O.dpt_regressor(up8, (518, 518), sd, f"dpt_regressor_head{v + 1}."))
            s = C3_STEP
            out[f"c3_pts3d_{v + 1}"] = pts.permute(0, 2, 3, 1)[:, ::s, ::s]
            out[f"c3_conf_{v + 1}"] = conf.permute(0, 2, 3, 1)[:, ::s, ::s]
    return out


if __name__ == "__main__":
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize.npz")
    only = sys.argv[1:]
    d = {k: torch.from_numpy(v) for k, v in np.load(path).items() if v.shape != ()} if (only and os.path.exists(path)) else {}
    if not only or "c5" in only:
        d.update(config5())
    if not only or "c3" in only:
        d.update(config3())
    d = {k: v.contiguous().numpy().astype(np.float32) for k, v in d.items()}
    d["c5_step"], d["c3_step"] = np.int32(C5_STEP), np.int32(C3_STEP)
    np.savez_compressed(path, **d)
    print(path, {k: getattr(v, "shape", v) for k, v in d.items()}, os.path.getsize(path), "bytes")
