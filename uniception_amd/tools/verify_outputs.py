"""Real-weights numeric check of the two-view path — the counterpart of the reference's verification block
(examples/models/dust3r/dust3r.py:150-230): load a checkpoint into the DUSt3R factory model, run ONE symmetrized image pair
(view1 = [img0, img1], view2 = [img1, img0], dust3r.py:190-194) through the HIP path and compare the four head outputs with a
reference `.npz` holding `head1_pts3d`, `head2_pts3d`, `head1_conf`, `head2_conf` (the file the reference's own harness reads,
produced by the vanilla DUSt3R code).  Both of the reference's metrics, with its definitions and bars (dust3r.py:223-230):

    abs_error = max |x - y| < 1e-2      and      rel_error = ||x - y|| / ||x|| < 1e-3          (x: this build's output, y: the file's)

    python -m uniception_amd.tools.verify_outputs --checkpoint dust3r_512_dpt.pth --original --images pair.npz \\
           --reference DUSt3R_ViTLarge_BaseDecoder_512_dpt_head_output.npz --head dpt --img 512 [--precision fp32|bf16x3|bf16]

`--original`: the checkpoint is in the original CroCo / DUSt3R key layout (converted on the fly by tools/convert_checkpoint.py);
otherwise a UniCeption-layout state_dict ({"model": ...} or bare).  `--images`: an .npz with `img0`, `img1` as HxWx3 uint8 (or float in
[0, 1]) — normalised like the reference ((x - 0.5) / 0.5, dust3r.py:184-187) — or already normalised 3xHxW float arrays.
The verification precision is fp32 ("fp32": exact kernels; "bf16x3": fp32-class arithmetic on the matrix pipe, meets the same
bars); "bf16" reports the performance mode's distance and is expected to miss the 1e-3 bar."""
import argparse
import sys
from typing import Dict, Tuple

import numpy as np
import torch

ABS_TOL, REL_TOL = 1e-2, 1e-3          # dust3r.py:230
KEYS = ("head1_pts3d", "head2_pts3d", "head1_conf", "head2_conf")


def abs_and_rel_error(x: np.ndarray, y: np.ndarray) -> Tuple[float, float]:
    "The reference's `compute_abs_and_rel_error` (dust3r.py:223): (max |x - y|, ||x - y|| / ||x||)."
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return float(np.abs(x - y).max()), float(np.linalg.norm(x - y) / np.linalg.norm(x))


def check_dict(res1: Dict, res2: Dict) -> Dict[str, np.ndarray]:
    "The four arrays the reference compares (dust3r.py:204-221), from the model's two result dicts."
    return {"head1_pts3d": res1["pts3d"].detach().float().cpu().numpy(),
            "head2_pts3d": res2["pts3d_in_other_view"].detach().float().cpu().numpy(),
            "head1_conf": res1["conf"].detach().squeeze(-1).float().cpu().numpy(),
            "head2_conf": res2["conf"].detach().squeeze(-1).float().cpu().numpy()}


def compare(outputs: Dict[str, np.ndarray], reference: Dict[str, np.ndarray], abs_tol: float = ABS_TOL, rel_tol: float = REL_TOL):
    """-> ({key: (abs_error, rel_error, ok)}, all_ok).  Shapes must agree (a [1, H, W] reference conf against [B, H, W] is an error,
    not a broadcast)."""
    report, all_ok = {}, True
    for k in KEYS:
        if k not in reference:
            raise KeyError(f"reference file has no '{k}' (found: {sorted(reference)})")
        x, y = outputs[k], np.asarray(reference[k])
        if x.shape != y.shape:
            raise ValueError(f"{k}: output shape {x.shape} vs reference {y.shape}")
        a, r = abs_and_rel_error(x, y)
        ok = a < abs_tol and r < rel_tol
        report[k] = (a, r, ok)
        all_ok &= ok
    return report, all_ok


def normalise_image(a: np.ndarray) -> torch.Tensor:
    "HxWx3 uint8 / float in [0, 1] -> 3xHxW in [-1, 1] ((x - 0.5) / 0.5, dust3r.py:184-187); 3xHxW float arrays pass through."
    a = np.asarray(a)
    if a.ndim == 3 and a.shape[0] == 3 and a.dtype != np.uint8:
        return torch.from_numpy(a.astype(np.float32))
    if a.ndim != 3 or a.shape[-1] < 3:
        raise ValueError(f"image array of shape {a.shape}: expected HxWx3 or 3xHxW")
    t = torch.from_numpy(a[..., :3].astype(np.float32))
    if a.dtype == np.uint8:
        t = t / 255
    return ((t - 0.5) / 0.5).permute(2, 0, 1).contiguous()


def symmetrized_views(img0: torch.Tensor, img1: torch.Tensor, device):
    img = torch.stack([img0, img1]).to(device)
    view1 = {"img": img, "instance": [0, 1], "data_norm_type": "dust3r"}
    view2 = {"img": img[[1, 0]].clone(), "instance": [1, 0], "data_norm_type": "dust3r"}
    return view1, view2


def run(model, view1, view2, precision: str = "fp32"):
    from .. import engine
    with torch.no_grad(), engine.precision(precision):
        return model(view1, view2)


def verify(model, img0: torch.Tensor, img1: torch.Tensor, reference: Dict[str, np.ndarray], precision: str = "fp32", device="cuda:0",
           out=sys.stdout):
    """Forward + both metrics on all four outputs; prints the reference harness's lines; returns (report, all_ok)."""
    view1, view2 = symmetrized_views(img0, img1, device)
    res1, res2 = run(model, view1, view2, precision)
    report, ok = compare(check_dict(res1, res2), reference)
    for k, (a, r, good) in report.items():
        print(f"{k} abs_error: {a}, rel_error: {r}" + ("" if good else "   <-- exceeds abs < 1e-2 and rel < 1e-3"), file=out)
    return report, ok


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--original", action="store_true", help="checkpoint is in the original CroCo / DUSt3R key layout")
    ap.add_argument("--images", required=True, help=".npz with img0, img1")
    ap.add_argument("--reference", required=True, help=".npz with head1_pts3d, head2_pts3d, head1_conf, head2_conf")
    ap.add_argument("--head", default="dpt", choices=["dpt", "linear"])
    ap.add_argument("--img", type=int, nargs="+", default=[512], help="model img_size: one value (square) or H W")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"])
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    from .. import _lib, engine
    from ..models.factory import DUSt3R
    from . import convert_checkpoint as cc
    _lib.load()
    size = (a.img[0], a.img[0]) if len(a.img) == 1 else (a.img[0], a.img[1])
    model = DUSt3R(name="verify", img_size=size, pred_head_type=a.head).eval()
    ckpt = torch.load(a.checkpoint, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    if a.original:
        cc.load_original_checkpoint(model, sd, strict=True)
    else:
        model.load_state_dict(sd, strict=True)
        engine.bump_weight_epoch()
    model = model.to(a.device)
    imgs = np.load(a.images)
    reference = dict(np.load(a.reference))
    print(f"===== Checking {a.checkpoint} ({a.head} head, {size[0]}x{size[1]}, {a.precision}) =====")
    _, ok = verify(model, normalise_image(imgs["img0"]), normalise_image(imgs["img1"]), reference, a.precision, a.device)
    print("PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
