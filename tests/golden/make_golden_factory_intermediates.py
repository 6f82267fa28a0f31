"""The factory cases' INTERMEDIATES from the REAL reference (in vitl_linear_224.npz / vitl_dpt_512.npz the outputs are
the reference's, the intermediates — encoder features, decoder outputs, DPT feature maps, decoded channels — were the oracle's, because
the reference's DUSt3R.forward returns none of them):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_factory_intermediates.py

Runs the reference's factory model with FORWARD HOOKS on its own sub-modules (encoder, info_sharing, dpt_feature_head{1,2},
dpt_regressor_head{1,2} / head{1,2}) — same filler weights and images as make_golden.py — and compares what they produced with the
intermediates stored in the fixtures, on the stored samples and norms: every one must agree to < 2e-5.  Writes
tests/golden/factory_intermediates_ref.json (per case and tensor: relative L2 on the samples, relative norm difference): data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import dust3r_oracle as O  # noqa: E402
from tests.golden.cases import CASES, GAINS, sample_indices  # noqa: E402

from uniception.models.factory.dust3r import DUSt3R  # noqa: E402


def run(name, c):
    torch.manual_seed(0)
    model = DUSt3R(name="g", img_size=tuple(c["img"]), pred_head_type=c["head"]).eval()
    O.fill_state_dict_(model.state_dict(), gain=1.0, gains=GAINS)
    got = {}
    B = c["B"]

    def enc_hook(_m, _i, out):          # one batched call (both views) or two calls: collect in order
        got.setdefault("_enc", []).append(out.features)          # [b, C, h, w]

    def info_hook(_m, _i, out):
        final, inter = (out, []) if not isinstance(out, tuple) else out
        got["dec_final1"], got["dec_final2"] = final.features[0], final.features[1]
        for j, o in enumerate(inter):
            got[f"dec_take{j}_1"], got[f"dec_take{j}_2"] = o.features[0], o.features[1]

    hooks = [model.encoder.register_forward_hook(enc_hook), model.info_sharing.register_forward_hook(info_hook)]
    for v in (1, 2):
        if c["head"] == "dpt":
            hooks.append(getattr(model, f"dpt_feature_head{v}").register_forward_hook(
                lambda _m, _i, out, v=v: got.__setitem__(f"dpt_up8_{v}", out.features_upsampled_8x)))
            hooks.append(getattr(model, f"dpt_regressor_head{v}").register_forward_hook(
                lambda _m, _i, out, v=v: got.__setitem__(f"decoded{v}", out.decoded_channels)))
        else:
            hooks.append(getattr(model, f"head{v}").register_forward_hook(
                lambda _m, _i, out, v=v: got.__setitem__(f"decoded{v}", out.decoded_channels)))
    img1, img2 = O.make_images(c["seed"], B, *c["img"])
    v1 = {"img": img1, "instance": [str(i) for i in range(B)], "data_norm_type": "dust3r"}
    v2 = {"img": img2, "instance": [str(100 + i) for i in range(B)], "data_norm_type": "dust3r"}
    with torch.no_grad():
        model(v1, v2)
    for h in hooks:
        h.remove()
    enc = torch.cat(got.pop("_enc"), 0)
    got["enc_feat1"], got["enc_feat2"] = enc[:B], enc[B:2 * B]
    z = np.load(os.path.join(HERE, name + ".npz"))
    stored = sorted({k.rsplit("__", 1)[0] for k in z.files if k.endswith("__samples")} - {"pts3d_1", "conf_1", "pts3d_2", "conf_2"})
    report, worst = {}, 0.0
    for k in stored:
        assert k in got, (name, k, sorted(got))
        t = got[k].detach().float()
        assert tuple(t.shape) == tuple(int(x) for x in z[k + "__shape"]), (name, k, tuple(t.shape), z[k + "__shape"])
        s_ref = t.flatten()[sample_indices(t.numel())].double()
        s_fix = torch.from_numpy(z[k + "__samples"]).double()
        e = float((s_fix - s_ref).norm() / s_ref.norm())
        en = abs(float(z[k + "__norm"]) - float(t.double().norm())) / float(t.double().norm())
        report[k] = {"samples_rel_l2": e, "norm_rel_diff": en}
        worst = max(worst, e, en)
        print(f"{name:18s} {k:14s} {tuple(t.shape)}  fixture vs reference: samples {e:.2e}  norm {en:.2e}", flush=True)
    assert worst < 2e-5, (name, worst)
    return report


def main():
    out = {name: run(name, c) for name, c in CASES.items() if c.get("factory")}
    with open(os.path.join(HERE, "factory_intermediates_ref.json"), "w") as f:
        json.dump({"what": "intermediates stored in the factory fixtures vs forward hooks on the REAL reference's sub-modules (rel-L2 on the "
                           "stored samples, relative difference of the norms); bar 2e-5", "cases": out}, f, indent=1)
    print("wrote factory_intermediates_ref.json")


if __name__ == "__main__":
    main()
