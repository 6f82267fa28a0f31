"""Shared test helpers: build the two-view model of a golden case from uniception_amd modules, load fixtures."""
import os

import numpy as np
import torch
import torch.nn as nn

from oracle import dust3r_oracle as O
from tests.golden.cases import CASES, GAINS, sample_indices

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class ComposedTwoView(nn.Module):
    """DUSt3R wiring with free dimensions (the factory hard-codes ViT-L), built from uniception_amd modules.
    Mirrors the composition used by tests/golden/make_golden.py on the reference's modules."""

    def __init__(self, c):
        super().__init__()
        from uniception_amd.models.encoders.croco import CroCoEncoder
        from uniception_amd.models.info_sharing.cross_attention_transformer import (
            MultiViewCrossAttentionTransformer, MultiViewCrossAttentionTransformerIFR)
        from uniception_amd.models.libs.croco.pos_embed import RoPE2D
        from uniception_amd.models.prediction_heads.adaptors import PointMapWithConfidenceAdaptor
        from uniception_amd.models.prediction_heads.dpt import DPTFeature, DPTRegressionProcessor
        from uniception_amd.models.prediction_heads.linear import LinearFeature

        rope = RoPE2D(freq=100.0)
        self.c = c
        self.encoder = CroCoEncoder(name="enc", data_norm_type="dust3r", img_size=tuple(c["img"]), patch_size=c["patch"],
                                    enc_embed_dim=c["enc_dim"], enc_depth=c["enc_depth"], enc_num_heads=c["enc_heads"])
        kw = dict(name="dec", input_embed_dim=c["enc_dim"], num_views=2, depth=c["dec_depth"], dim=c["dec_dim"],
                  num_heads=c["dec_heads"], custom_positional_encoding=rope)
        if c["head"] == "dpt":
            self.info_sharing = MultiViewCrossAttentionTransformerIFR(indices=list(c["indices"]), norm_intermediate=False, **kw)
            for v in (1, 2):
                setattr(self, f"dpt_feature_head{v}", DPTFeature(
                    patch_size=c["patch"], hooks=[0, 1, 2, 3], input_feature_dims=[c["enc_dim"]] + [c["dec_dim"]] * 3,
                    layer_dims=list(c["layer_dims"]), feature_dim=c["feature_dim"]))
                setattr(self, f"dpt_regressor_head{v}", DPTRegressionProcessor(input_feature_dim=c["feature_dim"], output_dim=4))
        else:
            self.info_sharing = MultiViewCrossAttentionTransformer(**kw)
            self.head1 = LinearFeature(c["dec_dim"], 4, c["patch"])
            self.head2 = LinearFeature(c["dec_dim"], 4, c["patch"])
        self.adaptor = PointMapWithConfidenceAdaptor(name="pointmap", pointmap_mode="exp", pointmap_vmin=-float("inf"),
                                                     pointmap_vmax=float("inf"), confidence_type="exp", confidence_vmin=1,
                                                     confidence_vmax=float("inf"))

    def forward(self, img1, img2, collect):
        from uniception_amd.models.encoders.base import ViTEncoderInput
        from uniception_amd.models.info_sharing.base import MultiViewTransformerInput
        from uniception_amd.models.prediction_heads.base import AdaptorInput, PredictionHeadInput, PredictionHeadLayeredInput

        c = self.c
        B, _, H, W = img1.shape
        feats = self.encoder(ViTEncoderInput(image=torch.cat([img1, img2], 0), data_norm_type="dust3r")).features
        f1, f2 = feats.chunk(2, dim=0)
        inp = MultiViewTransformerInput(features=[f1, f2])
        if c["head"] == "dpt":
            final, inter = self.info_sharing(inp)
        else:
            final, inter = self.info_sharing(inp), []
        collect.update(enc_feat1=f1, enc_feat2=f2, dec_final1=final.features[0], dec_final2=final.features[1])
        for j, t in enumerate(inter):
            collect[f"dec_take{j}_1"], collect[f"dec_take{j}_2"] = t.features[0], t.features[1]
        res = []
        for v in range(2):
            if c["head"] == "dpt":
                lay = [(f1, f2)[v], inter[0].features[v], inter[1].features[v], final.features[v]]
                up8 = getattr(self, f"dpt_feature_head{v + 1}")(PredictionHeadLayeredInput(list_features=lay, target_output_shape=(H, W)))
                collect[f"dpt_up8_{v + 1}"] = up8.features_upsampled_8x
                dec = getattr(self, f"dpt_regressor_head{v + 1}")(up8).decoded_channels
            else:
                dec = getattr(self, f"head{v + 1}")(PredictionHeadInput(last_feature=final.features[v])).decoded_channels
            collect[f"decoded{v + 1}"] = dec
            a = self.adaptor(AdaptorInput(adaptor_feature=dec, output_shape_hw=(H, W)))
            res.append((a.value.permute(0, 2, 3, 1).contiguous(), a.confidence.permute(0, 2, 3, 1).contiguous()))
        return ({"pts3d": res[0][0], "conf": res[0][1]}, {"pts3d_in_other_view": res[1][0], "conf": res[1][1]})


_CASE_MODELS = {}


def build_case_model(name):
    """uniception_amd model of golden case `name`, weights from the name-keyed filler (CPU tensors).  Built once per session and
    handed out as deep copies: constructing + filling the 570 M-parameter factory model takes ~10 s, a copy ~1 s, and two dozen
    tests ask for it."""
    import copy
    c = CASES[name]
    if name not in _CASE_MODELS:
        if c.get("factory"):
            from uniception_amd.models.factory import DUSt3R
            model = DUSt3R(name="g", img_size=tuple(c["img"]), pred_head_type=c["head"]).eval()
        else:
            model = ComposedTwoView(c).eval()
        O.fill_state_dict_(model.state_dict(), gain=1.0, gains=GAINS)
        _CASE_MODELS[name] = model
    return copy.deepcopy(_CASE_MODELS[name]), c


def case_images(c):
    return O.make_images(c["seed"], c["B"], *c["img"])


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


HEAD_OUTPUTS = ("pts3d_1", "pts3d_2", "conf_1", "conf_2")


def compare_to_golden(golden, tensors, c, tol, report=None, max_abs_tol=None, abs_report=None):
    """tensors: name -> torch tensor (any device/strides, BCHW- or BHWC-shaped like the reference's).
    Full-tensor cases compare everything; sampled cases compare the strided samples and the L2 norm.
    max_abs_tol: the second half of the reference's own acceptance gate (examples/models/dust3r/dust3r.py:223-230:
    max |x - y| < 1e-2 AND rel-L2 < 1e-3 on the four head outputs) — asserted on HEAD_OUTPUTS when given; abs_report
    receives the max-abs error of every compared tensor."""
    worst = ("", 0.0)
    for k, t in tensors.items():
        t = t.detach().float().cpu().contiguous()
        if c["store"] == "full":
            if k not in golden:
                continue
            g = golden[k]
            assert tuple(t.shape) == tuple(g.shape), f"{k}: shape {tuple(t.shape)} vs golden {g.shape}"
            err = rel_l2(t, g)
            aerr = float((t.double() - torch.as_tensor(g).double()).abs().max())
        else:
            if k + "__samples" not in golden:
                continue
            assert tuple(t.shape) == tuple(golden[k + "__shape"]), f"{k}: shape {tuple(t.shape)} vs golden {golden[k + '__shape']}"
            idx = sample_indices(t.numel())
            err = rel_l2(t.flatten()[idx], golden[k + "__samples"])
            aerr = float((t.flatten()[idx].double() - torch.as_tensor(golden[k + "__samples"]).double()).abs().max())
            nerr = abs(float(t.double().norm()) - float(golden[k + "__norm"])) / float(golden[k + "__norm"])
            err = max(err, nerr)
        if abs_report is not None:
            abs_report[k] = aerr
        if max_abs_tol is not None and k in HEAD_OUTPUTS:
            assert aerr < max_abs_tol, f"{k}: max-abs error {aerr:.3e} exceeds {max_abs_tol:.1e}"
        if report is not None:
            report[k] = err
        if err > worst[1]:
            worst = (k, err)
        assert err < tol, f"{k}: rel-L2 {err:.3e} exceeds {tol:.1e}"
    return worst
