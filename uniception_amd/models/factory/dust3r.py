"""DUSt3R two-view pointmap model (reference: factory/dust3r.py:21-332) wired from the HIP-backed modules.

Same constructor, same submodule names (and therefore the same state_dict keys and aliases: `head1.0.* ==
dpt_feature_head1.*` etc.), same `forward(view1, view2) -> (res1, res2)` contract.
"""
import os
from typing import List, Tuple

import torch
import torch.nn as nn

from ... import engine
from ..encoders.base import ViTEncoderInput
from ..encoders.croco import CroCoEncoder
from ..info_sharing.base import MultiViewTransformerInput
from ..info_sharing.cross_attention_transformer import MultiViewCrossAttentionTransformer, MultiViewCrossAttentionTransformerIFR
from ..libs.croco.pos_embed import RoPE2D
from ..prediction_heads.adaptors import PointMapWithConfidenceAdaptor
from ..prediction_heads.base import AdaptorInput, PredictionHeadInput, PredictionHeadLayeredInput
from ..prediction_heads.dpt import DPTFeature, DPTRegressionProcessor
from ..prediction_heads.linear import LinearFeature


def is_symmetrized(gt1, gt2):
    "True when the batch is made of (a,b),(b,a) pairs: instance ids of consecutive samples are swapped (dust3r.py:21-30)."
    x, y = gt1["instance"], gt2["instance"]
    if len(x) == len(y) and len(x) == 1:
        return False
    ok = True
    for i in range(0, len(x), 2):
        ok = ok and (x[i] == y[i + 1]) and (x[i + 1] == y[i])
    return ok


def interleave(tensor1, tensor2):
    "(t1[0], t2[0], t1[1], t2[1], ...) and the swapped sequence (dust3r.py:33-37)."
    res1 = torch.stack((tensor1, tensor2), dim=1).flatten(0, 1)
    res2 = torch.stack((tensor2, tensor1), dim=1).flatten(0, 1)
    return res1, res2


class DUSt3R(nn.Module):
    "DUSt3R defined with UniCeption modules, computed by the MI355X kernel library."

    def __init__(self, name: str, data_norm_type: str = "dust3r", img_size: tuple = (224, 224),
                 patch_embed_cls: str = "PatchEmbedDust3R", pred_head_type: str = "linear", pred_head_output_dim: int = 4,
                 pred_head_feature_dim: int = 256, depth_mode: Tuple[str, float, float] = ("exp", -float("inf"), float("inf")),
                 conf_mode: Tuple[str, float, float] = ("exp", 1, float("inf")), pos_embed: str = "RoPE100",
                 pretrained_checkpoint_path: str = None, pretrained_encoder_checkpoint_path: str = None,
                 pretrained_info_sharing_checkpoint_path: str = None,
                 pretrained_pred_head_checkpoint_paths: List[str] = [None, None],
                 pretrained_pred_head_regressor_checkpoint_paths: List[str] = [None, None],
                 override_encoder_checkpoint_attributes: bool = False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.name = name
        self.data_norm_type = data_norm_type
        self.img_size = img_size
        self.patch_embed_cls = patch_embed_cls
        self.pred_head_type = pred_head_type
        self.pred_head_output_dim = pred_head_output_dim
        self.depth_mode = depth_mode
        self.conf_mode = conf_mode
        self.pos_embed = pos_embed
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        self.pretrained_encoder_checkpoint_path = pretrained_encoder_checkpoint_path
        self.pretrained_info_sharing_checkpoint_path = pretrained_info_sharing_checkpoint_path
        self.pretrained_pred_head_checkpoint_paths = pretrained_pred_head_checkpoint_paths
        self.pretrained_pred_head_regressor_checkpoint_paths = pretrained_pred_head_regressor_checkpoint_paths
        self.override_encoder_checkpoint_attributes = override_encoder_checkpoint_attributes

        self.rope = RoPE2D(freq=float(pos_embed[len("RoPE"):]))
        self.encoder = CroCoEncoder(name=name, data_norm_type=data_norm_type, patch_embed_cls=patch_embed_cls,
                                    img_size=img_size, pretrained_checkpoint_path=pretrained_encoder_checkpoint_path,
                                    override_checkpoint_attributes=override_encoder_checkpoint_attributes)
        if pred_head_type == "linear":
            self.info_sharing = MultiViewCrossAttentionTransformer(
                name="base_info_sharing", input_embed_dim=self.encoder.enc_embed_dim, num_views=2,
                custom_positional_encoding=self.rope, pretrained_checkpoint_path=pretrained_info_sharing_checkpoint_path)
        elif pred_head_type == "dpt":
            self.info_sharing = MultiViewCrossAttentionTransformerIFR(
                name="base_info_sharing", input_embed_dim=self.encoder.enc_embed_dim, num_views=2, indices=[5, 8],
                norm_intermediate=False, custom_positional_encoding=self.rope,
                pretrained_checkpoint_path=pretrained_info_sharing_checkpoint_path)
        else:
            raise ValueError(f"Invalid prediction head type: {pred_head_type}. Must be 'linear' or 'dpt'.")

        if pred_head_type == "linear":
            for v in (1, 2):
                setattr(self, f"head{v}", LinearFeature(
                    input_feature_dim=self.info_sharing.dim, output_dim=pred_head_output_dim, patch_size=self.encoder.patch_size,
                    pretrained_checkpoint_path=pretrained_pred_head_checkpoint_paths[v - 1]))
        else:
            for v in (1, 2):
                feat = DPTFeature(patch_size=self.encoder.patch_size, hooks=[0, 1, 2, 3],
                                  input_feature_dims=[self.encoder.enc_embed_dim] + [self.info_sharing.dim] * 3,
                                  feature_dim=pred_head_feature_dim,
                                  pretrained_checkpoint_path=pretrained_pred_head_checkpoint_paths[v - 1])
                reg = DPTRegressionProcessor(input_feature_dim=pred_head_feature_dim, output_dim=pred_head_output_dim,
                                             pretrained_checkpoint_path=pretrained_pred_head_regressor_checkpoint_paths[v - 1])
                setattr(self, f"dpt_feature_head{v}", feat)
                setattr(self, f"dpt_regressor_head{v}", reg)
                setattr(self, f"head{v}", nn.Sequential(feat, reg))

        self.adaptor = PointMapWithConfidenceAdaptor(
            name="pointmap", pointmap_mode=depth_mode[0], pointmap_vmin=depth_mode[1], pointmap_vmax=depth_mode[2],
            confidence_type=conf_mode[0], confidence_vmin=conf_mode[1], confidence_vmax=conf_mode[2])

        if pretrained_checkpoint_path is not None:
            print(f"Loading pretrained DUSt3R weights from {pretrained_checkpoint_path} ...")
            ckpt = torch.load(pretrained_checkpoint_path, weights_only=False)
            print(self.load_state_dict(ckpt["model"]))

    def _encode_image_pairs(self, img1, img2, data_norm_type):
        "Both views go through the encoder as one batch when their shapes agree (dust3r.py:211-225)."
        if (img1.shape == img2.shape and engine.CONCURRENT and (not torch.is_grad_enabled() or engine.TRAIN_CONCURRENT) and img1.is_cuda
                and img1.shape[0] * (img1.shape[-2] // self.encoder.patch_size) * (img1.shape[-1] // self.encoder.patch_size) > engine.BRANCH_TOKENS_MAX):
            # large batch: the two views as two concurrent kernel streams instead of one concatenated batch (engine.CONCURRENT);
            # every row of every kernel depends on its own image only, so the features are those of the concatenated run
            enc = lambda im: self.encoder(ViTEncoderInput(image=im, data_norm_type=data_norm_type)).features   # noqa: E731
            return engine.run_branches(lambda: enc(img1), lambda: enc(img2), 0, inputs1=(img2,),
                                       warm_key=("enc", tuple(img1.shape), str(engine.compute_dtype()), torch.is_grad_enabled()), owner=self.encoder,
                                       shared_params=True)
        if img1.shape[-2:] == img2.shape[-2:]:
            out = self.encoder(ViTEncoderInput(image=torch.cat((img1, img2), dim=0), data_norm_type=data_norm_type)).features
            return engine.chunk_bchw(out, 2)
        # the reference's different-shape branch raises TypeError (ViTEncoderInput built without data_norm_type,
        # dust3r.py:221); here the two views are simply encoded separately
        out1 = self.encoder(ViTEncoderInput(image=img1, data_norm_type=data_norm_type)).features
        out2 = self.encoder(ViTEncoderInput(image=img2, data_norm_type=data_norm_type)).features
        return out1, out2

    def _encode_symmetrized(self, view1, view2):
        img1, img2 = view1["img"], view2["img"]
        if is_symmetrized(view1, view2):
            feat1, feat2 = self._encode_image_pairs(img1[::2], img2[::2], data_norm_type=view1["data_norm_type"])
            feat1, feat2 = interleave(feat1, feat2)
        else:
            feat1, feat2 = self._encode_image_pairs(img1, img2, data_norm_type=view1["data_norm_type"])
        return feat1, feat2

    def _downstream_head(self, head_num, decout, img_shape):
        head = getattr(self, f"head{head_num}")
        if self.pred_head_type == "linear":
            head_input = PredictionHeadInput(last_feature=decout[f"{head_num}"])
        else:
            head_input = PredictionHeadLayeredInput(list_features=decout[f"{head_num}"], target_output_shape=img_shape)
        return head(head_input)

    def forward(self, view1, view2):
        """view dicts {"img": [B,3,H,W], "instance": [...], "data_norm_type": str} ->
        ({"pts3d","conf"}, {"pts3d_in_other_view","conf"}) with [B,H,W,3] / [B,H,W,1] fp32 tensors."""
        _, _, h1, w1 = view1["img"].shape
        _, _, h2, w2 = view2["img"].shape
        shape1, shape2 = (int(h1), int(w1)), (int(h2), int(w2))

        feat1, feat2 = self._encode_symmetrized(view1, view2)
        info_in = MultiViewTransformerInput(features=[feat1, feat2])

        def f32(t):      # the reference hands fp32 features to its heads (dust3r.py:288-309); rows of a bf16 residual stream that a bf16
            return t if (t.dtype == torch.bfloat16 and engine.head_dtype() in (torch.bfloat16, torch.float16)) else t.float()   # 16-bit head reads as they are
        if self.pred_head_type == "linear":
            final = self.info_sharing(info_in)
            outs = {"1": f32(final.features[0]), "2": f32(final.features[1])}
        else:
            final, inter = self.info_sharing(info_in)
            outs = {str(v + 1): [f32((feat1, feat2)[v]), f32(inter[0].features[v]), f32(inter[1].features[v]),
                                 f32(final.features[v])] for v in range(2)}

        # (with-items are entered left to right: read the transformer's dtype BEFORE autocast is switched off, or bf16 that came from
        # torch.autocast would reach the heads as fp32)
        transformer_dtype = engine.compute_dtype()
        with torch.autocast("cuda", enabled=False), engine.ambient(transformer_dtype):
            def head(num, shape):
                ho = self._downstream_head(num, outs, shape)
                fo = self.adaptor(AdaptorInput(adaptor_feature=ho.decoded_channels, output_shape_hw=shape))
                return fo.value.permute(0, 2, 3, 1).contiguous(), fo.confidence.permute(0, 2, 3, 1).contiguous()

            # the two heads are independent: on two streams when the batch is too small to fill the chip
            feats2 = outs["2"] if isinstance(outs["2"], list) else [outs["2"]]
            n_tok = feats2[-1].shape[0] * feats2[-1].shape[2] * feats2[-1].shape[3]

            def run_heads():
                return engine.run_branches(lambda: head(1, shape1), lambda: head(2, shape2), n_tok, inputs1=tuple(feats2),
                                           warm_key=("heads", tuple(feats2[-1].shape), shape1, shape2, str(engine.head_dtype()), torch.is_grad_enabled()), owner=self,
                                           disjoint_params=self.head1 is not self.head2)      # (head1 / head2: separate parameters; the adaptor has none)
            ran_f16 = engine.head_dtype() == torch.float16
            (p1, c1), (p2, c2) = run_heads()
            if ran_f16 and not torch.is_grad_enabled() and engine.heads_saturated_now():
                # a scratch map left the fp16 range in THIS forward: the policy has fallen back to the transformer's bf16 — redo the two
                # heads with it and return those maps instead of the saturated ones
                (p1, c1), (p2, c2) = run_heads()
            res1 = {"pts3d": p1, "conf": c1}
            res2 = {"pts3d_in_other_view": p2, "conf": c2}
            engine.note_heads_ran()      # (captured forwards: asynchronous snapshot of the range-guard flag, engine.head_range_exceeded)
        return res1, res2
