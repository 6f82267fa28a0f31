"""Golden vectors of the DINOv2 encoder from an INDEPENDENT implementation: HuggingFace transformers' Dinov2Model /
Dinov2WithRegistersModel (the reference takes this network from torch.hub "facebookresearch/dinov2", which is neither vendored under
/root/reference nor fetchable here; transformers ships its own implementation of the same published architecture, checkpoint-
compatible with the hub weights through a key map).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo python3 -B /root/repo/tests/golden/make_golden_dinov2_hf.py

The hub-named parameters of tests/golden/dinov2_cases.py are mapped onto the transformers module (qkv split into query / key / value,
ls*.gamma -> layer_scale*.lambda1, ...), loaded strictly, and the final-norm token stream of a seeded 518x518 image is stored:
strided samples + norm of the patch-feature map, the class / register tokens in full; for DINOV2_HF_GRAD_CASES also the gradients
(transformers' autograd) of loss = <features, Wf> + <class / register tokens, Wr> with respect to every parameter, mapped back to the
hub names (query / key / value -> attn.qkv, lambda1 -> gamma): 512 strided samples + norm each (data only)."""
import os
import sys

import numpy as np
import torch
import transformers
from transformers import Dinov2Config, Dinov2Model, Dinov2WithRegistersConfig, Dinov2WithRegistersModel

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.cases import sample_indices  # noqa: E402
from tests.golden.dinov2_cases import (DINOV2_HF_CASES, DINOV2_HF_GRAD_CASES, SIZES, SWIGLU_HIDDEN, dinov2_grad_weights, dinov2_hub_state_dict,  # noqa: E402
                                       dinov2_image)


# FFN parameter names, hub -> transformers: GELU MLP (False) / SwiGLU (True: w12 -> weights_in, w3 -> weights_out)
FFN_KEYS = {False: (("mlp.fc1.weight", "mlp.fc1.weight"), ("mlp.fc1.bias", "mlp.fc1.bias"), ("mlp.fc2.weight", "mlp.fc2.weight"), ("mlp.fc2.bias", "mlp.fc2.bias")),
            True: (("mlp.w12.weight", "mlp.weights_in.weight"), ("mlp.w12.bias", "mlp.weights_in.bias"), ("mlp.w3.weight", "mlp.weights_out.weight"),
                   ("mlp.w3.bias", "mlp.weights_out.bias"))}


def hub_to_hf(sd, prefix, layers, D, regs):
    g = lambda k: sd[prefix + k]   # noqa: E731
    out = {"embeddings.cls_token": g("cls_token"), "embeddings.position_embeddings": g("pos_embed"),
           "embeddings.mask_token": torch.zeros(1, D),
           "embeddings.patch_embeddings.projection.weight": g("patch_embed.proj.weight"),
           "embeddings.patch_embeddings.projection.bias": g("patch_embed.proj.bias"),
           "layernorm.weight": g("norm.weight"), "layernorm.bias": g("norm.bias")}
    if regs:
        out["embeddings.register_tokens"] = g("register_tokens")
    for i in range(layers):
        b, h = f"blocks.{i}.", f"encoder.layer.{i}."
        qw, kw, vw = g(b + "attn.qkv.weight").split(D, 0)
        qb, kb, vb = g(b + "attn.qkv.bias").split(D, 0)
        out.update({h + "norm1.weight": g(b + "norm1.weight"), h + "norm1.bias": g(b + "norm1.bias"),
                    h + "attention.attention.query.weight": qw, h + "attention.attention.query.bias": qb,
                    h + "attention.attention.key.weight": kw, h + "attention.attention.key.bias": kb,
                    h + "attention.attention.value.weight": vw, h + "attention.attention.value.bias": vb,
                    h + "attention.output.dense.weight": g(b + "attn.proj.weight"), h + "attention.output.dense.bias": g(b + "attn.proj.bias"),
                    h + "layer_scale1.lambda1": g(b + "ls1.gamma"), h + "norm2.weight": g(b + "norm2.weight"),
                    h + "norm2.bias": g(b + "norm2.bias"), h + "layer_scale2.lambda1": g(b + "ls2.gamma")})
        for hub_k, hf_k in FFN_KEYS[prefix + b + "mlp.w12.weight" in sd]:
            out[h + hf_k] = g(b + hub_k)
    return {k: v.clone() for k, v in out.items()}


def main():
    store = {"transformers_version": np.array(transformers.__version__)}
    for name, c in DINOV2_HF_CASES.items():
        D, H = SIZES[c["size"]]
        kw = dict(hidden_size=D, num_hidden_layers=c["layers"], num_attention_heads=H, mlp_ratio=4, hidden_act="gelu",
                  layer_norm_eps=1e-6, image_size=518, patch_size=14, num_channels=3, qkv_bias=True, layerscale_value=1.0,
                  use_swiglu_ffn=c["size"] in SWIGLU_HIDDEN, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
        model = (Dinov2WithRegistersModel(Dinov2WithRegistersConfig(num_register_tokens=4, **kw)) if c["regs"]
                 else Dinov2Model(Dinov2Config(**kw))).eval()
        sd = dinov2_hub_state_dict(c)
        print(name, model.load_state_dict(hub_to_hf(sd, "model.", c["layers"], D, c["regs"]), strict=True))
        with torch.no_grad():
            tok = model(pixel_values=dinov2_image(c)).last_hidden_state          # [B, 1 + R + hw, D], final LayerNorm applied
        R = 4 if c["regs"] else 0
        h0, w0 = c["hw"][0] // 14, c["hw"][1] // 14
        feats = tok[:, 1 + R:].permute(0, 2, 1).reshape(c["B"], D, h0, w0).contiguous()
        regs = tok[:, :1 + R].permute(0, 2, 1).contiguous()
        idx = sample_indices(feats.numel())
        store[f"{name}/features__samples"] = feats.flatten()[idx].numpy()
        store[f"{name}/features__norm"] = np.float64(feats.double().norm().item())
        store[f"{name}/features__shape"] = np.array(feats.shape)
        store[f"{name}/registers"] = regs.numpy()
        print(name, tuple(feats.shape), tuple(regs.shape), f"|features| {feats.norm().item():.4f}")
        if name in DINOV2_HF_GRAD_CASES:
            tok = model(pixel_values=dinov2_image(c)).last_hidden_state
            feats = tok[:, 1 + R:].permute(0, 2, 1).reshape(c["B"], D, h0, w0)
            regs = tok[:, :1 + R].permute(0, 2, 1)
            wf, wr = dinov2_grad_weights(name, feats.shape, regs.shape)
            loss = (feats * wf).sum() + (regs * wr).sum()
            loss.backward()
            hf = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
            hub = {"cls_token": hf["embeddings.cls_token"], "pos_embed": hf["embeddings.position_embeddings"],
                   "patch_embed.proj.weight": hf["embeddings.patch_embeddings.projection.weight"],
                   "patch_embed.proj.bias": hf["embeddings.patch_embeddings.projection.bias"],
                   "norm.weight": hf["layernorm.weight"], "norm.bias": hf["layernorm.bias"]}
            if c["regs"]:
                hub["register_tokens"] = hf["embeddings.register_tokens"]
            for i in range(c["layers"]):
                b, h = f"blocks.{i}.", f"encoder.layer.{i}."
                a = h + "attention.attention."
                hub[b + "attn.qkv.weight"] = torch.cat([hf[a + "query.weight"], hf[a + "key.weight"], hf[a + "value.weight"]], 0)
                hub[b + "attn.qkv.bias"] = torch.cat([hf[a + "query.bias"], hf[a + "key.bias"], hf[a + "value.bias"]], 0)
                for hub_k, hf_k in (("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"), ("attn.proj.weight", "attention.output.dense.weight"),
                                    ("attn.proj.bias", "attention.output.dense.bias"), ("ls1.gamma", "layer_scale1.lambda1"),
                                    ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias"), ("ls2.gamma", "layer_scale2.lambda1")) + \
                        FFN_KEYS[c["size"] in SWIGLU_HIDDEN]:
                    hub[b + hub_k] = hf[h + hf_k]
            store[f"{name}/loss"] = np.float64(float(loss.detach()))
            for k, gr in hub.items():
                store[f"{name}/grad/model.{k}__samples"] = gr.flatten()[sample_indices(gr.numel(), 512)].float().numpy()
                store[f"{name}/grad/model.{k}__norm"] = np.float64(gr.double().norm().item())
            print(name, "gradients of", len(hub), "parameters; loss", float(loss.detach()))
    np.savez_compressed(os.path.join(HERE, "dinov2_hf.npz"), **store)
    print("wrote", os.path.join(HERE, "dinov2_hf.npz"))


if __name__ == "__main__":
    main()
