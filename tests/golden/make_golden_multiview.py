"""Golden vectors of the global / alternating multi-view transformers from the REAL reference (same recipe as make_golden.py:
build container only, reference imported from /root/reference with the two import stubs of SURVEY.md App. B):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_multiview.py

Writes tests/golden/multiview.npz: for every case of tests/golden/multiview_cases.py the reference outputs (per-view features,
extra-token features, intermediates).  Inputs and weights are regenerated from seeds / the name-keyed filler; data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.multiview_cases import DIMS, MV_CASES, RAND_SEED, fill, inputs, resolve  # noqa: E402

from uniception.models.info_sharing import INFO_SHARING_CLASSES  # noqa: E402
from uniception.models.info_sharing.base import MultiViewTransformerInput  # noqa: E402
from uniception.models.libs.croco.pos_embed import RoPE2D  # noqa: E402


def flatten(prefix, out, store):
    for v, f in enumerate(out.features):
        store[f"{prefix}feat{v}"] = f.detach().numpy()
    if out.additional_token_features is not None:
        store[f"{prefix}glob"] = out.additional_token_features.detach().numpy()
    if out.additional_token_features_per_view is not None:
        for v, f in enumerate(out.additional_token_features_per_view):
            store[f"{prefix}pv{v}"] = f.detach().numpy()


def main():
    store = {}
    for name, (key, extra, V, Tp, G, indices) in MV_CASES.items():
        cls, cls_ifr = INFO_SHARING_CLASSES[key]
        extra = resolve(extra, RoPE2D)
        if indices is not None:
            model = cls_ifr(name=name, indices=indices, **DIMS, **extra).eval()
        else:
            model = cls(name=name, **DIMS, **extra).eval()
        fill(model)
        feats, per_view, glob = inputs(name)
        torch.manual_seed(RAND_SEED)
        with torch.no_grad():
            res = model(MultiViewTransformerInput(features=feats, additional_input_tokens=glob, additional_input_tokens_per_view=per_view))
        if indices is not None:
            final, inter = res
            flatten(f"{name}/", final, store)
            for j, o in enumerate(inter):
                flatten(f"{name}/take{j}_", o, store)
        else:
            flatten(f"{name}/", res, store)
        print(name, "ok", {k: v.shape for k, v in store.items() if k.startswith(name + "/")})
    np.savez_compressed(os.path.join(HERE, "multiview.npz"), **store)
    print("wrote", os.path.join(HERE, "multiview.npz"))
    grads()


def grads():
    """Gradient fixtures (multiview_grads.npz): the reference's own autograd through its global / alternating transformer."""
    from tests.golden.cases import sample_indices
    from tests.golden.multiview_cases import MV_GRAD_CASES, case, grad_weights, output_list
    store = {}
    for name in MV_GRAD_CASES:
        key, extra, V, Tp, G, indices = case(name)
        assert indices is None
        cls, _ = INFO_SHARING_CLASSES[key]
        model = cls(name=name, **DIMS, **resolve(extra, RoPE2D)).train()
        fill(model)
        feats, per_view, glob = inputs(name)
        leaves = [t.requires_grad_(True) for t in feats + (per_view or []) + ([glob] if glob is not None else [])]
        torch.manual_seed(RAND_SEED)
        out = model(MultiViewTransformerInput(features=feats, additional_input_tokens=glob, additional_input_tokens_per_view=per_view))
        outs = output_list(out)
        ws = grad_weights(name, [tuple(t.shape) for t in outs])
        loss = sum((t * w).sum() for t, w in zip(outs, ws))
        loss.backward()
        store[f"{name}/loss"] = np.float64(float(loss.detach()))
        for i, t in enumerate(leaves):
            store[f"{name}/din{i}"] = t.grad.detach().numpy()
        for k, prm in model.named_parameters():
            if prm.grad is None:
                continue
            idx = sample_indices(prm.grad.numel(), 512)
            store[f"{name}/p/{k}__samples"] = prm.grad.flatten()[idx].float().numpy()
            store[f"{name}/p/{k}__norm"] = np.float64(prm.grad.double().norm().item())
        print(name, "grads ok: loss", float(loss.detach()), sum(1 for k in store if k.startswith(name + "/p/")) // 2, "parameter gradients")
    np.savez_compressed(os.path.join(HERE, "multiview_grads.npz"), **store)
    print("wrote", os.path.join(HERE, "multiview_grads.npz"))


if __name__ == "__main__":
    if sys.argv[1:] == ["grads"]:      # only the gradient fixtures (multiview.npz untouched)
        grads()
    else:
        main()
