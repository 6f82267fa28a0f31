"""Global / alternating multi-view transformers against golden vectors of the REAL reference
(tests/golden/make_golden_multiview.py): RoPE over all views' tokens, view positional encodings (sequential and the random
draw), per-view and global extra tokens, LayerScale, the intermediate-feature returners."""
import numpy as np
import os
import pytest
import torch

from tests.golden.multiview_cases import DIMS, MV_CASES, RAND_SEED, fill, inputs, resolve
from tests.helpers import GOLDEN_DIR, rel_l2

pytestmark = pytest.mark.gpu


def _flatten(prefix, out, store):
    for v, f in enumerate(out.features):
        store[f"{prefix}feat{v}"] = f
    if out.additional_token_features is not None:
        store[f"{prefix}glob"] = out.additional_token_features
    if out.additional_token_features_per_view is not None:
        for v, f in enumerate(out.additional_token_features_per_view):
            store[f"{prefix}pv{v}"] = f


@pytest.mark.parametrize("name", list(MV_CASES.keys()))
@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 3e-2)])
def test_multiview_transformer_parity(gpu, name, mode, tol):
    from uniception_amd import engine
    from uniception_amd.models.info_sharing import INFO_SHARING_CLASSES, MultiViewTransformerInput
    from uniception_amd.models.libs.croco.pos_embed import RoPE2D
    gold = np.load(os.path.join(GOLDEN_DIR, "multiview.npz"))
    key, extra, V, Tp, G, indices = MV_CASES[name]
    cls, cls_ifr = INFO_SHARING_CLASSES[key]
    extra = resolve(extra, RoPE2D)
    model = (cls_ifr(name=name, indices=indices, **DIMS, **extra) if indices is not None else cls(name=name, **DIMS, **extra)).eval()
    fill(model)
    model = model.to(gpu)
    feats, per_view, glob = inputs(name)
    mvi = MultiViewTransformerInput(features=[f.to(gpu) for f in feats],
                                    additional_input_tokens=None if glob is None else glob.to(gpu),
                                    additional_input_tokens_per_view=None if per_view is None else [t.to(gpu) for t in per_view])
    keep = feats[0].clone()
    torch.manual_seed(RAND_SEED)
    with torch.no_grad(), engine.precision(mode):
        res = model(mvi)
    torch.cuda.synchronize()
    got = {}
    if indices is not None:
        final, inter = res
        _flatten(f"{name}/", final, got)
        for j, o in enumerate(inter):
            _flatten(f"{name}/take{j}_", o, got)
    else:
        _flatten(f"{name}/", res, got)
    want = {k: gold[k] for k in gold.files if k.startswith(name + "/")}
    assert set(got) == set(want)
    worst = 0.0
    for k, w in want.items():
        assert tuple(got[k].shape) == tuple(w.shape), k
        e = rel_l2(got[k].float().cpu(), w)
        worst = max(worst, e)
        assert e < tol, f"{k}: rel-L2 {e:.3e}"
    assert torch.equal(mvi.features[0].cpu(), keep), "inputs are not modified (the view encoding is added to a copy)"
    print(f"\n[{mode}] {name}: worst rel-L2 {worst:.2e} over {len(want)} tensors")
