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


@pytest.mark.parametrize("name", ["global_rope_v3", "alt_tokens_v2", "alt_ls_v2", "global_ls_tokens_v2"])   # = multiview_cases.MV_GRAD_CASES
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_multiview_transformer_gradients_match_reference_autograd(gpu, name, mode):
    """Training through the global / alternating transformers: HIP forward + backward sub-layers against the reference's own
    autograd (tests/golden/multiview_grads.npz: L = sum of <output, seeded cotangent> over every output tensor) — gradients of the
    input features, of the extra tokens and of every parameter.  fp32 mode: rel-L2 < 1e-3; bf16 mode: cosine > 0.999."""
    from tests.golden.cases import sample_indices
    from tests.golden.multiview_cases import case, grad_weights, output_list
    from uniception_amd import engine
    from uniception_amd.models.info_sharing import INFO_SHARING_CLASSES, MultiViewTransformerInput
    from uniception_amd.models.libs.croco.pos_embed import RoPE2D
    gold = np.load(os.path.join(GOLDEN_DIR, "multiview_grads.npz"))
    key, extra, V, Tp, G, indices = case(name)
    cls, _ = INFO_SHARING_CLASSES[key]
    model = cls(name=name, **DIMS, **resolve(extra, RoPE2D)).train()
    fill(model)
    model = model.to(gpu)
    feats, per_view, glob = inputs(name)
    feats = [f.to(gpu).requires_grad_(True) for f in feats]
    per_view = None if per_view is None else [t.to(gpu).requires_grad_(True) for t in per_view]
    glob = None if glob is None else glob.to(gpu).requires_grad_(True)
    leaves = feats + (per_view or []) + ([glob] if glob is not None else [])
    torch.manual_seed(RAND_SEED)
    with engine.precision(mode):
        out = model(MultiViewTransformerInput(features=feats, additional_input_tokens=glob, additional_input_tokens_per_view=per_view))
        outs = output_list(out)
        ws = grad_weights(name, [tuple(t.shape) for t in outs])
        loss = sum((t * w.to(gpu)).sum() for t, w in zip(outs, ws))
    loss.backward()
    torch.cuda.synchronize()
    want_loss = float(gold[f"{name}/loss"])
    if mode == "fp32":      # (the loss is a signed sum with heavy cancellation: in bf16 only the gradients' direction is checked)
        assert abs(float(loss.detach()) - want_loss) <= 1e-4 * max(1.0, abs(want_loss)), (float(loss.detach()), want_loss)

    def check(got, want, what):
        got, want = got.double().flatten().cpu(), torch.from_numpy(np.asarray(want)).double().flatten()
        if mode == "fp32":
            assert rel_l2(got, want) < 1e-3, f"{what}: rel-L2 {rel_l2(got, want):.2e}"
        else:
            cos = float(torch.dot(got, want) / (got.norm() * want.norm()).clamp_min(1e-30))
            assert cos > 0.999, f"{what}: cosine {cos:.5f}"

    for i, t in enumerate(leaves):
        check(t.grad, gold[f"{name}/din{i}"], f"d input {i}")
    n = 0
    for k, prm in model.named_parameters():
        key_s = f"{name}/p/{k}__samples"
        if key_s not in gold.files:
            continue
        assert prm.grad is not None, k
        idx = sample_indices(prm.grad.numel(), 512)
        want_s, want_n = gold[key_s], float(gold[f"{name}/p/{k}__norm"])
        if want_n < 1e-12:
            continue
        got_n = float(prm.grad.double().norm())
        if mode == "fp32":
            assert abs(got_n - want_n) < 1e-3 * want_n, f"{k}: norm {got_n} vs {want_n}"
        if float(np.linalg.norm(want_s)) > 1e-3 * want_n * (len(idx) / prm.grad.numel()) ** 0.5:      # (a sample that carries signal)
            check(prm.grad.flatten()[torch.from_numpy(idx).to(gpu)], want_s, k)
        n += 1
    assert n >= 40
