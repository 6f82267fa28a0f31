"""Gradient checkpointing (reference: encoders/base.py:139-152, info_sharing/base.py:59-72; switched on by `gradient_checkpointing=True`
in the DINOv2 encoder and the multi-view transformers): wrapped blocks drop what their HIP sub-layer Functions saved and run their
forward again in the backward pass.  The property tested is the one the reference's wrapper has by construction: the same loss, the same
gradients (the re-computation is the same kernels on the same inputs; what differs is the order of fp32 atomic sums, 1e-7) — with less memory
held between the passes."""
import pytest
import torch

from tests.golden.multiview_cases import DIMS, RAND_SEED, fill, inputs, resolve

pytestmark = pytest.mark.gpu


def _run(model, leaves, forward, mode):
    from uniception_amd import engine
    for p in model.parameters():
        p.grad = None
    for t in leaves:
        t.grad = None
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    with engine.precision(mode):
        loss = forward()
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base          # what the recorded graph keeps alive between forward and backward
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), [t.grad.clone() for t in leaves], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, held


def _same(a, b, what):
    la, ia, pa, _ = a
    lb, ib, pb, _ = b
    # the forward is the same kernels on the same inputs: the same bits.  Gradients: the re-computed activations are the same bits too, but
    # LayerNorm / bias gradients are accumulated with fp32 atomics (column sums over rows), whose order differs from run to run even
    # without checkpointing: the last bits of a sum, 1e-7 relative
    assert la == lb, (what, la, lb)
    from tests.helpers import rel_l2
    for i, (x, y) in enumerate(zip(ia, ib)):
        assert rel_l2(x.double().cpu(), y.double().cpu()) < 1e-5, f"{what}: gradient of input {i} differs"
    assert pa.keys() == pb.keys() and len(pa) > 0
    for k in pa:
        assert rel_l2(pa[k].double().cpu(), pb[k].double().cpu()) < 1e-5, f"{what}: gradient of {k} differs ({rel_l2(pa[k].double().cpu(), pb[k].double().cpu()):.2e})"


@pytest.mark.parametrize("name", ["global_rope_v3", "alt_ls_v2", "global_ls_tokens_v2"])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_multiview_transformers_checkpointed_equal_plain(gpu, name, mode):
    from tests.golden.multiview_cases import case, grad_weights, output_list
    from uniception_amd.models.info_sharing import INFO_SHARING_CLASSES, MultiViewTransformerInput
    from uniception_amd.models.libs.croco.pos_embed import RoPE2D
    key, extra, V, Tp, G, indices = case(name)
    cls, _ = INFO_SHARING_CLASSES[key]
    res = {}
    for ck in (False, True):
        model = cls(name=name, **DIMS, **resolve(extra, RoPE2D), gradient_checkpointing=ck).train()
        fill(model)
        model = model.to(gpu)
        assert model.gradient_checkpointing == ck
        assert all((type(b).__name__.startswith("Checkpointed")) == ck for b in model.self_attention_blocks)
        feats, per_view, glob = inputs(name)
        feats = [torch.cat([f] * 8).to(gpu).requires_grad_(True) for f in feats]          # (a batch whose activations outweigh allocator granularity)
        per_view = None if per_view is None else [torch.cat([t] * 8).to(gpu).requires_grad_(True) for t in per_view]
        glob = None if glob is None else torch.cat([glob] * 8).to(gpu).requires_grad_(True)
        leaves = feats + (per_view or []) + ([glob] if glob is not None else [])

        def forward():
            torch.manual_seed(RAND_SEED)
            out = model(MultiViewTransformerInput(features=feats, additional_input_tokens=glob, additional_input_tokens_per_view=per_view))
            outs = output_list(out)
            g = torch.Generator().manual_seed(5)
            return sum((t * torch.randn(t.shape, generator=g).to(gpu)).sum() for t in outs)

        res[ck] = _run(model, leaves, forward, mode)
        with torch.no_grad():           # inference through a wrapped model: the fused pipeline, nothing recorded
            from uniception_amd import engine
            with engine.precision(mode):
                model.eval()(MultiViewTransformerInput(features=[f.detach() for f in feats], additional_input_tokens=None if glob is None else glob.detach(),
                                                       additional_input_tokens_per_view=None if per_view is None else [t.detach() for t in per_view]))
    _same(res[False], res[True], name)
    print(f"\n[checkpointing {mode}] {name}: held between the passes {res[False][3] / 2**20:.1f} MiB plain, {res[True][3] / 2**20:.1f} MiB checkpointed")
    assert res[True][3] < 0.6 * res[False][3]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_cross_attention_transformer_checkpointed_equal_plain(gpu, mode):
    "The DUSt3R decoder (two view branches, forked onto two kernel streams while autograd records): what the reference's wrapper means to do."
    from uniception_amd.models.info_sharing import MultiViewTransformerInput
    from uniception_amd.models.info_sharing.cross_attention_transformer import MultiViewCrossAttentionTransformer
    from uniception_amd.models.libs.croco.pos_embed import RoPE2D
    res = {}
    for ck in (False, True):
        model = MultiViewCrossAttentionTransformer(name="d", input_embed_dim=128, num_views=2, depth=3, dim=192, num_heads=3,
                                                   custom_positional_encoding=RoPE2D(100.0), gradient_checkpointing=ck).train()
        fill(model)
        model = model.to(gpu)
        assert all((type(b).__name__.startswith("Checkpointed")) == ck for br in model.multi_view_branches for b in br)
        g = torch.Generator().manual_seed(3)
        feats = [torch.randn(16, 128, 8, 12, generator=g).to(gpu).requires_grad_(True) for _ in range(2)]

        def forward():
            out = model(MultiViewTransformerInput(features=feats))
            gg = torch.Generator().manual_seed(6)
            return sum((t * torch.randn(t.shape, generator=gg).to(gpu)).sum() for t in out.features)

        res[ck] = _run(model, feats, forward, mode)
    _same(res[False], res[True], "cross-attention")
    print(f"\n[checkpointing {mode}] cross-attention: held {res[False][3] / 2**20:.1f} MiB plain, {res[True][3] / 2**20:.1f} MiB checkpointed")
    assert res[True][3] < 0.6 * res[False][3]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_dinov2_encoder_checkpointed_equal_plain(gpu, mode):
    from tests.golden.dinov2_cases import DINOV2_HF_CASES, dinov2_hub_state_dict, dinov2_image
    from uniception_amd.models.encoders import encoder_factory
    from uniception_amd.models.encoders.base import ViTEncoderInput
    c = DINOV2_HF_CASES["small_reg"]
    res = {}
    for ck in (False, True):
        enc = encoder_factory("dinov2", name="d", size=c["size"], with_registers=c["regs"], keep_first_n_layers=c["layers"], gradient_checkpointing=ck).train()
        enc.load_state_dict(dinov2_hub_state_dict(c), strict=True)
        enc = enc.to(gpu)
        assert all((type(b).__name__.startswith("Checkpointed")) == ck for b in enc.model.blocks)
        img = torch.cat([dinov2_image(c)] * 4).to(gpu)

        def forward():
            out = enc(ViTEncoderInput(image=img, data_norm_type="dinov2"))
            g = torch.Generator().manual_seed(7)
            return (out.features * torch.randn(out.features.shape, generator=g).to(gpu)).sum() + (out.registers * torch.randn(out.registers.shape, generator=g).to(gpu)).sum()

        res[ck] = _run(enc, [], forward, mode)
    _same(res[False], res[True], "dinov2")
    print(f"\n[checkpointing {mode}] dinov2: held {res[False][3] / 2**20:.1f} MiB plain, {res[True][3] / 2**20:.1f} MiB checkpointed")
    assert res[True][3] < 0.7 * res[False][3]


@pytest.mark.parametrize("kind", ["self", "cross"])
def test_checkpointed_block_with_random_masks_equals_the_plain_block_under_the_same_seed(gpu, kind):
    """proj_drop / the Mlp's drop / DropPath draw masks from PyTorch's generator in every forward, so a checkpointed block's
    re-computation has to see the generator state of its first forward (the reference's `checkpoint(..., use_reentrant=False)` keeps
    `preserve_rng_state=True`, encoders/base.py:139-152).  A wrapped block and the same block unwrapped, seeded alike, must produce the
    same output and the same gradients; with the state NOT restored the recomputed activations belong to other masks than the ones the
    saved ctx applies."""
    import copy
    from uniception_amd import engine
    from uniception_amd.models.utils.checkpointing import has_random_masks, wrap_module_with_gradient_checkpointing
    from uniception_amd.models.utils.transformer_blocks import CrossAttentionBlock, SelfAttentionBlock
    from tests.helpers import rel_l2
    B, N, Ny, C, H = 3, 40, 24, 128, 2
    torch.manual_seed(11)
    kw = dict(dim=C, num_heads=H, qkv_bias=True, proj_drop=0.2, drop_path=0.3, attn_drop=0.1)      # (attn_drop: its seed comes from the CPU generator)
    plain = (SelfAttentionBlock(**kw) if kind == "self" else CrossAttentionBlock(**kw)).to(gpu).train()
    ck = wrap_module_with_gradient_checkpointing(copy.deepcopy(plain))
    assert has_random_masks(plain) and not has_random_masks(SelfAttentionBlock(dim=C, num_heads=H))
    x0 = torch.randn(B, N, C, device=gpu)
    y0 = torch.randn(B, Ny, C, device=gpu)
    w = torch.randn(B, N, C, device=gpu)
    res = []
    for blk in (plain, ck):
        x = x0.clone().requires_grad_(True)
        y = y0.clone().requires_grad_(True)
        torch.manual_seed(1234)
        with engine.precision("fp32"):
            # two blocks' worth of generator use between the forward and the backward: the recomputation must not see the advanced state
            out = blk(x) if kind == "self" else blk(x, y)
            _ = torch.rand(1000, device=gpu)
            (out.float() * w).sum().backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), x.grad.clone(), y.grad.clone() if kind == "cross" else None,
                    {k: p.grad.clone() for k, p in blk.named_parameters()}))
    (o0, gx0, gy0, p0), (o1, gx1, gy1, p1) = res
    assert torch.equal(o0, o1)
    assert rel_l2(gx1.double().cpu(), gx0.double().cpu()) < 1e-5
    if kind == "cross":
        assert rel_l2(gy1.double().cpu(), gy0.double().cpu()) < 1e-5
    for k in p0:
        assert rel_l2(p1[k].double().cpu(), p0[k].double().cpu()) < 1e-5, k
