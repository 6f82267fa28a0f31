"""CPU: host-side logic of the nn.Module mirror — constructor/attribute surface, state_dict layout and aliases,
helper functions — and that no module silently computes on the CPU."""
import pytest
import torch

from uniception_amd._lib import UcHipError
from uniception_amd.models.encoders import ViTEncoderInput, encoder_factory
from uniception_amd.models.factory.dust3r import DUSt3R, interleave, is_symmetrized
from uniception_amd.models.info_sharing import INFO_SHARING_CLASSES, MultiViewTransformerInput
from uniception_amd.models.utils.intermediate_feature_return import feature_take_indices
from uniception_amd.models.utils.positional_encoding import PositionGetter


def test_feature_take_indices():
    assert feature_take_indices(12, None) == (list(range(12)), 11)
    assert feature_take_indices(12, 3) == ([9, 10, 11], 11)
    assert feature_take_indices(12, [5, 8]) == ([5, 8], 8)
    assert feature_take_indices(12, [-1, 0]) == ([11, 0], 11)
    with pytest.raises(AssertionError):
        feature_take_indices(12, [12])
    with pytest.raises(AssertionError):
        feature_take_indices(12, 13)


def test_symmetrized_detection_and_interleave():
    v1 = {"instance": ["a", "b", "c", "d"]}
    v2 = {"instance": ["b", "a", "d", "c"]}
    assert is_symmetrized(v1, v2)
    assert not is_symmetrized(v1, {"instance": ["b", "a", "c", "d"]})
    assert not is_symmetrized({"instance": ["a"]}, {"instance": ["a"]})  # batch-1 special case
    t1, t2 = torch.tensor([1, 3]), torch.tensor([2, 4])
    r1, r2 = interleave(t1, t2)
    assert r1.tolist() == [1, 2, 3, 4] and r2.tolist() == [2, 1, 4, 3]


def test_position_getter_row_major_yx():
    pos = PositionGetter()(2, 2, 3, "cpu")
    assert pos.shape == (2, 6, 2) and pos.dtype == torch.int64
    assert pos[0].tolist() == [[0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [1, 2]]


def test_dust3r_state_dict_surface():
    m = DUSt3R(name="x", img_size=(224, 224), pred_head_type="dpt")
    sd = m.state_dict()
    assert len(sd) == 1144 and sum(p.numel() for p in m.parameters()) == 568810120
    assert m.encoder.enc_embed_dim == 1024 and m.encoder.patch_size == 16 and m.info_sharing.dim == 768
    assert sd["encoder.patch_embed.proj.weight"].shape == (1024, 3, 16, 16)
    assert sd["encoder.enc_blocks.23.attn.qkv.weight"].shape == (3072, 1024)
    assert sd["info_sharing.proj_embed.weight"].shape == (768, 1024)
    assert sd["info_sharing.multi_view_branches.1.11.cross_attn.projk.weight"].shape == (768, 768)
    # aliases of the reference
    assert sd["head1.0.scratch.layer1_rn.weight"].data_ptr() == sd["dpt_feature_head1.scratch.layer_rn.0.weight"].data_ptr()
    assert sd["dpt_feature_head1.input_process.0.1.weight"].data_ptr() == sd["dpt_feature_head1.scratch.layer1_rn.weight"].data_ptr()
    assert sd["head2.1.conv2.2.bias"].data_ptr() == sd["dpt_regressor_head2.conv2.2.bias"].data_ptr()
    assert not any(k.startswith("dpt_feature_head1.scratch.refinenet4.resConfUnit1") for k in sd)
    assert sd["dpt_feature_head1.input_process.0.0.1.weight"].shape == (96, 96, 4, 4)  # ConvTranspose [Cin,Cout,k,k]
    lin = DUSt3R(name="x", img_size=(224, 224), pred_head_type="linear")
    assert sum(p.numel() for p in lin.parameters()) == 532342016
    assert lin.state_dict()["head1.linear.weight"].shape == (1024, 768, 1, 1)
    with pytest.raises(ValueError):
        DUSt3R(name="x", pred_head_type="nope")


def test_factories_and_registry():
    enc = encoder_factory("croco", name="e", data_norm_type="dust3r", img_size=(32, 32), enc_embed_dim=64, enc_depth=1, enc_num_heads=1)
    assert type(enc).__name__ == "CroCoEncoder"
    with pytest.raises(ValueError):
        encoder_factory("nope")
    assert set(INFO_SHARING_CLASSES) == {"cross_attention", "alternating_attention", "global_attention"}
    for key, (cls, ifr) in INFO_SHARING_CLASSES.items():
        assert issubclass(ifr, cls), key
    with pytest.raises(AssertionError):  # data-norm check fires before any compute
        enc(ViTEncoderInput(image=torch.zeros(1, 3, 32, 32), data_norm_type="croco"))


def test_no_silent_cpu_compute():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    enc = encoder_factory("croco", name="e", data_norm_type="dust3r", img_size=(32, 32), enc_embed_dim=64, enc_depth=1, enc_num_heads=1)
    with torch.no_grad(), pytest.raises(UcHipError):
        enc(ViTEncoderInput(image=torch.zeros(1, 3, 32, 32), data_norm_type="dust3r"))
    cls, _ = INFO_SHARING_CLASSES["cross_attention"]
    dec = cls(name="d", input_embed_dim=64, num_views=2, depth=1, dim=64, num_heads=1)
    with torch.no_grad(), pytest.raises(UcHipError):
        dec(MultiViewTransformerInput(features=[torch.zeros(1, 64, 2, 2), torch.zeros(1, 64, 2, 2)]))
    with pytest.raises(AssertionError):
        dec(MultiViewTransformerInput(features=[torch.zeros(1, 64, 2, 2)]))


def test_multiview_transformers_and_adaptors_construct_and_refuse_cpu_compute():
    """f2 / f4 constructors carry the reference's parameter names; compute on CPU tensors fails loudly."""
    from uniception_amd.models.prediction_heads import adaptors as A
    g_cls, g_ifr = INFO_SHARING_CLASSES["global_attention"]
    a_cls, _ = INFO_SHARING_CLASSES["alternating_attention"]
    glob = g_cls(name="g", input_embed_dim=48, max_num_views_for_pe=3, depth=2, dim=64, num_heads=2)
    alt = a_cls(name="a", input_embed_dim=48, depth=2, dim=64, num_heads=2, distinguish_ref_and_non_ref_views=True)
    assert len(glob.self_attention_blocks) == 2 and len(alt.self_attention_blocks) == 2
    feats = [torch.zeros(1, 48, 2, 2) for _ in range(2)]
    for m in (glob, alt):
        with torch.no_grad(), pytest.raises(UcHipError):
            m(MultiViewTransformerInput(features=feats))
    ifr = g_ifr(name="gi", input_embed_dim=48, max_num_views_for_pe=3, depth=4, dim=64, num_heads=2, indices=[1, 3])
    assert ifr.indices == [1, 3]
    # adaptor family: every class of the reference's adaptors module exists and is a UniCeptionAdaptorBase
    names = ["FlowAdaptor", "DepthAdaptor", "PointMapAdaptor", "RayDirectionsAdaptor", "ConfidenceAdaptor", "MaskAdaptor", "Covariance2DAdaptor",
             "ValueWithConfidenceAdaptor", "PointMapWithConfidenceAdaptor", "PointMapWithConfidenceAndMaskAdaptor", "RayMapPlusDepthAdaptor",
             "ScaleAdaptor", "CamTranslationPlusQuatsAdaptor"]
    from uniception_amd.models.prediction_heads.base import UniCeptionAdaptorBase
    for n in names:
        assert issubclass(getattr(A, n), UniCeptionAdaptorBase), n


def test_training_inputs_are_rejected_not_mishandled():
    """Modules without a HIP backward refuse inputs that need gradients (no silent CPU/ATen fallback); modules with
    one go to the HIP autograd path, which needs a device."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from uniception_amd.models.libs.croco.blocks import Mlp
    mlp = Mlp(64, 128)            # a bare layer module (the trainable pipelines are the blocks'): inference only
    with pytest.raises(UcHipError, match="backward"):
        mlp(torch.zeros(2, 3, 64))
    dino = encoder_factory("dinov2", name="d", size="small", keep_first_n_layers=1)   # trainable: HIP autograd path, needs a device
    with pytest.raises(UcHipError, match="HIP device only"):
        dino(ViTEncoderInput(image=torch.zeros(1, 3, 28, 28), data_norm_type="dinov2"))
    dino.requires_grad_(False)
    with pytest.raises(UcHipError, match="HIP device only"):
        dino(ViTEncoderInput(image=torch.zeros(1, 3, 28, 28), data_norm_type="dinov2"))
    enc = encoder_factory("croco", name="e", data_norm_type="dust3r", img_size=(32, 32), enc_embed_dim=64, enc_depth=1, enc_num_heads=1)
    with pytest.raises(UcHipError, match="HIP device only"):
        enc(ViTEncoderInput(image=torch.zeros(1, 3, 32, 32), data_norm_type="dust3r"))

def test_gradient_checkpointing_wraps_blocks_like_the_reference():
    """`gradient_checkpointing=True` swaps each block's class for a `_CheckpointingWrapper` subclass that remembers the original
    (`_restore_cls`, encoders/base.py:139-152): state_dict names and isinstance checks are unchanged, wrapping twice is a no-op.
    (The re-computation itself runs HIP kernels: tests/test_checkpointing_gpu.py.)"""
    from uniception_amd.models.encoders import encoder_factory
    from uniception_amd.models.info_sharing import INFO_SHARING_CLASSES
    from uniception_amd.models.utils.transformer_blocks import CrossAttentionBlock, SelfAttentionBlock
    plain = {}
    for key, blocks_of in (("cross_attention", lambda m: [b for br in m.multi_view_branches for b in br]),
                           ("global_attention", lambda m: list(m.self_attention_blocks)),
                           ("alternating_attention", lambda m: list(m.self_attention_blocks))):
        cls, _ = INFO_SHARING_CLASSES[key]
        kw = dict(name="t", input_embed_dim=64, dim=128, num_heads=2, depth=2)
        if key == "cross_attention":
            kw["num_views"] = 2
        a, b = cls(**kw), cls(**kw, gradient_checkpointing=True)
        assert list(a.state_dict().keys()) == list(b.state_dict().keys())
        for blk in blocks_of(b):
            assert type(blk).__name__.startswith("Checkpointed") and isinstance(blk, (CrossAttentionBlock, SelfAttentionBlock))
            assert type(blk)._restore_cls in (CrossAttentionBlock, SelfAttentionBlock)
            assert b.wrap_module_with_gradient_checkpointing(blk) is blk and type(blk)._restore_cls in (CrossAttentionBlock, SelfAttentionBlock)
        assert not any(type(blk).__name__.startswith("Checkpointed") for blk in blocks_of(a))
    enc = encoder_factory("dinov2", name="d", size="small", with_registers=True, keep_first_n_layers=2, gradient_checkpointing=True)
    assert enc.gradient_checkpointing and all(type(b).__name__.startswith("Checkpointed") for b in enc.model.blocks)


def test_dropout_masks_follow_nn_dropout_and_timm_droppath():
    """autograd.make_drops (host logic of the training-time dropout): None in eval mode or at rate 0; element masks keep with
    probability 1 - p and scale by 1 / (1 - p) (nn.Dropout); DropPath masks are one byte per SAMPLE, scale 1 / keep unless
    scale_by_keep is off (timm); PyTorch's generator owns the state (same seed, same masks)."""
    import torch
    from uniception_amd import autograd
    assert autograd.make_drops(False, "cpu", 2, 8, 16, p_out=0.5, p_path=0.5) is None
    assert autograd.make_drops(True, "cpu", 2, 8, 16) is None
    torch.manual_seed(3)
    d = autograd.make_drops(True, "cpu", 64, 32, 64, p_out=0.25, p_path=0.5, hidden=128, p_mid=0.1)
    assert d.out.shape == (64 * 32, 64) and d.out.dtype == torch.uint8 and abs(float(d.out.float().mean()) - 0.75) < 0.01
    assert d.mid.shape == (64 * 32, 128) and abs(float(d.mid.float().mean()) - 0.9) < 0.01
    assert d.path.shape == (64,) and d.path_rows == 32 and 0.2 < float(d.path.float().mean()) < 0.8
    assert abs(d.out_scale - 1 / 0.75) < 1e-6 and abs(d.mid_scale - 1 / 0.9) < 1e-6 and d.path_scale == 2.0 and d.has_out
    torch.manual_seed(3)
    d2 = autograd.make_drops(True, "cpu", 64, 32, 64, p_out=0.25, p_path=0.5, hidden=128, p_mid=0.1)
    assert torch.equal(d.out, d2.out) and torch.equal(d.path, d2.path) and torch.equal(d.mid, d2.mid)
    d3 = autograd.make_drops(True, "cpu", 4, 8, 16, p_path=0.25, scale_by_keep=False)
    assert d3.out is None and d3.mid is None and d3.path_scale == 1.0 and d3.has_out
    d4 = autograd.make_drops(True, "cpu", 4, 8, 16, hidden=32, p_mid=0.5)
    assert d4.mid is not None and not d4.has_out


def test_round6_host_logic_of_the_training_options():
    """Host side of the options round 6 closed: the masks' save / restore round trip of the sub-layer Functions (they travel through
    save_for_backward, the scales through ctx.meta), attention-dropout seeds from PyTorch's generator (reproducible under manual_seed,
    None in eval mode / at rate 0), which blocks make a checkpoint keep the generator state, and the constructor surface of
    latent_attn_dim / LayerScale in the cross-attention block / use_bn heads (the reference's state_dict keys)."""
    import torch
    from uniception_amd import autograd
    from uniception_amd.models.prediction_heads.dpt import DPTFeature, DPTSegmentationProcessor
    from uniception_amd.models.utils.checkpointing import has_random_masks
    from uniception_amd.models.utils.transformer_blocks import Attention, CrossAttentionBlock, SelfAttentionBlock
    torch.manual_seed(5)
    d = autograd.make_drops(True, "cpu", 4, 8, 16, p_out=0.25, p_path=0.5, hidden=32, p_mid=0.1)
    masks, spec = autograd._drops_saved(d)
    assert len(masks) == 3 and all(m.dtype == torch.uint8 for m in masks)
    x = torch.zeros(3)
    back, rest = autograd._drops_restore(spec, (x, x, *masks))
    assert rest == (x, x) and torch.equal(back.out, d.out) and torch.equal(back.path, d.path) and torch.equal(back.mid, d.mid)
    assert (back.out_scale, back.path_rows, back.path_scale, back.mid_scale) == (d.out_scale, d.path_rows, d.path_scale, d.mid_scale)
    assert autograd._drops_saved(None) == ((), None) and autograd._drops_restore(None, (x,)) == (None, (x,))
    d_path = autograd.make_drops(True, "cpu", 4, 8, 16, p_path=0.5)
    m2, s2 = autograd._drops_saved(d_path)
    b2, r2 = autograd._drops_restore(s2, (x, *m2))
    assert len(m2) == 1 and b2.out is None and b2.mid is None and torch.equal(b2.path, d_path.path) and r2 == (x,)
    # attention dropout: (p, seed) drawn from the CPU generator
    assert autograd.attn_dropout(False, 0.3) is None and autograd.attn_dropout(True, 0.0) is None
    torch.manual_seed(9)
    a = autograd.attn_dropout(True, 0.3)
    torch.manual_seed(9)
    b = autograd.attn_dropout(True, 0.3)
    c = autograd.attn_dropout(True, 0.3)
    assert a == b and a[0] == 0.3 and 0 <= a[1] < 2 ** 62 and c[1] != a[1]
    with pytest.raises(UcHipError):
        autograd.attn_dropout(True, 1.0)
    # which blocks consume the generator
    assert not has_random_masks(SelfAttentionBlock(dim=64, num_heads=1))
    for kw in (dict(proj_drop=0.1), dict(attn_drop=0.1), dict(drop_path=0.1)):
        assert has_random_masks(SelfAttentionBlock(dim=64, num_heads=1, **kw)), kw
        assert has_random_masks(CrossAttentionBlock(dim=64, num_heads=1, **kw)), kw
    # constructor surface: the reference's parameter shapes / state_dict keys
    att = Attention(128, latent_attn_dim=256, num_heads=4, qkv_bias=True)
    assert att.qkv.weight.shape == (768, 128) and att.proj.weight.shape == (128, 256) and att.head_dim == 64 and att.scale == 64 ** -0.5
    blk = CrossAttentionBlock(dim=64, num_heads=1, init_values=0.1)
    assert {"ls1.gamma", "ls2.gamma", "ls3.gamma"} <= set(blk.state_dict()) and blk._gammas()[0] is blk.ls1.gamma
    assert CrossAttentionBlock(dim=64, num_heads=1)._gammas() == (None, None, None)
    seg = DPTSegmentationProcessor(32, 5, hidden_dim=16, use_bn=True)
    assert {"conv.1.running_mean", "conv.1.weight"} <= set(seg.state_dict()) and seg.conv[0].bias is None
    feat = DPTFeature(patch_size=16, hooks=[0, 1, 2, 3], input_feature_dims=[32, 32, 32, 32], layer_dims=[16, 32, 64, 64], feature_dim=32, use_bn=True)
    keys = set(feat.state_dict())
    assert any(k.endswith("refinenet1.resConfUnit1.bn1.running_var") for k in keys) and not any(k.endswith("resConfUnit1.conv1.bias") for k in keys)
