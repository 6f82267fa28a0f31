"""CPU: checkpoint converter.  Real DUSt3R checkpoints are unreachable here, so the maps are exercised on
synthetic original-format checkpoints: a filled factory model is written with the ORIGINAL key names (the inverse map),
converted back and loaded strictly into a fresh model; the converted per-module checkpoints load strictly into the modules.
The GPU half (equal forward outputs through the HIP path) is tests/test_convert_checkpoint_gpu.py."""
import pytest
import torch

from oracle import dust3r_oracle as O
from tests.golden.cases import GAINS
from uniception_amd.models.factory import DUSt3R
from uniception_amd.tools import convert_checkpoint as cc


def _small(head):
    m = DUSt3R(name="c", img_size=(32, 48), pred_head_type=head).eval()
    m.encoder.enc_blocks = m.encoder.enc_blocks[:2]
    for br in m.info_sharing.multi_view_branches:
        del br[2:]
    m.info_sharing.depth = 2
    m.info_sharing.indices = [0, 1]     # the factory taps blocks 5 and 8 of its 12
    return m


@pytest.mark.parametrize("head", ["dpt", "linear"])
def test_round_trip_strict_load(head):
    src = _small(head)
    O.fill_state_dict_(src.state_dict(), gains=GAINS)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    orig = cc.uniception_to_original(sd)
    # the synthetic checkpoint carries the ORIGINAL names only, plus entries a real one has and the path does not use
    assert all(k.startswith(("patch_embed.", "enc_blocks.", "enc_norm.", "decoder_embed.", "dec_blocks.", "dec_blocks2.", "dec_norm.",
                             "downstream_head1.", "downstream_head2.")) for k in orig)
    assert cc.detect_head_type(orig) == head
    orig["mask_token"] = torch.zeros(1, 1, 768)
    if head == "dpt":
        assert "downstream_head1.dpt.act_postprocess.0.0.weight" in orig and "downstream_head2.dpt.head.4.bias" in orig
        assert "downstream_head1.dpt.scratch.layer1_rn.weight" in orig
        orig["downstream_head1.dpt.scratch.refinenet4.resConfUnit1.conv1.weight"] = torch.zeros(256, 256, 3, 3)
        orig["downstream_head1.head_local_features.0.weight"] = torch.zeros(4, 4)       # MASt3R extra
    else:
        assert orig["downstream_head1.proj.weight"].shape == (4 * 16 * 16, 768)          # nn.Linear layout in the original
    conv, dropped = cc.original_to_uniception(orig)
    assert "mask_token" in dropped and (head != "dpt" or len(dropped) == 3)
    dst = _small(head)
    res = dst.load_state_dict(conv, strict=True)          # every key incl. the aliases
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in dst.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # per-module checkpoints load into the modules
    mods = cc.split_modules(conv)
    assert dst.encoder.load_state_dict(mods["encoder"]["model"], strict=True) and mods["encoder"]["data_norm_type"] == "dust3r"
    dst.info_sharing.load_state_dict(mods["info_sharing"]["model"], strict=True)
    if head == "dpt":
        dst.dpt_feature_head1.load_state_dict(mods["dpt_feature_head1"]["model"], strict=True)
        dst.dpt_regressor_head2.load_state_dict(mods["dpt_regressor_head2"]["model"], strict=True)
    else:
        dst.head1.load_state_dict(mods["linear_feature_head1"]["model"], strict=True)


def test_croco_checkpoint_duplicates_the_single_decoder():
    src = _small("linear")
    O.fill_state_dict_(src.state_dict(), gains=GAINS)
    orig = cc.uniception_to_original(src.state_dict(), two_decoders=False)
    assert not any(k.startswith("dec_blocks2.") for k in orig)
    conv, _ = cc.original_to_uniception(orig)
    for k, v in conv.items():
        if k.startswith("info_sharing.multi_view_branches.1."):
            assert torch.equal(v, conv[k.replace("branches.1.", "branches.0.")])
    assert any(k.startswith("info_sharing.multi_view_branches.1.") for k in conv)


def test_unknown_keys_are_errors():
    with pytest.raises(KeyError):
        cc.original_to_uniception({"something_else.weight": torch.zeros(1)})


def _replay(head):
    """The original-layout checkpoint of tests/golden/keymap.json: every tensor filled with its id (expanded scalars, no memory)."""
    import json
    import os
    from tests.helpers import GOLDEN_DIR
    fx = json.load(open(os.path.join(GOLDEN_DIR, "keymap.json")))
    names = [k for k, _ in fx["original"][head]]
    orig = {k: torch.full((), float(i + 1)).expand(tuple(shape)) for i, (k, shape) in enumerate(fx["original"][head])}
    return fx, names, orig


def _same_map(got_sd, want, names):
    got = {k: v for k, v in got_sd.items() if torch.is_floating_point(v) and v.numel()}
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:8]
    for k, (src, shape) in want.items():
        v = got[k]
        assert list(v.shape) == shape, (k, tuple(v.shape), shape)
        i = int(round(float(v.reshape(-1)[0])))
        assert names[i - 1] == src and bool((v == float(i)).all()), (k, names[i - 1], src)


def test_key_map_equals_the_reference_scripts():
    """tests/golden/keymap.json was written by the REFERENCE's conversion script run on a synthetic original-layout checkpoint
    (tests/golden/make_golden_keymap.py): which original tensor lands under which UniCeption key of the cross-attention
    transformer, the DPT regression processors and the linear heads.  The converter here must produce the same map.  (The
    script's DPTFeature half does not load into the reference's current DPTFeature — recorded in the fixture; that part of
    the map is covered by the strict loads of test_round_trip_strict_load.)"""
    fx, names, orig = _replay("dpt")
    assert fx["dpt_feature_script"].startswith("reference script fails on its own DPTFeature")
    conv, _ = cc.original_to_uniception(orig)
    mods = cc.split_modules(conv)
    _same_map(mods["info_sharing"]["model"], fx["modules"]["cross_attn_transformer"], names)
    for h in ("1", "2"):
        _same_map(mods[f"dpt_regressor_head{h}"]["model"], fx["modules"][f"dpt_reg_processor{h}"], names)
    fx, names, orig = _replay("linear")
    conv, _ = cc.original_to_uniception(orig)
    mods = cc.split_modules(conv)
    for h in ("1", "2"):
        _same_map(mods[f"linear_feature_head{h}"]["model"], fx["modules"][f"linear_feature_head{h}"], names)


def test_original_dpt_weights_through_the_converter_match_the_original_module_on_cpu():
    """CPU half of the numeric pin of the DPT key map (the GPU half: test_convert_checkpoint_gpu.py): the weights of
    tests/golden/dpt_original.npz — ORIGINAL checkpoint names, output of the reference's `DPTOutputAdapter`
    (libs/croco/dpt_block.py:326-530) — converted and evaluated by the oracle's DPT restatement (which is keyed by the UniCeption
    names) reproduce that output.  A wrong map entry cannot pass: the oracle reads every converted key by name."""
    import os

    import numpy as np

    from oracle import dust3r_oracle as O
    from tests.golden.dpt_original_case import DPT_ORIGINAL as C
    from tests.helpers import GOLDEN_DIR, rel_l2
    gold = np.load(os.path.join(GOLDEN_DIR, "dpt_original.npz"))
    orig = {k[2:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w/")}
    conv, dropped = cc.original_to_uniception(orig)
    assert len(dropped) == 4 and all("refinenet4.resConfUnit1" in k for k in dropped)
    h, w = C["img"][0] // C["patch"], C["img"][1] // C["patch"]
    feats = [torch.from_numpy(gold[f"tokens{i}"]).transpose(1, 2).reshape(C["B"], -1, h, w) for i in range(4)]
    with torch.no_grad():
        up8 = O.dpt_feature(feats, conv, "dpt_feature_head1.")
        out = O.dpt_regressor(up8, tuple(C["img"]), conv, "dpt_regressor_head1.")
    assert rel_l2(out, gold["out"]) < 1e-5
    # ... and through the alias names the factory registers (head1.0.* / head1.1.*)
    with torch.no_grad():
        out2 = O.dpt_regressor(O.dpt_feature(feats, conv, "head1.0."), tuple(C["img"]), conv, "head1.1.")
    assert torch.equal(out, out2)
