"""CPU: the oracle restatement against the committed golden vectors (generated from the real reference by
tests/golden/make_golden.py).  Weights come from the name-keyed filler applied to the state_dict of the
uniception_amd modules, so this also proves their state_dict keys/shapes equal the reference's — any missing
or renamed parameter would change the weights and break parity."""
import pytest
import torch

from oracle import dust3r_oracle as O
from tests.golden.cases import CASES
from tests.helpers import build_case_model, case_images, compare_to_golden, load_golden


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_oracle_matches_reference_golden(name):
    torch.set_num_threads(max(1, torch.get_num_threads()))
    model, c = build_case_model(name)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    img1, img2 = case_images(c)
    collect = {}
    with torch.no_grad():
        o1, o2 = O.dust3r_forward(sd, img1, img2, head=c["head"], enc_depth=c["enc_depth"], enc_heads=c["enc_heads"],
                                  dec_depth=c["dec_depth"], dec_heads=c["dec_heads"], patch_size=c["patch"],
                                  indices=tuple(c["indices"]), collect=collect)
    tensors = dict(collect)
    tensors.update(pts3d_1=o1["pts3d"], conf_1=o1["conf"], pts3d_2=o2["pts3d_in_other_view"], conf_2=o2["conf"])
    worst = compare_to_golden(load_golden(name), tensors, c, tol=2e-5)
    print(f"{name}: worst {worst}")


def test_filler_is_name_keyed_and_deterministic():
    a = O.filler_tensor("encoder.enc_blocks.0.attn.qkv.weight", (6, 4))
    b = O.filler_tensor("encoder.enc_blocks.0.attn.qkv.weight", (6, 4))
    c = O.filler_tensor("encoder.enc_blocks.1.attn.qkv.weight", (6, 4))
    assert torch.equal(a, b) and not torch.equal(a, c)
    n = O.filler_tensor("encoder.enc_norm.weight", (1000,))
    assert abs(float(n.mean()) - 1.0) < 0.02  # LayerNorm gains are centred on 1


def test_rope_roundtrip_and_quarters():
    """fwd then inverse rotation is the identity; position 0 leaves tokens unchanged (curope2d.py:24-28)."""
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 3, 6, 64, generator=g)
    pos = O.grid_positions(2, 2, 3)
    r = O.rope2d(t, pos, 100.0, 1.0)
    back = O.rope2d(r, pos, 100.0, -1.0)
    assert (back - t).abs().max() < 1e-5
    assert torch.equal(r[:, :, 0], t[:, :, 0])  # token (0,0)
    # the x-half is untouched for tokens in column 0, the y-half for tokens in row 0
    assert torch.equal(r[:, :, 3, 32:], t[:, :, 3, 32:]) and torch.equal(r[:, :, 1, :32], t[:, :, 1, :32])
