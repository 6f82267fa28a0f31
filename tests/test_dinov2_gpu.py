"""DINOv2 encoder (BASELINE config 4) on the HIP kernels vs the oracle's restatement of the published architecture.
PARITY UNPINNED: the reference fetches this network from torch.hub (not vendored, no network here), so there is no reference
output to pin either side to; these tests only prove that the HIP module and the CPU restatement implement the same
reading of the architecture (cls token, resized position embedding, registers, LayerScale, output split)."""
import pytest
import torch

from oracle import dust3r_oracle as O
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu

GAINS = {"pos_embed": 300.0, "cls_token": 10.0, "register_tokens": 20.0}


def build(size, with_registers, layers):
    from uniception_amd.models.encoders import encoder_factory
    enc = encoder_factory("dinov2", name="d", size=size, with_registers=with_registers, keep_first_n_layers=layers).eval()
    O.fill_state_dict_(enc.state_dict(), gains=GAINS)
    return enc


@pytest.mark.parametrize("size,regs,hw", [("small", False, (70, 98)), ("small", True, (70, 98)), ("base", False, (518, 518)),
                                          ("large", True, (112, 84))])
def test_fp32_matches_restatement(gpu, size, regs, hw):
    from uniception_amd import engine
    from uniception_amd.models.encoders.base import ViTEncoderInput
    enc = build(size, regs, 2)
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, *hw, generator=g)
    heads = {"small": 6, "base": 12, "large": 16}[size]
    with torch.no_grad():
        f_ref, r_ref = O.dinov2_encoder(img, sd, "model.", num_heads=heads, num_registers=4 if regs else 0)
        with engine.precision("fp32"):
            out = enc.to(gpu)(ViTEncoderInput(image=img.to(gpu), data_norm_type="dinov2"))
    assert out.features.shape == f_ref.shape and out.registers.shape == r_ref.shape
    assert rel_l2(out.features.cpu(), f_ref) < 1e-3 and rel_l2(out.registers.cpu(), r_ref) < 1e-3
    print(f"\n[dinov2 fp32] {size} regs={regs} {hw}: features {rel_l2(out.features.cpu(), f_ref):.2e}, registers {rel_l2(out.registers.cpu(), r_ref):.2e}")


def test_bf16_and_intermediate_returner(gpu):
    from uniception_amd import engine
    from uniception_amd.models.encoders import feature_returner_encoder_factory
    from uniception_amd.models.encoders.base import ViTEncoderInput
    enc = feature_returner_encoder_factory("dinov2", name="d", size="small", keep_first_n_layers=3, indices=[0, 2]).eval()
    O.fill_state_dict_(enc.state_dict(), gains=GAINS)
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    img = torch.randn(1, 3, 56, 70, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        _, _, taken = O.dinov2_encoder(img, sd, "model.", num_heads=6, take=(0, 2))
        for mode, tol in (("fp32", 1e-3), ("bf16", 3e-2)):
            with engine.precision(mode):
                outs = enc.to(gpu)(ViTEncoderInput(image=img.to(gpu), data_norm_type="dinov2"))
            assert len(outs) == 2
            for o, t in zip(outs, taken):
                assert o.features.shape == (1, 384, 4, 5) and o.registers.shape == (1, 384, 1)
                assert rel_l2(o.features.cpu(), t[:, 1:].permute(0, 2, 1).reshape(1, 384, 4, 5)) < tol
                assert rel_l2(o.registers.cpu(), t[:, :1].permute(0, 2, 1)) < tol
