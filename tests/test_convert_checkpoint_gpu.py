"""GPU half of the checkpoint-converter check: a model loaded from a converted original-format checkpoint
computes exactly what the directly filled model computes, through the HIP path."""
import pytest
import torch

from oracle import dust3r_oracle as O
from tests.golden.cases import GAINS
from tests.test_convert_checkpoint import _small
from uniception_amd.tools import convert_checkpoint as cc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("head", ["dpt", "linear"])
def test_converted_checkpoint_gives_identical_outputs(gpu, head):
    from uniception_amd import engine
    src = _small(head)
    O.fill_state_dict_(src.state_dict(), gains=GAINS)
    orig = cc.uniception_to_original({k: v.clone() for k, v in src.state_dict().items()})
    dst = _small(head)
    cc.load_original_checkpoint(dst, orig, strict=True)
    a, b = O.make_images(11, 2, 32, 48)
    v1 = {"img": a.to(gpu), "instance": ["0", "1"], "data_norm_type": "dust3r"}
    v2 = {"img": b.to(gpu), "instance": ["2", "3"], "data_norm_type": "dust3r"}
    src, dst = src.to(gpu), dst.to(gpu)
    for mode in ("fp32", "bf16"):
        with torch.no_grad(), engine.precision(mode):
            r1, r2 = src(v1, v2)
            s1, s2 = dst(v1, v2)
        assert torch.equal(r1["pts3d"], s1["pts3d"]) and torch.equal(r2["conf"], s2["conf"])
        assert torch.isfinite(r1["pts3d"]).all()
