"""hipGraph replay of the two-view forward equals the eager forward and follows new inputs."""
import pytest
import torch

from tests.helpers import build_case_model, case_images

pytestmark = pytest.mark.gpu


def test_graphed_forward_matches_eager(gpu):
    from uniception_amd import engine
    from uniception_amd.graphs import GraphedTwoView
    from uniception_amd.models.factory import DUSt3R
    from oracle import dust3r_oracle as O
    torch.manual_seed(0)
    model = DUSt3R(name="g", img_size=(64, 96), pred_head_type="linear")
    O.fill_state_dict_(model.state_dict())
    model = model.to(gpu).eval()
    g = torch.Generator().manual_seed(1)

    def views(seed):
        gg = torch.Generator().manual_seed(seed)
        a, b = torch.randn(2, 3, 64, 96, generator=gg).to(gpu), torch.randn(2, 3, 64, 96, generator=gg).to(gpu)
        return ({"img": a, "instance": ["a0", "a1"], "data_norm_type": "dust3r"}, {"img": b, "instance": ["b0", "b1"], "data_norm_type": "dust3r"})

    v1, v2 = views(1)
    graphed = GraphedTwoView(model, v1, v2, precision="bf16")
    for seed in (1, 2, 3):
        v1, v2 = views(seed)
        with torch.no_grad(), engine.precision("bf16"):
            e1, e2 = model(v1, v2)
        r1, r2 = graphed(v1, v2)
        torch.cuda.synchronize()
        assert torch.equal(r1["pts3d"], e1["pts3d"]) and torch.equal(r2["conf"], e2["conf"])
    with pytest.raises(ValueError):
        graphed({"img": torch.zeros(1, 3, 64, 96, device=gpu)}, v2)


def test_two_stream_branches_equal_single_stream(gpu):
    """Small batches run the two decoder branches of a depth level and the two heads on two HIP streams; the result must be
    bitwise the single-stream result, eagerly and inside a captured graph."""
    from uniception_amd import engine
    from uniception_amd.graphs import GraphedTwoView
    from uniception_amd.models.factory import DUSt3R
    from oracle import dust3r_oracle as O
    model = DUSt3R(name="g", img_size=(64, 96), pred_head_type="dpt")
    O.fill_state_dict_(model.state_dict())
    model = model.to(gpu).eval()
    gg = torch.Generator().manual_seed(4)
    a, b = torch.randn(1, 3, 64, 96, generator=gg).to(gpu), torch.randn(1, 3, 64, 96, generator=gg).to(gpu)
    v1 = {"img": a, "instance": ["a0"], "data_norm_type": "dust3r"}
    v2 = {"img": b, "instance": ["b0"], "data_norm_type": "dust3r"}
    saved = engine.BRANCH_TOKENS_MAX
    try:
        engine.BRANCH_TOKENS_MAX = 0
        with torch.no_grad(), engine.precision("bf16"):
            s1, s2 = model(v1, v2)
        engine.BRANCH_TOKENS_MAX = 1 << 20
        for _ in range(3):   # repeated: allocator reuse across streams must not corrupt anything
            with torch.no_grad(), engine.precision("bf16"):
                t1, t2 = model(v1, v2)
            torch.cuda.synchronize()
            assert torch.equal(t1["pts3d"], s1["pts3d"]) and torch.equal(t2["conf"], s2["conf"]) and torch.equal(t2["pts3d_in_other_view"], s2["pts3d_in_other_view"])
        graphed = GraphedTwoView(model, v1, v2, precision="bf16")
        g1, g2 = graphed(v1, v2)
        torch.cuda.synchronize()
        assert torch.equal(g1["pts3d"], s1["pts3d"]) and torch.equal(g2["conf"], s2["conf"])
    finally:
        engine.BRANCH_TOKENS_MAX = saved


def test_capture_with_the_gpu_still_behind_the_host_at_warm_up(gpu):
    """Prepared weights carry a completion event that consumers on other streams poll; with the GPU far behind the host (large
    batches: seen from 40 pairs on) those events are still pending when the capture starts, and an event query inside a capture
    invalidates it.  Here the GPU is kept busy while a FRESH model (empty caches) warms up and is captured."""
    from uniception_amd.graphs import GraphedTwoView
    from uniception_amd.models.factory import DUSt3R
    from oracle import dust3r_oracle as O
    model = DUSt3R(name="g", img_size=(64, 96), pred_head_type="dpt")
    O.fill_state_dict_(model.state_dict())
    model = model.to(gpu).eval()
    gg = torch.Generator().manual_seed(5)
    a, b = torch.randn(2, 3, 64, 96, generator=gg).to(gpu), torch.randn(2, 3, 64, 96, generator=gg).to(gpu)
    v1 = {"img": a, "instance": ["a0", "a1"], "data_norm_type": "dust3r"}
    v2 = {"img": b, "instance": ["b0", "b1"], "data_norm_type": "dust3r"}
    big = torch.randn(8192, 8192, device=gpu)
    torch.cuda.synchronize()
    for _ in range(60):          # ~0.5 s of queued fp32 GEMMs: the warm-up's launches (and their events) wait behind them
        big @ big
    graphed = GraphedTwoView(model, v1, v2, precision="bf16")
    r1, r2 = graphed(v1, v2)
    torch.cuda.synchronize()
    from uniception_amd import engine
    with torch.no_grad(), engine.precision("bf16"):
        e1, e2 = model(v1, v2)
    assert torch.equal(r1["pts3d"], e1["pts3d"]) and torch.equal(r2["conf"], e2["conf"])
