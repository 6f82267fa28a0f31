"""GPU parity of the nn.Module mirror against the golden vectors of the real reference.

fp32 mode (exact-fp32 kernels) must meet the north-star bar: 1e-3 relative on every captured tensor (it lands
around 1e-6..1e-5).  bf16 mode (MFMA operands bf16, fp32 accumulate, bf16 residual stream — the reference's own stream under
autocast) is held, tensor class by tensor class, to the error the reference itself shows under bf16 autocast (~1e-2 rel-L2 on
encoder/decoder features): features <= 1e-2, everything behind the heads' exp / expm1 <= 3e-2."""
import pytest
import torch

from tests.golden.cases import CASES
from tests.helpers import HEAD_OUTPUTS, build_case_model, case_images, compare_to_golden, load_golden, rel_l2

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3
FP32_MAX_ABS = 1e-2   # the reference's own gate asks for both (examples/models/dust3r/dust3r.py:230)
# bf16: encoder/decoder features ~1e-2; decoded channels pass through exp/expm1, which amplifies absolute error
BF16_TOL = {"enc_feat": 1e-2, "dec_final": 1e-2, "dec_take": 1e-2, "dpt_up8": 1.5e-2, "decoded": 2e-2, "default": 3e-2}


def bf16_tol(key):
    return next((v for k, v in BF16_TOL.items() if key.startswith(k)), BF16_TOL["default"])


def run_case(name, gpu, mode):
    from uniception_amd import engine

    model, c = build_case_model(name)
    model = model.to(gpu)
    img1, img2 = (t.to(gpu) for t in case_images(c))
    collect = {}
    with torch.no_grad(), engine.precision(mode):
        if c.get("factory"):
            v1 = {"img": img1, "instance": [str(i) for i in range(c["B"])], "data_norm_type": "dust3r"}
            v2 = {"img": img2, "instance": [str(100 + i) for i in range(c["B"])], "data_norm_type": "dust3r"}
            r1, r2 = model(v1, v2)
        else:
            r1, r2 = model(img1, img2, collect)
    torch.cuda.synchronize()
    tensors = dict(collect)
    tensors.update(pts3d_1=r1["pts3d"], conf_1=r1["conf"], pts3d_2=r2["pts3d_in_other_view"], conf_2=r2["conf"])
    for k in ("pts3d_1", "pts3d_2"):
        assert tensors[k].shape == (c["B"], c["img"][0], c["img"][1], 3) and tensors[k].is_contiguous()
    for k in ("conf_1", "conf_2"):
        assert tensors[k].shape == (c["B"], c["img"][0], c["img"][1], 1) and tensors[k].is_contiguous()
    return tensors, c


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_fp32_parity_1e3(gpu, name):
    tensors, c = run_case(name, gpu, "fp32")
    report = {}
    abs_report = {}
    worst = compare_to_golden(load_golden(name), tensors, c, tol=FP32_TOL, report=report, max_abs_tol=FP32_MAX_ABS, abs_report=abs_report)
    print(f"\n[fp32] {name}: worst {worst[0]} {worst[1]:.2e}; head outputs max-abs " +
          ", ".join(f"{k}={abs_report[k]:.1e}" for k in HEAD_OUTPUTS if k in abs_report))


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_bf16_parity(gpu, name):
    tensors, c = run_case(name, gpu, "bf16")
    report = {}
    abs_report = {}
    gold = load_golden(name)
    worst = ("", 0.0)
    for k, t in tensors.items():          # one bar per tensor class (not one loose bar for everything)
        w = compare_to_golden(gold, {k: t}, c, tol=bf16_tol(k), report=report, abs_report=abs_report)
        worst = max(worst, w, key=lambda x: x[1])
    print(f"\n[bf16] {name}: worst {worst[0]} {worst[1]:.2e}; " + ", ".join(f"{k}={v:.1e}" for k, v in sorted(report.items())) +
          "; head outputs max-abs " + ", ".join(f"{k}={abs_report[k]:.1e}" for k in HEAD_OUTPUTS if k in abs_report))


def test_autocast_selects_bf16_and_heads_follow_reference_policy(gpu):
    """Under torch.autocast the transformer runs the bf16 MFMA path; results equal engine.precision('bf16')."""
    from uniception_amd import engine

    model, c = build_case_model("tiny_linear")
    model = model.to(gpu)
    img1, img2 = (t.to(gpu) for t in case_images(c))
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert engine.compute_dtype() == torch.bfloat16
            a1, _ = model(img1, img2, {})
        assert engine.compute_dtype() == torch.float32
        with engine.precision("bf16"):
            b1, _ = model(img1, img2, {})
    assert torch.equal(a1["pts3d"], b1["pts3d"])


@pytest.mark.parametrize("name", ["vitl_linear_224", "vitl_dpt_512"])
def test_factory_under_autocast_equals_precision_bf16(gpu, name):
    """The FACTORY runs its heads under autocast(enabled=False) like the reference (factory/dust3r.py:288-309) and carries the
    transformer's dtype into that region (engine.ambient): bf16 that comes from torch.autocast must give the very kernels — and
    bits — of engine.precision('bf16'), for the DPT and the linear heads (the ambient dtype used to be read after
    autocast was already off, and the heads silently fell back to the exact-fp32 kernels)."""
    from uniception_amd import engine, ops

    model, c = build_case_model(name)
    model = model.to(gpu)
    img1, img2 = (t.to(gpu) for t in case_images(c))
    v1 = {"img": img1, "instance": [str(i) for i in range(c["B"])], "data_norm_type": "dust3r"}
    v2 = {"img": img2, "instance": [str(100 + i) for i in range(c["B"])], "data_norm_type": "dust3r"}
    seen = {"precision": [], "autocast": []}
    orig = ops.gemm

    def spy(tag):
        def f(a, w, *args, **kw):
            seen[tag].append((a.dtype, tuple(w.shape)))
            return orig(a, w, *args, **kw)
        return f
    try:
        with torch.no_grad():
            ops.gemm = spy("precision")
            with engine.precision("bf16"):
                b1, b2 = model(v1, v2)
            ops.gemm = spy("autocast")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                a1, a2 = model(v1, v2)
    finally:
        ops.gemm = orig
    assert seen["precision"] and seen["autocast"] == seen["precision"], "autocast(bf16) took other GEMM kernels than precision('bf16'): the heads lost the ambient dtype"
    assert torch.equal(a1["pts3d"], b1["pts3d"]) and torch.equal(a1["conf"], b1["conf"])
    assert torch.equal(a2["pts3d_in_other_view"], b2["pts3d_in_other_view"]) and torch.equal(a2["conf"], b2["conf"])


def test_symmetrized_batch_shortcut(gpu):
    """(a,b),(b,a) batches encode each image once and interleave (factory/dust3r.py:227-238): same outputs."""
    from uniception_amd.models.factory import DUSt3R
    from oracle import dust3r_oracle as O
    from tests.golden.cases import GAINS

    model = DUSt3R(name="s", img_size=(32, 48), pred_head_type="linear").eval()
    # shrink: keep the test light by using only the first blocks
    model.encoder.enc_blocks = model.encoder.enc_blocks[:2]
    model.info_sharing.depth = 2
    O.fill_state_dict_(model.state_dict(), gains=GAINS)
    model = model.to(gpu)
    a, b = O.make_images(5, 1, 32, 48)
    img1 = torch.cat([a, b]).to(gpu)
    img2 = torch.cat([b, a]).to(gpu)
    sym1 = {"img": img1, "instance": ["p", "q"], "data_norm_type": "dust3r"}
    sym2 = {"img": img2, "instance": ["q", "p"], "data_norm_type": "dust3r"}
    ns1 = {"img": img1, "instance": ["p", "q"], "data_norm_type": "dust3r"}
    ns2 = {"img": img2, "instance": ["x", "y"], "data_norm_type": "dust3r"}
    with torch.no_grad():
        s1, s2 = model(sym1, sym2)
        n1, n2 = model(ns1, ns2)
    assert torch.allclose(s1["pts3d"], n1["pts3d"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(s2["conf"], n2["conf"], rtol=1e-5, atol=1e-6)


def test_contiguous_nchw_inputs_are_accepted(gpu):
    """Decoder and heads accept plain contiguous NCHW tensors (what reference-style callers pass) as well as the
    channels-last views produced by the HIP encoder."""
    from uniception_amd.models.info_sharing.base import MultiViewTransformerInput

    model, c = build_case_model("tiny_linear")
    model = model.to(gpu)
    img1, img2 = (t.to(gpu) for t in case_images(c))
    from uniception_amd.models.encoders.base import ViTEncoderInput
    with torch.no_grad():
        f = model.encoder(ViTEncoderInput(image=torch.cat([img1, img2]), data_norm_type="dust3r")).features
        f1, f2 = f.chunk(2)
        a = model.info_sharing(MultiViewTransformerInput(features=[f1, f2]))
        b = model.info_sharing(MultiViewTransformerInput(features=[f1.contiguous(), f2.contiguous()]))
    assert torch.equal(a.features[0], b.features[0]) and torch.equal(a.features[1], b.features[1])
    assert not f1.is_contiguous() and f1.contiguous().is_contiguous()


def test_concurrent_streams_are_bitwise_the_single_stream_forward(gpu):
    """engine.CONCURRENT (large batches: the two views through the encoder, the two decoder branches and the two heads on
    concurrent HIP streams) changes the schedule, never a value: bitwise the result of the same sub-graphs run in stream order.  Also on the first calls, when weight-derived caches are built on one stream and read on another
    (engine.BuiltOn) and shape-keyed ones are built by the sequential warm-up call (run_branches(warm_key=...))."""
    from oracle import dust3r_oracle as O
    from tests.golden.cases import GAINS
    from uniception_amd import engine
    from uniception_amd.models.factory import DUSt3R
    model = DUSt3R(name="g", img_size=(64, 96), pred_head_type="dpt").eval()      # the factory model (ViT-L widths) on a 4x6 token grid
    O.fill_state_dict_(model.state_dict(), gain=1.0, gains=GAINS)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(99)
    B = 12
    v1 = {"img": torch.randn(B, 3, 64, 96, generator=g).to(gpu), "instance": [str(i) for i in range(B)], "data_norm_type": "dust3r"}
    v2 = {"img": torch.randn(B, 3, 64, 96, generator=g).to(gpu), "instance": [str(100 + i) for i in range(B)], "data_norm_type": "dust3r"}

    def run(on):
        prev = engine.BRANCH_TOKENS_MAX
        engine.BRANCH_TOKENS_MAX = 0 if on else prev       # 12 x 24 tokens count as a "large" batch: every fork goes through CONCURRENT
        try:
            with torch.no_grad(), engine.precision("bf16"), engine.concurrent(on):
                r1, r2 = model(v1, v2)
        finally:
            engine.BRANCH_TOKENS_MAX = prev
        torch.cuda.synchronize()
        return r1, r2

    engine.invalidate_prepared()
    first = run(True)          # cold caches; the warm-up call of every fork point
    second = run(True)         # forked everywhere
    engine.FORK_STREAMS = False                            # the same decomposition, the halves one after the other on one stream
    try:
        ref = run(True)
    finally:
        engine.FORK_STREAMS = True
    for got in (first, second):
        for a, b in zip(got, ref):
            for k in a:
                assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_pairs_are_independent_to_the_bit(gpu, mode):
    """Every image pair is an independent unit of the path: permuting the pairs of a batch permutes the outputs,
    and a pair's result does not depend on its batch mates or on where in a kernel's tile its rows land — bit for bit (the
    epilogue arithmetic is written with explicit FMAs for exactly this; tools/probe_batch_invariance.py)."""
    from oracle import dust3r_oracle as O
    from tests.golden.cases import GAINS
    from uniception_amd import engine
    from uniception_amd.models.factory import DUSt3R
    model = DUSt3R(name="g", img_size=(64, 96), pred_head_type="dpt").eval()
    O.fill_state_dict_(model.state_dict(), gain=1.0, gains=GAINS)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(123)
    B = 5
    img1 = torch.randn(B, 3, 64, 96, generator=g).to(gpu)
    img2 = torch.randn(B, 3, 64, 96, generator=g).to(gpu)

    def run(idx):
        v1 = {"img": img1[idx].contiguous(), "instance": [f"a{i}" for i in idx], "data_norm_type": "dust3r"}
        v2 = {"img": img2[idx].contiguous(), "instance": [f"b{i}" for i in idx], "data_norm_type": "dust3r"}
        with torch.no_grad(), engine.precision(mode):
            r1, r2 = model(v1, v2)
        torch.cuda.synchronize()
        return {**{"1" + k: v for k, v in r1.items()}, **{"2" + k: v for k, v in r2.items()}}

    full = run([0, 1, 2, 3, 4])
    perm = [3, 0, 4, 2, 1]
    shuffled = run(perm)
    alone = run([2])
    pair = run([4, 2])
    for k, v in full.items():
        assert torch.equal(shuffled[k], v[perm]), f"{k}: permuting the batch must permute the outputs"
        assert torch.equal(alone[k][0], v[2]), f"{k}: a pair alone"
        assert torch.equal(pair[k][1], v[2]) and torch.equal(pair[k][0], v[4]), f"{k}: a pair in another batch"


def test_bench_sized_batch_reproduces_the_golden_pair_to_the_bit(gpu):
    """The bench's model (ViT-L + DPT at 512x512) on a batch large enough for everything the 64-pair bench turns on — two kernel
    streams (> engine.BRANCH_TOKENS_MAX tokens), full rounds of the eight-wave GEMM tiles, the streaming (non-temporal) residual
    epilogues of outputs over 128 MB: the reference's golden pair, placed at two positions of a 20-pair batch, gives bit for bit
    what it gives alone, and that is the reference's result within the bf16 bar.  (Parity at the bench's size through a
    size-independent property: the goldens themselves are single pairs.)"""
    from uniception_amd import engine

    model, c = build_case_model("vitl_dpt_512")
    model = model.to(gpu)
    a, b = case_images(c)
    g = torch.Generator().manual_seed(77)
    n1, n2 = torch.randn(18, 3, 512, 512, generator=g), torch.randn(18, 3, 512, 512, generator=g)
    img1 = torch.cat([a, n1[:12], a, n1[12:]]).to(gpu)       # the golden pair at positions 0 and 13
    img2 = torch.cat([b, n2[:12], b, n2[12:]]).to(gpu)
    B = img1.shape[0]
    assert B * (512 // 16) ** 2 > engine.BRANCH_TOKENS_MAX and 2 * B * 1024 * 1024 * 4 > 128 << 20

    def run(i1, i2):
        n = i1.shape[0]
        v1 = {"img": i1, "instance": [f"a{i}" for i in range(n)], "data_norm_type": "dust3r"}
        v2 = {"img": i2, "instance": [f"b{i}" for i in range(n)], "data_norm_type": "dust3r"}
        with torch.no_grad(), engine.precision("bf16"):
            r1, r2 = model(v1, v2)
        torch.cuda.synchronize()
        return dict(pts3d_1=r1["pts3d"], conf_1=r1["conf"], pts3d_2=r2["pts3d_in_other_view"], conf_2=r2["conf"])

    from uniception_amd import ops
    # (three default choices follow the number of tiles of a launch, i.e. the batch size, and each changes a summation order: the small-M
    #  path sums K >= 2048 in two halves, a 3x3 conv with fewer eight-wave row tiles than CUs stays on the implicit-GEMM kernel, and an
    #  attention call with fewer than 512 (batch, head, 256-query) items stays on the eight-wave kernel.
    #  Pinned — small_m_split 0, conv_rows 3: the eight-wave conv kernel, attn_p64 2: the persistent attention kernel wherever the shape
    #  allows — a pair's bits do not depend on its batch)
    with ops.tuning("small_m_split", 0), ops.tuning("conv_rows", 3), ops.tuning("attn_p64", 2):
        alone = run(img1[:1], img2[:1])
    alone_split = run(img1[:1], img2[:1])
    diffs = {k: rel_l2(alone_split[k].float().cpu(), v.float().cpu()) for k, v in alone.items()}
    print("\n[bf16] one pair, K >= 2048 summed in two halves vs in one chain: " + ", ".join(f"{k}={v:.1e}" for k, v in sorted(diffs.items())))
    assert max(diffs.values()) < BF16_TOL["default"]        # two bf16 roundings of the same forward: apart by what each is from fp32
    with ops.tuning("small_m_split", 0), ops.tuning("conv_rows", 3), ops.tuning("attn_p64", 2):      # (at 20 pairs the heads' smallest maps still make small-M launches)
        batch = run(img1, img2)
        again = run(img1, img2)            # second call: every fork point past its warm-up call
    for k, v in alone.items():
        for pos in (0, 13):
            assert torch.equal(batch[k][pos], v[0]), f"{k}: the golden pair at position {pos} of a {B}-pair batch differs from the pair alone"
        assert torch.equal(again[k], batch[k]), k
    report = {}
    compare_to_golden(load_golden("vitl_dpt_512"), {k: v[13:14] for k, v in batch.items()}, c, tol=BF16_TOL["default"], report=report)
    print("\n[bf16] golden pair inside a 20-pair 512x512 batch: " + ", ".join(f"{k}={v:.1e}" for k, v in sorted(report.items())))


@pytest.mark.parametrize("name", ["tiny_dpt", "vitl_dpt_512"])
def test_bf16_mode_with_an_fp32_residual_stream(gpu, name):
    """engine.bf16_stream(False): bf16 operands with the residual stream kept in fp32 (round 1's policy, more accurate than the
    reference's bf16 stream) still meets the bf16 bar, and lands closer to the fp32 reference than the default on the features."""
    from uniception_amd import engine
    rep = {}
    for on in (True, False):
        with engine.bf16_stream(on):
            tensors, c = run_case(name, gpu, "bf16")
        r = {}
        compare_to_golden(load_golden(name), tensors, c, tol=BF16_TOL["default"], report=r)
        rep[on] = r
    keys = sorted(rep[True])
    print(f"\n[bf16 stream vs fp32 stream] {name}: " + ", ".join(f"{k}={rep[True][k]:.1e}/{rep[False][k]:.1e}" for k in keys))


def test_round6_shape_routes_inside_the_model_at_224(gpu):
    """The two routes round 6 added, inside the factory ViT-L + DPT model at 224 x 224 (196 tokens per view: the packed-VT epilogue's
    token-quad stores; 56 / 112 / 224-wide head maps: the eight-wave 3x3 kernel's FLAT form once a launch has a CU's worth of tiles)
    against the routes they replace, same weights, same pairs, bf16 transformer + TF32-class heads: `conv_rows_flat` 0 (implicit-GEMM
    convolutions: the same products in another summation order) and the fp32 verification mode as the reference for both."""
    from uniception_amd import engine, ops
    from uniception_amd.models.factory import DUSt3R
    torch.manual_seed(0)
    model = DUSt3R(name="t224", img_size=(224, 224), pred_head_type="dpt").to(gpu).eval()
    g = torch.Generator().manual_seed(5)
    B = 6       # 6 pairs: the 224 x 224 128 -> 128 convolution of a head is 6 x 98 = 588 tiles of 512 pixels (>= 256: the default route)
    v1 = {"img": torch.randn(B, 3, 224, 224, generator=g).to(gpu), "instance": [str(i) for i in range(B)], "data_norm_type": "dust3r"}
    v2 = {"img": torch.randn(B, 3, 224, 224, generator=g).to(gpu), "instance": [str(100 + i) for i in range(B)], "data_norm_type": "dust3r"}

    def run(mode):
        with torch.no_grad(), engine.precision(mode):
            r1, r2 = model(v1, v2)
        torch.cuda.synchronize()
        return torch.cat([r1["pts3d"].float().flatten(), r1["conf"].float().flatten(), r2["pts3d_in_other_view"].float().flatten()])
    flat = run("bf16")
    with ops.tuning("conv_rows_flat", 0):
        implicit = run("bf16")
    exact = run("fp32")
    assert torch.isfinite(flat).all()
    e_routes = rel_l2(flat, implicit)
    e_flat, e_impl = rel_l2(flat, exact), rel_l2(implicit, exact)
    print(f"\n[224 x 224, 6 pairs] flat vs implicit-GEMM convolutions {e_routes:.2e}; vs fp32 mode: flat {e_flat:.2e}, implicit {e_impl:.2e}")
    assert not torch.equal(flat, implicit) and e_routes < 2e-3          # (fp16 maps: one rounding step of difference here and there)
    assert e_flat < 3e-2 and abs(e_flat - e_impl) < 3e-3                # the bf16 mode's distance to the exact arithmetic, either route
