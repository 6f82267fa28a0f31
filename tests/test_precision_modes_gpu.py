"""Precision modes beside the bf16 headline: fp32-class arithmetic on the matrix pipe.

* ops level: a split-operand (bf16x3) GEMM / 3x3 convolution reproduces the fp32 product to ~1e-6;
* model level: `engine.precision("bf16x3")` (every GEMM / conv split, fp32 attention) meets the north-star gate — rel-L2 < 1e-3 AND
  max-abs < 1e-2 on the head outputs against the reference goldens — on the full-size ViT-L + DPT 512x512 model;
* reference policy (`set_head_precision("fp32")` next to a bf16 transformer, what factory/dust3r.py:288-309 does under autocast):
  heads in fp32-class arithmetic; its error is the bf16 transformer's and is reported, not gated at 1e-3."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import HEAD_OUTPUTS, build_case_model, case_images, compare_to_golden, load_golden, rel_l2

pytestmark = pytest.mark.gpu


def test_split_gemm_and_conv_match_fp32(gpu):
    from uniception_amd import engine, ops
    g = torch.Generator().manual_seed(21)
    a = torch.randn(300, 256, generator=g)
    w = torch.randn(200, 256, generator=g) / 16
    b = torch.randn(200, generator=g)
    res = torch.randn(300, 200, generator=g)
    ref = (a.double() @ w.double().t() + b.double())
    with engine.precision("bf16x3"):
        y = ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu))
        y2 = ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), act="relu", residual=res.to(gpu))
    assert y.dtype == torch.float32 and rel_l2(y.cpu(), ref) < 5e-6
    assert rel_l2(y2.cpu(), F.relu(ref) + res.double()) < 5e-6
    a3 = ops.split_bf16x3(a.to(gpu), relu=True).cpu().float()
    assert torch.equal(a3[:, :256], a3[:, 256:512]) and rel_l2(a3[:, :256] + a3[:, 512:], F.relu(a)) < 1e-5
    # 3x3 convolution with ReLU-on-load, stride 1 and 2
    for (B, H, W, Cin, Cout, s) in [(2, 9, 11, 32, 40, 1), (1, 12, 8, 64, 24, 2)]:
        x = torch.randn(B, H, W, Cin, generator=g)
        wc = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        bc = torch.randn(Cout, generator=g)
        refc = F.conv2d(F.relu(x.permute(0, 3, 1, 2)).double(), wc.double(), bc.double(), stride=s, padding=1).permute(0, 2, 3, 1)
        wg = wc.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
        with engine.precision("bf16x3"):
            yc = ops.gemm(x.to(gpu), wg.to(gpu), bc.to(gpu), relu_a=True, conv=(B, H, W, Cin, s))
        assert rel_l2(yc.view(refc.shape).cpu(), refc) < 5e-6


@pytest.mark.parametrize("shape", [(2, 3, 196, 196), (1, 2, 1024, 1024), (2, 2, 150, 333), (1, 1, 64, 4096), (3, 2, 40, 8)])
def test_split_operand_attention_matches_fp64(gpu, shape):
    """uc_attention_fwd_x3 — both products of softmax(QK^T)V as three bf16 MFMA products of split operands, fp32 softmax — against
    the fp64 product: fp32-class (<= 3e-5), where the bf16 kernel is at 4e-3; strided q / k / v views of one fused qkv buffer, ragged
    and cross-attention key counts, peaked scores (a large scale), and agreement with the exact fp32 VALU kernel it replaces in
    engine.precision("bf16x3") (libs/croco/blocks.py:123-125, utils/transformer_blocks.py:244-246, 373-375)."""
    from uniception_amd import ops
    B, H, Nq, Nk = shape
    g = torch.Generator().manual_seed(sum(shape))
    if Nq == Nk:
        qkv = torch.randn(B, Nq, 3, H, 64, generator=g).to(gpu)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = torch.randn(B, Nq, H, 64, generator=g).to(gpu)
        kv = torch.randn(B, Nk, 2, H, 64, generator=g).to(gpu)
        k, v = kv[:, :, 0], kv[:, :, 1]
    for scale in (0.125, 1.0):          # 1.0: softmax dominated by a few keys (scores ~ N(0, 64))
        o = ops.attention_x3(q, k, v, scale)
        assert o.shape == (B, Nq, H, 64) and o.dtype == torch.float32 and o.is_contiguous()
        qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
        ref = (torch.softmax(qd @ kd.transpose(-1, -2) * scale, -1) @ vd).permute(0, 2, 1, 3)
        err = rel_l2(o.cpu(), ref.cpu())
        exact = ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), scale)
        print(f"[x3 attention] {shape} scale {scale}: rel-L2 {err:.2e} (exact fp32 kernel {rel_l2(exact.cpu(), ref.cpu()):.2e})")
        assert err < 3e-5
        assert float((o - ref.float()).abs().max()) < 2e-4 * float(ref.abs().max())
    # RoPE-2D fused into the operand split (ABI 10): the same bits as rotating q and k in place first (rope_2d_, the same cos / sin table)
    pq = torch.stack([torch.randint(0, 40, (B, Nq), generator=g), torch.randint(0, 50, (B, Nq), generator=g)], -1).to(gpu)
    pk = torch.stack([torch.randint(0, 40, (B, Nk), generator=g), torch.randint(0, 50, (B, Nk), generator=g)], -1).to(gpu)
    table = ops.rope_table(gpu, 64, 100.0, 1.0)
    fused = ops.attention_x3(q, k, v, 0.125, rope=(pq.reshape(-1, 2).contiguous(), pk.reshape(-1, 2).contiguous(), table))
    qr, kr = q.contiguous().clone(), k.contiguous().clone()
    ops.rope_2d_(qr, pq, 100.0, 1.0)
    ops.rope_2d_(kr, pk, 100.0, 1.0)
    two_pass = ops.attention_x3(qr, kr, v, 0.125)
    assert rel_l2(fused, two_pass) < 1e-6 and float((fused - two_pass).abs().max()) < 1e-5
    qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (qr, kr, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * 0.125, -1) @ vd).permute(0, 2, 1, 3)
    assert rel_l2(fused.cpu(), ref.cpu()) < 3e-5


def test_fp16_operand_gemm_and_conv_are_tf32_class(gpu):
    """uc_gemm with compute_dtype UC_F16 (the heads' TF32-class mode): fp16 MFMA operands, fp32 accumulate.  Against the fp64 product
    of the SAME fp16-rounded operands the kernel is exact to fp32 accumulation (<= 2e-6); against the fp32 operands it carries the
    operand rounding 2^-11 (TF32's) — 8x below bf16's 2^-8.  Dense (bias + GELU, fp32 out, fp16 residual), 3x3 convolution with
    ReLU-on-load / residual / stride 2, and the fused 128 -> 4 tail."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(31)
    a = torch.randn(520, 256, generator=g)
    w = torch.randn(328, 256, generator=g) / 16
    b = torch.randn(328, generator=g)
    res = torch.randn(520, 328, generator=g)
    a16, w16, r16 = a.half(), w.half(), res.half()
    exact = a16.double() @ w16.double().t() + b.double()
    y = ops.gemm(a16.to(gpu), w16.to(gpu), b.to(gpu), out_dtype=torch.float32)
    assert y.dtype == torch.float32 and rel_l2(y.cpu(), exact) < 2e-6
    y16 = ops.gemm(a16.to(gpu), w16.to(gpu), b.to(gpu), act="gelu")
    assert y16.dtype == torch.float16 and rel_l2(y16.float().cpu(), F.gelu(exact)) < 4e-4            # + one fp16 rounding of the output
    yr = ops.gemm(a16.to(gpu), w16.to(gpu), b.to(gpu), residual=r16.to(gpu), out_dtype=torch.float16)
    assert rel_l2(yr.float().cpu(), exact + r16.double()) < 4e-4
    true = a.double() @ w.double().t() + b.double()
    e16 = rel_l2(y.cpu(), true)
    ebf = rel_l2(ops.gemm(a.bfloat16().to(gpu), w.bfloat16().to(gpu), b.to(gpu), out_dtype=torch.float32).cpu(), true)
    print(f"\n[fp16 operands] dense: {e16:.2e} against the fp32 operands (bf16 operands: {ebf:.2e})")
    assert e16 < 5e-4 and e16 < ebf / 5
    for (B, H, W, Cin, Cout, s_) in [(2, 9, 11, 32, 40, 1), (1, 12, 8, 64, 24, 2), (1, 16, 16, 96, 64, 1)]:
        x = torch.randn(B, H, W, Cin, generator=g)
        wc = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        bc = torch.randn(Cout, generator=g)
        x16, wc16 = x.half(), wc.half()
        refc = F.conv2d(F.relu(x16.permute(0, 3, 1, 2)).double(), wc16.double(), bc.double(), stride=s_, padding=1).permute(0, 2, 3, 1)
        wg = wc16.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
        yc = ops.gemm(x16.to(gpu), wg.to(gpu), bc.to(gpu), relu_a=True, conv=(B, H, W, Cin, s_), out_dtype=torch.float32)
        assert rel_l2(yc.view(refc.shape).cpu(), refc) < 2e-6
    # fused tail: conv3x3(Cin -> 128) -> ReLU -> 1x1(128 -> 4)
    B, H, W, Cin = 2, 16, 24, 128
    x = torch.randn(B, H, W, Cin, generator=g).half()
    wc = (torch.randn(128, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).half()
    bc = torch.randn(128, generator=g) * 0.3
    w4, b4 = torch.randn(4, 128, generator=g) / math.sqrt(128), torch.randn(4, generator=g)
    yt = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wc.double(), bc.double(), padding=1))
    reft = torch.einsum("bchw,oc->bhwo", yt, w4.double()) + b4.double()
    out = ops.gemm(x.to(gpu), wc.permute(0, 2, 3, 1).reshape(128, -1).contiguous().to(gpu), bc.to(gpu), act="relu", conv=(B, H, W, Cin, 1),
                   tail=(w4.to(gpu), b4.to(gpu)))
    assert out.dtype == torch.float32 and rel_l2(out.view(B, H, W, 4).cpu(), reft) < 1e-5


def test_fp16_heads_are_tf32_class_against_exact_fp32_heads(gpu):
    """engine.set_head_precision("fp16") — the reference's fp32 heads on TF32-class arithmetic (fp16 MFMA operands, fp32 accumulate;
    TF32 is what "fp32" convolutions / linears run on in the reference's own environment, libs/croco/blocks.py:15) — on the SAME bf16
    transformer features as the exact-fp32 heads: the four outputs of the full-size ViT-L + DPT 512x512 model agree to ~1e-3 where
    the bf16 heads are at ~1e-2 (VERDICT r2 next #3b: reported and gated)."""
    exact, c = _run("vitl_dpt_512", gpu, "bf16", "fp32_exact")
    h16, _ = _run("vitl_dpt_512", gpu, "bf16", "fp16")
    hbf, _ = _run("vitl_dpt_512", gpu, "bf16", "follow")
    x3, _ = _run("vitl_dpt_512", gpu, "bf16", "fp32")
    e16 = {k: rel_l2(h16[k].cpu(), exact[k].cpu()) for k in HEAD_OUTPUTS}
    ebf = {k: rel_l2(hbf[k].cpu(), exact[k].cpu()) for k in HEAD_OUTPUTS}
    ex3 = {k: rel_l2(x3[k].cpu(), exact[k].cpu()) for k in HEAD_OUTPUTS}
    print("\n[heads only, same bf16 features, vs exact fp32 heads] fp16 heads " + ", ".join(f"{k}={v:.1e}" for k, v in e16.items()) +
          " | bf16 heads " + ", ".join(f"{k}={v:.1e}" for k, v in ebf.items()) + " | bf16x3 heads " + ", ".join(f"{k}={v:.1e}" for k, v in ex3.items()))
    for k in HEAD_OUTPUTS:
        assert e16[k] < 3e-3 and e16[k] < ebf[k] / 3, (k, e16[k], ebf[k])
        assert ex3[k] < 1e-4, (k, ex3[k])          # split-operand heads == exact fp32 heads to fp32-class accuracy
    # end to end against the reference golden: the bf16 transformer's error class
    rep = {}
    compare_to_golden(load_golden("vitl_dpt_512"), h16, c, tol=3e-2, report=rep)
    print("[bf16 transformer + fp16 heads vs reference golden] " + ", ".join(f"{k}={rep[k]:.1e}" for k in HEAD_OUTPUTS))
    # and the small models incl. the linear head
    for name in ("tiny_dpt", "tiny_linear", "tiny_dpt_odd"):
        t, cc = _run(name, gpu, "bf16", "fp16")
        compare_to_golden(load_golden(name), t, cc, tol=3e-2)


def _run(name, gpu, mode, head_mode):
    from uniception_amd import engine
    model, c = build_case_model(name)
    model = model.to(gpu)
    img1, img2 = (t.to(gpu) for t in case_images(c))
    with engine.head_precision(head_mode):
        with torch.no_grad(), engine.precision(mode):
            if c.get("factory"):
                v1 = {"img": img1, "instance": [str(i) for i in range(c["B"])], "data_norm_type": "dust3r"}
                v2 = {"img": img2, "instance": [str(100 + i) for i in range(c["B"])], "data_norm_type": "dust3r"}
                r1, r2 = model(v1, v2)
            else:
                r1, r2 = model(img1, img2, {})
    torch.cuda.synchronize()
    return dict(pts3d_1=r1["pts3d"], conf_1=r1["conf"], pts3d_2=r2["pts3d_in_other_view"], conf_2=r2["conf"]), c


@pytest.mark.parametrize("name", ["tiny_dpt", "tiny_linear", "vitl_dpt_512"])
def test_bf16x3_everything_meets_the_reference_gate(gpu, name):
    tensors, c = _run(name, gpu, "bf16x3", "follow")
    report, abs_report = {}, {}
    compare_to_golden(load_golden(name), tensors, c, tol=1e-3, report=report, max_abs_tol=1e-2, abs_report=abs_report)
    print(f"\n[bf16x3] {name}: rel-L2 " + ", ".join(f"{k}={report[k]:.1e}" for k in HEAD_OUTPUTS) +
          "; max-abs " + ", ".join(f"{k}={abs_report[k]:.1e}" for k in HEAD_OUTPUTS))


@pytest.mark.parametrize("name", ["tiny_dpt"])     # (the full-size model: test_fp16_heads_are_tf32_class_against_exact_fp32_heads compares all head policies on it)
def test_reference_policy_bf16_transformer_fp32_class_heads(gpu, name):
    ref_pol, c = _run(name, gpu, "bf16", "fp32")
    follow, _ = _run(name, gpu, "bf16", "follow")
    exact, _ = _run(name, gpu, "bf16", "fp32_exact")
    g = load_golden(name)
    rep_pol, rep_fol = {}, {}
    compare_to_golden(g, ref_pol, c, tol=4e-2, report=rep_pol)
    compare_to_golden(g, follow, c, tol=4e-2, report=rep_fol)
    # split-operand heads == exact fp32 heads on the same bf16 features, to fp32-class accuracy
    for k in HEAD_OUTPUTS:
        assert rel_l2(ref_pol[k].cpu(), exact[k].cpu()) < 1e-4, k   # ~2^-16 per product through ~15 chained convolutions
    print(f"\n[reference policy] {name}: bf16 transformer + fp32-class heads " + ", ".join(f"{k}={rep_pol[k]:.1e}" for k in HEAD_OUTPUTS) +
          " | bf16 heads " + ", ".join(f"{k}={rep_fol[k]:.1e}" for k in HEAD_OUTPUTS))


def test_fp16_heads_range_guard_saturates_reports_and_falls_back(gpu):
    """VERDICT r3 #7 / ADVICE r3: fp16 operands carry TF32's mantissa, not its exponent.  A DPT head whose scratch maps reach 1e5-1e6
    (the layer_rn convolutions scaled up by F, the last 1x1 convolution down by 1/F: the head is positively homogeneous up to its biases)
    must not turn into inf / NaN under the default TF32-class policy: the fp16 stores saturate, the launches raise the device flag,
    engine.head_range_exceeded() reports it, and the policy runs the bf16 fallback — which is the "follow" policy to the bit and within
    the bf16 bar of the exact-fp32 heads of the same scaled model — for the SAME forward already (eager inference reads the flag at the
    end of the forward and re-runs the heads) and from then on.  With the unscaled model the flag stays down."""
    from uniception_amd import engine
    F_ = 3.0e4

    def scaled_model():
        model, c = build_case_model("tiny_dpt")
        with torch.no_grad():
            for k, p in model.named_parameters():
                if ".input_process." in k and k.endswith(".1.weight") and p.dim() == 4 and p.shape[-1] == 3:
                    p.mul_(F_)                       # layer_rn 3x3 convolutions: every scratch map is F times larger
                elif "regressor" in k and ".conv2.2." in k:
                    p.mul_(1.0 / F_ if k.endswith("weight") else 1.0)
        return model.to(gpu), c

    def run(model, c, head_mode):
        img1, img2 = (t.to(gpu) for t in case_images(c))
        with engine.head_precision(head_mode), torch.no_grad(), engine.precision("bf16"):
            r1, r2 = model(img1, img2, {})
        torch.cuda.synchronize()
        return dict(pts3d_1=r1["pts3d"], conf_1=r1["conf"], pts3d_2=r2["pts3d_in_other_view"], conf_2=r2["conf"])

    engine.head_range_exceeded(reset=True)
    try:
        # unscaled: in range, flag down, fp16 heads stay fp16
        model0, c0 = build_case_model("tiny_dpt")
        run(model0.to(gpu), c0, "fp16")
        assert not engine.head_range_exceeded()
        assert engine.head_precision and engine._f16_tripped is False
        # scaled, a pipeline COMPOSED from the modules (nobody owns the whole forward): saturated but finite, reported, and the policy falls back
        model, c = scaled_model()
        exact = run(model, c, "fp32_exact")
        assert all(torch.isfinite(v).all() for v in exact.values())
        sat = run(model, c, "fp16")
        assert all(torch.isfinite(v).all() for v in sat.values()), "fp16 heads must saturate, not overflow"
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            assert engine.head_range_exceeded()
        after = run(model, c, "fp16")               # the guard has tripped: bf16 heads now
        follow = run(model, c, "follow")
        for k in HEAD_OUTPUTS:
            assert torch.equal(after[k], follow[k]), k
            assert rel_l2(after[k].cpu(), exact[k].cpu()) < 4e-2, (k, rel_l2(after[k].cpu(), exact[k].cpu()))
        err_sat = max(rel_l2(sat[k].cpu(), exact[k].cpu()) for k in HEAD_OUTPUTS)
        print(f"\n[range guard] maps x{F_:.0e}: saturated fp16 heads {err_sat:.2e} from exact fp32 heads (finite, flagged); fallback (bf16) "
              + ", ".join(f"{k}={rel_l2(after[k].cpu(), exact[k].cpu()):.1e}" for k in HEAD_OUTPUTS))
    finally:
        engine.head_range_exceeded(reset=True)


def test_fp16_heads_range_guard_protects_the_forward_that_trips_it(gpu):
    """DUSt3R.forward (the factory model owns the whole forward) reads the saturation flag at the end of an eager inference
    forward; when a head map left the fp16 range in THAT call, the policy falls back and the two heads are re-run in the transformer's
    bf16 before anything is returned: the caller gets the "follow" policy's maps — to the bit — from the call that tripped the guard,
    not the saturated ones.  With UNICEPTION_AMD_HEAD_RANGE_SYNC off (and inside hipGraph captures) the asynchronous form remains."""
    from uniception_amd import engine
    F_ = 3.0e4
    model, c = build_case_model("vitl_dpt_512")
    with torch.no_grad():
        for k, p in model.named_parameters():
            if ".input_process." in k and k.endswith(".1.weight") and p.dim() == 4 and p.shape[-1] == 3:
                p.mul_(F_)
            elif "regressor" in k and ".conv2.2." in k:
                p.mul_(1.0 / F_ if k.endswith("weight") else 1.0)
    model = model.to(gpu)
    a, b = case_images(c)
    v1 = {"img": a.to(gpu), "instance": ["a"], "data_norm_type": "dust3r"}
    v2 = {"img": b.to(gpu), "instance": ["b"], "data_norm_type": "dust3r"}

    def run(head_mode):
        with engine.head_precision(head_mode), torch.no_grad(), engine.precision("bf16"):
            r1, r2 = model(v1, v2)
        torch.cuda.synchronize()
        return dict(pts3d_1=r1["pts3d"], conf_1=r1["conf"], pts3d_2=r2["pts3d_in_other_view"], conf_2=r2["conf"])

    engine.head_range_exceeded(reset=True)
    try:
        follow = run("follow")
        assert all(torch.isfinite(v).all() for v in follow.values())
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            first = run("fp16")
        for k in HEAD_OUTPUTS:
            assert torch.equal(first[k], follow[k]), f"{k}: the forward that saturated did not return the fallback's maps"
        assert engine._f16_tripped
        # asynchronous form: the saturated (finite) maps come back, the flag is seen afterwards
        engine.head_range_exceeded(reset=True)
        prev, engine.HEAD_RANGE_SYNC = engine.HEAD_RANGE_SYNC, False
        try:
            sat = run("fp16")
            assert all(torch.isfinite(v).all() for v in sat.values())
            assert any(not torch.equal(sat[k], follow[k]) for k in HEAD_OUTPUTS)
            with pytest.warns(RuntimeWarning, match="fp16 range"):
                assert engine.head_range_exceeded()
        finally:
            engine.HEAD_RANGE_SYNC = prev
    finally:
        engine.head_range_exceeded(reset=True)
