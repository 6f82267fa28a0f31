"""Kernel-level parity: every C-ABI entry point against a plain fp32 PyTorch-CPU statement of the same op.

fp32 kernels: 1e-5-class agreement.  bf16 MFMA kernels: inputs are rounded to bf16 first, the CPU
reference runs in fp32 on the SAME rounded inputs, so the tolerance only has to absorb fp32
accumulation order + the bf16 rounding of the output.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rope_ref(tokens_bnhd, pos, base, fwd):
    """curope arithmetic (curope.cpp:21-46) in fp32 on [B,N,H,D]."""
    t = tokens_bnhd.float().clone()
    B, N, H, D = t.shape
    Q = D // 4
    inv = fwd / (base ** (torch.arange(Q, dtype=torch.float32) / Q))
    for axis in range(2):
        ang = pos[:, :, axis].float()[:, :, None] * inv[None, None, :]  # B,N,Q
        c, s = ang.cos()[:, :, None, :], ang.sin()[:, :, None, :]
        u = t[..., axis * 2 * Q: axis * 2 * Q + Q].clone()
        v = t[..., axis * 2 * Q + Q: axis * 2 * Q + 2 * Q].clone()
        t[..., axis * 2 * Q: axis * 2 * Q + Q] = u * c - v * s
        t[..., axis * 2 * Q + Q: axis * 2 * Q + 2 * Q] = v * c + u * s
    return t


def grid_pos(B, h, w):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([ys.reshape(-1), xs.reshape(-1)], -1)[None].expand(B, -1, -1).contiguous()


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
@pytest.mark.parametrize("shape", [(2, 6, 9, 3, 64), (1, 2, 3, 2, 32), (1, 3, 5, 1, 20)])
def test_rope2d_contiguous(gpu, dtype, tol, shape):
    from uniception_amd import ops
    B, h, w, H, D = shape
    g = torch.Generator().manual_seed(1)
    tok = torch.randn(B, h * w, H, D, generator=g).to(dtype)
    pos = grid_pos(B, h, w)
    ref = rope_ref(tok, pos, 100.0, 1.0)
    t = tok.to(gpu)
    ops.rope_2d_(t, pos.to(gpu), 100.0, 1.0)
    assert rel_l2(t.cpu().float(), ref) < tol
    # inverse rotation restores the input (the backward path of curope2d.py:24-28)
    ops.rope_2d_(t, pos.to(gpu), 100.0, -1.0)
    assert rel_l2(t.cpu().float(), tok.float()) < 2 * tol


def test_rope2d_against_the_references_compiled_cpu_loop(gpu):
    """uc_rope2d (the curope drop-in) against the REFERENCE'S OWN code: curope.cpp compiled from /root/reference in the build container
    into oracle/_ref/curope_ref.so (oracle/build_ref.py; it travels to the GPU box with the snapshot) — `rope_2d` on CPU tensors is
    its `rope_2d_cpu` loop (curope.cpp:11-46).  Forward and inverse rotation, random (non-grid) positions, 64- and 32-wide heads."""
    from oracle import build_ref
    from uniception_amd import ops
    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref/curope_ref.so is not in this snapshot")
    g = torch.Generator().manual_seed(4)
    for (B, N, H, D) in ((2, 77, 3, 64), (1, 200, 2, 32)):
        t = torch.randn(B, N, H, D, generator=g)
        pos = torch.randint(0, 64, (B, N, 2), generator=g)
        for fwd in (1.0, -1.0):
            want = t.clone()
            ref.rope_2d(want, pos, 100.0, fwd)
            got = t.to(gpu)
            ops.rope_2d_(got, pos.to(gpu), 100.0, fwd)
            assert float((got.cpu() - want).abs().max()) < 2e-5, (B, N, H, D, fwd)


def test_rope2d_strided_qkv_view(gpu):
    """q and k views of a fused qkv buffer rotated in place; v untouched (blocks.py:105-114)."""
    from uniception_amd import ops
    B, h, w, H, D = 2, 4, 7, 3, 64
    N = h * w
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B, N, 3, H, D, generator=g)
    pos = grid_pos(B, h, w)
    dev = qkv.to(gpu)
    q, k = dev[:, :, 0], dev[:, :, 1]
    ops.rope_2d_(q, pos.to(gpu), 100.0, 1.0)
    ops.rope_2d_(k, pos.to(gpu), 100.0, 1.0)
    out = dev.cpu()
    assert rel_l2(out[:, :, 0], rope_ref(qkv[:, :, 0], pos, 100.0, 1.0)) < 2e-6
    assert rel_l2(out[:, :, 1], rope_ref(qkv[:, :, 1], pos, 100.0, 1.0)) < 2e-6
    assert torch.equal(out[:, :, 2], qkv[:, :, 2])


def test_rope2d_error_behaviour(gpu):
    from uniception_amd import ops
    t = torch.zeros(1, 4, 2, 64, device=gpu)
    with pytest.raises(RuntimeError):
        ops.rope_2d_(t[0], torch.zeros(1, 4, 2, dtype=torch.int64, device=gpu), 100.0, 1.0)
    with pytest.raises(RuntimeError):
        ops.rope_2d_(t, torch.zeros(1, 5, 2, dtype=torch.int64, device=gpu), 100.0, 1.0)
    with pytest.raises(RuntimeError):
        ops.rope_2d_(t, torch.zeros(1, 4, 3, dtype=torch.int64, device=gpu), 100.0, 1.0)
    with pytest.raises(RuntimeError):  # CPU tensor: no fallback
        ops.rope_2d_(torch.zeros(1, 4, 2, 64), torch.zeros(1, 4, 2, dtype=torch.int64), 100.0, 1.0)


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [64, 768, 1024, 100, 2052])
@pytest.mark.parametrize("rows", [1, 7, 513])
def test_layernorm(gpu, C, rows):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, C, generator=g) * 3 + 0.5
    w = torch.randn(C, generator=g)
    b = torch.randn(C, generator=g)
    ref = F.layer_norm(x, (C,), w, b, 1e-6)
    y = ops.layernorm(x.to(gpu), w.to(gpu), b.to(gpu), 1e-6, torch.float32)
    assert rel_l2(y.cpu(), ref) < 3e-6
    yb = ops.layernorm(x.to(gpu), w.to(gpu), b.to(gpu), 1e-6, torch.bfloat16)
    assert rel_l2(yb.cpu().float(), ref) < 5e-3
    xb = x.bfloat16()
    yb2 = ops.layernorm(xb.to(gpu), w.to(gpu), b.to(gpu), 1e-6, torch.float32)
    assert rel_l2(yb2.cpu(), F.layer_norm(xb.float(), (C,), w, b, 1e-6)) < 3e-6


# ------------------------------------------------------------------------------------------
GEMM_SHAPES = [(128, 128, 64), (256, 384, 128), (130, 70, 96), (1, 4, 8), (777, 200, 864), (64, 3072, 768)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_f32(gpu, M, N, K):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(4)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = F.gelu(a @ w.t() + bias) + res
    out = ops.gemm(a.to(gpu), w.to(gpu), bias.to(gpu), act="gelu", residual=res.to(gpu))
    assert rel_l2(out.cpu(), ref) < 3e-6
    out2 = ops.gemm(a.to(gpu), w.to(gpu))
    assert rel_l2(out2.cpu(), a @ w.t()) < 3e-6


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bf16(gpu, M, N, K):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.float() @ w.float().t()
    out = ops.gemm(a.to(gpu), w.to(gpu), out_dtype=torch.float32)
    assert rel_l2(out.cpu(), ref) < 2e-5, "asymmetric random operands: catches a transposed C write"
    ref2 = F.gelu(ref + bias) + res
    out2 = ops.gemm(a.to(gpu), w.to(gpu), bias.to(gpu), act="gelu", residual=res.to(gpu), out_dtype=torch.float32)
    assert rel_l2(out2.cpu(), ref2) < 2e-5
    out3 = ops.gemm(a.to(gpu), w.to(gpu), bias.to(gpu), act="relu", out_dtype=torch.bfloat16)
    assert rel_l2(out3.cpu().float(), F.relu(ref + bias)) < 4e-3
    # bf16 residual, strided A (row stride > K), in-place accumulate into the residual buffer
    abig = torch.zeros(M, K + 8, dtype=torch.bfloat16)
    abig[:, :K] = a
    resb = res.bfloat16()
    rdev = resb.to(gpu)
    ops.gemm(abig.to(gpu)[:, :K], w.to(gpu), bias.to(gpu), residual=rdev, out=rdev)
    assert rel_l2(rdev.cpu().float(), ref + bias + resb.float()) < 6e-3


def test_gemm_bf16_rope_and_vt_epilogue(gpu):
    """Fused QKV epilogue: RoPE on q,k columns, V written in the packed VT layout."""
    from uniception_amd import ops
    B, h, w, H = 2, 5, 7, 3   # N = 35 tokens: not a multiple of 16 -> general VT path
    # (8,8): N = 64, aligned fast path; (14,14) = the 196 tokens of a 224 x 224 view and (6,6): multiples of 4 but not of 16 — the
    # quad-store path (round 6), wave tiles straddling image borders
    for (h, w) in [(5, 7), (8, 8), (14, 14), (6, 6)]:
        N = h * w
        Cd = H * 64
        g = torch.Generator().manual_seed(6)
        x = torch.randn(B * N, 128, generator=g).bfloat16()
        wq = (torch.randn(3 * Cd, 128, generator=g) / math.sqrt(128)).bfloat16()
        bias = torch.randn(3 * Cd, generator=g)
        pos = grid_pos(B, h, w)
        table = ops.rope_table(gpu, max(h, w), 100.0)
        vt = ops.vt_buffer(B, H, N, gpu)
        vt.zero_()
        qk = ops.gemm(x.to(gpu), wq.to(gpu), bias.to(gpu), out_dtype=torch.bfloat16,
                      rope=(pos.to(gpu).view(-1, 2), table, 2 * Cd), vt=(2 * Cd, vt, N))
        assert qk.shape == (B * N, 2 * Cd)
        full = (x.float() @ wq.float().t() + bias).view(B, N, 3, H, 64)
        q_ref = rope_ref(full[:, :, 0], pos, 100.0, 1.0)
        k_ref = rope_ref(full[:, :, 1], pos, 100.0, 1.0)
        got = qk.cpu().float().view(B, N, 2, H, 64)
        assert rel_l2(got[:, :, 0], q_ref) < 4e-3
        assert rel_l2(got[:, :, 1], k_ref) < 4e-3
        # VT layout: vt[b,h,d, 16*(n//16) + perm(n%16)] == v[b,n,h,d]
        v_ref = full[:, :, 2]  # B,N,H,64
        n = torch.arange(N)
        wv = n % 16
        posn = (n // 16) * 16 + ((wv >> 2) & 1) * 8 + (wv & 3) + 4 * (wv >> 3)
        vt_cpu = vt.cpu().float()
        got_v = vt_cpu[:, :, :, posn].permute(0, 3, 1, 2)  # B,N,H,64
        assert rel_l2(got_v, v_ref) < 4e-3
        # uc_vt_pack produces the same layout from a row-major V
        packed = ops.vt_pack(v_ref.bfloat16().to(gpu))
        assert torch.equal(packed.cpu()[:, :, :, posn], v_ref.bfloat16().permute(0, 2, 3, 1))


@pytest.mark.parametrize("geom", [(2, 8, 16), (8, 5, 13)], ids=["M256", "M520_ragged"])
@pytest.mark.parametrize("variant", ["auto", "0", "1", "2", "3", "4", "6", "7"])
def test_gemm_folded_layernorm(gpu, variant, geom, monkeypatch):
    """LayerNorm fused into the GEMMs around it (blocks.py:158-161, transformer_blocks.py:643-646): the producer's fp32 epilogue
    emits a bf16 twin + per-row block statistics, the consumer GEMM on the RAW twin with gamma folded into W reproduces
    LN(x) W^T + b through its epilogue — plain, GELU, RoPE and VT tiles, with a non-zero row mean."""
    from uniception_amd import ops
    if variant != "auto":
        ops.tuning_set("gemm_variant", int(variant))       # (reset to automatic after every test: conftest.py)
    g = torch.Generator().manual_seed(77)
    # (M520_ragged: a last row tile of 8 rows and 65-token images — the eight-wave kernel's LDS side panel clamps its row / column
    # pieces there; the bitwise comparisons with the block-partial form below are comparisons with the global-load epilogue)
    (B, h, w_), H = geom, 3
    N, C = h * w_, H * 64
    M = B * N
    a = torch.randn(M, 128, generator=g).bfloat16()
    wp = (torch.randn(C, 128, generator=g) / math.sqrt(128)).bfloat16()
    bp = torch.randn(C, generator=g) + 0.7                      # row mean well away from zero
    res = torch.randn(M, C, generator=g) * 2 + 0.5
    x_ref = a.float() @ wp.float().t() + bp + res
    x = ops.gemm(a.to(gpu), wp.to(gpu), bp.to(gpu), residual=res.to(gpu), out_dtype=torch.float32, emit_ln=True)
    side = x.uc_ln
    assert rel_l2(x.cpu(), x_ref) < 2e-5
    assert torch.equal(side.twin.cpu(), x.cpu().bfloat16()), "twin = bf16 rounding of the stored fp32 rows"
    st = side.stats(1e-6).cpu()
    xs = x.cpu().double()
    assert (st[:, 0].double() - xs.mean(1)).abs().max() < 1e-5
    assert ((st[:, 1].double() - 1 / torch.sqrt(xs.var(1, unbiased=False) + 1e-6)) / st[:, 1].double()).abs().max() < 1e-5
    # producer without residual (embedding GEMMs)
    x2 = ops.gemm(a.to(gpu), wp.to(gpu), bp.to(gpu), out_dtype=torch.float32, emit_ln=True)
    assert rel_l2(x2.cpu(), a.float() @ wp.float().t() + bp) < 2e-5 and torch.equal(x2.uc_ln.twin.cpu(), x2.cpu().bfloat16())
    # consumer
    gamma, beta = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    h_ref = F.layer_norm(x.cpu(), (C,), gamma, beta, 1e-6)
    for act in (None, "gelu"):
        wq = (torch.randn(3 * C, C, generator=g) / math.sqrt(C))
        bq = torch.randn(3 * C, generator=g)
        wf = (wq * gamma[None, :]).bfloat16()
        bias = (wq @ beta + bq)
        cs = wf.float().sum(1)
        y = ops.gemm(side.twin, wf.to(gpu), bias.to(gpu), act=act, ln=(side.stats(1e-6), cs.to(gpu)))
        # the block partials merged inside the consumer's epilogue (uc_gemm_desc.ln_nblk: small batches, no finalize launch): same bits
        y_m = ops.gemm(side.twin, wf.to(gpu), bias.to(gpu), act=act, ln=(ops.LnPartial(side.partial, 1e-6), cs.to(gpu)))
        assert torch.equal(y, y_m)
        y_ref = h_ref @ wq.t() + bq
        y_ref = F.gelu(y_ref) if act else y_ref
        assert rel_l2(y.cpu().float(), y_ref) < 6e-3
        # same operands through the unfused route: LayerNorm kernel (bf16 out) -> GEMM
        hb = ops.layernorm(x, gamma.to(gpu), beta.to(gpu), 1e-6, torch.bfloat16)
        y_unf = ops.gemm(hb, wq.bfloat16().to(gpu), bq.to(gpu), act=act)
        assert rel_l2(y.cpu().float(), y_ref) < 1.5 * rel_l2(y_unf.cpu().float(), y_ref) + 1e-3
    # RoPE + VT tiles behind the folded LayerNorm
    pos = grid_pos(B, h, w_)
    table = ops.rope_table(gpu, max(h, w_), 100.0)
    vt = ops.vt_buffer(B, H, N, gpu)
    qk = ops.gemm(side.twin, wf.to(gpu), bias.to(gpu), rope=(pos.to(gpu).view(-1, 2), table, 2 * C), vt=(2 * C, vt, N),
                  ln=(side.stats(1e-6), cs.to(gpu)))
    vt_m = ops.vt_buffer(B, H, N, gpu)
    qk_m = ops.gemm(side.twin, wf.to(gpu), bias.to(gpu), rope=(pos.to(gpu).view(-1, 2), table, 2 * C), vt=(2 * C, vt_m, N),
                    ln=(ops.LnPartial(side.partial, 1e-6), cs.to(gpu)))
    assert torch.equal(qk, qk_m) and torch.equal(vt, vt_m)
    full = (h_ref @ wq.t() + bq).view(B, N, 3, H, 64)
    got = qk.cpu().float().view(B, N, 2, H, 64)
    assert rel_l2(got[:, :, 0], rope_ref(full[:, :, 0], pos, 100.0, 1.0)) < 6e-3
    assert rel_l2(got[:, :, 1], rope_ref(full[:, :, 1], pos, 100.0, 1.0)) < 6e-3
    n = torch.arange(N)
    wv = n % 16
    posn = (n // 16) * 16 + ((wv >> 2) & 1) * 8 + (wv & 3) + 4 * (wv >> 3)
    assert rel_l2(vt.cpu().float()[:, :, :, posn].permute(0, 3, 1, 2), full[:, :, 2]) < 6e-3


@pytest.mark.parametrize("variant", ["auto", "0", "1", "2", "3", "4", "6", "7"])
def test_gemm_bf16_residual_stream_producer(gpu, variant, monkeypatch):
    """The producer GEMM of a bf16 residual stream (the reference's stream under autocast): bf16 output = round(acc + bias + bf16
    residual) in ONE rounding, row statistics of the STORED (rounded) rows, the output its own twin; ragged M;
    and the consumer's folded LayerNorm on it."""
    from uniception_amd import ops
    if variant != "auto":
        ops.tuning_set("gemm_variant", int(variant))       # (reset to automatic after every test: conftest.py)
    g = torch.Generator().manual_seed(78)
    for M in (520, 256):
        C, K = 192, 128
        a = torch.randn(M, K, generator=g).bfloat16()
        wp = (torch.randn(C, K, generator=g) / math.sqrt(K)).bfloat16()
        bp = torch.randn(C, generator=g) + 0.7
        res = (torch.randn(M, C, generator=g) * 2 + 0.5).bfloat16()
        want = a.float() @ wp.float().t() + bp + res.float()
        x = ops.gemm(a.to(gpu), wp.to(gpu), bp.to(gpu), residual=res.to(gpu), out_dtype=torch.bfloat16, emit_ln=True)
        assert x.dtype == torch.bfloat16 and x.uc_ln.twin is x
        # one rounding of the fp32 sum: equal to the bf16 of the reference up to the accumulation-order noise at rounding ties
        xd = x.cpu().float()
        assert rel_l2(xd, want) < 3e-3
        assert ((xd - want).abs() <= want.abs() * 2.0 ** -8 + 1e-6).all()
        st = x.uc_ln.stats(1e-6).cpu().double()
        assert (st[:, 0] - xd.double().mean(1)).abs().max() < 1e-5, "statistics are those of the stored rows"
        assert ((st[:, 1] - 1 / torch.sqrt(xd.double().var(1, unbiased=False) + 1e-6)) / st[:, 1]).abs().max() < 1e-5
        # without a residual (embedding GEMMs)
        x0 = ops.gemm(a.to(gpu), wp.to(gpu), bp.to(gpu), out_dtype=torch.bfloat16, emit_ln=True)
        assert rel_l2(x0.cpu().float(), a.float() @ wp.float().t() + bp) < 3e-3 and x0.uc_ln.twin is x0
        # bf16 residual without statistics (the last sub-layer of a stack)
        x1 = ops.gemm(a.to(gpu), wp.to(gpu), bp.to(gpu), residual=res.to(gpu), out_dtype=torch.bfloat16)
        assert torch.equal(x1.cpu(), ops.gemm(a.to(gpu), wp.to(gpu), bp.to(gpu), residual=res.to(gpu), out_dtype=torch.bfloat16, emit_ln=True).cpu())
        # consumer: folded LayerNorm on the stream
        gamma, beta = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
        wq = torch.randn(2 * C, C, generator=g) / math.sqrt(C)
        bq = torch.randn(2 * C, generator=g)
        wf = (wq * gamma[None, :]).bfloat16()
        y = ops.gemm(x, wf.to(gpu), (wq @ beta + bq).to(gpu), ln=(x.uc_ln.stats(1e-6), wf.float().sum(1).to(gpu)))
        y_ref = F.layer_norm(x.cpu().float(), (C,), gamma, beta, 1e-6) @ wq.t() + bq
        assert rel_l2(y.cpu().float(), y_ref) < 6e-3


@pytest.mark.parametrize("variant", ["0", "1", "2", "3", "4", "6", "7"])
def test_gemm_bf16_tile_variants(gpu, variant, monkeypatch):
    """Every tile variant of the direct-to-LDS kernel (uc_tuning_set "gemm_variant") against the fp32 product, through
    each specialised epilogue: bf16 store (+GELU), fp32 residual add, RoPE + VT, the generic drain (ragged N, bf16 residual)."""
    from uniception_amd import ops
    ops.tuning_set("gemm_variant", int(variant))           # (reset to automatic after every test: conftest.py)
    g = torch.Generator().manual_seed(50 + int(variant))
    for (M, N, K) in [(512, 384, 256), (300, 200, 128), (1024, 768, 64), (520, 328, 192)]:
        a = torch.randn(M, K, generator=g).bfloat16()
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, generator=g)
        res = torch.randn(M, N, generator=g)
        ref = a.float() @ w.float().t()
        ad, wd, bd = a.to(gpu), w.to(gpu), bias.to(gpu)
        assert rel_l2(ops.gemm(ad, wd, out_dtype=torch.float32).cpu(), ref) < 2e-5
        assert rel_l2(ops.gemm(ad, wd, bd, act="gelu").cpu().float(), F.gelu(ref + bias)) < 4e-3
        assert rel_l2(ops.gemm(ad, wd, bd).cpu().float(), ref + bias) < 4e-3
        assert rel_l2(ops.gemm(ad, wd, bd, residual=res.to(gpu), out_dtype=torch.float32).cpu(), ref + bias + res) < 2e-5
        rb = res.bfloat16().to(gpu)
        assert rel_l2(ops.gemm(ad, wd, bd, residual=rb, out_dtype=torch.bfloat16).cpu().float(), ref + bias + res.bfloat16().float()) < 6e-3
    B, h, w_, H = 3, 8, 16, 3          # 128 tokens per image: the whole-row VT store path
    N, Cd = h * w_, H * 64
    x = torch.randn(B * N, 128, generator=g).bfloat16()
    wq = (torch.randn(3 * Cd, 128, generator=g) / math.sqrt(128)).bfloat16()
    bias = torch.randn(3 * Cd, generator=g)
    pos = grid_pos(B, h, w_)
    table = ops.rope_table(gpu, max(h, w_), 100.0)
    vt = ops.vt_buffer(B, H, N, gpu)
    qk = ops.gemm(x.to(gpu), wq.to(gpu), bias.to(gpu), rope=(pos.to(gpu).view(-1, 2), table, 2 * Cd), vt=(2 * Cd, vt, N))
    full = (x.float() @ wq.float().t() + bias).view(B, N, 3, H, 64)
    got = qk.cpu().float().view(B, N, 2, H, 64)
    assert rel_l2(got[:, :, 0], rope_ref(full[:, :, 0], pos, 100.0, 1.0)) < 4e-3
    assert rel_l2(got[:, :, 1], rope_ref(full[:, :, 1], pos, 100.0, 1.0)) < 4e-3
    n = torch.arange(N)
    wv = n % 16
    posn = (n // 16) * 16 + ((wv >> 2) & 1) * 8 + (wv & 3) + 4 * (wv >> 3)
    assert rel_l2(vt.cpu().float()[:, :, :, posn].permute(0, 3, 1, 2), full[:, :, 2]) < 4e-3


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("geom", [(2, 16, 24, 128, "relu"), (1, 37, 29, 64, "relu"), (3, 8, 8, 32, None), (1, 40, 40, 128, "gelu")])
def test_conv3x3_fused_tail(gpu, geom, mode):
    """conv3x3(Cin -> 128) -> act -> Conv2d(128 -> 4, 1x1) in one kernel (uc_gemm_desc.tail_*; dpt.py:271-277): ragged M, every
    tile variant the dispatcher may pick, plain bf16 and split-operand (fp32-class) arithmetic."""
    from uniception_amd import engine, ops
    B, H, W, Cin, act = geom
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, H, W, Cin, generator=g)
    wc = torch.randn(128, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bc = torch.randn(128, generator=g) * 0.3
    w4 = torch.randn(4, 128, generator=g) / math.sqrt(128)
    b4 = torch.randn(4, generator=g)
    xin = x.bfloat16().float() if mode == "bf16" else x
    win = wc.bfloat16().float() if mode == "bf16" else wc
    y = F.conv2d(xin.permute(0, 3, 1, 2).double(), win.double(), bc.double(), padding=1)
    y = {"relu": F.relu, "gelu": F.gelu, None: lambda t: t}[act](y)
    ref = torch.einsum("bchw,oc->bhwo", y, w4.double()) + b4.double()
    wg = wc.permute(0, 2, 3, 1).reshape(128, -1).contiguous()
    for variant in ("auto", "0", "1", "3"):
        ops.tuning_set("gemm_variant", -3 if variant == "auto" else int(variant))
        try:
            if mode == "bf16":
                out = ops.gemm(x.bfloat16().to(gpu), wg.bfloat16().to(gpu), bc.to(gpu), act=act, conv=(B, H, W, Cin, 1), tail=(w4.to(gpu), b4.to(gpu)))
            else:
                with engine.precision("bf16x3"):
                    out = ops.gemm(x.to(gpu), wg.to(gpu), bc.to(gpu), act=act, conv=(B, H, W, Cin, 1), tail=(w4.to(gpu), b4.to(gpu)))
        finally:
            ops.tuning_set("gemm_variant", -3)
        assert out.shape == (B * H * W, 4) and out.dtype == torch.float32
        tol = 3e-4 if act == "gelu" else (3e-5 if mode == "bf16" else 1e-5)      # (the epilogue's GELU is a 3e-5-accurate polynomial)
        assert rel_l2(out.view(B, H, W, 4).cpu(), ref) < tol, (variant, mode)
    out = ops.gemm(x.bfloat16().to(gpu), wg.bfloat16().to(gpu), None, act=act, conv=(B, H, W, Cin, 1), tail=(w4.to(gpu), None))
    y0 = {"relu": F.relu, "gelu": F.gelu, None: lambda t: t}[act](F.conv2d(x.bfloat16().double().permute(0, 3, 1, 2), wc.bfloat16().double(), None, padding=1))
    assert rel_l2(out.view(B, H, W, 4).cpu(), torch.einsum("bchw,oc->bhwo", y0, w4.double())) < (3e-4 if act == "gelu" else 3e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(2, 9, 11, 16, 24, 1), (1, 37, 37, 32, 40, 2), (2, 8, 8, 96, 256, 1), (1, 6, 5, 8, 8, 2),
                                  (1, 9, 11, 64, 72, 1), (2, 10, 7, 128, 40, 2), (3, 19, 23, 64, 300, 1)])
def test_conv3x3_implicit_gemm(gpu, dtype, geom):
    """3x3 conv, pad 1, stride 1/2 (dpt_block.py:34-69, dpt.py:161-172) incl. the ReLU-on-load of the RCU."""
    from uniception_amd import ops
    B, H, W, Cin, Cout, s = geom
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype)
    wt = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    bias = torch.randn(Cout, generator=g)
    for relu_a in (False, True):
        xin = F.relu(x.float()) if relu_a else x.float()
        ref = F.conv2d(xin, wt.float(), bias, stride=s, padding=1)  # B,Cout,Ho,Wo
        x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(gpu)
        w_r = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(gpu)  # (ky,kx,c)
        out = ops.gemm(x_nhwc, w_r, bias.to(gpu), conv=(B, H, W, Cin, s), relu_a=relu_a, out_dtype=torch.float32)
        Ho, Wo = ref.shape[2:]
        got = out.cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
        assert rel_l2(got, ref) < (3e-6 if dtype == torch.float32 else 2e-5)


@pytest.mark.parametrize("M,N,K", [(2048, 1024, 4096), (1024, 768, 3072), (2048, 1024, 1024), (1000, 1024, 1152)])
def test_gemm_small_m_fused_k_split(gpu, M, N, K):
    """Small-M path: a dense launch whose 128x128 tiles cover at most half the CUs splits K in two across twice the workgroups and hands
    the first half's accumulators over inside the kernel (uc_gemm_desc unchanged; csrc/gemm_glds.h fuse_split2).  Against the unsplit
    kernel (gemm_variant 0 forced) and fp64: every epilogue family (bf16 store + GELU, bf16 residual stream with row statistics, fp32
    residual), a ragged M, two streams at once (each has its own hand-over workspace) and a hipGraph replay (the flags must be back
    at zero after every launch)."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(gpu)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(gpu)
    b = torch.randn(N, generator=g).to(gpu)
    res16 = torch.randn(M, N, generator=g).bfloat16().to(gpu)
    res32 = torch.randn(M, N, generator=g).to(gpu)
    ref = a.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    cases = {"gelu": (lambda: ops.gemm(a, w, b, act="gelu"), F.gelu(ref), 6e-3),
             "bf16 stream": (lambda: ops.gemm(a, w, b, residual=res16, emit_ln=True), ref + res16.double().cpu(), 6e-3),
             "f32 residual": (lambda: ops.gemm(a, w, b, residual=res32, out_dtype=torch.float32), ref + res32.double().cpu(), 2e-5)}
    for name, (fn, want, tol) in cases.items():
        y = fn()
        with ops.tuning("gemm_variant", 0):
            y0 = fn()
        assert rel_l2(y.float().cpu(), want) < tol, name
        assert rel_l2(y.float(), y0.float()) < (2e-6 if y.dtype == torch.float32 else 2e-3), name     # two partial sums instead of one chain
        if name == "bf16 stream":
            assert torch.allclose(y.uc_ln.stats(1e-6), y0.uc_ln.stats(1e-6), rtol=2e-2, atol=2e-2)
    if K == 4096:
        # the 3x3 conv form (K = 9 Cin = 2304, 64 tiles) and fp16 operands (the prediction heads' mode) take the same path
        for dt in (torch.bfloat16, torch.float16):
            x = torch.randn(1, 64, 64, 256, generator=g).to(dt).to(gpu)
            wc = (torch.randn(256, 9 * 256, generator=g) / 48).to(dt).to(gpu)
            r = torch.randn(4096, 256, generator=g).to(dt).to(gpu)
            fn = lambda: ops.gemm(x, wc, b[:256].contiguous(), conv=(1, 64, 64, 256, 1), relu_a=True, residual=r, out_dtype=torch.float32)
            y = fn()
            with ops.tuning("small_m_split", 0):
                y0 = fn()
            assert rel_l2(y, y0) < 2e-6 and not torch.equal(y, y0), dt
            a16 = a.to(dt)
            w16 = w.to(dt)
            y = ops.gemm(a16, w16, b, out_dtype=torch.float32)
            with ops.tuning("small_m_split", 0):
                y0 = ops.gemm(a16, w16, b, out_dtype=torch.float32)
            assert rel_l2(y, y0) < 2e-6 and not torch.equal(y, y0), dt
    # two streams at once, many launches back to back
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    torch.cuda.synchronize()
    for _ in range(8):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                outs.append(ops.gemm(a, w, b, residual=res32, out_dtype=torch.float32))
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # graph replay
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = torch.empty(M, N, device=gpu, dtype=torch.float32)
        ops.gemm(a, w, b, residual=res32, out=out)          # (the stream's workspace is created outside the capture)
        torch.cuda.synchronize()
        # (a capture belongs to a graph: ops.capture_scope(token) names it, fuse_ws_release(token) hands its buffers back — what
        # uniception_amd.graphs does; a capture WITHOUT a scope has no owner for a hand-over buffer and runs unsplit: correct, other bits)
        graph0 = torch.cuda.CUDAGraph()
        out0 = torch.empty(M, N, device=gpu, dtype=torch.float32)
        with torch.cuda.graph(graph0, stream=st):
            ops.gemm(a, w, b, residual=res32, out=out0)
        graph0.replay()
        torch.cuda.synchronize()
        with ops.tuning("small_m_split", 0):
            assert torch.equal(out0, ops.gemm(a, w, b, residual=res32, out_dtype=torch.float32))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st), ops.capture_scope(0x7e57_0001):
            ops.gemm(a, w, b, residual=res32, out=out)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, outs[0])
    # a second graph, recorded on the same capture stream, replayed at the same time on another stream: each graph owns its hand-over buffers
    with torch.cuda.stream(st):
        out2 = torch.empty(M, N, device=gpu, dtype=torch.float32)
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2, stream=st), ops.capture_scope(0x7e57_0002):
            for _ in range(4):
                ops.gemm(a, w, b, residual=res32, out=out2)
    torch.cuda.synchronize()
    for _ in range(5):
        out.zero_(); out2.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(4):
                graph.replay()
        with torch.cuda.stream(s2):
            graph2.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, outs[0]) and torch.equal(out2, outs[0])
    del graph, graph2
    ops.fuse_ws_release(0x7e57_0001)
    ops.fuse_ws_release(0x7e57_0002)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(4, 128, 128, 64, 128), (16, 64, 64, 128, 128), (1, 256, 256, 64, 256), (1, 128, 512, 64, 128), (2, 192, 64, 64, 256),
                                  (1, 256, 256, 256, 128)])
def test_conv3x3_row_walking_kernel(gpu, dtype, geom):
    """Stride-1 convs on maps >= 64 wide whose rows tile 256 pixels, with >= 256 tiles, take conv3x3_rows_kernel (a slab of input
    pixels per kernel row, the three horizontal taps as row offsets of the fragment reads): same contract as the implicit-GEMM
    kernel — against torch's conv on the same rounded operands; tiles of one, two and four image-row segments, two segments per
    image row (W = 512), image borders, several images, ReLU on load, bias + ReLU, two residuals, fp32 output, the fused 1x1 tail,
    bf16 and fp16 operands."""
    from uniception_amd import ops
    B, H, W, Cin, Cout = geom
    g = torch.Generator().manual_seed(B * H + W + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).to(dtype)
    wt = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    bias = torch.randn(Cout, generator=g)
    r1 = torch.randn(B * H * W, Cout, generator=g).to(dtype)
    r2 = torch.randn(B * H * W, Cout, generator=g).to(dtype)
    w_r = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(gpu)
    xg = x.to(gpu)
    tol16 = 6e-3 if dtype == torch.bfloat16 else 8e-4          # 16-bit outputs: one rounding of the result
    ops.tuning_set("conv_rows", 2)                              # every eligible shape (the default takes it where it wins only); conftest resets
    for relu_a in (False, True):
        xin = F.relu(x.double()) if relu_a else x.double()
        ref = F.conv2d(xin.permute(0, 3, 1, 2), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
        out = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), relu_a=relu_a, out_dtype=torch.float32)
        assert rel_l2(out.cpu(), ref) < 2e-5
        out = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), relu_a=relu_a, act="relu")
        assert out.dtype == dtype and rel_l2(out.float().cpu(), F.relu(ref)) < tol16
        out = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), relu_a=relu_a, residual=r1.to(gpu), residual2=r2.to(gpu))
        assert rel_l2(out.float().cpu(), ref + r1.double() + r2.double()) < tol16
    # the two kernels against each other (same products, another summation order), and the default routing on the shape it is for
    outs = {}
    for mode in (0, 1, 2):
        ops.tuning_set("conv_rows", mode)
        outs[mode] = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), out_dtype=torch.float32)
    assert rel_l2(outs[2], outs[0]) < 2e-6           # (one 64-channel chunk: the same K order, the same bits)
    # default routing: the eight-wave 512-pixel kernel wherever its shape rules hold, else the 256-pixel one where it wins
    R8 = 1 if W >= 512 else 512 // W
    rows8 = W >= 128 and (W % 512 == 0 or 512 % W == 0) and H % R8 == 0 and (B * H * W) % 512 == 0 and (B * H * W // 512) * (Cout // 128) >= 256
    if rows8:
        ops.tuning_set("conv_rows", 3)
        assert torch.equal(outs[1], ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), out_dtype=torch.float32))
    else:
        assert torch.equal(outs[1], outs[2] if (Cout == 128 and Cin >= 256) else outs[0])
    ops.tuning_set("conv_rows", 2)
    if Cout == 128 and dtype == torch.bfloat16:
        w4 = torch.randn(4, 128, generator=g) / math.sqrt(128)
        b4 = torch.randn(4, generator=g)
        ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), wt.double(), bias.double(), padding=1))
        out = ops.gemm(xg, w_r, bias.to(gpu), act="relu", conv=(B, H, W, Cin, 1), tail=(w4.to(gpu), b4.to(gpu)))
        assert rel_l2(out.view(B, H, W, 4).cpu(), torch.einsum("bchw,oc->bhwo", ref, w4.double()) + b4.double()) < 3e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(8, 128, 128, 64, 128), (1, 256, 512, 64, 128), (2, 256, 256, 128, 128), (1, 128, 1024, 64, 128), (1, 512, 128, 64, 256),
                                  (3, 192, 256, 64, 128),
                                  # round 6, the FLAT form (rows that do not tile 512 pixels; tiles of 512 consecutive pixels): the DINOv2-518
                                  # head's 148 / 296 / 592-wide maps, the 224 x 224 head's 56 / 112 / 224, the 64^2 level, a 17-wide odd map
                                  # whose 1700 pixels leave a ragged last tile and put several images into one tile
                                  (2, 148, 148, 64, 128), (1, 37, 296, 128, 128), (1, 30, 592, 64, 128), (3, 56, 56, 64, 128), (2, 112, 112, 64, 256),
                                  (1, 224, 224, 64, 128), (4, 64, 64, 64, 128), (5, 20, 17, 64, 128)])
def test_conv3x3_eight_wave_row_walking_kernel(gpu, dtype, geom):
    """conv3x3_rows8_kernel (512 pixels x 128 output channels per workgroup, eight waves of 128 x 64, 32-channel super-steps of three
    taps, register-prefetched fragments): same contract as the implicit-GEMM kernel — against torch's conv on the same rounded
    operands; tiles of one, two and four image-row segments, two tiles per image row (W = 1024), image borders, several images,
    ReLU on load, bias + ReLU, two residuals, fp32 output, two column tiles, the fused 1x1 tail, bf16 and fp16 operands.
    Round 6: any width from 16 pixels and any pixel count through the kernel's flat form (edge columns zeroed in registers, the
    vertical padding by the DMA's zero fill, the last tile by the epilogues' row bound); `conv_rows_flat` 0 keeps those maps on the
    implicit-GEMM kernel (same products, another summation order)."""
    from uniception_amd import ops
    B, H, W, Cin, Cout = geom
    g = torch.Generator().manual_seed(B * H + W + Cin + 1)
    x = torch.randn(B, H, W, Cin, generator=g).to(dtype)
    wt = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    bias = torch.randn(Cout, generator=g)
    r1 = torch.randn(B * H * W, Cout, generator=g).to(dtype)
    r2 = torch.randn(B * H * W, Cout, generator=g).to(dtype)
    w_r = wt.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(gpu)
    xg = x.to(gpu)
    tol16 = 6e-3 if dtype == torch.bfloat16 else 8e-4          # 16-bit outputs: one rounding of the result
    ops.tuning_set("conv_rows", 3)                              # conftest resets
    for relu_a in (False, True):
        xin = F.relu(x.double()) if relu_a else x.double()
        ref = F.conv2d(xin.permute(0, 3, 1, 2), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Cout)
        out = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), relu_a=relu_a, out_dtype=torch.float32)
        assert rel_l2(out.cpu(), ref) < 2e-5
        out = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), relu_a=relu_a, act="relu")
        assert out.dtype == dtype and rel_l2(out.float().cpu(), F.relu(ref)) < tol16
        out = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), relu_a=relu_a, residual=r1.to(gpu), residual2=r2.to(gpu))
        assert rel_l2(out.float().cpu(), ref + r1.double() + r2.double()) < tol16
    # against the implicit-GEMM kernel: the same products in another summation order — and NOT the same launch
    outs = {}
    for mode in (0, 3):
        ops.tuning_set("conv_rows", mode)
        outs[mode] = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), out_dtype=torch.float32)
    assert rel_l2(outs[3], outs[0]) < 2e-6 and not torch.equal(outs[3], outs[0])
    ops.tuning_set("conv_rows", 3)
    flat = not (W >= 128 and (W % 512 == 0 or 512 % W == 0) and H % max(1, 512 // W) == 0 and (B * H * W) % 512 == 0)
    if flat:      # the knob that keeps such maps off the kernel: then this IS the implicit-GEMM launch
        with ops.tuning("conv_rows_flat", 0):
            off = ops.gemm(xg, w_r, bias.to(gpu), conv=(B, H, W, Cin, 1), out_dtype=torch.float32)
        assert torch.equal(off, outs[0])
    if Cout == 128 and dtype == torch.bfloat16:
        w4 = torch.randn(4, 128, generator=g) / math.sqrt(128)
        b4 = torch.randn(4, generator=g)
        ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), wt.double(), bias.double(), padding=1))
        out = ops.gemm(xg, w_r, bias.to(gpu), act="relu", conv=(B, H, W, Cin, 1), tail=(w4.to(gpu), b4.to(gpu)))
        assert rel_l2(out.view(B, H, W, 4).cpu(), torch.einsum("bchw,oc->bhwo", ref, w4.double()) + b4.double()) < 3e-5


# ------------------------------------------------------------------------------------------
def sdpa_ref(q, k, v, scale):
    """q [B,Nq,H,D] etc. -> [B,Nq,H,D], fp32 softmax(QK^T*scale)V."""
    qh, kh, vh = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    att = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (att @ vh).permute(0, 2, 1, 3)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 3, 50, 50, 64), (1, 2, 200, 77, 32), (1, 1, 1, 1, 64), (2, 2, 129, 300, 64)])
def test_attention_f32(gpu, B, H, Nq, Nk, D):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(8)
    q = torch.randn(B, Nq, H, D, generator=g)
    k = torch.randn(B, Nk, H, D, generator=g)
    v = torch.randn(B, Nk, H, D, generator=g)
    ref = sdpa_ref(q, k, v, D ** -0.5)
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), D ** -0.5)
    assert rel_l2(out.cpu(), ref) < 3e-6


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 196, 196), (1, 2, 1024, 1024), (1, 2, 1369, 1369), (2, 1, 37, 300), (1, 1, 5, 3), (1, 2, 130, 64)])
def test_attention_bf16(gpu, B, H, Nq, Nk):
    from uniception_amd import ops
    D = 64
    g = torch.Generator().manual_seed(9)
    q = (torch.randn(B, Nq, H, D, generator=g) * 1.5).bfloat16()
    k = (torch.randn(B, Nk, H, D, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    ref = sdpa_ref(q, k, v, D ** -0.5)
    vt = ops.vt_pack(v.to(gpu))
    out = ops.attention(q.to(gpu), k.to(gpu), vt, D ** -0.5, v_packed=True)
    # P is rounded to bf16 before PV and O to bf16 on store: ~2^-8 relative
    assert rel_l2(out.cpu().float(), ref) < 8e-3
    # strided views of a fused [B,N,2,H,D] q|k buffer (what the QKV GEMM produces)
    if Nq == Nk:
        qk = torch.stack([q, k], dim=2).to(gpu)
        out2 = ops.attention(qk[:, :, 0], qk[:, :, 1], vt, D ** -0.5, v_packed=True)
        assert torch.equal(out2, out)


def test_attention_bf16_spiked_keys(gpu):
    """One key dominating a late tile forces the running-max rescale branch (online softmax)."""
    from uniception_amd import ops
    B, H, N, D = 1, 1, 256, 64
    g = torch.Generator().manual_seed(10)
    q = torch.randn(B, N, H, D, generator=g).bfloat16()
    k = torch.randn(B, N, H, D, generator=g).bfloat16()
    v = torch.randn(B, N, H, D, generator=g).bfloat16()
    k[0, 200, 0] = q[0, 17, 0] * 6  # spike: row 17 max jumps in tile 3
    k[0, 70, 0] = q[0, 99, 0] * 4
    ref = sdpa_ref(q, k, v, D ** -0.5)
    out = ops.attention(q.to(gpu), k.to(gpu), ops.vt_pack(v.to(gpu)), D ** -0.5, v_packed=True)
    assert (out.cpu().float() - ref).abs().max() < 3e-2
    assert rel_l2(out.cpu().float(), ref) < 8e-3


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 1024, 1024), (1, 2, 1369, 1369), (2, 1, 300, 1370), (1, 2, 256, 100), (3, 2, 512, 700), (1, 1, 260, 4096),
                                       (1, 2, 130, 65), (9, 16, 196, 196)])
def test_attention_persistent_kernel(gpu, B, H, Nq, Nk):
    """attn_bf16_p64_kernel (tuning knob attn_p64 = 2: wherever the shape allows): 64 queries per wave, persistent workgroups, every
    exponential against the row maximum of the item's first 32 keys.  Whole and ragged key tiles, query counts that are not a
    multiple of 64 or 256, more and fewer items than workgroups, strided q | k views, the log-sum-exp output."""
    from uniception_amd import ops
    D = 64
    g = torch.Generator().manual_seed(2000 + Nq + Nk)
    q = (torch.randn(B, Nq, H, D, generator=g) * 1.5).bfloat16()
    k = (torch.randn(B, Nk, H, D, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    ref = sdpa_ref(q, k, v, D ** -0.5)
    ref_lse = torch.logsumexp(torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * D ** -0.5, dim=-1)
    vt = ops.vt_pack(v.to(gpu))
    with ops.tuning("attn_p64", 2):
        lse = torch.full((B, H, Nq), float("nan"), device=gpu)
        out = ops.attention(q.to(gpu), k.to(gpu), vt, D ** -0.5, v_packed=True, lse=lse)
        assert torch.isfinite(out).all()
        assert rel_l2(out.cpu().float(), ref) < 8e-3
        assert (lse.cpu() - ref_lse).abs().max() < 2e-2
        if Nq == Nk:
            qk = torch.stack([q, k], dim=2).to(gpu)
            out2 = ops.attention(qk[:, :, 0], qk[:, :, 1], vt, D ** -0.5, v_packed=True)
            assert torch.equal(out2, out)
    with ops.tuning("attn_p64", 0):
        old = ops.attention(q.to(gpu), k.to(gpu), vt, D ** -0.5, v_packed=True)
    assert (old.float() - out.float()).abs().max() < 6.5e-2      # (two bf16 output steps at |o| < 8)


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 196, 196), (2, 2, 65, 65), (1, 2, 130, 1370), (2, 12, 300, 130)])
def test_attention_persistent_kernel_ragged_keys_next_to_poisoned_memory(gpu, B, H, Nq, Nk):
    "The persistent kernel's descriptors: a key row past Nk must read as zeros (and be masked), never as the NaNs next to it."
    from uniception_amd import ops
    g = torch.Generator().manual_seed(19 * Nq + Nk)
    buf = torch.full((B, Nk, 3, H, 64), float("nan")).bfloat16()
    buf[:, :, 1] = torch.randn(B, Nk, H, 64, generator=g).bfloat16()
    buf = buf.to(gpu)
    k = buf[:, :, 1]
    qbuf = torch.full((B, Nq, 2, H, 64), float("nan")).bfloat16()
    qbuf[:, :, 0] = torch.randn(B, Nq, H, 64, generator=g).bfloat16()
    q = qbuf.to(gpu)[:, :, 0]
    v = torch.randn(B, Nk, H, 64, generator=g).bfloat16().to(gpu)
    with ops.tuning("attn_p64", 2):
        o = ops.attention(q, k, ops.vt_pack(v), 0.125, v_packed=True)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 0.125
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.float())
    assert torch.isfinite(o).all()
    assert rel_l2(o.float().cpu(), ref.cpu()) < 4e-3


@pytest.mark.parametrize("case", ["stale_maximum", "overflow_late", "underflow_partner_row", "overflow_last_ragged_tile"])
def test_attention_persistent_kernel_score_range(gpu, case):
    """The stale maximum and its way out.  A key ~69 scaled nats above the first 32 keys' maximum is still handled in the kernel
    (P up to 2^100); beyond that — or when the lane's OTHER row (32 queries on) owns a first-tile maximum that pushes this row's
    exponentials under 2^-100 — the wave flags its 64-query block and attn_bf16_fixup_kernel recomputes it with the exact online
    softmax: the outputs must match the reference either way, with no sentinel (NaN) left behind."""
    from uniception_amd import ops
    B, H, D = 2, 2, 64
    Nq, Nk = (300, 1000) if case == "overflow_last_ragged_tile" else (512, 768)
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, Nq, H, D, generator=g).bfloat16()
    k = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    v = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    if case == "stale_maximum":
        k[0, 700, 0] = q[0, 17, 0] * 6          # ~ +69 in the exp2 domain against a first-tile maximum of a few: inside the range
        k[1, 300, 1] = q[1, 400, 1] * 5
    elif case == "overflow_late":
        k[0, 700, 0] = q[0, 17, 0] * 16         # ~ +185: the row sum overflows 2^100
        k[1, 40, 1] = q[1, 500, 1] * 14         # second half of the first tile: after the maximum was taken
    elif case == "underflow_partner_row":
        k[0, 5, 0] = q[0, 17, 0] * 16           # row 17's first-tile maximum ~185 is row 49's stale maximum too: row 49 underflows
        k[1, 20, 1] = q[1, 300, 1] * 15
    else:
        k[0, 990, 0] = q[0, 280, 0] * 16        # in the ragged last tile, for a query of the ragged last block
        k[1, 970, 1] = q[1, 3, 1] * 16
    ref = sdpa_ref(q, k, v, D ** -0.5)
    ref_lse = torch.logsumexp(torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * D ** -0.5, dim=-1)
    with ops.tuning("attn_p64", 2):
        lse = torch.empty(B, H, Nq, device=gpu)
        out = ops.attention(q.to(gpu), k.to(gpu), ops.vt_pack(v.to(gpu)), D ** -0.5, v_packed=True, lse=lse)
    assert torch.isfinite(out).all()
    assert (out.cpu().float() - ref).abs().max() < 6.5e-2       # (two bf16 output steps at |o| < 8)
    assert rel_l2(out.cpu().float(), ref) < 8e-3
    # (Q is pre-multiplied by scale * log2(e) and re-rounded to bf16 inside the kernel: against a key 16 x a query, |q| |k| scale ~ 130 nats,
    #  every score carries ~5e-4 of that.  Training's LSE calls DO take this kernel under the default policy (uc_attention_fwd has no lse
    #  gate): the backward kernels re-round scale * log2(e) * Q the same way, so the P they rebuild from this LSE sums to one — the
    #  gradient is that of the re-rounded Q; INTEGRATION.md §3 states the accepted LSE noise)
    assert (lse.cpu() - ref_lse).abs().max() < 0.12


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,H,W", [(16, 32, 48), (14, 28, 42), (4, 8, 8)])
def test_patch_gather(gpu, P, H, W):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(11)
    img = torch.randn(2, 3, H, W, generator=g)   # P=14 takes the scalar path (patch size not a multiple of 4)
    ref = F.unfold(img, kernel_size=P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)  # (c,u,v) columns
    out = ops.patch_gather(img.to(gpu), P, torch.float32)
    assert torch.equal(out.cpu(), ref)
    outb = ops.patch_gather(img.to(gpu), P, torch.bfloat16)
    assert torch.equal(outb.cpu(), ref.bfloat16())


def test_convert_to_fp16_saturates_flags_and_keeps_nan(gpu):
    """the fp16 saturation must not hide NaN producers — a NaN stays NaN in the fp16 output (v_med3 alone would turn it into
    65504) and trips the range flag like an overflow does (the detector's maximum is the NaN-propagating one)."""
    from uniception_amd import ops
    flag = ops.f16_sat_flag()
    for bad, expect_flag in ((None, False), (1.0e6, True), (float("nan"), True), (float("-inf"), True)):
        flag.zero_()
        x = torch.linspace(-4, 4, 4096 + 7, device=gpu)
        if bad is not None:
            x[1234] = bad
        y = ops.convert(x.contiguous(), torch.float16)
        torch.cuda.synchronize()
        assert bool(flag.item()) == expect_flag, (bad, flag.item())
        if bad is None:
            assert torch.equal(y, x.half())
        elif bad != bad:
            assert torch.isnan(y[1234]) and torch.isfinite(torch.cat([y[:1234], y[1235:]])).all()
        else:
            assert y[1234].item() == (65504.0 if bad > 0 else -65504.0)
    flag.zero_()


def test_layout_conversions(gpu):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 70, 5, 9, generator=g)
    y = ops.nchw_to_nhwc(x.to(gpu), torch.float32)
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    yb = ops.nchw_to_nhwc(x.to(gpu), torch.bfloat16)
    assert torch.equal(yb.cpu(), x.permute(0, 2, 3, 1).contiguous().bfloat16())
    z = ops.nhwc_to_nchw(yb, torch.float32)
    assert torch.equal(z.cpu(), x.bfloat16().float())
    assert torch.equal(ops.convert(x.to(gpu), torch.bfloat16).cpu(), x.bfloat16())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(2, 5, 7, 16, 10, 14, None), (1, 19, 19, 8, 38, 38, (37, 37)), (1, 37, 37, 8, 65, 65, None), (1, 1, 1, 8, 4, 4, None)])
def test_bilinear(gpu, dtype, geom):
    from uniception_amd import ops
    B, Hi, Wi, C, Ho, Wo, crop = geom
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, C, Hi, Wi, generator=g).to(dtype)
    ref = F.interpolate(x.float(), size=(Ho, Wo), mode="bilinear", align_corners=True)
    if crop:
        ref = ref[:, :, :crop[0], :crop[1]]
    out = ops.bilinear_nhwc(x.permute(0, 2, 3, 1).contiguous().to(gpu), Ho, Wo, crop)
    got = out.cpu().float().permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < (2e-6 if dtype == torch.float32 else 4e-3)


def test_bilinear_scale2_matches_scale_factor(gpu):
    """scale_factor=2, align_corners=True (dpt_block.py:251-253) == size=(2H,2W)."""
    from uniception_amd import ops
    x = torch.randn(1, 8, 6, 9)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    out = ops.bilinear_nhwc(x.permute(0, 2, 3, 1).contiguous().to(gpu), 12, 18)
    assert rel_l2(out.cpu().permute(0, 3, 1, 2), ref) < 2e-6


@pytest.mark.parametrize("k", [2, 4])
def test_convtranspose_as_gemm_plus_scatter(gpu, k):
    """ConvTranspose2d(k=s) == GEMM with weight [(u,v,o), c] + pixel scatter (dpt.py:116-140)."""
    from uniception_amd import ops
    B, Cin, Cout, h, w = 2, 24, 16, 5, 6
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, Cin, h, w, generator=g)
    wt = torch.randn(Cin, Cout, k, k, generator=g) / math.sqrt(Cin)
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x, wt, bias, stride=k)
    a = x.permute(0, 2, 3, 1).reshape(B * h * w, Cin).contiguous().to(gpu)
    wg = wt.permute(2, 3, 1, 0).reshape(k * k * Cout, Cin).contiguous().to(gpu)  # rows (u,v,o)
    bg = bias.repeat(k * k).to(gpu)
    y = ops.gemm(a, wg, bg)
    out = ops.convt_scatter(y, B, h, w, k, Cout)
    assert rel_l2(out.cpu().permute(0, 3, 1, 2), ref) < 3e-6


def test_pixel_shuffle(gpu):
    from uniception_amd import ops
    B, h, w, P, Cout = 2, 3, 5, 4, 4
    y = torch.randn(B, Cout * P * P, h, w)
    ref = F.pixel_shuffle(y, P)
    src = y.permute(0, 2, 3, 1).reshape(B * h * w, Cout * P * P).contiguous()
    out = ops.pixel_shuffle(src.to(gpu), B, h, w, P, Cout)
    assert torch.equal(out.cpu(), ref)


def test_pointmap_adaptor(gpu):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(15)
    x = torch.randn(2, 4, 6, 7, generator=g) * 2
    x[0, :3, 0, 0] = 0  # zero vector: d clipped at 1e-8
    xyz, c = x[:, :3], x[:, 3:]
    d = xyz.norm(dim=1, keepdim=True)
    pts_ref = (xyz / d.clip(min=1e-8) * torch.expm1(d)).permute(0, 2, 3, 1)
    conf_ref = (1 + c.exp().clip(max=float("inf"))).permute(0, 2, 3, 1)
    pts, conf = ops.pointmap_adaptor(x.to(gpu), 1.0, float("inf"))
    assert rel_l2(pts.cpu(), pts_ref) < 2e-6 and rel_l2(conf.cpu(), conf_ref) < 2e-6
    # channels-last input (what the DPT tail produces) gives the same result
    xcl = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).to(gpu)
    pts2, conf2 = ops.pointmap_adaptor(xcl, 1.0, float("inf"))
    assert torch.equal(pts2, pts) and torch.equal(conf2, conf)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cin", [128, 16, 256, 8, 72])   # 8 * 2^k: lane-cooperative kernel; 72: one-thread-per-pixel fallback
def test_conv1x1_to4(gpu, dtype, Cin):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(16 + Cin)
    f = torch.randn(2, 37, 29, Cin, generator=g).to(dtype)
    w = torch.randn(4, Cin, generator=g) / 11
    b = torch.randn(4, generator=g)
    ref = f.float() @ w.t() + b
    out = ops.conv1x1_to4(f.to(gpu), w.to(gpu), b.to(gpu))
    assert rel_l2(out.cpu(), ref) < 3e-6


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 4, 4), (2, 3, 9, 18), (1, 2, 196, 196), (2, 2, 65, 65), (2, 12, 4, 8), (1, 2, 130, 1370)])
def test_attention_ragged_keys_next_to_poisoned_memory(gpu, B, H, Nq, Nk):
    """Ragged key counts on the LDS-DMA kernel: K is a strided view inside a buffer whose other slots are NaN, so a key row
    read past Nk (instead of the descriptor's zero fill) or an unmasked pad position would surface as NaN / a wrong softmax."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(17 * Nq + Nk)
    buf = torch.full((B, Nk, 3, H, 64), float("nan")).bfloat16()
    buf[:, :, 1] = torch.randn(B, Nk, H, 64, generator=g).bfloat16()
    buf = buf.to(gpu)
    k = buf[:, :, 1]
    q = torch.randn(B, Nq, H, 64, generator=g).bfloat16().to(gpu)
    v = torch.randn(B, Nk, H, 64, generator=g).bfloat16().to(gpu)
    o = ops.attention(q, k, ops.vt_pack(v), 0.125, v_packed=True)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 0.125
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.float())
    assert torch.isfinite(o).all()
    assert rel_l2(o.float().cpu(), ref.cpu()) < 4e-3


def test_curope2d_module_and_autograd_function(gpu):
    """The drop-in for the reference's only native boundary (curope2d.py:12-38): cuRoPE2D on a strided [B,H,N,D] view of a fused
    qkv buffer rotates in place and returns the same tensor; cuRoPE2D_func's backward applies the inverse rotation to the
    incoming gradient — also when that gradient is not contiguous — and matches autograd through the oracle's RoPE."""
    from oracle import dust3r_oracle as O
    from uniception_amd.models.libs.croco.curope import cuRoPE2D, cuRoPE2D_func
    B, h, w, H, D = 2, 5, 7, 3, 64
    N = h * w
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, N, 3, H, D, generator=g)
    pos = grid_pos(B, h, w)
    rope = cuRoPE2D(100.0)
    dev = qkv.to(gpu)
    q_view = dev[:, :, 0].transpose(1, 2)                      # [B,H,N,D], strides of the fused buffer
    assert not q_view.is_contiguous()
    out = rope(q_view, pos.to(gpu))
    assert out.data_ptr() == q_view.data_ptr()
    ref = O.rope2d(qkv[:, :, 0].transpose(1, 2), pos, 100.0)   # oracle: [B,H,N,D]
    assert rel_l2(out.cpu(), ref) < 2e-6
    assert torch.equal(dev[:, :, 1].cpu(), qkv[:, :, 1]), "k and v of the fused buffer are untouched"
    # autograd: forward on [B,N,H,D] (the Function's layout), loss through a NON-contiguous consumer of the result
    x = torch.randn(B, N, H, D, generator=g)
    wgt = torch.randn(B, H, N, D, generator=g)
    xr = x.clone().requires_grad_(True)
    (O.rope2d(xr.transpose(1, 2), pos, 100.0) * wgt).sum().backward()
    xd = x.to(gpu).requires_grad_(True)
    y = cuRoPE2D_func.apply(xd.clone(), pos.to(gpu), 100.0, 1.0)
    assert rel_l2(y.detach().transpose(1, 2).cpu(), O.rope2d(x.transpose(1, 2), pos, 100.0)) < 2e-6
    (y.transpose(1, 2) * wgt.to(gpu)).sum().backward()         # gradient reaches the Function as a transposed view
    assert rel_l2(xd.grad.cpu(), xr.grad) < 2e-6


@pytest.mark.parametrize("K", [64, 128, 192, 256, 1024])
def test_gemm_four_wave_kernel_is_bitwise_the_sixteen_wave_kernel(gpu, K):
    """gemm_bf16_glds4_kernel (variant 7: 128x128 wave tiles, accumulators pinned in AGPRs, hand-scheduled inline-asm K-loop) adds up
    the same MFMA products in the same order as the 16-wave kernel: every epilogue family must agree with it to the bit, for 1, 2, 3
    (the three loop forms: last / no-DMA / full) and many K-steps, several tiles, a ragged last row panel and a partial column tile."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(900 + K)
    for (M, N) in [(512, 512), (776, 640), (256, 264)]:
        a = torch.randn(M, K, generator=g).bfloat16().to(gpu)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(gpu)
        bias = torch.randn(N, generator=g).to(gpu)
        res16 = torch.randn(M, N, generator=g).bfloat16().to(gpu)
        res32 = torch.randn(M, N, generator=g).to(gpu)
        outs = {}
        for variant in (2, 7):
            with ops.tuning("gemm_variant", variant):
                outs[variant] = [ops.gemm(a, w, bias), ops.gemm(a, w, bias, act="gelu"), ops.gemm(a, w, out_dtype=torch.float32),
                                 ops.gemm(a, w, bias, residual=res32, out_dtype=torch.float32),
                                 ops.gemm(a, w, bias, residual=res16, out_dtype=torch.bfloat16, emit_ln=(N % 64 == 0))]
        ref = a.float() @ w.float().t() + bias
        assert rel_l2(outs[7][0].float().cpu(), ref.cpu()) < 4e-3
        for i, (x, y) in enumerate(zip(outs[2], outs[7])):
            assert torch.equal(x, y), (M, N, K, i, float((x.float() - y.float()).abs().max()))
        if N % 64 == 0:
            s2, s7 = outs[2][4].uc_ln, outs[7][4].uc_ln
            assert torch.equal(s2.stats(1e-6), s7.stats(1e-6))


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 1024, 1024), (1, 2, 1369, 1369), (2, 1, 300, 37), (1, 2, 256, 64), (3, 2, 512, 700), (1, 1, 260, 4096)])
def test_attention_role_split_kernel_is_bitwise_the_dma_kernel(gpu, B, H, Nq, Nk):
    """attn_bf16_rs_kernel (tuning knob attn_role_split): the eight-wave forward cut into matrix / vector segments that the two halves
    of a workgroup run one segment apart.  Same MFMAs on the same operands in the same order, same softmax arithmetic: the outputs
    and the log-sum-exp must equal the one-barrier-per-tile kernel's to the bit — one and many key tiles, ragged key / query counts,
    and a spiked key that takes the rescale branch in a late tile."""
    from uniception_amd import ops
    D = 64
    g = torch.Generator().manual_seed(1000 + Nq + Nk)
    q = (torch.randn(B, Nq, H, D, generator=g) * 1.5).bfloat16()
    k = (torch.randn(B, Nk, H, D, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    if Nk > 200:
        k[0, Nk - 7, 0] = q[0, 17, 0] * 6
    vt = ops.vt_pack(v.to(gpu))
    outs = {}
    for rs in (0, 1):
        with ops.tuning("attn_role_split", rs):
            lse = torch.empty(B, H, Nq, device=gpu)
            outs[rs] = (ops.attention(q.to(gpu), k.to(gpu), vt, D ** -0.5, v_packed=True, lse=lse), lse)
    assert rel_l2(outs[1][0].cpu().float(), sdpa_ref(q, k, v, D ** -0.5)) < 8e-3
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
