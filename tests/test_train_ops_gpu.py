"""Backward-kernel parity: every training-path C-ABI entry point against torch autograd on the CPU (fp32) of the forward
expression the header says it differentiates.  The reference has no hand-written backward — autograd over its modules IS
the reference behaviour — so an autograd gradient of the same expression on the same (bf16-rounded where the kernel
takes bf16) inputs is the oracle here.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [256, 768, 1024])
@pytest.mark.parametrize("dy_dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-5)])
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_bwd(gpu, C, dy_dtype, tol, with_res):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(C)
    rows = 777
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g).to(dy_dtype)
    dres = torch.randn(rows, C, generator=g) if with_res else None
    y = F.layer_norm(x, (C,), gamma, beta, 1e-6)
    y.backward(dy.float())
    dx_ref = x.grad + (dres if with_res else 0)
    dgam = torch.zeros(C, device=gpu)
    dbet = torch.zeros(C, device=gpu)
    dx = ops.layernorm_bwd(x.detach().to(gpu), gamma.detach().to(gpu), dy.to(gpu), 1e-6, dgam, dbet,
                           dres.to(gpu) if with_res else None)
    assert rel_l2(dx.cpu(), dx_ref) < tol
    assert rel_l2(dgam.cpu(), gamma.grad) < tol
    assert rel_l2(dbet.cpu(), beta.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,ld", [(1000, 768, 768), (5, 1024, 3072), (2049, 30, 32), (1, 4, 4)])
def test_colsum(gpu, dtype, M, N, ld):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(M + N)
    full = torch.randn(M, ld, generator=g).to(dtype)
    src = full[:, :N]
    out = torch.full((N,), 0.5, device=gpu)
    ops.colsum_(full.to(gpu)[:, :N], out)
    ref = src.double().sum(0) + 0.5
    assert float((out.cpu().double() - ref).abs().max()) < 1e-4 * max(1.0, math.sqrt(M))


@pytest.mark.parametrize("act", ["gelu", "relu"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 5e-3)])
def test_act_bwd(gpu, act, dtype, tol):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(3)
    u = (torch.randn(513, 300, generator=g) * 2).to(dtype)
    dg = torch.randn(513, 300, generator=g).to(dtype)
    uu = u.float().requires_grad_(True)
    (F.gelu(uu) if act == "gelu" else F.relu(uu)).backward(dg.float())
    du = ops.act_bwd(dg.to(gpu), u.to(gpu), act)
    assert du.dtype == dtype
    assert rel_l2(du.cpu().float(), uu.grad) < tol


@pytest.mark.parametrize("R,S", [(64, 64), (197, 1000), (3, 5), (1024, 3072)])
@pytest.mark.parametrize("src,dst", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                     (torch.float32, torch.bfloat16)])
def test_transpose2d(gpu, R, S, src, dst):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(R)
    x = torch.randn(R, S, generator=g).to(src)
    y = ops.transpose2d(x.to(gpu), dst)
    assert y.shape == (S, R) and y.dtype == dst
    assert torch.equal(y.cpu(), x.t().contiguous().to(dst))


def test_transpose2d_of_a_map_with_more_than_four_million_rows(gpu):
    """Round 6: a [16 x 512 x 512 pixels, C] gradient map has 65 536 row tiles of 64 — one more than grid.y takes; the longer tile axis
    rides on grid.x now (the fp32-class heads' training step at bench sizes failed with 'too many rows')."""
    from uniception_amd import ops
    R, S = 65536 * 64 + 70, 8
    x = torch.arange(R * S, device=gpu, dtype=torch.float32).view(R, S) % 1021
    y = ops.transpose2d(x, torch.bfloat16)
    assert y.shape == (S, R) and torch.equal(y, x.t().contiguous().to(torch.bfloat16))
    z = ops.transpose2d(x[:300].t().contiguous(), torch.float32)           # the other orientation: few rows, the long axis in S
    assert torch.equal(z, x[:300])


@pytest.mark.parametrize("P,Cout,h,w", [(16, 4, 3, 5), (14, 4, 2, 2), (2, 3, 4, 4)])
def test_pixel_unshuffle_is_inverse_of_pixel_shuffle(gpu, P, Cout, h, w):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(P)
    B = 2
    rows = torch.randn(B * h * w, Cout * P * P, generator=g)
    img = ops.pixel_shuffle(rows.to(gpu), B, h, w, P, Cout)
    ref = F.pixel_shuffle(rows.view(B, h, w, -1).permute(0, 3, 1, 2), P)
    assert torch.equal(img.cpu(), ref)
    back = ops.pixel_unshuffle(img, P, torch.float32)
    assert torch.equal(back.cpu(), rows)
    # and it is the autograd gradient of pixel_shuffle
    gimg = torch.randn(B, Cout, P * h, P * w, generator=g)
    rr = rows.clone().requires_grad_(True)
    F.pixel_shuffle(rr.view(B, h, w, -1).permute(0, 3, 1, 2), P).backward(gimg)
    assert torch.equal(ops.pixel_unshuffle(gimg.to(gpu), P, torch.float32).cpu(), rr.grad)


# ------------------------------------------------------------------------------------------
def _adaptor_loss_ref(x, gt, alpha):
    """x [B,4,H,W]; the adaptor of adaptors.py:337-342,1080-1083 followed by the confidence-weighted regression loss."""
    xyz = x[:, :3].permute(0, 2, 3, 1)
    d = xyz.norm(dim=-1, keepdim=True)
    pts = xyz / d.clamp(min=1e-8) * torch.expm1(d)
    conf = 1 + x[:, 3].exp()
    r = (pts - gt).norm(dim=-1)
    return (conf * r - alpha * conf.log()).sum()


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_pointmap_loss_forward_backward(gpu, layout):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 24, 40
    x = torch.randn(B, 4, H, W, generator=g, dtype=torch.float64) * 0.7
    x[0, :3, 0, :5] *= 1e-3       # tiny-norm pixels exercise the series branch of s'(d)
    gt = torch.randn(B, H, W, 3, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    loss_ref = _adaptor_loss_ref(xr, gt, 0.2)
    loss_ref.backward()
    xd = x.float().to(gpu)
    if layout == "nhwc":
        xd = xd.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    loss = torch.zeros(1, device=gpu)
    dx = ops.pointmap_loss(xd, gt.float().to(gpu), 0.2, 0.5, loss)
    assert dx.stride() == xd.stride()
    assert abs(float(loss.cpu()) - float(loss_ref.detach())) / abs(float(loss_ref.detach())) < 1e-5
    assert rel_l2(dx.cpu(), 0.5 * xr.grad) < 1e-5


def test_adamw_matches_torch(gpu):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 100_003
    p0 = torch.randn(n, generator=g)
    pref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pref], lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.to(gpu), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        pref.grad = grad.clone()
        opt.step()
        ops.adamw_(p, (grad * 4).to(gpu), m, v, 1e-3, 0.9, 0.95, 1e-8, 0.05, step, grad_scale=0.25)
    assert rel_l2(p.cpu(), pref.detach()) < 1e-6


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,sk", [(768, 768, 6272, 8), (1024, 3072, 4096, 4), (200, 72, 1024, 3)])
def test_gemm_split_k(gpu, M, N, K, sk):
    """weight-gradient shape: few output tiles, long K; every K slice stores an fp32 partial slab, then one reduction."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = torch.randn(N, K, generator=g).bfloat16()
    ws = ops.gemm(a.to(gpu), w.to(gpu), out_dtype=torch.float32, split_k=sk)
    assert ws.shape == (sk, M, N)
    ref = a.float() @ w.float().t()
    c = ops.splitk_reduce(ws)
    assert rel_l2(c.cpu(), ref) < 2e-5
    base = torch.randn(M, N, generator=g)
    acc = ops.splitk_reduce(ws, out=base.to(gpu), accumulate=True)
    assert rel_l2(acc.cpu(), ref + base) < 2e-5


def test_gemm_preact_out(gpu):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 300, 512, 256
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / 16).bfloat16()
    b = torch.randn(N, generator=g)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=gpu)
    y = ops.gemm(a.to(gpu), w.to(gpu), bias=b.to(gpu), act="gelu", out_dtype=torch.bfloat16, preact_out=pre)
    u = a.float() @ w.float().t() + b
    assert rel_l2(pre.cpu().float(), u) < 4e-3
    assert rel_l2(y.cpu().float(), F.gelu(u)) < 5e-3


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 196, 196), (1, 2, 64, 64), (1, 1, 77, 130), (2, 12, 1024, 1024), (1, 2, 33, 300)])
def test_attention_lse_and_backward(gpu, B, H, Nq, Nk):
    from uniception_amd import ops
    D = 64
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q = torch.randn(B, Nq, H, D, generator=g).bfloat16()
    k = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    v = torch.randn(B, Nk, H, D, generator=g).bfloat16()
    do = torch.randn(B, Nq, H, D, generator=g).bfloat16()
    scale = D ** -0.5
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * scale
    o_ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vf)
    o_ref.backward(do.float())
    lse_ref = s.logsumexp(-1)

    qd, kd, vd = q.to(gpu), k.to(gpu), v.to(gpu)
    lse = torch.empty(B, H, Nq, dtype=torch.float32, device=gpu)
    o = ops.attention(qd, kd, ops.vt_pack(vd), scale, v_packed=True, lse=lse)
    assert rel_l2(o.cpu().float(), o_ref.detach()) < 6e-3
    assert float((lse.cpu() - lse_ref.detach()).abs().max()) < 2e-3
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, o, do.to(gpu), lse, scale)
    assert rel_l2(dv.cpu().float(), vf.grad) < 1e-2
    assert rel_l2(dq.cpu().float(), qf.grad) < 1.5e-2
    assert rel_l2(dk.cpu().float(), kf.grad) < 1.5e-2


def test_attention_backward_on_fused_qkv_views(gpu):
    """self-attention layout of the encoder: q/k/v are strided views of one [B,N,3,H,D] buffer."""
    from uniception_amd import ops
    B, N, H, D = 2, 100, 4, 64
    g = torch.Generator().manual_seed(21)
    qkv = torch.randn(B, N, 3, H, D, generator=g).bfloat16()
    do = torch.randn(B, N, H, D, generator=g).bfloat16()
    qf, kf, vf = (qkv[:, :, i].float().requires_grad_(True) for i in range(3))
    F.scaled_dot_product_attention(qf.transpose(1, 2), kf.transpose(1, 2), vf.transpose(1, 2)).transpose(1, 2).backward(do.float())
    dev = qkv.to(gpu)
    qd, kd, vd = dev[:, :, 0], dev[:, :, 1], dev[:, :, 2]
    lse = torch.empty(B, H, N, dtype=torch.float32, device=gpu)
    o = ops.attention(qd, kd, ops.vt_pack(vd), D ** -0.5, v_packed=True, lse=lse)
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, o, do.to(gpu), lse, D ** -0.5)
    for got, ref in ((dq, qf.grad), (dk, kf.grad), (dv, vf.grad)):
        assert rel_l2(got.cpu().float(), ref) < 1.5e-2


def test_attention_backward_with_the_inverse_rope_inside(gpu):
    """rope=(qpos, kpos, base, F0): dq / dk come back as gradients of the UN-rotated q / k — the arithmetic of two uc_rope2d passes with
    -F0 over the plain backward's dq / dk (curope.cpp:21-46 with the sign flipped), done in the kernels' epilogues; cross-attention
    shapes (Nq != Nk, ragged against the 128-row tiles) and separate q / k position grids."""
    from uniception_amd import ops
    B, H, D, scale = 2, 3, 64, 64 ** -0.5
    for (gh, gw, kh, kw) in ((9, 7, 9, 7), (5, 6, 11, 13)):
        Nq, Nk = gh * gw, kh * kw
        g = torch.Generator().manual_seed(Nq * 7 + Nk)
        q = torch.randn(B, Nq, H, D, generator=g).bfloat16().to(gpu)
        kv = torch.randn(B, Nk, 2, H, D, generator=g).bfloat16().to(gpu)
        do = torch.randn(B, Nq, H, D, generator=g).bfloat16().to(gpu)
        qpos = torch.cartesian_prod(torch.arange(gh), torch.arange(gw)).repeat(B, 1).contiguous().to(gpu)
        kpos = (torch.cartesian_prod(torch.arange(kh), torch.arange(kw)) + 3).repeat(B, 1).contiguous().to(gpu)
        k, v = kv[:, :, 0], kv[:, :, 1]
        lse = torch.empty(B, H, Nq, dtype=torch.float32, device=gpu)
        o = ops.attention(q, k, ops.vt_pack(v.contiguous()), scale, v_packed=True, lse=lse)
        dq0, dk0, dv0 = ops.attention_bwd(q, k, v, o, do, lse, scale)
        ops.rope_2d_(dq0, qpos.view(B, Nq, 2), 100.0, -1.0)
        ops.rope_2d_(dk0, kpos.view(B, Nk, 2), 100.0, -1.0)
        dq1, dk1, dv1 = ops.attention_bwd(q, k, v, o, do, lse, scale, rope=(qpos, kpos, 100.0, 1.0))
        assert torch.equal(dv0, dv1)
        assert rel_l2(dq1.float().cpu(), dq0.float().cpu()) < 6e-3 and rel_l2(dk1.float().cpu(), dk0.float().cpu()) < 6e-3   # (one bf16 rounding instead of two)


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 196, 196), (1, 2, 1024, 1024), (1, 2, 1370, 1370), (1, 1, 77, 130), (1, 2, 300, 1000), (2, 2, 1000, 300),
                                       (3, 5, 256, 512), (20, 16, 512, 512)])
def test_attention_backward_64_row_kernels_are_bitwise_the_32_row_kernels(gpu, B, H, Nq, Nk):
    """attn_bwd_dq64_kernel / attn_bwd_dkv64_kernel (attention_bwd64.h; tuning knob attn_bwd64 = 2: wherever the shape allows) run the SAME
    MFMA chains in the same order as the 32-row kernels (knob 0) — persistent workgroups, 64 rows per wave, hand-placed slots: every
    bit of dQ, dK, dV must agree.  Shapes: ragged against the 64-row tiles and the 256-row workgroups, cross-attention both ways,
    (batch, head) counts that are not multiples of the 8 XCDs, and more items than workgroups (20 x 16 x 2 = 640 > 256: item seams);
    with and without the inverse RoPE in the epilogues, on strided views of one fused buffer."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(Nq * 13 + Nk)
    q = torch.randn(B, Nq, 2, H, 64, generator=g).bfloat16().to(gpu)[:, :, 1]
    kv = torch.randn(B, Nk, 2, H, 64, generator=g).bfloat16().to(gpu)
    k, v = kv[:, :, 0], kv[:, :, 1]
    do = torch.randn(B, Nq, H, 64, generator=g).bfloat16().to(gpu)
    lse = torch.empty(B, H, Nq, dtype=torch.float32, device=gpu)
    o = ops.attention(q, k, ops.vt_pack(v.contiguous()), 0.125, v_packed=True, lse=lse)
    qpos = torch.randint(0, 40, (B * Nq, 2), generator=g).to(gpu)
    kpos = torch.randint(0, 40, (B * Nk, 2), generator=g).to(gpu)
    for rope in (None, (qpos, kpos, 100.0, 1.0)):
        with ops.tuning("attn_bwd64", 0):
            ref = ops.attention_bwd(q, k, v, o, do, lse, 0.125, rope=rope)
        with ops.tuning("attn_bwd64", 2):
            got = ops.attention_bwd(q, k, v, o, do, lse, 0.125, rope=rope)
        for a, b, name in zip(got, ref, ("dq", "dk", "dv")):
            assert torch.equal(a, b), f"{name} differs (rope={rope is not None}): max |diff| {float((a.float() - b.float()).abs().max())}"
        assert all(bool(torch.isfinite(t.float()).all()) for t in got)


# ------------------------------------------------------------------------------------------
# DPT head backward helpers
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("geom", [(5, 7, 10, 14, None), (3, 4, 6, 8, (5, 7)), (37, 37, 64, 50, None), (8, 8, 8, 8, None), (1, 1, 2, 2, None)])
def test_bilinear_backward_is_the_adjoint(gpu, dtype, tol, geom):
    from uniception_amd import ops
    Hi, Wi, Ho, Wo, crop = geom
    g = torch.Generator().manual_seed(Hi * 31 + Wo)
    B, C = 2, 16
    x = torch.randn(B, C, Hi, Wi, generator=g).to(dtype).float().requires_grad_(True)
    y = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True)
    if crop is not None:
        y = y[:, :, :crop[0], :crop[1]]
    dy = torch.randn(y.shape, generator=g).to(dtype)
    y.backward(dy.float())
    dx = ops.bilinear_nhwc_bwd(dy.permute(0, 2, 3, 1).contiguous().to(gpu), Hi, Wi, Ho, Wo)
    assert dx.shape == (B, Hi, Wi, C)
    assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2), x.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 4])
def test_convt_gather_inverts_scatter(gpu, dtype, k):
    from uniception_amd import ops
    B, h, w, Cout = 2, 3, 5, 16
    g = torch.Generator().manual_seed(k)
    rows = torch.randn(B * h * w, k * k * Cout, generator=g).to(dtype).to(gpu)
    img = ops.convt_scatter(rows, B, h, w, k, Cout)
    assert torch.equal(ops.convt_gather(img, k), rows)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stride,relu,H,W,Cin", [(1, False, 6, 9, 16), (1, True, 5, 7, 72), (2, False, 5, 7, 16), (2, True, 8, 8, 64)])
def test_im2col_t_matches_unfold(gpu, dtype, stride, relu, H, W, Cin):
    from uniception_amd import ops
    B = 2
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).to(dtype)
    out = ops.im2col_t(x.to(gpu), stride, relu)
    xa = x.float().relu() if relu else x.float()
    cols = F.unfold(xa.permute(0, 3, 1, 2), kernel_size=3, padding=1, stride=stride)      # [B, Cin*9, L], rows (c, ky, kx)
    L = cols.shape[-1]
    ref = cols.view(B, Cin, 9, L).permute(2, 1, 0, 3).reshape(9 * Cin, B * L)              # rows (tap, c), cols (b, oy, ox)
    npix = B * L
    assert out.shape[0] == 9 * Cin and out.shape[1] % 64 == 0 and out.shape[1] >= npix
    assert torch.equal(out[:, :npix].float().cpu(), ref.to(dtype).float())
    assert float(out[:, npix:].abs().sum()) == 0.0


def test_dilate_nhwc(gpu):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(1)
    src = torch.randn(2, 3, 4, 8, generator=g)
    out = ops.dilate_nhwc(src.to(gpu), 5, 7, 2).cpu()
    ref = torch.zeros(2, 5, 7, 8)
    ref[:, ::2, ::2] = src
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("Cin", [128, 16, 72])
def test_conv1x1_to4_backward(gpu, dtype, tol, Cin):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(Cin)
    B, H, W = 2, 9, 13
    feat = torch.randn(B, H, W, Cin, generator=g).to(dtype)
    w = torch.randn(4, Cin, generator=g) / 8
    b = torch.randn(4, generator=g)
    dout = torch.randn(B, H, W, 4, generator=g)
    f, ww, bb = feat.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (f @ ww.t() + bb).backward(dout)
    dw, db = torch.zeros(4, Cin, device=gpu), torch.zeros(4, device=gpu)
    dfeat = ops.conv1x1_to4_bwd(feat.to(gpu), w.to(gpu), dout.to(gpu), dw, db)
    assert rel_l2(dfeat.float().cpu(), f.grad) < tol
    assert rel_l2(dw.cpu(), ww.grad) < 1e-5 and rel_l2(db.cpu(), bb.grad) < 1e-5


# ------------------------------------------------------------------------------------------
# TN contraction (weight gradients without transposes; ds_read_b64_tr_b16 operand loads)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,I,J,sk", [(64, 256, 256, 1), (4096, 768, 768, 5), (1000, 1024, 3072, 2), (777, 72, 200, 3),
                                      (130, 8, 8, 1), (16384, 1024, 1024, 16)])
def test_gemm_tn_dense(gpu, T, I, J, sk):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(T + I)
    a = torch.randn(T, I, generator=g).bfloat16()
    b = torch.randn(T, J, generator=g).bfloat16()
    ws, cs = ops.gemm_tn(a.to(gpu), b.to(gpu), split_k=sk, colsum=True)
    assert ws.shape == (sk, I, J) and cs.shape == (sk, I)
    c = ops.splitk_reduce(ws)
    ref = a.float().t() @ b.float()
    assert rel_l2(c.cpu(), ref) < 2e-5
    # fused bias gradient: column sums of the first operand
    assert rel_l2(cs.sum(0).cpu(), a.float().sum(0)) < 2e-5
    assert torch.equal(ops.splitk_reduce(ops.gemm_tn(a.to(gpu), b.to(gpu), split_k=sk)), c)


def test_gemm_tn_strided_operands(gpu):
    """operands that are column slices of wider buffers (dq|dk|dv views of a fused qkv gradient)."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(5)
    T = 500
    a_full = torch.randn(T, 3 * 128, generator=g).bfloat16()
    b_full = torch.randn(T, 2 * 192, generator=g).bfloat16()
    c = ops.splitk_reduce(ops.gemm_tn(a_full.to(gpu)[:, 128:256], b_full.to(gpu)[:, 192:], split_k=2))
    ref = a_full[:, 128:256].float().t() @ b_full[:, 192:].float()
    assert rel_l2(c.cpu(), ref) < 2e-5


@pytest.mark.parametrize("stride,relu,B,H,W,Cin,Cout,sk", [(1, False, 2, 6, 9, 16, 32, 1), (1, True, 1, 16, 16, 64, 256, 2),
                                                            (2, False, 2, 5, 7, 16, 24, 1), (2, True, 1, 32, 32, 64, 64, 3),
                                                            (1, False, 1, 64, 64, 128, 128, 4)])
def test_gemm_tn_conv_weight_gradient(gpu, stride, relu, B, H, W, Cin, Cout, sk):
    from uniception_amd import ops
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(B, H, W, Cin, generator=g).bfloat16()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.randn(B, Ho, Wo, Cout, generator=g).bfloat16()
    wref = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    xa = x.float().relu() if relu else x.float()
    F.conv2d(xa.permute(0, 3, 1, 2), wref, stride=stride, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ws, cs = ops.gemm_tn(dy.view(-1, Cout).to(gpu), x.to(gpu), split_k=sk, conv=(stride, relu), colsum=True)
    dW = ops.splitk_reduce(ws).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    assert rel_l2(dW.cpu(), wref.grad) < 2e-5
    assert rel_l2(cs.sum(0).cpu(), dy.float().sum((0, 1, 2))) < 2e-5


@pytest.mark.parametrize("relu,B,H,W,Cin,Cout,sk", [(False, 1, 5, 64, 128, 128, 1), (True, 2, 7, 128, 128, 256, 3), (False, 1, 3, 192, 256, 128, 5),
                                                    (True, 3, 64, 64, 256, 256, 21), (False, 1, 1, 64, 128, 128, 2)])
def test_gemm_tn_conv_rows_kernel(gpu, relu, B, H, W, Cin, Cout, sk):
    """Stride-1 convs on maps a multiple of 64 wide with 128-channel tiles take conv_dw_rows_kernel (one kernel row per workgroup, the
    taps kx as row shifts of the staged pixels): the same contract as the implicit-im2col kernel — weight gradient and bias gradient
    against torch's conv backward on the same bf16 inputs, image borders (first / last row and column segments, a one-row image),
    several images, ReLU on load, K slices that do not divide the segment count, more slices than segments."""
    from uniception_amd import ops
    assert ops.gemm_tn_conv_tiles(Cout, H, W, Cin, 1) == 3 * (Cout // 128) * (Cin // 128)
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).bfloat16()
    dy = torch.randn(B, H, W, Cout, generator=g).bfloat16()
    wref = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    xa = x.float().relu() if relu else x.float()
    F.conv2d(xa.permute(0, 3, 1, 2), wref, stride=1, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ws, cs = ops.gemm_tn(dy.view(-1, Cout).to(gpu), x.to(gpu), split_k=sk, conv=(1, relu), colsum=True)
    dW = ops.splitk_reduce(ws).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    assert rel_l2(dW.cpu(), wref.grad) < 2e-5
    assert rel_l2(cs.sum(0).cpu(), dy.float().sum((0, 1, 2))) < 2e-5
    into = torch.full((Cout,), 2.0, device=gpu)
    ws2 = ops.gemm_tn(dy.view(-1, Cout).to(gpu), x.to(gpu), split_k=sk, conv=(1, relu), colsum_into=into)
    assert torch.equal(ws2, ws) and rel_l2((into - 2.0).cpu(), dy.float().sum((0, 1, 2))) < 2e-5


@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_gemm_fused_activation_backward(gpu, act):
    """du = (dy W) * act'(u) in the data-gradient GEMM's epilogue == act_bwd(gemm(dy, W), u)."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(17)
    M, N, K = 777, 512, 256
    dy = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / 16).bfloat16()
    u = (torch.randn(M, N, generator=g) * 1.5).bfloat16()
    fused = ops.gemm(dy.to(gpu), w.to(gpu), dact=(u.to(gpu), act))
    uu = u.float().requires_grad_(True)
    (F.gelu(uu) if act == "gelu" else F.relu(uu)).backward(dy.float() @ w.float().t())
    assert rel_l2(fused.float().cpu(), uu.grad) < 5e-3


@pytest.mark.parametrize("T,I,J,sk", [(4096, 128, 1152, 7), (1000, 64, 256, 2), (300, 128, 72, 1), (2048, 96, 2304, 3)])
def test_gemm_tn_half_height_tile(gpu, T, I, J, sk):
    """I <= 128 takes the 128 x 256 tile (8 waves): same results, no half-empty MFMA tiles."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(T + I + J)
    a = torch.randn(T, I, generator=g).bfloat16()
    b = torch.randn(T, J, generator=g).bfloat16()
    ws, cs = ops.gemm_tn(a.to(gpu), b.to(gpu), split_k=sk, colsum=True)
    assert rel_l2(ops.splitk_reduce(ws).cpu(), a.float().t() @ b.float()) < 2e-5
    assert rel_l2(cs.sum(0).cpu(), a.float().sum(0)) < 2e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("M,H", [(37, 8), (300, 4096), (1, 264)])
def test_swiglu_gate_and_its_backward(gpu, dtype, tol, M, H):
    """uc_swiglu / uc_swiglu_bwd (DINOv2 giant's FFN gate): silu(x1) * x2 of t = [x1 | x2] and its gradient against fp64 autograd
    on the same rounded inputs; ragged row counts, one row, a width that is a multiple of 8 only."""
    from uniception_amd import ops
    g = torch.Generator().manual_seed(M + H)
    t = (2.0 * torch.randn(M, 2 * H, generator=g)).to(dtype)
    dg = torch.randn(M, H, generator=g).to(dtype)
    td = t.double().requires_grad_(True)
    ref = F.silu(td[:, :H]) * td[:, H:]
    ref.backward(dg.double())
    out = ops.swiglu(t.to(gpu))
    assert out.dtype == dtype and out.shape == (M, H)
    assert rel_l2(out.double().cpu(), ref.detach()) < tol
    dt = ops.swiglu_bwd(dg.to(gpu), t.to(gpu))
    assert dt.dtype == dtype and dt.shape == (M, 2 * H)
    assert rel_l2(dt.double().cpu(), td.grad) < tol
    with pytest.raises(Exception):
        ops.swiglu(torch.zeros(4, 12, device=gpu))       # H = 6: not a multiple of 8


@pytest.mark.parametrize("dtype,B,H,Nq,Nk,D", [(torch.bfloat16, 2, 3, 200, 136, 64), (torch.bfloat16, 1, 2, 128, 64, 64), (torch.bfloat16, 2, 2, 70, 300, 64),
                                             (torch.float32, 2, 2, 100, 77, 64), (torch.float32, 1, 3, 33, 50, 32)])
def test_attention_with_dropout_forward_and_backward(gpu, dtype, B, H, Nq, Nk, D):
    """uc_attention_fwd_drop / uc_attention_bwd_drop / uc_attention_bwd_f32_drop against fp32 PyTorch autograd over
    softmax(S) o mask / (1 - p) @ V with the mask uc_attention_drop_mask reports: output, LSE (of the UNdropped scores), dQ, dK, dV;
    ragged query and key counts, strided q / k / v views, the inverse RoPE riding in the bf16 backward as without dropout."""
    from uniception_amd import ops
    p, seed = 0.3, 0x1234_5678_9ABC_DEF1
    g = torch.Generator().manual_seed(Nq * 13 + Nk)
    qkv = torch.randn(B, max(Nq, Nk), 3, H, D, generator=g).to(dtype)
    q, k, v = qkv[:, :Nq, 0], qkv[:, :Nk, 1], qkv[:, :Nk, 2]
    do = torch.randn(B, Nq, H, D, generator=g).to(dtype)
    scale = D ** -0.5
    dev = qkv.to(gpu)
    qd, kd, vd = dev[:, :Nq, 0], dev[:, :Nk, 1], dev[:, :Nk, 2]
    mask = ops.attention_drop_mask(B, H, Nq, Nk, p, seed, gpu)
    assert abs(float(mask.float().mean()) - (1 - p)) < 0.02
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * scale
    pr = s.softmax(-1) * mask.cpu().float() / (1 - p)
    o_ref = torch.einsum("bhqk,bkhd->bqhd", pr, vf)
    o_ref.backward(do.float())
    lse = torch.empty(B, H, Nq, dtype=torch.float32, device=gpu)
    bf = dtype == torch.bfloat16
    o = ops.attention(qd, kd, ops.vt_pack(vd) if bf else vd, scale, v_packed=bf, lse=lse, dropout=(p, seed))
    assert rel_l2(o.cpu().float(), o_ref.detach()) < (8e-3 if bf else 3e-6)
    assert float((lse.cpu() - s.detach().logsumexp(-1)).abs().max()) < (2e-3 if bf else 1e-5)
    dq, dk, dv = ops.attention_bwd(qd, kd, vd, o, do.to(gpu), lse, scale, dropout=(p, seed))
    tol = 1.5e-2 if bf else 1e-5
    assert rel_l2(dv.cpu().float(), vf.grad) < tol
    assert rel_l2(dq.cpu().float(), qf.grad) < tol
    assert rel_l2(dk.cpu().float(), kf.grad) < tol
    # another seed is another mask; p = 0 is the plain kernel
    o2 = ops.attention(qd, kd, ops.vt_pack(vd) if bf else vd, scale, v_packed=bf, dropout=(p, seed + 1))
    assert rel_l2(o2.float(), o.float()) > 0.1
    o0 = ops.attention(qd, kd, ops.vt_pack(vd) if bf else vd, scale, v_packed=bf, dropout=(0.0, seed))
    assert torch.equal(o0, ops.attention(qd, kd, ops.vt_pack(vd) if bf else vd, scale, v_packed=bf))


def test_attention_dropout_mask_statistics(gpu):
    """The counter-based keep function: keep rate 1 - p per (batch, head) and per key column, no correlation between neighbouring
    queries / keys / heads, a different mask per seed."""
    from uniception_amd import ops
    B, H, Nq, Nk = 2, 4, 512, 384
    for p in (0.1, 0.5):
        m = ops.attention_drop_mask(B, H, Nq, Nk, p, 42, gpu).float()
        sd = (p * (1 - p)) ** 0.5                     # (bounds: 5 standard deviations of a mean over that many Bernoulli draws)
        assert float((m.mean((2, 3)) - (1 - p)).abs().max()) < 5 * sd / (Nq * Nk) ** 0.5
        assert float((m.mean((0, 1, 2)) - (1 - p)).abs().max()) < 5 * sd / (B * H * Nq) ** 0.5
        assert float((m.mean((0, 1, 3)) - (1 - p)).abs().max()) < 5 * sd / (B * H * Nk) ** 0.5
        c = m - m.mean()
        var = float((c * c).mean())
        for a, b in ((c[:, :, :-1], c[:, :, 1:]), (c[:, :, :, :-1], c[:, :, :, 1:]), (c[:, :-1], c[:, 1:]), (c[:1], c[1:])):
            assert abs(float((a * b).mean())) / var < 0.01
        m2 = ops.attention_drop_mask(B, H, Nq, Nk, p, 43, gpu).float()
        assert abs(float(((m - m.mean()) * (m2 - m2.mean())).mean())) / var < 0.01
