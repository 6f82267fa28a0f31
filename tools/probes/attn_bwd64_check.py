"""attn_bwd_dkv64_kernel against the 32-key kernel and an fp32 reference, and its time (tools/probes: run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def case(B, H, Nq, Nk, ref=True, time=False):
    g = torch.Generator(device="cpu").manual_seed(Nq * 7 + Nk)
    q = torch.randn(B, Nq, H, 64, generator=g).bfloat16().to(dev)
    k = torch.randn(B, Nk, H, 64, generator=g).bfloat16().to(dev)
    v = torch.randn(B, Nk, H, 64, generator=g).bfloat16().to(dev)
    do = torch.randn(B, Nq, H, 64, generator=g).bfloat16().to(dev)
    lse = torch.empty(B, H, Nq, device=dev)
    with ops.tuning("attn_p64", 0):
        o = ops.attention(q, k, ops.vt_pack(v), 0.125, v_packed=True, lse=lse)
    with ops.tuning("attn_bwd64", 0):
        dq0, dk0, dv0 = ops.attention_bwd(q, k, v, o, do, lse, 0.125)
    with ops.tuning("attn_bwd64", 2):
        dq1, dk1, dv1 = ops.attention_bwd(q, k, v, o, do, lse, 0.125)
    msg = f"B={B} H={H} Nq={Nq} Nk={Nk}: new vs old dk {rel(dk1, dk0):.2e} dv {rel(dv1, dv0):.2e} dq {rel(dq1, dq0):.2e} finite {bool(torch.isfinite(dk1.float()).all() and torch.isfinite(dv1.float()).all())}"
    if ref:
        qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
        s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * 0.125
        torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vf).backward(do.float())
        msg += f" | vs fp32: old dk {rel(dk0, kf.grad):.2e} dv {rel(dv0, vf.grad):.2e}; new dk {rel(dk1, kf.grad):.2e} dv {rel(dv1, vf.grad):.2e}"
    if time:
        fl = 2.5 * 4.0 * B * H * Nq * Nk * 64
        with ops.tuning("attn_bwd64", 0):
            t0 = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, 0.125))
        with ops.tuning("attn_bwd64", 2):
            t1 = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, 0.125))
        msg += f" | {t0*1e6:.1f} -> {t1*1e6:.1f} us ({fl/t0/1e12:.0f} -> {fl/t1/1e12:.0f} TF/s algorithmic)"
    print(msg, flush=True)


for shp in [(1, 2, 64, 64), (2, 3, 196, 196), (1, 1, 77, 130), (1, 2, 33, 300), (2, 4, 1024, 1024), (1, 2, 1370, 1370), (1, 2, 300, 1000)]:
    case(*shp)
for shp in [(128, 16, 1024, 1024), (64, 12, 1024, 1024), (16, 16, 4096, 4096), (32, 16, 1370, 1370)]:
    case(*shp, ref=False, time=True)
