"""The forward's dense GEMM calls with their real epilogues (RoPE + VT, fp32 residual, erf-GELU, plain bf16), per tile variant.
usage: python tools/bench_model_gemms.py [pairs=64] [variants=auto,3,4,5] [which=enc,dec]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = (sys.argv[2] if len(sys.argv) > 2 else "auto,3,4,5").split(",")
which = (sys.argv[3] if len(sys.argv) > 3 else "enc,dec").split(",")
N = 1024


def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s, scale=0.5): return (torch.randn(*s, device=dev) * scale).bfloat16()


def cases(tag, Bimg, C, H):
    M = Bimg * N
    h = rnd(M, C); x = torch.randn(M, C, device=dev); hid = rnd(M, 4 * C)
    pos = torch.cartesian_prod(torch.arange(32), torch.arange(32)).repeat(Bimg, 1).to(dev).contiguous()
    table = ops.rope_table(dev, 1024, 100.0, 1.0)
    w = lambda n, k: rnd(n, k, scale=1 / math.sqrt(k))
    b = lambda n: torch.randn(n, device=dev) * 0.1
    wqkv, bqkv, wp, bp, w1, b1, w2, b2 = w(3 * C, C), b(3 * C), w(C, C), b(C), w(4 * C, C), b(4 * C), w(C, 4 * C), b(C)
    vt = ops.vt_buffer(Bimg, H, N, dev)
    out32 = torch.empty(M, C, device=dev)
    yield f"{tag} qkv rope+vt", 2 * M * 3 * C * C, lambda: ops.gemm(h, wqkv, bqkv, rope=(pos, table, 2 * C), vt=(2 * C, vt, N))
    yield f"{tag} qkv plain  ", 2 * M * 3 * C * C, lambda: ops.gemm(h, wqkv, bqkv)
    yield f"{tag} proj +res32", 2 * M * C * C, lambda: ops.gemm(h, wp, bp, residual=x, out=out32)
    yield f"{tag} proj plain ", 2 * M * C * C, lambda: ops.gemm(h, wp, bp)
    yield f"{tag} fc1 gelu   ", 2 * M * 4 * C * C, lambda: ops.gemm(h, w1, b1, act="gelu")
    yield f"{tag} fc1 plain  ", 2 * M * 4 * C * C, lambda: ops.gemm(h, w1, b1)
    yield f"{tag} fc2 +res32 ", 2 * M * 4 * C * C, lambda: ops.gemm(hid, w2, b2, residual=x, out=out32)
    if tag == "dec":
        wkv, bkv = w(2 * C, C), b(2 * C)
        yield f"{tag} projq rope ", 2 * M * C * C, lambda: ops.gemm(h, wp, bp, rope=(pos, table, C))
        yield f"{tag} kv rope+vt ", 2 * M * 2 * C * C, lambda: ops.gemm(h, wkv, bkv, rope=(pos, table, C), vt=(C, vt, N))


res = {}
for tag, Bimg, C, H in (("enc", 2 * B, 1024, 16), ("dec", B, 768, 12)):
    if tag not in which: continue
    for name, fl, fn in cases(tag, Bimg, C, H):
        for v in variants:
            if v == "auto": ops.tuning_set("gemm_variant", -3)
            else: ops.tuning_set("gemm_variant", int(v))
            t = timeit(fn)
            res[(name, v)] = (t, fl)
        print(f"{name}: " + " | ".join(f"v{v} {res[(name, v)][0]*1e6:7.1f}us {fl/res[(name, v)][0]/1e12:6.1f}TF" for v in variants), flush=True)
ops.tuning_set("gemm_variant", -3)
