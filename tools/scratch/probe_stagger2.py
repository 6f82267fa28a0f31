"""Stagger experiment on the bf16-store GEMMs (UC_GEMM_STAGGER forces it for every family)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
def rnd(*s, scale=0.5): return (torch.randn(*s, device=dev) * scale).bfloat16()
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = []
N = 1024
for tag, Bimg, C, H in (("enc", 128, 1024, 16), ("dec", 64, 768, 12)):
    M = Bimg * N
    h = rnd(M, C)
    pos = torch.cartesian_prod(torch.arange(32), torch.arange(32)).repeat(Bimg, 1).to(dev).contiguous()
    table = ops.rope_table(dev, 1024, 100.0, 1.0)
    wq = rnd(3 * C, C, scale=1 / 32); bq = torch.randn(3 * C, device=dev) * 0.1
    w1 = rnd(4 * C, C, scale=1 / 32); b1 = torch.randn(4 * C, device=dev) * 0.1
    vt = ops.vt_buffer(Bimg, H, N, dev)
    cases += [(f"{tag} qkv rope+vt", lambda h=h, wq=wq, bq=bq, pos=pos, vt=vt, C=C: ops.gemm(h, wq, bq, rope=(pos, table, 2 * C), vt=(2 * C, vt, N))),
              (f"{tag} fc1 gelu", lambda h=h, w1=w1, b1=b1: ops.gemm(h, w1, b1, act="gelu"))]
for rep in range(2):
    for st in (0, 50, 100, 200):
        ops.tuning_set("gemm_stagger", int(st))
        print(f"stagger {st:4d}: " + " | ".join(f"{n} {timeit(f):7.1f}" for n, f in cases), flush=True)
