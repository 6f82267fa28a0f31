"""Stagger experiment on the fp32-residual GEMMs (UC_GEMM_STAGGER = 100-MHz ticks per phase group)."""
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
for tag, M, C in (("enc", 131072, 1024), ("dec", 65536, 768)):
    h = rnd(M, C); x = torch.randn(M, C, device=dev); hid = rnd(M, 4 * C)
    wp = rnd(C, C, scale=1 / 32); w2 = rnd(C, 4 * C, scale=1 / 64); bp = torch.randn(C, device=dev) * 0.1
    w1 = rnd(4 * C, C, scale=1 / 32); b1 = torch.randn(4 * C, device=dev) * 0.1
    out32 = torch.empty(M, C, device=dev)
    cases += [(f"{tag} proj +res32 emit", lambda h=h, wp=wp, bp=bp, x=x, o=out32: ops.gemm(h, wp, bp, residual=x, out=o, emit_ln=True)),
              (f"{tag} fc2 +res32 emit", lambda hid=hid, w2=w2, bp=bp, x=x, o=out32: ops.gemm(hid, w2, bp, residual=x, out=o, emit_ln=True)),
              (f"{tag} fc1 gelu", lambda h=h, w1=w1, b1=b1: ops.gemm(h, w1, b1, act="gelu"))]
for st in (0, 150, 300, 500):
    ops.tuning_set("gemm_stagger", int(st))
    print(f"stagger {st:4d}: " + " | ".join(f"{n} {timeit(f):7.1f}" for n, f in cases), flush=True)
