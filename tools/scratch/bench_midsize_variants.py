"""Tile variants of the dense bf16 GEMM at the batch sweep's mid sizes (2 - 16 pairs: M = 2048 .. 32768 rows): which tile wins where.
Families: plain bf16 store (+ GELU) and the bf16 residual stream (proj / fc2)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")


def t_of(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for pairs in ([int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (2, 4, 8, 16)):
    for (tag, M, N, K, resid, act) in [("enc qkv", pairs * 2048, 3072, 1024, False, None), ("enc proj", pairs * 2048, 1024, 1024, True, None),
                                       ("enc fc1", pairs * 2048, 4096, 1024, False, "gelu"), ("enc fc2", pairs * 2048, 1024, 4096, True, None),
                                       ("dec qkv", pairs * 1024, 2304, 768, False, None), ("dec proj", pairs * 1024, 768, 768, True, None),
                                       ("dec fc1", pairs * 1024, 3072, 768, False, "gelu"), ("dec fc2", pairs * 1024, 768, 3072, True, None)]:
        a = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16() if resid else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        row = []
        for var in (-3, 0, 1, 3, 2, 6, 7, 4):
            with ops.tuning("gemm_variant", var):
                row.append(t_of(lambda: ops.gemm(a, w, b, act=act or "none", residual=res, out=out)))
        best = min(range(1, len(row)), key=lambda i: row[i])
        print(f"{pairs:2d} pairs {tag:9s} M={M:6d} N={N:5d} K={K:5d}: auto {row[0]:7.1f} | v0 {row[1]:7.1f} v1 {row[2]:7.1f} v3 {row[3]:7.1f} v2 {row[4]:7.1f} v6 {row[5]:7.1f} v7 {row[6]:7.1f} v4 {row[7]:7.1f} us"
              f"  best v{(0, 1, 3, 2, 6, 7, 4)[best - 1]} ({100 * (row[0] / row[best] - 1):+.0f} % over auto)", flush=True)
