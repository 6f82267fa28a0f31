"""Stress: the 64-row attention backward kernels against the 32-row ones, bit by bit, many repetitions on small ragged shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
bad = 0
for (B, H, Nq, Nk) in [(2, 3, 196, 196), (2, 2, 1000, 300), (4, 4, 512, 200), (3, 5, 256, 130), (1, 16, 1024, 1024), (40, 16, 196, 196), (5, 3, 700, 260), (2, 2, 260, 700)]:
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q = torch.randn(B, Nq, H, 64, generator=g).bfloat16().to(dev); k = torch.randn(B, Nk, H, 64, generator=g).bfloat16().to(dev)
    v = torch.randn(B, Nk, H, 64, generator=g).bfloat16().to(dev); do = torch.randn(B, Nq, H, 64, generator=g).bfloat16().to(dev)
    lse = torch.empty(B, H, Nq, device=dev)
    o = ops.attention(q, k, ops.vt_pack(v), 0.125, v_packed=True, lse=lse)
    with ops.tuning("attn_bwd64", 0):
        ref = ops.attention_bwd(q, k, v, o, do, lse, 0.125)
    n_bad = 0
    for it in range(60):
        with ops.tuning("attn_bwd64", 2):
            got = ops.attention_bwd(q, k, v, o, do, lse, 0.125)
        if it % 7 == 0:      # (other work in between: different timing)
            _ = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)
        n_bad += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
    print((B, H, Nq, Nk), "mismatching repetitions:", n_bad, flush=True)
    bad += n_bad
print("TOTAL", bad)
