"""bf16 attention backward (uc_attention_bwd: dQ kernel + dK/dV kernel) at the model's shapes: time per call and TFLOP/s on the five
algorithmic products (2.5x the forward's flops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (B, H, N) in [(128, 16, 1024), (64, 12, 1024), (16, 16, 4096), (32, 16, 1370)]:
    q = torch.randn(B, N, H, 64, device=dev).bfloat16()
    k = torch.randn(B, N, H, 64, device=dev).bfloat16()
    v = torch.randn(B, N, H, 64, device=dev).bfloat16()
    do = torch.randn(B, N, H, 64, device=dev).bfloat16()
    lse = torch.empty(B, H, N, device=dev)
    with ops.tuning("attn_p64", 0):
        o = ops.attention(q, k, ops.vt_pack(v), 0.125, v_packed=True, lse=lse)
    fl = 2.5 * 4.0 * B * H * N * N * 64
    t = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, 0.125))
    print(f"B={B} H={H} N={N}: backward {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s algorithmic ({fl/t/2.5e15:.3f} of the bf16 peak)", flush=True)
