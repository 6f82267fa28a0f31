"""bf16 vs fp8 attention forward at the model's shapes."""
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


for (B, H, N) in [(64, 16, 1024), (32, 12, 1024), (16, 16, 4096), (8, 12, 4096), (64, 16, 1370)]:
    q = torch.randn(B, N, H, 64, device=dev).bfloat16()
    k = torch.randn(B, N, H, 64, device=dev).bfloat16()
    v = torch.randn(B, N, H, 64, device=dev).bfloat16()
    vt, vt8 = ops.vt_pack(v), ops.vt_pack_fp8(v)
    fl = 4.0 * B * H * N * N * 64
    t16 = timeit(lambda: ops.attention(q, k, vt, 0.125, v_packed=True))
    t8 = timeit(lambda: ops.attention_fp8(q, k, vt8, 0.125))   # includes the K pre-pack when N % 64 == 0
    tp = timeit(lambda: ops.vt_pack_fp8(v))
    print(f"B={B} H={H} N={N}: bf16 {t16*1e6:8.1f} us {fl/t16/1e12:7.1f} TF/s | fp8 {t8*1e6:8.1f} us {fl/t8/1e12:7.1f} TF/s | "
          f"fp8 V pack {tp*1e6:6.1f} us", flush=True)
