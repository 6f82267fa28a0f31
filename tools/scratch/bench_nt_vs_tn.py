"""NT (uc_gemm) vs TN (uc_gemm_tn) MFMA throughput on matching shapes; HIP-event timing on the launch stream."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def nt(M, N, K, sk=1):
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    if sk == 1:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, w, out=out))
    else:
        ws = torch.empty(sk, M, N, device=dev)
        t = timeit(lambda: ops.gemm(a, w, out=ws, out_dtype=torch.float32, split_k=sk))
    print(f"NT  M={M:6d} N={N:5d} K={K:6d} sk={sk:2d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)


def tn(T, I, J, sk):
    a = (torch.randn(T, I, device=dev) * 0.5).bfloat16()
    b = (torch.randn(T, J, device=dev) * 0.5).bfloat16()
    t = timeit(lambda: ops.gemm_tn(a, b, split_k=sk))
    print(f"TN  T={T:6d} I={I:5d} J={J:5d} sk={sk:2d}: {t*1e6:8.1f} us  {2*T*I*J/t/1e12:7.1f} TFLOP/s", flush=True)


# forward shapes at 32 pairs (M = 65536 tokens) and the weight-gradient shapes at 8 pairs (T = 16384)
for (M, N, K) in [(65536, 4096, 1024), (65536, 1024, 4096), (65536, 3072, 1024), (65536, 1024, 1024), (16384, 4096, 1024)]:
    nt(M, N, K)
for (M, N, K, sk) in [(4096, 1024, 16384, 4), (1024, 1024, 16384, 16), (4096, 4096, 16384, 1), (8192, 8192, 8192, 1), (4096, 4096, 1024, 1)]:
    nt(M, N, K, sk)
for (T, I, J, sk) in [(16384, 4096, 1024, 4), (16384, 1024, 1024, 16), (16384, 4096, 4096, 1), (8192, 8192, 8192, 1), (1024, 4096, 4096, 1)]:
    tn(T, I, J, sk)


def nt_epi(M, N, K, act):
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, w, b, act=act, out=out))
    print(f"NT+bias+{act}  M={M:6d} N={N:5d} K={K:6d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)


nt_epi(65536, 4096, 1024, "gelu")
nt_epi(65536, 4096, 1024, None)
nt_epi(65536, 3072, 768, "gelu")
