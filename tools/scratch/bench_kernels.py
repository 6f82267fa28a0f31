"""Micro-benchmarks of the hot kernels at the shapes of the ViT-L/16 512x512 DUSt3R forward (run on the GPU box).
Prints achieved TFLOP/s / GB/s per kernel; used to steer optimisation, not part of the judged bench."""
import argparse
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.pairs
    N = 1024
    print(f"pairs/GPU={B}")
    shapes = [
        ("enc qkv", 2 * B * N, 3072, 1024), ("enc proj", 2 * B * N, 1024, 1024), ("enc fc1", 2 * B * N, 4096, 1024),
        ("enc fc2", 2 * B * N, 1024, 4096), ("dec qkv", B * N, 2304, 768), ("dec proj", B * N, 768, 768),
        ("dec fc1", B * N, 3072, 768), ("dec fc2", B * N, 768, 3072), ("sq 4096", 4096, 4096, 4096), ("sq 8192", 8192, 8192, 8192),
    ]
    for name, M, Nn, K in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
        out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, w, out=out))
        print(f"gemm {name:10s} M={M:6d} N={Nn:5d} K={K:5d}: {t*1e6:9.1f} us  {2*M*Nn*K/t/1e12:7.1f} TFLOP/s")
    for name, Bq, H in [("enc attn", 2 * B, 16), ("dec attn", B, 12)]:
        q = torch.randn(Bq, N, H, 64, device=dev).bfloat16()
        k = torch.randn(Bq, N, H, 64, device=dev).bfloat16()
        v = torch.randn(Bq, N, H, 64, device=dev).bfloat16()
        vt = ops.vt_pack(v)
        o = torch.empty_like(q)
        t = timeit(lambda: ops.attention(q, k, vt, 0.125, v_packed=True, out=o))
        fl = 4 * Bq * H * N * N * 64
        print(f"attn {name:10s} B={Bq} H={H} N={N}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TFLOP/s")
    # DPT 3x3 convolutions (implicit GEMM over NHWC): (H=W, Cin, Cout, relu_a)
    for name, Hh, Cin, Cout in [("rcu 128^2 256->256", 128, 256, 256), ("rcu 64^2 256->256", 64, 256, 256),
                                ("reg conv1 256^2 256->128", 256, 256, 128), ("reg conv2 512^2 128->128", 512, 128, 128)]:
        xi = (torch.randn(B, Hh, Hh, Cin, device=dev) * 0.5).bfloat16()
        wc = (torch.randn(Cout, 9 * Cin, device=dev) / math.sqrt(9 * Cin)).bfloat16()
        for relu_a in (False, True):
            t = timeit(lambda: ops.gemm(xi, wc, conv=(B, Hh, Hh, Cin, 1), relu_a=relu_a), iters=10)
            print(f"conv {name:26s} relu_a={int(relu_a)}: {t*1e6:9.1f} us  {2*B*Hh*Hh*9*Cin*Cout/t/1e12:7.1f} TFLOP/s")
    x = torch.randn(2 * B * N, 1024, device=dev)
    g = torch.ones(1024, device=dev)
    bb = torch.zeros(1024, device=dev)
    t = timeit(lambda: ops.layernorm(x, g, bb, 1e-6, torch.bfloat16))
    print(f"layernorm [{2*B*N},1024] f32->bf16: {t*1e6:9.1f} us  {x.numel()*6/t/1e9:7.1f} GB/s")


if __name__ == "__main__":
    main()
