"""3x3 implicit-GEMM conv throughput at the DPT head's shapes (32 images per head)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (B, H, W, Cin, Cout, relu) in [(32, 128, 128, 256, 256, True), (32, 64, 64, 256, 256, True), (32, 32, 32, 256, 256, False),
                                   (32, 256, 256, 256, 128, False), (32, 512, 512, 128, 128, False), (32, 128, 128, 96, 256, False)]:
    x = (torch.randn(B, H, W, Cin, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (3 * Cin ** 0.5)).bfloat16()
    b = torch.randn(Cout, device=dev)
    t = timeit(lambda: ops.gemm(x, w, b, relu_a=relu, conv=(B, H, W, Cin, 1)))
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"conv B={B} {H}x{W} Cin={Cin} Cout={Cout} relu_in={relu}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s", flush=True)
