"""3x3 conv forward at the DPT head's shapes, every kernel form side by side (conv_rows 0 implicit GEMM, 2 the 256-pixel row-walking
kernel, 3 the eight-wave 512-pixel one): us and TFLOP/s.  argv: [f16] [relu] [res: two 16-bit residuals, no activation — the residual conv
units' second convolution]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
dt = torch.float16 if "f16" in sys.argv else torch.bfloat16
relu_a = "relu" in sys.argv
res = "res" in sys.argv
for (B, H, W, Cin, Cout, tail) in [(64, 512, 512, 128, 128, True), (64, 512, 512, 128, 128, False), (64, 256, 256, 256, 128, False), (64, 256, 256, 256, 256, False),
                                   (128, 128, 128, 256, 256, False), (64, 128, 128, 256, 256, False)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(dt); w = (torch.randn(Cout, 9 * Cin, device=dev) / math.sqrt(9 * Cin)).to(dt)
    b = torch.randn(Cout, device=dev)
    t4 = (torch.randn(4, 128, device=dev) / 11.0, torch.randn(4, device=dev)) if tail and dt == torch.bfloat16 else None
    if res and tail:
        continue
    r1 = torch.randn(B * H * W, Cout, device=dev).to(dt) if res else None
    r2 = torch.randn(B * H * W, Cout, device=dev).to(dt) if res else None
    line = f"B={B} {H}x{W} {Cin}->{Cout}{' +tail' if t4 else ''} {str(dt)[6:]}{' relu_a' if relu_a else ''}{' +2 residuals' if res else ''}:"
    for mode in (0, 2, 3):
        ops.tuning_set("conv_rows", mode)
        f = (lambda: ops.gemm(x, w, b, conv=(B, H, W, Cin, 1), relu_a=relu_a, residual=r1, residual2=r2)) if res else \
            (lambda: ops.gemm(x, w, b, conv=(B, H, W, Cin, 1), act="relu", relu_a=relu_a, tail=t4))
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        line += f"  [{mode}] {t*1e6:8.1f} us {2.0*B*H*W*9*Cin*Cout/t/1e12:6.1f} TF"
    print(line, flush=True)
