"""3x3 conv forward (implicit GEMM / row-walking kernel) at the DPT head's shapes: TFLOP/s.  UC_CONV_ROWS=0 selects the implicit-GEMM kernel."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
for (B, H, W, Cin, Cout) in [(64, 512, 512, 128, 128), (64, 256, 256, 256, 128), (64, 128, 128, 256, 256), (64, 64, 64, 256, 256)]:
    x = torch.randn(B, H, W, Cin, device=dev).to(dt); w = (torch.randn(Cout, 9 * Cin, device=dev) / math.sqrt(9 * Cin)).to(dt)
    b = torch.randn(Cout, device=dev)
    f = lambda: ops.gemm(x, w, b, conv=(B, H, W, Cin, 1), act="relu")
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"B={B} {H}x{W} {Cin}->{Cout} {str(dt)[6:]}: {t*1e6:9.1f} us  {2.0*B*H*W*9*Cin*Cout/t/1e12:7.1f} TFLOP/s", flush=True)
