"""Conv weight gradient (uc_gemm_tn, conv form) at the DPT head's shapes: TFLOP/s.  UC_CONV_DW_ROWS=0 selects the implicit-im2col kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
for (B, H, W, Cin, Cout, relu) in [(32, 512, 512, 128, 128, False), (32, 256, 256, 256, 128, False), (32, 128, 128, 256, 256, True), (32, 64, 64, 256, 256, True), (32, 256, 256, 256, 256, True)]:
    x = torch.randn(B, H, W, Cin, device=dev).bfloat16(); dy = torch.randn(B * H * W, Cout, device=dev).bfloat16()
    sk = max(1, min(dy.shape[0] // 512, 256 // ops.gemm_tn_conv_tiles(Cout, H, W, Cin, 1)))
    f = lambda: ops.splitk_reduce(ops.gemm_tn(dy, x, split_k=sk, conv=(1, relu)))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"B={B} {H}x{W} {Cin}->{Cout} sk={sk}: {t*1e6:9.1f} us  {2.0*B*H*W*9*Cin*Cout/t/1e12:7.1f} TFLOP/s", flush=True)
