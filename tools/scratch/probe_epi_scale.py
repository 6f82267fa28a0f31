"""Is the fp32-residual epilogue bound per CU or by the chip's HBM burst?  Same tile work with 32 .. 256 workgroups resident (one tile each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
C = 1024
for tiles_m in (8, 16, 32, 64, 128, 512):
    M = tiles_m * 256
    h = (torch.randn(M, C, device=dev) * 0.5).bfloat16(); x = torch.randn(M, C, device=dev)
    wp = (torch.randn(C, C, device=dev) / 32).bfloat16(); bp = torch.randn(C, device=dev) * 0.1
    out32 = torch.empty(M, C, device=dev)
    for name, fn in (("res32", lambda: ops.gemm(h, wp, bp, residual=x, out=out32)), ("bf16 ", lambda: ops.gemm(h, wp, bp))):
        print(f"tiles {tiles_m * 4:5d} {name}", flush=True); sys.stderr.flush()
        fn(); fn(); torch.cuda.synchronize()
