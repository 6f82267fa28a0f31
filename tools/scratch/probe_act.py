"""fc1-shaped GEMM under each activation epilogue: wall time and per-workgroup epilogue time (UC_GEMM_TRACE=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
M, C = 131072, 1024
h = (torch.randn(M, C, device=dev) * 0.5).bfloat16()
w1 = (torch.randn(4 * C, C, device=dev) / 32).bfloat16(); b1 = torch.randn(4 * C, device=dev) * 0.1
for act in (None, "relu", "gelu"):
    print(f"fc1 act={act}", flush=True); sys.stderr.flush()
    for _ in range(2): ops.gemm(h, w1, b1, act=act); torch.cuda.synchronize()
