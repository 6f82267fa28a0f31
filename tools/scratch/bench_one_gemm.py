"""Run one bf16 GEMM shape a few times (for rocprofv3 --pmc passes)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
