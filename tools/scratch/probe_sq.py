"""One shape, 10 launches: the 8192^3 GEMM (and the fc1 shape) for PMC counter collection."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
for M, Nn, K in [(8192, 8192, 8192), (131072, 4096, 1024)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
