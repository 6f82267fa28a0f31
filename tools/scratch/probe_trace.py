"""Per-workgroup timeline of one GEMM launch (UC_GEMM_TRACE=1): prologue / K-loop / epilogue / same-CU dispatch gap."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
for name, M, Nn, K, f32out in [("enc fc1", 131072, 4096, 1024, False), ("enc proj", 131072, 1024, 1024, True), ("sq8192", 8192, 8192, 8192, False)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    res = torch.randn(M, Nn, device=dev) if f32out else None
    for _ in range(3):
        ops.gemm(a, w, out=out, residual=res)
    torch.cuda.synchronize()
