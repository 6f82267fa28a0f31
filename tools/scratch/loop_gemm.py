"""Run one GEMM shape for N seconds (power / clock probes)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
M, N, K, secs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
kind = sys.argv[5] if len(sys.argv) > 5 else "uc"
dev = torch.device("cuda:0")
a = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); wt = w.t()
fn = (lambda: ops.gemm(a, w, out=out)) if kind == "uc" else (lambda: torch.mm(a, wt, out=out))
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
print(f"{kind} M={M} N={N} K={K}: {2*M*N*K*n/dt/1e12:.1f} TFLOP/s over {dt:.1f}s")
