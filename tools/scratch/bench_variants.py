"""Long-K NT GEMM under the tile variants (UC_GEMM_VARIANT): steady-state main-loop comparison."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(8192, 8192, 8192), (16384, 4096, 4096), (65536, 4096, 1024), (65536, 1024, 4096)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"variant {os.environ.get('UC_GEMM_VARIANT','auto')}: M={M} N={N} K={K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)
