"""Calibration only (not part of the product or the judged bench): what does the vendor GEMM (torch.mm -> hipBLASLt) reach
at the forward's shapes on this box, next to uc_gemm without an epilogue?  Tells how far the hand-written kernel is from
the practical ceiling and which macro-tile the vendor heuristics pick (kernel names via rocprofv3 --kernel-trace)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 1024
shapes = [("enc qkv", 2*B*N, 3072, 1024), ("enc proj", 2*B*N, 1024, 1024), ("enc fc1", 2*B*N, 4096, 1024), ("enc fc2", 2*B*N, 1024, 4096),
          ("dec qkv", B*N, 2304, 768), ("dec proj", B*N, 768, 768), ("dec kv", B*N, 1536, 768), ("dec fc1", B*N, 3072, 768),
          ("dec fc2", B*N, 768, 3072), ("sq 8192", 8192, 8192, 8192)]
for name, M, Nn, K in shapes:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    t_uc = timeit(lambda: ops.gemm(a, w, out=out))
    wt = w.t()
    t_v = timeit(lambda: torch.mm(a, wt, out=out))
    fl = 2 * M * Nn * K
    print(f"{name:9s} M={M:6d} N={Nn:5d} K={K:5d}: uc_gemm {t_uc*1e6:8.1f} us {fl/t_uc/1e12:7.1f} TF | hipBLASLt {t_v*1e6:8.1f} us {fl/t_v/1e12:7.1f} TF", flush=True)
