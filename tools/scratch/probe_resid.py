"""proj / fc2 shapes with the in-place fp32 residual epilogue (what the model runs)."""
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
for name, M, Nn, K in [("enc proj", 131072, 1024, 1024), ("enc fc2", 131072, 1024, 4096), ("dec proj", 65536, 768, 768), ("dec fc2", 65536, 768, 3072)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    x = torch.randn(M, Nn, device=dev)
    b = torch.randn(Nn, device=dev)
    t = timeit(lambda: ops.gemm(a, w, b, out=x, residual=x))
    print(f"stg={os.environ.get('UC_GEMM_STAGGER','0'):>4s} {name:9s}: {t*1e6:8.1f} us  {2*M*Nn*K/t/1e12:7.1f} TF  HBM-side {M*Nn*8/t/1e12:5.2f} TB/s", flush=True)
