"""Decompose the per-tile fixed cost of the 256x256 GEMM: run with UC_GEMM_DBG=0/4/8/12 (4: no epilogue, 8: one K-step)."""
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
for name, M, Nn, K, f32out in [("enc fc1", 131072, 4096, 1024, False), ("enc proj", 131072, 1024, 1024, True), ("sq8192", 8192, 8192, 8192, False),
                               ("1 round", 4096, 4096, 1024, False), ("dec qkv", 65536, 2304, 768, False)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    res = torch.randn(M, Nn, device=dev) if f32out else None
    t = timeit(lambda: ops.gemm(a, w, out=out, residual=res))
    tiles = math.ceil(M / 256) * math.ceil(Nn / 256)
    print(f"DBG={os.environ.get('UC_GEMM_DBG','0'):>2s} {name:9s}: {t*1e6:8.1f} us  {2*M*Nn*K/t/1e12:7.1f} TF  per-tile-round {t*1e6/max(1,tiles/256):6.2f} us", flush=True)
