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
for name, M, Nn, K in [("n128 k256", 2097152, 128, 256), ("n96 k1024", 131072, 96, 1024), ("n384 k768", 131072, 384, 768), ("n192 k768", 131072, 192, 768), ("n128 k1152", 1048576, 128, 1152)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, w, out=out))
    print(f"co={os.environ.get('UC_GEMM_CORESIDENT','1')} {name:12s}: {t*1e6:8.1f} us  {2*M*Nn*K/t/1e12:7.1f} TF", flush=True)
