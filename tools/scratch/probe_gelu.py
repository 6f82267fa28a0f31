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
for name, M, Nn, K in [("enc fc1", 131072, 4096, 1024), ("dec fc1", 65536, 3072, 768)]:
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(Nn, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(Nn, device=dev)
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    for act, bias in [("none", None), ("none", b), ("relu", b), ("gelu", b)]:
        t = timeit(lambda: ops.gemm(a, w, bias, act=act, out=out))
        print(f"{os.environ.get('UC_HIP_LIB','cur')[-16:]} {name} act={act:5s} bias={'y' if bias is not None else 'n'}: {t*1e6:8.1f} us  {2*M*Nn*K/t/1e12:7.1f} TF", flush=True)
