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
B, N, C, H = 128, 1024, 1024, 16
a = (torch.randn(B * N, C, device=dev) * 0.5).bfloat16()
w = (torch.randn(3 * C, C, device=dev) / math.sqrt(C)).bfloat16()
b = torch.randn(3 * C, device=dev)
vt = ops.vt_buffer(B, H, N, dev)
for _ in range(2):
    t = timeit(lambda: ops.gemm(a, w, b, vt=(2 * C, vt, N)))
    print(f"dbg={os.environ.get('UC_GEMM_DBG','0')} enc qkv with VT epilogue: {t*1e6:8.1f} us {2*B*N*3*C*C/t/1e12:7.1f} TF", flush=True)
