import os, sys
sys.path.insert(0, os.getcwd())
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (B, H, N) in [(128, 16, 1024), (64, 12, 1024)]:
    q = torch.randn(B, N, H, 64, device=dev).bfloat16(); k = torch.randn(B, N, H, 64, device=dev).bfloat16(); v = torch.randn(B, N, H, 64, device=dev).bfloat16()
    vt = ops.vt_pack(v); fl = 4.0 * B * H * N * N * 64
    t = timeit(lambda: ops.attention(q, k, vt, 0.125, v_packed=True))
    print(f"RS={os.environ.get('UC_ATTN_RS','0')} PRIO={os.environ.get('UC_ATTN_PRIO','0')} B={B} H={H} N={N}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s", flush=True)
