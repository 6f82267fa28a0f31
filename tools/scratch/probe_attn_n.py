import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (B, H, N) in [(64, 16, 1024), (64, 16, 1280), (64, 16, 1344), (64, 16, 1370), (64, 16, 1408), (64, 16, 1536), (32, 16, 1370), (64, 16, 2048)]:
    q = torch.randn(B, N, H, 64, device=dev).bfloat16(); k = torch.randn(B, N, H, 64, device=dev).bfloat16(); v = torch.randn(B, N, H, 64, device=dev).bfloat16()
    vt = ops.vt_pack(v)
    t = timeit(lambda: ops.attention(q, k, vt, 0.125, v_packed=True))
    print(f"dma={os.environ.get('UC_ATTN_DMA','1')} B={B} H={H} N={N}: {t*1e6:8.1f} us {4.0*B*H*N*N*64/t/1e12:7.1f} TF/s", flush=True)
