"""bf16 attention forward: the persistent 64-queries-per-wave kernel (attn_p64 = 2) against the eight-wave DMA kernel (attn_p64 = 0),
same process, same box, interleaved rounds; plus max-abs difference of the two outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


shapes = [(256, 16, 1024), (128, 16, 1024), (128, 12, 1024), (64, 16, 1370), (16, 16, 4096), (8, 12, 4096), (512, 12, 196), (16, 16, 1024), (2, 16, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (B, H, N) in shapes:
    q = torch.randn(B, N, H, 64, device=dev).bfloat16()
    k = torch.randn(B, N, H, 64, device=dev).bfloat16()
    v = torch.randn(B, N, H, 64, device=dev).bfloat16()
    vt = ops.vt_pack(v)
    fl = 4.0 * B * H * N * N * 64
    outs, best = {}, {0: 1e9, 2: 1e9}
    for rnd in range(3):
        for mode in (0, 2):
            with ops.tuning("attn_p64", mode):
                outs[mode] = ops.attention(q, k, vt, 0.125, v_packed=True)
                best[mode] = min(best[mode], timeit(lambda: ops.attention(q, k, vt, 0.125, v_packed=True)))
    d = (outs[0].float() - outs[2].float()).abs().max().item()
    print(f"B={B} H={H} N={N}: eight-wave {best[0]*1e6:8.1f} us {fl/best[0]/1e12:7.1f} TF/s | p64 {best[2]*1e6:8.1f} us {fl/best[2]/1e12:7.1f} TF/s "
          f"({best[0]/best[2]:.3f}x) | max |diff| {d:.4f}", flush=True)
