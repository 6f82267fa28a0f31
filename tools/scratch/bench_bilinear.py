import torch, sys, os
sys.path.insert(0, "/root/repo")
from uniception_amd import ops
dev = torch.device("cuda:0")
for (B, H, W, C) in [(64, 256, 256, 128), (64, 128, 128, 256), (64, 64, 64, 256)]:
    x = torch.randn(B, H, W, C, device=dev).bfloat16()
    f = lambda: ops.bilinear_nhwc(x, 2 * H, 2 * W)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    gb = (x.numel() * 2 * 5) / 1e9
    print(f"rows2={os.environ.get('UC_BILINEAR_ROWS2','1')} B={B} {H}x{W}x{C}: {t*1e6:8.1f} us  {gb/t/1e3:.2f} TB/s")
