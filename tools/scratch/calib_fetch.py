"""FETCH_SIZE / WRITE_SIZE calibration workload (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
access pattern").  Launches, each a few times, with footprints far past the 256-MiB Infinity Cache:
  * convert bf16 -> fp16 over 1 GiB of input (16 B / lane streaming read and write: bytes known exactly);
  * the dense GEMM at N = 256 — ONE column of 256-wide tiles, so A (M x K) can only be read once: bytes known exactly,
    through the kernel's own LDS-DMA access pattern;
  * the model's four ViT-L shapes (qkv / proj / fc1 / fc2 at 64 pairs) for the per-shape re-read factor.
Run under rocprofv3 --pmc by tools/calib_fetch.sh; rows are told apart by grid size."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
M = 131072
x = torch.randn(512 * 1024 * 1024, device=dev).bfloat16()          # 1 GiB
for _ in range(3):
    y = ops.convert(x, torch.float16)
del x, y
shapes = [(M, 256, 1024), (M, 256, 4096), (M, 3072, 1024), (M, 1024, 1024), (M, 4096, 1024), (M, 1024, 4096)]
for (m, n, k) in shapes:
    a = (torch.randn(m, k, device=dev) * 0.5).bfloat16(); w = (torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    print(f"gemm M={m} N={n} K={k}: A {m*k*2/1e6:.1f} MB, W {n*k*2/1e6:.1f} MB, C {m*n*2/1e6:.1f} MB; grid = {((m+255)//256)*((n+255)//256)} workgroups")
