"""Four-wave GEMM (gemm_variant 7) of the loaded library (UC_HIP_LIB selects a tools/_libs variant) against the automatic choice:
bitwise check on three epilogue families, then sustained TFLOP/s per shape.  usage: python tools/probe_g4.py [secs]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
M, N, K = 2048 * 20, 1024, 1024        # 640 tiles: 2-3 per workgroup of the persistent form
a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(dev); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(dev)
b = torch.randn(N, generator=g).to(dev); res = torch.randn(M, N, generator=g).to(dev)
bad = 0
for K2 in (64, 128, 256, 384, 1024):
    for name, fn in {"bf16 plain": lambda: ops.gemm(a[:, :K2].contiguous(), w[:, :K2].contiguous(), b).float(),
                     "bf16 gelu": lambda: ops.gemm(a[:, :K2].contiguous(), w[:, :K2].contiguous(), b, act="gelu").float(),
                     "f32 bias+res": lambda: ops.gemm(a[:, :K2].contiguous(), w[:, :K2].contiguous(), b, residual=res, out_dtype=torch.float32)}.items():
        with ops.tuning("gemm_variant", 2): y2 = fn()
        for var in (7, 8):
            with ops.tuning("gemm_variant", var): y7 = fn()
            n = int((y2 != y7).sum()); bad += n
            if n: print(f"K={K2} variant {var} {name}: {n} / {y2.numel()} differ, max {float((y2 - y7).abs().max()):.3e}")
print("bitwise:", "OK" if bad == 0 else f"FAILED ({bad})", flush=True)
for (m, n, k) in [(8192, 8192, 8192), (131072, 4096, 1024), (131072, 1024, 1024), (131072, 1024, 4096), (131072, 3072, 1024)]:
    a = (torch.randn(m, k, device=dev) * 0.5).bfloat16(); w = (torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    r = []
    for var in (-3, 7, 8):
        with ops.tuning("gemm_variant", var):
            for _ in range(5): ops.gemm(a, w, out=out)
            torch.cuda.synchronize(); t0 = time.time(); it = 0
            while time.time() - t0 < secs:
                for _ in range(10): ops.gemm(a, w, out=out)
                torch.cuda.synchronize(); it += 10
            r.append(2 * m * n * k * it / (time.time() - t0) / 1e12)
    line = f"M={m} N={n} K={k}: auto {r[0]:7.1f}  four-wave {r[1]:7.1f}  persistent {r[2]:7.1f}"
    for stg in [int(x) for x in os.environ.get("STAGGER", "").split(",") if x]:
        ops.tuning_set("gemm_stagger", stg)
        with ops.tuning("gemm_variant", 8):
            for _ in range(5): ops.gemm(a, w, out=out)
            torch.cuda.synchronize(); t0 = time.time(); it = 0
            while time.time() - t0 < secs:
                for _ in range(10): ops.gemm(a, w, out=out)
                torch.cuda.synchronize(); it += 10
            line += f"  stg{stg} {2 * m * n * k * it / (time.time() - t0) / 1e12:7.1f}"
        ops.tuning_set("gemm_stagger", -1)
    print(line + " TFLOP/s", flush=True)
