"""The 16-wave (variant 2) and eight-wave (variant 6) 256x256 GEMM kernels against each other, bit by bit, per epilogue family."""
import torch, sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniception_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
M, N, K = 1024, 1024, 1024
a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().to(dev); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(dev)
b = torch.randn(N, generator=g).to(dev); res = torch.randn(M, N, generator=g).to(dev)
pos = torch.cartesian_prod(torch.arange(32), torch.arange(32)).to(dev).contiguous()
table = ops.rope_table(dev, 1024, 100.0, 1.0)
def run(var, fn):
    ops.tuning_set("gemm_variant", int(var))
    try: return fn()
    finally: ops.tuning_set("gemm_variant", -3)
cases = {
 "f32 plain": lambda: ops.gemm(a, w, out_dtype=torch.float32),
 "f32 bias+res": lambda: ops.gemm(a, w, b, residual=res, out_dtype=torch.float32),
 "bf16 plain": lambda: ops.gemm(a, w, b).float(),
 "bf16 gelu": lambda: ops.gemm(a, w, b, act="gelu").float(),
 "bf16 rope": lambda: ops.gemm(a, w, b, rope=(pos, table, N)).float(),
 "f32 emit_ln twin": lambda: ops.gemm(a, w, b, residual=res, out_dtype=torch.float32, emit_ln=True).uc_ln.twin.float(),
 "f32 emit_ln partial": lambda: ops.gemm(a, w, b, residual=res, out_dtype=torch.float32, emit_ln=True).uc_ln.partial,
}
for name, fn in cases.items():
    y2, y6 = run("2", fn), run("6", fn)
    d = (y2 - y6).abs()
    print(f"{name:22s} max abs diff {float(d.max()):.3e}  n_diff {int((d > 0).sum())} / {d.numel()}")
