"""Epilogue anatomy of the encoder QKV / proj GEMMs: each epilogue option alone, plus the per-workgroup timeline (UC_GEMM_TRACE)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, C, H = 1024, 1024, 16
Bimg = 2 * B
M = Bimg * N
def rnd(*s, scale=0.5): return (torch.randn(*s, device=dev) * scale).bfloat16()
h = rnd(M, C); x = torch.randn(M, C, device=dev)
pos = torch.cartesian_prod(torch.arange(32), torch.arange(32)).repeat(Bimg, 1).to(dev).contiguous()
table = ops.rope_table(dev, 1024, 100.0, 1.0)
wqkv = rnd(3 * C, C, scale=1 / 32); bqkv = torch.randn(3 * C, device=dev) * 0.1
wp = rnd(C, C, scale=1 / 32); bp = torch.randn(C, device=dev) * 0.1
vt = ops.vt_buffer(Bimg, H, N, dev)
out32 = torch.empty(M, C, device=dev)
outb = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
xb = x.bfloat16()
cases = [
    ("qkv plain", lambda: ops.gemm(h, wqkv, bqkv)),
    ("qkv rope(2C) no vt", lambda: ops.gemm(h, wqkv, bqkv, rope=(pos, table, 2 * C))),
    ("qkv vt only", lambda: ops.gemm(h, wqkv, bqkv, vt=(2 * C, vt, N))),
    ("qkv rope+vt", lambda: ops.gemm(h, wqkv, bqkv, rope=(pos, table, 2 * C), vt=(2 * C, vt, N))),
    ("proj plain bf16", lambda: ops.gemm(h, wp, bp)),
    ("proj f32 out no res", lambda: ops.gemm(h, wp, bp, out=out32)),
    ("proj +res32", lambda: ops.gemm(h, wp, bp, residual=x, out=out32)),
    ("proj +res bf16 -> bf16", lambda: ops.gemm(h, wp, bp, residual=xb, out=outb)),
]
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
if os.environ.get("UC_GEMM_TRACE"):
    for name, fn in cases:
        print(name, flush=True); sys.stderr.flush()
        fn(); torch.cuda.synchronize()
else:
    for name, fn in cases:
        print(f"{name:28s} {timeit(fn):8.1f} us", flush=True)

# ---- folded LayerNorm: producer (emit) and consumer (ln=) costs
import torch.nn.functional as F
xx = ops.gemm(h, wp, bp, residual=x, out_dtype=torch.float32, emit_ln=True)
side = xx.uc_ln
st = side.stats(1e-6)
cs = wqkv.float().sum(1).contiguous()
w1 = rnd(4 * C, C, scale=1 / 32); b1 = torch.randn(4 * C, device=dev) * 0.1; cs1 = w1.float().sum(1).contiguous()
gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
fold_cases = [
    ("proj +res32 emit twin+stats", lambda: ops.gemm(h, wp, bp, residual=x, out=out32, emit_ln=True)),
    ("ln finalize", lambda: ops.LnSide(side.twin, side.partial).stats(1e-6)),
    ("layernorm kernel f32->bf16", lambda: ops.layernorm(x, gam, bet, 1e-6, torch.bfloat16)),
    ("qkv rope+vt ln-fold", lambda: ops.gemm(side.twin, wqkv, bqkv, rope=(pos, table, 2 * C), vt=(2 * C, vt, N), ln=(st, cs))),
    ("qkv rope+vt", lambda: ops.gemm(h, wqkv, bqkv, rope=(pos, table, 2 * C), vt=(2 * C, vt, N))),
    ("fc1 gelu ln-fold", lambda: ops.gemm(side.twin, w1, b1, act="gelu", ln=(st, cs1))),
    ("fc1 gelu", lambda: ops.gemm(h, w1, b1, act="gelu")),
]
if os.environ.get("UC_GEMM_TRACE"):
    for name, fn in fold_cases:
        print(name, flush=True); sys.stderr.flush()
        fn(); torch.cuda.synchronize()
else:
    for rep in range(2):
        for name, fn in fold_cases:
            print(f"{name:28s} {timeit(fn):8.1f} us", flush=True)
