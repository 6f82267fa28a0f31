"""Anatomy of the bf16-stream residual epilogue (diag build: UNICEPTION_AMD_DIAG_LIB=1): enc / dec proj with UC_GEMM_DBG bits
128 (no residual read), 512 (no row statistics), 1024 (statistics computed, not stored), and the start stagger on / off.
The env var is read once per process: run once per bit set.  usage: UC_GEMM_DBG=<bits> python tools/probe_bs_anatomy.py [pairs]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64

def timeit(fn, seconds=0.3):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    iters = max(5, int(seconds / (e0.elapsed_time(e1) * 1e-3)))
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-6 * 1e3

def rnd(*s, scale=0.5): return (torch.randn(*s, device=dev) * scale).bfloat16()
cells = []
for tag, M, C in (("enc", 2 * B * 1024, 1024), ("dec", B * 1024, 768)):
    a, w, bias, xs = rnd(M, C), rnd(C, C, scale=1 / math.sqrt(C)), torch.randn(C, device=dev) * 0.1, rnd(M, C)
    out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    for stg in (-1, 0):
        ops.tuning_set("gemm_stagger", stg)
        t = timeit(lambda: ops.gemm(a, w, bias, residual=xs, out=out, emit_ln=True))
        t2 = timeit(lambda: ops.gemm(a, w, bias, residual=xs, out=out))
        t3 = timeit(lambda: ops.gemm(a, w, bias))
        cells.append(f"{tag} stagger={stg:3d}: res+stats {t*1e6:6.1f}us  res only {t2*1e6:6.1f}us  plain {t3*1e6:6.1f}us")
print(f"UC_GEMM_DBG={os.environ.get('UC_GEMM_DBG','0')}\n  " + "\n  ".join(cells), flush=True)
