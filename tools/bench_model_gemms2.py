"""The forward's dense GEMM calls as the bf16 model issues them TODAY (bf16 residual stream + row statistics, folded LayerNorm in
the consumers, RoPE + VT, polynomial GELU), per tile variant, sustained.
usage: python tools/bench_model_gemms2.py [pairs=64] [variants=auto,2,3,6,7] [which=enc,dec] [seconds=0.5]
Prints one line per shape (us, TFLOP/s per variant) and a JSON table at the end (the per-shape table VERDICT r3 next #1 asks for)."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniception_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = (sys.argv[2] if len(sys.argv) > 2 else "auto,2,3,6,7").split(",")
which = (sys.argv[3] if len(sys.argv) > 3 else "enc,dec").split(",")
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
N = 1024


def timeit(fn, seconds):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    iters = max(5, int(seconds / (e0.elapsed_time(e1) * 1e-3)))
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*s, scale=0.5): return (torch.randn(*s, device=dev) * scale).bfloat16()


def cases(tag, Bimg, C, H):
    M = Bimg * N
    h = rnd(M, C); xs = rnd(M, C); hid = rnd(M, 4 * C)
    pos = torch.cartesian_prod(torch.arange(32), torch.arange(32)).repeat(Bimg, 1).to(dev).contiguous()
    table = ops.rope_table(dev, 1024, 100.0, 1.0)
    w = lambda n, k: rnd(n, k, scale=1 / math.sqrt(k))
    b = lambda n: torch.randn(n, device=dev) * 0.1
    wqkv, bqkv, wp, bp, w1, b1, w2, b2 = w(3 * C, C), b(3 * C), w(C, C), b(C), w(4 * C, C), b(4 * C), w(C, 4 * C), b(C)
    vt = ops.vt_buffer(Bimg, H, N, dev)
    stats = torch.stack([torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5], 1).contiguous()
    cs = lambda n: torch.randn(n, device=dev)
    cs3, cs4, cs1, cs2 = cs(3 * C), cs(4 * C), cs(C), cs(2 * C)
    outb = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    yield f"{tag} qkv ln+rope+vt", (M, 3 * C, C), lambda: ops.gemm(h, wqkv, bqkv, rope=(pos, table, 2 * C), vt=(2 * C, vt, N), ln=(stats, cs3))
    yield f"{tag} qkv plain     ", (M, 3 * C, C), lambda: ops.gemm(h, wqkv, bqkv)
    yield f"{tag} proj +res+stat", (M, C, C), lambda: ops.gemm(h, wp, bp, residual=xs, out=outb, emit_ln=True)
    yield f"{tag} proj plain    ", (M, C, C), lambda: ops.gemm(h, wp, bp)
    yield f"{tag} fc1 ln+gelu   ", (M, 4 * C, C), lambda: ops.gemm(h, w1, b1, act="gelu", ln=(stats, cs4))
    yield f"{tag} fc1 plain     ", (M, 4 * C, C), lambda: ops.gemm(h, w1, b1)
    yield f"{tag} fc2 +res+stat ", (M, C, 4 * C), lambda: ops.gemm(hid, w2, b2, residual=xs, out=outb, emit_ln=True)
    yield f"{tag} fc2 plain     ", (M, C, 4 * C), lambda: ops.gemm(hid, w2, b2)
    if tag == "dec":
        wkv, bkv = w(2 * C, C), b(2 * C)
        yield f"{tag} projq ln+rope ", (M, C, C), lambda: ops.gemm(h, wp, bp, rope=(pos, table, C), ln=(stats, cs1))
        yield f"{tag} kv ln+rope+vt ", (M, 2 * C, C), lambda: ops.gemm(h, wkv, bkv, rope=(pos, table, C), vt=(C, vt, N), ln=(stats, cs2))


rows = []
for tag, Bimg, C, H in (("enc", 2 * B, 1024, 16), ("dec", B, 768, 12)):
    if tag not in which: continue
    for name, (M, Nn, K), fn in cases(tag, Bimg, C, H):
        fl = 2.0 * M * Nn * K
        rec = {"shape": name.strip(), "M": M, "N": Nn, "K": K}
        for v in variants:
            ops.tuning_set("gemm_variant", -3 if v == "auto" else int(v))
            try:
                t = timeit(fn, secs)
                rec[f"v{v}"] = {"us": round(t * 1e6, 1), "tflops": round(fl / t / 1e12, 1), "frac": round(fl / t / 2.5e15, 3)}
            except Exception as e:  # a variant that does not take this epilogue
                rec[f"v{v}"] = {"error": str(e)[:80]}
        rows.append(rec)
        print(f"{name}: " + " | ".join(
            (f"v{v} {rec[f'v{v}']['us']:7.1f}us {rec[f'v{v}']['tflops']:6.1f}TF" if "us" in rec[f"v{v}"] else f"v{v} n/a") for v in variants), flush=True)
ops.tuning_set("gemm_variant", -3)
print("JSON " + json.dumps({"pairs": B, "seconds_per_point": secs, "rows": rows}))
