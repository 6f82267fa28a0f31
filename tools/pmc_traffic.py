"""HBM traffic per launch of the dominant kernel (dense bf16 GEMM, all tile variants / epilogue families) from the FETCH_SIZE
and WRITE_SIZE tables tools/profile_round.sh wrote, with the gfx950 correction, stamped with the kernel fingerprint so that
bench.py only reports it for the build it was measured on.  usage: python tools/pmc_traffic.py <dir> <tag>"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uniception_amd import build
from bench import FWD_PAIRS as P          # the headline's pairs per GPU: the tables are named after it (tools/profile_round.sh)
d, tag = sys.argv[1], sys.argv[2]


def table(counter):
    tot, n = 0.0, 0
    cur = None
    for line in open(os.path.join(d, f"{tag}_pmc_{counter}_bench_pairs{P}.txt")):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        m = re.search(r"total=\s*([0-9.]+)\s+per-dispatch=\s*([0-9.]+)\s+\(n=(\d+)\)", line)
        # dense kernels: A_MODE (6th template argument) == 0
        dense = re.match(r"gemm_bf16_glds_kernel<\d+, \d+, \d+, \d+, \d+, 0,", cur) or cur.startswith(("gemm_bf16_glds8_kernel<", "gemm_bf16_glds4_kernel<"))   # the eight- and four-wave kernels are dense only
        if m and dense and counter in line:
            tot += float(m.group(1)); n += int(m.group(3))
    return tot, n


f, nf = table("FETCH_SIZE")
w, nw = table("WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
rec = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 ({P} pairs/GPU, 512x512, DPT); tables profiles/{tag}_pmc_*_bench_pairs{P}.txt",
       "kernel": "gemm_bf16_glds_kernel<*, A_MODE dense, *> + gemm_bf16_glds8_kernel<*> + gemm_bf16_glds4_kernel<*> (all dense tile variants and epilogue families)",
       "kernel_fingerprint": build.loaded_fingerprint(),
       "config": {"pairs_per_gpu": P, "img": 512, "head": "dpt", "precision": "bf16"},
       "dispatches": nf, "FETCH_SIZE_KB_per_launch": round(f / nf, 1), "WRITE_SIZE_KB_per_launch": round(w / nw, 1),
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
       "traffic_bytes_per_launch": int((2 * f + w) / nf * 1024)}
print(json.dumps(rec, indent=1))
