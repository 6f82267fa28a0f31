"""Why the encoder + decoder leg of bench.py sometimes reads 345 instead of 470 pairs/s: per-step times of the linear-head model built
after the DPT model has run, with and without an idle gap before it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import engine
from uniception_amd.models.factory import DUSt3R
dev = torch.device("cuda:0")
P = 64
g = torch.Generator().manual_seed(1)
v1 = {"img": torch.randn(P, 3, 512, 512, generator=g).to(dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(P, 3, 512, 512, generator=g).to(dev), "instance": [str(100 + i) for i in range(P)], "data_norm_type": "dust3r"}


def steps(m, n, tag):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.time()
        with torch.no_grad(), engine.precision("bf16"):
            m(v1, v2)
        torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    print(tag, " ".join(f"{t:6.1f}" for t in ts), flush=True)


dpt = DUSt3R(name="b", img_size=(512, 512), pred_head_type="dpt").to(dev).eval()
steps(dpt, 6, "dpt      ")
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
time.sleep(gap)
lin = DUSt3R(name="l", img_size=(512, 512), pred_head_type="linear").to(dev).eval()
steps(lin, 10, f"lin gap{gap:3.0f}")
del lin
torch.cuda.empty_cache()
lin = DUSt3R(name="l", img_size=(512, 512), pred_head_type="linear").to(dev).eval()
steps(lin, 6, "lin again ")
print("mem", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
