"""The DUSt3R forward under any engine precision mode / head policy, for profiling:  python tools/bench_mode.py bf16x3 [pairs=16] [steps=3] [head_precision=follow]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import engine
from uniception_amd.models.factory import DUSt3R
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
headp = sys.argv[4] if len(sys.argv) > 4 else "follow"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DUSt3R(name="b", img_size=(512, 512), pred_head_type="dpt").to(dev).eval()
g = torch.Generator().manual_seed(1)
v1 = {"img": torch.randn(pairs, 3, 512, 512, generator=g).to(dev), "instance": [f"a{i}" for i in range(pairs)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(pairs, 3, 512, 512, generator=g).to(dev), "instance": [f"b{i}" for i in range(pairs)], "data_norm_type": "dust3r"}
engine.set_head_precision(headp)
if os.environ.get("SINGLE_STREAM") == "1":
    engine.CONCURRENT = False
def f():
    with torch.no_grad(), engine.precision(mode):
        return model(v1, v2)
f(); f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"mode {mode} heads {headp} pairs {pairs}: {dt*1e3:.1f} ms/step, {pairs/dt:.1f} pairs/s")
