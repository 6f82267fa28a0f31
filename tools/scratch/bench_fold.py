"""Forward with / without the fused LayerNorm path, same process (A/B of engine.set_ln_fold)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import engine
from uniception_amd.models.factory import DUSt3R
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
head = sys.argv[2] if len(sys.argv) > 2 else "dpt"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DUSt3R(name="b", img_size=(512, 512), pred_head_type=head).to(dev).eval()
g = torch.Generator().manual_seed(1000)
v1 = {"img": torch.randn(pairs, 3, 512, 512, generator=g).to(dev), "instance": [f"a{i}" for i in range(pairs)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(pairs, 3, 512, 512, generator=g).to(dev), "instance": [f"b{i}" for i in range(pairs)], "data_norm_type": "dust3r"}
outs = {}
for rnd in range(2):
    for fold in (False, True):
        engine.set_ln_fold(fold)
        with torch.no_grad(), engine.precision("bf16"):
            for _ in range(2): r = model(v1, v2)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): r = model(v1, v2)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        outs[fold] = r[0]["pts3d"].float()
        print(f"round {rnd} fold={fold}: {dt*1e3:.1f} ms/step, {pairs/dt:.1f} pairs/s", flush=True)
d = (outs[True] - outs[False]).norm() / outs[False].norm()
print(f"pts3d fold vs no-fold rel-L2 {d:.3e}")
