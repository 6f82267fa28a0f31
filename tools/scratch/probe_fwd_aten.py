"""Which Python lines launch torch's own (at::native / copy) kernels in a steady-state bf16 forward of the bench model (TorchDispatchMode + stack)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from uniception_amd import engine
from uniception_amd.models.factory import DUSt3R
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = DUSt3R(name="bench", img_size=(512, 512), pred_head_type="dpt").to(dev).eval()
g = torch.Generator().manual_seed(1)
v1 = {"img": torch.randn(P, 3, 512, 512, generator=g).to(dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(P, 3, 512, 512, generator=g).to(dev), "instance": [str(100 + i) for i in range(P)], "data_norm_type": "dust3r"}


def step():
    with torch.no_grad(), engine.precision("bf16"):
        return model(v1, v2)


for _ in range(3):
    step()
torch.cuda.synchronize()
SKIP = ("empty", "view", "as_strided", "select", "slice", "reshape", "permute", "transpose", "t.", "detach", "alias", "unsqueeze", "squeeze",
        "expand", "_unsafe_view", "split", "narrow", "unbind", "chunk", "unflatten", "_local_scalar", "lift_fresh", "set_", "is_nonzero", "stride", "sym_")
agg = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            fr = [f for f in traceback.extract_stack() if "uniception_amd/" in f.filename]
            where = " <- ".join(f"{f.filename[f.filename.index('uniception_amd/') + 15:]}:{f.lineno}" for f in reversed(fr[-3:])) if fr else "(no repo frame)"
            numel = next((a_.numel() for a_ in args if isinstance(a_, torch.Tensor)), 0)
            v = agg.setdefault((name, where), [0, 0]); v[0] += 1; v[1] += numel
        return func(*args, **(kwargs or {}))


with Spy():
    step()
    torch.cuda.synchronize()
for (name, where), (n, numel) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{n:5d} {numel/1e6:9.1f}M  {name:32s} {where[:150]}")
