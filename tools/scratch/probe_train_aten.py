"""Which Python lines launch the at::native kernels of a training step (torch.profiler, CPU ops grouped by stack)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from uniception_amd import autograd, engine
from uniception_amd.models.factory import DUSt3R
from uniception_amd.training import Trainer
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = DUSt3R(name="bench", img_size=(512, 512), pred_head_type="dpt").to(dev).train()
g = torch.Generator().manual_seed(1)
v1 = {"img": torch.randn(P, 3, 512, 512, generator=g).to(dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(P, 3, 512, 512, generator=g).to(dev), "instance": [str(100 + i) for i in range(P)], "data_norm_type": "dust3r"}
gt1 = torch.randn(P, 512, 512, 3, generator=g).to(dev); gt2 = torch.randn(P, 512, 512, 3, generator=g).to(dev)
trainer = Trainer(model, lr=1e-5, weight_decay=0.05)


def step():
    trainer.zero_grad()
    with engine.precision("bf16"):
        r1, r2 = model(v1, v2)
        loss = autograd.conf_loss(r1["pts3d"], r1["conf"], gt1) + autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gt2)
    loss.backward()
    trainer.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
SKIP = ("empty", "view", "as_strided", "select", "slice", "reshape", "permute", "transpose", "t.", "detach", "alias", "unsqueeze", "squeeze",
        "expand", "_unsafe_view", "split", "narrow", "unbind", "chunk", "unflatten", "_local_scalar", "lift_fresh", "set_", "is_nonzero", "stride", "sym_")
agg = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            fr = [f for f in traceback.extract_stack() if "uniception_amd/" in f.filename]
            where = " <- ".join(f"{f.filename[f.filename.index('uniception_amd/') + 15:]}:{f.lineno}" for f in reversed(fr[-3:])) if fr else "(autograd engine: no repo frame)"
            numel = 0
            for a_ in args:
                if isinstance(a_, torch.Tensor):
                    numel = a_.numel(); break
            k = (name, where)
            v = agg.setdefault(k, [0, 0]); v[0] += 1; v[1] += numel
        return func(*args, **(kwargs or {}))


with Spy():
    step()
    torch.cuda.synchronize()
for (name, where), (n, numel) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{n:5d} {numel/1e6:9.1f}M  {name:32s} {where[:140]}")
