import sys, os
sys.path.insert(0, "/root/repo")
import torch
from uniception_amd import engine, ops
from uniception_amd.models.factory import DUSt3R
dev = torch.device("cuda:0")
model = DUSt3R(name="b", img_size=(512, 512), pred_head_type=sys.argv[3] if len(sys.argv) > 3 else "dpt").to(dev).eval()
P = int(sys.argv[1]); mode = sys.argv[2]
v1 = {"img": torch.randn(P, 3, 512, 512, device=dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(P, 3, 512, 512, device=dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
def fwd():
    with torch.no_grad(), engine.precision("bf16"):
        return model(v1, v2)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd(); fwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
if len(sys.argv) > 4 and sys.argv[4] == "single":
    ctx = engine.concurrent(False)
else:
    import contextlib; ctx = contextlib.nullcontext()
ops.fuse_ws_reserve(4)
g = torch.cuda.CUDAGraph()
try:
    with ctx, ops.capture_scope(12345), torch.cuda.graph(g, capture_error_mode=mode):
        out = fwd()
    g.replay(); torch.cuda.synchronize()
    print(P, mode, sys.argv[3:], "ok", torch.cuda.max_memory_allocated() / 1e9)
except Exception as e:
    print(P, mode, sys.argv[3:], "FAILED", str(e)[:120])
