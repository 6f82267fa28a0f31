import sys, os
sys.path.insert(0, "/root/repo")
import torch
from uniception_amd import engine
from uniception_amd.graphs import GraphedTwoView
from uniception_amd.models.factory import DUSt3R
dev = torch.device("cuda:0")
model = DUSt3R(name="b", img_size=(512, 512), pred_head_type="dpt").to(dev).eval()
for P in [int(a) for a in sys.argv[1:]]:
    v1 = {"img": torch.randn(P, 3, 512, 512, device=dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
    v2 = {"img": torch.randn(P, 3, 512, 512, device=dev), "instance": [str(i) for i in range(P)], "data_norm_type": "dust3r"}
    try:
        g = GraphedTwoView(model, v1, v2)
        g(v1, v2); torch.cuda.synchronize()
        print(P, "ok", flush=True)
        del g
    except Exception as e:
        print(P, "FAILED", str(e)[:200], flush=True)
        break
