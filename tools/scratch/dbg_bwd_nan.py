import sys; sys.path.insert(0, "/root/repo")
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
for (B,H,Nq,Nk) in [(2,3,196,196),(1,2,300,1000),(2,2,1000,300),(1,1,77,130)]:
    g = torch.Generator().manual_seed(Nq*7+Nk)
    q = torch.randn(B,Nq,H,64,generator=g).bfloat16().to(dev); k = torch.randn(B,Nk,H,64,generator=g).bfloat16().to(dev)
    v = torch.randn(B,Nk,H,64,generator=g).bfloat16().to(dev); do = torch.randn(B,Nq,H,64,generator=g).bfloat16().to(dev)
    lse = torch.empty(B,H,Nq,device=dev)
    o = ops.attention(q,k,ops.vt_pack(v),0.125,v_packed=True,lse=lse)
    for mode in (0,1,2):
        with ops.tuning("attn_bwd64", mode):
            dq,dk,dv = ops.attention_bwd(q,k,v,o,do,lse,0.125)
        torch.cuda.synchronize()
        print((B,H,Nq,Nk), mode, [int(torch.isnan(t.float()).sum()) for t in (dq,dk,dv)], flush=True)
