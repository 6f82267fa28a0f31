import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniception_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, Nq, Nk) in [(2, 3, 4, 4), (2, 3, 16, 16), (2, 3, 9, 18), (1, 2, 196, 196), (2, 3, 64, 64), (2, 12, 4, 8), (1, 1, 1, 1), (2, 2, 65, 65)]:
    # K as a view into a fused buffer [B,N,3,H,64] placed so that memory after the last key row is NaN
    buf = torch.full((B, Nk, 3, H, 64), float("nan"), device=dev).bfloat16()
    buf[:, :, 1] = torch.randn(B, Nk, H, 64, device=dev).bfloat16()
    k = buf[:, :, 1]
    q = torch.randn(B, Nq, H, 64, device=dev).bfloat16()
    v = torch.randn(B, Nk, H, 64, device=dev).bfloat16()
    vt = ops.vt_pack(v)
    o = ops.attention(q, k, vt, 0.125, v_packed=True)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 0.125
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.float())
    err = float((o.float() - ref).norm() / ref.norm())
    print(B, H, Nq, Nk, "finite", bool(torch.isfinite(o).all()), "err", err)
