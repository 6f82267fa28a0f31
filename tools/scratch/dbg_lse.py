import sys, torch
sys.path.insert(0, "/root/repo")
from uniception_amd import ops
gpu = torch.device("cuda:0")
B, H, D, Nq, Nk = 2, 2, 64, 512, 768
g = torch.Generator().manual_seed(77)
q = torch.randn(B, Nq, H, D, generator=g).bfloat16(); k = torch.randn(B, Nk, H, D, generator=g).bfloat16(); v = torch.randn(B, Nk, H, D, generator=g).bfloat16()
k[0, 700, 0] = q[0, 17, 0] * 16; k[1, 40, 1] = q[1, 500, 1] * 14
ref_lse = torch.logsumexp(torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * D ** -0.5, dim=-1)
for mode in (2, 0):
    with ops.tuning("attn_p64", mode):
        lse = torch.empty(B, H, Nq, device=gpu)
        out = ops.attention(q.to(gpu), k.to(gpu), ops.vt_pack(v.to(gpu)), D ** -0.5, v_packed=True, lse=lse)
    d = (lse.cpu() - ref_lse).abs()
    idx = torch.nonzero(d > 2e-2)
    print("mode", mode, "max", d.max().item(), "n bad", len(idx))
    for i in idx[:10]:
        b, h, qq = i.tolist(); print("   ", b, h, qq, lse[b, h, qq].item(), ref_lse[b, h, qq].item())
