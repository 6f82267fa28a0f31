"""Full-size sanity soak: ViT-L + DPT @512, fixed synthetic batch, N AdamW steps; the loss must fall and stay finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uniception_amd import autograd, engine
from uniception_amd.models.factory import DUSt3R
from uniception_amd.training import Trainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
pairs, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 30
model = DUSt3R(name="soak", img_size=(512, 512), pred_head_type="dpt").to(dev).train()
# SOAK_DROP=p (round 6): every dropout of the transformer blocks at rate p — proj_drop / Mlp drops (uc_mask_scale) and attn_drop (the
# attention kernels' counter-based mask, forward and backward) — at the full model's shapes (1024 tokens, 16 / 12 heads)
drop = float(os.environ.get("SOAK_DROP", "0"))
if drop > 0:
    n = 0
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = drop
            n += 1
        if hasattr(m, "dropout_p"):
            m.dropout_p = drop
    print(f"dropout {drop} on {n} Dropout modules", flush=True)
tr = Trainer(model, lr=3e-5, weight_decay=0.05)
g = torch.Generator().manual_seed(1)
v1 = {"img": torch.randn(pairs, 3, 512, 512, generator=g).to(dev), "instance": [str(i) for i in range(pairs)], "data_norm_type": "dust3r"}
v2 = {"img": torch.randn(pairs, 3, 512, 512, generator=g).to(dev), "instance": [str(100 + i) for i in range(pairs)], "data_norm_type": "dust3r"}
gt1 = torch.randn(pairs, 512, 512, 3, generator=g).to(dev)
gt2 = torch.randn(pairs, 512, 512, 3, generator=g).to(dev)
losses = []
for it in range(steps):
    tr.zero_grad()
    with engine.precision("bf16"):
        r1, r2 = model(v1, v2)
        loss = autograd.conf_loss(r1["pts3d"], r1["conf"], gt1) + autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gt2)
    loss.backward()
    tr.step()
    losses.append(float(loss.detach()))
    if it % 5 == 0 or it == steps - 1:
        print(f"step {it:3d} loss {losses[-1]:.5f} grad-norm {float(tr.flat.grad.norm()):.4e}", flush=True)
assert all(l == l and abs(l) < 1e6 for l in losses), "non-finite loss"
assert losses[-1] < losses[0], (losses[0], losses[-1])
print(f"OK: loss {losses[0]:.4f} -> {losses[-1]:.4f} over {steps} steps, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
