"""The global / alternating multi-view transformers at a size nobody had run them at: 4 views of 1024 tokens (global attention over 4096
tokens), ViT-L-encoder-sized inputs, 12 blocks of width 768; forward (bf16), training step (with and without gradient checkpointing,
with dropout of every kind).  Finite outputs, finite gradients, times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniception_amd import engine
from uniception_amd.models.info_sharing import INFO_SHARING_CLASSES, MultiViewTransformerInput
from uniception_amd.models.libs.croco.pos_embed import RoPE2D
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, V, h, w = 8, 4, 32, 32
for key in ("global_attention", "alternating_attention"):
    for extra in (dict(), dict(gradient_checkpointing=True), dict(proj_drop=0.1, attn_drop=0.1, drop_path=0.1, init_values=0.5)):
        cls, _ = INFO_SHARING_CLASSES[key]
        kw = dict(name="t", input_embed_dim=1024, dim=768, num_heads=12, depth=12, custom_positional_encoding=RoPE2D(100.0) if key != "global_attention" else "rope")
        if key == "global_attention":
            kw["use_rand_idx_pe_for_non_reference_views"] = False
        m = cls(**kw, **extra).to(dev)
        feats = [torch.randn(B, 1024, h, w, device=dev) for _ in range(V)]
        m.eval()
        with torch.no_grad(), engine.precision("bf16"):
            out = m(MultiViewTransformerInput(features=feats))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m(MultiViewTransformerInput(features=feats))
            torch.cuda.synchronize(); tf = time.perf_counter() - t0
        assert all(torch.isfinite(f).all() for f in out.features)
        m.train()
        for it in range(2):
            for p in m.parameters():
                p.grad = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with engine.precision("bf16"):
                out = m(MultiViewTransformerInput(features=feats))
                loss = sum(f.float().square().mean() for f in out.features)
            loss.backward()
            torch.cuda.synchronize(); tt = time.perf_counter() - t0
        gn = torch.sqrt(sum(p.grad.float().square().sum() for p in m.parameters() if p.grad is not None))
        assert torch.isfinite(loss) and torch.isfinite(gn)
        print(f"{key:22s} {str(extra):90s} fwd {tf*1e3:7.1f} ms  train step {tt*1e3:7.1f} ms  loss {float(loss):.4f} |g| {float(gn):.3e}  peak {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
        del m, out, loss
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
