"""Golden GRADIENTS of the reference's transformer blocks with qk_norm=True (LayerNorm of q / k over head_dim before the positional
encoding, utils/transformer_blocks.py:196-197, 229, 308-309, 348) — autograd over the REAL reference blocks on CPU (build container
only; stubs as in make_golden.py):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_qknorm_grads.py

Writes tests/golden/qknorm_blocks_grads.npz: per case the block's state_dict (fp32), the reference output, and the gradients of
loss = sum(out * w) with respect to every parameter (evenly spaced samples + L2 norm for the large ones) and input.  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.qknorm_block_cases import CASES, make_inputs, sample_idx  # noqa: E402

from uniception.models.libs.croco.pos_embed import RoPE2D  # noqa: E402
from uniception.models.utils.transformer_blocks import CrossAttentionBlock, SelfAttentionBlock  # noqa: E402


def main():
    store = {}
    for name, c in CASES.items():
        torch.manual_seed(c["seed"])
        rope = RoPE2D(freq=100.0) if c["rope"] else None
        kw = dict(dim=c["dim"], num_heads=c["heads"], qkv_bias=True, qk_norm=True, custom_positional_encoding=rope)
        blk = (SelfAttentionBlock(**kw) if c["kind"] == "self" else CrossAttentionBlock(**kw)).train()
        with torch.no_grad():
            for k, p in blk.named_parameters():      # non-trivial norm parameters
                if "norm" in k:
                    p.copy_(torch.randn_like(p) * 0.3 + (1.0 if k.endswith("weight") else 0.0))
        ins = make_inputs(c)
        x = ins["x"].clone().requires_grad_(True)
        if c["kind"] == "self":
            out = blk(x, ins["xpos"] if c["rope"] else None)
        else:
            y = ins["y"].clone().requires_grad_(True)
            out = blk(x, y, ins["xpos"] if c["rope"] else None, ins["ypos"] if c["rope"] else None)
        (out * ins["w"]).sum().backward()
        for k, v in blk.state_dict().items():
            store[f"{name}/sd/{k}"] = v.detach().numpy()
        store[f"{name}/out"] = out.detach().numpy()
        store[f"{name}/dx"] = x.grad.numpy()
        if c["kind"] == "cross":
            store[f"{name}/dy"] = y.grad.numpy()
        for k, p in blk.named_parameters():      # (large gradients: 4096 evenly spaced entries + the L2 norm)
            store[f"{name}/grad/{k}"] = p.grad.flatten()[sample_idx(p.numel())].numpy()
            store[f"{name}/gnorm/{k}"] = np.float64(float(p.grad.norm()))
        print(name, tuple(out.shape), float(out.abs().mean()), len(list(blk.named_parameters())), "parameters")
    np.savez_compressed(os.path.join(HERE, "qknorm_blocks_grads.npz"), **store)
    print("wrote qknorm_blocks_grads.npz")


if __name__ == "__main__":
    main()
