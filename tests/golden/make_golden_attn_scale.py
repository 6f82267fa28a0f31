"""Golden vectors of the reference's Attention / CrossAttention layers with their optional q scalings — `use_scalable_softmax`
(q * log N) and `use_entropy_scaling` (q * sqrt(growth * log N / log base)), utils/transformer_blocks.py:231-241, 360-370 — and of
Attention with `latent_attn_dim` (q / k / v in a latent width, :178-199) from the
REAL reference (build container only; stubs as in make_golden.py):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/tmp/oracle_stubs:/root/reference:/root/repo python3 -B /root/repo/tests/golden/make_golden_attn_scale.py

Writes tests/golden/attn_scale_opts.npz: per case the layer's state_dict (fp32) and the reference output (inputs are re-made from the
case's seed by tests/golden/attn_opts_cases.make_inputs).  Data only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.attn_opts_cases import SCALE_CASES, make_inputs  # noqa: E402

from uniception.models.libs.croco.pos_embed import RoPE2D  # noqa: E402
from uniception.models.utils.transformer_blocks import Attention, CrossAttention  # noqa: E402


def main():
    store = {}
    for name, c in SCALE_CASES.items():
        torch.manual_seed(c["seed"])
        rope = RoPE2D(freq=100.0) if c["rope"] else None
        kw = dict(dim=c["dim"], num_heads=c["heads"], qkv_bias=True, qk_norm=c["qk_norm"], custom_positional_encoding=rope,
                  use_scalable_softmax=c["scalable"], use_entropy_scaling=c["entropy"])
        if c.get("latent"):
            kw["latent_attn_dim"] = c["latent"]
        layer = (Attention(**kw) if c["kind"] == "self" else CrossAttention(**kw)).eval()
        with torch.no_grad():
            for k, p in layer.named_parameters():
                if "norm" in k:
                    p.copy_(torch.randn_like(p) * 0.3 + (1.0 if k.endswith("weight") else 0.0))
        ins = make_inputs(c)
        with torch.no_grad():
            if c["kind"] == "self":
                out = layer(ins["x"], ins["xpos"] if c["rope"] else None)
            else:
                out = layer(ins["q"], ins["k"], ins["k"], ins["qpos"] if c["rope"] else None, ins["kpos"] if c["rope"] else None)
        for k, v in layer.state_dict().items():
            store[f"{name}/sd/{k}"] = v.numpy()
        store[f"{name}/out"] = out.numpy()
        print(name, tuple(out.shape), float(out.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "attn_scale_opts.npz"), **store)
    print("wrote attn_scale_opts.npz")


if __name__ == "__main__":
    main()
