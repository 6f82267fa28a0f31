"""Golden-vector case table shared by the generator (make_golden.py) and the parity tests."""
import numpy as np

# gain overrides of the name-keyed filler: scale the last head layer so the decoded channels are O(1) and the
# adaptor's expm1 / exp run in their curved regime without overflowing (gain 1 gives |decoded| ~ 10).
GAINS = {"conv2.2.weight": 0.4, "head1.linear.weight": 0.5, "head2.linear.weight": 0.5}

_TINY = dict(patch=16, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=192, dec_depth=3, dec_heads=3,
             indices=(0, 1), layer_dims=(16, 32, 64, 128), feature_dim=32, store="full")

CASES = {
    # composed from the reference modules with small dims (head_dim 64 everywhere so the MFMA path applies)
    "tiny_dpt": dict(_TINY, head="dpt", img=(64, 96), B=2, seed=11),
    "tiny_linear": dict(_TINY, head="linear", img=(64, 96), B=2, seed=12),
    # odd 5x7 token grid: N=35 (attention/VT tails), DPT stride-2 level 3x4 -> x2 -> crop to 5x7
    "tiny_dpt_odd": dict(_TINY, head="dpt", img=(80, 112), B=1, seed=13),
    # patch 14 (BASELINE config 4's grid class): 5x7 tokens, 8x DPT map 40x56 -> non-integer bilinear resize to 70x98
    "tiny_dpt_p14": dict(_TINY, head="dpt", img=(70, 98), B=1, seed=14, patch=14),
    # BASELINE config 0: ViT-B/16 encoder + 6-block decoder + linear head, 224x224, batch 2
    "cfg1_vitb_linear_224": dict(patch=16, enc_dim=768, enc_depth=12, enc_heads=12, dec_dim=768, dec_depth=6, dec_heads=12,
                                 indices=(), head="linear", img=(224, 224), B=2, seed=21, store="samples"),
    # the reference factory model (ViT-L/16 + 12-block decoder), linear head @224 and DPT head @512 (BASELINE config 1/2)
    "vitl_linear_224": dict(factory=True, patch=16, enc_dim=1024, enc_depth=24, enc_heads=16, dec_dim=768, dec_depth=12,
                            dec_heads=12, indices=(5, 8), head="linear", img=(224, 224), B=1, seed=31, store="samples"),
    "vitl_dpt_512": dict(factory=True, patch=16, enc_dim=1024, enc_depth=24, enc_heads=16, dec_dim=768, dec_depth=12,
                         dec_heads=12, indices=(5, 8), head="dpt", img=(512, 512), B=1, seed=32, store="samples"),
}


def sample_indices(numel: int, n: int = 4096) -> np.ndarray:
    """Evenly spaced flat indices (all of them when the tensor is small)."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    return (np.arange(n, dtype=np.int64) * (numel - 1)) // (n - 1)


# cases that also have a gradient fixture (<case>__grads.npz, written by make_golden_grads.py)
GRAD_CASES = ("tiny_linear", "tiny_dpt", "cfg1_vitb_linear_224", "vitl_dpt_512")


def grad_targets(c):
    """Seeded synthetic pointmap targets [B,H,W,3] for the two views of case dict `c`."""
    import torch
    H, W = c["img"]
    out = []
    for v in (0, 1):
        rng = np.random.Generator(np.random.Philox(key=7000 + 10 * c["seed"] + v))
        out.append(torch.from_numpy(rng.standard_normal(size=(c["B"], H, W, 3), dtype=np.float32)))
    return out
