"""Case of the original-layout DPT golden (make_golden_dpt_original.py / test_convert_checkpoint_gpu.py)."""
DPT_ORIGINAL = dict(patch=16, img=(64, 96), B=2, token_dims=(128, 192, 192, 192), layer_dims=(16, 32, 64, 128), feature_dim=32, seed=91)


def dpt_original_tokens():
    import torch
    c = DPT_ORIGINAL
    g = torch.Generator().manual_seed(c["seed"])
    n = (c["img"][0] // c["patch"]) * (c["img"][1] // c["patch"])
    return [torch.randn(c["B"], n, d, generator=g) for d in c["token_dims"]]
