"""DINOv2 cases pinned to an independent implementation (HuggingFace transformers' Dinov2Model / Dinov2WithRegistersModel) — shared
by the golden generator (make_golden_dinov2_hf.py), the oracle test and the GPU test.  Native 37x37 grids only (518 / 14): the two
implementations resize the position embedding differently for other grids (hub: 0.1-offset scale factor / antialias; HF: plain
size-based bicubic), so only the native grid is a common ground."""
import torch

SIZES = {"small": (384, 6), "base": (768, 12), "large": (1024, 16), "giant": (1536, 24)}   # embed dim, heads
SWIGLU_HIDDEN = {"giant": 4096}      # giant: SwiGLU FFN, hidden (int(4 D * 2 / 3) + 7) // 8 * 8 (hub SwiGLUFFNFused = transformers Dinov2SwiGLUFFN)
GAINS = {"pos_embed": 300.0, "cls_token": 10.0, "register_tokens": 20.0}
DINOV2_HF_CASES = {
    "small_noreg": dict(size="small", regs=False, layers=2, hw=(518, 518), B=1, seed=11),
    "small_reg": dict(size="small", regs=True, layers=3, hw=(518, 518), B=1, seed=12),
    "base_reg": dict(size="base", regs=True, layers=1, hw=(518, 518), B=1, seed=13),
    # the size BASELINE configs[3] names: ViT-L/14, all 24 blocks, 518 x 518 (the reference's defaults: size="large", with_registers=False,
    # encoders/dinov2.py:18-24)
    "large_full": dict(size="large", regs=False, layers=24, hw=(518, 518), B=1, seed=14),
    # other grids than the checkpoint's 37 x 37: the position table goes through the resize of the *_reg hub models (bicubic, antialias,
    # explicit size), which transformers' Dinov2WithRegistersEmbeddings.interpolate_pos_encoding implements independently — down, up,
    # non-square.  (The non-register models' resize — a 0.1 offset folded into a scale factor, no antialias — has no second
    # implementation here: transformers' Dinov2Model resizes to an explicit size instead.)
    "small_reg_224": dict(size="small", regs=True, layers=2, hw=(224, 224), B=2, seed=15),
    "base_reg_448x336": dict(size="base", regs=True, layers=1, hw=(448, 336), B=1, seed=16),
    "small_reg_700x560": dict(size="small", regs=True, layers=1, hw=(700, 560), B=1, seed=17),
    # giant (ViT-g/14: 1536 wide, 24 heads, SwiGLU FFN of 4096 hidden), the first two blocks, with and without registers
    "giant_reg_224": dict(size="giant", regs=True, layers=2, hw=(224, 224), B=1, seed=18),
    "giant_noreg": dict(size="giant", regs=False, layers=1, hw=(518, 518), B=1, seed=19),
}


def dinov2_image(c):
    g = torch.Generator().manual_seed(c["seed"])
    return torch.randn(c["B"], 3, *c["hw"], generator=g)


def dinov2_hub_state_dict(c, prefix="model."):
    "Hub-named parameters (the names DINOv2Encoder / the oracle read) filled by the name-keyed filler."
    from oracle import dust3r_oracle as O
    D, _ = SIZES[c["size"]]
    R = 4 if c["regs"] else 0
    sd = {"cls_token": torch.empty(1, 1, D), "pos_embed": torch.empty(1, 1 + 37 * 37, D),
          "patch_embed.proj.weight": torch.empty(D, 3, 14, 14), "patch_embed.proj.bias": torch.empty(D),
          "norm.weight": torch.empty(D), "norm.bias": torch.empty(D)}
    if R:
        sd["register_tokens"] = torch.empty(1, R, D)
    for i in range(c["layers"]):
        b = f"blocks.{i}."
        for n, shp in (("norm1.weight", (D,)), ("norm1.bias", (D,)), ("attn.qkv.weight", (3 * D, D)), ("attn.qkv.bias", (3 * D,)),
                       ("attn.proj.weight", (D, D)), ("attn.proj.bias", (D,)), ("ls1.gamma", (D,)), ("norm2.weight", (D,)),
                       ("norm2.bias", (D,)), ("ls2.gamma", (D,))):
            sd[b + n] = torch.empty(*shp)
        Hs = SWIGLU_HIDDEN.get(c["size"])
        ffn = ((("mlp.fc1.weight", (4 * D, D)), ("mlp.fc1.bias", (4 * D,)), ("mlp.fc2.weight", (D, 4 * D)), ("mlp.fc2.bias", (D,))) if Hs is None else
               (("mlp.w12.weight", (2 * Hs, D)), ("mlp.w12.bias", (2 * Hs,)), ("mlp.w3.weight", (D, Hs)), ("mlp.w3.bias", (D,))))
        for n, shp in ffn:
            sd[b + n] = torch.empty(*shp)
    sd = {prefix + k: v for k, v in sd.items()}
    O.fill_state_dict_(sd, gains=GAINS)
    return sd


# cases that also carry gradients (transformers' autograd): loss = <features, Wf> + <cls/registers, Wr>, seeded cotangents
DINOV2_HF_GRAD_CASES = ("small_noreg", "small_reg", "giant_reg_224")


def dinov2_grad_weights(name, f_shape, r_shape):
    g = torch.Generator().manual_seed(sum(map(ord, name)) + 4099)
    return torch.randn(*f_shape, generator=g), torch.randn(*r_shape, generator=g)
