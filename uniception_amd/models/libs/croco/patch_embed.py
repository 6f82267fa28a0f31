"""Patch embedding (reference: libs/croco/patch_embed.py:13-127): the k=s=P conv is a patch gather + one GEMM."""
import torch
import torch.nn as nn

from .... import autograd, engine, ops
from ...utils.positional_encoding import PositionGetter  # noqa: F401  (same helper as the reference's local copy)
from .blocks import to_2tuple


def get_patch_embed(patch_embed_cls, img_size, patch_size, enc_embed_dim):
    assert patch_embed_cls in ["PatchEmbedCroCo", "PatchEmbedDust3R", "ManyAR_PatchEmbed"]
    return {"PatchEmbedCroCo": PatchEmbedCroCo, "PatchEmbedDust3R": PatchEmbedDust3R,
            "ManyAR_PatchEmbed": ManyAR_PatchEmbed}[patch_embed_cls](img_size, patch_size, 3, enc_embed_dim)


class PatchEmbedCroCo(nn.Module):
    """Conv2d(3->D, k=s=P) tokens [B,N,D] (row-major grid) + int64 (y,x) positions (patch_embed.py:34-65)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        self.position_getter = PositionGetter()

    def _embed(self, x):
        """[B,3,H,W] -> (tokens fp32 [B,N,D], pos)."""
        if self.patch_size[0] != self.patch_size[1]:
            raise engine.UcHipError("non-square patches are not supported by the HIP patch gather")
        if not self.flatten:
            raise engine.UcHipError("flatten=False is not supported by the HIP patch embedding")
        B, C, H, W = x.shape
        P = self.patch_size[0]
        dt = engine.compute_dtype()
        if (C * P * P) % 8 != 0:
            dt = torch.float32   # e.g. patch 14: K = 588 is not a multiple of the 16-byte bf16 chunk; this GEMM is 0.1 % of the FLOPs
        img = x.float().contiguous() if (x.dtype != torch.float32 or not x.is_contiguous()) else x
        train = autograd.grad_needed(x, self.proj.weight)
        if train:
            # (stream dtype: bf16 next to bf16 operands — engine.stream_dtype — unless a norm layer follows, which takes fp32)
            sdt = engine.stream_dtype(dt, self.proj.weight.shape[0]) if isinstance(self.norm, nn.Identity) else torch.float32
            tok = autograd.patch_embed(img, self.proj, P, dt, sdt).view(B, (H // P) * (W // P), -1)
        else:
            cols = ops.patch_gather(img, P, dt)
            w, b = engine.patch_weights(self.proj, dt)
            emit = isinstance(self.norm, nn.Identity) and engine.fold_ok(dt, w.shape[0], w.shape[1])   # first block's LayerNorm folds into its QKV GEMM
            tok2d = ops.gemm(cols, w, b, out_dtype=engine.stream_dtype(dt, w.shape[0], w.shape[1]) if emit else torch.float32, emit_ln=emit)
            tok = engine.carry_ln(tok2d, tok2d.view(B, (H // P) * (W // P), -1))
        pos = self.position_getter(B, H // P, W // P, x.device)
        if not isinstance(self.norm, nn.Identity):
            if train:
                tok = autograd.layer_norm(tok.view(-1, tok.shape[-1]), self.norm, torch.float32).view(tok.shape)
            else:
                tok = engine.layernorm(tok, self.norm, torch.float32)
        return tok, pos

    def forward(self, x, **kw):
        B, C, H, W = x.shape
        torch._assert(H == self.img_size[0], f"Input image height ({H}) doesn't match model ({self.img_size[0]}).")
        torch._assert(W == self.img_size[1], f"Input image width ({W}) doesn't match model ({self.img_size[1]}).")
        return self._embed(x)

    def _init_weights(self):
        w = self.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))


class PatchEmbedDust3R(PatchEmbedCroCo):
    """Any image size that is a multiple of the patch size (patch_embed.py:68-82)."""

    def forward(self, x, **kw):
        B, C, H, W = x.shape
        assert H % self.patch_size[0] == 0, f"Input image height ({H}) is not a multiple of patch size ({self.patch_size[0]})."
        assert W % self.patch_size[1] == 0, f"Input image width ({W}) is not a multiple of patch size ({self.patch_size[1]})."
        return self._embed(x)


class ManyAR_PatchEmbed(PatchEmbedCroCo):
    """Landscape batches whose samples may be transposed portraits (patch_embed.py:85-127)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        self.embed_dim = embed_dim
        super().__init__(img_size, patch_size, in_chans, embed_dim, norm_layer, flatten)

    def forward(self, img, true_shape):
        B, C, H, W = img.shape
        assert W >= H, f"img should be in landscape mode, but got {W=} {H=}"
        assert H % self.patch_size[0] == 0, f"Input image height ({H}) is not a multiple of patch size ({self.patch_size[0]})."
        assert W % self.patch_size[1] == 0, f"Input image width ({W}) is not a multiple of patch size ({self.patch_size[1]})."
        assert true_shape.shape == (B, 2), f"true_shape has the wrong shape={true_shape.shape}"
        height, width = true_shape.T
        is_landscape = (width >= height).to(img.device)
        if bool(is_landscape.all()):
            return self._embed(img)
        # mixed batch: portraits are embedded on their transposed image and get transposed-grid positions
        Wt, Ht = W // self.patch_size[0], H // self.patch_size[1]
        x = img.new_zeros((B, Ht * Wt, self.embed_dim), dtype=torch.float32)
        pos = torch.zeros((B, Ht * Wt, 2), dtype=torch.int64, device=img.device)
        is_portrait = ~is_landscape
        if bool(is_landscape.any()):
            t, p = self._embed(img[is_landscape].contiguous())
            x[is_landscape], pos[is_landscape] = t, p
        t, p = self._embed(img[is_portrait].swapaxes(-1, -2).contiguous())
        x[is_portrait], pos[is_portrait] = t, p
        return x, pos
