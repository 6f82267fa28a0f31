"""ORACLE — test infrastructure, not product code.

A functional, CPU-only, fp32 restatement of the reference's DUSt3R two-view pointmap path, written from
the arithmetic in SURVEY.md Appendix A.  Every function takes a flat ``state_dict``-style mapping whose
keys are the reference's parameter names, so the same weights drive the reference, this oracle and the
HIP implementation.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
The product package ``uniception_amd`` never does.

Pinning: UniCeption holds no golden vectors for this path, so the oracle is pinned against
the reference itself, imported in the build container by ``tests/golden/make_golden.py``; the resulting
fixtures live in ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` replays them.

Reference citations are file:line under /root/reference/uniception/models/.
"""
import math
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ---------------------------------------------------------------------------------------------
# deterministic, name-keyed weight filler (independent of module construction order / torch RNG)
# ---------------------------------------------------------------------------------------------
def filler_tensor(name: str, shape: Sequence[int], gain: float = 1.0) -> Tensor:
    """Values for parameter `name`: Philox stream keyed by crc32(name).
    * >=2-D weights : N(0, gain^2 / fan_in), fan_in = prod(shape[1:])
    * 1-D '...norm*.weight' : 1 + 0.1 N(0,1)      * other 1-D (biases) : 0.1 N(0,1)
    """
    seed = zlib.crc32(name.encode("utf-8"))
    rng = np.random.Generator(np.random.Philox(key=seed))
    x = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        x *= np.float32(gain / math.sqrt(fan_in))
    else:
        leaf = name.rsplit(".", 2)
        is_norm_weight = name.endswith(".weight") and "norm" in (leaf[-2] if len(leaf) >= 2 else "")
        x = (1.0 + 0.1 * x) if is_norm_weight else 0.1 * x
        x = x.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x))


def fill_state_dict_(state_dict: SD, gain: float = 1.0, gains: Optional[Dict[str, float]] = None) -> None:
    """In-place fill of every floating tensor of a module's state_dict, keys visited in sorted order
    (aliased entries end with the value of their last alias — identical for any module with the same keys).
    `gains` maps a key substring to a gain override."""
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        if not torch.is_floating_point(t):
            continue
        g = gain
        if gains:
            for sub, gv in gains.items():
                if sub in key:
                    g = gv
        t.copy_(filler_tensor(key, t.shape, g).to(t.dtype))


def make_images(seed: int, B: int, H: int, W: int) -> Tuple[Tensor, Tensor]:
    """Two synthetic views, randn like examples/models/dust3r/profile_dust3r.py:34-39 but numpy-seeded."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    a = rng.standard_normal(size=(2, B, 3, H, W), dtype=np.float32)
    return torch.from_numpy(a[0].copy()), torch.from_numpy(a[1].copy())


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def grid_positions(B: int, h: int, w: int) -> Tensor:
    """(y, x) int64 coordinates of a row-major h x w token grid, [B, h*w, 2]
    (libs/croco/patch_embed.py:25-31, utils/positional_encoding.py:15-23)."""
    ys = torch.arange(h).repeat_interleave(w)
    xs = torch.arange(w).repeat(h)
    return torch.stack([ys, xs], dim=-1)[None].expand(B, -1, -1).contiguous()


def rope2d(tokens: Tensor, positions: Tensor, base: float = 100.0, F0: float = 1.0) -> Tensor:
    """2-D rotary embedding on tokens [B,H,N,D] with positions [B,N,2] (y,x).
    Quarter layout [u_y | v_y | u_x | v_x], omega_i = base^(-i/Q), Q = D/4; angles in fp32
    (libs/croco/curope/kernels.cu:36-81, curope.cpp:21-46; equals libs/croco/pos_embed.py:116-155)."""
    B, H, N, D = tokens.shape
    Q = D // 4
    omega = F0 / (base ** (torch.arange(Q, dtype=torch.float32) / Q))
    out = tokens.clone()
    for axis in range(2):
        ang = positions[:, :, axis].to(torch.float32)[:, None, :, None] * omega  # B,1,N,Q
        c, s = torch.cos(ang), torch.sin(ang)
        lo = slice(axis * 2 * Q, axis * 2 * Q + Q)
        hi = slice(axis * 2 * Q + Q, axis * 2 * Q + 2 * Q)
        u, v = tokens[..., lo], tokens[..., hi]
        out[..., lo] = u * c - v * s
        out[..., hi] = v * c + u * s
    return out


def layer_norm(x: Tensor, sd: SD, prefix: str, eps: float = 1e-6) -> Tensor:
    """nn.LayerNorm(eps=1e-6) (encoders/croco.py:32; info_sharing/cross_attention_transformer.py:42)."""
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def linear(x: Tensor, sd: SD, prefix: str) -> Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def sdpa(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(Dh)) v, explicit (the naive branch of libs/croco/blocks.py:116-120)."""
    att = torch.softmax((q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5), dim=-1)
    return att @ v


def mlp(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """fc2(GELU_erf(fc1(x))) (libs/croco/blocks.py:80-86; utils/transformer_blocks.py:82-89)."""
    return linear(F.gelu(linear(x, sd, prefix + ".fc1")), sd, prefix + ".fc2")


def _qk_norm(t: Tensor, sd: SD, prefix: str) -> Tensor:
    """qk_norm=True: nn.LayerNorm(head_dim) (default eps 1e-5: the layers' norm_layer default, utils/transformer_blocks.py:150, 196-197)
    on q / k [B,H,N,Dh] BEFORE the positional encoding (:229); identity when the layer has no such parameters."""
    if prefix + ".weight" not in sd:
        return t
    return F.layer_norm(t, (t.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def q_scaling(n_tokens: int, scalable_softmax: bool = False, entropy_scaling: bool = False, base_token_count: int = 444,
              growth_factor: float = 1.4) -> float:
    """The reference's optional scalings of q after the positional encoding (utils/transformer_blocks.py:231-241, 360-370):
    scalable softmax q * log(N), entropy scaling q * sqrt(growth * log(N) / log(base count)); N = the QUERY count."""
    import math
    m = 1.0
    if scalable_softmax:
        m *= math.log(n_tokens)
    if entropy_scaling:
        m *= math.sqrt(growth_factor * math.log(n_tokens) / math.log(base_token_count))
    return m


def self_attention(x: Tensor, pos: Optional[Tensor], sd: SD, prefix: str, num_heads: int, base: float, q_scale: float = 1.0) -> Tensor:
    """qkv -> [q_norm, k_norm] -> RoPE(q), RoPE(k) -> [q * q_scale] -> SDPA -> proj (libs/croco/blocks.py:105-129;
    utils/transformer_blocks.py:219-256).  Wqkv rows: [0:D]=Q, [D:2D]=K, [2D:3D]=V, head-major.  pos None: no positional encoding."""
    B, N, _ = x.shape
    qkv = linear(x, sd, prefix + ".qkv")
    Cd = qkv.shape[-1] // 3           # the model width, or the layer's latent_attn_dim (utils/transformer_blocks.py:178-199)
    qkv = qkv.view(B, N, 3, num_heads, Cd // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = _qk_norm(qkv[0], sd, prefix + ".q_norm"), _qk_norm(qkv[1], sd, prefix + ".k_norm"), qkv[2]
    if pos is not None:
        q, k = rope2d(q, pos, base), rope2d(k, pos, base)
    if q_scale != 1.0:
        q = q * q_scale
    o = sdpa(q, k, v).transpose(1, 2).reshape(B, N, Cd)
    return linear(o, sd, prefix + ".proj")


def cross_attention(xq: Tensor, y: Tensor, qpos: Optional[Tensor], kpos: Optional[Tensor], sd: SD, prefix: str, num_heads: int,
                    base: float, value: Optional[Tensor] = None, q_scale: float = 1.0) -> Tensor:
    """projq/projk/projv, [q_norm, k_norm], RoPE on q (own positions) and k (other view's), SDPA, proj
    (utils/transformer_blocks.py:345-386).  value: the value tokens when they are not the key tokens y (:341-348)."""
    B, Nq, Cd = xq.shape
    Nk = y.shape[1]
    Dh = Cd // num_heads
    q = linear(xq, sd, prefix + ".projq").view(B, Nq, num_heads, Dh).transpose(1, 2)
    k = linear(y, sd, prefix + ".projk").view(B, Nk, num_heads, Dh).transpose(1, 2)
    v = linear(y if value is None else value, sd, prefix + ".projv").view(B, Nk, num_heads, Dh).transpose(1, 2)
    q, k = _qk_norm(q, sd, prefix + ".q_norm"), _qk_norm(k, sd, prefix + ".k_norm")
    if qpos is not None:
        q, k = rope2d(q, qpos, base), rope2d(k, kpos, base)
    if q_scale != 1.0:
        q = q * q_scale
    o = sdpa(q, k, v).transpose(1, 2).reshape(B, Nq, Cd)
    return linear(o, sd, prefix + ".proj")


def encoder_block(x: Tensor, pos: Tensor, sd: SD, prefix: str, num_heads: int, base: float) -> Tensor:
    """Pre-LN MHSA + pre-LN MLP with residuals (libs/croco/blocks.py:158-161)."""
    x = x + self_attention(layer_norm(x, sd, prefix + ".norm1"), pos, sd, prefix + ".attn", num_heads, base)
    return x + mlp(layer_norm(x, sd, prefix + ".norm2"), sd, prefix + ".mlp")


def decoder_block(x: Tensor, y: Tensor, xpos: Tensor, ypos: Tensor, sd: SD, prefix: str, num_heads: int,
                  base: float) -> Tensor:
    """self-attn, cross-attn against LN_y(other view), MLP (utils/transformer_blocks.py:643-646)."""
    x = x + self_attention(layer_norm(x, sd, prefix + ".norm1"), xpos, sd, prefix + ".attn", num_heads, base)
    yn = layer_norm(y, sd, prefix + ".norm_y")
    x = x + cross_attention(layer_norm(x, sd, prefix + ".norm2"), yn, xpos, ypos, sd, prefix + ".cross_attn",
                            num_heads, base)
    return x + mlp(layer_norm(x, sd, prefix + ".norm3"), sd, prefix + ".mlp")


# ---------------------------------------------------------------------------------------------
# encoder / decoder
# ---------------------------------------------------------------------------------------------
def patch_embed(img: Tensor, sd: SD, prefix: str, P: int) -> Tuple[Tensor, Tensor]:
    """Conv2d(3->D, k=s=P)+bias -> [B,N,D], row-major tokens, plus (y,x) positions
    (libs/croco/patch_embed.py:47,69-82)."""
    B, _, H, W = img.shape
    x = F.conv2d(img, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"], stride=P)
    return x.flatten(2).transpose(1, 2), grid_positions(B, H // P, W // P)


def croco_encoder(img: Tensor, sd: SD, prefix: str, *, depth: int, num_heads: int, patch_size: int = 16,
                  base: float = 100.0) -> Tensor:
    """encoders/croco.py:147-182: patch embed -> `depth` blocks -> LN -> BCHW features."""
    B, _, H, W = img.shape
    x, pos = patch_embed(img, sd, prefix + "patch_embed", patch_size)
    for i in range(depth):
        x = encoder_block(x, pos, sd, f"{prefix}enc_blocks.{i}", num_heads, base)
    x = layer_norm(x, sd, prefix + "enc_norm")
    return x.permute(0, 2, 1).reshape(B, -1, H // patch_size, W // patch_size).contiguous()


def cross_attention_transformer(feats: List[Tensor], sd: SD, prefix: str, *, depth: int, num_heads: int,
                                indices: Sequence[int] = (), norm_intermediate: bool = True,
                                base: float = 100.0) -> Tuple[List[Tensor], List[List[Tensor]]]:
    """info_sharing/cross_attention_transformer.py:390-505 (and :191-275 when `indices` is empty).
    Returns (final per-view BCHW features, [per take-index [per-view BCHW]])."""
    V = len(feats)
    B, _, h, w = feats[0].shape
    xs = [f.permute(0, 2, 3, 1).reshape(B, h * w, -1) for f in feats]
    pos = [grid_positions(B, h, w) for _ in range(V)]
    if prefix + "proj_embed.weight" in sd:  # Identity when input_embed_dim == dim (:114-118)
        xs = [linear(x, sd, prefix + "proj_embed") for x in xs]
    dim = xs[0].shape[-1]

    def to_bchw(t):
        return t.reshape(B, h, w, dim).permute(0, 3, 1, 2).contiguous()

    taken = []
    for d in range(depth):
        new = []
        for v in range(V):
            others = [u for u in range(V) if u != v]
            y = torch.cat([xs[u] for u in others], dim=1)
            ypos = torch.cat([pos[u] for u in others], dim=1)
            new.append(decoder_block(xs[v], y, pos[v], ypos, sd, f"{prefix}multi_view_branches.{v}.{d}", num_heads, base))
        xs = new  # every view reads the previous depth's features (:470)
        if d in indices:
            taken.append([to_bchw(layer_norm(x, sd, prefix + "norm") if norm_intermediate else x) for x in xs])
    final = [to_bchw(layer_norm(x, sd, prefix + "norm")) for x in xs]
    return final, taken


# ---------------------------------------------------------------------------------------------
# heads
# ---------------------------------------------------------------------------------------------
def conv(x: Tensor, sd: SD, prefix: str, stride: int = 1, padding: int = 0) -> Tensor:
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride=stride, padding=padding)


def residual_conv_unit(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """x + conv2(relu(conv1(relu(x)))) with 3x3/p1 convs; ReLU not in place (libs/croco/dpt_block.py:156-177)."""
    y = conv(F.relu(x), sd, prefix + ".conv1", padding=1)
    y = conv(F.relu(y), sd, prefix + ".conv2", padding=1)
    return y + x


def fusion_block(path: Tensor, skip: Optional[Tensor], sd: SD, prefix: str) -> Tensor:
    """(+RCU1(skip)) -> RCU2 -> x2 bilinear align_corners=True -> 1x1 out_conv (libs/croco/dpt_block.py:225-255)."""
    out = path
    if skip is not None:
        out = out + residual_conv_unit(skip, sd, prefix + ".resConfUnit1")
    out = residual_conv_unit(out, sd, prefix + ".resConfUnit2")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return conv(out, sd, prefix + ".out_conv")


def dpt_feature(feats: List[Tensor], sd: SD, prefix: str) -> Tensor:
    """prediction_heads/dpt.py:107-232: reassemble 4 maps, project to feature_dim, fuse coarse-to-fine."""
    p = prefix + "input_process."
    l0 = F.conv_transpose2d(conv(feats[0], sd, p + "0.0.0"), sd[p + "0.0.1.weight"], sd[p + "0.0.1.bias"], stride=4)
    l1 = F.conv_transpose2d(conv(feats[1], sd, p + "1.0.0"), sd[p + "1.0.1.weight"], sd[p + "1.0.1.bias"], stride=2)
    l2 = conv(feats[2], sd, p + "2.0.0")
    l3 = conv(conv(feats[3], sd, p + "3.0.0"), sd, p + "3.0.1", stride=2, padding=1)
    layers = [conv(l, sd, f"{p}{i}.1", padding=1) for i, l in enumerate((l0, l1, l2, l3))]  # layer_rn: 3x3, no bias
    s = prefix + "scratch."
    path4 = fusion_block(layers[3], None, sd, s + "refinenet4")[:, :, : layers[2].shape[2], : layers[2].shape[3]]
    path3 = fusion_block(path4, layers[2], sd, s + "refinenet3")
    path2 = fusion_block(path3, layers[1], sd, s + "refinenet2")
    return fusion_block(path2, layers[0], sd, s + "refinenet1")


def dpt_regressor(x: Tensor, out_hw: Tuple[int, int], sd: SD, prefix: str) -> Tensor:
    """conv3x3 -> bilinear to (H,W) align_corners=True -> conv3x3 -> ReLU -> conv1x1 (prediction_heads/dpt.py:271-305)."""
    x = conv(x, sd, prefix + "conv1", padding=1)
    x = F.interpolate(x, size=out_hw, mode="bilinear", align_corners=True)
    x = F.relu(conv(x, sd, prefix + "conv2.0", padding=1))
    return conv(x, sd, prefix + "conv2.2")


def linear_head(x: Tensor, sd: SD, prefix: str, patch_size: int) -> Tensor:
    """1x1 conv to out_dim*P^2 channels, then pixel_shuffle(P) (prediction_heads/linear.py:47-54,81-82)."""
    return F.pixel_shuffle(conv(x, sd, prefix + "linear"), patch_size)


def pointmap_adaptor(x: Tensor, conf_vmin: float = 1.0, conf_vmax: float = float("inf")) -> Tuple[Tensor, Tensor]:
    """PointMapAdaptor 'exp' + ConfidenceAdaptor 'exp' (prediction_heads/adaptors.py:337-342, 1080-1083)."""
    xyz, c = x[:, :3], x[:, 3:4]
    d = xyz.norm(dim=1, keepdim=True)
    pts = xyz / d.clip(min=1e-8) * torch.expm1(d)
    conf = conf_vmin + c.exp().clip(max=conf_vmax - conf_vmin)
    return pts, conf


# ---------------------------------------------------------------------------------------------
# DINOv2 ViT-S/B/L-14 encoder (BASELINE config 4).  The reference loads this network from torch.hub
# ("facebookresearch/dinov2", encoders/dinov2.py:91-102) — third-party, not vendored, no pinned revision, not fetchable
# here — so this is a restatement of the PUBLISHED architecture (prepare_tokens_with_masks / interpolate_pos_encoding /
# NestedTensorBlock without drop-path / forward_features), anchored on the reference's call sites
# (encoders/dinov2.py:140-163 SDPA attention, :188-216 output split) and PINNED at the native 37x37 grid to an independent
# implementation of the same network: HuggingFace transformers 5.15.0's Dinov2Model / Dinov2WithRegistersModel
# (tests/golden/make_golden_dinov2_hf.py -> dinov2_hf.npz; agreement ~1e-7).  Other grids: the *_reg models' position-embedding
# resize (bicubic, antialias, explicit size) is pinned to transformers' Dinov2WithRegistersEmbeddings too (16x16, 32x24 and 50x40
# cases).  PARITY UNPINNED only for the NON-register models on other grids: the hub's resize there (a 0.1 offset folded into a scale
# factor, no antialias — dinov2_pos_embed below) is restated from the published code; transformers' Dinov2Model resizes to an explicit size.
# ---------------------------------------------------------------------------------------------
def dinov2_pos_embed(sd: SD, prefix: str, h0: int, w0: int, num_registers: int) -> Tensor:
    pe = sd[prefix + "pos_embed"].float()
    N = pe.shape[1] - 1
    M = int(math.sqrt(N))
    if (h0, w0) == (M, M):
        return pe
    grid = pe[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2)
    if num_registers > 0:      # *_reg hub models: interpolate_antialias=True, interpolate_offset=0.0
        grid = F.interpolate(grid, size=(h0, w0), mode="bicubic", antialias=True)
    else:                      # plain hub models: interpolate_offset=0.1 passed as a scale factor
        grid = F.interpolate(grid, scale_factor=((h0 + 0.1) / M, (w0 + 0.1) / M), mode="bicubic", antialias=False)
    return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, h0 * w0, -1)], dim=1)


def dinov2_encoder(img: Tensor, sd: SD, prefix: str, *, num_heads: int, num_registers: int = 0, patch_size: int = 14,
                   norm: bool = True, take: Sequence[int] = ()):
    """-> (features [B,D,h,w], registers [B,D,1+R]) and, if `take`, the (normed) token streams after those blocks."""
    B, _, H, W = img.shape
    h0, w0 = H // patch_size, W // patch_size
    x = F.conv2d(img, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=patch_size)
    x = x.flatten(2).transpose(1, 2)                                            # [B, hw, D]
    x = torch.cat([sd[prefix + "cls_token"].expand(B, -1, -1), x], dim=1) + dinov2_pos_embed(sd, prefix, h0, w0, num_registers)
    if num_registers > 0:
        x = torch.cat([x[:, :1], sd[prefix + "register_tokens"].expand(B, -1, -1), x[:, 1:]], dim=1)
    depth = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in sd if k.startswith(prefix + "blocks."))
    Cd = x.shape[-1]
    taken = []
    for i in range(depth):
        bp = f"{prefix}blocks.{i}."
        h = layer_norm(x, sd, bp + "norm1")
        N = h.shape[1]
        qkv = linear(h, sd, bp + "attn.qkv").view(B, N, 3, num_heads, Cd // num_heads).permute(2, 0, 3, 1, 4)
        a = linear(sdpa(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, N, Cd), sd, bp + "attn.proj")
        x = x + sd[bp + "ls1.gamma"] * a
        h2 = layer_norm(x, sd, bp + "norm2")
        if bp + "mlp.w12.weight" in sd:      # giant: the hub's SwiGLUFFNFused — x1, x2 = w12(x).chunk(2); w3(silu(x1) * x2)
            x1, x2 = linear(h2, sd, bp + "mlp.w12").chunk(2, dim=-1)
            x = x + sd[bp + "ls2.gamma"] * linear(F.silu(x1) * x2, sd, bp + "mlp.w3")
        else:
            x = x + sd[bp + "ls2.gamma"] * mlp(h2, sd, bp + "mlp")
        if i in take:
            taken.append(layer_norm(x, sd, prefix + "norm") if norm else x)
    xn = layer_norm(x, sd, prefix + "norm") if norm else x
    R = num_registers
    feats = xn[:, 1 + R:].permute(0, 2, 1).reshape(B, Cd, h0, w0)
    regs = xn[:, :1 + R].permute(0, 2, 1)
    return (feats, regs, taken) if take else (feats, regs)


# ---------------------------------------------------------------------------------------------
# two-view model (factory/dust3r.py:250-332)
# ---------------------------------------------------------------------------------------------
def dust3r_forward(sd: SD, img1: Tensor, img2: Tensor, *, head: str, enc_depth: int = 24, enc_heads: int = 16,
                   dec_depth: int = 12, dec_heads: int = 12, patch_size: int = 16, indices: Sequence[int] = (5, 8),
                   base: float = 100.0, collect: Optional[dict] = None):
    """Both views through one encoder batch, the 2-view decoder (un-normed intermediates at `indices` for DPT),
    per-view heads, adaptor, BHWC outputs.  `collect`, if given, receives named intermediates."""
    B, _, H, W = img1.shape
    feats = croco_encoder(torch.cat([img1, img2], 0), sd, "encoder.", depth=enc_depth, num_heads=enc_heads,
                          patch_size=patch_size, base=base)
    f1, f2 = feats[:B], feats[B:]
    final, taken = cross_attention_transformer([f1, f2], sd, "info_sharing.", depth=dec_depth, num_heads=dec_heads,
                                               indices=indices if head == "dpt" else (), norm_intermediate=False, base=base)
    if collect is not None:
        collect.update(enc_feat1=f1, enc_feat2=f2, dec_final1=final[0], dec_final2=final[1])
        for j, t in enumerate(taken):
            collect[f"dec_take{j}_1"], collect[f"dec_take{j}_2"] = t[0], t[1]
    outs = []
    for v in range(2):
        if head == "dpt":
            layered = [(f1, f2)[v], taken[0][v], taken[1][v], final[v]]
            up8 = dpt_feature(layered, sd, f"dpt_feature_head{v + 1}.")
            dec = dpt_regressor(up8, (H, W), sd, f"dpt_regressor_head{v + 1}.")
            if collect is not None:
                collect[f"dpt_up8_{v + 1}"] = up8
        elif head == "linear":
            dec = linear_head(final[v], sd, f"head{v + 1}.", patch_size)
        else:
            raise ValueError(head)
        if collect is not None:
            collect[f"decoded{v + 1}"] = dec
        pts, conf = pointmap_adaptor(dec)
        outs.append((pts.permute(0, 2, 3, 1).contiguous(), conf.permute(0, 2, 3, 1).contiguous()))
    res1 = {"pts3d": outs[0][0], "conf": outs[0][1]}
    res2 = {"pts3d_in_other_view": outs[1][0], "conf": outs[1][1]}
    return res1, res2
