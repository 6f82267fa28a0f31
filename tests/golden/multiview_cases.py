"""Cases of the global / alternating multi-view transformers shared by the golden generator and the tests."""
DIMS = dict(input_embed_dim=128, dim=192, num_heads=3, depth=4)      # head_dim 64 so the MFMA path applies
B, H, W = 2, 4, 6

# name -> (class key, constructor extras, number of views, per-view extra tokens, global extra tokens, IFR indices)
MV_CASES = {
    "global_rope_v3": ("global_attention", dict(custom_positional_encoding="rope", use_rand_idx_pe_for_non_reference_views=False), 3, 0, 0, None),
    "global_tokens_v2": ("global_attention", dict(use_rand_idx_pe_for_non_reference_views=True), 2, 2, 3, None),
    "global_ifr_v2": ("global_attention", dict(custom_positional_encoding="rope", use_rand_idx_pe_for_non_reference_views=False,
                                               norm_intermediate=False), 2, 0, 0, [1, 3]),
    "alt_rope_v3": ("alternating_attention", dict(custom_positional_encoding="ROPE_OBJECT"), 3, 0, 0, None),
    "alt_tokens_v2": ("alternating_attention", dict(use_pe_for_non_reference_views=True, use_rand_idx_pe_for_non_reference_views=True), 2, 2, 3, None),
    "alt_ifr_v4": ("alternating_attention", dict(custom_positional_encoding="ROPE_OBJECT", init_values=0.5), 4, 0, 0, 2),
}
# gradient-fixture-only cases (no forward golden in multiview.npz): LayerScale blocks (init_values) under training
MV_GRAD_ONLY_CASES = {
    "alt_ls_v2": ("alternating_attention", dict(custom_positional_encoding="ROPE_OBJECT", init_values=0.5), 2, 0, 0, None),
    "global_ls_tokens_v2": ("global_attention", dict(use_rand_idx_pe_for_non_reference_views=False, init_values=0.7), 2, 1, 2, None),
}


def case(name):
    return MV_CASES[name] if name in MV_CASES else MV_GRAD_ONLY_CASES[name]


# "ROPE_OBJECT": a RoPE2D(100.0) instance of the side that builds the model (only the global transformer resolves the string "rope")
RAND_SEED = 1234      # torch.manual_seed before every forward: the random view-index draw (global_attention_transformer.py:378-380)


def inputs(name):
    import torch
    _, _, V, Tp, G, _ = case(name)
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    feats = [torch.randn(B, DIMS["input_embed_dim"], H, W, generator=g) for _ in range(V)]
    per_view = [torch.randn(B, DIMS["input_embed_dim"], Tp, generator=g) for _ in range(V)] if Tp else None
    glob = torch.randn(B, DIMS["input_embed_dim"], G, generator=g) if G else None
    return feats, per_view, glob


def fill(model):
    """Name-keyed filler on parameters only (the sinusoid view table is a buffer both sides compute themselves)."""
    from oracle import dust3r_oracle as O
    sd = {k: v for k, v in model.state_dict().items() if k != "view_pos_table"}
    O.fill_state_dict_(sd)


def resolve(extra, rope_cls):
    e = dict(extra)
    if e.get("custom_positional_encoding") == "ROPE_OBJECT":
        e["custom_positional_encoding"] = rope_cls(100.0)
    return e


# cases that also have a gradient fixture (multiview_grads.npz): reference autograd of  L = sum_k <out_k, R_k>  over every output
# tensor (per-view features, extra-token features), R_k seeded
MV_GRAD_CASES = ("global_rope_v3", "alt_tokens_v2", "alt_ls_v2", "global_ls_tokens_v2")


def grad_weights(name, shapes):
    """Seeded cotangents, one per output tensor, in the order (features of view 0.., global extra tokens, per-view extra tokens)."""
    import torch
    g = torch.Generator().manual_seed(7 + sum(map(ord, name)))
    return [torch.randn(*s, generator=g) for s in shapes]


def output_list(out):
    ts = list(out.features)
    if out.additional_token_features is not None:
        ts.append(out.additional_token_features)
    if out.additional_token_features_per_view is not None:
        ts += list(out.additional_token_features_per_view)
    return ts
