"""Cases of tests/golden/attn_opts.npz (shared by the generator, which imports the reference, and the tests, which do not)."""
import torch

CASES = {
    "self_qknorm_rope": dict(kind="self", dim=128, heads=2, qk_norm=True, rope=True, sep_v=False, B=2, gh=6, gw=5, gkh=6, gkw=5, seed=11),
    "self_qknorm": dict(kind="self", dim=192, heads=3, qk_norm=True, rope=False, sep_v=False, B=1, gh=4, gw=7, gkh=4, gkw=7, seed=12),
    "cross_qknorm_rope": dict(kind="cross", dim=128, heads=2, qk_norm=True, rope=True, sep_v=False, B=2, gh=5, gw=4, gkh=6, gkw=7, seed=13),
    "cross_sepv_rope": dict(kind="cross", dim=128, heads=2, qk_norm=False, rope=True, sep_v=True, B=2, gh=5, gw=4, gkh=3, gkw=8, seed=14),
    "cross_sepv_qknorm": dict(kind="cross", dim=192, heads=3, qk_norm=True, rope=False, sep_v=True, B=1, gh=4, gw=4, gkh=5, gkw=5, seed=15),
}


def make_inputs(c):
    g = torch.Generator().manual_seed(1000 + c["seed"])
    Nq, Nk = c["gh"] * c["gw"], c["gkh"] * c["gkw"]

    def pos(h, w):
        return torch.cartesian_prod(torch.arange(h), torch.arange(w)).unsqueeze(0).repeat(c["B"], 1, 1).contiguous()
    if c["kind"] == "self":
        return {"x": torch.randn(c["B"], Nq, c["dim"], generator=g), "xpos": pos(c["gh"], c["gw"])}
    return {"q": torch.randn(c["B"], Nq, c["dim"], generator=g), "k": torch.randn(c["B"], Nk, c["dim"], generator=g),
            "v": torch.randn(c["B"], Nk, c["dim"], generator=g), "qpos": pos(c["gh"], c["gw"]), "kpos": pos(c["gkh"], c["gkw"])}


# tests/golden/attn_scale_opts.npz (round 6, make_golden_attn_scale.py): the reference's optional q scalings — scalable softmax
# (q * log N) and entropy scaling (q * sqrt(growth * log N / log base)), utils/transformer_blocks.py:231-241, 360-370
SCALE_CASES = {
    "self_scalable_rope": dict(kind="self", dim=128, heads=2, qk_norm=False, rope=True, sep_v=False, B=2, gh=6, gw=5, gkh=6, gkw=5, seed=21,
                               scalable=True, entropy=False),
    "self_entropy": dict(kind="self", dim=192, heads=3, qk_norm=False, rope=False, sep_v=False, B=1, gh=4, gw=7, gkh=4, gkw=7, seed=22,
                         scalable=False, entropy=True),
    "cross_both_rope": dict(kind="cross", dim=128, heads=2, qk_norm=False, rope=True, sep_v=False, B=2, gh=5, gw=4, gkh=6, gkw=7, seed=23,
                            scalable=True, entropy=True),
    "cross_entropy_qknorm": dict(kind="cross", dim=128, heads=2, qk_norm=True, rope=False, sep_v=False, B=1, gh=4, gw=4, gkh=5, gkw=5, seed=24,
                                 scalable=False, entropy=True),
    # latent attention (utils/transformer_blocks.py:178-199): q / k / v in `latent` channels, proj back to dim
    "self_latent_rope": dict(kind="self", dim=128, heads=4, qk_norm=False, rope=True, sep_v=False, B=2, gh=6, gw=5, gkh=6, gkw=5, seed=25,
                             scalable=False, entropy=False, latent=256),
    "self_latent_narrow_qknorm": dict(kind="self", dim=192, heads=2, qk_norm=True, rope=False, sep_v=False, B=1, gh=4, gw=7, gkh=4, gkw=7, seed=26,
                                      scalable=True, entropy=False, latent=64),
}
