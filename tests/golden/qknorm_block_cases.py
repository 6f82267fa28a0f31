"""Cases of tests/golden/qknorm_blocks_grads.npz (shared by the generator, which imports the reference, and the tests, which do not)."""
import torch

CASES = {
    "self_block_qknorm_rope": dict(kind="self", dim=128, heads=2, rope=True, B=2, gh=6, gw=5, gkh=6, gkw=5, seed=21),
    "self_block_qknorm": dict(kind="self", dim=128, heads=2, rope=False, B=1, gh=4, gw=7, gkh=4, gkw=7, seed=22),
    "cross_block_qknorm_rope": dict(kind="cross", dim=128, heads=2, rope=True, B=2, gh=5, gw=4, gkh=6, gkw=7, seed=23),
}


def make_inputs(c):
    g = torch.Generator().manual_seed(2000 + c["seed"])
    Nq, Nk = c["gh"] * c["gw"], c["gkh"] * c["gkw"]

    def pos(h, w):
        return torch.cartesian_prod(torch.arange(h), torch.arange(w)).unsqueeze(0).repeat(c["B"], 1, 1).contiguous()
    out = {"x": torch.randn(c["B"], Nq, c["dim"], generator=g), "xpos": pos(c["gh"], c["gw"]),
           "w": torch.randn(c["B"], Nq, c["dim"], generator=g)}          # loss = sum(out * w)
    if c["kind"] == "cross":
        out.update({"y": torch.randn(c["B"], Nk, c["dim"], generator=g), "ypos": pos(c["gkh"], c["gkw"])})
    return out


def sample_idx(n, k=4096):
    "evenly spaced sample of a flattened gradient (all of it when it has at most k entries)"
    return torch.arange(n) if n <= k else torch.linspace(0, n - 1, k).long()
