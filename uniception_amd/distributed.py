"""Multi-GPU plumbing of the two-view path: one process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm).

Image pairs are independent in the forward (no cross-sample op anywhere on the path), so inference
shards the batch of pairs across ranks with NO data-path collective; a pair's two views always stay on one rank
(the cross-attention decoder couples them), and symmetrized (a,b),(b,a) neighbours are kept together so the
encoder's dedup shortcut still applies.  Collectives appear only for measurement (max over ranks) and, optionally,
to gather results on every rank.
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_pairs: int, rank: int, world: int, granule: int = 2) -> Tuple[int, int]:
    """[lo, hi) of the pairs owned by `rank`: contiguous, balanced to within one granule (2 keeps symmetrized
    neighbours together).  Every pair is owned by exactly one rank."""
    assert 0 <= rank < world and n_pairs >= 0 and granule >= 1
    g = (n_pairs + granule - 1) // granule
    per, extra = divmod(g, world)
    lo_g = rank * per + min(rank, extra)
    hi_g = lo_g + per + (1 if rank < extra else 0)
    return min(lo_g * granule, n_pairs), min(hi_g * granule, n_pairs)


def shard_views(view1: Dict, view2: Dict, rank: int, world: int, granule: int = 2) -> Tuple[Dict, Dict]:
    """Slice the view dicts of DUSt3R.forward ("img" tensor, "instance" list, other keys passed through)."""
    n = view1["img"].shape[0]
    lo, hi = shard_bounds(n, rank, world, granule)

    def cut(v):
        out = dict(v)
        out["img"] = v["img"][lo:hi]
        if "instance" in v:
            out["instance"] = list(v["instance"][lo:hi])
        return out

    return cut(view1), cut(view2)


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a host scalar (the timing contract of bench.py); identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_outputs(res: Dict[str, torch.Tensor], n_pairs: int, granule: int = 2) -> Dict[str, torch.Tensor]:
    """Concatenate per-rank result dicts (e.g. {"pts3d": [b,H,W,3], "conf": [b,H,W,1]}) back to the global pair order.
    Shards may be uneven, so every rank pads to the largest shard before the all_gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return res
    world = dist.get_world_size()
    sizes = [shard_bounds(n_pairs, r, world, granule) for r in range(world)]
    biggest = max(hi - lo for lo, hi in sizes)
    out = {}
    for k, t in res.items():
        pad = torch.zeros((biggest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        parts: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out[k] = torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
    return out
