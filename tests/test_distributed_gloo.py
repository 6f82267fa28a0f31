"""CPU, world_size 2, gloo: the multi-GPU plumbing of the path (pair sharding, result gather, max-over-ranks timing).
The data path itself has no collective: pairs are independent."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from uniception_amd.distributed import allgather_outputs, max_over_ranks, shard_bounds, shard_views


def test_shard_bounds_partition_every_pair_once():
    for n in (0, 1, 2, 7, 8, 16, 33):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert lo <= hi and (lo % 2 == 0 or lo == n)
                seen += list(range(lo, hi))
            assert seen == list(range(n)), (n, world)
    # balanced to one granule
    sizes = [shard_bounds(34, r, 8)[1] - shard_bounds(34, r, 8)[0] for r in range(8)]
    assert max(sizes) - min(sizes) <= 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        img1 = torch.randn(n_pairs, 3, 4, 4, generator=g)
        img2 = torch.randn(n_pairs, 3, 4, 4, generator=g)
        v1 = {"img": img1, "instance": [f"a{i}" for i in range(n_pairs)], "data_norm_type": "dust3r"}
        v2 = {"img": img2, "instance": [f"b{i}" for i in range(n_pairs)], "data_norm_type": "dust3r"}
        s1, s2 = shard_views(v1, v2, rank, world)
        lo, hi = shard_bounds(n_pairs, rank, world)
        assert s1["img"].shape[0] == hi - lo and s1["instance"] == v1["instance"][lo:hi] and s2["data_norm_type"] == "dust3r"
        # a stand-in "model": any per-pair function of both views (pairs are independent)
        local = {"pts3d": (s1["img"] * 2 + s2["img"]).permute(0, 2, 3, 1).contiguous()}
        full = allgather_outputs(local, n_pairs)
        ref = (img1 * 2 + img2).permute(0, 2, 3, 1)
        assert torch.equal(full["pts3d"], ref)
        # timing contract: every rank reports the slowest rank's time
        assert max_over_ranks(1.0 + rank) == float(world)
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_gather_and_timing():
    for n_pairs in (8, 7):
        mp.spawn(_worker, args=(2, _free_port(), n_pairs), nprocs=2, join=True)
