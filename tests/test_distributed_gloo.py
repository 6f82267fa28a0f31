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


# ---------------------------------------------------------------------------------------------------------------
# training exchange step (BASELINE config 3): flat gradient buffer, bucketed in-place all-reduce from autograd hooks
# ---------------------------------------------------------------------------------------------------------------
def _toy_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.LayerNorm(16), torch.nn.Linear(16, 16),
                               torch.nn.GELU(), torch.nn.Linear(16, 3))


def _grad_worker(rank, world, port, bucket_bytes):
    from uniception_amd.training import FlatParameters, GradientBuckets
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _toy_model()
        flat = FlatParameters(model)
        buckets = GradientBuckets(flat, bucket_bytes=bucket_bytes)
        assert buckets.buckets[-1][0] == 0 and buckets.buckets[0][1] == flat.numel      # the buckets tile the flat buffer
        assert all(a[0] == b[1] for a, b in zip(buckets.buckets[:-1], buckets.buckets[1:]))
        g = torch.Generator().manual_seed(100)
        data = torch.randn(world, 5, 6, generator=g)
        for step in range(2):   # twice: the per-step bookkeeping must reset
            flat.zero_grad()
            buckets.start_step()
            model(data[rank] * (step + 1)).square().mean().backward()
            buckets.finish()
            # reference: the sum over ranks of the single-process gradients
            ref_model = _toy_model()
            tot = [torch.zeros_like(p) for p in ref_model.parameters()]
            for r in range(world):
                ref_model.zero_grad()
                ref_model(data[r] * (step + 1)).square().mean().backward()
                for t, p in zip(tot, ref_model.parameters()):
                    t += p.grad
            for (n, p), t in zip(model.named_parameters(), tot):
                assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[n][0]   # still a view of the flat buffer
                assert torch.allclose(p.grad, t, rtol=1e-5, atol=1e-6), n
    finally:
        dist.destroy_process_group()


def test_two_rank_bucketed_gradient_allreduce():
    for bucket_bytes in (1 << 30, 256):   # one bucket for everything / many small buckets
        mp.spawn(_grad_worker, args=(2, _free_port(), bucket_bytes), nprocs=2, join=True)


def test_flat_parameters_keep_module_semantics():
    from uniception_amd.training import FlatParameters
    model = _toy_model()
    x = torch.randn(4, 6)
    y0 = model(x)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    flat = FlatParameters(model)
    assert torch.equal(model(x), y0) and all(torch.equal(v, sd0[k]) for k, v in model.state_dict().items())
    # decayed tensors (matrices) first, then biases / norm parameters
    names = [n for n, _ in flat.order]
    first_nd = next(i for i, n in enumerate(names) if n.endswith("bias") or dict(model.named_parameters())[n].dim() <= 1)
    assert all(dict(model.named_parameters())[n].dim() > 1 for n in names[:first_nd])
    assert flat.n_decay == sum(p.numel() for p in model.parameters() if p.dim() > 1)
    flat.param.mul_(2.0)   # parameters are views of the flat buffer
    assert torch.equal(model[0].weight, sd0["0.weight"] * 2)
