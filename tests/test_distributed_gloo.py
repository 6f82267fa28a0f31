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
        flat = FlatParameters(model, bucket_bytes)
        buckets = GradientBuckets(flat)
        assert buckets.buckets[0][0] == 0 and buckets.buckets[-1][1] == flat.numel      # the buckets tile the flat buffer
        assert all(a[1] == b[0] for a, b in zip(buckets.buckets[:-1], buckets.buckets[1:]))
        g = torch.Generator().manual_seed(100)
        data = torch.randn(world, 5, 6, generator=g)

        def reference(scale_fn, skip_last_on_rank=None):
            """sum over ranks of the single-process gradients"""
            ref_model = _toy_model()
            tot = {n: torch.zeros_like(p) for n, p in ref_model.named_parameters()}
            for r in range(world):
                ref_model.zero_grad()
                _toy_loss(ref_model, scale_fn(r), skip_last=(skip_last_on_rank == r)).backward()
                for n, p in ref_model.named_parameters():
                    if p.grad is not None:
                        tot[n] += p.grad
            return tot

        def check(tot):
            for n, p in model.named_parameters():
                assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[n][0]   # still a view of the flat buffer
                assert torch.allclose(p.grad, tot[n], rtol=1e-5, atol=1e-6), n

        for step in range(2):   # twice: the per-step bookkeeping must reset
            flat.zero_grad()
            buckets.start_step()
            _toy_loss(model, data[rank] * (step + 1)).backward()
            buckets.finish()
            check(reference(lambda r: data[r] * (step + 1)))
        # a rank whose loss does not reach the last layer leaves buckets incomplete there: finish() issues the rest in bucket
        # order, so both ranks still run the same sequence of collectives
        flat.zero_grad()
        buckets.start_step()
        _toy_loss(model, data[rank], skip_last=(rank == 1)).backward()
        buckets.finish()
        check(reference(lambda r: data[r], skip_last_on_rank=1))
        # gradient accumulation: micro-batches under `accumulating` only add locally, the last backward exchanges the sum
        flat.zero_grad()
        buckets.start_step()
        buckets.accumulating = True
        _toy_loss(model, data[rank]).backward()
        buckets.accumulating = False
        _toy_loss(model, data[rank] * 3).backward()
        buckets.finish()
        t1, t3 = reference(lambda r: data[r]), reference(lambda r: data[r] * 3)
        check({n: t1[n] + t3[n] for n in t1})
        # a second backward outside of it is an error, not a silent divergence
        flat.zero_grad()
        buckets.start_step()
        _toy_loss(model, data[rank]).backward()
        try:
            _toy_loss(model, data[rank]).backward()
            raised = False
        except RuntimeError as e:
            raised = "no_sync" in str(e)
        assert raised
        buckets.finish()
        # Gradient sink (autograd._wgrad under the Trainer): a Function that adds its weight gradient straight into p.grad and hands
        # autograd None.  The AccumulateGrad node of such a parameter still runs (undefined gradient) and fires the post-accumulate
        # hook on this PyTorch build, so the bucket is issued DURING the backward, not in finish() (if a build skipped the
        # hook, finish() would still reduce it — this assertion is what tells the two apart).
        class Sunk(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x, w)
                return x @ w.t()

            @staticmethod
            def backward(ctx, g):
                x, w = ctx.saved_tensors
                w.grad.add_(g.t() @ x)
                return g @ w, None
        flat.zero_grad()
        buckets.start_step()
        h = model[:-1](data[rank])
        Sunk.apply(h, model[-1].weight).square().mean().backward()     # (the last layer's bias gets no gradient: frozen-like)
        issued_in_backward = buckets._next
        buckets.finish()
        first = flat.buckets[0]["names"]
        if all(n == "5.weight" or n == "5.bias" for n in first):        # many-small-buckets layout: bucket 0 = the last layer's weight (+ bias)
            assert issued_in_backward >= (1 if "5.bias" not in first else 0)
        ref_model = _toy_model()
        tot = torch.zeros_like(ref_model[-1].weight)
        for r in range(world):
            ref_model.zero_grad()
            (ref_model[:-1](data[r]) @ ref_model[-1].weight.t()).square().mean().backward()
            tot += ref_model[-1].weight.grad
        assert torch.allclose(model[-1].weight.grad, tot, rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()


def _toy_loss(model, x, skip_last=False):
    h = model[:-1](x)
    return h.square().mean() if skip_last else model[-1](h).square().mean()


def test_two_rank_bucketed_gradient_allreduce():
    for bucket_bytes in (1 << 30, 256):   # one bucket for everything / many small buckets
        mp.spawn(_grad_worker, args=(2, _free_port(), bucket_bytes), nprocs=2, join=True)


def test_flat_parameters_keep_module_semantics():
    from uniception_amd.training import SLOT_ALIGN, FlatParameters
    model = _toy_model()
    x = torch.randn(4, 6)
    y0 = model(x)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    flat = FlatParameters(model, bucket_bytes=512)
    assert torch.equal(model(x), y0) and all(torch.equal(v, sd0[k]) for k, v in model.state_dict().items())
    params = dict(model.named_parameters())
    assert len(flat.buckets) > 1
    # buckets follow the backward: the LAST registered parameters sit in bucket 0; inside a bucket the decayed tensors
    # (matrices) come first, then biases / norm parameters; every slot is 64-byte aligned and slots do not overlap
    assert "5.weight" in flat.buckets[0]["names"] and "0.weight" in flat.buckets[-1]["names"]
    end = 0
    for b in flat.buckets:
        assert b["lo"] == end
        for n in b["names"]:
            off, k = flat.offsets[n]
            assert off % SLOT_ALIGN == 0 and off >= end and k == params[n].numel()
            assert (off < b["split"]) == (params[n].dim() > 1)
            end = off + k
        end = b["hi"]
    assert end == flat.numel
    flat.param.mul_(2.0)   # parameters are views of the flat buffer
    assert torch.equal(model[0].weight, sd0["0.weight"] * 2)


# ---------------------------------------------------------------------------------------------------------------
# bench.py starts its own ranks: `python bench.py --gpus N` from a plain shell
# ---------------------------------------------------------------------------------------------------------------
def _run_bench(args, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_self_launches_two_ranks_from_plain_python():
    """No torchrun, no WORLD_SIZE in the environment: bench.py spawns the ranks, they rendezvous on 127.0.0.1, rank 0 prints
    ONE line with n_gpus == 2 (the launch check replaces the model, which needs a GPU)."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"UNICEPTION_AMD_BENCH_LAUNCH_CHECK": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["launch_check"] is True
    assert lines[0]["max_rank_s"] >= 0.02          # the slowest rank's time (rank 1 sleeps 20 ms)
    # the N > 1 line reports the gradient exchange it ran — ranks read back from the process group, bucket count and
    # bytes, communication per step and the part the backward did not hide (same GradientBuckets.comm_stats / exchange_summary code as
    # the GPU ranks' `train_step` leg; here over gloo on a toy model)
    assert lines[0]["rccl_ranks"] == 2 and lines[0]["backend"] == "gloo"
    ex = lines[0]["train_step"]["exchange"]
    assert ex["rccl_ranks"] == 2 and ex["buckets"] == len(ex["bucket_bytes"]) >= 2 and sum(ex["bucket_bytes"]) >= 4 * (6 * 32 + 32 * 32 + 32 * 3)
    assert ex["steps_measured"] == 2 and ex["comm_ms_per_step"] > 0 and ex["exposed_comm_ms_per_step"] >= 0
    assert ex["exposed_comm_ms_per_step"] <= ex["comm_ms_per_step"] * 1.5 + 1.0 and ex["busbw_GBps"] is not None


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r, lines = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "UNICEPTION_AMD_BENCH_LAUNCH_CHECK": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and not lines
