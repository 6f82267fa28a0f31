"""Trainer (flat buffers + uc_adamw + weight-cache invalidation) against torch.optim.AdamW driven by the same HIP backward."""
import copy

import pytest
import torch

from tests.golden.cases import grad_targets
from tests.helpers import build_case_model, case_images, rel_l2

pytestmark = pytest.mark.gpu


def _loss(model, imgs, gts, mode):
    from uniception_amd import autograd, engine
    with engine.precision(mode):
        r1, r2 = model(imgs[0], imgs[1], {})
        return autograd.conf_loss(r1["pts3d"], r1["conf"], gts[0]) + autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gts[1])


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_trainer_steps_match_torch_adamw(gpu, mode):
    from uniception_amd.training import Trainer, _no_decay

    model, c = build_case_model("tiny_linear")
    model = model.to(gpu).train()
    ref = copy.deepcopy(model)
    imgs = [t.to(gpu) for t in case_images(c)]
    gts = [t.to(gpu) for t in grad_targets(c)]
    hp = dict(lr=2e-3, betas=(0.9, 0.95), eps=1e-8)
    tr = Trainer(model, weight_decay=0.05, **hp)
    named = list(ref.named_parameters())
    opt = torch.optim.AdamW([{"params": [p for n, p in named if not _no_decay(n, p)], "weight_decay": 0.05},
                             {"params": [p for n, p in named if _no_decay(n, p)], "weight_decay": 0.0}], **hp)
    losses = []
    for _ in range(3):
        tr.zero_grad()
        loss = _loss(model, imgs, gts, mode)
        loss.backward()
        tr.step()
        opt.zero_grad(set_to_none=True)
        lr_ = _loss(ref, imgs, gts, mode)
        lr_.backward()
        opt.step()
        losses.append((float(loss.detach()), float(lr_.detach())))
    # identical kernels on both sides -> the only difference is the optimizer implementation
    for (a, b) in losses:
        assert abs(a - b) / abs(b) < (1e-5 if mode == "fp32" else 2e-3), losses
    assert losses[-1][0] < losses[0][0], f"loss did not decrease: {losses}"
    # bf16: the sink path sums split-K slabs in a different order than autograd's accumulation; Adam's g / sqrt(v) turns a
    # last-bit difference of a near-zero gradient into a full-size update of that element (observed 3e-3 .. 5e-3)
    tol = 2e-5 if mode == "fp32" else 1e-2
    pm, pr = dict(model.named_parameters()), dict(ref.named_parameters())
    worst = max(rel_l2(pm[k].detach().cpu(), pr[k].detach().cpu()) for k in pm)
    print(f"\n[{mode}] losses {losses}, worst parameter deviation after 3 steps {worst:.2e}")
    assert worst < tol
    # every parameter and gradient is still a view of the flat buffers
    for n, p in model.named_parameters():
        off, k = tr.flat.offsets[n]
        assert p.data_ptr() == tr.flat.param.data_ptr() + 4 * off and p.grad.data_ptr() == tr.flat.grad.data_ptr() + 4 * off


# ---------------------------------------------------------------------------------------------------------------
# two ranks (gloo transport, both on cuda:0 — RCCL refuses two ranks on one device) running the REAL training step:
# sharded pairs, hook-launched bucket all-reduce on GPU gradients, uc_adamw — against one process on the full batch
# ---------------------------------------------------------------------------------------------------------------
def _ddp_worker(rank, world, port, out_path, mode, use_sink=True):
    import os
    import torch.distributed as dist
    from uniception_amd.training import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        model, c = build_case_model("tiny_linear")
        model = model.to(dev).train()
        imgs = [t[rank:rank + 1].to(dev) for t in case_images(c)]          # this rank's pair
        gts = [t[rank:rank + 1].to(dev) for t in grad_targets(c)]
        tr = Trainer(model, lr=2e-3, weight_decay=0.05, bucket_bytes=1 << 20)   # several buckets
        tr.broadcast_parameters(0)
        if not use_sink:
            from uniception_amd import autograd
            autograd.set_grad_sink(False)
        assert len(tr.buckets.buckets) > 1
        grads = None
        for it in range(2):
            tr.zero_grad()
            _loss(model, imgs, gts, mode).backward()
            if it == 0:     # the all-reduced gradient of the first step (before Adam amplifies rounding-level differences)
                tr.buckets.finish()
                grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
            tr.step()
        if rank == 0:
            torch.save({"params": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "grads": grads}, out_path)
    finally:
        dist.destroy_process_group()


def _spawn_two_ranks(tmp_path, tag, mode, use_sink=True):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_path = str(tmp_path / f"rank0_{tag}.pt")
    mp.spawn(_ddp_worker, args=(2, port, out_path, mode, use_sink), nprocs=2, join=True)
    return torch.load(out_path)


def test_two_rank_training_step_equals_single_process_full_batch(gpu, tmp_path):
    from uniception_amd.training import Trainer
    two_rank = _spawn_two_ranks(tmp_path, "fp32", "fp32")["params"]
    model, c = build_case_model("tiny_linear")
    model = model.to(gpu).train()
    imgs = [t.to(gpu) for t in case_images(c)]
    gts = [t.to(gpu) for t in grad_targets(c)]
    tr = Trainer(model, lr=2e-3, weight_decay=0.05)
    for _ in range(2):
        tr.zero_grad()
        _loss(model, imgs, gts, "fp32").backward()
        tr.step()
    worst = max(rel_l2(two_rank[k], v.detach().cpu()) for k, v in model.state_dict().items())
    print(f"\n[2-rank vs 1-process, fp32] worst parameter deviation after 2 steps {worst:.2e}")
    assert worst < 1e-5


def test_two_rank_bf16_gradient_sink_equals_autograd_accumulation(gpu, tmp_path):
    """bf16: TN weight gradients are reduced straight into the flat buffer (the Functions return None for them); the same
    two-rank run with the sink disabled (autograd accumulates) must give the same all-reduced gradients."""
    with_sink = _spawn_two_ranks(tmp_path, "sink", "bf16", True)["grads"]
    without = _spawn_two_ranks(tmp_path, "nosink", "bf16", False)["grads"]
    again = _spawn_two_ranks(tmp_path, "nosink2", "bf16", False)["grads"]
    # run-to-run noise floor (fp32 atomics in the LayerNorm-backward reductions) vs the sink's deviation
    noise = max(rel_l2(again[k], without[k]) for k in without)
    worst = max(rel_l2(with_sink[k], without[k]) for k in without)
    print(f"\n[2-rank bf16 all-reduced gradients] sink vs autograd accumulation {worst:.2e} (run-to-run noise {noise:.2e})")
    assert worst < max(1e-5, 4 * noise)


def test_checkpoint_resume_continues_identically(gpu, tmp_path):
    """train 2 steps, checkpoint (model + trainer state), train 2 more; a fresh process state restored from the checkpoint
    must reproduce the last 2 steps (fp32 kernels: bit-for-bit up to the atomics' summation order)."""
    from uniception_amd.training import Trainer
    model, c = build_case_model("tiny_linear")
    model = model.to(gpu).train()
    imgs = [t.to(gpu) for t in case_images(c)]
    gts = [t.to(gpu) for t in grad_targets(c)]
    tr = Trainer(model, lr=2e-3, weight_decay=0.05)

    def steps(trainer, mdl, n):
        for _ in range(n):
            trainer.zero_grad()
            _loss(mdl, imgs, gts, "fp32").backward()
            trainer.step()

    steps(tr, model, 2)
    ckpt = {"model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "trainer": tr.state_dict()}
    torch.save(ckpt, tmp_path / "ckpt.pt")
    steps(tr, model, 2)

    model2, _ = build_case_model("tiny_linear")
    model2 = model2.to(gpu).train()
    tr2 = Trainer(model2, lr=1.0)                      # hyper-parameters come back from the checkpoint
    ck = torch.load(tmp_path / "ckpt.pt")
    model2.load_state_dict(ck["model"])
    tr2.parameters_changed()
    tr2.load_state_dict(ck["trainer"])
    assert tr2.steps == 2 and tr2.lr == 2e-3
    steps(tr2, model2, 2)
    worst = max(rel_l2(v.detach().cpu(), model.state_dict()[k].detach().cpu()) for k, v in model2.state_dict().items())
    print(f"\n[resume] worst parameter deviation after resuming {worst:.2e}")
    assert worst < 1e-5
    for n, p in model2.named_parameters():   # still views of the flat buffer after load_state_dict
        off, k = tr2.flat.offsets[n]
        assert p.data_ptr() == tr2.flat.param.data_ptr() + 4 * off


# ---------------------------------------------------------------------------------------------------------------
# the exchange step on the REAL backend: RCCL ("nccl") with world_size 1 — hooks fire from the autograd thread, the
# gradient sink writes through raw pointers, the event-gated communication stream launches ncclAllReduce on slices of
# the flat buffer, uc_adamw consumes the result.  Must equal the same steps without a process group.
# ---------------------------------------------------------------------------------------------------------------
def _rccl_worker(rank, world, port, out_path, mode):
    import os
    import torch.distributed as dist
    from uniception_amd.training import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        model, c = build_case_model("tiny_linear")
        model = model.to(dev).train()
        imgs = [t.to(dev) for t in case_images(c)]
        gts = [t.to(dev) for t in grad_targets(c)]
        tr = Trainer(model, lr=2e-3, weight_decay=0.05, bucket_bytes=1 << 20, force_collectives=True)
        assert tr.buckets.active and len(tr.buckets.buckets) > 1 and len(tr.buckets._hooks) == len(tr.flat.order)
        issued = []
        tr.enable_comm_timing(True)      # (bench.py's multi-GPU line: events around every bucket's ncclAllReduce on the communication stream)
        for _ in range(2):
            tr.zero_grad()
            _loss(model, imgs, gts, mode).backward()
            issued.append(tr.buckets._next)           # buckets launched by hooks DURING the backward
            tr.step()
        stats = tr.comm_stats()
        # ---- plain DistributedDataParallel over the same modules: the Functions hand ordinary gradients to
        # autograd when no Trainer owns them, so DDP's reducer sees every parameter.  One rank: the all-reduce is the identity and the
        # gradients must equal a bare backward's.  Single-stream training: DDP's bucket copies are not ordered against side streams.
        from torch.nn.parallel import DistributedDataParallel as DDP
        from uniception_amd import autograd as uc_autograd, engine
        uc_autograd.set_grad_sink(False)
        prev_tc, engine.TRAIN_CONCURRENT = engine.TRAIN_CONCURRENT, False
        try:
            m2, _ = build_case_model("tiny_linear")
            m2 = m2.to(dev).train()
            _loss(m2, imgs, gts, mode).backward()
            bare = {k: p.grad.detach().clone() for k, p in m2.named_parameters() if p.grad is not None}
            m2.zero_grad(set_to_none=True)
            ddp = DDP(m2, device_ids=[0], find_unused_parameters=True)
            with engine.precision(mode):
                r1, r2 = ddp(imgs[0], imgs[1], {})
                (uc_autograd.conf_loss(r1["pts3d"], r1["conf"], gts[0])
                 + uc_autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gts[1])).backward()
            torch.cuda.synchronize()
            ddp_grads = {k: p.grad.detach().cpu() for k, p in m2.named_parameters() if p.grad is not None}
            bare = {k: v.cpu() for k, v in bare.items()}
        finally:
            engine.TRAIN_CONCURRENT = prev_tc
        torch.save({"params": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "issued": issued,
                    "nbuckets": len(tr.buckets.buckets), "stats": stats, "ddp": ddp_grads, "bare": bare}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_one_rank_rccl_exchange_step(gpu, tmp_path, mode):
    import socket
    import torch.multiprocessing as mp
    from uniception_amd.training import Trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out_path = str(tmp_path / f"rccl_{mode}.pt")
    mp.spawn(_rccl_worker, args=(1, port, out_path, mode), nprocs=1, join=True)
    got = torch.load(out_path)
    assert got["issued"][0] >= got["nbuckets"] - 1, "hooks must launch the collectives during the backward, not finish()"
    st = got["stats"]
    assert st["ranks"] == 1 and st["backend"] == "nccl" and st["buckets"] == got["nbuckets"] == len(st["bucket_bytes"]) and st["steps"] == 2
    assert st["comm_ms"] > 0 and 0 <= st["exposed_ms"] and st["overlapped_frac"] is not None
    print(f"\n[1-rank RCCL exchange, {mode}] {st['buckets']} buckets, comm {st['comm_ms']} ms / step, exposed {st['exposed_ms']} ms")
    assert got["ddp"].keys() == got["bare"].keys() and len(got["ddp"]) > 10
    worst_ddp = max(rel_l2(got["ddp"][k], got["bare"][k]) for k in got["bare"] if float(got["bare"][k].norm()) > 0)
    print(f"[DistributedDataParallel vs bare backward, {mode}] worst gradient deviation {worst_ddp:.2e}")
    assert worst_ddp < (1e-5 if mode == "fp32" else 1e-2)
    model, c = build_case_model("tiny_linear")
    model = model.to(gpu).train()
    imgs = [t.to(gpu) for t in case_images(c)]
    gts = [t.to(gpu) for t in grad_targets(c)]
    tr = Trainer(model, lr=2e-3, weight_decay=0.05, bucket_bytes=1 << 20)
    for _ in range(2):
        tr.zero_grad()
        _loss(model, imgs, gts, mode).backward()
        tr.step()
    worst = max(rel_l2(got["params"][k], v.detach().cpu()) for k, v in model.state_dict().items())
    print(f"\n[1-rank RCCL vs no process group, {mode}] worst parameter deviation after 2 steps {worst:.2e}")
    # (two separate runs: bias gradients accumulate with fp32 atomics in arrival order, and AdamW turns a last-bit difference of a
    # near-zero gradient into an lr-sized step of that element — seen once at 1.3e-6 in a full-suite run)
    assert worst < (2e-5 if mode == "fp32" else 1e-2)


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_two_stream_training_step_equals_the_single_stream_one(gpu, mode, tol):
    """Round 4: while autograd records, the decoder's two view branches and the two heads (disjoint parameters) and the encoder's two
    views (SHARED parameters: the forked view's weight gradients leave the read-modify-write gradient sink to autograd's accumulation)
    run on two HIP streams, forward and backward.  On the factory model with the trainer's flat gradient buffer and sink in place, the
    loss and every parameter gradient of the forked step equal the single-stream step's (fp32: to summation order; bf16: the encoder
    sees two half batches instead of one — same rows, same bits per row — and the weight-gradient sums split differently)."""
    from uniception_amd import autograd, engine
    from uniception_amd.training import Trainer

    def step(fork):
        model, c = build_case_model("vitl_linear_224")      # the factory model (the encoder fork lives in DUSt3R._encode_image_pairs)
        assert c.get("factory")
        model = model.to(gpu).train()
        tr = Trainer(model, lr=1e-3)
        img1, img2 = (t.to(gpu) for t in case_images(c))
        gt1, gt2 = (t.to(gpu) for t in grad_targets(c))
        v1 = {"img": img1, "instance": [str(i) for i in range(c["B"])], "data_norm_type": "dust3r"}
        v2 = {"img": img2, "instance": [str(100 + i) for i in range(c["B"])], "data_norm_type": "dust3r"}
        prev = (engine.TRAIN_CONCURRENT, engine.BRANCH_TOKENS_MAX)
        engine.TRAIN_CONCURRENT, engine.BRANCH_TOKENS_MAX = fork, (0 if fork else prev[1])     # (0: fork whatever the batch size)
        try:
            losses = []
            for _ in range(2):      # twice: the first call of a fork point runs its branches in order (warm-up), the second overlaps
                tr.zero_grad()
                with engine.precision(mode):
                    r1, r2 = model(v1, v2)
                    loss = autograd.conf_loss(r1["pts3d"], r1["conf"], gt1) + autograd.conf_loss(r2["pts3d_in_other_view"], r2["conf"], gt2)
                loss.backward()
                torch.cuda.synchronize()
                losses.append(float(loss.detach()))
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        finally:
            engine.TRAIN_CONCURRENT, engine.BRANCH_TOKENS_MAX = prev
            tr.close() if hasattr(tr, "close") else None
            autograd.set_grad_sink(False)
        return losses, grads

    l0, g0 = step(False)
    l1, g1 = step(True)
    assert abs(l0[1] - l1[1]) / abs(l0[1]) < (1e-6 if mode == "fp32" else 2e-3), (l0, l1)
    worst = max((rel_l2(g1[k].cpu(), g0[k].cpu()), k) for k in g0)
    print(f"\n[{mode}] forked vs single-stream training step: loss {l1[1]:.6f} vs {l0[1]:.6f}, worst gradient deviation {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < tol, worst
