"""Data-parallel training step for the DUSt3R hot path (BASELINE config 3): flat fp32 parameter / gradient / moment
buffers, bucketed gradient all-reduce over RCCL (xGMI) overlapped with the backward, AdamW kernels on contiguous ranges.

Layout (per GPU, 288 GB HBM: a full fp32 replica of the 569 M-parameter model + grads + two moments is 9.1 GB):
    flat_param [P] fp32   — every trainable nn.Parameter is a view into it
    flat_grad  [P] fp32   — every param.grad is a view into it; autograd (and the gradient sink) accumulate in place
    exp_avg, exp_avg_sq [P] fp32
The flat buffers are laid out in REVERSE registration order — the order in which the backward finishes gradients (heads
first, patch embedding last) — and cut into buckets of >= bucket_bytes.  Inside a bucket the weight-decayed tensors come
first and the biases / norm parameters after them, so a bucket is one contiguous all-reduce and two contiguous AdamW ranges;
no bucket has to wait for parameters from the other end of the network.  Every slot starts on a 64-byte boundary (the
split-K reduction that writes weight gradients straight into the buffer uses 16-byte accesses).

Gradient exchange: when the last gradient of a bucket has been accumulated (post-accumulate-grad hook) the bucket's slice of
flat_grad is all-reduced asynchronously IN PLACE — no staging copies.  Collectives are issued strictly in bucket order on
every rank (a bucket that completes early waits for its predecessors), so ranks can never disagree on the sequence.  The
producer-side ordering is explicit: an event recorded on the stream that wrote the gradients gates a dedicated
communication stream, from which the collective is launched.  xGMI rings are per-link bound, so buckets are large (default
256 MiB: ~9 collectives for the whole model) rather than the 25 MiB NVSwitch-era default.  `step()` waits for the handles
and applies uc_adamw with grad_scale = 1/world_size (the mean).
"""
import contextlib
import time
from typing import List

import torch
import torch.distributed as dist
import torch.nn as nn

from . import autograd, engine, ops

SLOT_ALIGN = 16   # elements (64 bytes)


def _no_decay(name: str, p: torch.Tensor) -> bool:
    return p.dim() <= 1 or name.endswith(".bias")


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class FlatParameters:
    """Re-homes the trainable parameters of `module` into one flat fp32 buffer with a matching gradient buffer."""

    def __init__(self, module: nn.Module, bucket_bytes: int = 256 << 20):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("module has no trainable parameters")
        dev = named[0][1].device
        for n, p in named:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError(f"parameter {n}: expected fp32 on {dev}, got {p.dtype} on {p.device}")
        # buckets: runs of the reversed registration order, each re-ordered decayed-first
        cap = max(1, bucket_bytes // 4)
        groups, cur, cur_n = [], [], 0
        for n, p in reversed(named):
            cur.append((n, p))
            cur_n += _round_up(p.numel(), SLOT_ALIGN)
            if cur_n >= cap:
                groups.append(cur)
                cur, cur_n = [], 0
        if cur:
            groups.append(cur)
        self.order = []
        self.offsets = {}
        self.buckets = []        # dicts: lo, split (end of the decayed tensors), hi, names
        off = 0
        for grp in groups:
            grp = grp[::-1]   # registration order inside a bucket (keeps e.g. the K and V biases of a fused projection adjacent)
            decay = [(n, p) for n, p in grp if not _no_decay(n, p)]
            nodecay = [(n, p) for n, p in grp if _no_decay(n, p)]
            lo = off
            for n, p in decay:
                self.offsets[n] = (off, p.numel())
                off += _round_up(p.numel(), SLOT_ALIGN)
            split = off
            for n, p in nodecay:
                self.offsets[n] = (off, p.numel())
                off += _round_up(p.numel(), SLOT_ALIGN)
            self.buckets.append({"lo": lo, "split": split, "hi": off, "names": [n for n, _ in decay + nodecay]})
            self.order += decay + nodecay
        self.numel = off
        self.param = torch.zeros(self.numel, dtype=torch.float32, device=dev)   # slot padding stays zero (zero gradient, zero update)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p in self.order:
                o, k = self.offsets[n]
                self.param[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.param[o:o + k].view(p.shape)
                p.grad = self.grad[o:o + k].view(p.shape)

    def zero_grad(self) -> None:
        self.grad.zero_()
        for n, p in self.order:   # re-pin a view if user code dropped or replaced it (optimizer.zero_grad(set_to_none=True), ...)
            off, k = self.offsets[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view(p.shape)


class GradientBuckets:
    """Asynchronous in-place all-reduce of the buckets of the flat gradient buffer, launched from autograd hooks."""

    def __init__(self, flat: FlatParameters, process_group=None, force_collectives: bool = False):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collectives and dist.is_available() and dist.is_initialized())
        self.handles: List = []
        self.buckets = [(b["lo"], b["hi"]) for b in flat.buckets]
        self._pending = [len(b["names"]) for b in flat.buckets]
        self._bucket_of = {n: i for i, b in enumerate(flat.buckets) for n in b["names"]}
        self.accumulating = False          # inside Trainer.no_sync(): gradients accumulate locally, nothing is reduced
        self._comm_stream = torch.cuda.Stream(device=flat.grad.device) if flat.grad.is_cuda else None
        self._main_stream = torch.cuda.current_stream(flat.grad.device) if flat.grad.is_cuda else None   # re-recorded by start_step()
        self._hooks = []
        # communication timing (bench.py's multi-GPU line): off by default — events only when asked for
        self.measure = False
        self._timing: List = []            # per finished step: (bwd_done, comm_done, [(start, end, bytes)]) events / host seconds
        self._step_marks: List = []
        if self.active:
            for n, p in flat.order:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(n)))
        self.start_step()

    def start_step(self) -> None:
        if self._comm_stream is not None:
            self._main_stream = torch.cuda.current_stream(self.flat.grad.device)   # the stream forward / backward are launched from
        self._left = list(self._pending)
        self._done = set()
        self._complete = [False] * len(self.buckets)
        self._next = 0                     # buckets [0, _next) have been issued
        self.handles = []

    def _make_hook(self, name):
        def hook(_p):
            self._ready(name)
        return hook

    def _ready(self, name) -> None:
        if self.accumulating:
            return
        if name in self._done:
            # a second backward before step(): the bucket holding this gradient may already be all-reduced, so the new local
            # contribution would be added to a cross-rank sum — replicas would silently diverge
            raise RuntimeError(f"gradient of {name} reported twice in one step: for gradient accumulation run the earlier "
                               "micro-batches under Trainer.no_sync() (only the last backward exchanges gradients)")
        self._done.add(name)
        b = self._bucket_of[name]
        self._left[b] -= 1
        if self._left[b] == 0:
            self._complete[b] = True
            self._issue_ready()

    def _issue_ready(self) -> None:
        # strictly in bucket order on every rank: a bucket that completes early waits for its predecessors
        while self._next < len(self.buckets) and self._complete[self._next]:
            self._issue(self._next)
            self._next += 1

    def _issue(self, b: int) -> None:
        lo, hi = self.buckets[b]
        g = self.flat.grad[lo:hi]
        if self._comm_stream is not None:
            # every kernel that wrote this bucket's gradients (autograd accumulation, uc_splitk_reduce / uc_gemm_tn through
            # the gradient sink) was launched on the current stream before this point: the event orders the collective
            # behind them explicitly, whatever stream the process group uses internally
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(g.device))
            self._comm_stream.wait_event(ev)
            from . import engine
            # a hook can fire inside an AccumulateGrad node that lives on a SIDE stream (parameters first used in a forked branch):
            # the "current stream" is then that side stream, and the sink kernels of the other branch were launched on the main one
            if self._main_stream is not None:
                self._comm_stream.wait_stream(self._main_stream)
            for s in engine.all_side_streams(g.device):       # backward nodes of forked sub-graphs (engine.run_branches) write gradients there
                self._comm_stream.wait_stream(s)
            g.record_stream(self._comm_stream)
            with torch.cuda.stream(self._comm_stream):
                if self.measure:           # the point where the collective MAY start: every producer of the bucket is behind it
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record(self._comm_stream)
                    self._step_marks.append([ev0, None, (hi - lo) * g.element_size()])
                self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            if self.measure:
                self._step_marks.append([time.perf_counter(), None, (hi - lo) * g.element_size()])
            self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self) -> None:
        """Issue what the hooks could not (frozen / unused parameters leave buckets incomplete) — in bucket order, the same
        sequence on every rank — and wait for everything."""
        if not self.active:
            return
        cuda = self._comm_stream is not None
        bwd_done = None
        if self.measure:
            if cuda:
                # "the backward is over" = the last kernel of the main stream AND of every side stream a forked branch's backward ran on
                from . import engine
                cur = torch.cuda.current_stream(self.flat.grad.device)
                for s in engine.all_side_streams(self.flat.grad.device):
                    cur.wait_stream(s)
                bwd_done = torch.cuda.Event(enable_timing=True)
                bwd_done.record(cur)
            else:
                bwd_done = time.perf_counter()
        for b in range(self._next, len(self.buckets)):
            self._issue(b)
        self._next = len(self.buckets)
        if cuda:
            # the waits run on the communication stream (it then holds every collective's end: an event after a wait is that
            # collective's completion time), and the compute stream joins the communication stream once
            with torch.cuda.stream(self._comm_stream):
                for i, h in enumerate(self.handles):
                    h.wait()
                    if self.measure and i < len(self._step_marks):
                        ev1 = torch.cuda.Event(enable_timing=True)
                        ev1.record(self._comm_stream)
                        self._step_marks[i][1] = ev1
            torch.cuda.current_stream(self.flat.grad.device).wait_stream(self._comm_stream)
        else:
            for i, h in enumerate(self.handles):
                h.wait()
                if self.measure and i < len(self._step_marks):
                    self._step_marks[i][1] = time.perf_counter()
        if self.measure:
            if cuda:
                comm_done = torch.cuda.Event(enable_timing=True)
                comm_done.record(torch.cuda.current_stream(self.flat.grad.device))
            else:
                comm_done = time.perf_counter()
            self._timing.append((bwd_done, comm_done, self._step_marks))
            self._step_marks = []
        self.handles = []

    def enable_comm_timing(self, on: bool = True) -> None:
        "Record, from the next step on, when each bucket's collective may start / has ended and when the backward / the exchange end."
        self.measure = bool(on) and self.active
        self._timing, self._step_marks = [], []

    def comm_stats(self) -> dict:
        """Per-step averages over the steps finished since enable_comm_timing(): `comm_ms` = sum over buckets of (collective end −
        the point it could start), `exposed_ms` = what the compute stream waited for the exchange AFTER the backward's last kernel
        (communication not hidden under the backward), `overlapped_frac` = 1 − exposed / comm.  Synchronizes the device."""
        world = self.world
        out = {"ranks": world, "backend": (dist.get_backend(self.pg) if (dist.is_available() and dist.is_initialized()) else None),
               "buckets": len(self.buckets), "bucket_bytes": [(hi - lo) * self.flat.grad.element_size() for lo, hi in self.buckets],
               "steps": len(self._timing)}
        if not self._timing:
            return out
        cuda = self._comm_stream is not None
        if cuda:
            torch.cuda.synchronize(self.flat.grad.device)
        el = (lambda a, b: a.elapsed_time(b)) if cuda else (lambda a, b: (b - a) * 1e3)
        comm = [sum(el(m[0], m[1]) for m in marks if m[1] is not None) for _b, _c, marks in self._timing]
        exposed = [max(0.0, el(b, c)) for b, c, _m in self._timing]
        out["comm_ms"] = round(sum(comm) / len(comm), 3)
        out["exposed_ms"] = round(sum(exposed) / len(exposed), 3)
        out["overlapped_frac"] = round(1.0 - out["exposed_ms"] / out["comm_ms"], 4) if out["comm_ms"] > 0 else None
        total = sum(out["bucket_bytes"])
        # ring all-reduce moves 2 (n-1)/n of the buffer per rank: the bus bandwidth the exchange ran at while it was on the wire
        out["busbw_GBps"] = round(2.0 * (world - 1) / max(world, 1) * total / (out["comm_ms"] * 1e-3) / 1e9, 2) if out["comm_ms"] > 0 else None
        return out


class Trainer:
    """zero_grad() -> forward/backward (user code) -> step().  One process per GPU; torch.distributed (RCCL) optional."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 process_group=None, bucket_bytes: int = 256 << 20, force_collectives: bool = False):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.flat = FlatParameters(model, bucket_bytes)
        self.buckets = GradientBuckets(self.flat, process_group, force_collectives)
        self.exp_avg = torch.zeros_like(self.flat.param)
        self.exp_avg_sq = torch.zeros_like(self.flat.param)
        self.steps = 0
        autograd.set_grad_sink(True)   # TN weight gradients are reduced straight into flat.grad
        engine.bump_weight_epoch()     # the parameters were re-homed: prepared copies keyed on the old storage are stale

    @property
    def world_size(self) -> int:
        return self.buckets.world

    def enable_comm_timing(self, on: bool = True) -> None:
        self.buckets.enable_comm_timing(on)

    def comm_stats(self) -> dict:
        "GradientBuckets.comm_stats(): bucket count / bytes, per-step communication time and the part the backward did not hide."
        return self.buckets.comm_stats()

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world_size > 1:
            dist.broadcast(self.flat.param, src=src, group=self.buckets.pg)
            engine.bump_weight_epoch()

    def zero_grad(self) -> None:
        cur = torch.cuda.current_stream(self.flat.grad.device)
        for s in engine.all_side_streams(self.flat.grad.device):     # (a backward whose gradients were never stepped may still be writing)
            cur.wait_stream(s)
        self.flat.zero_grad()
        self.buckets.start_step()

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only add to the local gradient buffer; the backward
        that follows outside of it exchanges the accumulated sum."""
        prev = self.buckets.accumulating
        self.buckets.accumulating = True
        try:
            yield
        finally:
            self.buckets.accumulating = prev

    def step(self) -> None:
        self.buckets.finish()
        # gradients written by forked branches' backward nodes and by the weight-gradient stream (engine.run_branches / WGRAD_STREAM)
        cur = torch.cuda.current_stream(self.flat.grad.device)
        for s in engine.all_side_streams(self.flat.grad.device):
            cur.wait_stream(s)
        self.steps += 1
        gs = 1.0 / self.world_size
        b1, b2 = self.betas
        f = self.flat
        for b in f.buckets:
            lo, sp, hi = b["lo"], b["split"], b["hi"]
            if sp > lo:
                ops.adamw_(f.param[lo:sp], f.grad[lo:sp], self.exp_avg[lo:sp], self.exp_avg_sq[lo:sp], self.lr, b1, b2,
                           self.eps, self.weight_decay, self.steps, grad_scale=gs)
            if hi > sp:
                ops.adamw_(f.param[sp:hi], f.grad[sp:hi], self.exp_avg[sp:hi], self.exp_avg_sq[sp:hi], self.lr, b1, b2,
                           self.eps, 0.0, self.steps, grad_scale=gs)
        engine.bump_weight_epoch()   # the kernel wrote through raw pointers: invalidate the prepared-weight cache
        # the exchange of this step is over: a backward that follows without zero_grad() (gradients then accumulate on top of the
        # reduced ones, as with torch optimizers) starts a new round of bookkeeping instead of tripping the double-backward check
        self.buckets.start_step()

    # ---- checkpoint / resume -------------------------------------------------------------------
    def state_dict(self) -> dict:
        """Optimizer state for checkpointing (the parameters themselves are in model.state_dict(), whose tensors are views of
        the flat buffer).  Layout-independent: moments are stored per parameter name."""
        out = {"steps": self.steps, "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
               "exp_avg": {}, "exp_avg_sq": {}}
        for n, (off, k) in self.flat.offsets.items():
            out["exp_avg"][n] = self.exp_avg[off:off + k].detach().cpu().clone()
            out["exp_avg_sq"][n] = self.exp_avg_sq[off:off + k].detach().cpu().clone()
        return out

    def load_state_dict(self, state: dict) -> None:
        self.steps = int(state["steps"])
        self.lr, self.betas, self.eps, self.weight_decay = state["lr"], tuple(state["betas"]), state["eps"], state["weight_decay"]
        for n, (off, k) in self.flat.offsets.items():
            self.exp_avg[off:off + k].copy_(state["exp_avg"][n].reshape(-1))
            self.exp_avg_sq[off:off + k].copy_(state["exp_avg_sq"][n].reshape(-1))

    def parameters_changed(self) -> None:
        """Call after writing parameters from outside (model.load_state_dict on the flattened model copies INTO the flat
        buffer): invalidates the prepared bf16 / transposed weight copies."""
        engine.bump_weight_epoch()
