"""Data-parallel training step for the DUSt3R hot path (BASELINE config 3): flat fp32 parameter / gradient / moment
buffers, bucketed gradient all-reduce over RCCL (xGMI) overlapped with the backward, one AdamW kernel per step.

Layout (per GPU, 288 GB HBM: a full fp32 replica of the 569 M-parameter model + grads + two moments is 9.1 GB):
    flat_param [P] fp32   — every trainable nn.Parameter is a view into it (decayed tensors first, then biases/norms)
    flat_grad  [P] fp32   — every param.grad is a view into it; autograd accumulates in place
    exp_avg, exp_avg_sq [P] fp32
Gradient exchange: parameters are bucketed in reverse registration order (the order the backward produces them); when the
last gradient of a bucket has been accumulated (post-accumulate-grad hook) the bucket's slice of flat_grad is all-reduced
asynchronously IN PLACE — no staging copies.  xGMI rings are per-link bound, so buckets are large (default 256 MiB: ~9
collectives for the whole model) rather than the 25 MiB NVSwitch-era default.  `step()` waits for the handles and applies
uc_adamw with grad_scale = 1/world_size (the mean) on the two contiguous ranges.
"""
from typing import List

import torch
import torch.distributed as dist
import torch.nn as nn

from . import autograd, engine, ops


def _no_decay(name: str, p: torch.Tensor) -> bool:
    return p.dim() <= 1 or name.endswith(".bias")


class FlatParameters:
    """Re-homes the trainable parameters of `module` into one flat fp32 buffer with a matching gradient buffer."""

    def __init__(self, module: nn.Module):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("module has no trainable parameters")
        dev = named[0][1].device
        for n, p in named:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError(f"parameter {n}: expected fp32 on {dev}, got {p.dtype} on {p.device}")
        decay = [(n, p) for n, p in named if not _no_decay(n, p)]
        nodecay = [(n, p) for n, p in named if _no_decay(n, p)]
        self.order = decay + nodecay
        self.n_decay = sum(p.numel() for _, p in decay)
        self.numel = sum(p.numel() for _, p in self.order)
        self.param = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.offsets = {}
        off = 0
        with torch.no_grad():
            for n, p in self.order:
                k = p.numel()
                self.param[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.param[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                self.offsets[n] = (off, k)
                off += k

    def zero_grad(self) -> None:
        self.grad.zero_()
        for n, p in self.order:   # re-pin a view if user code dropped or replaced it (optimizer.zero_grad(set_to_none=True), ...)
            off, k = self.offsets[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view(p.shape)


class GradientBuckets:
    """Asynchronous in-place all-reduce of slices of the flat gradient buffer, launched from autograd hooks."""

    def __init__(self, flat: FlatParameters, process_group=None, bucket_bytes: int = 256 << 20):
        self.flat = flat
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.handles: List = []
        self.buckets = []        # (lo, hi) element ranges of flat.grad
        self._pending = []       # gradients still missing per bucket
        self._bucket_of = {}
        self._build(bucket_bytes)
        self._hooks = []
        if self.world > 1:
            for n, p in flat.order:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(n)))

    def _build(self, bucket_bytes: int) -> None:
        # contiguous runs of the flat buffer, walked from the END (backward produces the last layers first)
        cap = max(1, bucket_bytes // 4)
        names = [n for n, _ in self.flat.order]
        hi = self.flat.numel
        cur = []
        for n in reversed(names):
            off, k = self.flat.offsets[n]
            cur.append(n)
            if hi - off >= cap:
                self._add_bucket(off, hi, cur)
                hi, cur = off, []
        if cur:
            self._add_bucket(0, hi, cur)

    def _add_bucket(self, lo, hi, names) -> None:
        b = len(self.buckets)
        self.buckets.append((lo, hi))
        self._pending.append(len(names))
        for n in names:
            self._bucket_of[n] = b

    def _ready(self, name) -> None:
        if name in self._done:          # a gradient is complete once per step, whoever reports it
            return
        self._done.add(name)
        b = self._bucket_of[name]
        self._left[b] -= 1
        if self._left[b] == 0:
            lo, hi = self.buckets[b]
            self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _make_hook(self, name):
        def hook(_p):
            self._ready(name)
        return hook

    def start_step(self) -> None:
        self._left = list(self._pending)
        self._done = set()
        self.handles = []

    def _ready(self, name) -> None:
        if name in self._done:          # a gradient is complete once per step, whoever reports it
            return
        self._done.add(name)
        b = self._bucket_of[name]
        self._left[b] -= 1
        if self._left[b] == 0:
            lo, hi = self.buckets[b]
            self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _make_hook(self, name):
        def hook(_p):
            self._ready(name)
        return hook

    def start_step(self) -> None:
        self._left = list(self._pending)
        self._done = set()
        self.handles = []

    def finish(self) -> None:
        """Wait for the in-flight buckets; reduce any bucket whose hooks did not all fire (frozen / unused parameters)."""
        if self.world == 1:
            return
        for b, left in enumerate(self._left):
            if left > 0:
                lo, hi = self.buckets[b]
                self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self._left[b] = 0
        for h in self.handles:
            h.wait()
        self.handles = []


class Trainer:
    """zero_grad() -> forward/backward (user code) -> step().  One process per GPU; torch.distributed (RCCL) optional."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 process_group=None, bucket_bytes: int = 256 << 20):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.flat = FlatParameters(model)
        self.buckets = GradientBuckets(self.flat, process_group, bucket_bytes)
        self.exp_avg = torch.zeros_like(self.flat.param)
        self.exp_avg_sq = torch.zeros_like(self.flat.param)
        self.steps = 0
        self.buckets.start_step()
        autograd.set_grad_sink(True)   # TN weight gradients are reduced straight into flat.grad

    @property
    def world_size(self) -> int:
        return self.buckets.world

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world_size > 1:
            dist.broadcast(self.flat.param, src=src, group=self.buckets.pg)
            engine.bump_weight_epoch()

    def zero_grad(self) -> None:
        self.flat.zero_grad()
        self.buckets.start_step()

    def step(self) -> None:
        self.buckets.finish()
        self.steps += 1
        nd, n = self.flat.n_decay, self.flat.numel
        gs = 1.0 / self.world_size
        b1, b2 = self.betas
        if nd > 0:
            ops.adamw_(self.flat.param[:nd], self.flat.grad[:nd], self.exp_avg[:nd], self.exp_avg_sq[:nd], self.lr, b1, b2,
                       self.eps, self.weight_decay, self.steps, grad_scale=gs)
        if n > nd:
            ops.adamw_(self.flat.param[nd:], self.flat.grad[nd:], self.exp_avg[nd:], self.exp_avg_sq[nd:], self.lr, b1, b2,
                       self.eps, 0.0, self.steps, grad_scale=gs)
        engine.bump_weight_epoch()   # the kernel wrote through raw pointers: invalidate the prepared-weight cache

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

