"""hipGraph capture of the two-view forward for latency-bound serving (1-4 pairs per call).

One 512x512 pair is ~640 kernel launches.  `GraphedTwoView` records the launches of one forward into a hipGraph
(torch.cuda.CUDAGraph on ROCm; every uc_hip entry point launches on the stream it is handed, so the kernels land in the
capture) and replays it with the inputs copied into static buffers.  Shapes, precision mode and weights are frozen at capture
time; call `recapture()` after changing weights (the prepared bf16 weight copies are graph inputs by address, and a rebuilt
cache would leave the graph reading freed memory).

Measured (MI355X, ViT-L + DPT, 512x512): the replay takes the same time for 1 / 2 / 4 pairs (15.2 / 17.6 / 24.3 ms) as the eager
path — the asynchronous launch path already keeps ahead of the GPU, whose small-batch time is set by tile-count-starved GEMMs
(M = 2048 tokens gives 128 tiles for 256 CUs).  The graph therefore buys CPU headroom (one replay call instead of ~640
launches per forward), not latency.
"""
from typing import Dict, Tuple

import torch

from . import engine, ops


class GraphedTwoView:
    def __init__(self, model, view1: Dict, view2: Dict, precision: str = "bf16", attention: str = "bf16", warmup: int = 2):
        self.model = model
        self.precision, self.attention = precision, attention
        self.v1 = dict(view1, img=view1["img"].clone())
        self.v2 = dict(view2, img=view2["img"].clone())
        self._warmup = warmup
        # frozen into the graph: whether the batch is a symmetrized one (the factory then encodes img[::2] only and interleaves,
        # factory/dust3r.py:227-238) and the normalization tag the encoder checked
        self._signature = self._view_signature(view1, view2)
        self.recapture()

    @staticmethod
    def _view_signature(view1: Dict, view2: Dict):
        from .models.factory.dust3r import is_symmetrized
        return (bool(is_symmetrized(view1, view2)), view1.get("data_norm_type"), view2.get("data_norm_type"))

    def _forward(self):
        with torch.no_grad(), engine.precision(self.precision), engine.attention_precision(self.attention):
            return self.model(self.v1, self.v2)

    def recapture(self) -> None:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the default stream: builds weight caches, sets kernel attributes
            for _ in range(self._warmup):
                self._forward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # hand-over buffers of the small-M GEMM path: the launches recorded below own theirs for as long as this graph lives
        # (ops.capture_scope); they come from a reserve filled here, outside the capture (allocation is not allowed inside one)
        ops.fuse_ws_release(id(self))          # (recapture: the device was synchronised above, no replay of the old graph is in flight)
        ops.fuse_ws_reserve(4)
        self.graph = torch.cuda.CUDAGraph()
        with ops.capture_scope(id(self)), torch.cuda.graph(self.graph):
            self.out = self._forward()

    def __del__(self):
        try:
            # a replay may still be in flight: its split GEMMs own the hand-over buffers until it has finished — wait before the
            # buffers return to the free list (the next stream that pops one would share flag words and partial sums with it)
            torch.cuda.synchronize()
            self.graph = None
            ops.fuse_ws_release(id(self))
        except Exception:
            pass

    def __call__(self, view1: Dict, view2: Dict) -> Tuple[Dict, Dict]:
        """Replays the captured forward on new images of the captured shape.  The returned tensors are the graph's static
        output buffers: they are overwritten by the next call (clone them to keep a result)."""
        if view1["img"].shape != self.v1["img"].shape or view2["img"].shape != self.v2["img"].shape:
            raise ValueError(f"captured for {tuple(self.v1['img'].shape)}, got {tuple(view1['img'].shape)}")
        sig = self._view_signature(view1, view2)
        if sig != self._signature:
            raise ValueError(f"captured for (symmetrized, data_norm_type x2) = {self._signature}, got {sig}: the pair layout and the "
                             "normalization tag are frozen into the graph — build another GraphedTwoView for these views")
        self.v1["img"].copy_(view1["img"], non_blocking=True)
        self.v2["img"].copy_(view2["img"], non_blocking=True)
        self.graph.replay()
        return self.out
