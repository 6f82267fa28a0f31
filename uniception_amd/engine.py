"""Execution engine of the DUSt3R hot path: precision policy, prepared-weight cache and the fused
token-stream / NHWC pipelines that the nn.Module shells in ``uniception_amd.models`` call.

Data layout in HBM
  * token stream  : [B*N, C] row-major ("NLC"), GEMM operands in the compute dtype.  Residual stream: bf16 in bf16 inference (the
                    reference's stream under autocast: the patch embedding and every sub-layer's output linear produce bf16, and
                    bf16 + bf16 stays bf16; only LayerNorm outputs are fp32) — `set_bf16_stream(False)` keeps it in fp32 —
                    and fp32 in verification mode and in training.
  * q|k buffer    : [B*N, 2C] compute dtype, RoPE already applied by the QKV GEMM epilogue (bf16 mode);
                    V is written by the same GEMM in the packed "VT" layout [B,H,64,Npad] the attention kernel wants.
  * dense maps    : NHWC in the compute dtype inside the DPT head; BCHW-shaped tensors at the public API are
                    channels-last *views* of the same memory (no transposes), contiguous NCHW inputs are converted once.

Precision: fp32 ("verification mode", exact-fp32 kernels) unless CUDA autocast is active or
``precision("bf16")`` is in force, in which case GEMM/attention operands are bf16 with fp32 accumulation.
"""
import contextlib
import os
import weakref
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from ._lib import UcHipError

_forced_dtype: Optional[torch.dtype] = None
_forced_x3: bool = False       # precision("bf16x3"): fp32 tensors, every GEMM / conv on split bf16 operands
# Default "fp16": the reference keeps its prediction heads in fp32 under autocast (factory/dust3r.py:288-309) — which its own
# environment multiplies in TF32 (allow_tf32, libs/croco/blocks.py:15; cuDNN's convolution default).  fp16 MFMA operands carry TF32's
# 10-bit mantissa at the bf16 rate: 2e-3 from the exact-fp32 heads on the full-size model where bf16 heads are at 1.7e-2, for 1 % of
# the forward (372 -> 368 pairs/s).  "follow" = round 1-2's bf16 heads.
_head_mode: str = os.environ.get("UNICEPTION_AMD_HEAD_PRECISION", "fp16")  # "fp16" | "follow" | "fp32" | "fp32_exact"
ROPE_TABLE_NPOS = 1024  # positions covered by the fused-epilogue cos/sin table (16k px at patch 16)


@contextlib.contextmanager
def precision(name: Optional[str]):
    """Force the compute precision of the transformer GEMMs/attention: None (follow autocast) or
    "fp32"   verification mode: exact fp32 kernels (k-ascending FMA chains) everywhere;
    "bf16"   bf16 MFMA operands, fp32 accumulate — the performance mode;
    "bf16x3" fp32 tensors, every GEMM / convolution AND both products of the attention on split bf16 operands (hi.hi + hi.lo +
             lo.hi, fp32 accumulate: fp32-class results on the matrix pipe, see ops.split_bf16x3 / ops.attention_x3), fp32 softmax:
             the fast 1e-3-grade mode."""
    global _forced_dtype, _forced_x3
    prev = (_forced_dtype, _forced_x3)
    _forced_dtype = {None: None, "fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                     "bfloat16": torch.bfloat16, "bf16x3": torch.float32}[name]
    _forced_x3 = name == "bf16x3"
    try:
        yield
    finally:
        _forced_dtype, _forced_x3 = prev


def set_head_precision(mode: str) -> None:
    """"follow": prediction heads use the compute dtype;
    "fp16" (default): TF32-class heads next to a bf16 transformer — fp16 MFMA operands (10-bit mantissa, exactly TF32's: what the reference's "fp32" heads multiply with on
            its own GPUs, allow_tf32 = True in libs/croco/blocks.py:15 and cuDNN's default for convolutions), fp32 accumulate, fp16
            maps between the layers, fp32 final layer + adaptor: the reference's head policy at the cost of the bf16 heads;
    "fp32": the reference's policy (it disables autocast around the heads, factory/dust3r.py:288-309): fp32 tensors, GEMMs and
            convolutions on split bf16 operands (bf16x3, ~1e-6 relative) when the transformer runs bf16, exact kernels in
            verification mode;
    "fp32_exact": heads always on the exact fp32 kernels (slow)."""
    global _head_mode
    assert mode in ("follow", "fp16", "fp32", "fp32_exact")
    _head_mode = mode


def head_dtype_name() -> str:
    "The head policy in force (for reports)."
    return _head_mode


def train_head_dtype_name() -> str:
    "What the prediction heads compute in while autograd records (for reports): see head_dtype()."
    if _head_mode in ("fp32", "fp32_exact"):
        return "fp32-class heads in training (policy %s): fp32 tensors, split bf16 operands forward and backward" % _head_mode
    return "bf16 kernels, forward and backward (policy %s applies to inference; the reference trains its heads in fp32 / TF32)" % _head_mode


@contextlib.contextmanager
def head_precision(mode: str):
    "Scoped set_head_precision."
    prev = _head_mode
    set_head_precision(mode)
    try:
        yield
    finally:
        set_head_precision(prev)


_mm_override: Optional[str] = None      # set by the autograd Functions around their backward: the mode their forward ran in


def _fp32_matmul() -> str:
    """How an fp32-operand GEMM runs right now (hook of ops.gemm)."""
    if _mm_override is not None:
        return _mm_override
    if _forced_x3:
        return "bf16x3"
    if _head_mode == "fp32" and compute_dtype() == torch.bfloat16:
        return "bf16x3"      # the only fp32 GEMMs next to a bf16 transformer are the heads'
    return "exact"


ops.fp32_matmul_hook = _fp32_matmul


_x3_attention: bool = os.environ.get("UNICEPTION_AMD_X3_ATTENTION", "1") != "0"   # bf16x3 mode: MFMA split-operand attention (0: the exact fp32 VALU kernel)
_attn_mode: str = os.environ.get("UNICEPTION_AMD_ATTENTION", "bf16")   # "bf16" | "fp8": matrix format of the bf16 path's attention


@contextlib.contextmanager
def attention_precision(name: str):
    """"fp8": softmax(QK^T)V of the bf16 compute path runs on the e4m3 K=64 MFMA kernel (BASELINE config 5; inference only,
    head_dim 64, ~5e-2 relative error on the attention output — the format's precision); "bf16": default."""
    global _attn_mode
    assert name in ("bf16", "fp8")
    prev = _attn_mode
    _attn_mode = name
    try:
        yield
    finally:
        _attn_mode = prev


def _fp8_attention() -> bool:
    return _attn_mode == "fp8" and not torch.is_grad_enabled()


_ambient_dtype: Optional[torch.dtype] = None


@contextlib.contextmanager
def ambient(dt: torch.dtype):
    """Carry the transformer's compute dtype into a region where autocast is switched off (the factory runs its heads under
    autocast(enabled=False) like the reference): "follow"-mode heads and the bf16x3 head policy then see the same dtype
    whether bf16 came from torch.autocast or from engine.precision."""
    global _ambient_dtype
    prev = _ambient_dtype
    _ambient_dtype = dt
    try:
        yield
    finally:
        _ambient_dtype = prev


def compute_dtype() -> torch.dtype:
    if _forced_dtype is not None:
        return _forced_dtype
    if torch.is_autocast_enabled("cuda"):
        # fp16 autocast (the reference's profile_dust3r.py default) is served by the bf16 MFMA path
        return torch.bfloat16
    if _ambient_dtype is not None:
        return _ambient_dtype
    return torch.float32


def head_dtype() -> torch.dtype:
    if _head_mode == "fp16":
        # next to a bf16 transformer in inference; the verification modes (fp32, bf16x3) and training (bf16 backward kernels) follow
        cd = compute_dtype()
        if cd == torch.bfloat16 and not torch.is_grad_enabled():
            _poll_head_range()
            if not _f16_tripped:
                return torch.float16
        return cd
    return torch.float32 if _head_mode in ("fp32", "fp32_exact") else compute_dtype()


# ---------------------------------------------------------------------------------------------
# Range guard of the TF32-class (fp16-operand) heads.  fp16 has TF32's 10-bit mantissa but not its 8-bit exponent: a map beyond
# +-65504 has no fp16 value.  The kernels SATURATE there (never inf / NaN) and raise a device flag (ops.f16_sat_flag: GEMM / conv
# epilogues with fp16 outputs, conversions into fp16).  The engine reads the flag without stalling the stream: after the heads of a
# forward it copies the flag to pinned host memory behind an event; the next time a head asks for its dtype and that event has
# completed with the flag set, the policy FALLS BACK for the rest of the process to the transformer's bf16 (fp32 exponent range, the
# "follow" policy: 1.7e-2 instead of 2e-3 from exact-fp32 heads) and says so once.  head_range_exceeded() is the synchronous form —
# call it after a forward to know whether THAT result saw saturated maps (uniception_amd/tools/verify_outputs.py does).
# ---------------------------------------------------------------------------------------------
_f16_tripped: bool = False
_f16_pending = {}        # device index -> (pinned host int32, event)


def note_heads_ran() -> None:
    "Called by the factory after the heads of a forward (fp16 head policy only): snapshot the saturation flag asynchronously."
    if _head_mode != "fp16" or _f16_tripped or torch.cuda.is_current_stream_capturing():
        return
    from . import ops
    dev = torch.cuda.current_device()
    host, ev = _f16_pending.get(dev) or (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    host.copy_(ops.f16_sat_flag(), non_blocking=True)
    ev.record()
    _f16_pending[dev] = (host, ev)


HEAD_RANGE_SYNC: bool = os.environ.get("UNICEPTION_AMD_HEAD_RANGE_SYNC", "1") != "0"


def heads_saturated_now() -> bool:
    """Round 5: the guard protects the forward that trips it.  Called by the factory right after the heads of an eager
    (non-captured) inference forward that ran them in fp16: reads the 4-byte flag synchronously — the caller is about to consume the
    outputs anyway — and, if a map saturated, trips the fallback NOW; the factory then re-runs the two heads in the transformer's bf16
    and returns those maps.  (Inside a hipGraph capture no read is possible: the asynchronous path below stays, documented in
    INTEGRATION.md section 3b.  UNICEPTION_AMD_HEAD_RANGE_SYNC=0 restores the asynchronous behaviour everywhere.)"""
    if not HEAD_RANGE_SYNC or _head_mode != "fp16" or _f16_tripped or torch.cuda.is_current_stream_capturing():
        return False
    from . import ops
    dev = torch.cuda.current_device()
    host, ev = _f16_pending.get(dev) or (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    host.copy_(ops.f16_sat_flag(), non_blocking=True)
    ev.record()
    _f16_pending[dev] = (host, ev)
    ev.synchronize()
    if int(host[0]) != 0:
        _trip_head_range()
        return True
    return False


def _poll_head_range() -> None:
    global _f16_tripped
    if _f16_tripped or not _f16_pending or torch.cuda.is_current_stream_capturing():     # (an event query is not allowed inside a capture)
        return
    for dev, (host, ev) in list(_f16_pending.items()):
        if ev.query() and int(host[0]) != 0:
            _trip_head_range()
            return


def _trip_head_range() -> None:
    global _f16_tripped
    import warnings
    from . import ops
    _f16_tripped = True
    for t in ops._f16_sat_flags.values():
        t.zero_()
    warnings.warn("uniception_amd: a prediction-head map left the fp16 range (|x| > 65504, saturated) under the TF32-class head policy; "
                  "falling back to bf16 heads (engine.set_head_precision('follow')) for the rest of this process — "
                  "set_head_precision('fp32') gives fp32-class heads at 2/3 of the speed", RuntimeWarning, stacklevel=3)


def head_range_exceeded(reset: bool = False) -> bool:
    """Synchronous: has any fp16 head launch on the current device saturated a value since the flag was last cleared?  (True also
    switches the fp16 policy to its bf16 fallback, like the asynchronous poll.)  reset=True clears the flag AND the fallback."""
    global _f16_tripped
    from . import ops
    hit = bool(int(ops.f16_sat_flag().item()) != 0) or _f16_tripped
    if hit and not _f16_tripped:
        _trip_head_range()
    if reset:
        ops.f16_sat_flag().zero_()
        _f16_pending.clear()
        _f16_tripped = False
    return hit


# ---------------------------------------------------------------------------------------------
# LayerNorm folded into the GEMMs around it (bf16 inference): the GEMM that writes the fp32 residual stream also writes a
# bf16 twin of the rows and their per-row statistics (ops.gemm(emit_ln=True) -> out.uc_ln); the next sub-layer's first GEMM
# takes the twin as its A operand with gamma folded into the weight and applies rstd * (acc - mean * colsum) + (b + W beta)
# in its epilogue (ops.gemm(ln=...)).  No LayerNorm kernel, no normalized copy of the stream in HBM.
# ---------------------------------------------------------------------------------------------
_ln_fold: bool = os.environ.get("UNICEPTION_AMD_LN_FOLD", "1") != "0"


def set_ln_fold(on: bool) -> None:
    """Switch the fused LayerNorm + GEMM path (default on; off: every LayerNorm is the stand-alone kernel)."""
    global _ln_fold
    _ln_fold = bool(on)


def fold_ok(dt: torch.dtype, *dims: int) -> bool:
    """The folded path exists for bf16 operands without autograd; channel counts must be multiples of the 64-wide tile."""
    return _ln_fold and dt == torch.bfloat16 and not torch.is_grad_enabled() and all(d % 64 == 0 for d in dims)


# bf16 residual stream (bf16 inference with the folded LayerNorm): the stream between the sub-layers is stored in bf16 — what the
# reference's stream is under torch.autocast (bf16 linear outputs added to a bf16 x) — 2 + 2 bytes per element in the residual
# epilogues instead of 4 + 4 + 2 (fp32 read + fp32 write + bf16 twin): the stored rows ARE the next GEMM's A operand and the row
# statistics are those of the rounded rows.  Off: the fp32 stream of round 1 (more accurate than the reference's policy).
_bf16_stream: bool = os.environ.get("UNICEPTION_AMD_BF16_STREAM", "1") != "0"


def set_bf16_stream(on: bool) -> None:
    global _bf16_stream
    _bf16_stream = bool(on)


def bf16_stream_enabled() -> bool:
    return _bf16_stream and _ln_fold


@contextlib.contextmanager
def bf16_stream(on: bool):
    "Scoped switch of the residual-stream dtype of bf16 inference (True: bf16, the default; False: fp32)."
    global _bf16_stream
    prev = _bf16_stream
    _bf16_stream = bool(on)
    try:
        yield
    finally:
        _bf16_stream = prev


# ... and in TRAINING (round 4): under autocast the reference's residual stream is bf16 in training too (bf16 linear outputs added to a
# bf16 x; autograd hands bf16 gradients down it).  With the switch on (default) a bf16 training step keeps x, the sub-layer outputs and
# the residual gradients in bf16: LayerNorm forward / backward and the residual epilogues move half the bytes (the LayerNorm backward
# alone 8 instead of 16 bytes per element).  Weight gradients, LayerNorm statistics and every accumulation stay fp32.
_bf16_train_stream: bool = os.environ.get("UNICEPTION_AMD_BF16_TRAIN_STREAM", "1") != "0"


def set_bf16_train_stream(on: bool) -> None:
    global _bf16_train_stream
    _bf16_train_stream = bool(on)


@contextlib.contextmanager
def bf16_train_stream(on: bool):
    "Scoped set_bf16_train_stream."
    global _bf16_train_stream
    prev = _bf16_train_stream
    _bf16_train_stream = bool(on)
    try:
        yield
    finally:
        _bf16_train_stream = prev


def stream_dtype(dt: torch.dtype, *dims: int) -> torch.dtype:
    "dtype of the token residual stream: bf16 next to bf16 operands (inference: with the LayerNorm fold; training: see above), else fp32."
    if torch.is_grad_enabled():
        return torch.bfloat16 if (_bf16_train_stream and dt == torch.bfloat16 and all(d % 64 == 0 for d in dims)) else torch.float32
    return torch.bfloat16 if (_bf16_stream and fold_ok(dt, *dims)) else torch.float32


def carry_ln(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """`dst` is a reshaped view of the same rows as `src`: hand the producer's twin / statistics over."""
    side = getattr(src, "uc_ln", None)
    if side is not None:
        dst.uc_ln = side
    return dst


def finalize_ln(x2d: torch.Tensor, ln: nn.Module) -> None:
    """Materialize the (mean, rstd) rows a later ln_operand(x2d, ln) will use on the CURRENT stream (call before forking
    work that shares x2d across HIP streams: the lazily built statistics are cached on the tensor)."""
    side = getattr(x2d, "uc_ln", None)
    if side is not None and isinstance(ln, nn.LayerNorm):
        side.stats_arg(ln.eps)       # (small batches: nothing to materialize, the consumers merge the block partials themselves)


def ln_operand(x2d: torch.Tensor, ln: nn.Module, dt: torch.dtype):
    """A operand of the GEMM behind LayerNorm `ln` of the stream x2d.  Returns (operand, fold): fold is None and operand =
    LN(x) in dt (stand-alone kernel), or fold = (stats [M,2], ln) and operand = the producer's raw bf16 twin."""
    side = getattr(x2d, "uc_ln", None)
    if (side is not None and isinstance(ln, nn.LayerNorm) and ln.elementwise_affine and fold_ok(dt, x2d.shape[-1])
            and side.twin.shape == x2d.shape):
        return side.twin, (side.stats_arg(ln.eps), ln)
    return layernorm(x2d, ln, dt), None


def require_inference(*tensors) -> None:
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise UcHipError(
            "uniception_amd: this module has no HIP backward (the CroCo encoder, the cross-attention decoder, the DPT and "
            "linear heads, the adaptor and the loss do); run it under torch.no_grad() or freeze it (requires_grad_(False))."
        )


# ---------------------------------------------------------------------------------------------
# prepared weights: compute-dtype copies / re-laid-out tensors derived from nn.Parameters, cached per
# (owner module, tag) and invalidated when any source parameter is modified or replaced.
# ---------------------------------------------------------------------------------------------
_prep_cache = weakref.WeakKeyDictionary()
_weight_epoch = 0


def bump_weight_epoch() -> None:
    """Invalidate every prepared weight: call after parameters were modified through raw pointers (uc_adamw on the flat
    buffer, a broadcast into it) — such writes do not advance the tensors' autograd version counters."""
    global _weight_epoch
    _weight_epoch += 1


_check_weights: bool = os.environ.get("UNICEPTION_AMD_CHECK_WEIGHTS", "0") == "1"


def invalidate_prepared(module: Optional[nn.Module] = None) -> None:
    """Drop prepared weight copies: of `module` and its children, or of everything (None).  Same effect as
    bump_weight_epoch() but scoped.  REQUIRED after writing parameters through `.data` (`p.data.copy_(ema)`,
    `p.data.mul_()`), raw pointers or any other route that does not advance the tensors' version counters — the cache cannot
    see such writes and would keep serving bf16 / transposed / LayerNorm-folded copies of the OLD values."""
    if module is None:
        bump_weight_epoch()
        return
    for m in module.modules():
        _prep_cache.pop(m, None)


class BuiltOn:
    """Where and when a cached device object was produced: a consumer on ANOTHER stream waits for the producing stream's event
    until it has completed (micro-batch streams and two-stream branches share the prepared weights, rope tables, ...)."""
    __slots__ = ("stream", "event")

    def __init__(self):
        self.stream = self.event = None
        # (built while a hipGraph records: the launches are nodes of that graph, ordered by its own edges — and an event recorded into
        # a capture may never be queried afterwards)
        if torch.cuda.is_available() and torch.cuda.is_initialized() and not torch.cuda.is_current_stream_capturing():
            self.stream = torch.cuda.current_stream()
            self.event = torch.cuda.Event()
            self.event.record(self.stream)

    def sync(self):
        ev = self.event
        if ev is None:
            return
        if torch.cuda.is_current_stream_capturing():
            # hipEventQuery is not allowed while a stream captures (it invalidates the capture: seen at >= 40 pairs, where the GPU is
            # still behind the host at the warm-up's second pass and the events survive it).  Every capture starts from a device-wide
            # synchronize (torch.cuda.graph.__enter__, graphs.GraphedTwoView.recapture): what was built before it is complete.
            return
        if ev.query():
            self.event = None          # completed: visible to every stream from now on
            return
        cur = torch.cuda.current_stream()
        if cur != self.stream:
            cur.wait_event(ev)


def prepared(owner: nn.Module, tag, sources: Sequence[Optional[torch.Tensor]], build):
    stamp = (_weight_epoch,) + tuple((s.data_ptr(), s._version, s.device, s.dtype) if s is not None else None for s in sources)
    if _check_weights:   # debug (UNICEPTION_AMD_CHECK_WEIGHTS=1): a content checksum catches `.data` writes, at a sync per call
        stamp += tuple(float(s.detach().double().abs().sum()) if s is not None else None for s in sources)
    slot = _prep_cache.setdefault(owner, {})
    hit = slot.get(tag)
    if hit is not None and hit[0] == stamp:
        hit[2].sync()
        return hit[1]
    with torch.no_grad():
        val = build()
    slot[tag] = (stamp, val, BuiltOn())
    return val


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().float().contiguous()


def lin_weights(lin: nn.Linear, dtype: torch.dtype):
    """(W [N,K] in dtype, bias fp32|None) of an nn.Linear."""
    return prepared(lin, ("lin", dtype), (lin.weight, lin.bias),
                    lambda: (lin.weight.detach().to(dtype).contiguous(), _f32c(lin.bias)))


def kv_weights(projk: nn.Linear, projv: nn.Linear, dtype: torch.dtype):
    """Concatenated [Wk; Wv] so K and V of the other view come out of one GEMM."""
    def build():
        w = torch.cat([projk.weight.detach(), projv.weight.detach()], 0).to(dtype).contiguous()
        if projk.bias is None and projv.bias is None:
            b = None
        else:
            zk = projk.bias.detach() if projk.bias is not None else torch.zeros_like(projk.weight[:, 0])
            zv = projv.bias.detach() if projv.bias is not None else torch.zeros_like(projv.weight[:, 0])
            b = torch.cat([zk, zv]).float().contiguous()
        return w, b
    return prepared(projk, ("kv", dtype), (projk.weight, projk.bias, projv.weight, projv.bias), build)


def _fold_ln(w: torch.Tensor, b: Optional[torch.Tensor], ln: nn.LayerNorm, dtype: torch.dtype):
    """(W * gamma in dtype, b + W beta fp32, row sums of the ROUNDED W * gamma fp32): LN(x) W^T + b = rstd (x W'^T - mean colsum) + b'."""
    w32 = w.detach().float()
    wf = (w32 * ln.weight.detach().float()[None, :]).to(dtype).contiguous()
    bias = (w32 * ln.bias.detach().float()[None, :]).sum(1)      # (one-time weight prep; an elementwise product + row sum, not a vendor-BLAS gemv)
    if b is not None:
        bias = bias + b.detach().float()
    return wf, bias.contiguous(), wf.float().sum(1).contiguous()


def ln_lin_weights(lin: nn.Linear, ln: nn.LayerNorm, dtype: torch.dtype):
    """Weights of `lin` applied to LayerNorm `ln`'s output, in the folded form (see _fold_ln)."""
    return prepared(lin, ("lnlin", dtype), (lin.weight, lin.bias, ln.weight, ln.bias),
                    lambda: _fold_ln(lin.weight, lin.bias, ln, dtype))


def ln_kv_weights(projk: nn.Linear, projv: nn.Linear, ln: nn.LayerNorm, dtype: torch.dtype):
    def build():
        w = torch.cat([projk.weight.detach(), projv.weight.detach()], 0)
        zk = projk.bias.detach() if projk.bias is not None else torch.zeros_like(projk.weight[:, 0])
        zv = projv.bias.detach() if projv.bias is not None else torch.zeros_like(projv.weight[:, 0])
        return _fold_ln(w, torch.cat([zk, zv]), ln, dtype)
    return prepared(projk, ("lnkv", dtype), (projk.weight, projk.bias, projv.weight, projv.bias, ln.weight, ln.bias), build)


def conv1x1_weights(conv: nn.Conv2d, dtype: torch.dtype):
    return prepared(conv, ("c1", dtype), (conv.weight, conv.bias),
                    lambda: (conv.weight.detach().reshape(conv.out_channels, -1).to(dtype).contiguous(), _f32c(conv.bias)))


def conv3x3_weights(conv: nn.Conv2d, dtype: torch.dtype):
    """[Cout, 9*Cin] with K ordered (ky,kx,c) to match the NHWC implicit-GEMM gather."""
    return prepared(conv, ("c3", dtype), (conv.weight, conv.bias),
                    lambda: (conv.weight.detach().permute(0, 2, 3, 1).reshape(conv.out_channels, -1).to(dtype).contiguous(),
                             _f32c(conv.bias)))


def conv3x3_bn_weights(conv: nn.Conv2d, bn: nn.BatchNorm2d, dtype: torch.dtype):
    """conv3x3 -> BatchNorm2d in EVAL mode (running statistics) as one convolution: with s = gamma / sqrt(var + eps),
    BN(conv(x)) = conv_{W s}(x) + (beta + (b - mean) s) — the DPT head's `use_bn=True` residual conv units
    (libs/croco/dpt_block.py:125-176).  Same layout as conv3x3_weights."""
    def build():
        s = (bn.weight.detach().float() if bn.weight is not None else torch.ones_like(bn.running_var)) / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = (conv.weight.detach().float() * s[:, None, None, None]).permute(0, 2, 3, 1).reshape(conv.out_channels, -1).to(dtype).contiguous()
        b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(s)
        beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(s)
        return w, (beta + (b0 - bn.running_mean.detach().float()) * s).contiguous()
    return prepared(conv, ("c3bn", dtype), (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var), build)


def convt_weights(ct: nn.ConvTranspose2d, dtype: torch.dtype):
    """ConvTranspose2d(k=s): weight [Cin,Cout,k,k] -> GEMM weight [(u,v,o), Cin]; bias repeated per (u,v)."""
    k = ct.kernel_size[0]

    def build():
        w = ct.weight.detach().permute(2, 3, 1, 0).reshape(k * k * ct.out_channels, ct.in_channels).to(dtype).contiguous()
        b = None if ct.bias is None else ct.bias.detach().float().repeat(k * k).contiguous()
        return w, b
    return prepared(ct, ("ct", dtype), (ct.weight, ct.bias), build)


def patch_weights(conv: nn.Conv2d, dtype: torch.dtype):
    return prepared(conv, ("pe", dtype), (conv.weight, conv.bias),
                    lambda: (conv.weight.detach().reshape(conv.out_channels, -1).to(dtype).contiguous(), _f32c(conv.bias)))


def patch_weights_padded(conv: nn.Conv2d, dtype: torch.dtype, kpad: int):
    "patch_weights with the K dimension (Cin P P) zero-padded to `kpad` columns: a 14 x 14 patch has 588, not a multiple of the MFMA kernels' 64"
    def build():
        w = conv.weight.detach().reshape(conv.out_channels, -1).to(dtype)
        wp = torch.zeros((conv.out_channels, kpad), dtype=dtype, device=w.device)
        wp[:, :w.shape[1]] = w
        return wp, _f32c(conv.bias)
    return prepared(conv, ("pe_pad", dtype, kpad), (conv.weight, conv.bias), build)


def layerscale_lin_weights(lin: nn.Linear, gamma: torch.Tensor, dtype: torch.dtype):
    """LayerScale folded into the preceding linear: gamma * (x W^T + b) = x (gamma[:,None] W)^T + gamma*b — zero kernel work."""
    def build():
        g = gamma.detach().float()
        w = (lin.weight.detach().float() * g[:, None]).to(dtype).contiguous()
        b = (lin.bias.detach().float() * g).contiguous() if lin.bias is not None else None
        return w, b
    return prepared(lin, ("ls", dtype), (lin.weight, lin.bias, gamma), build)


def ln_params(ln: nn.LayerNorm):
    return prepared(ln, "ln", (ln.weight, ln.bias), lambda: (_f32c(ln.weight), _f32c(ln.bias)))


# ---------------------------------------------------------------------------------------------
# layout helpers at the public BCHW boundary
# ---------------------------------------------------------------------------------------------
def nlc_as_bchw(x2d: torch.Tensor, B: int, h: int, w: int) -> torch.Tensor:
    """[B*h*w, C] -> BCHW-shaped channels-last view (no copy).  A bf16 twin written by the producing LayerNorm
    (ops.layernorm(twin=True)) travels with the view as `uc_twin_nhwc` = (NHWC twin, version of the fp32 storage at creation):
    bchw_to_nhwc hands it out instead of a cast pass as long as nobody has written to the fp32 tensor since."""
    v = x2d.view(B, h, w, x2d.shape[-1]).permute(0, 3, 1, 2)
    tw = getattr(x2d, "uc_twin", None)
    if tw is None and getattr(x2d, "uc_ln", None) is not None:
        tw = x2d.uc_ln.twin           # un-normed residual-stream rows (norm_intermediate=False): the producer GEMM's bf16 twin
    if tw is not None and tw.shape == x2d.shape and tw.is_contiguous():
        v.uc_twin_nhwc = (tw.view(B, h, w, x2d.shape[-1]), x2d._version)
    return v


def chunk_bchw(feat: torch.Tensor, n: int):
    """feat.chunk(n, dim=0) that keeps the bf16 twins of nlc_as_bchw with the pieces."""
    parts = feat.chunk(n, dim=0)
    side = getattr(feat, "uc_twin_nhwc", None)
    if side is not None and len(parts) == n:
        for p_, t_ in zip(parts, side[0].chunk(n, dim=0)):
            p_.uc_twin_nhwc = (t_, side[1])
    return parts


def bchw_to_nhwc(feat: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """BCHW-shaped tensor -> contiguous NHWC tensor in `dtype` (free when it already is a channels-last view)."""
    B, C, H, W = feat.shape
    nhwc = feat.permute(0, 2, 3, 1)
    if _train(feat):   # differentiable hop: the permute is a view for channels-last features, the cast is a Function
        from . import autograd
        return autograd.convert(nhwc, dtype)
    side = getattr(feat, "uc_twin_nhwc", None)
    if (side is not None and dtype == torch.bfloat16 and feat.dtype == torch.float32 and side[1] == feat._version
            and side[0].shape == nhwc.shape and side[0].is_contiguous()):
        return side[0]     # the producer's bf16 copy of exactly these values (views share the version counter of their storage)
    if nhwc.is_contiguous():
        return nhwc if nhwc.dtype == dtype else ops.convert(nhwc, dtype)
    if not feat.is_contiguous():
        feat = feat.contiguous()
    return ops.nchw_to_nhwc(feat, dtype)


def layernorm(x: torch.Tensor, ln: nn.LayerNorm, out_dtype: torch.dtype, twin: bool = False) -> torch.Tensor:
    if torch.is_grad_enabled() and (x.requires_grad or ln.weight.requires_grad):
        from . import autograd
        x2 = x.reshape(-1, x.shape[-1])
        return autograd.layer_norm(x2, ln, out_dtype).view(x.shape)
    g, b = ln_params(ln)
    # output features (fp32 by contract) whose consumers take bf16 operands: write the bf16 copy in the same pass
    want_twin = twin and out_dtype == torch.float32 and compute_dtype() == torch.bfloat16
    return ops.layernorm(x, g, b, ln.eps, out_dtype, twin=want_twin)


# ---------------------------------------------------------------------------------------------
# RoPE plumbing
# ---------------------------------------------------------------------------------------------
def is_native_rope(rope) -> bool:
    return rope is not None and getattr(rope, "_uc_native_rope", False)


def _rope_epilogue(rope, pos2d: torch.Tensor, cols: int):
    table = ops.rope_table(pos2d.device, ROPE_TABLE_NPOS, rope.base, rope.F0)
    return (pos2d, table, cols)


def _pos2d(pos: torch.Tensor) -> torch.Tensor:
    if pos.dtype != torch.int64:
        pos = pos.long()
    return pos.reshape(-1, 2).contiguous()


# ---------------------------------------------------------------------------------------------
# attention sub-graphs on the token stream.  x2d: [B*N, C]; returns the attention-branch output
# (proj applied) with `residual` added by the proj GEMM epilogue when given.
# ---------------------------------------------------------------------------------------------
def _folded(lin: nn.Linear, fold, dtype: torch.dtype):
    """(W, b, ln-argument of ops.gemm) of `lin` behind a LayerNorm: folded when `fold` = (stats, ln) from ln_operand."""
    if fold is None:
        return lin_weights(lin, dtype) + (None,)
    w, b, cs = ln_lin_weights(lin, fold[1], dtype)
    return w, b, (fold[0], cs)


def _qk_norm(t: torch.Tensor, norm: Optional[nn.Module]) -> torch.Tensor:
    """qk_norm (utils/transformer_blocks.py:196-197, 229): LayerNorm over head_dim of q / k, [B, N, H, Dh] view -> contiguous
    [B, N, H, Dh] in the same dtype.  Runs BEFORE the positional encoding, so these layers take the unfused route (GEMM without
    the RoPE / VT epilogue, uc_layernorm over B.N.H rows of Dh, uc_rope2d in place, attention)."""
    if norm is None or isinstance(norm, nn.Identity):
        return t
    if not isinstance(norm, nn.LayerNorm) or norm.weight is None or norm.bias is None:
        raise UcHipError(f"qk_norm with {type(norm).__name__} has no HIP path (nn.LayerNorm with affine parameters only)")
    if torch.is_grad_enabled() and (t.requires_grad or norm.weight.requires_grad):
        raise UcHipError("qk_norm=True has no HIP backward: run these layers under torch.no_grad()")
    return ops.layernorm(t.contiguous(), norm.weight.detach().float(), norm.bias.detach().float(), norm.eps, t.dtype)


def _has_norm(*norms) -> bool:
    return any(n is not None and not isinstance(n, nn.Identity) for n in norms)


# token counts that are not multiples of 4: V row-major out of the QKV / KV GEMM + uc_vt_pack instead of the element-wise VT epilogue
VT_PACK_ODD: bool = os.environ.get("UNICEPTION_AMD_VT_PACK_ODD", "1") != "0"


def self_attention(h2d: torch.Tensor, B: int, N: int, qkv: nn.Linear, proj: nn.Linear, num_heads: int, rope, pos,
                   scale: float, residual: Optional[torch.Tensor], out_dtype: torch.dtype, proj_wb=None, fold=None,
                   emit_ln: bool = False, q_norm=None, k_norm=None) -> torch.Tensor:
    """proj_wb: optional prepared (W, b) overriding proj's own (e.g. with a LayerScale folded in).
    fold: from ln_operand — h2d is then the RAW bf16 stream and the LayerNorm is applied by the QKV GEMM's epilogue.
    emit_ln: the proj GEMM also writes the twin / statistics the next sub-layer's folded LayerNorm consumes.
    q_norm / k_norm: the layer's qk_norm modules (LayerNorm over head_dim, before the positional encoding)."""
    dtype = h2d.dtype
    M, Cm = h2d.shape
    Cd = qkv.out_features // 3          # width of q / k / v: the model width, or `latent_attn_dim` (utils/transformer_blocks.py:178-199)
    Dh = Cd // num_heads
    wq, bq, lnq = _folded(qkv, fold, dtype)
    wp, bp = proj_wb if proj_wb is not None else lin_weights(proj, dtype)
    native = (rope is None or is_native_rope(rope)) and not _has_norm(q_norm, k_norm)      # (qk_norm: the unfused route below)
    if dtype == torch.bfloat16 and Dh == 64 and native and _fp8_attention():
        ep = _rope_epilogue(rope, _pos2d(pos), 2 * Cd) if rope is not None else None
        t5 = ops.gemm(h2d, wq, bq, rope=ep, ln=lnq).view(B, N, 3, num_heads, Dh)
        o = ops.attention_fp8(t5[:, :, 0], t5[:, :, 1], ops.vt_pack_fp8(t5[:, :, 2]), scale)
    elif dtype == torch.bfloat16 and Dh == 64 and native and N % 4 != 0 and VT_PACK_ODD:
        # token counts that are not multiples of 4 (DINOv2: 1370 / 1369 tokens): the QKV GEMM's packed-VT epilogue would store such
        # V tiles element by element (2-byte stores: its launches ran at 0.20-0.31 of peak) — V leaves the GEMM row-major through the
        # plain 16-byte drain and one uc_vt_pack pass (HBM-bound, ~5 TB/s) re-lays it out
        ep = _rope_epilogue(rope, _pos2d(pos), 2 * Cd) if rope is not None else None
        t5 = ops.gemm(h2d, wq, bq, rope=ep, ln=lnq).view(B, N, 3, num_heads, Dh)
        o = ops.attention(t5[:, :, 0], t5[:, :, 1], ops.vt_pack(t5[:, :, 2]), scale, v_packed=True)
    elif dtype == torch.bfloat16 and Dh == 64 and native:
        vt = ops.vt_buffer(B, num_heads, N, h2d.device)
        ep = _rope_epilogue(rope, _pos2d(pos), 2 * Cd) if rope is not None else None
        qk = ops.gemm(h2d, wq, bq, rope=ep, vt=(2 * Cd, vt, N), ln=lnq)
        qk5 = qk.view(B, N, 2, num_heads, Dh)
        o = ops.attention(qk5[:, :, 0], qk5[:, :, 1], vt, scale, v_packed=True)
    else:
        if dtype == torch.bfloat16 and Dh != 64:
            raise UcHipError(f"bf16 attention needs head_dim 64 (got {Dh}); use fp32 precision for this model")
        t = ops.gemm(h2d, wq, bq, ln=lnq).view(B, N, 3, num_heads, Dh)
        q, k, v = _qk_norm(t[:, :, 0], q_norm), _qk_norm(t[:, :, 1], k_norm), t[:, :, 2]
        o = _x3_rope_attention(rope, q, k, v, pos, pos, scale)
        if o is None:
            q, k = _apply_rope(rope, q, k, pos, pos)
            o = _attention_generic(q, k, v, scale)
    emit = emit_ln and out_dtype in (torch.float32, torch.bfloat16) and fold_ok(dtype, Cm)
    return ops.gemm(o.view(M, Cd), wp, bp, residual=residual, out_dtype=out_dtype, emit_ln=emit)


def _apply_rope(rope, q, k, qpos, kpos):
    """q,k: [B,N,H,D] views.  Native rope rotates in place; a foreign callable gets the reference's [B,H,N,D] view."""
    if rope is None:
        return q, k
    if is_native_rope(rope):
        ops.rope_2d_(q, qpos.contiguous(), rope.base, rope.F0)
        ops.rope_2d_(k, kpos.contiguous(), rope.base, rope.F0)
        return q, k
    return rope(q.transpose(1, 2), qpos).transpose(1, 2), rope(k.transpose(1, 2), kpos).transpose(1, 2)


def _x3_rope_attention(rope, q, k, v, qpos, kpos, scale):
    """precision("bf16x3") with the native RoPE-2D: the rotation of q and k rides in the operand split of the split-operand attention
    (uc_attention_fwd_x3) instead of making two passes of its own over q and k.  Returns None when that path does not apply."""
    if not (_forced_x3 and _x3_attention and rope is not None and is_native_rope(rope) and q.dtype == torch.float32 and q.shape[-1] == 64
            and not torch.is_grad_enabled() and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1):
        return None
    # the kernel reads cos / sin from a table of ROPE_TABLE_NPOS positions and clamps beyond it: this is the VERIFICATION mode, so grids
    # past the table (or negative positions) take the exact-angle path instead (rope_2d_ + attention_x3 without rope) — ADVICE r3
    if not (_pos_in_table(qpos) and _pos_in_table(kpos)):
        return None
    table = ops.rope_table(q.device, ROPE_TABLE_NPOS, rope.base, rope.F0)
    return ops.attention_x3(q, k, v, scale, rope=(_pos2d(qpos), _pos2d(kpos), table))


_pos_range_cache = {}     # (data_ptr, version, numel) of a position tensor -> its positions all lie in [0, ROPE_TABLE_NPOS)


def _pos_in_table(pos: torch.Tensor) -> bool:
    key = (pos.data_ptr(), pos._version, pos.numel())
    hit = _pos_range_cache.get(key)
    if hit is None:
        if len(_pos_range_cache) > 256:
            _pos_range_cache.clear()
        lo, hi = torch.aminmax(pos)            # (one tiny reduction per position tensor, cached: positions are built once per shape)
        hit = _pos_range_cache[key] = bool(lo.item() >= 0 and hi.item() < ROPE_TABLE_NPOS)
    return hit


def _attention_generic(q, k, v, scale):
    if q.dtype == torch.bfloat16:
        q = q if q.stride(3) == 1 else q.contiguous()
        k = k if k.stride(3) == 1 else k.contiguous()
        v = v if v.stride(3) == 1 else v.contiguous()
        return ops.attention(q, k, ops.vt_pack(v), scale, v_packed=True)
    q = q if q.stride(3) == 1 else q.contiguous()
    k = k if k.stride(3) == 1 else k.contiguous()
    v = v if v.stride(3) == 1 else v.contiguous()
    if _forced_x3 and q.shape[-1] == 64 and _x3_attention and not torch.is_grad_enabled():
        # precision("bf16x3"): the two products of the attention as three bf16 MFMA products of split operands each, fp32 softmax
        return ops.attention_x3(q, k, v, scale)
    return ops.attention(q, k, v, scale)


def cross_attention(hq2d: torch.Tensor, hkv2d: torch.Tensor, B: int, Nq: int, Nk: int, projq: nn.Linear, projk: nn.Linear,
                    projv: nn.Linear, proj: nn.Linear, num_heads: int, rope, qpos, kpos, scale: float,
                    residual: Optional[torch.Tensor], out_dtype: torch.dtype, fold_q=None, fold_kv=None,
                    emit_ln: bool = False, q_norm=None, k_norm=None, hv2d: Optional[torch.Tensor] = None, proj_wb=None) -> torch.Tensor:
    """fold_q / fold_kv: from ln_operand for the query / key-value streams (see self_attention).
    proj_wb: optional prepared (W, b) overriding proj's own (a LayerScale folded in).
    q_norm / k_norm: the layer's qk_norm modules.  hv2d: the VALUE tokens when they are not the key tokens
    (utils/transformer_blocks.py:341-348: projk(key), projv(value) — two GEMMs instead of the fused [Wk; Wv] one)."""
    dtype = hq2d.dtype
    Cd = hq2d.shape[1]
    Dh = Cd // num_heads
    if hv2d is not None or _has_norm(q_norm, k_norm):
        return _cross_attention_unfused(hq2d, hkv2d, hv2d, B, Nq, Nk, projq, projk, projv, proj, num_heads, rope, qpos, kpos, scale,
                                        residual, out_dtype, fold_q, fold_kv, emit_ln, q_norm, k_norm, proj_wb)
    wq, bq, lnq = _folded(projq, fold_q, dtype)
    if fold_kv is None:
        (wkv, bkv), lnkv = kv_weights(projk, projv, dtype), None
    else:
        wkv, bkv, cskv = ln_kv_weights(projk, projv, fold_kv[1], dtype)
        lnkv = (fold_kv[0], cskv)
    wp, bp = proj_wb if proj_wb is not None else lin_weights(proj, dtype)
    native = rope is None or is_native_rope(rope)
    if dtype == torch.bfloat16 and Dh == 64 and native and _fp8_attention():
        epq = _rope_epilogue(rope, _pos2d(qpos), Cd) if rope is not None else None
        epk = _rope_epilogue(rope, _pos2d(kpos), Cd) if rope is not None else None
        q = ops.gemm(hq2d, wq, bq, rope=epq, ln=lnq).view(B, Nq, num_heads, Dh)
        kv5 = ops.gemm(hkv2d, wkv, bkv, rope=epk, ln=lnkv).view(B, Nk, 2, num_heads, Dh)
        o = ops.attention_fp8(q, kv5[:, :, 0], ops.vt_pack_fp8(kv5[:, :, 1]), scale)
    elif dtype == torch.bfloat16 and Dh == 64 and native and Nk % 4 != 0 and VT_PACK_ODD:      # (see self_attention)
        epq = _rope_epilogue(rope, _pos2d(qpos), Cd) if rope is not None else None
        epk = _rope_epilogue(rope, _pos2d(kpos), Cd) if rope is not None else None
        q = ops.gemm(hq2d, wq, bq, rope=epq, ln=lnq).view(B, Nq, num_heads, Dh)
        kv5 = ops.gemm(hkv2d, wkv, bkv, rope=epk, ln=lnkv).view(B, Nk, 2, num_heads, Dh)
        o = ops.attention(q, kv5[:, :, 0], ops.vt_pack(kv5[:, :, 1]), scale, v_packed=True)
    elif dtype == torch.bfloat16 and Dh == 64 and native:
        vt = ops.vt_buffer(B, num_heads, Nk, hq2d.device)
        epq = _rope_epilogue(rope, _pos2d(qpos), Cd) if rope is not None else None
        epk = _rope_epilogue(rope, _pos2d(kpos), Cd) if rope is not None else None
        q = ops.gemm(hq2d, wq, bq, rope=epq, ln=lnq).view(B, Nq, num_heads, Dh)
        k = ops.gemm(hkv2d, wkv, bkv, rope=epk, vt=(Cd, vt, Nk), ln=lnkv).view(B, Nk, num_heads, Dh)
        o = ops.attention(q, k, vt, scale, v_packed=True)
    else:
        if dtype == torch.bfloat16 and Dh != 64:
            raise UcHipError(f"bf16 attention needs head_dim 64 (got {Dh}); use fp32 precision for this model")
        q = ops.gemm(hq2d, wq, bq, ln=lnq).view(B, Nq, num_heads, Dh)
        kv = ops.gemm(hkv2d, wkv, bkv, ln=lnkv).view(B, Nk, 2, num_heads, Dh)
        k, v = kv[:, :, 0], kv[:, :, 1]
        o = _x3_rope_attention(rope, q, k, v, qpos, kpos, scale)
        if o is None:
            q, k = _apply_rope(rope, q, k, qpos, kpos)
            o = _attention_generic(q, k, v, scale)
    emit = emit_ln and out_dtype in (torch.float32, torch.bfloat16) and fold_ok(dtype, Cd)
    return ops.gemm(o.reshape(B * Nq, Cd), wp, bp, residual=residual, out_dtype=out_dtype, emit_ln=emit)


def _cross_attention_unfused(hq2d, hk2d, hv2d, B, Nq, Nk, projq, projk, projv, proj, num_heads, rope, qpos, kpos, scale, residual,
                             out_dtype, fold_q, fold_k, emit_ln, q_norm, k_norm, proj_wb=None):
    """CrossAttention with options the fused pipeline does not carry: qk_norm (LayerNorm of q / k over head_dim before the positional
    encoding) and value tokens that are not the key tokens.  q, k, v from three GEMMs (the query / key LayerNorm folds still apply),
    uc_layernorm for the norms, uc_rope2d in place, attention on row-major V."""
    dtype = hq2d.dtype
    Cd = hq2d.shape[1]
    Dh = Cd // num_heads
    if dtype == torch.bfloat16 and Dh != 64:
        raise UcHipError(f"bf16 attention needs head_dim 64 (got {Dh}); use fp32 precision for this model")
    if hv2d is not None and fold_k is not None:
        raise UcHipError("a folded key LayerNorm cannot serve separate value tokens")
    wq, bq, lnq = _folded(projq, fold_q, dtype)
    wk, bk, lnk = _folded(projk, fold_k, dtype)
    q = ops.gemm(hq2d, wq, bq, ln=lnq).view(B, Nq, num_heads, Dh)
    k = ops.gemm(hk2d, wk, bk, ln=lnk).view(B, Nk, num_heads, Dh)
    if hv2d is None:
        wv, bv, lnv = _folded(projv, fold_k, dtype)
        v = ops.gemm(hk2d, wv, bv, ln=lnv).view(B, Nk, num_heads, Dh)
    else:
        assert hv2d.shape[0] == B * Nk, "key and value must have the same number of tokens"
        wv, bv = lin_weights(projv, dtype)
        v = ops.gemm(hv2d, wv, bv).view(B, Nk, num_heads, Dh)
    q, k = _qk_norm(q, q_norm), _qk_norm(k, k_norm)
    o = _x3_rope_attention(rope, q, k, v, qpos, kpos, scale)
    if o is None:
        q, k = _apply_rope(rope, q, k, qpos, kpos)
        o = _attention_generic(q, k, v, scale)
    wp, bp = proj_wb if proj_wb is not None else lin_weights(proj, dtype)
    emit = emit_ln and out_dtype in (torch.float32, torch.bfloat16) and fold_ok(dtype, Cd)
    return ops.gemm(o.reshape(B * Nq, Cd), wp, bp, residual=residual, out_dtype=out_dtype, emit_ln=emit)


def mlp(h2d: torch.Tensor, fc1: nn.Linear, fc2: nn.Linear, act: str, residual: Optional[torch.Tensor],
        out_dtype: torch.dtype, fc2_wb=None, fold=None, emit_ln: bool = False) -> torch.Tensor:
    """fold / emit_ln: see self_attention (fc1 takes the folded LayerNorm, fc2 writes the next one's twin / statistics)."""
    if fold is not None and act not in ("gelu", "none", None):
        raise UcHipError("the folded LayerNorm epilogue exists for GELU / no activation")
    w1, b1, ln1 = _folded(fc1, fold, h2d.dtype)
    w2, b2 = fc2_wb if fc2_wb is not None else lin_weights(fc2, h2d.dtype)
    g = ops.gemm(h2d, w1, b1, act=act, ln=ln1)
    emit = emit_ln and out_dtype in (torch.float32, torch.bfloat16) and fold_ok(h2d.dtype, w2.shape[0])
    return ops.gemm(g, w2, b2, residual=residual, out_dtype=out_dtype, emit_ln=emit)


def mlp_swiglu(h2d: torch.Tensor, w12: nn.Linear, w3: nn.Linear, residual: Optional[torch.Tensor], out_dtype: torch.dtype,
               w3_wb=None, fold=None, emit_ln: bool = False) -> torch.Tensor:
    """DINOv2 giant's FFN (the hub's SwiGLUFFNFused): x1, x2 = w12(h).chunk(2); residual + w3(silu(x1) * x2).  w12 takes the folded
    LayerNorm like fc1 does, the gate is one HBM-bound pass (uc_swiglu), w3 writes the next LayerNorm's statistics (emit_ln)."""
    w1, b1, ln1 = _folded(w12, fold, h2d.dtype)
    w2, b2 = w3_wb if w3_wb is not None else lin_weights(w3, h2d.dtype)
    g = ops.swiglu(ops.gemm(h2d, w1, b1, ln=ln1))
    emit = emit_ln and out_dtype in (torch.float32, torch.bfloat16) and fold_ok(h2d.dtype, w2.shape[0])
    return ops.gemm(g, w2, b2, residual=residual, out_dtype=out_dtype, emit_ln=emit)


def act_name(act_module: nn.Module) -> str:
    if isinstance(act_module, nn.GELU) and getattr(act_module, "approximate", "none") == "none":
        return "gelu"
    if isinstance(act_module, nn.ReLU):
        return "relu"
    if isinstance(act_module, nn.Identity):
        return "none"
    raise UcHipError(f"activation {type(act_module).__name__} has no fused HIP epilogue (supported: GELU(erf), ReLU)")


# ---------------------------------------------------------------------------------------------
# DPT pieces on NHWC maps
# ---------------------------------------------------------------------------------------------
def _train(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def bilinear(x: torch.Tensor, Ho: int, Wo: int, crop=None) -> torch.Tensor:
    if _train(x):
        from . import autograd
        return autograd.bilinear(x, Ho, Wo, crop)
    return ops.bilinear_nhwc(x, Ho, Wo, crop)


def conv1x1(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    if _train(x, conv.weight):
        from . import autograd
        return autograd.conv1x1(x, conv)
    B, H, W, Cin = x.shape
    w, b = conv1x1_weights(conv, x.dtype)
    return ops.gemm(x.view(-1, Cin), w, b).view(B, H, W, -1)


def conv3x3(x: torch.Tensor, conv: nn.Conv2d, relu_in: bool = False, act=None, residual: Optional[torch.Tensor] = None,
            residual2: Optional[torch.Tensor] = None, grad_mask_cell=None, bn: Optional[nn.BatchNorm2d] = None) -> torch.Tensor:
    """bn: a BatchNorm2d applied to the convolution's output BEFORE act / the residuals, folded into its weights (eval mode only)."""
    if bn is not None and (bn.training or bn.running_mean is None):
        raise UcHipError("BatchNorm in the DPT head runs folded into its convolution: eval mode with running statistics only "
                         "(batch statistics in training have no HIP path)")
    if _train(x, conv.weight, residual, residual2):
        if bn is not None:
            raise UcHipError("BatchNorm in the DPT head has no HIP backward (use_bn=True models run in inference only)")
        from . import autograd
        return autograd.conv3x3(x, conv, relu_in, act, residual, residual2, grad_mask_cell)
    B, H, W, Cin = x.shape
    s = conv.stride[0]
    w, b = conv3x3_weights(conv, x.dtype) if bn is None else conv3x3_bn_weights(conv, bn, x.dtype)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    r1 = None if residual is None else residual.reshape(-1, residual.shape[-1])
    r2 = None if residual2 is None else residual2.reshape(-1, residual2.shape[-1])
    y = ops.gemm(x, w, b, act=act, residual=r1, residual2=r2, relu_a=relu_in, conv=(B, H, W, Cin, s))
    return y.view(B, Ho, Wo, -1)


def conv3x3_tail4(x: torch.Tensor, conv: nn.Conv2d, act, last: nn.Conv2d) -> Optional[torch.Tensor]:
    """conv3x3 (128 output channels) -> act -> Conv2d(128 -> 4, 1x1) in ONE kernel (the regressor tail, dpt.py:271-277): the
    128-channel map is never stored.  Returns fp32 [B,H,W,4], or None when the fused form does not apply (training, exact fp32
    kernels, other channel counts): the caller then runs the two layers separately."""
    if _train(x, conv.weight, last.weight) or conv.out_channels != 128 or last.out_channels != 4 or last.kernel_size != (1, 1):
        return None
    B, H, W, Cin = x.shape
    if conv.stride[0] != 1 or Cin % 32 != 0 or os.environ.get("UNICEPTION_AMD_FUSED_TAIL", "1") == "0":
        return None
    if not (x.dtype in (torch.bfloat16, torch.float16) or (x.dtype == torch.float32 and ops.fp32_matmul_hook() == "bf16x3")):
        return None
    w, b = conv3x3_weights(conv, x.dtype)
    w4, b4 = prepared(last, "c1x4", (last.weight, last.bias),
                      lambda: (last.weight.detach().reshape(4, -1).float().contiguous(),
                               last.bias.detach().float().contiguous() if last.bias is not None
                               else torch.zeros(4, device=last.weight.device)))
    return ops.gemm(x, w, b, act=act, conv=(B, H, W, Cin, 1), tail=(w4, b4)).view(B, H, W, 4)


def conv_transpose_ks(x: torch.Tensor, ct: nn.ConvTranspose2d) -> torch.Tensor:
    if _train(x, ct.weight):
        from . import autograd
        return autograd.conv_transpose_ks(x, ct)
    B, H, W, Cin = x.shape
    k = ct.kernel_size[0]
    w, b = convt_weights(ct, x.dtype)
    y = ops.gemm(x.view(-1, Cin), w, b)
    return ops.convt_scatter(y, B, H, W, k, ct.out_channels)


# ---------------------------------------------------------------------------------------------
# two-stream execution of independent branches (latency regime)
# ---------------------------------------------------------------------------------------------
_side_streams = {}
# Two-stream execution pays while a branch's launches leave CUs idle: -25 % time at 1 pair, -12 % at 8, -5 % at 16 pairs; at
# 64 pairs it is +1-3 % but every kernel then overlaps a foreign one, which inflates per-kernel durations (the dense GEMM's
# average 464 -> 607 us) and blurs the roofline accounting of the headline run — so it is limited to <= 16 pairs x 1024 tokens.
BRANCH_TOKENS_MAX = int(os.environ.get("UNICEPTION_AMD_BRANCH_TOKENS_MAX", "16384"))


def side_stream(device, index: int = 0) -> "torch.cuda.Stream":
    s = _side_streams.get((device, index))
    if s is None:
        s = _side_streams[(device, index)] = torch.cuda.Stream(device=device)
    return s


# CONCURRENT (UNICEPTION_AMD_CONCURRENT, default on): independent sub-graphs of a LARGE batch also run on two streams.  Every
# tile of a launch costs the same, so on one stream all 256 CUs reach their HBM-heavy epilogues — and the last, partly filled
# round of workgroups — together while the matrix pipes idle; with two kernel streams in flight one's epilogue bursts and
# kernel tails run under the other's MFMA phases: +2 % on the 64-pair forward (two views through the encoder, two decoder
# branches, two heads).  Per-kernel durations are then not defined (kernels overlap): bench.py's roofline pass and the
# committed profiles run with concurrent(False).
CONCURRENT = os.environ.get("UNICEPTION_AMD_CONCURRENT", "1") != "0"
_branch_warm = set()                          # warm keys without an owner
_branch_warm_by_owner = weakref.WeakKeyDictionary()   # owner module -> its warm keys (dropped with the module: no id() reuse, no growth)
FORK_STREAMS = True      # tests: False keeps every decomposition (two views, two branches, two heads) but runs the halves in stream order


@contextlib.contextmanager
def concurrent(on: bool):
    "Scoped switch of the large-batch two-stream execution (small batches keep theirs)."
    global CONCURRENT
    prev, CONCURRENT = CONCURRENT, bool(on)
    try:
        yield
    finally:
        CONCURRENT = prev


# Two kernel streams while autograd records (round 4).  PyTorch runs every backward node on the stream its forward ran on and orders
# nodes across streams itself, so forking the forward forks the backward too.  Weight gradients are accumulated into the flat gradient
# buffer by plain read-modify-write kernels (the gradient sink), and two streams adding into the same parameter's slice would race:
# sub-graphs with DISJOINT parameters fork as they are (the decoder's two view branches, the two heads); for sub-graphs that SHARE
# parameters (the encoder's two views) the Functions of BOTH halves return their weight gradients to autograd instead
# (autograd._sink_aware): the two contributions meet in one AccumulateGrad node per parameter, and the autograd engine orders the
# producing streams with that node (round 4 kept the sink for the first half: nothing ordered its read-modify-write kernels on the
# main stream against the AccumulateGrad of the forked half).
# uniception_amd.training orders its collectives behind every side stream (GradientBuckets._issue).
TRAIN_CONCURRENT: bool = os.environ.get("UNICEPTION_AMD_TRAIN_CONCURRENT", "1") != "0"


_acc_warn_off = [False]


def _quiet_accumulate_grad_stream_warning() -> None:
    """Forked branches produce gradients on a side stream for parameters whose AccumulateGrad node lives on the main stream: intended
    here (autograd orders the two), so PyTorch's once-per-process notice about it is switched off where the API exists."""
    if not _acc_warn_off[0]:
        _acc_warn_off[0] = True
        fn = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if fn is not None:
            fn(False)


def all_side_streams(device=None):
    "The side streams run_branches has forked to so far (for callers that must order work behind them)."
    return [s for (d, _i), s in _side_streams.items() if device is None or torch.device(d) == torch.device(device)]


def run_branches(fn0, fn1, rows: int, inputs0=(), inputs1=(), warm_key=None, owner=None, disjoint_params: bool = False,
                 shared_params: bool = False):
    """Runs two independent sub-graphs; when they are small (`rows` tokens/pixels each <= BRANCH_TOKENS_MAX: their kernels
    launch fewer workgroups than the chip has CUs) or CONCURRENT is on, and no gradient is recorded, the second one goes to a
    side HIP stream.  `inputs1` are the tensors fn1 reads that were produced on the current stream (they are
    recorded on the side stream for the caching allocator); outputs of fn1 are recorded on the current stream.
    `disjoint_params`: the two sub-graphs own different parameters — they may fork while autograd records (see TRAIN_CONCURRENT).
    `shared_params`: they share parameters (the encoder's two views): they may fork in training too, the forked one's weight gradients
    taking autograd's accumulation instead of the gradient sink.
    `warm_key`: the first call with a given key runs the two sub-graphs one after the other (fn1 on the side stream, fn0 behind
    it) — whatever they cache by shape (position grids, tables) is then built in order; weight-derived caches carry their own
    event (BuiltOn).  `owner`: the module the warm state belongs to (kept in a WeakKeyDictionary — a key built from id(module)
    would outlive the module and be inherited by whatever object reuses the id)."""
    if (torch.is_grad_enabled() and not ((disjoint_params or shared_params) and TRAIN_CONCURRENT and CONCURRENT)) or (rows > BRANCH_TOKENS_MAX and not CONCURRENT) \
            or not inputs1 or not inputs1[0].is_cuda:
        return fn0(), fn1()
    if not FORK_STREAMS:
        return fn0(), fn1()
    # first call with this key: fn1 still runs on the side stream (its allocator pool fills now, not in somebody's timed
    # region) but fn0 waits for it — nothing overlaps while shape-keyed caches are being built
    warm = _branch_warm if owner is None else _branch_warm_by_owner.setdefault(owner, set())
    serialize = warm_key is not None and warm_key not in warm
    if serialize:
        warm.add(warm_key)
    main = torch.cuda.current_stream()
    side = side_stream(inputs1[0].device)
    side.wait_stream(main)
    for t in inputs1:
        t.record_stream(side)
    with torch.cuda.stream(side):
        if torch.is_grad_enabled():
            _quiet_accumulate_grad_stream_warning()
        if shared_params and torch.is_grad_enabled():
            # the forked branch shares parameters with fn0: its weight gradients must not take the read-modify-write gradient sink
            # concurrently with fn0's (autograd._sink_aware) — they go through autograd's own accumulation instead
            from . import autograd
            prev, autograd._sink_fwd_ok[0] = autograd._sink_fwd_ok[0], False
            try:
                out1 = fn1()
            finally:
                autograd._sink_fwd_ok[0] = prev
        else:
            out1 = fn1()
    if serialize:
        main.wait_stream(side)
    if shared_params and torch.is_grad_enabled():
        # fn0 shares parameters with the forked fn1: BOTH leave the gradient sink (a sink kernel of fn0's backward on the main stream
        # and AccumulateGrad of fn1's dW — whose node lives on whichever stream touched the parameter first — would read-modify-write
        # the same slice of the flat buffer with nothing ordering them).  With both returning dW, the two contributions meet in ONE
        # AccumulateGrad node and the autograd engine synchronises the producing streams with it.
        from . import autograd
        prev0, autograd._sink_fwd_ok[0] = autograd._sink_fwd_ok[0], False
        try:
            out0 = fn0()
        finally:
            autograd._sink_fwd_ok[0] = prev0
    else:
        out0 = fn0()
    main.wait_stream(side)
    for t in (out1 if isinstance(out1, (tuple, list)) else (out1,)):
        if torch.is_tensor(t):
            t.record_stream(main)
    return out0, out1

