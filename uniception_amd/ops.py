"""Tensor-level wrappers over the C ABI (include/uc_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function below
hands raw device pointers + sizes/strides to libuc_hip.so.  Nothing in this module computes with
torch ops, and nothing falls back to the CPU: tensors must live on a HIP device.
"""
import ctypes as C
import weakref
from typing import Optional, Tuple

import contextlib

import torch

from . import _lib
from ._lib import (GemmDesc, UC_A_CONV3X3, UC_A_DENSE, UC_ACT_GELU_ERF, UC_ACT_NONE, UC_ACT_RELU, UC_BF16, UC_F16,
                   UC_F32, UC_V_PACKED_T, UC_V_ROWMAJOR, UcHipError)

_DT = {torch.float32: UC_F32, torch.bfloat16: UC_BF16, torch.float16: UC_F16}
ACT = {None: UC_ACT_NONE, "none": UC_ACT_NONE, "gelu": UC_ACT_GELU_ERF, "relu": UC_ACT_RELU}


def _dt(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise UcHipError(f"dtype {t} is not supported by the HIP kernels")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise UcHipError(
                "uniception_amd kernels run on a HIP device only (got a CPU tensor); there is no CPU fallback. "
                "Use the reference implementation or oracle/ for CPU execution."
            )


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def tuning_set(name: str, value: int) -> None:
    """Run-time switchable tuning knobs of libuc_hip.so (uc_tuning_set): "gemm_variant" (-3 automatic, -1 register-staged kernel,
    0..3 / 6 / 7 tile variants of the direct-to-LDS bf16 GEMM), "gemm_stagger" (-1 launcher policy), "attn_role_split", "attn_p64" (persistent 64-queries-per-wave attention forward: 0 never, 1 policy, 2 wherever the shape allows), "attn_bwd64" (64-rows-per-wave attention backward kernels: 0 never, 1 policy, 2 wherever the shape allows; bitwise the same results), "conv_rows" (row-walking 3x3
    conv kernels: 0 never, 1 where they win, 2 the 256-pixel one wherever the shape allows, 3 the eight-wave 512-pixel one wherever the shape allows), "conv_rows_flat" (1: the eight-wave kernel's flat form also takes maps whose rows do not tile 512 pixels; 0: those stay on the implicit-GEMM kernel), "small_m_split" (smallest K for which a dense launch on at most half
    the CUs splits K in two inside the kernel; 0: never — results are then bit-identical across batch sizes).  Every value selects a correct
    kernel; everything else the library reads from the environment, once (csrc/knobs.h)."""
    _lib.check(_lib.load().uc_tuning_set(name.encode(), int(value)), f"uc_tuning_set({name})")
    if name == "small_m_split":
        _small_m_cache[0] = int(value)


def tuning_get(name: str) -> int:
    import ctypes
    v = ctypes.c_int(0)
    _lib.check(_lib.load().uc_tuning_get(name.encode(), ctypes.byref(v)), f"uc_tuning_get({name})")
    return v.value


@contextlib.contextmanager
def tuning(name: str, value: int):
    "Scoped tuning_set (tests, micro-benchmarks: run one tile variant, then restore)."
    prev = tuning_get(name)
    tuning_set(name, value)
    try:
        yield
    finally:
        tuning_set(name, prev)


# --------------------------------------------------------------------------------------------
def rope_2d_(tokens: torch.Tensor, positions: torch.Tensor, base: float, fwd: float) -> None:
    """In-place RoPE-2D on tokens [B,N,H,D] (any strides with stride(3)==1); positions [B,N,2] int64.
    Same contract as curope.rope_2d (curope.cpp:49-69)."""
    if tokens.dim() != 4:
        raise RuntimeError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise RuntimeError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise RuntimeError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise RuntimeError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise RuntimeError("positions.shape[2] must be equal to 2")
    if tokens.is_cuda != positions.is_cuda:
        raise RuntimeError("tokens and positions are not on the same device")
    _need_gpu(tokens, positions)
    if tokens.stride(3) != 1:
        raise RuntimeError("tokens are not contiguous in the last dimension")
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("positions must be a contiguous int64 tensor")
    B, N, H, D = tokens.shape
    lib = _lib.load()
    _lib.check(lib.uc_rope2d(tokens.data_ptr(), positions.data_ptr(), B, N, H, D, tokens.stride(0), tokens.stride(1),
                             tokens.stride(2), float(base), float(fwd), _dt(tokens.dtype), _stream()), "uc_rope2d")


_rope_tables = {}


def rope_table(device, npos: int, base: float, F0: float = 1.0) -> torch.Tensor:
    """[npos,16,2] fp32 cos/sin table for the fused GEMM epilogue (head_dim 64 -> Q=16); cached per device."""
    npos = max(64, (npos + 63) // 64 * 64)
    key = (device.index if device.index is not None else torch.cuda.current_device(), npos, float(base), float(F0))
    t = _rope_tables.get(key)
    if t is None:
        t = torch.empty(npos, 16, 2, dtype=torch.float32, device=device)
        _lib.check(_lib.load().uc_rope_table(t.data_ptr(), npos, 16, float(base), float(F0), _stream()), "uc_rope_table")
        t.uc_rope_base, t.uc_rope_f0 = float(base), float(F0)   # travel with the table into uc_gemm's descriptor
        from .engine import BuiltOn
        t.uc_built = BuiltOn()
        _rope_tables[key] = t
    else:
        t.uc_built.sync()       # a consumer on another stream waits for the stream that filled the table
    return t


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, out_dtype: torch.dtype,
              twin: bool = False) -> torch.Tensor:
    """x [..., C] contiguous (fp32|bf16) -> same shape in out_dtype.  twin: a bf16 copy of the result is written in the same
    pass and rides on the returned tensor as ``y.uc_twin`` (for consumers that take bf16 operands)."""
    _need_gpu(x, weight, bias)
    assert x.is_contiguous() and weight.dtype == torch.float32 and bias.dtype == torch.float32
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    tw = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if twin else None
    _lib.check(_lib.load().uc_layernorm_twin(x.data_ptr(), _dt(x.dtype), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                             _dt(out_dtype), _p(tw), rows, Cn, float(eps), _stream()), "uc_layernorm")
    if twin:
        y.uc_twin = tw
    return y


# ---------------------------------------------------------------------------------------------
# Hand-over buffers of uc_gemm's small-M path (uc_gemm_desc.fuse_ws, ABI 11): the library allocates nothing, the HOST does.
# A buffer is uc_gemm_fuse_ws_bytes() (8 MiB) of UNCACHED device memory — the two halves of a split tile may run on different
# XCDs, whose L2s are not coherent; PyTorch's allocator has no such flavour, so the HIP runtime is called directly (ctypes on the
# libamdhip64 the process already has) — zero-filled once, used by one launch at a time: one per (device, stream), and one per
# (device, stream, graph) for launches recorded into a hipGraph (its replays may overlap eager launches on the same stream handle).
# Allocation is not allowed while a stream is capturing: captures draw from a reserve filled beforehand (fuse_ws_reserve, called by
# graphs.GraphedTwoView); an empty reserve means the launch runs unsplit (uc_gemm's documented behaviour without a buffer).
# ---------------------------------------------------------------------------------------------
_HIP_MALLOC_UNCACHED = 0x3      # hipDeviceMallocUncached
_hip_rt = None
_fuse_ws_sets = {}              # (device, stream handle, capture token | 0) -> device pointer
_fuse_ws_free = {}              # device -> [device pointers not handed out]
_capture_token = 0              # set by capture_scope(): identifies the graph being recorded


def _hip_runtime():
    "The HIP runtime library this process has loaded (PyTorch's): hipExtMallocWithFlags / hipMemset / hipFree."
    global _hip_rt
    if _hip_rt is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        _hip_rt = C.CDLL(path or "libamdhip64.so")
        _hip_rt.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        _hip_rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip_rt.hipFree.argtypes = [C.c_void_p]
    return _hip_rt


def _fuse_ws_new(device_index: int) -> Optional[int]:
    hip = _hip_runtime()
    nbytes = int(_lib.load().uc_gemm_fuse_ws_bytes())
    ptr = C.c_void_p()
    with torch.cuda.device(device_index):
        if hip.hipExtMallocWithFlags(C.byref(ptr), nbytes, _HIP_MALLOC_UNCACHED) != 0 or not ptr.value:
            return None
        if hip.hipMemset(ptr, 0, nbytes) != 0:        # (synchronous; the flag words must start at zero)
            hip.hipFree(ptr)
            return None
    return ptr.value


def fuse_ws_reserve(n: int = 4, device=None) -> None:
    """Make sure n hand-over buffers are ready to be handed to streams (call OUTSIDE a stream capture, e.g. before recording a graph
    whose small-M launches should split)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    free = _fuse_ws_free.setdefault(dev, [])
    while len(free) < n:
        ptr = _fuse_ws_new(dev)
        if ptr is None:
            break
        free.append(ptr)


def fuse_ws_release(token: int) -> None:
    "Return the buffers of a destroyed graph (capture token) to the reserve."
    for key in [k for k in _fuse_ws_sets if k[2] == token and token != 0]:
        _fuse_ws_free.setdefault(key[0], []).append(_fuse_ws_sets.pop(key))


def fuse_ws_free_all() -> None:
    "hipFree every hand-over buffer (only when no launch that uses one can still be in flight: synchronizes first)."
    torch.cuda.synchronize()
    hip = _hip_runtime()
    for ptr in list(_fuse_ws_sets.values()) + [p for v in _fuse_ws_free.values() for p in v]:
        hip.hipFree(C.c_void_p(ptr))
    _fuse_ws_sets.clear()
    _fuse_ws_free.clear()


@contextlib.contextmanager
def capture_scope(token: int):
    "Launches recorded inside belong to the graph identified by `token` (their hand-over buffers live as long as it does)."
    global _capture_token
    prev = _capture_token
    _capture_token = token
    try:
        yield
    finally:
        _capture_token = prev


def _fuse_ws_for_launch(M: int, N: int, K: int) -> Optional[int]:
    """The buffer a dense bf16 launch of this shape may use for the in-kernel K split (None: it runs unsplit).  Only shapes uc_gemm
    can split get one (at most 128 tiles of 128 x 128 on at most half the CUs, K at least the small_m_split knob)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > 128 or K % 128 != 0:
        return None
    smk = _small_m_k()
    if smk <= 0 or K < smk or 2 * tiles > _num_cus():
        return None
    dev = torch.cuda.current_device()
    capturing = torch.cuda.is_current_stream_capturing()
    # a captured launch belongs to its GRAPH, not to the stream it was recorded on (PyTorch records every graph on a capture stream, and
    # two graphs may be replayed at the same time on different streams): keyed by the capture_scope token, else by the capture's own id
    token = 0
    if capturing:
        if not _capture_token:
            # a capture outside ops.capture_scope has no owner to hand the buffer back when the graph dies (every such graph would
            # keep 8 MiB for good, and once the reserve is drained later captures would silently run unsplit): it runs unsplit,
            # always — wrap captures in ops.capture_scope(token) + fuse_ws_release(token) (uniception_amd.graphs does)
            return None
        token = _capture_token
    key = (dev, _stream(), token)
    ptr = _fuse_ws_sets.get(key)
    if ptr is None:
        free = _fuse_ws_free.setdefault(dev, [])
        if not capturing and len(free) < _FUSE_WS_SPARE + 1:
            fuse_ws_reserve(_FUSE_WS_SPARE + 1, dev)      # this one + spares for the streams a later capture brings along
        if not free:
            return None
        ptr = free.pop()
        _fuse_ws_sets[key] = ptr
    return ptr


_FUSE_WS_SPARE = 3


_small_m_cache = [None]
_num_cus_cache = {}


def _small_m_k() -> int:
    if _small_m_cache[0] is None:
        _small_m_cache[0] = tuning_get("small_m_split")
    return _small_m_cache[0]


def _num_cus() -> int:
    dev = torch.cuda.current_device()
    if dev not in _num_cus_cache:
        _num_cus_cache[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    return _num_cus_cache[dev]


# ---------------------------------------------------------------------------------------------
# fp16 range guard (TF32-class heads): every launch that can push a value out of fp16's range (GEMM / conv epilogues with fp16
# outputs, conversions into fp16) saturates at +-65504 and ORs 1 into this per-device int32 — the engine polls it (engine.
# head_range_exceeded) and falls back to a wider head format.
# ---------------------------------------------------------------------------------------------
_f16_sat_flags = {}


def f16_sat_flag() -> torch.Tensor:
    dev = torch.cuda.current_device()
    t = _f16_sat_flags.get(dev)
    if t is None:
        t = _f16_sat_flags[dev] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", dev))
    return t


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
         residual: Optional[torch.Tensor] = None, residual2: Optional[torch.Tensor] = None,
         out_dtype: Optional[torch.dtype] = None,
         out: Optional[torch.Tensor] = None, relu_a: bool = False,
         rope: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None,
         vt: Optional[Tuple[int, torch.Tensor, int]] = None,
         conv: Optional[Tuple[int, int, int, int, int]] = None, preact_out: Optional[torch.Tensor] = None,
         split_k: int = 1, dact: Optional[Tuple[torch.Tensor, str]] = None,
         ln: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, emit_ln: bool = False,
         tail: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None) -> torch.Tensor:
    """C[M,N] = epilogue(A . W^T).

    a    : dense [M,K] (row stride a.stride(0), unit column stride) or, with conv=(B,H,W,Cin,stride), an NHWC image.
    w    : [N,K] contiguous, same dtype as a (fp32 -> exact fp32 kernel, bf16 -> MFMA kernel).
    rope : (positions [M,2] int64, table, rope_cols) -> fused RoPE-2D on the first rope_cols columns (bf16 only).
    vt   : (vt_col0, vt_out [B,H,64,Npad] bf16, ntok) -> V columns written in the packed VT layout (bf16 only).
    ln   : (stats [M,2] fp32 (mean, rstd), colsum [N] fp32) -> folded LayerNorm: a holds the RAW rows, w has gamma folded in,
           the epilogue computes rstd * (acc - mean * colsum) + bias (bf16 output only).
    tail : (w4 [4,N] fp32, b4 [4] fp32 | None) with N == 128 on a 3x3 convolution — the [M,128] result is not stored; returns
           fp32 [M,4] = b4 + act(acc + bias) . w4^T (the DPT regressor's conv3x3 -> ReLU -> conv1x1(128 -> 4) in one kernel).
    emit_ln : also write per-row 64-column-block statistics and, for an fp32 output, a bf16 twin of it (a bf16 output — the bf16
           residual stream — is its own twin); they ride on the returned tensor as ``out.uc_ln`` (an LnSide: twin, partial,
           finalized-stats cache) for the consumer's ``ln=``.
    """
    _need_gpu(a, w, bias, residual)
    assert a.dtype == w.dtype and w.is_contiguous() and w.dim() == 2
    cd = _dt(a.dtype)
    N, K = w.shape
    if (cd == UC_F32 and fp32_matmul_hook() == "bf16x3" and rope is None and vt is None and preact_out is None
            and (split_k <= 1 or (conv is None and bias is None and act in (None, "none") and residual is None and not relu_a))
            and dact is None and ln is None and not emit_ln and (K // (9 if conv is not None else 1)) % 8 == 0 and a.is_contiguous()):
        # fp32-class product on the bf16 matrix pipe: [hi | hi | lo] rows against [Wh | Wl | Wh] weights (uc_split_bf16x3)
        # (split_k > 1, round 6: the weight gradients of the fp32-class heads — a handful of output tiles, millions of reduction steps —
        #  return fp32 slabs [split_k, M, N] like the bf16 path's)
        a3 = split_bf16x3(a, relu=relu_a)
        conv3 = None if conv is None else (conv[0], conv[1], conv[2], 3 * conv[3], conv[4])
        return gemm(a3, split_weight_bf16x3(w, 9 if conv is not None else 1), bias, act=act, residual=residual, residual2=residual2,
                    out_dtype=out_dtype or torch.float32, out=out, conv=conv3, tail=tail, split_k=split_k)
    d = GemmDesc()
    d.compute_dtype = cd
    d.relu_a = 1 if relu_a else 0
    d.A, d.W = a.data_ptr(), w.data_ptr()
    if conv is None:
        assert a.dim() == 2 and a.stride(1) == 1 and a.shape[1] == K
        M = a.shape[0]
        d.a_mode, d.lda = UC_A_DENSE, a.stride(0)
    else:
        Bc, Hc, Wc, Cin, s = conv
        assert a.is_contiguous() and a.numel() == Bc * Hc * Wc * Cin and K == 9 * Cin
        Ho, Wo = (Hc - 1) // s + 1, (Wc - 1) // s + 1
        M = Bc * Ho * Wo
        d.a_mode, d.lda = UC_A_CONV3X3, 0
        d.conv_B, d.conv_H, d.conv_W, d.conv_Cin, d.conv_stride, d.conv_Ho, d.conv_Wo = Bc, Hc, Wc, Cin, s, Ho, Wo
    d.M, d.N, d.K = M, N, K
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N
    d.bias = _p(bias)
    d.act = ACT[act]
    if residual is not None:
        assert residual.dim() == 2 and residual.stride(1) == 1 and residual.shape == (M, N)
        d.residual, d.res_dtype, d.ldr = residual.data_ptr(), _dt(residual.dtype), residual.stride(0)
        if residual2 is not None:
            assert residual2.shape == residual.shape and residual2.stride() == residual.stride() and residual2.dtype == residual.dtype
            d.residual2 = residual2.data_ptr()
    n_out = N
    d.vt_col0 = -1
    if vt is not None:
        vt_col0, vt_out, ntok = vt
        assert vt_out.dtype == torch.bfloat16 and vt_out.is_contiguous()
        d.vt_col0, d.vt_out, d.vt_ntok, d.vt_npad = vt_col0, vt_out.data_ptr(), ntok, vt_out.shape[-1]
        n_out = vt_col0
    if rope is not None:
        pos, table, rope_cols = rope
        assert pos.dtype == torch.int64 and pos.is_contiguous() and pos.numel() == 2 * M
        d.rope_cols, d.rope_pos, d.rope_table, d.rope_npos = rope_cols, pos.data_ptr(), table.data_ptr(), table.shape[0]
        d.rope_base, d.rope_f0 = table.uc_rope_base, table.uc_rope_f0
    if tail is not None:
        w4, b4 = tail
        assert conv is not None and N == 128 and out is None and residual is None and rope is None and vt is None and split_k <= 1
        assert w4.dtype == torch.float32 and w4.is_contiguous() and w4.shape == (4, N)
        assert b4 is None or (b4.dtype == torch.float32 and b4.is_contiguous() and b4.numel() == 4)
        out4 = torch.empty((M, 4), dtype=torch.float32, device=a.device)
        d.tail_w, d.tail_b, d.tail_out = w4.data_ptr(), _p(b4), out4.data_ptr()
        d.C, d.out_dtype, d.ldc = None, (UC_F16 if cd == UC_F16 else UC_BF16), N      # (nothing is stored: the storage dtype of the operands)
        if cd == UC_F16:
            d.sat_flag = _p(f16_sat_flag())
        _lib.check(_lib.load().uc_gemm(C.byref(d), _stream()), "uc_gemm")
        return out4
    if out is None and split_k > 1:
        out = torch.empty((split_k, M, N), dtype=torch.float32, device=a.device)
        d.C, d.out_dtype, d.ldc = out.data_ptr(), _dt(out.dtype), N
    elif out is None:
        out = torch.empty((M, n_out), dtype=out_dtype or a.dtype, device=a.device)
    elif split_k > 1:   # workspace of split_k [M,N] fp32 slabs
        assert out.shape == (split_k, M, N) and out.is_contiguous() and out.dtype == torch.float32
        d.C, d.out_dtype, d.ldc = out.data_ptr(), _dt(out.dtype), N
    else:
        assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] == M and out.shape[1] >= n_out
    if split_k <= 1 or out.dim() == 2:
        d.C, d.out_dtype, d.ldc = out.data_ptr(), _dt(out.dtype), out.stride(0)
    if preact_out is not None:
        assert preact_out.shape == out.shape and preact_out.stride() == out.stride() and preact_out.dtype == out.dtype
        d.preact_out = preact_out.data_ptr()
    d.split_k = int(split_k)
    if dact is not None:   # fused activation backward: out *= act'(u)
        u, act_name = dact
        assert u.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and u.shape == out.shape and u.stride() == out.stride()
        d.dact_u, d.dact_act = u.data_ptr(), ACT[act_name]
    if ln is not None:
        st, cs = ln
        assert cs.dtype == torch.float32 and cs.is_contiguous() and cs.numel() == N
        if isinstance(st, LnPartial):     # block partials merged in this GEMM's epilogue (small batches: no finalize launch)
            pt = st.partial
            assert pt.dtype == torch.float32 and pt.is_contiguous() and pt.shape == (K // 64, M, 2)     # block-major
            d.ln_stats, d.ln_nblk, d.ln_eps = pt.data_ptr(), K // 64, float(st.eps)
        else:
            assert st.dtype == torch.float32 and st.is_contiguous() and st.shape == (M, 2)
            d.ln_stats = st.data_ptr()
        d.ln_colsum = cs.data_ptr()
    side = None
    if emit_ln:
        assert out.dim() == 2 and N % 64 == 0 and vt is None
        partial = torch.empty((N // 64, M, 2), dtype=torch.float32, device=a.device)     # block-major: [N/64][M] (sum, M2) pairs
        if out.dtype == torch.bfloat16:      # bf16 residual stream: the stored rows are their own twin
            assert out.is_contiguous() and (residual is None or residual.dtype == torch.bfloat16)
            side = LnSide(out, partial)
            d.stats_out = partial.data_ptr()
        else:
            assert out.dtype == torch.float32
            side = LnSide(torch.empty((M, N), dtype=torch.bfloat16, device=a.device), partial)
            d.twin_out, d.ldt, d.stats_out = side.twin.data_ptr(), N, side.partial.data_ptr()
    if cd == UC_F16:
        d.sat_flag = _p(f16_sat_flag())
    if cd != UC_F32 and split_k <= 1:      # small-M path: hand uc_gemm a hand-over buffer when this launch can split K inside the kernel
        ws = _fuse_ws_for_launch(M, N, K)
        if ws is not None:
            d.fuse_ws = ws
    _lib.check(_lib.load().uc_gemm(C.byref(d), _stream()), "uc_gemm")
    if side is not None:
        out.uc_ln = side
    return out


# fp32 GEMMs: "exact" (fp32 FMA chain, the verification kernels) or "bf16x3" (split operands on the bf16 MFMA pipe); the
# engine installs a hook that answers per call (precision context, head policy)
fp32_matmul_hook = lambda: "exact"   # noqa: E731


def split_bf16x3(x: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """fp32 [..., C] contiguous -> bf16 [..., 3C] = [hi | hi | lo] (uc_split_bf16x3)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 8 == 0
    Cn = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (3 * Cn,), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().uc_split_bf16x3(x.data_ptr(), out.data_ptr(), x.numel() // Cn, Cn, 1 if relu else 0, _stream()),
               "uc_split_bf16x3")
    return out


_w3_cache = {}   # id(weight tensor) -> (weakref, stamp, split copy); entries die with their tensor


def split_weight_bf16x3(w: torch.Tensor, taps: int = 1) -> torch.Tensor:
    """fp32 weight [N, taps*Cin] -> bf16 [N, taps*3*Cin] with every tap's block laid out [Wh | Wl | Wh]; cached per tensor."""
    key = id(w)
    hit = _w3_cache.get(key)
    stamp = (w._version, w.data_ptr(), taps)
    if hit is not None and hit[0]() is w and hit[1] == stamp:
        hit[3].sync()
        return hit[2]
    with torch.no_grad():
        N = w.shape[0]
        w32 = w.detach().float().view(N, taps, -1)
        hi = w32.bfloat16()
        lo = (w32 - hi.float()).bfloat16()
        w3 = torch.cat([hi, lo, hi], dim=2).reshape(N, -1).contiguous()
    from .engine import BuiltOn
    _w3_cache[key] = (weakref.ref(w, lambda _r, k=key: _w3_cache.pop(k, None)), stamp, w3, BuiltOn())
    return w3


class LnPartial:
    "ln= argument of gemm(): the producer's per-block row statistics, merged into (mean, rstd) inside the consumer's epilogue."
    __slots__ = ("partial", "eps")

    def __init__(self, partial, eps):
        self.partial, self.eps = partial, eps


# rows up to which the consumer GEMMs merge the LayerNorm block statistics themselves (uc_gemm_desc.ln_nblk) instead of reading the
# output of a uc_ln_stats_finalize launch.  Measured (512x512 pairs, ms per batch at 1 / 2 / 4 / 8 pairs): eager 12.8 / 13.7 / 15.0 /
# 25.1 -> 11.6 / 10.8 / 16.5 / 27.0 — the eager forward of 1-2 pairs is HOST-bound (~640 launches through ctypes), so 120 fewer
# launches are worth 1-3 ms there; replayed from a hipGraph (GPU-bound) 8.2 / 9.8 / 14.8 / 24.5 -> 8.4 / 10.3 / 16.3 / 26.9: every
# column tile of a row panel repeats the merge at the head of its epilogue, which costs more than the 4-us launches it saves.  Hence:
# up to 2048 rows (two pairs per stream), and never while a graph is being captured.
LN_MERGE_IN_EPILOGUE_MAX_ROWS = int(__import__("os").environ.get("UNICEPTION_AMD_LN_MERGE_ROWS", "2048"))


class LnSide:
    """What a producer GEMM leaves next to its fp32 output rows for the folded LayerNorm of the consumer: a bf16 copy of the
    rows and their per-block statistics; (mean, rstd) per row are finalized on first use, per eps."""
    __slots__ = ("twin", "partial", "_stats")

    def __init__(self, twin, partial):
        self.twin, self.partial, self._stats = twin, partial, {}

    def stats_arg(self, eps: float):
        """What the consumer GEMM gets as its LayerNorm statistics: the block partials themselves for small batches (merged in its
        epilogue: same bits, no launch), the finalized (mean, rstd) rows otherwise."""
        if (self.partial.shape[1] <= LN_MERGE_IN_EPILOGUE_MAX_ROWS and eps not in self._stats
                and not torch.cuda.is_current_stream_capturing()):
            return LnPartial(self.partial, eps)
        return self.stats(eps)

    def stats(self, eps: float) -> torch.Tensor:
        st = self._stats.get(eps)
        if st is None:
            nblk, M, _ = self.partial.shape
            st = torch.empty((M, 2), dtype=torch.float32, device=self.partial.device)
            _lib.check(_lib.load().uc_ln_stats_finalize(self.partial.data_ptr(), M, nblk, float(eps), st.data_ptr(), _stream()),
                       "uc_ln_stats_finalize")
            self._stats[eps] = st
        return st


def vt_buffer(B: int, H: int, ntok: int, device) -> torch.Tensor:
    npad = (ntok + 63) // 64 * 64
    vt = torch.empty((B, H, 64, npad), dtype=torch.bfloat16, device=device)
    if npad != ntok:
        # pad key positions must be finite (the attention kernel multiplies them by P = 0).  Keys are permuted inside every
        # group of 16 positions, so the pads of a ragged group are interleaved with its keys: clear the whole last group(s)
        vt[..., (ntok // 16) * 16:].zero_()
    return vt


def vt_pack(v: torch.Tensor) -> torch.Tensor:
    """v: [B,N,H,D] bf16 view (stride(3)==1) -> packed VT [B,H,D,Npad]."""
    _need_gpu(v)
    B, N, H, D = v.shape
    assert v.dtype == torch.bfloat16 and v.stride(3) == 1
    npad = (N + 63) // 64 * 64
    out = torch.empty((B, H, D, npad), dtype=torch.bfloat16, device=v.device)
    _lib.check(_lib.load().uc_vt_pack(v.data_ptr(), out.data_ptr(), B, H, N, D, v.stride(0), v.stride(1), v.stride(2),
                                      _stream()), "uc_vt_pack")
    return out


def vt_pack_fp8(v: torch.Tensor) -> torch.Tensor:
    """v: [B,N,H,64] bf16 view -> e4m3 V^T [B,H,64,Npad] (uint8) in the k-slot order of the fp8 attention kernel."""
    _need_gpu(v)
    B, N, H, D = v.shape
    assert v.dtype == torch.bfloat16 and v.stride(3) == 1 and D == 64
    npad = (N + 63) // 64 * 64
    out = torch.empty((B, H, D, npad), dtype=torch.uint8, device=v.device)
    _lib.check(_lib.load().uc_vt_pack_fp8(v.data_ptr(), out.data_ptr(), B, H, N, D, v.stride(0), v.stride(1), v.stride(2),
                                          _stream()), "uc_vt_pack_fp8")
    return out


def k_pack_fp8(k: torch.Tensor) -> torch.Tensor:
    """k: [B,N,H,64] bf16 view -> e4m3 rows [B,H,Npad,64] (uint8), zero padded: the K operand of the DMA-staged fp8 kernel."""
    _need_gpu(k)
    B, N, H, D = k.shape
    assert k.dtype == torch.bfloat16 and k.stride(3) == 1 and D == 64
    npad = (N + 63) // 64 * 64
    out = torch.empty((B, H, npad, D), dtype=torch.uint8, device=k.device)
    _lib.check(_lib.load().uc_k_pack_fp8(k.data_ptr(), out.data_ptr(), B, H, N, k.stride(0), k.stride(1), k.stride(2), _stream()),
               "uc_k_pack_fp8")
    return out


def attention_fp8(q: torch.Tensor, k: torch.Tensor, vt8: torch.Tensor, scale: float, prepack_k: Optional[bool] = None) -> torch.Tensor:
    """q [B,Nq,H,64], k [B,Nk,H,64] bf16 views; vt8 from vt_pack_fp8.  Returns O [B,Nq,H,64] bf16.
    prepack_k (default: whenever Nk % 64 == 0): convert K once (k_pack_fp8) and run the LDS-DMA kernel
    (uc_attention_fwd_fp8_k8); otherwise the kernel converts K tiles while staging them (uc_attention_fwd_fp8)."""
    _need_gpu(q, k, vt8)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    assert q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and D == 64 and q.stride(3) == 1 and k.stride(3) == 1
    assert vt8.dtype == torch.uint8 and vt8.is_contiguous() and vt8.shape == (B, H, 64, (Nk + 63) // 64 * 64)
    out = torch.empty((B, Nq, H, D), dtype=torch.bfloat16, device=q.device)
    if prepack_k is None:
        prepack_k = Nk % 64 == 0
    if prepack_k:
        k8 = k_pack_fp8(k)
        _lib.check(_lib.load().uc_attention_fwd_fp8_k8(
            q.data_ptr(), k8.data_ptr(), vt8.data_ptr(), out.data_ptr(), B, H, Nq, Nk, q.stride(0), q.stride(1), q.stride(2),
            out.stride(0), out.stride(1), out.stride(2), float(scale), _stream()), "uc_attention_fwd_fp8_k8")
        return out
    _lib.check(_lib.load().uc_attention_fwd_fp8(
        q.data_ptr(), k.data_ptr(), vt8.data_ptr(), out.data_ptr(), B, H, Nq, Nk, q.stride(0), q.stride(1), q.stride(2),
        k.stride(0), k.stride(1), k.stride(2), out.stride(0), out.stride(1), out.stride(2), float(scale), _stream()),
        "uc_attention_fwd_fp8")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, v_packed: bool = False,
              out: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None, dropout=None) -> torch.Tensor:
    """q [B,Nq,H,D], k [B,Nk,H,D] strided views (stride(3)==1); v same, or packed VT [B,H,D,Npad] when v_packed.
    Returns O [B,Nq,H,D] contiguous (== [B,Nq,H*D]).
    dropout = (p, seed): dropout of the attention probabilities inside the kernel (uc_attention_fwd_drop: a counter-based keep
    function of (seed, batch, head, query, key); attention_drop_mask materialises it; the backward takes the same pair)."""
    _need_gpu(q, k, v)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    assert q.stride(3) == 1 and k.stride(3) == 1 and q.dtype == k.dtype == v.dtype
    if out is None:
        out = torch.empty((B, Nq, H, D), dtype=q.dtype, device=q.device)
    if v_packed:
        vs = (0, 0, 0)
        assert v.is_contiguous() and v.shape[-1] == (Nk + 63) // 64 * 64
    else:
        assert v.stride(3) == 1
        vs = (v.stride(0), v.stride(1), v.stride(2))
    if dropout is not None and float(dropout[0]) > 0.0:
        _lib.check(_lib.load().uc_attention_fwd_drop(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _dt(q.dtype), UC_V_PACKED_T if v_packed else UC_V_ROWMAJOR,
            B, H, Nq, Nk, D, q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2), vs[0], vs[1], vs[2],
            out.stride(0), out.stride(1), out.stride(2), float(scale), _p(lse), float(dropout[0]), int(dropout[1]), _stream()),
            "uc_attention_fwd_drop")
        return out
    _lib.check(_lib.load().uc_attention_fwd(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _dt(q.dtype), UC_V_PACKED_T if v_packed else UC_V_ROWMAJOR,
        B, H, Nq, Nk, D, q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2), vs[0], vs[1], vs[2],
        out.stride(0), out.stride(1), out.stride(2), float(scale), _p(lse), _stream()), "uc_attention_fwd")
    return out


def attention_x3(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None,
                 lse: Optional[torch.Tensor] = None, rope: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """fp32-class attention on the matrix pipe (uc_attention_fwd_x3: split bf16 operands, three MFMA products per product, fp32
    softmax).  q [B,Nq,H,64], k / v [B,Nk,H,64]: fp32 strided views with stride(3) == 1.  Returns fp32 O [B,Nq,H,64] contiguous.
    rope = (q positions [B*Nq, 2] int64, k positions [B*Nk, 2] int64, table from rope_table): q and k are rotated by RoPE-2D inside the
    operand split (the arithmetic of rope_2d_), instead of in a pass of their own."""
    _need_gpu(q, k, v)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    assert D == 64 and q.dtype == k.dtype == v.dtype == torch.float32
    assert q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1 and k.shape == v.shape
    if out is None:
        out = torch.empty((B, Nq, H, D), dtype=torch.float32, device=q.device)
    qp = kp = tab = None
    npos = 0
    if rope is not None:
        qp, kp, tab = rope
        assert qp.dtype == kp.dtype == torch.int64 and qp.is_contiguous() and kp.is_contiguous() and qp.numel() == B * Nq * 2 and kp.numel() == B * Nk * 2
        assert tab.dtype == torch.float32 and tab.is_contiguous() and tab.numel() % 32 == 0
        npos = tab.numel() // 32
    lib = _lib.load()
    ws = torch.empty(int(lib.uc_attention_x3_workspace_bytes(B, H, Nq, Nk)), dtype=torch.uint8, device=q.device)
    _lib.check(lib.uc_attention_fwd_x3(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, Nq, Nk,
                                       q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                       v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2), float(scale),
                                       _p(lse), _p(qp), _p(kp), _p(tab), npos, _stream()), "uc_attention_fwd_x3")
    return out


def patch_gather(img: torch.Tensor, P: int, out_dtype: torch.dtype) -> torch.Tensor:
    _need_gpu(img)
    assert img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4
    B, Cin, H, W = img.shape
    cols = torch.empty((B * (H // P) * (W // P), Cin * P * P), dtype=out_dtype, device=img.device)
    _lib.check(_lib.load().uc_patch_gather(img.data_ptr(), cols.data_ptr(), _dt(out_dtype), B, Cin, H, W, P, _stream()),
               "uc_patch_gather")
    return cols


def nchw_to_nhwc(x: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """x [B,C,H,W] contiguous -> [B,H,W,C] contiguous in out_dtype."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 4
    B, Cn, H, W = x.shape
    y = torch.empty((B, H, W, Cn), dtype=out_dtype, device=x.device)
    _lib.check(_lib.load().uc_nchw_to_nhwc(x.data_ptr(), _dt(x.dtype), y.data_ptr(), _dt(out_dtype), B, Cn, H, W, _stream()),
               "uc_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """x [B,H,W,C] contiguous -> [B,C,H,W] contiguous in out_dtype."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 4
    B, H, W, Cn = x.shape
    y = torch.empty((B, Cn, H, W), dtype=out_dtype, device=x.device)
    _lib.check(_lib.load().uc_nhwc_to_nchw(x.data_ptr(), _dt(x.dtype), y.data_ptr(), _dt(out_dtype), B, Cn, H, W, _stream()),
               "uc_nhwc_to_nchw")
    return y


def convert(x: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    _need_gpu(x)
    assert x.is_contiguous()
    if x.dtype == out_dtype:
        return x
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_lib.load().uc_convert(x.data_ptr(), _dt(x.dtype), y.data_ptr(), _dt(out_dtype), x.numel(),
                                      _p(f16_sat_flag()) if out_dtype == torch.float16 else None, _stream()), "uc_convert")
    return y


def bilinear_nhwc(x: torch.Tensor, Ho: int, Wo: int, crop: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """align_corners=True bilinear resize of NHWC x to (Ho,Wo), optionally keeping only the top-left crop."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 4
    B, Hi, Wi, Cn = x.shape
    ch, cw = crop if crop is not None else (Ho, Wo)
    y = torch.empty((B, ch, cw, Cn), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().uc_bilinear_nhwc(x.data_ptr(), y.data_ptr(), _dt(x.dtype), B, Hi, Wi, Cn, Ho, Wo, ch, cw, _stream()),
               "uc_bilinear_nhwc")
    return y


def convt_scatter(x: torch.Tensor, B: int, h: int, w: int, k: int, Cout: int) -> torch.Tensor:
    _need_gpu(x)
    assert x.is_contiguous() and x.shape == (B * h * w, k * k * Cout)
    y = torch.empty((B, k * h, k * w, Cout), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().uc_convt_scatter(x.data_ptr(), y.data_ptr(), _dt(x.dtype), B, h, w, k, Cout, _stream()),
               "uc_convt_scatter")
    return y


def pixel_shuffle(x: torch.Tensor, B: int, h: int, w: int, P: int, Cout: int) -> torch.Tensor:
    _need_gpu(x)
    assert x.is_contiguous() and x.shape == (B * h * w, Cout * P * P)
    y = torch.empty((B, Cout, P * h, P * w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().uc_pixel_shuffle(x.data_ptr(), _dt(x.dtype), y.data_ptr(), B, h, w, P, Cout, _stream()),
               "uc_pixel_shuffle")
    return y


def pointmap_adaptor(x: torch.Tensor, conf_vmin: float, conf_vmax: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """x: fp32 4-channel map given as a BCHW-shaped tensor (contiguous NCHW or channels-last strides).
    Returns (pts [B,H,W,3], conf [B,H,W,1]) fp32 contiguous."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 4
    B, _, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if sh != W * sw:
        raise UcHipError("pointmap_adaptor: rows of the 4-channel map must be densely packed")
    pts = torch.empty((B, H, W, 3), dtype=torch.float32, device=x.device)
    conf = torch.empty((B, H, W, 1), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().uc_pointmap_adaptor(x.data_ptr(), sb, sc, sw, pts.data_ptr(), conf.data_ptr(), B, H, W,
                                               float(conf_vmin), float(conf_vmax), _stream()), "uc_pointmap_adaptor")
    return pts, conf


def adaptor_program(x: torch.Tensor, segs, cout: int) -> torch.Tensor:
    """x: fp32 BCHW-shaped map (contiguous NCHW or a channels-last view); segs: list of _lib.AdaptorSeg -> fp32 NHWC [B,H,W,cout]."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.dim() == 4
    B, _, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if sh != W * sw:
        raise UcHipError("adaptor_program: rows of the input map must be densely packed")
    out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    arr = (_lib.AdaptorSeg * len(segs))(*segs)
    _lib.check(_lib.load().uc_adaptor_program(x.data_ptr(), sb, sc, sw, out.data_ptr(), B, H, W, cout, arr, len(segs), _stream()),
               "uc_adaptor_program")
    return out


def adaptor_program_bwd(x: torch.Tensor, dout: torch.Tensor, segs) -> torch.Tensor:
    """Gradient of adaptor_program with respect to x: x as in the forward, dout fp32 NHWC [B,H,W,cout] -> dx with x's strides."""
    _need_gpu(x, dout)
    assert x.dtype == torch.float32 and x.dim() == 4 and dout.dtype == torch.float32 and dout.is_contiguous() and dout.dim() == 4
    B, C, H, W = x.shape
    assert dout.shape[:3] == (B, H, W)
    sb, sc, sh, sw = x.stride()
    if sh != W * sw:
        raise UcHipError("adaptor_program_bwd: rows of the input map must be densely packed")
    dx = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device)
    arr = (_lib.AdaptorSeg * len(segs))(*segs)
    _lib.check(_lib.load().uc_adaptor_program_bwd(x.data_ptr(), sb, sc, sw, dout.data_ptr(), dx.data_ptr(), B, H, W, C, dout.shape[3],
                                                  arr, len(segs), _stream()), "uc_adaptor_program_bwd")
    return dx


def conv1x1_to4(feat: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """feat NHWC [B,H,W,Cin]; w fp32 [4,Cin]; b fp32 [4] -> fp32 NHWC [B,H,W,4]."""
    _need_gpu(feat, w, b)
    assert feat.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous() and w.shape[0] == 4
    B, H, W, Cin = feat.shape
    out = torch.empty((B, H, W, 4), dtype=torch.float32, device=feat.device)
    _lib.check(_lib.load().uc_conv1x1_to4(feat.data_ptr(), _dt(feat.dtype), w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                          B * H * W, Cin, _stream()), "uc_conv1x1_to4")
    return out


def add_view_pe_(x: torch.Tensor, pe: torch.Tensor, T: int) -> torch.Tensor:
    """x fp32 [B, L, C] += pe[v] on the T tokens of each of the V = pe.shape[0] views (in place; rows past V*T untouched)."""
    _need_gpu(x, pe)
    assert x.dtype == torch.float32 and pe.dtype == torch.float32 and x.is_contiguous() and pe.is_contiguous() and x.dim() == 3
    B, L, Cn = x.shape
    _lib.check(_lib.load().uc_add_view_pe(x.data_ptr(), pe.data_ptr(), B, L, T, pe.shape[0], Cn, _stream()), "uc_add_view_pe")
    return x


def assemble_tokens(tok: torch.Tensor, cls: torch.Tensor, reg: Optional[torch.Tensor], pos: torch.Tensor) -> torch.Tensor:
    """tok fp32 [B,hw,D], cls [D], reg [R,D]|None, pos [1+hw,D] -> [B, 1+R+hw, D] = [cls+pos0 | reg | tok+pos]."""
    _need_gpu(tok, cls, reg, pos)
    B, hw, D = tok.shape
    R = 0 if reg is None else reg.shape[0]
    for t in (tok, cls, reg, pos):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    assert cls.numel() == D and pos.shape == (1 + hw, D)
    out = torch.empty((B, 1 + R + hw, D), dtype=torch.float32, device=tok.device)
    _lib.check(_lib.load().uc_assemble_tokens(tok.data_ptr(), cls.data_ptr(), _p(reg), pos.data_ptr(), out.data_ptr(), B, hw, R, D,
                                              _stream()), "uc_assemble_tokens")
    return out


def token_slice(src: torch.Tensor, start: int, n: int) -> torch.Tensor:
    """src fp32 [B,Ns,D] -> contiguous [B,n,D] copy of rows start..start+n."""
    _need_gpu(src)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 3
    B, Ns, D = src.shape
    dst = torch.empty((B, n, D), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().uc_token_slice(src.data_ptr(), dst.data_ptr(), B, Ns, n, start, 0, n, D, _stream()), "uc_token_slice")
    return dst


# --------------------------------------------------------------------------------------------
# training path
# --------------------------------------------------------------------------------------------
def layernorm_bwd(x: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, eps: float, dgamma: torch.Tensor,
                  dbeta: torch.Tensor, dres: Optional[torch.Tensor] = None, bf16_twin: bool = False):
    """dx (x's dtype: fp32, or bf16 on a bf16 training stream) of y = LN(x); dgamma/dbeta are accumulated into (fp32, caller-zeroed).
    dres: gradient of a parallel residual branch to add into dx (same dtype as x).  bf16_twin (fp32 stream only): also return a bf16
    copy of dx written in the same pass."""
    _need_gpu(x, gamma, dy, dgamma, dbeta, dres)
    assert x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    dx = torch.empty_like(x)
    twin = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if (bf16_twin and x.dtype == torch.float32) else None
    if dres is not None:
        assert dres.dtype == x.dtype and dres.is_contiguous() and dres.shape == x.shape
    _lib.check(_lib.load().uc_layernorm_bwd(x.data_ptr(), _dt(x.dtype), gamma.data_ptr(), dy.data_ptr(), _dt(dy.dtype), _p(dres), dx.data_ptr(),
                                            _p(twin), dgamma.data_ptr(), dbeta.data_ptr(), rows, Cn, float(eps), _stream()),
               "uc_layernorm_bwd")
    return (dx, twin if twin is not None else dx) if bf16_twin else dx


def gemm_tn(a: torch.Tensor, b: torch.Tensor, split_k: int = 1, conv: Optional[Tuple[int, bool]] = None,
            colsum: bool = False, colsum_into: Optional[torch.Tensor] = None):
    """Weight-gradient contraction over the slow axis: returns fp32 slabs [split_k, I, J] of  sum_t a[t,i] * b[t,j].
    a: [T,I] bf16 (unit column stride).  b: [T,J] bf16, or with conv=(stride, relu) the NHWC input [B,H,W,Cin] of a
    3x3/pad-1 conv whose im2col ([T, 9*Cin], T = output pixels) is formed implicitly.
    colsum=True: also returns slabs [split_k, I] of sum_t a[t,i] (the bias gradient).
    colsum_into: fp32 [I] buffer that receives += sum_t a[t,i] atomically instead (e.g. the bias's gradient buffer)."""
    _need_gpu(a, b)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and a.stride(1) == 1
    T, I = a.shape
    if conv is None:
        assert b.dim() == 2 and b.stride(1) == 1 and b.shape[0] == T
        J, ldb, cg, relu = b.shape[1], b.stride(0), (0, 0, 0, 0, 0), 0
    else:
        stride, relu = conv
        assert b.dim() == 4 and b.is_contiguous()
        Bc, Hc, Wc, Cin = b.shape
        J, ldb, cg = 9 * Cin, 0, (Bc, Hc, Wc, Cin, stride)
    out = torch.empty((split_k, I, J), dtype=torch.float32, device=a.device)
    if colsum_into is not None:
        assert colsum_into.dtype == torch.float32 and colsum_into.is_contiguous() and colsum_into.numel() == I and not colsum
        cs, atomic = colsum_into, 1
    else:
        cs, atomic = (torch.empty((split_k, I), dtype=torch.float32, device=a.device) if colsum else None), 0
    _lib.check(_lib.load().uc_gemm_tn(a.data_ptr(), a.stride(0), b.data_ptr(), ldb, T, I, J, *cg, 1 if relu else 0, out.data_ptr(),
                                      _p(cs), atomic, split_k, _stream()), "uc_gemm_tn")
    return (out, cs) if colsum else out


def gemm_tn_conv_tiles(Cout: int, H: int, W: int, Cin: int, stride: int) -> int:
    "Workgroups one K-slice of gemm_tn(conv=...) occupies (uc_gemm_tn_conv_tiles): sizes split_k."
    return int(_lib.load().uc_gemm_tn_conv_tiles(int(Cout), int(H), int(W), int(Cin), int(stride)))


def splitk_reduce(ws: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """ws [sk, M, N] fp32 slabs (uc_gemm split_k / uc_gemm_tn output, or a row range ws_full[:, r0:r1] of them) ->
    out [M,N] (= or += the sum over slabs)."""
    _need_gpu(ws, out)
    assert ws.dtype == torch.float32 and ws.dim() == 3 and ws.stride(2) == 1 and ws.stride(1) == ws.shape[2]
    if out is None:
        assert not accumulate
        out = torch.empty(ws.shape[1:], dtype=torch.float32, device=ws.device)
    assert out.is_contiguous() and out.numel() == ws.shape[1] * ws.shape[2] and out.dtype == torch.float32
    stride = ws.stride(0) if ws.shape[0] > 1 else out.numel()
    _lib.check(_lib.load().uc_splitk_reduce(ws.data_ptr(), ws.shape[0], out.numel(), stride, out.data_ptr(), 1 if accumulate else 0,
                                            _stream()), "uc_splitk_reduce")
    return out


def colsum_(src: torch.Tensor, out: torch.Tensor) -> None:
    """out[n] += sum_m src[m, n] (out fp32)."""
    _need_gpu(src, out)
    assert src.dim() == 2 and src.stride(1) == 1 and out.dtype == torch.float32 and out.numel() == src.shape[1]
    _lib.check(_lib.load().uc_colsum(src.data_ptr(), _dt(src.dtype), src.shape[0], src.shape[1], src.stride(0), out.data_ptr(),
                                     _stream()), "uc_colsum")


def mask_scale(x: torch.Tensor, mask: torch.Tensor, rows_per_mask: int, scale: float, residual: Optional[torch.Tensor] = None,
               out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Dropout / DropPath (training): out [rows, cols] = (residual +) x * (mask ? scale : 0) (uc_mask_scale).  mask uint8: one per element
    (rows_per_mask == 0) or one per group of rows_per_mask rows.  Its own backward when applied to the gradient without residual."""
    _need_gpu(x, mask, residual)
    assert x.dim() == 2 and x.is_contiguous() and mask.dtype == torch.uint8 and mask.is_contiguous() and x.shape[1] % 4 == 0
    rows, cols = x.shape
    assert mask.numel() == (x.numel() if rows_per_mask == 0 else (rows + rows_per_mask - 1) // rows_per_mask)
    od = out_dtype or (residual.dtype if residual is not None else x.dtype)
    if residual is not None:
        assert residual.shape == x.shape and residual.is_contiguous() and residual.dtype == od
    out = torch.empty((rows, cols), dtype=od, device=x.device)
    _lib.check(_lib.load().uc_mask_scale(x.data_ptr(), _dt(x.dtype), mask.data_ptr(), int(rows_per_mask), float(scale), _p(residual),
                                         out.data_ptr(), _dt(od), rows, cols, _stream()), "uc_mask_scale")
    return out


def act_bwd(dg: torch.Tensor, u: torch.Tensor, act: str) -> torch.Tensor:
    _need_gpu(dg, u)
    assert dg.is_contiguous() and u.is_contiguous() and dg.shape == u.shape and dg.dtype == u.dtype
    du = torch.empty_like(dg)
    _lib.check(_lib.load().uc_act_bwd(dg.data_ptr(), u.data_ptr(), du.data_ptr(), _dt(dg.dtype), ACT[act], dg.numel(), _stream()),
               "uc_act_bwd")
    return du


def swiglu(t: torch.Tensor) -> torch.Tensor:
    """t [M, 2H] (= w12(x)) -> silu(t[:, :H]) * t[:, H:]  [M, H]: the gate of DINOv2 giant's SwiGLU FFN.  fp32 / bf16, H % 8 == 0."""
    M, H2 = t.shape
    if not t.is_contiguous() or H2 % 16 != 0 or t.dtype not in (torch.float32, torch.bfloat16):
        raise UcHipError(f"swiglu: contiguous fp32 / bf16 [M, 2H] with H % 8 == 0 expected (got {tuple(t.shape)}, {t.dtype})")
    g = torch.empty((M, H2 // 2), dtype=t.dtype, device=t.device)
    _lib.check(_lib.load().uc_swiglu(t.data_ptr(), g.data_ptr(), _dt(t.dtype), M, H2 // 2, _stream()), "uc_swiglu")
    return g


def swiglu_bwd(dg: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    "Gradient of swiglu with respect to t: dg [M, H], t [M, 2H] -> dt [M, 2H]."
    M, H2 = t.shape
    if not (t.is_contiguous() and dg.is_contiguous()) or dg.shape != (M, H2 // 2) or dg.dtype != t.dtype or H2 % 16 != 0:
        raise UcHipError("swiglu_bwd: contiguous dg [M, H] and t [M, 2H] of one dtype expected")
    dt_ = torch.empty_like(t)
    _lib.check(_lib.load().uc_swiglu_bwd(dg.data_ptr(), t.data_ptr(), dt_.data_ptr(), _dt(t.dtype), M, H2 // 2, _stream()), "uc_swiglu_bwd")
    return dt_


def transpose2d(x: torch.Tensor, out_dtype: Optional[torch.dtype] = None, pad_to: int = 1, with_copy: bool = False):
    """[R,S] contiguous -> [S,Rp] contiguous, Rp = R rounded up to `pad_to` (<= 64; pad columns are zeros).
    with_copy: also return the un-transposed matrix converted to out_dtype (one pass over the source)."""
    _need_gpu(x)
    assert x.dim() == 2 and x.is_contiguous() and 1 <= pad_to <= 64
    out_dtype = out_dtype or x.dtype
    R, S = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    y = torch.empty((S, Rp), dtype=out_dtype, device=x.device)
    c = torch.empty((R, S), dtype=out_dtype, device=x.device) if with_copy else None
    _lib.check(_lib.load().uc_transpose2d(x.data_ptr(), _dt(x.dtype), y.data_ptr(), _dt(out_dtype), _p(c), R, S, Rp, _stream()),
               "uc_transpose2d")
    return (y, c) if with_copy else y


def pointmap_adaptor_bwd(x: torch.Tensor, dpts: Optional[torch.Tensor], dconf: Optional[torch.Tensor], conf_vmin: float,
                         conf_vmax: float) -> torch.Tensor:
    _need_gpu(x, dpts, dconf)
    assert x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 4
    B, _, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    assert sh == W * sw
    for t in (dpts, dconf):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    dx = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device)
    vmax = conf_vmax if conf_vmax != float("inf") else 3.0e38
    _lib.check(_lib.load().uc_pointmap_adaptor_bwd(x.data_ptr(), sb, sc, sw, _p(dpts), _p(dconf), float(conf_vmin), float(vmax),
                                                   dx.data_ptr(), B, H, W, _stream()), "uc_pointmap_adaptor_bwd")
    return dx


def conf_loss(pts: torch.Tensor, conf: torch.Tensor, gt: torch.Tensor, alpha: float, grad_scale: float,
              loss_sum: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_gpu(pts, conf, gt, loss_sum)
    for t in (pts, conf, gt):
        assert t.dtype == torch.float32 and t.is_contiguous()
    npix = conf.numel()
    assert pts.numel() == 3 * npix and gt.numel() == 3 * npix
    dpts, dconf = torch.empty_like(pts), torch.empty_like(conf)
    _lib.check(_lib.load().uc_conf_loss(pts.data_ptr(), conf.data_ptr(), gt.data_ptr(), float(alpha), float(grad_scale),
                                        loss_sum.data_ptr(), dpts.data_ptr(), dconf.data_ptr(), npix, _stream()), "uc_conf_loss")
    return dpts, dconf


def pointmap_loss(x: torch.Tensor, gt: torch.Tensor, alpha: float, grad_scale: float, loss_sum: torch.Tensor) -> torch.Tensor:
    """x: fp32 4-channel BCHW-shaped map (dense rows); gt [B,H,W,3] fp32. Adds the summed loss into loss_sum[0] and
    returns d(loss_sum)/dx * grad_scale with the memory layout of x."""
    _need_gpu(x, gt, loss_sum)
    assert x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 4 and gt.is_contiguous() and gt.dtype == torch.float32
    B, _, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    assert sh == W * sw
    dx = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().uc_pointmap_loss(x.data_ptr(), sb, sc, sw, gt.data_ptr(), float(alpha), float(grad_scale),
                                            loss_sum.data_ptr(), dx.data_ptr(), B, H, W, _stream()), "uc_pointmap_loss")
    return dx


def pixel_unshuffle(g: torch.Tensor, P: int, out_dtype: torch.dtype) -> torch.Tensor:
    """g fp32 NCHW [B,Cout,P*h,P*w] contiguous -> [B*h*w, Cout*P*P]."""
    _need_gpu(g)
    assert g.dtype == torch.float32 and g.is_contiguous()
    B, Cout, Hd, Wd = g.shape
    h, w = Hd // P, Wd // P
    y = torch.empty((B * h * w, Cout * P * P), dtype=out_dtype, device=g.device)
    _lib.check(_lib.load().uc_pixel_unshuffle(g.data_ptr(), y.data_ptr(), _dt(out_dtype), B, h, w, P, Cout, _stream()),
               "uc_pixel_unshuffle")
    return y


def adamw_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
           weight_decay: float, step: int, grad_scale: float = 1.0) -> None:
    _need_gpu(p, g, m, v)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    _lib.check(_lib.load().uc_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1),
                                    float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _stream()), "uc_adamw")


def attention_drop_mask(B: int, H: int, Nq: int, Nk: int, p: float, seed: int, device) -> torch.Tensor:
    "uint8 [B,H,Nq,Nk] keep mask (1 = kept) of attention(..., dropout=(p, seed)): the function the kernels evaluate, for references."
    mask = torch.empty((B, H, Nq, Nk), dtype=torch.uint8, device=device)
    _need_gpu(mask)
    _lib.check(_lib.load().uc_attention_drop_mask(mask.data_ptr(), B, H, Nq, Nk, float(p), int(seed), _stream()), "uc_attention_drop_mask")
    return mask


def attention_bwd(q, k, v, o, do, lse, scale: float, out=None, rope=None, dropout=None):
    """q,o,do [B,Nq,H,64]; k,v [B,Nk,H,64] bf16 views (unit last stride); lse fp32 [B,H,Nq].
    dropout = the forward's (p, seed) when it dropped attention probabilities (the kernels re-evaluate the mask).
    Returns dq, dk, dv ([B,N,H,64] bf16): fresh contiguous tensors, or the three views passed as `out`
    (e.g. slices of one fused dqkv buffer).
    rope = (qpos int64 [B*Nq,2], kpos int64 [B*Nk,2], base, F0) (bf16 only): q and k were RoPE-rotated before the forward; dq / dk come
    back as gradients of the un-rotated q / k (the inverse rotation runs inside the backward kernels)."""
    _need_gpu(q, k, v, o, do, lse)
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    dt = q.dtype
    if not ((dt == torch.bfloat16 and D == 64) or (dt == torch.float32 and D <= 64)):
        raise UcHipError("attention backward needs bf16 with head_dim 64, or fp32 with head_dim <= 64")
    assert lse.dtype == torch.float32 and lse.is_contiguous()
    if do.stride() != o.stride() or o.stride(3) != 1:
        o, do = o.contiguous(), do.contiguous()
    if out is None:
        dq = torch.empty((B, Nq, H, D), dtype=dt, device=q.device)
        dk = torch.empty((B, Nk, H, D), dtype=dt, device=q.device)
        dv = torch.empty((B, Nk, H, D), dtype=dt, device=q.device)
    else:
        dq, dk, dv = out
        assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
        assert all(t.dtype == dt and t.stride(3) == 1 for t in out)
    # scratch: fp32 verification kernels [B,H,Nq]; bf16 kernels [B*H][2][Nq rounded up to 128] (ABI 13: the dQ kernel leaves -lse*log2(e)
    # and -rowsum(dO*O) there as the dK / dV kernel's accumulator start values)
    delta = torch.empty((B, H, Nq) if dt == torch.float32 else (B * H, 2, (Nq + 127) // 128 * 128), dtype=torch.float32, device=q.device)
    if rope is not None:
        assert dt == torch.bfloat16 and rope[0].dtype == torch.int64 and rope[1].dtype == torch.int64
        assert rope[0].is_contiguous() and rope[1].is_contiguous() and rope[0].numel() == 2 * B * Nq and rope[1].numel() == 2 * B * Nk
    drop = dropout is not None and float(dropout[0]) > 0.0
    if dt == torch.float32 and drop:
        _lib.check(_lib.load().uc_attention_bwd_f32_drop(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(),
            dv.data_ptr(), delta.data_ptr(), B, H, Nq, Nk, D,
            q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
            o.stride(0), o.stride(1), o.stride(2), dq.stride(0), dq.stride(1), dq.stride(2), dk.stride(0), dk.stride(1),
            dk.stride(2), dv.stride(0), dv.stride(1), dv.stride(2), float(scale), float(dropout[0]), int(dropout[1]), _stream()),
            "uc_attention_bwd_f32_drop")
        return dq, dk, dv
    if drop:
        _lib.check(_lib.load().uc_attention_bwd_drop(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, Nq, Nk,
            q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
            o.stride(0), o.stride(1), o.stride(2), dq.stride(0), dq.stride(1), dq.stride(2), dk.stride(0), dk.stride(1), dk.stride(2),
            dv.stride(0), dv.stride(1), dv.stride(2), float(scale),
            _p(rope[0]) if rope is not None else None, _p(rope[1]) if rope is not None else None,
            float(rope[2]) if rope is not None else 0.0, float(rope[3]) if rope is not None else 0.0,
            float(dropout[0]), int(dropout[1]), _stream()), "uc_attention_bwd_drop")
        return dq, dk, dv
    if dt == torch.float32:
        _lib.check(_lib.load().uc_attention_bwd_f32(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(),
            dv.data_ptr(), delta.data_ptr(), B, H, Nq, Nk, D,
            q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
            o.stride(0), o.stride(1), o.stride(2), dq.stride(0), dq.stride(1), dq.stride(2), dk.stride(0), dk.stride(1),
            dk.stride(2), dv.stride(0), dv.stride(1), dv.stride(2), float(scale), _stream()), "uc_attention_bwd_f32")
        return dq, dk, dv
    _lib.check(_lib.load().uc_attention_bwd(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
        dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, Nq, Nk,
        q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
        o.stride(0), o.stride(1), o.stride(2), dq.stride(0), dq.stride(1), dq.stride(2), dk.stride(0), dk.stride(1), dk.stride(2),
        dv.stride(0), dv.stride(1), dv.stride(2), float(scale),
        _p(rope[0]) if rope is not None else None, _p(rope[1]) if rope is not None else None,
        float(rope[2]) if rope is not None else 0.0, float(rope[3]) if rope is not None else 0.0, _stream()), "uc_attention_bwd")
    return dq, dk, dv


def bilinear_nhwc_bwd(dy: torch.Tensor, Hi: int, Wi: int, Ho: int, Wo: int) -> torch.Tensor:
    """dy [B,crop_h,crop_w,C] (gradient of bilinear_nhwc's output) -> dx [B,Hi,Wi,C]."""
    _need_gpu(dy)
    assert dy.is_contiguous() and dy.dim() == 4
    B, ch, cw, C = dy.shape
    dx = torch.empty((B, Hi, Wi, C), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().uc_bilinear_nhwc_bwd(dy.data_ptr(), dx.data_ptr(), _dt(dy.dtype), B, Hi, Wi, C, Ho, Wo, ch, cw, _stream()),
               "uc_bilinear_nhwc_bwd")
    return dx


def convt_gather(dy: torch.Tensor, k: int) -> torch.Tensor:
    """dy NHWC [B,k*h,k*w,Cout] -> [B*h*w, k*k*Cout]."""
    _need_gpu(dy)
    assert dy.is_contiguous() and dy.dim() == 4
    B, Hk, Wk, Cout = dy.shape
    h, w = Hk // k, Wk // k
    out = torch.empty((B * h * w, k * k * Cout), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().uc_convt_gather(dy.data_ptr(), out.data_ptr(), _dt(dy.dtype), B, h, w, k, Cout, _stream()), "uc_convt_gather")
    return out


def im2col_t(x: torch.Tensor, stride: int, relu: bool, pad_to: int = 64) -> torch.Tensor:
    """x NHWC [B,H,W,Cin] -> [9*Cin, npix_padded] (see uc_im2col_t)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dim() == 4
    B, H, W, Cin = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    npix = B * Ho * Wo
    ld = (npix + pad_to - 1) // pad_to * pad_to
    out = torch.empty((9 * Cin, ld), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().uc_im2col_t(x.data_ptr(), out.data_ptr(), _dt(x.dtype), B, H, W, Cin, stride, 1 if relu else 0, ld, _stream()),
               "uc_im2col_t")
    return out


def dilate_nhwc(src: torch.Tensor, H: int, W: int, stride: int) -> torch.Tensor:
    _need_gpu(src)
    assert src.is_contiguous() and src.dim() == 4
    B, h, w, C = src.shape
    out = torch.empty((B, H, W, C), dtype=src.dtype, device=src.device)
    _lib.check(_lib.load().uc_dilate_nhwc(src.data_ptr(), out.data_ptr(), _dt(src.dtype), B, h, w, H, W, C, stride, _stream()),
               "uc_dilate_nhwc")
    return out


def conv1x1_to4_bwd(feat: torch.Tensor, w: torch.Tensor, dout: torch.Tensor, dw: torch.Tensor, db: torch.Tensor,
                    relu_mask: bool = False) -> torch.Tensor:
    """feat NHWC [...,Cin], w fp32 [4,Cin], dout fp32 [...,4] -> dfeat (feat's dtype); dw/db accumulated (fp32).
    relu_mask: feat is a ReLU's output; its backward (zero where feat <= 0) is applied to dfeat in the same pass."""
    _need_gpu(feat, w, dout, dw, db)
    assert feat.is_contiguous() and dout.is_contiguous() and dout.dtype == torch.float32 and w.is_contiguous()
    Cin = feat.shape[-1]
    npix = feat.numel() // Cin
    dfeat = torch.empty_like(feat)
    _lib.check(_lib.load().uc_conv1x1_to4_bwd(feat.data_ptr(), _dt(feat.dtype), w.data_ptr(), dout.data_ptr(), dfeat.data_ptr(),
                                              dw.data_ptr(), db.data_ptr(), npix, Cin, 1 if relu_mask else 0, _stream()), "uc_conv1x1_to4_bwd")
    return dfeat
