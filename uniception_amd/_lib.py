"""ctypes binding of libuc_hip.so (C ABI declared in include/uc_hip.h).

The product path has no CPU or PyTorch fallback: if the library is missing or a kernel call
fails, an exception is raised (``UcHipError``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UC_HIP_LIB: A/B-test another build; UNICEPTION_AMD_DIAG_LIB=1: the diagnostics build (python -m uniception_amd.build --diag), tools only
LIB_PATH = os.environ.get("UC_HIP_LIB") or os.path.join(_HERE, "libuc_hip_diag.so" if os.environ.get("UNICEPTION_AMD_DIAG_LIB") == "1"
                                                        else "libuc_hip.so")

UC_F32, UC_BF16, UC_F16 = 0, 1, 2
UC_A_DENSE, UC_A_CONV3X3 = 0, 1
UC_ACT_NONE, UC_ACT_GELU_ERF, UC_ACT_RELU = 0, 1, 2
UC_V_ROWMAJOR, UC_V_PACKED_T = 0, 1

i32, i64, f32, vp = C.c_int, C.c_int64, C.c_float, C.c_void_p
u64 = C.c_uint64


class UcHipError(RuntimeError):
    pass


class AdaptorSeg(C.Structure):
    # field order == struct uc_adaptor_seg in include/uc_hip.h
    _fields_ = [("op", i32), ("mode", i32), ("flags", i32), ("c0", i32), ("n", i32), ("o0", i32), ("p", f32 * 4), ("vmin", f32), ("vmax", f32)]


UC_AD_ELEM, UC_AD_NORM, UC_AD_ZEXP, UC_AD_DIR, UC_AD_CONF_EXP, UC_AD_CONF_SIGMOID, UC_AD_MASK, UC_AD_FLOW, UC_AD_FLOWCOORD, UC_AD_COV2D = range(1, 11)


class GemmDesc(C.Structure):
    # field order == struct uc_gemm_desc in include/uc_hip.h
    _fields_ = [
        ("compute_dtype", i32), ("a_mode", i32), ("relu_a", i32),
        ("A", vp), ("lda", i64), ("W", vp), ("M", i64), ("N", i64), ("K", i64),
        ("conv_B", i32), ("conv_H", i32), ("conv_W", i32), ("conv_Cin", i32), ("conv_stride", i32),
        ("conv_Ho", i32), ("conv_Wo", i32),
        ("bias", vp), ("act", i32), ("residual", vp), ("residual2", vp), ("res_dtype", i32), ("ldr", i64),
        ("rope_cols", i64), ("rope_pos", vp), ("rope_table", vp), ("rope_npos", i32), ("rope_base", f32), ("rope_f0", f32),
        ("vt_col0", i64), ("vt_out", vp), ("vt_ntok", i32), ("vt_npad", i32),
        ("preact_out", vp), ("split_k", i32), ("dact_u", vp), ("dact_act", i32),
        ("C", vp), ("out_dtype", i32), ("ldc", i64),
        ("twin_out", vp), ("ldt", i64), ("stats_out", vp), ("ln_stats", vp), ("ln_colsum", vp),
        ("tail_w", vp), ("tail_b", vp), ("tail_out", vp),
        ("ln_nblk", i32), ("ln_eps", f32), ("fuse_ws", vp), ("sat_flag", vp),
    ]


# name -> argtypes (every function returns int except uc_last_error)
SIGNATURES = {
    "uc_abi_version": [],
    "uc_tuning_set": [C.c_char_p, i32],
    "uc_tuning_get": [C.c_char_p, C.POINTER(i32)],
    "uc_rope2d": [vp, vp, i32, i32, i32, i32, i64, i64, i64, f32, f32, i32, vp],
    "uc_rope_table": [vp, i32, i32, f32, f32, vp],
    "uc_layernorm": [vp, i32, vp, vp, vp, i32, i64, i32, f32, vp],
    "uc_layernorm_twin": [vp, i32, vp, vp, vp, i32, vp, i64, i32, f32, vp],
    "uc_gemm": [C.POINTER(GemmDesc), vp],
    "uc_ln_stats_finalize": [vp, i64, i32, f32, vp, vp],
    "uc_split_bf16x3": [vp, vp, i64, i32, i32, vp],
    "uc_add_view_pe": [vp, vp, i64, i32, i32, i32, i32, vp],
    "uc_attention_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32] + [i64] * 12 + [f32, vp, vp],
    "uc_attention_x3_workspace_bytes": [i32, i32, i32, i32],
    "uc_gemm_fuse_ws_bytes": [],
    "uc_attention_fwd_x3": [vp, vp, vp, vp, vp, i32, i32, i32, i32] + [i64] * 12 + [f32, vp, vp, vp, vp, i32, vp],
    "uc_attention_fwd_fp8": [vp, vp, vp, vp, i32, i32, i32, i32] + [i64] * 9 + [f32, vp],
    "uc_vt_pack_fp8": [vp, vp, i32, i32, i32, i32, i64, i64, i64, vp],
    "uc_attention_fwd_fp8_k8": [vp, vp, vp, vp, i32, i32, i32, i32] + [i64] * 6 + [f32, vp],
    "uc_k_pack_fp8": [vp, vp, i32, i32, i32, i64, i64, i64, vp],
    "uc_vt_pack": [vp, vp, i32, i32, i32, i32, i64, i64, i64, vp],
    "uc_patch_gather": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "uc_nchw_to_nhwc": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "uc_nhwc_to_nchw": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "uc_convert": [vp, i32, vp, i32, i64, vp, vp],
    "uc_bilinear_nhwc": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "uc_convt_scatter": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "uc_pixel_shuffle": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "uc_pointmap_adaptor": [vp, i64, i64, i64, vp, vp, i32, i32, i32, f32, f32, vp],
    "uc_conv1x1_to4": [vp, i32, vp, vp, vp, i64, i32, vp],
    "uc_adaptor_program": [vp, i64, i64, i64, vp, i32, i32, i32, i32, vp, i32, vp],
    "uc_assemble_tokens": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "uc_token_slice": [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "uc_layernorm_bwd": [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i64, i32, f32, vp],
    "uc_gemm_tn": [vp, i64, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp],
    "uc_gemm_tn_conv_tiles": [i64, i32, i32, i32, i32],
    "uc_splitk_reduce": [vp, i32, i64, i64, vp, i32, vp],
    "uc_colsum": [vp, i32, i64, i64, i64, vp, vp],
    "uc_act_bwd": [vp, vp, vp, i32, i32, i64, vp],
    "uc_mask_scale": [vp, i32, vp, i64, f32, vp, vp, i32, i64, i32, vp],
    "uc_swiglu": [vp, vp, i32, i64, i64, vp],
    "uc_swiglu_bwd": [vp, vp, vp, i32, i64, i64, vp],
    "uc_transpose2d": [vp, i32, vp, i32, vp, i64, i64, i64, vp],
    "uc_pointmap_adaptor_bwd": [vp, i64, i64, i64, vp, vp, f32, f32, vp, i32, i32, i32, vp],
    "uc_adaptor_program_bwd": [vp, i64, i64, i64, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp],
    "uc_conf_loss": [vp, vp, vp, f32, f32, vp, vp, vp, i64, vp],
    "uc_pointmap_loss": [vp, i64, i64, i64, vp, f32, f32, vp, vp, i32, i32, i32, vp],
    "uc_pixel_unshuffle": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "uc_adamw": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp],
    "uc_attention_bwd": [vp] * 10 + [i32] * 4 + [i64] * 21 + [f32, vp, vp, f32, f32, vp],
    "uc_bilinear_nhwc_bwd": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "uc_convt_gather": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "uc_im2col_t": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, vp],
    "uc_dilate_nhwc": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "uc_conv1x1_to4_bwd": [vp, i32, vp, vp, vp, vp, vp, i64, i32, i32, vp],
    "uc_attention_bwd_f32": [vp] * 10 + [i32] * 5 + [i64] * 21 + [f32, vp],
    "uc_attention_fwd_drop": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32] + [i64] * 12 + [f32, vp, f32, u64, vp],
    "uc_attention_bwd_drop": [vp] * 10 + [i32] * 4 + [i64] * 21 + [f32, vp, vp, f32, f32, f32, u64, vp],
    "uc_attention_bwd_f32_drop": [vp] * 10 + [i32] * 5 + [i64] * 21 + [f32, f32, u64, vp],
    "uc_attention_drop_mask": [vp, i32, i32, i32, i32, f32, u64, vp],
}

_lib = None


ABI_VERSION = 14   # UC_ABI_VERSION of include/uc_hip.h this binding was written against


def load():
    """Load libuc_hip.so (once). Raises UcHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UcHipError(
            f"{LIB_PATH} not found: build it with `python -m uniception_amd.build` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for the HIP path."
        )
    lib = C.CDLL(LIB_PATH)
    lib.uc_last_error.restype = C.c_char_p
    lib.uc_last_error.argtypes = []
    lib.uc_build_flavor.restype = C.c_char_p
    lib.uc_build_flavor.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = i64 if name.endswith("_bytes") else i32
        fn.argtypes = args
    if lib.uc_abi_version() != ABI_VERSION:
        raise UcHipError(f"{LIB_PATH} has ABI version {lib.uc_abi_version()}, this binding expects {ABI_VERSION}: rebuild it "
                         "with `python -m uniception_amd.build`")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().uc_last_error().decode("utf-8", "replace")
        raise UcHipError(f"{what} failed with status {status}: {msg}")
