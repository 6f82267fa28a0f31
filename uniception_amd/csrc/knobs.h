// Tuning knobs of libuc_hip.so.
//
// Every environment variable the library honours is read ONCE, on first use, under std::call_once (uc_knobs()): no lazily
// initialised function-local statics, no getenv on a launch path.  All of them choose between CORRECT code paths (tile variants,
// cache policies, workgroup sizes); none changes results beyond what the chosen kernel's own arithmetic order does.
// Two of them can also be switched at run time through the C ABI (uc_tuning_set: atomics) — tests and micro-benchmarks run every
// tile variant inside one process.
//
// Diagnostics that produce WRONG results or allocate / synchronise (UC_GEMM_DBG, UC_ATTN_DBG, UC_GEMM_TRACE) exist only in a
// build with -DUC_DIAG (python -m uniception_amd.build --diag -> libuc_hip_diag.so, never the shipped library): in the release
// build UC_DBG(...) / UC_TRACE(...) are compile-time constants and the code behind them is not in the binary.
#pragma once
#include <atomic>

struct UcKnobs {
    int gemm_group_m;        // UC_GEMM_GROUP_M      row panels per L2-sharing tile group (default 4)
    int gemm_coresident;     // UC_GEMM_CORESIDENT   256x128x32 tiles with two workgroups per CU for narrow outputs (default 1)
    int gemm_nt;             // UC_GEMM_NT           non-temporal epilogue mask override (-1: policy)
    int gemm_8wave;          // UC_GEMM_8WAVE        eight-wave 256x256 kernel: 0 off, 1 bf16-store family (default), 2 all, 3 + bf16 stream
    int conv_dw_rows;        // UC_CONV_DW_ROWS      row-walking conv weight-gradient kernel where the shape allows (default 1; 0: implicit im2col everywhere)
    int gemm_4wave_min_k;    // UC_GEMM_4WAVE_MIN_K  ... for launches at least this deep (2048: where it beats the 16-wave kernel)
    int gemm_side_lds;       // UC_GEMM_SIDE_LDS     eight-wave kernel, folded-LayerNorm family: statistics / column sums / bias / RoPE positions of a tile DMA-staged into LDS before the K-loop (1) or loaded at the head of the epilogue (0)
    int gemm_4wave;          // UC_GEMM_4WAVE        four-wave 256x256 kernel (128x128 wave tiles, asm K-loop): 0 off, 1 bf16-store family, 2 + bf16 stream, 3 all (default)
    int gemm_small_stages;   // UC_GEMM_SMALL_STAGES 3-stage ring for launches with fewer workgroups than CUs (default 3)
    int attn_nw;             // UC_ATTN_NW           waves per attention workgroup: 0 policy, 4, 8
    int attn_dma;            // UC_ATTN_DMA          LDS-DMA attention kernel (default 1)
    int attn_prio;           // UC_ATTN_PRIO         eight-wave attention: static s_setprio 1 for waves 4-7 (default 0)
    int bilinear_rows2;      // UC_BILINEAR_ROWS2    output rows per work item of the upsampling form (default 4)
    int ln_nt;               // UC_LN_NT             non-temporal LayerNorm loads override (-1: policy)
#ifdef UC_DIAG
    int gemm_dbg;            // UC_GEMM_DBG          (diag build only) wrong-result anatomy switches of the GEMM kernels
    int attn_dbg;            // UC_ATTN_DBG          (diag build only) wrong-result anatomy switches of the attention kernel
    int gemm_trace;          // UC_GEMM_TRACE        (diag build only) per-workgroup timeline: allocates, synchronises, prints
#endif
};
const UcKnobs& uc_knobs();

// run-time switchable (uc_tuning_set): -3 / -1 mean "automatic / launcher policy"
extern std::atomic<int> g_uc_gemm_variant;   // UC_GEMM_VARIANT: -3 automatic, -1 register-staged kernel, 0..3, 6, 7 direct-to-LDS tile variants
extern std::atomic<int> g_uc_gemm_stagger;   // UC_GEMM_STAGGER: -1 launcher policy, >= 0 ticks per phase group
extern std::atomic<int> g_uc_small_m_split;  // UC_GEMM_SMALLM: small-M path of the dense GEMM — smallest K for which a launch on <= half the CUs splits K in two inside the kernel (default 2048: K = 1024 is level, 3072 / 4096 gain 26-28 %; 0: never, and a pair's bits then do not depend on its batch size)
extern std::atomic<int> g_uc_conv_rows;      // UC_CONV_ROWS: row-walking 3x3 conv kernels: 0 never, 1 where they win, 2 the 256-pixel kernel wherever the shape allows, 3 the eight-wave 512-pixel kernel wherever the shape allows
extern std::atomic<int> g_uc_conv_rows_flat; // UC_CONV_ROWS_FLAT: the eight-wave row kernel's flat form (tiles of 512 consecutive pixels, edge lanes zeroed in registers) for maps whose rows do not tile 512 pixels: 1 (default) / 0
extern std::atomic<int> g_uc_attn_p64;       // UC_ATTN_P64: persistent 64-queries-per-wave bf16 attention (attention_p64.h): 0 never, 1 where the launch has enough items (default), 2 wherever the shape allows
extern std::atomic<int> g_uc_attn_bwd64;     // UC_ATTN_BWD64: 64-rows-per-wave attention backward kernels (attention_bwd64.h): 0 never, 1 where a workgroup's rows are mostly real (default), 2 always
extern std::atomic<int> g_uc_attn_rs;        // UC_ATTN_RS: eight-wave bf16 attention as role-split segments (matrix beside vector on every SIMD): 0 / 1

#ifdef UC_DIAG
#define UC_DBG(p, bits) ((p).dbg & (bits))
#define UC_TRACE(p) ((p).trace)
#else
#define UC_DBG(p, bits) (0)
#define UC_TRACE(p) ((unsigned long long*)nullptr)
#endif
