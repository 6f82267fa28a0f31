// Parameters / launcher of the direct-to-LDS dense bf16 GEMM (gemm_glds.hip), shared with the uc_gemm dispatcher.
#pragma once
#include "common.h"

struct GldsParams {
    const bf16_t* A;
    int64_t lda;
    const bf16_t* W;
    int64_t M, N, K;
    const float* bias;
    int act;
    const void* residual;
    const void* residual2;
    int res_dtype;
    int64_t ldr;
    int64_t rope_cols;
    const int64_t* rope_pos;
    const float2* rope_table;
    int rope_npos;
    float rope_ratio;       // base^(-1/16): frequency ratio of neighbouring channels of a quarter
    float rope_turn0;       // rotation per unit position of channel 0, in turns: F0 / (2 pi); channel i: rope_turn0 * rope_ratio^i
    float rope_l2ratio;     // log2(rope_ratio)
    int64_t vt_col0;
    bf16_t* vt_out;
    int vt_ntok, vt_npad;
    void* C;
    int out_dtype;
    int64_t ldc;
    // LayerNorm folded into this GEMM (consumer side): A holds raw rows x, W has gamma folded in; the bf16-store epilogues
    // compute rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n]
    const float2* ln_stats;   // [M] (mean, rstd)
    const float* ln_colsum;   // [N] sum_k W'[n,k]
    // ... or, instead of finalized ln_stats, the producer's per-block partials: merged per row in the epilogue (uc_ln_merge_row) —
    // small batches, where a 4-us merge launch per LayerNorm is a tenth of the forward
    const float2* ln_partial; // [ln_nblk][M] (block-major) (sum, squared deviations from the block mean) per 64-column block of x
    int ln_nblk;
    float ln_eps;
    // producer side (fp32-output epilogue): bf16 twin of the stored rows and per-row statistics of every 64-column block
    bf16_t* twin;             // [M, ldt] bf16 copy of C, or NULL
    int64_t ldt;
    float2* stats_out;        // [N/64][M] (block-major) (sum, sum of squared deviations from the block mean), or NULL
    // fused narrow tail (N == 128): out4[m][o] = tail_b[o] + sum_n act(acc + bias)[m][n] * tail_w[o][n]; C is not stored
    const float* tail_w;      // [4][N]
    const float* tail_b;      // [4] or NULL
    float* tail_out;          // [M][4]
    void* preact;   // optional pre-activation copy (same dtype / ld as C)
    const bf16_t* dact_u;   // optional: multiply the result by act'(u), u bf16 [M, ldc]
    int dact_act;
    int split_k;    // >1: K split over blockIdx groups, slice s writes its fp32 partial product to C + s*M*ldc
    // fused two-way split of K (the latency regime's small dense launches, 128x128 tiles on at most half the CUs): workgroups [0, nwg)
    // compute the first half of K and hand their accumulators over through fs_ws; workgroups [nwg, 2 nwg) compute the second half, wait
    // for their partner's flag, add, and run the normal epilogue
    int fuse_split2;
    float* fs_ws;           // [tiles][128 * 128] fp32
    unsigned* fs_flags;     // [tiles], 0 between launches
    int tiles_m, tiles_n;
    int group_m;  // row panels per L2-sharing tile group (tile traversal order)
    uc_fastdiv dNwg, dPerGroup, dGm, dGmLast;   // exact fast division by tiles_m*tiles_n, group_m*tiles_n, group_m, tiles_m % group_m
    int vec_ok;   // C / residual / bias satisfy the alignment needed by the 4-wide vector epilogue
    // implicit-GEMM 3x3 convolution over an NHWC image (a_mode == UC_A_CONV3X3): K = 9*Cin, Cin % 64 == 0
    int dbg;      // diagnostics only (UC_GEMM_DBG): 1 skip the in-loop DMA, 2 skip the in-loop barrier, 4 no epilogue, 8 one K-step, 16 generic epilogue only, 32 (eight-wave kernel) no wait for the DMA, 64 (conv) A tiles staged for tap 0 only, 128 / 256 (fp32 residual epilogue) no residual read / no twin write
    int stagger;  // experiment: 100-MHz ticks of start delay per phase group for the first round of workgroups (0 = off)
    int side_lds; // (eight-wave kernel, BF16 family; set by its launcher) the tile's row statistics / column sums / bias / RoPE positions are
                  // DMA-staged into 8 KiB of LDS behind the ring at kernel start: the epilogue's first loads are LDS reads, not an exposed L2 / HBM round trip
    int nt_out;   // output (+ residual) streams of more than half the 256 MB Infinity Cache: non-temporal epilogue loads / stores
    unsigned long long* trace;   // diagnostics (UC_GEMM_TRACE): per-workgroup {start, loop start, loop end, end} 100-MHz ticks + HW id
    int* sat_flag; // fp16 outputs: set to 1 (atomic or) when a value beyond +-65504 was saturated; NULL: not reported
    int f16;      // operands / 16-bit outputs / 16-bit residuals are fp16 instead of bf16 (the heads' TF32-class mode): EPI_ALL family only
    int a_mode, relu_a;
    int cH, cW, cCin, cStride, cHo, cWo;
    uc_fastdiv dWo, dHo, dHWo, dCin;   // exact fast division by cWo, cHo, cHo*cWo, cCin (conv index math)
};

// variant: 0 = 128x128 tile (4 waves), 1 = 256x128 (8 waves), 2 = 256x256 (16 waves)
// auto_variant: the variant came from the dispatcher's heuristic (not UC_GEMM_VARIANT): the launcher may refine it per epilogue family
int uc_launch_gemm_glds(const GldsParams& p, int variant, hipStream_t st, bool auto_variant = false);
