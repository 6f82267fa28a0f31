/*
 * uc_hip.h — C ABI of libuc_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * UniCeption DUSt3R two-view pointmap hot path.
 *
 * Boundary rules:
 *   - extern "C", plain pointers + explicit sizes/strides, no torch types.
 *   - every entry point is asynchronous on the hipStream_t it is given, allocates
 *     nothing (no hipMalloc* is reachable from the release library: tests/test_abi.py reads
 *     its import table), takes every workspace from the caller (uc_attention_x3_workspace_bytes,
 *     uc_gemm_fuse_ws_bytes) and keeps no mutable global state except the thread-local
 *     last-error text and the tuning knobs below (environment read ONCE under
 *     std::call_once, a few run-time switchable atomics — all of them choose between
 *     correct code paths); returns 0 on success or a negative uc_status.
 *   - the shipped (release) build contains no diagnostics: the wrong-result anatomy
 *     switches and the allocating / synchronising timeline trace of the kernel work exist
 *     only in a -DUC_DIAG build (libuc_hip_diag.so, uc_build_flavor() == "diag").
 *   - all pointers are DEVICE pointers unless the name says "host".
 *
 * Each entry point cites the reference interface (file:line under the UniCeption tree)
 * whose arithmetic it implements.
 */
#ifndef UC_HIP_H
#define UC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* uc_stream_t; /* hipStream_t */

enum uc_dtype { UC_F32 = 0, UC_BF16 = 1, UC_F16 = 2 };

enum uc_status {
    UC_OK = 0,
    UC_ERR_BAD_ARG = -1,      /* rank / shape / stride / dtype the kernel does not support */
    UC_ERR_UNSUPPORTED = -2,  /* valid request, not implemented for this dtype/shape        */
    UC_ERR_LAUNCH = -3        /* hipGetLastError() != hipSuccess after the launch           */
};

/* Text of the last error on the calling thread ("" if none). */
const char* uc_last_error(void);
/* ABI version; bumped when a signature or the uc_gemm_desc layout changes.
 *   1: forward path.  2: training entry points, uc_gemm_desc gained preact_out / split_k / dact_u, uc_attention_fwd gained lse,
 *      fp8 attention, DINOv2 token ops.  3: uc_attention_fwd_fp8_k8, uc_k_pack_fp8 added.  4/5: see INTEGRATION.md.
 *   6: uc_adaptor_program_bwd added.  7: uc_build_flavor, uc_tuning_set / uc_tuning_get (environment knobs read once; no
 *      diagnostics in the release build), uc_attention_fwd_x3.  8: uc_gemm_desc gained ln_nblk / ln_eps.  9: uc_gemm_tn_conv_tiles added.
 *   10: uc_attention_fwd_x3 takes RoPE-2D positions (rotation fused into its operand split).
 *   11: the folded LayerNorm's block statistics (uc_gemm_desc.stats_out, ln_stats with ln_nblk > 0, uc_ln_stats_finalize) are
 *       block-major [N/64][M][2] instead of [M][N/64][2]; uc_gemm_desc gained fuse_ws (caller-provided hand-over buffer of the
 *       small-M path, uc_gemm_fuse_ws_bytes): the library no longer allocates.
 *   12: uc_swiglu / uc_swiglu_bwd added (DINOv2 giant's SwiGLU FFN); tuning knob conv_rows takes 3 (eight-wave row-walking 3x3
 *       convolution wherever the shape allows).
 *   14 (round 6): attention dropout — uc_attention_fwd_drop, uc_attention_bwd_drop, uc_attention_bwd_f32_drop, uc_attention_drop_mask
 *       added; tuning knob conv_rows_flat. */
#define UC_ABI_VERSION 14
int uc_abi_version(void);
/* "release" (the shipped library: no diagnostics compiled in) or "diag" (-DUC_DIAG: UC_GEMM_DBG / UC_ATTN_DBG / UC_GEMM_TRACE honoured). */
const char* uc_build_flavor(void);
/* Tuning knobs switchable at run time (process-wide atomics; every value selects a correct kernel):
 *   "gemm_variant": -3 automatic (default), -1 register-staged kernel, 0 128x128, 1 256x128, 2 256x256, 3 256x128x32 co-resident,
 *                   6 eight-wave / 7 four-wave (128x128 wave tiles, hand-scheduled K-loop) 256x256 tile of the direct-to-LDS bf16 GEMM;  "gemm_stagger": -1 launcher policy, >= 0 ticks.
 *   "attn_role_split": 0 / 1 — the eight-wave bf16 attention forward as role-split segments (bitwise the same results).
 * Their initial values come from UC_GEMM_VARIANT / UC_GEMM_STAGGER / UC_ATTN_RS; every other UC_* environment knob (csrc/knobs.h) is read
 * once, on first use. */
int uc_tuning_set(const char* name, int value);
int uc_tuning_get(const char* name, int* value);

/* ------------------------------------------------------------------------------------
 * RoPE-2D, in place — drop-in for the reference's only native entry point
 *   curope.rope_2d(tokens[B,N,H,D], positions[B,N,2] int64, base, fwd)
 *   (uniception/models/libs/croco/curope/curope.cpp:49-69, kernels.cu:17-108).
 * tokens is addressed as tokens[b*sb + n*sn + h*sh + d] (element strides, d contiguous)
 * so q/k views of a fused qkv buffer can be rotated in place.  fwd=+F0 rotates forward,
 * fwd=-F0 applies the inverse rotation (the gradient path, curope2d.py:24-28).
 * dtype: UC_F32 | UC_BF16 | UC_F16 (the reference has no bf16 dispatch).
 * ---------------------------------------------------------------------------------- */
int uc_rope2d(void* tokens, const int64_t* positions, int B, int N, int H, int D,
              int64_t sb, int64_t sn, int64_t sh, float base, float fwd, int dtype,
              uc_stream_t stream);

/* cos/sin table used by the fused GEMM epilogue: table[p][i] = (cos, sin)(p * F0 * base^(-i/Q)),
 * p in [0,npos), i in [0,Q), fp32 pairs.  Same angle formula as uc_rope2d. */
int uc_rope_table(float* table, int npos, int Q, float base, float F0, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dim: y = (x-mean)/sqrt(var_biased+eps)*gamma+beta
 *   (nn.LayerNorm(eps=1e-6): encoders/croco.py:32, info_sharing/cross_attention_transformer.py:42).
 * x: [rows, C] (x_dtype f32|bf16), gamma/beta fp32 [C], y: [rows, C] (y_dtype f32|bf16).
 * ---------------------------------------------------------------------------------- */
int uc_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                 int y_dtype, int64_t rows, int C, float eps, uc_stream_t stream);
/* The same with a second, bf16 copy of y written in the same pass (y_twin_bf16 [rows, C], or NULL): the features a
 * transformer hands out in fp32 (encoders/croco.py:177-180, cross_attention_transformer.py:497-503) are consumed as bf16
 * operands by the next module (decoder input projection, DPT heads) — no separate cast pass over them. */
int uc_layernorm_twin(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                      void* y_twin_bf16, int64_t rows, int C, float eps, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * GEMM family: C[M,N] = epilogue( A_op[M,K] . W[N,K]^T )
 * One descriptor serves nn.Linear (blocks.py:80-129, transformer_blocks.py:219-256,345-386),
 * the k=s patch-embed conv after im2col (patch_embed.py:47), 1x1 convs and — with
 * a_mode=UC_A_CONV3X3 — the DPT 3x3 convolutions as implicit GEMMs over an NHWC input
 * (dpt_block.py:21-80,156-177; dpt.py:107-178,271-277).
 * ---------------------------------------------------------------------------------- */
enum uc_a_mode { UC_A_DENSE = 0, UC_A_CONV3X3 = 1 };
enum uc_act { UC_ACT_NONE = 0, UC_ACT_GELU_ERF = 1, UC_ACT_RELU = 2 };

typedef struct uc_gemm_desc {
    int compute_dtype;   /* UC_F32: A,W fp32, exact-fp32 FMA chain; UC_BF16: A,W bf16, MFMA, fp32 accumulate; UC_F16 (round 3): A,W fp16
                            on the f16 MFMA (10-bit mantissa = TF32's: what the reference's fp32 heads run on under allow_tf32,
                            libs/croco/blocks.py:15, factory/dust3r.py:288-309) at the bf16 rate; 16-bit outputs / residuals are
                            then fp16; bias / activation / residual(s) / fused tail only (no RoPE, VT, LayerNorm fold, split-K) */
    int a_mode;          /* uc_a_mode */
    int relu_a;          /* apply ReLU to A elements while loading (DPT ResidualConvUnit pre-activation) */
    const void* A;       /* dense: [M,K] row-major, leading dim lda.  conv: NHWC [B,H,W,Cin] */
    int64_t lda;
    const void* W;       /* [N,K] row-major (nn.Linear weight layout); conv: [N, 9*Cin] ordered (ky,kx,c) */
    int64_t M, N, K;
    /* conv3x3 (pad 1) geometry: M = B*Ho*Wo, K = 9*Cin */
    int conv_B, conv_H, conv_W, conv_Cin, conv_stride, conv_Ho, conv_Wo;
    /* epilogue */
    const float* bias;   /* [N] fp32 or NULL */
    int act;             /* uc_act, applied after bias */
    const void* residual; /* [M,N] (ld = ldr) added after act, or NULL */
    const void* residual2; /* optional second addend, same dtype and ld as residual (DPT fusion: path + RCU(skip)) */
    int res_dtype;
    int64_t ldr;
    /* fused RoPE-2D on output columns [0, rope_cols): columns are (head, d) with head_dim 64;
       rows are tokens with positions rope_pos[m] = (y,x).  rope_table from uc_rope_table. */
    int64_t rope_cols;
    const int64_t* rope_pos;   /* [M,2] int64 */
    const float* rope_table;   /* [npos][16][2] fp32 */
    int rope_npos;
    float rope_base, rope_f0;  /* the frequencies behind rope_table (angle = pos * F0 * base^(-i/16), kernels.cu:36-81): the bf16
                                  store epilogue rotates with hardware sin/cos of that angle instead of table loads */
    /* "VT" epilogue (bf16 path): output columns [vt_col0, N) are V channels (head-major, head_dim 64);
       instead of C they are written transposed+permuted to vt_out [B,H,64,vt_npad] (see uc_attention_fwd).
       Rows are tokens, vt_ntok per batch element (M = B*vt_ntok).  vt_col0 < 0 disables. */
    int64_t vt_col0;
    void* vt_out;
    int vt_ntok, vt_npad;
    /* training-path extras (bf16 path):
       preact_out: if non-NULL, the value BEFORE the activation (acc + bias) is also stored there, same dtype/ld as C
                   (saved for the activation's backward);
       split_k   : > 1 splits the K loop over that many workgroup groups; slice s stores its fp32 partial product to
                   the slab C + s*M*ldc (C must hold split_k slabs of [M, ldc] fp32; no zero-init needed); no
                   bias/act/residual/rope/vt.  uc_splitk_reduce sums the slabs.  Used by the weight-gradient GEMMs
                   (few output tiles, very long K) — plain stores: fp32 atomics run at ~75 G/s and would dominate. */
    void* preact_out;
    int split_k;
    /* dact_u: if non-NULL, the result is multiplied by act'(u[m,n]) with act = dact_act (UC_ACT_GELU_ERF | UC_ACT_RELU) and
       u laid out like C in the compute dtype — the activation backward of the consumer fused into this data-gradient
       GEMM (du = (dy W) * act'(u)).  bf16 direct-to-LDS kernels only. */
    const void* dact_u;
    int dact_act;
    void* C;             /* [M,N] row-major, leading dim ldc (only columns < vt_col0 are written when vt is on) */
    int out_dtype;       /* UC_F32 | UC_BF16 */
    int64_t ldc;
    /* LayerNorm fused into the GEMMs around it (bf16 direct-to-LDS path, N % 64 == 0) — the "fused LayerNorm + QKV / fc1" of
       the pre-LN sub-layers (libs/croco/blocks.py:158-161, utils/transformer_blocks.py:643-646):
       producer (the GEMM that writes the fp32 residual stream: proj, fc2, patch / input embedding; out_dtype UC_F32):
         twin_out : if non-NULL, the stored rows are also written as bf16 to twin_out [M, ldt] — the A operand of the consumer;
         stats_out: if non-NULL, [N/64][M][2] fp32, BLOCK-major (ABI 11; [M][N/64][2] before): per 64-column block and row (sum, sum
                    of squared deviations from the block mean) of the stored values — a wave's 64 rows of one block are 512
                    contiguous bytes, written by one store; uc_ln_stats_finalize merges the blocks into (mean, rstd) per row;
       consumer (qkv, q / kv projections, fc1; out_dtype UC_BF16, no residual):
         A = the bf16 twin (RAW rows x), W = W * gamma[k] (bf16), bias = b + W beta,
         ln_stats : [M][2] fp32 (mean, rstd) of the rows of x, ln_colsum: [N] fp32 = sum_k W'[n,k] of the bf16 W';
         the epilogue forms rstd[m] * (acc[m,n] - mean[m] * ln_colsum[n]) + bias[n] == LayerNorm(x)[m,:] . W[n,:] + b[n]
         before the activation / RoPE / VT steps.  NULL disables. */
    void* twin_out;
    int64_t ldt;
    float* stats_out;
    const float* ln_stats;
    const float* ln_colsum;
    /* Fused narrow tail (bf16 direct-to-LDS path, N == 128): the DPT regressor's conv3x3 -> ReLU -> Conv2d(128 -> 4, 1x1)
       (prediction_heads/dpt.py:271-277, 304-309).  With tail_out non-NULL the [M,128] result is NOT stored (C may be NULL);
       instead tail_out[m][o] = tail_b[o] + sum_n act(acc[m,n] + bias[n]) * tail_w[o*N + n], o < 4, fp32 [M,4] — the
       4.3 GB per-head intermediate of a 512x512 batch is never written or read.  No residual / rope / vt / split_k. */
    const float* tail_w;   /* [4][N] fp32 */
    const float* tail_b;   /* [4] fp32 or NULL */
    float* tail_out;       /* [M][4] fp32 */
    /* Consumer side of the folded LayerNorm WITHOUT the uc_ln_stats_finalize launch (ABI 8): with ln_nblk > 0, ln_stats points at the
       producer's per-block partials [ln_nblk][M][2] (its stats_out, block-major; K == 64 * ln_nblk) and every row's (mean, rstd) is merged in
       the epilogue — the same arithmetic, bit for bit, as uc_ln_stats_finalize with ln_eps.  For small batches, where the ~120
       merge launches of a forward are a tenth of its time; at large M the stand-alone merge is cheaper (every column tile of a
       row panel repeats the in-epilogue merge).  ln_nblk == 0: ln_stats holds finalized (mean, rstd) rows. */
    int ln_nblk;
    float ln_eps;
    /* Hand-over buffer of the small-M path (ABI 11; see the notes below), or NULL: CALLER-provided — uc_gemm allocates nothing.
       uc_gemm_fuse_ws_bytes() bytes of UNCACHED device memory (hipExtMallocWithFlags(.., hipDeviceMallocUncached): the two halves of
       a split tile may run on different XCDs, whose L2s are not coherent), zero-filled once by the caller, 256-byte aligned, used
       by ONE launch at a time — one buffer per stream that issues such launches, and one per stream of a captured graph for as
       long as the graph lives.  The library leaves its flag words zero after every launch. */
    void* fuse_ws;
    /* fp16 operand form (compute_dtype UC_F16, the TF32-class prediction heads): fp16 stores SATURATE at +-65504 instead of
       becoming inf (fp16 has TF32's mantissa, not its exponent); with sat_flag non-NULL the launch ORs 1 into *sat_flag (device
       int, zeroed by the caller) when it saturated a value — the host then knows the maps left fp16's range and can rerun /
       continue with a wider head format (engine: "follow" = bf16, or "fp32").  NULL: not reported. */
    int* sat_flag;
} uc_gemm_desc;

/* Notes on uc_gemm's behaviour outside the descriptor:
 *   small-M path — a dense bf16 launch whose 128 x 128 tiles cover at most half the CUs (and at most 128 tiles), with K >= the
 *   tuning knob small_m_split (UC_GEMM_SMALLM, default 2048; 0 = never) AND a caller-provided hand-over buffer (desc->fuse_ws),
 *   splits K in two across twice the workgroups and hands the first half's accumulators over inside the kernel.  Without a buffer
 *   the launch runs unsplit.  The result is the same sum taken in two halves: not bit-identical to the unsplit kernel (and therefore
 *   to the same rows inside a larger batch); small_m_split = 0, or fuse_ws = NULL, keeps one chain per accumulator for every batch
 *   size.  Nothing in libuc_hip.so allocates device memory or keeps per-stream state (round 3 kept a lazily grown pool inside
 *   uc_gemm: removed). */
/* Size in bytes of a hand-over buffer for desc->fuse_ws: 128 tiles x 128 x 128 fp32 partial sums + 128 flag words. */
int64_t uc_gemm_fuse_ws_bytes(void);
int uc_gemm(const uc_gemm_desc* desc, uc_stream_t stream);

/* View positional encoding of the global / alternating multi-view transformers
 * (info_sharing/global_attention_transformer.py:365-392): x[b, v*T + t, :] += pe[v, :], x fp32 [B, L, C] in place (rows past
 * V*T untouched), pe fp32 [V, C].  C % 4 == 0. */
int uc_add_view_pe(float* x, const float* pe, int64_t B, int L, int T, int V, int C, uc_stream_t stream);

/* bf16x3 operand split (fp32-class GEMMs on the bf16 matrix pipe): x fp32 [rows, C] -> out bf16 [rows, 3C] = [hi | hi | lo],
 * hi = bf16(x), lo = bf16(x - hi); relu != 0 clamps x at zero first.  A weight laid out [Wh | Wl | Wh] per K block makes
 * uc_gemm accumulate xh.wh + xh.wl + xl.wh in fp32 (what the reference computes in fp32 for its prediction heads,
 * factory/dust3r.py:288-309, to ~2^-16 per product).  C % 8 == 0. */
int uc_split_bf16x3(const float* x, void* out, int64_t rows, int C, int relu, uc_stream_t stream);

/* Merge the per-block row statistics a producer GEMM wrote (stats_out: [nblk][rows][2], block-major = (sum, squared deviations from the
 * block mean) over 64-column blocks, C = 64 * nblk columns) into LayerNorm statistics: out[row] = (mean, 1/sqrt(var_biased + eps)). */
int uc_ln_stats_finalize(const float* partial, int64_t rows, int nblk, float eps, float* out, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * fp32-class attention on the matrix pipe (head_dim 64; engine.precision("bf16x3")): softmax(scale Q K^T) V with both products as
 * THREE bf16 MFMA products of split operands (x = hi + lo; hi.hi + hi.lo + lo.hi, fp32 accumulate), fp32 softmax — ~1e-5 relative
 * on the output, the arithmetic of F.scaled_dot_product_attention in fp32 (libs/croco/blocks.py:123-125,
 * models/utils/transformer_blocks.py:244-246, 373-375) at MFMA rate.  Q / K / V: fp32 [B, N, H, 64] addressed by element strides
 * (unit channel stride); O: fp32, same addressing; lse: optional fp32 [B, H, Nq].  `workspace`: device scratch of
 * uc_attention_x3_workspace_bytes(B, H, Nq, Nk) bytes (the split bf16 operands; the call allocates nothing).
 * ---------------------------------------------------------------------------------- */
int64_t uc_attention_x3_workspace_bytes(int B, int H, int Nq, int Nk);
int uc_attention_fwd_x3(const float* Q, const float* K, const float* V, float* O, void* workspace, int B, int H, int Nq, int Nk,
                        int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                        int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, float* lse,
                        const int64_t* q_pos, const int64_t* k_pos, const float* rope_table, int rope_npos, uc_stream_t stream);
/* (q_pos [B, Nq, 2] / k_pos [B, Nk, 2] int64 + rope_table [rope_npos][16] (cos, sin) from uc_rope_table: Q and K are rotated by RoPE-2D — the
 *  arithmetic of uc_rope2d, libs/croco/pos_embed.py:113-155 — inside the operand split, so the rotated rows never make a pass of their
 *  own; all NULL / 0: Q and K are taken as they are.) */

/* ------------------------------------------------------------------------------------
 * Scaled-dot-product attention, no mask, no dropout:  O = softmax(Q K^T * scale) V
 *   (F.scaled_dot_product_attention call sites: libs/croco/blocks.py:123-125,
 *    utils/transformer_blocks.py:244-246, 373-375).
 * Element addressing (d contiguous):  Q[b*q_sb + n*q_sn + h*q_sh + d], same for K and O, so
 * self-attention can pass views of a fused qkv buffer and cross-attention Nk != Nq.
 *
 * V comes in one of two layouts:
 *   v_layout = UC_V_ROWMAJOR : V[b*v_sb + n*v_sn + h*v_sh + d]                    (UC_F32 path)
 *   v_layout = UC_V_PACKED_T : "VT" — V transposed per head with the key index permuted inside
 *       every group of 16 keys so that each lane's MFMA operand is one 16-byte LDS read:
 *         VT[((b*H + h)*D + d) * Npad + 16*(n/16) + uc_vt_perm(n%16)],  Npad = roundup(Nk,64)
 *         uc_vt_perm(w) = ((w>>2)&1)*8 + (w&3) + 4*(w>>3)
 *       (required by the UC_BF16 MFMA path; produced by uc_gemm's vt epilogue or uc_vt_pack;
 *        v_sb/v_sn/v_sh are ignored).  Positions that hold no key (n >= Nk: the tail of the last 16-key group, where
 *        they are interleaved with its keys, and the groups up to Npad) must hold finite values — the kernel multiplies
 *        them by probability 0.  uc_vt_pack writes zeros there; a caller that lets uc_gemm fill VT clears
 *        positions 16*(Nk/16) .. Npad of every row first.
 * UC_BF16 requires D == 64; UC_F32 supports D <= 64.
 * ---------------------------------------------------------------------------------- */
enum uc_v_layout { UC_V_ROWMAJOR = 0, UC_V_PACKED_T = 1 };

int uc_attention_fwd(const void* Q, const void* K, const void* V, void* O, int dtype, int v_layout,
                     int B, int H, int Nq, int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh,
                     int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                     int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                     float* lse /* optional fp32 [B,H,Nq]: log-sum-exp of the scaled scores, saved for the backward */,
                     uc_stream_t stream);

/* FP8 attention forward (BASELINE config 5): same contract as uc_attention_fwd with UC_BF16 operands and D == 64, but both
 * products run on the K=64 block-scaled e4m3 MFMA (unit scales).  Q, K: bf16 strided views (converted to e4m3 inside the
 * kernel); VT8: V transposed per head in e4m3, [B,H,64,Npad] bytes, Npad = roundup(Nk,64), keys permuted inside every group
 * of 64 so that position 32*hh + 16*kb + r holds key 32*kb + (r&3) + 8*(r>>2) + 4*hh, zero padded — produced by
 * uc_vt_pack_fp8 from a row-major bf16 V.  O: bf16.  Inference only (no LSE). */
int uc_attention_fwd_fp8(const void* Q, const void* K, const void* VT8, void* O, int B, int H, int Nq, int Nk, int64_t q_sb,
                         int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t o_sb, int64_t o_sn,
                         int64_t o_sh, float scale, uc_stream_t stream);
int uc_vt_pack_fp8(const void* V, void* VT8, int B, int H, int Nk, int D, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                   uc_stream_t stream);
/* The same attention with K pre-packed too (Nk % 64 == 0): K8 = e4m3 rows [B,H,Nk,64] from uc_k_pack_fp8 (one conversion
 * per key instead of one per key AND query tile); both tiles are staged by LDS-DMA.  Q: bf16 strided view. */
int uc_attention_fwd_fp8_k8(const void* Q, const void* K8, const void* VT8, void* O, int B, int H, int Nq, int Nk, int64_t q_sb,
                            int64_t q_sn, int64_t q_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, uc_stream_t stream);
int uc_k_pack_fp8(const void* K, void* K8, int B, int H, int Nk, int64_t k_sb, int64_t k_sn, int64_t k_sh, uc_stream_t stream);

/* Row-major bf16 V[b*v_sb + n*v_sn + h*v_sh + d] -> packed VT [B,H,D,Npad] (layout above). */
int uc_vt_pack(const void* V, void* VT, int B, int H, int Nk, int D, int64_t v_sb, int64_t v_sn,
               int64_t v_sh, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Patch gather for the k=s=P patch-embed conv (libs/croco/patch_embed.py:47,69-82):
 * img fp32 NCHW [B,Cin,H,W] -> cols[B*(H/P)*(W/P), Cin*P*P] (out_dtype), column order
 * (c,u,v) == Conv2d weight.view(D,-1) order, token order row-major (i,j).
 * ---------------------------------------------------------------------------------- */
int uc_patch_gather(const float* img, void* cols, int out_dtype, int B, int Cin, int H, int W,
                    int P, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Layout/dtype conversion at the public BCHW boundary.
 *   uc_nchw_to_nhwc: src [B,C,H,W] (src_dtype) -> dst [B,H,W,C] (dst_dtype)
 *   uc_nhwc_to_nchw: src [B,H,W,C] -> dst [B,C,H,W]
 * (reference hops: encoders/croco.py:177-180, info_sharing/cross_attention_transformer.py:428-431,497-503)
 * ---------------------------------------------------------------------------------- */
int uc_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C,
                    int H, int W, uc_stream_t stream);
int uc_nhwc_to_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C,
                    int H, int W, uc_stream_t stream);
/* contiguous element-wise dtype conversion (n elements).  Conversions INTO fp16 saturate at +-65504; with sat_flag non-NULL they OR 1
   into *sat_flag (device int) when a value was beyond the fp16 range (see uc_gemm_desc.sat_flag).  NULL: not reported. */
int uc_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, int* sat_flag,
               uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=True, NHWC (dpt_block.py:251-253 scale_factor=2; dpt.py:304 size=(H,W)).
 * src [B,Hi,Wi,C] -> dst [B,Ho,Wo,C]; same dtype.  Optionally only the top-left (crop_h,crop_w)
 * window of the resized image is produced (dpt.py:213: refinenet4 output cropped to level-2 size):
 * dst is then [B,crop_h,crop_w,C].  Pass crop_h=Ho, crop_w=Wo for no crop.
 * ---------------------------------------------------------------------------------- */
int uc_bilinear_nhwc(const void* src, void* dst, int dtype, int B, int Hi, int Wi, int C, int Ho,
                     int Wo, int crop_h, int crop_w, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Pixel scatter after a GEMM with N = k*k*Cout columns ordered (u,v,o):
 *   ConvTranspose2d(k=s, p=0) (dpt.py:116-140):  dst NHWC [B, k*h, k*w, Cout],
 *   dst[b, k*i+u, k*j+v, o] = src[(b*h+i)*w+j, (u*k+v)*Cout + o]   (bias already added by the GEMM).
 * ---------------------------------------------------------------------------------- */
int uc_convt_scatter(const void* src, void* dst, int dtype, int B, int h, int w, int k, int Cout,
                     uc_stream_t stream);

/* F.pixel_shuffle(P) of the linear head (linear.py:81-82): src [B*h*w, Cout*P*P] row-major
 * (column = c*P*P + u*P + v) -> dst fp32 NCHW [B,Cout,P*h,P*w]. */
int uc_pixel_shuffle(const void* src, int src_dtype, float* dst, int B, int h, int w, int P,
                     int Cout, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * PointMapWithConfidenceAdaptor("exp", "exp"; adaptors.py:337-342, 1080-1083) fused with the
 * BCHW->BHWC permute of factory/dust3r.py:323-330.
 *   in : x fp32, element (b,c,y,x) at x[b*x_sb + c*x_sc + (y*W+x)*x_sp]  (NCHW: sc=H*W, sp=1; NHWC: sc=1, sp=4)
 *   out: pts [B,H,W,3] fp32, conf [B,H,W,1] fp32
 *   pts = xyz / max(|xyz|,1e-8) * expm1(|xyz|);  conf = conf_vmin + min(exp(c), conf_vmax - conf_vmin)
 * ---------------------------------------------------------------------------------- */
int uc_pointmap_adaptor(const float* x, int64_t x_sb, int64_t x_sc, int64_t x_sp, float* pts,
                        float* conf, int B, int H, int W, float conf_vmin, float conf_vmax,
                        uc_stream_t stream);

/* Generic adaptor pass (prediction_heads/adaptors.py:25-2300): every adaptor of the reference splits the decoded channels,
 * transforms each group per pixel and concatenates the results; a list of segments expresses the whole composition in one
 * kernel.  x: fp32 BCHW-shaped (strides sb, sc, sw in elements; rows dense), out: fp32 NHWC [B,H,W,Cout]. */
enum uc_adaptor_op {
    UC_AD_ELEM = 1,          /* n channels: mode 0 x | 1 x^2 | 2 exp(x); clip             (Depth/Scale/SceneFlow, "linear" pointmaps) */
    UC_AD_NORM = 2,          /* n-vector v: v/max(|v|,1e-8) * f(|v|), f = |v|^2 (mode 1) | expm1|v| (mode 2); clip   (PointMap, RayOrigins, CamTranslation) */
    UC_AD_ZEXP = 3,          /* (x e^z, y e^z, e^z); clip                                   (PointMap "z_exp") */
    UC_AD_DIR = 4,           /* clip; flags&1: last channel = max(last, p[0]); flags&2: unit norm; flags&4: divide by the last channel   (RayDirections, Quaternions) */
    UC_AD_CONF_EXP = 5,      /* vmin + min(exp(x), vmax - vmin)                             (Confidence "exp") */
    UC_AD_CONF_SIGMOID = 6,  /* sigmoid(x) (vmax - vmin) + vmin                             (Confidence "sigmoid") */
    UC_AD_MASK = 7,          /* two outputs: (x, sigmoid(x))                                (Mask: logits, mask) */
    UC_AD_FLOW = 8,          /* (x p[0] + p[1], y p[2] + p[3])                              (Flow: std / mean scaled to the output shape) */
    UC_AD_FLOWCOORD = 9,     /* 0.5 (x + 1) p[0] + 0.5 - (col + 0.5), same with p[1] and the row   (Flow, output_normalized_coordinate) */
    UC_AD_COV2D = 10         /* (c1, c2, s) + offset p[0] on c1, c2 -> 7 outputs: covariance (e^c1, e^c2, tanh s e^((c1+c2)/2)), log det, inverse
                                covariance                                                  (Covariance2D "exp_tanh") */
};
#define UC_ADAPTOR_MAX_SEGS 8
typedef struct uc_adaptor_seg {
    int op, mode, flags;
    int c0, n;               /* input channels [c0, c0 + n), n <= 4 */
    int o0;                  /* first output channel */
    float p[4];
    float vmin, vmax;        /* clip bounds (-inf / +inf: none) */
} uc_adaptor_seg;
int uc_adaptor_program(const float* x, int64_t sb, int64_t sc, int64_t sw, float* out, int B, int H, int W, int Cout,
                       const uc_adaptor_seg* segs, int nseg, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DPT regressor tail: ReLU'd features -> Conv1x1(Cin -> 4) + bias -> decoded channels
 * (dpt.py:271-277 conv2[2]).  feat NHWC [npix, Cin] (dtype), w fp32 [4,Cin], b fp32 [4];
 * out fp32 NHWC [npix,4].  Cin <= 256, multiple of 8.
 * ---------------------------------------------------------------------------------- */
int uc_conv1x1_to4(const void* feat, int dtype, const float* w, const float* b, float* out,
                   int64_t npix, int Cin, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Token-sequence assembly for cls/register-token ViTs (DINOv2 `prepare_tokens_with_masks`, reached through
 * encoders/dinov2.py:188): out [B, 1+R+hw, D] = [cls + pos[0] | reg[0..R) | tok[b,i] + pos[1+i]], all fp32.
 * uc_token_slice copies rows src[b, src_off + i, :] -> dst[b, dst_off + i, :], i < n (splitting cls/registers from
 * the patch tokens after the final norm, encoders/dinov2.py:191-216).
 * ---------------------------------------------------------------------------------- */
int uc_assemble_tokens(const float* tok, const float* cls, const float* reg, const float* pos, float* out, int B, int hw,
                       int R, int D, uc_stream_t stream);
int uc_token_slice(const float* src, float* dst, int B, int Ns, int Nd, int src_off, int dst_off, int n, int D,
                   uc_stream_t stream);

/* ====================================================================================
 * Training path (backward of the same hot path; config 3 of BASELINE.json).
 * The reference has no hand-written backward: it relies on PyTorch autograd over the modules cited above, so each
 * entry point below states the forward expression it differentiates.
 * ==================================================================================== */

/* LayerNorm backward (forward: uc_layernorm).  x [rows,C] in x_dtype (f32, or bf16: the bf16 training stream — dres and dx are then
 * bf16 too and dx_bf16 must be NULL), dy [rows,C] (dy_dtype f32|bf16), gamma fp32 [C].
 *   dx[r,:]  = rstd*(a - mean(a) - xhat*mean(a*xhat)) (+ dres[r,:] if dres != NULL),  a = dy*gamma, xhat = (x-mean)*rstd
 *   dgamma[c] += sum_r dy*xhat, dbeta[c] += sum_r dy     (fp32 atomic accumulation: zero them first)
 *   dx_bf16 (optional, fp32 stream): a bf16 copy of dx written in the same pass — dx is the gradient of the residual stream, which
 *   the previous sub-layer's backward GEMMs consume in bf16.
 * Any C; widths 256*{1,2,3,4,6,8} take the register-resident kernel. */
int uc_layernorm_bwd(const void* x, int x_dtype, const float* gamma, const void* dy, int dy_dtype, const void* dres, void* dx,
                     void* dx_bf16, float* dgamma, float* dbeta, int64_t rows, int C, float eps, uc_stream_t stream);

/* "TN" contraction over tokens / pixels for weight gradients, no operand transposes (bf16 in, fp32 out):
 *     C[s][i,j] = sum_{t in K-slice s} A[t,i] * B[t,j]        A = dY [T,I] (lda),  B = X [T,J] (ldb),  dW = dY^T X
 *   conv_B > 0: B is the IMPLICIT im2col of the NHWC image [conv_B,conv_H,conv_W,conv_Cin] of a 3x3 / pad-1 conv with the
 *   given stride: B[t,(ky*3+kx)*Cin+c] = act(x[b,oy*s-1+ky,ox*s-1+kx,c]), t=(b,oy,ox), J = 9*Cin, relu_b = ReLU-on-load.
 *   C holds split_k slabs of [I,J] fp32 (sum them with uc_splitk_reduce).  I, J, lda, ldb multiples of 8.
 *   colsum_a (optional): sum_t A[t,i] — the bias gradient, formed from the A fragments already in registers (no extra pass
 *   over dY): split_k slabs of [I] fp32, or with colsum_atomic != 0 ONE [I] buffer every K slice adds to atomically (+=),
 *   e.g. the bias's gradient buffer itself.
 *   (the reference gets these products from autograd's addmm / conv backward.) */
int uc_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t T, int64_t I, int64_t J, int conv_B, int conv_H,
               int conv_W, int conv_Cin, int conv_stride, int relu_b, float* C, float* colsum_a, int colsum_atomic,
               int split_k, uc_stream_t stream);
/* Output tiles one K-slice of the conv form of uc_gemm_tn occupies (workgroups per slice) — what a caller sizes split_k with so that
 * tiles * split_k fills the CUs.  Stride-1 convs on maps a multiple of 64 wide with Cin and Cout multiples of 128 take the
 * row-walking kernel (one kernel row ky per workgroup, the three taps kx as row shifts of the staged pixels: no im2col decode, every
 * input pixel fetched three times instead of nine): 3 * (Cout/128) * (Cin/128) tiles; every other shape the implicit-im2col kernel. */
int uc_gemm_tn_conv_tiles(int64_t Cout, int conv_H, int conv_W, int conv_Cin, int conv_stride);

/* out[i] = (accumulate ? out[i] : 0) + sum_s ws[s*slab_stride + i], i < n (n, slab_stride multiples of 4): reduction of the
 * split_k slabs of uc_gemm / uc_gemm_tn (a row range of every slab when slab_stride > n); with accumulate it adds the result
 * straight into a gradient buffer. */
int uc_splitk_reduce(const float* ws, int split_k, int64_t n, int64_t slab_stride, float* out, int accumulate,
                     uc_stream_t stream);

/* Column sums (bias gradients): out[n] += sum_m src[m*ld + n]; out fp32, zero-initialised by the caller. */
int uc_colsum(const void* src, int dtype, int64_t M, int64_t N, int64_t ld, float* out, uc_stream_t stream);

/* Activation backward: du = dg * act'(u), u = saved pre-activation (uc_gemm preact_out).  act: UC_ACT_GELU_ERF | UC_ACT_RELU. */
int uc_act_bwd(const void* dg, const void* u, void* du, int dtype, int act, int64_t n, uc_stream_t stream);
/* Dropout / DropPath in training (ABI 13; nn.Dropout, timm DropPath — reference blocks.py:64-86, 120-161, utils/transformer_blocks.py:145-208):
 * out [rows, cols] = (residual +) x * (mask ? scale : 0).  mask: uint8, one per element (rows_per_mask == 0) or one per group of
 * rows_per_mask consecutive rows (DropPath: rows_per_mask = tokens per sample).  x: fp32 / bf16; residual (optional) and out share
 * out_dtype (fp32 / bf16); cols % 4 == 0.  Applied to the gradient (without residual) it is its own backward. */
int uc_mask_scale(const void* x, int x_dtype, const unsigned char* mask, int64_t rows_per_mask, float scale, const void* residual,
                  void* out, int out_dtype, int64_t rows, int cols, uc_stream_t stream);

/* SwiGLU gate (DINOv2 giant's FFN; the hub's SwiGLUFFNFused, reference encoders/dinov2.py:68-84 loads it through torch.hub):
 * t [M, 2H] row-major = w12(x); g[m, j] = silu(t[m, j]) * t[m, H + j], g [M, H].  dtype UC_F32 | UC_BF16, H % 8 == 0. */
int uc_swiglu(const void* t, void* g, int dtype, int64_t M, int64_t H, uc_stream_t stream);
/* ... and its backward: dt[m, j] = dg x2 s (1 + x1 (1 - s)), dt[m, H + j] = dg x1 s with x1 = t[m, j], x2 = t[m, H + j],
 * s = sigmoid(x1); dg [M, H], t and dt [M, 2H]. */
int uc_swiglu_bwd(const void* dg, const void* t, void* dt, int dtype, int64_t M, int64_t H, uc_stream_t stream);

/* 2-D transpose src[R,S] -> dst[S, ld_dst] (row-major; f32->f32, bf16->bf16 or f32->bf16), used to put the reduction
 * dimension of the weight-gradient GEMMs (dW = dY^T X) on the contiguous axis.  ld_dst in [R, R+64): columns R..ld_dst-1
 * are written as zeros so the GEMM K dimension can be padded to a multiple of 64.  If rowmajor_copy != NULL the
 * converted (dst dtype) un-transposed [R,S] matrix is written there in the same pass. */
int uc_transpose2d(const void* src, int src_dtype, void* dst, int dst_dtype, void* rowmajor_copy, int64_t R, int64_t S,
                   int64_t ld_dst, uc_stream_t stream);

/* Backward of uc_pointmap_adaptor for an arbitrary downstream loss: dpts [B,H,W,3], dconf [B,H,W,1] (either may be
 * NULL = zero) -> dx addressed like x.  The confidence clamp at conf_vmax passes no gradient. */
int uc_pointmap_adaptor_bwd(const float* x, int64_t x_sb, int64_t x_sc, int64_t x_sp, const float* dpts,
                            const float* dconf, float conf_vmin, float conf_vmax, float* dx, int B, int H, int W,
                            uc_stream_t stream);

/* Backward of uc_adaptor_program for an arbitrary downstream loss (what torch autograd computes through the reference's
 * adaptors, prediction_heads/adaptors.py:25-2300): dout fp32 NHWC [B,H,W,Cout] -> dx fp32 with the strides of x ([B,Cin,H,W]-shaped).
 * clip / clamp pass the gradient where the unclipped value lies inside the bounds (bounds included, as torch.clip does);
 * input channels no segment reads get a zero gradient; two segments may not read the same input channel; Cin <= 64. */
int uc_adaptor_program_bwd(const float* x, int64_t sb, int64_t sc, int64_t sw, const float* dout, float* dx, int B, int H, int W,
                           int Cin, int Cout, const uc_adaptor_seg* segs, int nseg, uc_stream_t stream);

/* Confidence-weighted regression loss on adaptor outputs (the DUSt3R training objective; the reference ships no loss):
 *   loss_sum[0] += sum_pix conf*|pts-gt| - alpha*log(conf);  dpts, dconf = its gradients * grad_scale. */
int uc_conf_loss(const float* pts, const float* conf, const float* gt, float alpha, float grad_scale, float* loss_sum,
                 float* dpts, float* dconf, int64_t npix, uc_stream_t stream);

/* Fused adaptor + loss, forward and backward in one pass over the decoded channels of one view
 * (forward: uc_pointmap_adaptor; loss = mean over pixels of conf*|pts-gt| - alpha*log(conf), the DUSt3R confidence loss).
 *   x fp32 4-channel map addressed like uc_pointmap_adaptor; gt fp32 [B,H,W,3];
 *   loss_sum[0] += sum over pixels (caller divides by the pixel count); dx gets d(loss_sum)/dx * grad_scale, same addressing as x. */
int uc_pointmap_loss(const float* x, int64_t x_sb, int64_t x_sc, int64_t x_sp, const float* gt, float alpha,
                     float grad_scale, float* loss_sum, float* dx, int B, int H, int W, uc_stream_t stream);

/* Inverse of uc_pixel_shuffle for gradients: src fp32 NCHW [B,Cout,P*h,P*w] -> dst [B*h*w, Cout*P*P] (dst_dtype). */
int uc_pixel_unshuffle(const float* src, void* dst, int dst_dtype, int B, int h, int w, int P, int Cout, uc_stream_t stream);

/* AdamW step on flat fp32 buffers: p, g, m, v of n elements (decoupled weight decay).  step >= 1. */
int uc_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int step, float grad_scale, uc_stream_t stream);

/* Attention backward (forward: uc_attention_fwd, which must have been called with a non-NULL lse).
 *   inputs : Q,K,V,O,dO as [B,N,H,64] strided views (bf16; dO addressed with O's strides), LSE fp32 [B,H,Nq] (natural log of
 *            the softmax denominator of the scaled scores);
 *   outputs: dQ [B,Nq,H,64], dK, dV [B,Nk,H,64] bf16 (own strides, e.g. slices of one fused dqkv buffer).
 *   delta is fp32 scratch of 2 * B * H * (Nq rounded up to 128) floats (ABI 13; before: B * H * Nq): per (batch, head) the dQ kernel
 *   leaves -LSE * log2(e) (absent queries: -1e30) and -rowsum(dO*O) (0) there, the dK / dV kernel starts its accumulators from them.
 *   All operands are row-major: the transposed MFMA operands are formed inside the kernels with LDS transpose-reads.
 *   rope_qpos / rope_kpos (both or neither; int64 [B*Nq,2] / [B*Nk,2] (y,x)): Q and K were rotated by RoPE-2D (base, F0) before the
 *   forward — dQ and dK are then returned as gradients of the UN-rotated q / k (the inverse rotation rides in the kernels' epilogues
 *   instead of two uc_rope2d passes).  NULL: gradients of the rotated operands. */
int uc_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                     void* dQ, void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int64_t q_sb, int64_t q_sn,
                     int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                     int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb,
                     int64_t dk_sn, int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale,
                     const int64_t* rope_qpos, const int64_t* rope_kpos, float rope_base, float rope_f0,
                     uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * DPT head backward helpers (forward ops: uc_bilinear_nhwc, uc_convt_scatter, uc_gemm conv mode, uc_conv1x1_to4).
 * ---------------------------------------------------------------------------------- */
/* Adjoint of uc_bilinear_nhwc: dy [B,crop_h,crop_w,C] -> dx [B,Hi,Wi,C] (same dtype), identical index/clamp rules. */
int uc_bilinear_nhwc_bwd(const void* dy, void* dx, int dtype, int B, int Hi, int Wi, int C, int Ho, int Wo, int crop_h,
                         int crop_w, uc_stream_t stream);
/* Inverse of uc_convt_scatter: src NHWC [B,k*h,k*w,Cout] -> dst [B*h*w, k*k*Cout] (columns (u,v,o)). */
int uc_convt_gather(const void* src, void* dst, int dtype, int B, int h, int w, int k, int Cout, uc_stream_t stream);
/* Transposed im2col of a 3x3/pad-1 conv input for the weight-gradient GEMM:
 *   dst[(ky*3+kx)*Cin + c, p] = act(x[b, oy*stride-1+ky, ox*stride-1+kx, c]),  p = (b*Ho+oy)*Wo+ox,
 * zero outside the image and for p in [B*Ho*Wo, ld); act = ReLU when relu != 0 (the conv's ReLU-on-load). */
int uc_im2col_t(const void* x, void* dst, int dtype, int B, int H, int W, int Cin, int stride, int relu, int64_t ld,
                uc_stream_t stream);
/* Zero-stuffing (data gradient of a strided conv): dst [B,H,W,C] = src [B,h,w,C] placed at multiples of `stride`. */
int uc_dilate_nhwc(const void* src, void* dst, int dtype, int B, int h, int w, int H, int W, int C, int stride,
                   uc_stream_t stream);
/* Backward of uc_conv1x1_to4: dfeat[p,c] = sum_o dout[p,o] w[o,c] (dtype of feat); dw [4,Cin] and db [4] are
 * accumulated with fp32 atomics (zero them first).  relu_mask != 0: feat is the output of a ReLU whose backward is applied here
 * (dfeat = 0 where feat <= 0): the regressor's conv3x3 -> ReLU -> conv1x1 tail without a mask pass of its own. */
int uc_conv1x1_to4_bwd(const void* feat, int dtype, const float* w, const float* dout, void* dfeat, float* dw, float* db,
                       int64_t npix, int Cin, int relu_mask, uc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Attention dropout (training with attn_drop > 0; replaces nn.Dropout on the softmax probabilities / the dropout_p of the fused SDPA in
 * /root/reference/uniception/models/utils/transformer_blocks.py:198, 245, 251, 374, 380 and libs/croco/blocks.py:98-125).
 * No mask is stored: whether probability (b, h, q, k) is kept is a counter-based hash of (seed, b * H + h, q, k) against
 * round(drop_p * 2^32), evaluated identically by the forward kernel, by both backward kernels (which rebuild P from the LSE) and by
 * uc_attention_drop_mask.  O = (P o mask / (1 - drop_p)) V; the LSE written is that of the undropped scores.  drop_p in [0, 1).
 * ---------------------------------------------------------------------------------- */
/* uc_attention_fwd's argument list + (drop_p, seed).  bf16 (packed VT, head_dim 64) and fp32 (row-major V, head_dim <= 64). */
int uc_attention_fwd_drop(const void* Q, const void* K, const void* V, void* O, int dtype, int v_layout, int B, int H, int Nq,
                          int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                          int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                          float* lse, float drop_p, unsigned long long seed, uc_stream_t stream);
/* uc_attention_bwd's argument list + the forward's (drop_p, seed); O is the forward's (dropped) output. */
int uc_attention_bwd_drop(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                          void* dQ, void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int64_t q_sb, int64_t q_sn,
                          int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                          int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb,
                          int64_t dk_sn, int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale,
                          const int64_t* rope_qpos, const int64_t* rope_kpos, float rope_base, float rope_f0, float drop_p,
                          unsigned long long seed, uc_stream_t stream);
/* uc_attention_bwd_f32's argument list + the forward's (drop_p, seed). */
int uc_attention_bwd_f32_drop(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                              const float* LSE, float* dQ, float* dK, float* dV, float* delta, int B, int H, int Nq, int Nk,
                              int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                              int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                              int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb, int64_t dk_sn, int64_t dk_sh,
                              int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, float drop_p, unsigned long long seed,
                              uc_stream_t stream);
/* The keep mask those kernels apply, as bytes [B, H, Nq, Nk] (1 = kept): for reference implementations and tests. */
int uc_attention_drop_mask(void* mask, int B, int H, int Nq, int Nk, float drop_p, unsigned long long seed, uc_stream_t stream);

/* fp32 verification-mode attention backward: same math, row-major fp32 operands (head_dim D <= 64), no packed
 * transposes needed.  delta fp32 [B,H,Nq] is scratch. */
int uc_attention_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                         const float* LSE, float* dQ, float* dK, float* dV, float* delta, int B, int H, int Nq, int Nk,
                         int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                         int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                         int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb, int64_t dk_sn, int64_t dk_sh,
                         int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, uc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UC_HIP_H */
