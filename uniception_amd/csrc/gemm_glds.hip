// Launcher of the direct-to-LDS bf16 GEMM kernels (gemm_glds_kernel.h).  The kernel template is instantiated per operand
// mode and epilogue family in separate translation units (gemm_glds_*.hip): they compile in parallel and each kernel carries
// only the epilogue code its launches can reach.
#include "gemm_glds.h"
#include "knobs.h"

void glds_launch_dense_bf16(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_dense_f32(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_dense_bs(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_dense_all(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_conv(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_conv_f16(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_dense_all_f16(const GldsParams& p, int variant, hipStream_t st);

int uc_launch_gemm_glds(const GldsParams& p, int variant, hipStream_t st, bool auto_variant) {
    if (p.f16) {      // fp16 operands: one family (EPI_ALL), 16-wave / co-resident tiles only
        if (variant == 6 || variant == 7) variant = 2;
        if (p.a_mode == UC_A_CONV3X3) glds_launch_conv_f16(p, variant, st); else glds_launch_dense_all_f16(p, variant, st);
        return 0;
    }
    if (p.a_mode == UC_A_CONV3X3) { glds_launch_conv(p, variant, st); return 0; }
    // a descriptor goes to a single-family kernel only when EVERY wave of the launch takes that family's epilogue
    const bool plain = p.vec_ok && p.N % 64 == 0 && p.split_k <= 1 && !p.preact && !p.dact_u && !UC_DBG(p, 16);
    // 256x256 tiles: the eight-wave form (variant 6, 128x64 per wave, next K-chunk's fragments register-resident) where it wins.
    // UC_GEMM_8WAVE: 0 off, 1 bf16-store family (default), 2 every family, 3 bf16-store family + bf16 residual stream.
    const int eight = uc_knobs().gemm_8wave;
    // bf16 residual stream (out bf16 + bf16 residual and / or row statistics): the residual family's drain, 2 + 2 bytes per element
    const bool bf16_stream = plain && p.out_dtype == UC_BF16 && p.act == UC_ACT_NONE && p.vt_col0 < 0 && p.rope_cols <= 0 && !p.ln_stats && !p.ln_partial &&
                             !p.residual2 && ((p.residual && p.res_dtype == UC_BF16) || p.stats_out);
    const bool bf16_fam = plain && p.out_dtype == UC_BF16 && !p.residual && !bf16_stream;
    if (auto_variant && variant == 2 && p.M % 8 == 0 && p.N % 8 == 0 &&
        (eight == 2 || ((eight == 1 || eight == 3) && bf16_fam) || (eight == 3 && bf16_stream))) variant = 6;
    // ... or the four-wave form (variant 7: 128x128 wave tiles, accumulators in AGPRs, hand-scheduled K-loop).
    // UC_GEMM_4WAVE: 0 off, 1 bf16-store family, 2 + bf16 residual stream, 3 every family.
    const int four = uc_knobs().gemm_4wave;
    const bool f32_fam = plain && p.out_dtype == UC_F32 && (!p.residual || p.res_dtype == UC_F32) && p.act == UC_ACT_NONE && p.vt_col0 < 0;
    if (auto_variant && (variant == 2 || variant == 6) && four > 0 && p.K >= uc_knobs().gemm_4wave_min_k &&
        (four >= 3 || (bf16_fam && four >= 1) || (bf16_stream && four >= 2) || (f32_fam && four >= 3))) variant = 7;
    if (bf16_fam) glds_launch_dense_bf16(p, variant, st);
    else if (bf16_stream || (plain && p.out_dtype == UC_F32 && (!p.residual || p.res_dtype == UC_F32) && p.act == UC_ACT_NONE && p.vt_col0 < 0)) {
        // The fp32 epilogues (residual read + fp32 store + bf16 twin) move 4-5x the bytes of a bf16 store and all CUs reach
        // them together: an HBM burst with idle matrix pipes.  Launches long enough to amortise the ramp (>= 6 tiles per CU)
        // start their first round of workgroups in 8 phase groups 1.5 us apart (by row panel, see the kernel):
        // encoder proj 480 -> 432 us, fc2 1095 -> 1043 us; neutral-to-worse for shorter launches, hence the threshold.
        GldsParams q = p;
        // (bf16 residual stream, round 4: 4 bytes per element in the epilogue — the stagger measures 316 vs 322 us AGAINST it on the encoder's
        //  proj GEMM: off for that family)
        if (q.stagger < 0) q.stagger = (!bf16_stream && (variant == 2 || variant == 6 || variant == 7) && (int64_t)ceil_div64(p.M, 256) * ceil_div64(p.N, 256) >= 6 * 256) ? 150 : 0;
        if (bf16_stream) glds_launch_dense_bs(q, variant, st);
        else glds_launch_dense_f32(q, variant, st);
        return 0;
    }
    else glds_launch_dense_all(p, variant, st);
    return 0;
}
