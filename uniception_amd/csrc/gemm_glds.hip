// Launcher of the direct-to-LDS bf16 GEMM kernels (gemm_glds_kernel.h).  The kernel template is instantiated per operand
// mode and epilogue family in separate translation units (gemm_glds_*.hip): they compile in parallel and each kernel carries
// only the epilogue code its launches can reach.
#include "gemm_glds.h"

void glds_launch_dense_bf16(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_dense_f32(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_dense_all(const GldsParams& p, int variant, hipStream_t st);
void glds_launch_conv(const GldsParams& p, int variant, hipStream_t st);

int uc_launch_gemm_glds(const GldsParams& p, int variant, hipStream_t st) {
    if (p.a_mode == UC_A_CONV3X3) { glds_launch_conv(p, variant, st); return 0; }
    // a descriptor goes to a single-family kernel only when EVERY wave of the launch takes that family's epilogue
    const bool plain = p.vec_ok && p.N % 64 == 0 && p.split_k <= 1 && !p.preact && !p.dact_u && !(p.dbg & 16);
    if (plain && p.out_dtype == UC_BF16 && !p.residual) glds_launch_dense_bf16(p, variant, st);
    else if (plain && p.out_dtype == UC_F32 && (!p.residual || p.res_dtype == UC_F32) && p.act == UC_ACT_NONE && p.vt_col0 < 0) glds_launch_dense_f32(p, variant, st);
    else glds_launch_dense_all(p, variant, st);
    return 0;
}
