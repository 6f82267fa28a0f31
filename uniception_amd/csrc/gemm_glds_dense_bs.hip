// Dense direct-to-LDS GEMM kernels, epilogue family GLDS_EPI_BS: the bf16 residual stream (see gemm_glds_kernel.h).
#include "gemm_glds_kernel.h"
void glds_launch_dense_bs(const GldsParams& p, int variant, hipStream_t st) { glds_launch_variants<UC_A_DENSE, GLDS_EPI_BS>(p, variant, st); }
