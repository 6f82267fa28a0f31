// Implicit-GEMM 3x3 convolution kernels of the direct-to-LDS family (all epilogues: bf16 store with activation, residual drains).
#include "gemm_glds_kernel.h"
void glds_launch_conv(const GldsParams& p, int variant, hipStream_t st) {
    if (glds_launch_conv_res16<false>(p, variant, st)) return;
    glds_launch_variants<UC_A_CONV3X3, GLDS_EPI_ALL>(p, variant, st);
}
