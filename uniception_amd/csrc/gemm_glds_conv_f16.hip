// fp16-operand instantiation of the implicit-GEMM 3x3 convolution kernels (the prediction heads' TF32-class mode).
#include "gemm_glds_kernel.h"
void glds_launch_conv_f16(const GldsParams& p, int variant, hipStream_t st) {
    if (glds_launch_conv_res16<true>(p, variant, st)) return;
    glds_launch_variants<UC_A_CONV3X3, GLDS_EPI_ALL, true>(p, variant, st);
}
