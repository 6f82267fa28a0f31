// fp16-operand instantiation of the dense direct-to-LDS GEMM kernels (the prediction heads' TF32-class mode: 1x1 convolutions,
// ConvTranspose-as-GEMM, linear heads).
#include "gemm_glds_kernel.h"
void glds_launch_dense_all_f16(const GldsParams& p, int variant, hipStream_t st) { glds_launch_variants<UC_A_DENSE, GLDS_EPI_ALL, true>(p, variant, st); }
