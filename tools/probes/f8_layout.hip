// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, unit scales) on gfx950.
// Hypothesis h: byte e (0..31) of lane l holds  A[row = l&31][k(h,l,e)]  and  B[k(h,l,e)][col = l&31].
//   h=0: k = 32*(l>>5) + e            h=1: k = 16*(l>>5) + (e&15) + 32*(e>>4)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ int kmap(int h, int l, int e) { return h == 0 ? 32 * (l >> 5) + e : 16 * (l >> 5) + (e & 15) + 32 * (e >> 4); }

__global__ void probe(const float* A, const float* B, float* D, int h, int scale_word) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        int wa = 0, wb = 0;
        float av[4], bv[4];
        for (int e = 0; e < 4; ++e) {
            const int k = kmap(h, l, 4 * w + e);
            av[e] = A[(l & 31) * 64 + k];
            bv[e] = B[k * 32 + (l & 31)];
        }
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(av[0], av[1], wa, false);
        wa = __builtin_amdgcn_cvt_pk_fp8_f32(av[2], av[3], wa, true);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(bv[0], bv[1], wb, false);
        wb = __builtin_amdgcn_cvt_pk_fp8_f32(bv[2], bv[3], wb, true);
        a[w] = wa; b[w] = wb;
    }
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, scale_word, 0, scale_word);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = acc[r];
    }
}

int main() {
    float hA[32 * 64], hB[64 * 32], hD[32 * 32], ref[32 * 32];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) hA[i] = (float)(rand() % 9 - 4);
    for (int i = 0; i < 64 * 32; ++i) hB[i] = (float)(rand() % 7 - 3) * 0.5f;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < 64; ++k) s += hA[i * 64 + k] * hB[k * 32 + j];
            ref[i * 32 + j] = s;
        }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int h = 0; h < 2; ++h)
        for (int sw : {0x7f7f7f7f, 0}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, h, sw);
            hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
            double err = 0, mag = 0;
            for (int i = 0; i < 1024; ++i) { err += fabs(hD[i] - ref[i]); mag += fabs(ref[i]); }
            printf("hypothesis %d scale_word 0x%08x: sum|D-ref| = %g (sum|ref| = %g) D[0]=%g ref[0]=%g\n", h, sw, err, mag, hD[0], ref[0]);
        }
    return 0;
}
