// Issue cost of the vector / transcendental / matrix instructions of the attention inner loop on gfx950, ONE wave per SIMD
// (256-thread workgroups, one per CU, all CUs busy): cycles (s_memtime) per instruction of an unrolled stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rates.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define ITERS 64

template <int T, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k(float* out, unsigned long long* cyc, const float* in) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[(threadIdx.x + i * 7) & 255];
    float s0 = v[0] * 0.001f, s1 = v[1] * 0.001f, s2 = v[2] * 0.001f, s3 = v[3] * 0.001f;
    float16_t acc0 = (float16_t)(0.f), acc1 = (float16_t)(0.f), acc2 = (float16_t)(0.f), acc3 = (float16_t)(0.f);
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)v[i]; b[i] = (__bf16)v[i + 8]; }
    unsigned u0 = 0, u1 = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (T == 0) {        // 16 independent v_exp_f32
            asm volatile(REP4("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else if constexpr (T == 1) { // 16 independent v_add_f32 (4 chains)
            asm volatile(REP4("v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %4, %1\n\tv_add_f32 %2, %4, %2\n\tv_add_f32 %3, %4, %3\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(s0));
        } else if constexpr (T == 2) { // 16 v_cvt_pk_bf16_f32
            asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %3, %2\n\t") : "+v"(u0), "+v"(u1) : "v"(v[0]), "v"(v[1]));
        } else if constexpr (T == 3) { // exp, add alternating (8 + 8)
            asm volatile(REP4("v_exp_f32 %0, %0\n\tv_add_f32 %2, %4, %2\n\tv_exp_f32 %1, %1\n\tv_add_f32 %3, %4, %3\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(s0));
        } else if constexpr (T == 4) { // 16 v_exp_f16
            asm volatile(REP4("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else if constexpr (T == 5) { // 16 MFMA, 4 accumulators
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b));
        } else if constexpr (T == 6) { // the attention slot: MFMA + 2 add + cvt + 2 exp, 16 times
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %10, %11, %0\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %1, %10, %11, %1\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %9, %6, %7\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %2, %10, %11, %2\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %3, %10, %11, %3\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %9, %6, %7\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(s0), "+v"(s1), "+v"(v[0]), "+v"(v[1]), "+v"(u0), "+v"(u1) : "v"(a), "v"(b));
        } else if constexpr (T == 7) { // MFMA + 5 plain adds, 16 times
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %10, %11, %0\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_add_f32 %8, %6, %8\n\tv_add_f32 %9, %7, %9\n\tv_add_f32 %4, %7, %4\n\t"
                              "v_mfma_f32_32x32x16_bf16 %1, %10, %11, %1\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_add_f32 %8, %6, %8\n\tv_add_f32 %9, %7, %9\n\tv_add_f32 %4, %7, %4\n\t"
                              "v_mfma_f32_32x32x16_bf16 %2, %10, %11, %2\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_add_f32 %8, %6, %8\n\tv_add_f32 %9, %7, %9\n\tv_add_f32 %4, %7, %4\n\t"
                              "v_mfma_f32_32x32x16_bf16 %3, %10, %11, %3\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_add_f32 %8, %6, %8\n\tv_add_f32 %9, %7, %9\n\tv_add_f32 %4, %7, %4\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(s0), "+v"(s1), "+v"(v[0]), "+v"(v[1]), "+v"(s2), "+v"(s3) : "v"(a), "v"(b));
        } else if constexpr (T == 8) { // MFMA + 1 exp + 2 add + 1 cvt (half the exps)
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %10, %11, %0\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_exp_f32 %6, %6\n\t"
                              "v_mfma_f32_32x32x16_bf16 %1, %10, %11, %1\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %9, %6, %7\n\tv_exp_f32 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %2, %10, %11, %2\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_exp_f32 %6, %6\n\t"
                              "v_mfma_f32_32x32x16_bf16 %3, %10, %11, %3\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %9, %6, %7\n\tv_exp_f32 %7, %7\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(s0), "+v"(s1), "+v"(v[0]), "+v"(v[1]), "+v"(u0), "+v"(u1) : "v"(a), "v"(b));
        } else if constexpr (T == 9) { // 16 v_pk_fma_f32 (4 chains)
            asm volatile(REP4("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4\n\t")
                         : "+v"(*(double*)&v[0]), "+v"(*(double*)&v[2]), "+v"(*(double*)&v[4]), "+v"(*(double*)&v[6]) : "v"(*(double*)&v[8]));
        } else if constexpr (T == 10) { // 16 v_pk_fma_f16
            asm volatile(REP4("v_pk_fma_f16 %0, %0, %4, %4\n\tv_pk_fma_f16 %1, %1, %4, %4\n\tv_pk_fma_f16 %2, %2, %4, %4\n\tv_pk_fma_f16 %3, %3, %4, %4\n\t") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(s0));
        } else if constexpr (T == 11) { // MFMA + 2 exp only
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\t"
                              "v_mfma_f32_32x32x16_bf16 %1, %6, %7, %1\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\t"
                              "v_mfma_f32_32x32x16_bf16 %2, %6, %7, %2\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\t"
                              "v_mfma_f32_32x32x16_bf16 %3, %6, %7, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(v[0]), "+v"(v[1]) : "v"(a), "v"(b));
        } else if constexpr (T == 12) { // 16 ds_read_b128 + wait (LDS latency/throughput from one wave per SIMD)
            extern __shared__ char sm[];
            float4 r0, r1, r2, r3;
            const unsigned ad = (threadIdx.x & 63) * 16;
            asm volatile(REP4("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t") "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(ad));
            v[0] += r0.x + r1.x + r2.x + r3.x;
        } else if constexpr (T == 13) { // MFMA + 2 v_exp_f16 + 2 add + cvt
            asm volatile(REP4("v_mfma_f32_32x32x16_bf16 %0, %10, %11, %0\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %1, %10, %11, %1\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %9, %6, %7\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %2, %10, %11, %2\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %8, %6, %7\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7\n\t"
                              "v_mfma_f32_32x32x16_bf16 %3, %10, %11, %3\n\tv_add_f32 %4, %6, %4\n\tv_add_f32 %5, %7, %5\n\tv_cvt_pk_bf16_f32 %9, %6, %7\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(s0), "+v"(s1), "+v"(v[0]), "+v"(v[1]), "+v"(u0), "+v"(u1) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += v[i] + acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + s0 + s1 + s2 + s3 + (float)u0 + (float)u1;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int T, int WPS>
static int run(const char* name, int per_iter) {
    float *out, *in; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * WPS * 4)); CK(hipMalloc(&in, 1024)); CK(hipMalloc(&cyc, 256 * 4 * WPS * 8));
    std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = -((i * 37) % 97) / 16.0f;
    CK(hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<T, WPS>), dim3(256), dim3(256 * WPS), T == 12 ? 4096 : 0, 0, out, cyc, in);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(256 * 4 * WPS); CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto x : c) s += (double)x; s /= c.size();
    printf("%-44s waves/SIMD %d: %8.1f ticks per block of %2d = %6.2f per instruction (per SIMD: %6.2f)\n", name, WPS, s / ITERS, per_iter, s / ITERS / per_iter, s / ITERS / per_iter / WPS);
    hipFree(out); hipFree(in); hipFree(cyc);
    return 0;
}
#define RUN(T, name, n) if (run<T, 1>(name, n)) return 1; if (run<T, 2>(name, n)) return 1;
int main() {
    RUN(0, "v_exp_f32 x16", 16)
    RUN(1, "v_add_f32 x16", 16)
    RUN(2, "v_cvt_pk_bf16_f32 x16", 16)
    RUN(3, "exp / add alternating x16", 16)
    RUN(4, "v_exp_f16 x16", 16)
    RUN(5, "MFMA 32x32x16 bf16 x16", 16)
    RUN(6, "slot: MFMA + 2 add + cvt + 2 exp  x16", 16)
    RUN(7, "slot: MFMA + 5 add  x16", 16)
    RUN(8, "slot: MFMA + 2 add + cvt + 1 exp  x16", 16)
    RUN(9, "v_pk_fma_f32 x16", 16)
    RUN(10, "v_pk_fma_f16 x16", 16)
    RUN(11, "slot: MFMA + 2 exp  x16", 16)
    RUN(12, "ds_read_b128 x16 + wait", 16)
    RUN(13, "slot: MFMA + 2 add + cvt + 2 exp_f16  x16", 16)
    return 0;
}
