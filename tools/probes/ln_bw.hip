// Probe: LayerNorm f32->bf16 bandwidth variants vs a plain copy kernel of the same traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
typedef float float2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { float2v_t v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v_t)); }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// copy with the same traffic: read f32 write bf16
__global__ void copy_k(const float* x, unsigned short* y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4_t v = *(const float4_t*)(x + i * 4);
        uint2 r; r.x = pk(v.x, v.y); r.y = pk(v.z, v.w);
        *(uint2*)(y + i * 4) = r;
    }
}
template <int NV, int ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void ln_k(const float* x, const float* g, const float* b, unsigned short* y, long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS_PER_WAVE;
    float4_t v[ROWS_PER_WAVE][NV];
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; ++r)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[r][i] = *(const float4_t*)(x + (row0 + r) * C + (i * 64 + lane) * 4);
    float4_t gg[NV], bb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { gg[i] = *(const float4_t*)(g + (i * 64 + lane) * 4); bb[i] = *(const float4_t*)(b + (i * 64 + lane) * 4); }
#pragma unroll
    for (int r = 0; r < ROWS_PER_WAVE; ++r) {
        float s = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = wsum(s) / C;
        float q = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) { float a = v[r][i].x - mean, bq = v[r][i].y - mean, c = v[r][i].z - mean, d = v[r][i].w - mean; q += (a * a + bq * bq) + (c * c + d * d); }
        const float rstd = rsqrtf(wsum(q) / C + 1e-6f);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            uint2 o;
            o.x = pk((v[r][i].x - mean) * rstd * gg[i].x + bb[i].x, (v[r][i].y - mean) * rstd * gg[i].y + bb[i].y);
            o.y = pk((v[r][i].z - mean) * rstd * gg[i].z + bb[i].z, (v[r][i].w - mean) * rstd * gg[i].w + bb[i].w);
            *(uint2*)(y + (row0 + r) * C + (i * 64 + lane) * 4) = o;
        }
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f(); (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
    for (int C : {1024, 768}) {
        const long rows = 32768; float *x, *g, *b; unsigned short* y;
        (void)hipMalloc(&x, rows * C * 4); (void)hipMalloc(&y, rows * C * 2); (void)hipMalloc(&g, C * 4); (void)hipMalloc(&b, C * 4);
        (void)hipMemset(x, 0, rows * C * 4); (void)hipMemset(g, 0, C * 4); (void)hipMemset(b, 0, C * 4);
        const double bytes = (double)rows * C * 6;
        float t = timeit([&] { copy_k<<<8192, 256>>>(x, y, rows * C / 4); });
        printf("C=%d copy f32->bf16           : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        if (C == 1024) {
            t = timeit([&] { ln_k<4, 1><<<rows / 4, 256>>>(x, g, b, y, rows, C); });  printf("C=%d LN 1 row/wave            : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
            t = timeit([&] { ln_k<4, 2><<<rows / 8, 256>>>(x, g, b, y, rows, C); });  printf("C=%d LN 2 rows/wave           : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
            t = timeit([&] { ln_k<4, 4><<<rows / 16, 256>>>(x, g, b, y, rows, C); }); printf("C=%d LN 4 rows/wave           : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        } else {
            t = timeit([&] { ln_k<3, 1><<<rows / 4, 256>>>(x, g, b, y, rows, C); });  printf("C=%d LN 1 row/wave            : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
            t = timeit([&] { ln_k<3, 2><<<rows / 8, 256>>>(x, g, b, y, rows, C); });  printf("C=%d LN 2 rows/wave           : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
            t = timeit([&] { ln_k<3, 4><<<rows / 16, 256>>>(x, g, b, y, rows, C); }); printf("C=%d LN 4 rows/wave           : %7.1f us %6.2f TB/s\n", C, t * 1e3, bytes / t / 1e9);
        }
    }
    return 0;
}
