// Probe: cost of a 256x256-tile epilogue store tail per CU for different lane->address maps (one 16-wave WG per CU,
// 128 KiB LDS to pin one WG per CU like the GEMM).  Build: hipcc --offload-arch=gfx950 -O3 store_tail.hip -o store_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(1024) void tail(char* C, const char* R, long ldc_elems, int tiles_n, int reps) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, g = lane >> 4;
    for (int rep = 0; rep < reps; ++rep) {
    const int t = blockIdx.x + rep * gridDim.x; const long tm = t / tiles_n, tn = t % tiles_n;
    if (threadIdx.x == 5000) smem[0] = 1;
    const long row0 = tm * 256 + wr * 64, col0 = tn * 256 + wc * 64;
    uint4_t v = {(unsigned)lane, 2u, 3u, 4u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (P == 0) {          // bf16, 8 B/lane, 16 rows x 32 B per instruction (current)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long r = row0 + 16 * i + frow, c = col0 + 16 * j + 4 * g;
                *reinterpret_cast<uint2_t*>(C + (r * ldc_elems + c) * 2) = uint2_t{v.x, v.y};
            }
        } else if constexpr (P == 1) {   // bf16, 16 B/lane, 16 rows x 64 B per instruction
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                long r = row0 + 16 * i + frow, c = col0 + 32 * j + 8 * g;
                *reinterpret_cast<uint4_t*>(C + (r * ldc_elems + c) * 2) = v;
            }
        } else if constexpr (P == 2) {   // bf16, 16 B/lane, 8 rows x 128 B per instruction (full lines)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                long r = row0 + 16 * i + 8 * j + (lane >> 3), c = col0 + 8 * (lane & 7);
                *reinterpret_cast<uint4_t*>(C + (r * ldc_elems + c) * 2) = v;
            }
        } else if constexpr (P == 3 || P == 5) {   // fp32, 16 B/lane, 16 rows x 64 B (current); 5: + residual load
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long r = row0 + 16 * i + frow, c = col0 + 16 * j + 4 * g;
                uint4_t x = v;
                if constexpr (P == 5) { uint4_t y = *reinterpret_cast<const uint4_t*>(R + (r * ldc_elems + c) * 4); x += y; }
                *reinterpret_cast<uint4_t*>(C + (r * ldc_elems + c) * 4) = x;
            }
        } else if constexpr (P == 4 || P == 6) {   // fp32, 16 B/lane, 4 rows x 256 B (full lines); 6: + residual load
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long r = row0 + 16 * i + 4 * j + (lane >> 4), c = col0 + 4 * (lane & 15);
                uint4_t x = v;
                if constexpr (P == 6) { uint4_t y = *reinterpret_cast<const uint4_t*>(R + (r * ldc_elems + c) * 4); x += y; }
                *reinterpret_cast<uint4_t*>(C + (r * ldc_elems + c) * 4) = x;
            }
        } else if constexpr (P == 7) {   // bf16, whole WG cooperates: a wave writes 2 full 512-B rows per instruction
#pragma unroll
            for (int j = 0; j < 2; ++j) {   // 16 waves x 8 instr x 2 rows = 256 rows
                long r = tm * 256 + wave * 16 + (i * 2 + j) * 2 + (lane >> 5), c = tn * 256 + 8 * (lane & 31);
                *reinterpret_cast<uint4_t*>(C + (r * ldc_elems + c) * 2) = v;
            }
        }
    }
    }
}

template <int P>
static void run(const char* name, char* C, char* R, long N, int rounds, int elt, int wgs = 256, int reps = 1) {
    const int tiles_n = (int)(N / 256);
    const int tiles = wgs * rounds;
    auto k = tail<P>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(tiles), dim3(1024), 131072, 0, C, R, N, tiles_n, reps);
    hipEventRecord(e0);
    const int it = 10;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL(k, dim3(tiles), dim3(1024), 131072, 0, C, R, N, tiles_n, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it;
    const double bytes = (double)tiles * reps * 65536 * elt * ((P == 5 || P == 6) ? 2 : 1);
    printf("%-48s wgs=%3d rounds=%3d reps=%3d: %8.1f us  %6.2f us/tile-slot  %6.2f TB/s\n", name, wgs, rounds, reps, us, us / rounds / reps, bytes / us * 1e-6);
}

int main() {
    const long N = 4096; const long Mmax = 256L * 256 * 64 / (N / 256) ;   // rows for 64 rounds
    char *C, *R; hipMalloc(&C, Mmax * N * 4); hipMalloc(&R, Mmax * N * 4);
    hipMemset(C, 0, Mmax * N * 4); hipMemset(R, 0, Mmax * N * 4);
    for (int rounds : {1, 32}) {
        run<0>("bf16 8B/lane 16x32B (current)", C, R, N, rounds, 2);
        run<1>("bf16 16B/lane 16x64B", C, R, N, rounds, 2);
        run<2>("bf16 16B/lane 8x128B full lines", C, R, N, rounds, 2);
        run<7>("bf16 16B/lane 2x512B WG-cooperative rows", C, R, N, rounds, 2);
        run<3>("fp32 16B/lane 16x64B (current)", C, R, N, rounds, 4);
        run<4>("fp32 16B/lane 4x256B full lines", C, R, N, rounds, 4);
        run<5>("fp32 16x64B + residual load (current)", C, R, N, rounds, 4);
        run<6>("fp32 4x256B + residual load", C, R, N, rounds, 4);
    }
    for (int wgs : {8, 32, 128}) {
        run<0>("bf16 8B/lane 16x32B (current)", C, R, N, 1, 2, wgs, 32);
        run<1>("bf16 16B/lane 16x64B", C, R, N, 1, 2, wgs, 32);
        run<2>("bf16 16B/lane 8x128B full lines", C, R, N, 1, 2, wgs, 32);
        run<3>("fp32 16B/lane 16x64B (current)", C, R, N, 1, 4, wgs, 32);
        run<4>("fp32 16B/lane 4x256B full lines", C, R, N, 1, 4, wgs, 32);
        run<5>("fp32 16x64B + residual load (current)", C, R, N, 1, 4, wgs, 32);
        run<6>("fp32 4x256B + residual load", C, R, N, 1, 4, wgs, 32);
    }
    return 0;
}
