// Probe (round 4): what an LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave-instruction) costs per CU as a function of how its 64
// lanes' 16-byte chunks are laid out in global memory: ONE contiguous KiB, 8 rows x 128 B (a 64-deep bf16 K-step), 16 rows x 64 B
// (a 32-deep K-step), 4 rows x 256 B (a 128-deep K-step).  Source = an L2-resident window per workgroup (the GEMM's operand panels
// are L2 hits), rows a K = 1024 pitch (2 KiB) apart.  Reports GB/s per CU and cycles per piece per CU at 2.4 GHz nominal.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_seg tools/probes/dma_seg.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// SEG = bytes per row segment (64 / 128 / 256 / 1024); a piece = 1024 / SEG rows, `pitch` bytes apart
template <int SEG, int D, int DEPTH>
__global__ void k(const char* src, size_t window, int pitch, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    char* base = smem + wave * (D * DEPTH * 1024);
    constexpr int LPR = SEG / 16;                 // lanes per row segment
    const int rows = 1024 / SEG;                  // rows per piece
    const size_t lane_off = (size_t)(lane / LPR) * pitch + (size_t)(lane % LPR) * 16;
    const char* wbase = src + (size_t)blockIdx.x * window;
    // the window holds (window / pitch) rows of `pitch` bytes; a piece advances by `rows` rows, a K-step by SEG bytes inside the rows
    const size_t nrows = window / pitch;
    size_t row = (size_t)wave * rows, col = 0;
    for (int it = 0; it < iters; ++it) {
        const int slot = it % DEPTH;
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const char* g = wbase + ((row + (size_t)q * nw * rows) % nrows) * pitch + col + lane_off;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(base + (slot * D + q) * 1024), 16, 0, 0);
        }
        row += (size_t)D * nw * rows;
        if (row >= nrows) { row -= nrows; col = (col + SEG) % (size_t)pitch; }
        wait_vmcnt<D*(DEPTH - 1)>();
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((float*)smem)[0];
}

template <int SEG, int D, int DEPTH>
void run(const char* src, size_t window, int pitch, int nwaves, float* sink) {
    const int iters = 4000;
    const size_t lds = (size_t)nwaves * D * DEPTH * 1024;
    auto fn = k<SEG, D, DEPTH>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fn<<<256, nwaves * 64, lds>>>(src, window, pitch, 100, sink);
    hipEventRecord(e0);
    fn<<<256, nwaves * 64, lds>>>(src, window, pitch, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double pieces_per_cu = (double)nwaves * D * iters;
    const double bytes = 256.0 * pieces_per_cu * 1024.0;
    printf("seg=%4dB pitch=%5d window=%4zuKB waves/CU=%2d D=%d depth=%d : %7.1f GB/s/CU %6.2f TB/s chip  %6.1f cyc@2.4GHz per piece per CU (err=%d)\n",
           SEG, pitch, window >> 10, nwaves, D, DEPTH, bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / pieces_per_cu, (int)hipGetLastError());
}

int main() {
    char* src; float* sink;
    const size_t total = (size_t)1 << 30;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&sink, 4096);
    // window per workgroup: 64 KiB (all 256 windows = 16 MiB: L2 + MALL resident) and 2 MiB (512 MiB: streams from HBM / MALL)
    for (size_t window : {(size_t)64 << 10, (size_t)2 << 20}) {
        for (int nw : {4, 8, 16}) {
            run<1024, 4, 2>(src, window, 1024, nw, sink);
            run<256, 4, 2>(src, window, 2048, nw, sink);
            run<128, 4, 2>(src, window, 2048, nw, sink);
            run<64, 4, 2>(src, window, 2048, nw, sink);
            if (nw <= 8) { run<128, 8, 2>(src, window, 2048, nw, sink); run<64, 8, 2>(src, window, 2048, nw, sink); }
        }
    }
    return 0;
}
