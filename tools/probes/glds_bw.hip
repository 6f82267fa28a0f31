// Probe: achievable global->LDS DMA fill rate per CU from an L2-resident source, vs waves/CU and queue depth.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// each wave: per iteration issues D 1-KiB DMA instrs; keeps DEPTH iterations in flight
template <int D, int DEPTH>
__global__ void k(const char* src, size_t span_mask, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    char* base = smem + wave * (D * DEPTH * 1024);
    size_t off = ((size_t)blockIdx.x * 7919 * 1024 + (size_t)wave * 65536) & span_mask;
    for (int it = 0; it < iters; ++it) {
        const int slot = it % DEPTH;
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const char* g = src + ((off + (size_t)q * 1024) & span_mask) + lane * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(base + (slot * D + q) * 1024), 16, 0, 0);
        }
        off = (off + (size_t)nw * D * 1024) & span_mask;
        wait_vmcnt<D*(DEPTH - 1)>();
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((float*)smem)[0];
}

template <int D, int DEPTH>
void run(const char* src, size_t span, int nwaves, float* sink) {
    const int iters = 2000;
    const size_t lds = (size_t)nwaves * D * DEPTH * 1024;
    auto fn = k<D, DEPTH>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fn<<<256, nwaves * 64, lds>>>(src, span - 1, 50, sink);
    hipEventRecord(e0);
    fn<<<256, nwaves * 64, lds>>>(src, span - 1, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = 256.0 * nwaves * D * 1024.0 * iters;
    printf("span=%5zuKB waves/CU=%2d D=%d depth=%d in-flight/CU=%4dKB : %7.1f GB/s/CU  %6.2f TB/s chip  (err=%d)\n", span >> 10, nwaves, D, DEPTH,
           nwaves * D * DEPTH, bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12, (int)hipGetLastError());
}

int main() {
    char* src; float* sink;
    hipMalloc(&src, (size_t)1 << 30); hipMemset(src, 1, (size_t)1 << 30); hipMalloc(&sink, 4096);
    for (size_t span : {(size_t)1 << 21, (size_t)1 << 24, (size_t)1 << 27, (size_t)1 << 30}) {
        for (int nw : {4, 8, 16}) {
            run<4, 1>(src, span, nw, sink); run<4, 2>(src, span, nw, sink); run<8, 2>(src, span, nw, sink);
            if (nw <= 8) run<8, 4>(src, span, nw, sink);
        }
    }
    return 0;
}
