// Probe: buffer_load_dwordx4 ... offen lds on gfx950 — LDS addressing above 64 KiB, zero fill of out-of-range lanes,
// soffset handling of the range check.  Build on the box: hipcc --offload-arch=gfx950 -O3 buffer_lds.hip -o /tmp/bl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned* a, unsigned nbytes, unsigned* out, unsigned lds_off, unsigned soff, unsigned oob_lane_mask_lo) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long pa = (unsigned long long)a;
    v4u srd;
    srd.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    srd.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    srd.z = __builtin_amdgcn_readfirstlane(nbytes);
    srd.w = 0x00020000u;
    unsigned off = lane * 16;
    if ((oob_lane_mask_lo >> (lane & 31)) & 1u) off = 0xffffffffu;    // out-of-range lanes
    const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + lds_off);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(off), "s"(srd), "s"(ldsaddr), "s"(so) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = ((unsigned*)(smem + lds_off))[lane * 4 + i];
    if (lane == 0) out[256] = ((unsigned*)smem)[0];   // untouched sentinel at LDS offset 0 (when lds_off > 0)
}

int main() {
    const int n = 4096;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000 + i;
    unsigned *a, *out;
    hipMalloc(&a, n * 4); hipMalloc(&out, 260 * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    struct T { const char* name; unsigned nbytes, lds_off, soff, mask; } tests[] = {
        {"plain, lds_off 0", 16384, 0, 0, 0}, {"lds_off 100 KiB", 16384, 102400, 0, 0}, {"lanes 1,3 out of range (0xffffffff)", 16384, 102400, 0, 0xa},
        {"num_records 512: lanes >= 32 out of range", 512, 0, 0, 0}, {"soffset 1024", 16384, 0, 1024, 0}, {"soffset 1024 + oob lanes", 16384, 0, 1024, 0xa},
        {"soffset 1024, num_records 1536", 1536, 0, 1024, 0}};
    for (auto& t : tests) {
        hipMemset(out, 0, 260 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, a, t.nbytes, out, t.lds_off, t.soff, t.mask);
        std::vector<unsigned> o(260);
        hipMemcpy(o.data(), out, 260 * 4, hipMemcpyDeviceToHost);
        printf("%-44s: lane0 %u %u | lane1 %u | lane3 %u | lane31 %u | lane32 %u | lane63 %u %u | sentinel %x\n", t.name, o[0], o[1], o[4], o[12], o[124], o[128],
               o[252], o[255], o[256]);
    }
    return 0;
}
