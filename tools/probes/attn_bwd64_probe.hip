// Stand-alone timing of attn_bwd_dkv64_kernel (no torch; correctness is tools/probes/attn_bwd64_check.py's job): random operands,
// plausible start values, time per launch, and with -DB64_TIMING per-section s_memtime ticks of wave 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DB64_TIMING] [-DB64_VARIANT=n] attn_bwd64_probe.hip -o /tmp/b64 && /tmp/b64 B H N
#include "../../uniception_amd/csrc/attention_bwd.hip"
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <cstring>
void uc_set_error(const char*, ...) {}
std::atomic<int> g_uc_attn_bwd64{2};
static UcKnobs g_k;
const UcKnobs& uc_knobs() { return g_k; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 16, N = argc > 3 ? atoi(argv[3]) : 1024, D = 64;
    const size_t n_el = (size_t)B * N * H * D;
    std::vector<unsigned short> h(n_el);
    srand(1);
    for (auto& x : h) x = f2bf(((rand() & 0xffff) / 65536.0f - 0.5f) * 2.0f);
    unsigned short* buf[8];
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&buf[i], n_el * 2)); CK(hipMemcpy(buf[i], h.data(), n_el * 2, hipMemcpyHostToDevice)); }
    const int nq_pad = (N + 127) / 128 * 128;
    std::vector<float> aux((size_t)B * H * 2 * nq_pad);
    for (size_t i = 0; i < aux.size(); ++i) aux[i] = ((i / nq_pad) & 1) ? -0.01f : -12.0f;
    float* daux; CK(hipMalloc(&daux, aux.size() * 4)); CK(hipMemcpy(daux, aux.data(), aux.size() * 4, hipMemcpyHostToDevice));
    AttnBwdParams p;
    memset(&p, 0, sizeof(p));
    p.Q = (bf16_t*)buf[0]; p.K = (bf16_t*)buf[1]; p.V = (bf16_t*)buf[2]; p.O = (bf16_t*)buf[3]; p.dO = (bf16_t*)buf[4];
    p.dQ = (bf16_t*)buf[5]; p.dK = (bf16_t*)buf[6]; p.dV = (bf16_t*)buf[7];
    p.aux = daux; p.delta = daux; p.nq_pad = nq_pad; p.B = B; p.H = H; p.Nq = N; p.Nk = N; p.scale = 0.125f;
    p.q_sb = p.k_sb = p.v_sb = p.o_sb = p.dq_sb = p.dk_sb = p.dv_sb = (int64_t)N * H * D;
    p.q_sn = p.k_sn = p.v_sn = p.o_sn = p.dq_sn = p.dk_sn = p.dv_sn = H * D;
    p.q_sh = p.k_sh = p.v_sh = p.o_sh = p.dq_sh = p.dk_sh = p.dv_sh = D;
    const unsigned items = (unsigned)(((N + 255) / 256) * H * B); const unsigned grid = std::min(argc > 4 ? (unsigned)atoi(argv[4]) : 256u, (items + 7) / 8 * 8);
    auto launch = [&] { hipLaunchKernelGGL(attn_bwd_dkv64_kernel, dim3(grid), dim3(256), 0, 0, p); };
    float* dlse; CK(hipMalloc(&dlse, (size_t)B * H * N * 4));
    { std::vector<float> l((size_t)B * H * N, 8.0f); CK(hipMemcpy(dlse, l.data(), l.size() * 4, hipMemcpyHostToDevice)); }
    p.LSE = dlse;
    const unsigned qitems = (unsigned)(((N + 255) / 256) * H * B), qgrid = std::min(256u, (qitems + 7) / 8 * 8);
    auto launch_q = [&] { hipLaunchKernelGGL(attn_bwd_dq64_kernel, dim3(qgrid), dim3(256), 0, 0, p); };
    auto launch_q_old = [&] { hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(((N + 127) / 128) * H * B)), dim3(256), 0, 0, p); };
    auto launch_old = [&] { hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(((N + 127) / 128) * H * B)), dim3(256), 0, 0, p); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fl = 4.0 * 2.0 * B * H * (double)N * N * D;
    for (int which = 0; which < 4; ++which) {
        auto go = [&] { which == 0 ? launch_old() : which == 1 ? launch() : which == 2 ? launch_q_old() : launch_q(); };
        for (int i = 0; i < 3; ++i) go();
        CK(hipDeviceSynchronize());
        const int iters = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) go();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double t = ms * 1e-3 / iters;
        const double f2 = which < 2 ? fl : fl * 0.75;
        const char* nm4[4] = {"dkv32", "dkv64", "dq32", "dq64"};
        printf("B=%d H=%d N=%d %s: %.1f us  %.1f TFLOP/s executed (%.3f of 2500)\n", B, H, N, nm4[which], t * 1e6, f2 / t / 1e12, f2 / t / 2.5e15);
    }
#ifdef B64_TIMING
    {
        unsigned long long* dd; const size_t nd = 65536 + 4 * (size_t)grid + 16;
        CK(hipMalloc(&dd, nd * 8)); CK(hipMemset(dd, 0, nd * 8));
        p.dbg = dd;
        launch(); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hd(nd); CK(hipMemcpy(hd.data(), dd, nd * 8, hipMemcpyDeviceToHost));
        const char* nm[8] = {"loop-top", "A0", "B0", "wait+barrier", "A1", "B1", "prologue+seams", "tail+readout"};
        const int nt = (N + 63) / 64;
        for (unsigned wg : {0u, 9u, 100u, 255u}) if (wg < grid && wg < 8192) {
            printf("  wg %4u (s_memtime ticks):", wg);
            const int items_wg = (items + grid - 1) / grid; for (int i = 0; i < 8; ++i) printf(" %s %.0f%s", nm[i], (double)hd[wg * 8 + i] / (i < 6 ? nt * items_wg : items_wg), i < 6 ? "/tile" : "/item");
            printf("\n");
        }
        unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0; double avg = 0, avgr = 0;
        for (unsigned w = 0; w < grid; ++w) { t0 = std::min(t0, hd[65536 + 4 * w]); t1 = std::max(t1, hd[65536 + 4 * w + 1]); r0 = std::min(r0, hd[65536 + 4 * w + 2]); r1 = std::max(r1, hd[65536 + 4 * w + 3]);
            avg += (double)(hd[65536 + 4 * w + 1] - hd[65536 + 4 * w]); avgr += (double)(hd[65536 + 4 * w + 3] - hd[65536 + 4 * w + 2]); }
        printf("  kernel span %.1f us, %.3f GHz; mean workgroup life %.0f ticks = %.1f us\n", (r1 - r0) * 0.01, (double)(t1 - t0) / ((r1 - r0) * 10.0), avg / grid, avgr / grid * 0.01);
    }
#endif
    return 0;
}
