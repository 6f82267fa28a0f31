// Stand-alone check + timing of attn_bf16_p64_kernel (no torch): random Q / K / V, VT packed on the host, a sample of rows
// compared with an fp64 softmax, then timed.   hipcc --offload-arch=gfx950 -O3 -std=c++17 attn_p64_probe.hip -o /tmp/p64 && /tmp/p64 B H N
#include "../../uniception_amd/csrc/attention_p64.h"
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstring>
void uc_set_error(const char*, ...) {}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static int vt_key_of_pos(int pp) { const int hi = pp >> 3, j = pp & 7; return (j & 3) + 8 * (j >> 2) + 4 * hi; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, H = argc > 2 ? atoi(argv[2]) : 16, N = argc > 3 ? atoi(argv[3]) : 1024;
    const int Nk = argc > 4 ? atoi(argv[4]) : N;
    const int spike = argc > 5 ? atoi(argv[5]) : 0;      // 1: a key 6x a query (~92 in the exp2 domain: stale-maximum path); 2: 9x (~138: flagged for the fix-up kernel)
    const int D = 64, npad = (Nk + 63) / 64 * 64;
    const size_t nq_el = (size_t)B * N * H * D, nk_el = (size_t)B * Nk * H * D, nvt = (size_t)B * H * D * npad;
    std::vector<unsigned short> q(nq_el), k(nk_el), v(nk_el), vt(nvt, 0), o(nq_el);
    srand(1);
    auto rnd = [] { return ((rand() & 0xffff) / 65536.0f - 0.5f) * 4.0f; };
    for (auto& x : q) x = f2bf(rnd());
    for (auto& x : k) x = f2bf(rnd());
    for (auto& x : v) x = f2bf(rnd());
    if (spike) {   // a key far above the rest in a LATE tile: the stale maximum must be outgrown by > 2^30
        for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) {
            const int key = Nk - 1 - (h % 70);
            if (key < 0) continue;
            for (int d = 0; d < D; ++d) k[(((size_t)b * Nk + key) * H + h) * D + d] = f2bf(bf2f(q[(((size_t)b * N + (7 * h) % N) * H + h) * D + d]) * (spike == 2 ? 9.0f : 6.0f));
        }
    }
    for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int d = 0; d < D; ++d) for (int pos = 0; pos < npad; ++pos) {
        const int key = (pos & ~15) + vt_key_of_pos(pos & 15);
        vt[(((size_t)b * H + h) * D + d) * npad + pos] = key < Nk ? v[(((size_t)b * Nk + key) * H + h) * D + d] : 0;
    }
    unsigned short *dq, *dk, *dvt, *dout; float* dlse;
    CK(hipMalloc(&dq, nq_el * 2)); CK(hipMalloc(&dk, nk_el * 2)); CK(hipMalloc(&dvt, nvt * 2)); CK(hipMalloc(&dout, nq_el * 2)); CK(hipMalloc(&dlse, (size_t)B * H * N * 4));
    CK(hipMemcpy(dq, q.data(), nq_el * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, k.data(), nk_el * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvt, vt.data(), nvt * 2, hipMemcpyHostToDevice)); CK(hipMemset(dout, 0xff, nq_el * 2));
    AttnP64Params p;
    memset(&p, 0, sizeof(p));
    p.Q = dq; p.K = dk; p.V = dvt; p.O = dout; p.lse = dlse; p.B = B; p.H = H; p.Nq = N; p.Nk = Nk;
    p.q_sb = (int64_t)N * H * D; p.q_sn = H * D; p.q_sh = D; p.k_sb = (int64_t)Nk * H * D; p.k_sn = H * D; p.k_sh = D;
    p.o_sb = p.q_sb; p.o_sn = p.q_sn; p.o_sh = D; p.npad = npad; p.c = 0.125f * 1.44269504088896340736f; p.q_prescaled = getenv("P64_PRESCALED") ? 1 : 0;   // (timing only: results are then wrong by the missing factor)
    p.nq = (N + 255) / 256; p.dNq = uc_make_fastdiv((unsigned)p.nq); p.dH = uc_make_fastdiv((unsigned)H);
    const int nbh = B * H;
    const int items = nbh * p.nq;
    int grid = argc > 6 ? atoi(argv[6]) : 512; while (grid > 8 && (grid / 8) > (items + 7) / 8) grid -= 8;
    printf("B=%d H=%d Nq=%d Nk=%d items=%d grid=%d ragged=%d\n", B, H, N, Nk, items, grid, (Nk & 63) != 0);
    auto launch = [&] {
        if (Nk & 63) hipLaunchKernelGGL(attn_bf16_p64_kernel<true>, dim3(grid), dim3(256), 0, 0, p);
        else hipLaunchKernelGGL(attn_bf16_p64_kernel<false>, dim3(grid), dim3(256), 0, 0, p);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, nq_el * 2, hipMemcpyDeviceToHost));
    std::vector<float> lse((size_t)B * H * N);
    CK(hipMemcpy(lse.data(), dlse, lse.size() * 4, hipMemcpyDeviceToHost));
    // check a sample of rows (every row of the spiked queries)
    double worst = 0, worst_lse = 0; int bad = 0, checked = 0, flagged = 0;
    for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int blk = 0; blk * 64 < N; ++blk) {
        const size_t at = (((size_t)b * N + blk * 64) * H + h) * D;
        if (o[at] == 0x7fc5 && o[at + 1] == 0x7fc5) ++flagged;
    }
    for (int b = 0; b < B; b += (B > 4 ? B / 4 : 1)) for (int h = 0; h < H; h += (H > 4 ? 5 : 1)) for (int qi = 0; qi < N; ++qi) {
        if (!(qi % 37 == 0 || qi == N - 1 || qi == (7 * h) % N || (qi & 255) == 255 || (qi & 255) == 64)) continue;
        {
            const size_t at = (((size_t)b * N + (qi & ~63)) * H + h) * D;
            if (o[at] == 0x7fc5 && o[at + 1] == 0x7fc5) continue;     // block flagged for recomputation
        }
        std::vector<double> sc(Nk);
        double mx = -1e300;
        for (int key = 0; key < Nk; ++key) {
            double a = 0;
            for (int d = 0; d < D; ++d) a += (double)bf2f(q[(((size_t)b * N + qi) * H + h) * D + d]) * bf2f(k[(((size_t)b * Nk + key) * H + h) * D + d]);
            sc[key] = a * 0.125; mx = std::max(mx, sc[key]);
        }
        double lsum = 0; for (int key = 0; key < Nk; ++key) { sc[key] = exp(sc[key] - mx); lsum += sc[key]; }
        for (int d = 0; d < D; ++d) {
            double a = 0;
            for (int key = 0; key < Nk; ++key) a += sc[key] * bf2f(v[(((size_t)b * Nk + key) * H + h) * D + d]);
            a /= lsum;
            const double got = bf2f(o[(((size_t)b * N + qi) * H + h) * D + d]);
            const double err = fabs(got - a);
            worst = std::max(worst, err);
            if (!(err < 0.03)) { if (bad < 8) printf("  BAD b=%d h=%d q=%d d=%d got %f want %f\n", b, h, qi, d, got, a); ++bad; }
        }
        const double want_lse = mx + log(lsum), e2 = fabs(lse[((size_t)b * H + h) * N + qi] - want_lse);
        worst_lse = std::max(worst_lse, e2);
        ++checked;
    }
    printf("checked %d rows: max abs err %.4g (lse %.4g), bad %d; %d blocks flagged for the fix-up kernel\n", checked, worst, worst_lse, bad, flagged);
    // timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    const int iters = getenv("P64_LONG") ? atoi(getenv("P64_LONG")) : 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms * 1e-3 / iters, fl = 4.0 * B * H * (double)N * Nk * D;
#ifdef P64_TIMING
    {
        unsigned long long* dd; CK(hipMalloc(&dd, 16384 * 8)); CK(hipMemset(dd, 0, 16384 * 8));
        p.dbg = dd; launch(); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hd(16384); CK(hipMemcpy(hd.data(), dd, 16384 * 8, hipMemcpyDeviceToHost));
        const char* nm[16] = {"lookahead", "regionA", "regionB", "end_iter", "pro:end", "tail:stores", "seam", "item_setup", "pro:qfrags", "pro:qk0", "pro:max", "pro:qk1+sm", "tail:pv+sm", "tail:l", "-", "-"};
        const int items_wg = (items + grid - 1) / grid, iters_wg = items_wg * ((Nk + 63) / 64 - 1);
        for (int wg : {0, 9, 100}) if (wg < grid) {
            printf("  wg %3d (s_memtime ticks, %d items, %d loop iterations):", wg, items_wg, iters_wg);
            unsigned long long tot = 0;
            for (int i = 0; i < 16; ++i) tot += hd[wg * 16 + i];
            for (int i = 0; i < 14; ++i) printf(" %s %.0f/%s", nm[i], (double)hd[wg * 16 + i] / (i < 4 ? iters_wg : items_wg), i < 4 ? "it" : "item");
            printf("  total %llu\n", tot);
        }
        {
            unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
            for (int w = 0; w < grid; ++w) { t0 = std::min(t0, hd[8192 + 4 * w]); t1 = std::max(t1, hd[8192 + 4 * w + 1]); r0 = std::min(r0, hd[8192 + 4 * w + 2]); r1 = std::max(r1, hd[8192 + 4 * w + 3]); }
            int late = 0; double avg = 0, avgr = 0;
            for (int w = 0; w < grid; ++w) { late += hd[8192 + 4 * w + 2] - r0 > (r1 - r0) / 4; avg += (double)(hd[8192 + 4 * w + 1] - hd[8192 + 4 * w]); avgr += (double)(hd[8192 + 4 * w + 3] - hd[8192 + 4 * w + 2]); }
            int nb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, attn_bf16_p64_kernel<false>, 256, 0);
            printf("  kernel span: %llu s_memtime ticks = %llu s_memrealtime ticks (100 MHz: %.1f us) -> %.3f GHz; mean workgroup life %.0f ticks = %.1f us; %d of %d workgroups started later than a quarter into the span; occupancy API: %d blocks per CU\n",
                   t1 - t0, r1 - r0, (r1 - r0) * 0.01, (double)(t1 - t0) / ((r1 - r0) * 10.0), avg / grid, avgr / grid * 0.01, late, grid, nb);
        }
        p.dbg = nullptr;
    }
#endif
    printf("time %.1f us  %.1f TFLOP/s  (%.3f of 2500)\n", t * 1e6, fl / t / 1e12, fl / t / 2.5e15);
    return bad ? 1 : 0;
}
