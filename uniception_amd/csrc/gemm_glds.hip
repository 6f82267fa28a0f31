// Dense bf16 GEMM for gfx950 with direct global->LDS staging (global_load_lds_dwordx4, 1 KiB per wave-instruction).
//
//   C[M,N] = epilogue(A[M,K] . W[N,K]^T),  K % 64 == 0, A/W bf16 row-major.
//
// Tile geometry: BM x BN x 64 per workgroup, every wavefront owns a 64x64 sub-tile (4x4 MFMA 16x16x32 fragments).
// LDS image of a stage: (BM + BN) rows x 128 B.  The DMA writes lane-linear (wave base + lane*16), so the
// XOR swizzle that makes the ds_read_b128 fragment loads conflict-free (chunk ^= (row>>1)&7) is applied to the
// per-lane SOURCE address; the fragment reads apply the same involution.  Two stages; the loads of K-step t+1
// are issued before the MFMAs of step t and drained by the barrier that ends the step.
//
// Operand orientation: for ordinary tiles the MFMA computes C^T fragments (first operand = W rows), so a lane owns
// 4 CONSECUTIVE output columns of one row: bias/residual/stores are 8/16-byte vector accesses and, with the W-row
// permutation 16*(a>>2)+4j+(a&3), a lane's 16 values are 16 consecutive columns.  Tiles that take the RoPE epilogue
// keep the identity column map (partner channel d+16 = next fragment, same lane/register); V tiles that are written
// in the packed VT layout use the un-swapped orientation (a lane owns 4 consecutive TOKENS of one channel).
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

#include "gemm_glds.h"
#include <stdlib.h>
#include <type_traits>

// erf-GELU of the bf16 MFMA path.  libm's erff is a branchy two-range evaluation (~50 VALU ops per element once both
// sides of the branch run in a wave); with only 16 K-steps per fc1 tile that epilogue cost as much as the MFMA loop.
// Abramowitz-Stegun 7.1.26: erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1/(1 + p z), |error| <= 1.5e-7 — at fp32
// rounding level and three orders below the bf16 output's own rounding.  (The fp32 verification kernel keeps erff.)
__device__ __forceinline__ float glds_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);            // erf(|x|/sqrt2)
    const float erf_s = __builtin_copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf_s);
}
// d/dx of the activation at pre-activation x: GELU' = Phi(x) + x phi(x); ReLU' = [x > 0]
__device__ __forceinline__ float glds_dact(float x, int act) {
    if (act == UC_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);      // exp(-x^2/2)
    const float erf_s = __builtin_copysignf(fmaf(-poly * t, e, 1.0f), x);
    return fmaf(0.5f, erf_s, 0.5f) + x * e * 0.39894228040143267794f;
}
__device__ __forceinline__ float glds_act(float v, int act) {
    if (act == UC_ACT_GELU_ERF) return glds_gelu(v);
    if (act == UC_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

__device__ __forceinline__ int glds_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// 128 zero bytes: DMA source for im2col positions that fall into the convolution's zero padding
__device__ uint4 g_zero_chunk[8];

__device__ __forceinline__ uint4 glds_relu_bf16x8(uint4 v) {
    unsigned* q = reinterpret_cast<unsigned*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned neg = (q[i] >> 15) & 0x00010001u;
        q[i] &= ~(neg * 0xffffu);
    }
    return v;
}

// One 1-KiB LDS-DMA piece issued through inline asm.  With the __builtin form hipcc tracks the DMA as an LDS write it
// cannot disambiguate from the fragment ds_reads of the OTHER stage buffer and inserts `s_waitcnt vmcnt(0)` in front of
// them — the whole global->LDS latency is then exposed in every K-step (measured: matrix pipe 39 % busy, 57 % of wave
// cycles parked).  The asm form is invisible to that pass; completion is enforced by hand with counted s_waitcnt
// vmcnt(N) + s_barrier (see the K-loops).  M0 carries the wave-uniform LDS byte address; it is compiler-reserved, so it
// is saved and restored inside the statement; s_nop 0 covers the M0-write -> LDS-DMA hazard.
__device__ __forceinline__ void dma16_to_lds(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// Shared epilogue: bias -> activation -> fused RoPE-2D -> residual(s) -> store, or the packed-VT store of V tiles.
template <int FA>
__device__ __forceinline__ void glds_epilogue(const GldsParams& p, float4_t (&acc)[FA][4], int mode, int64_t wave_m,
                                              int64_t wave_n, int lane, int ksplit) {
    const int frow = lane & 15;
    // =============================== epilogue ===============================
    const int g = lane >> 4;
    if (wave_n >= p.N) return;

    if (mode == 2) {
        // acc[i][j][r]: token row m = wave_m + 16i + 4g + r, channel column wave_n + 16j + frow
        const int head = (int)((wave_n - p.vt_col0) >> 6);
        const int nheads = (int)((p.N - p.vt_col0) >> 6);
        const bool aligned = (p.vt_ntok & 15) == 0;
        float bcol[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bcol[j] = p.bias ? p.bias[wave_n + 16 * j + frow] : 0.f;
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int64_t mb = wave_m + 16 * i + 4 * g;
            if (aligned) {
                if (mb >= p.M) continue;
                const int b = (int)(mb / p.vt_ntok);
                const int tok = (int)(mb % p.vt_ntok);
                const int pos = (tok & ~15) + ((g & 1) << 3) + ((g >> 1) << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = 16 * j + frow;
                    bf16_t* dst = p.vt_out + (((int64_t)b * nheads + head) * 64 + d) * p.vt_npad + pos;
                    uint2 pk;
                    pk.x = pack_bf16x2(acc[i][j][0] + bcol[j], acc[i][j][1] + bcol[j]);
                    pk.y = pack_bf16x2(acc[i][j][2] + bcol[j], acc[i][j][3] + bcol[j]);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = mb + r;
                    if (m >= p.M) continue;
                    const int b = (int)(m / p.vt_ntok);
                    const int tok = (int)(m % p.vt_ntok);
                    const int w = tok & 15;
                    const int pos = (tok & ~15) + (((w >> 2) & 1) << 3) + (w & 3) + ((w >> 3) << 2);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        p.vt_out[(((int64_t)b * nheads + head) * 64 + 16 * j + frow) * p.vt_npad + pos] =
                            f32_to_bf16(acc[i][j][r] + bcol[j]);
                }
            }
        }
        return;
    }

    // swapped modes: acc[i][j][r]: row m = wave_m + 16i + frow; 4 consecutive columns wave_n + 16j + 4g + r
    float b4[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t nb = wave_n + 16 * j + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[j][r] = 0.f;
        if (p.bias) {
            if (p.vec_ok && nb + 3 < p.N) {
                const float4_t bb = *reinterpret_cast<const float4_t*>(p.bias + nb);
                b4[j][0] = bb.x; b4[j][1] = bb.y; b4[j][2] = bb.z; b4[j][3] = bb.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[j][r] = (nb + r < p.N) ? p.bias[nb + r] : 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const int64_t m = wave_m + 16 * i + frow;
        if (m >= p.M) continue;
        float v[4][4];
        if (p.split_k > 1) {   // partial product of one K slice -> its own [M, ldc] slab of the workspace, nothing else
            float* slab = (float*)p.C + (int64_t)ksplit * p.M * p.ldc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t nb = wave_n + 16 * j + 4 * g;
                if (p.vec_ok && nb + 3 < p.N) {
                    *reinterpret_cast<float4_t*>(slab + m * p.ldc + nb) = acc[i][j];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) slab[m * p.ldc + nb + r] = acc[i][j][r];
                }
            }
            continue;
        }
        if (p.preact) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t nb = wave_n + 16 * j + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nb + r >= p.N) continue;
                    const float u = acc[i][j][r] + b4[j][r];
                    if (p.out_dtype == UC_F32) ((float*)p.preact)[m * p.ldc + nb + r] = u;
                    else ((bf16_t*)p.preact)[m * p.ldc + nb + r] = f32_to_bf16(u);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = glds_act(acc[i][j][r] + b4[j][r], p.act);
        if (mode == 1) {
            int py = (int)p.rope_pos[m * 2 + 0];
            int px = (int)p.rope_pos[m * 2 + 1];
            py = min(max(py, 0), p.rope_npos - 1);
            px = min(max(px, 0), p.rope_npos - 1);
            const float4_t* ty = reinterpret_cast<const float4_t*>(p.rope_table + py * 16 + 4 * g);
            const float4_t* tx = reinterpret_cast<const float4_t*>(p.rope_table + px * 16 + 4 * g);
            const float4_t cy0 = ty[0], cy1 = ty[1], cx0 = tx[0], cx1 = tx[1];  // (cos,sin) x 4
            const float cyc[4] = {cy0.x, cy0.z, cy1.x, cy1.z}, cys[4] = {cy0.y, cy0.w, cy1.y, cy1.w};
            const float cxc[4] = {cx0.x, cx0.z, cx1.x, cx1.z}, cxs[4] = {cx0.y, cx0.w, cx1.y, cx1.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u0 = v[0][r], w0 = v[1][r], u1 = v[2][r], w1 = v[3][r];
                v[0][r] = u0 * cyc[r] - w0 * cys[r];
                v[1][r] = w0 * cyc[r] + u0 * cys[r];
                v[2][r] = u1 * cxc[r] - w1 * cxs[r];
                v[3][r] = w1 * cxc[r] + u1 * cxs[r];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t nb = wave_n + 16 * j + 4 * g;
            if (nb >= p.N) continue;
            const bool full = p.vec_ok && nb + 3 < p.N;
            if (p.residual) {
                if (full && p.res_dtype == UC_F32) {
                    const float4_t r4 = *reinterpret_cast<const float4_t*>((const float*)p.residual + m * p.ldr + nb);
                    v[j][0] += r4.x; v[j][1] += r4.y; v[j][2] += r4.z; v[j][3] += r4.w;
                    if (p.residual2) {
                        const float4_t s4 = *reinterpret_cast<const float4_t*>((const float*)p.residual2 + m * p.ldr + nb);
                        v[j][0] += s4.x; v[j][1] += s4.y; v[j][2] += s4.z; v[j][3] += s4.w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (nb + r >= p.N) continue;
                        const int64_t idx = m * p.ldr + nb + r;
                        v[j][r] += p.res_dtype == UC_F32 ? ((const float*)p.residual)[idx] : bf16_to_f32(((const bf16_t*)p.residual)[idx]);
                        if (p.residual2)
                            v[j][r] += p.res_dtype == UC_F32 ? ((const float*)p.residual2)[idx] : bf16_to_f32(((const bf16_t*)p.residual2)[idx]);
                    }
                }
            }
            if (p.dact_u) {   // fused activation backward: out = v * act'(u)
                float u4[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) {
                    const uint2 uu = *reinterpret_cast<const uint2*>(p.dact_u + m * p.ldc + nb);
                    u4[0] = __uint_as_float(uu.x << 16); u4[1] = __uint_as_float(uu.x & 0xffff0000u);
                    u4[2] = __uint_as_float(uu.y << 16); u4[3] = __uint_as_float(uu.y & 0xffff0000u);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) u4[r] = bf16_to_f32(p.dact_u[m * p.ldc + nb + r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] *= glds_dact(u4[r], p.dact_act);
            }
            if (full) {
                if (p.out_dtype == UC_F32) {
                    *reinterpret_cast<float4_t*>((float*)p.C + m * p.ldc + nb) = (float4_t){v[j][0], v[j][1], v[j][2], v[j][3]};
                } else {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[j][0], v[j][1]);
                    pk.y = pack_bf16x2(v[j][2], v[j][3]);
                    *reinterpret_cast<uint2*>((bf16_t*)p.C + m * p.ldc + nb) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nb + r >= p.N) continue;
                    if (p.out_dtype == UC_F32) ((float*)p.C)[m * p.ldc + nb + r] = v[j][r];
                    else ((bf16_t*)p.C)[m * p.ldc + nb + r] = f32_to_bf16(v[j][r]);
                }
            }
        }
    }
}

// BM_ x BN_ workgroup tile, WAVES_M x WAVES_N wavefronts; a wave owns (16*FA) x 64 outputs, FA = BM_/WAVES_M/16.
template <int BM_, int BN_, int WAVES_M, int WAVES_N, int STAGES, int A_MODE>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_glds_kernel(GldsParams p) {
    static_assert(BN_ / WAVES_N == 64, "a wave owns 64 output columns (one 64-wide head)");
    constexpr int WTM = BM_ / WAVES_M;
    constexpr int FA = WTM / 16;          // A-row fragments per wave (4 or 8)
    static_assert(WTM % 16 == 0 && (FA == 4 || FA == 8), "wave tile rows");
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int STAGE_BYTES = (BM_ + BN_) * 128;
    constexpr int NI = (BM_ + BN_) / 8;   // 1-KiB DMA instructions per stage
    constexpr int PER = NI / NW;          // per wave
    static_assert(NI % NW == 0, "DMA instructions must split evenly over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    const int nwg = p.tiles_m * p.tiles_n;
    const int ksplit = (int)blockIdx.x / nwg;                 // split-K slice (0 when split_k == 1)
    const int t = glds_xcd_remap((int)blockIdx.x - ksplit * nwg, nwg);
    // Tile order inside an XCD's run: groups of GM row panels swept column by column, so the ~32 tiles an XCD runs
    // concurrently form a GM x (32/GM) block that shares GM A-panels and 32/GM W-panels in its L2 (a plain row-major
    // order shares 2 A-panels but streams ALL of W through every pair of row panels: 2.4x algorithmic fetch traffic).
    int tm, tn;
    {
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int grp = t / per_group, within = t - grp * per_group;
        const int first_m = grp * GM;
        const int gsz = min(GM, p.tiles_m - first_m);
        tm = first_m + within % gsz;
        tn = within / gsz;
    }
    const int64_t m0 = (int64_t)tm * BM_;
    const int64_t n0 = (int64_t)tn * BN_;
    const int64_t wave_m = m0 + wr * WTM;
    const int64_t wave_n = n0 + wc * 64;

    // ---- per-wave epilogue mode (wave-uniform) ----
    const bool is_vt = A_MODE == UC_A_DENSE && p.vt_col0 >= 0 && wave_n >= p.vt_col0;        // conv tiles take neither epilogue
    const bool is_rope = A_MODE == UC_A_DENSE && !is_vt && p.rope_cols > 0 && wave_n < p.rope_cols;
    const int mode = is_vt ? 2 : (is_rope ? 1 : 0);

    // ---- DMA source pointers: instruction I = wave*PER + q covers combined-tile rows [8I, 8I+8) ----
    // dense: src[q] + k0.   conv: A rows are gathered per K-step from the 3x3 window (one tap per 64-channel K-step,
    // because Cin % 64 == 0); positions inside the zero padding read from g_zero_chunk instead.
    // Dense: one 64-bit source pointer per instruction.  Conv: two 32-bit words per instruction — A rows keep the packed
    // window origin (oy*s | ox*s << 16) and the image's first pixel index, W rows a 32-bit element offset; the lane's
    // channel chunk is re-derived — so the 16-wave 256x256 tile stays inside its 128-VGPR budget.
    const bf16_t* src[A_MODE == UC_A_DENSE ? PER : 1];
    int st0[A_MODE == UC_A_DENSE ? 1 : PER], st1[A_MODE == UC_A_DENSE ? 1 : PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int rr = (wave * PER + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((rr >> 1) & 7);   // logical chunk stored at physical chunk (lane&7) of row rr
        if (rr < BM_) {
            int64_t m = m0 + rr;
            if (m >= p.M) m = p.M - 1;
            if constexpr (A_MODE == UC_A_DENSE) {
                src[q] = p.A + m * p.lda + c * 8;
            } else {
                const int ox = (int)(m % p.cWo);
                const int oy = (int)((m / p.cWo) % p.cHo);
                const int b = (int)(m / ((int64_t)p.cWo * p.cHo));
                st0[q] = (oy * p.cStride) | ((ox * p.cStride) << 16);
                st1[q] = b * p.cH * p.cW;
            }
        } else {
            int64_t n = n0 + (rr - BM_);
            if (n >= p.N) n = p.N - 1;
            if constexpr (A_MODE == UC_A_DENSE) {
                src[q] = p.W + n * p.K + c * 8;
            } else {
                st0[q] = (int)(n * p.K + c * 8);      // < 2^31 elements (checked by the launcher)
                st1[q] = 0;
            }
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;   // LDS byte address of the dynamic region
    auto issue_stage = [&](int stage, int64_t k0) {
        int ky = 0, kx = 0, ch0 = 0;
        if (A_MODE != UC_A_DENSE) {
            const int tap = (int)(k0 / p.cCin);
            ch0 = (int)(k0 % p.cCin);
            ky = tap / 3; kx = tap - 3 * ky;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const bf16_t* g;
            if constexpr (A_MODE == UC_A_DENSE) {
                g = src[q] + k0;
            } else {
                const bool is_a = (wave * PER + q) * 8 < BM_;   // wave-uniform: an instruction is all-A or all-W
                if (is_a) {
                    const int rr = (wave * PER + q) * 8 + (lane >> 3);
                    const int c8 = ((lane & 7) ^ ((rr >> 1) & 7)) * 8;
                    const int iy = (st0[q] & 0xffff) - 1 + ky, ix = (st0[q] >> 16) - 1 + kx;
                    const bool ok = iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
                    g = ok ? p.A + ((int64_t)(st1[q] + iy * p.cW + ix) * p.cCin + ch0 + c8)
                           : reinterpret_cast<const bf16_t*>(g_zero_chunk) + c8;
                } else {
                    g = p.W + st0[q] + k0;
                }
            }
            dma16_to_lds(g, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE_BYTES + wave * (PER * 1024) + q * 1024)));
        }
    };

    // ---- fragment addressing (identity row maps: conflict-free under the (row>>1)&7 chunk swizzle) ----
    // Row r of fragment i is wr*WTM + 16 i + frow (A) / BM + wc*64 + 16 j + frow (W): the swizzle key (r>>1)&7 only depends
    // on frow (all other terms are multiples of 16), and the row offsets are one base + compile-time multiples of 2 KiB —
    // two base registers and two swizzled chunk offsets (one per 32-wide K half) address all 8..12 fragment reads.
    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int f_sw = (frow >> 1) & 7;
    const int a_base = (wr * WTM + frow) * 128;
    const int w_base = (BM_ + wc * 64 + frow) * 128;
    const int ch_off[2] = {((0 * 4 + fk) ^ f_sw) << 4, ((1 * 4 + fk) ^ f_sw) << 4};

    float4_t acc[FA][4];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // K range of this workgroup (whole K unless split_k > 1)
    const int nk_total = (int)(p.K / 64);
    const int nk_per = (nk_total + p.split_k - 1) / p.split_k;
    const int kt0 = ksplit * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const int64_t kbase = (int64_t)kt0 * 64;
    // SWAP: first MFMA operand = W rows -> C^T fragments (lane owns 4 consecutive columns of one row);
    // !SWAP (VT tiles): first operand = A rows (lane owns 4 consecutive tokens of one channel).
    auto compute_stage = [&](const char* st, auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[FA], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(st + w_base + ch_off[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < FA; ++i) {
                uint4 raw = *reinterpret_cast<const uint4*>(st + a_base + ch_off[ks] + i * 2048);
                if constexpr (A_MODE != UC_A_DENSE) {
                    if (p.relu_a) raw = glds_relu_bf16x8(raw);   // uniform flag: ReLU of the DPT residual conv unit, applied on load
                }
                af[i] = __builtin_bit_cast(bf16x8_t, raw);
            }
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    auto main_loop = [&](auto swap_tag) {
        if constexpr (STAGES == 2) {
            // 2-stage ring: the DMA of step kt+1 is in flight while the MFMAs of step kt run.
            if (nk > 0) issue_stage(0, kbase);
            for (int kt = 0; kt < nk; ++kt) {
                wait_vmcnt<0>();                 // this wave's pieces of stage kt have landed
                __builtin_amdgcn_s_barrier();    // ... and everyone else's; every wave is done reading stage kt-1
                asm volatile("" ::: "memory");
                if (kt + 1 < nk && !(p.dbg & 1)) issue_stage((kt + 1) & 1, kbase + (int64_t)(kt + 1) * 64);
                compute_stage(smem + (kt & 1) * STAGE_BYTES, swap_tag);
            }
        } else {
            // 3-stage ring, DMA two K-steps ahead: the loads of step kt+1 stay in flight across the barrier of step kt.
            if (nk > 0) issue_stage(0, kbase);
            if (nk > 1) issue_stage(1, kbase + 64);
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) wait_vmcnt<PER>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                int nxt = cur + 2; if (nxt >= 3) nxt -= 3;
                if (kt + 2 < nk) issue_stage(nxt, kbase + (int64_t)(kt + 2) * 64);
                compute_stage(smem + cur * STAGE_BYTES, swap_tag);
                cur = (cur == 2) ? 0 : cur + 1;
            }
        }
    };
    if constexpr (A_MODE == UC_A_DENSE) {
        if (mode == 2) main_loop(std::false_type{}); else main_loop(std::true_type{});
    } else {
        main_loop(std::true_type{});
    }

    glds_epilogue<FA>(p, acc, mode, wave_m, wave_n, lane, ksplit);
}

template <int BM_, int BN_, int WM_, int WN_, int STAGES, int A_MODE>
static void launch_variant_mode(GldsParams p, hipStream_t st) {
    p.tiles_m = (int)ceil_div64(p.M, BM_);
    p.tiles_n = (int)ceil_div64(p.N, BN_);
    auto kfn = gemm_bf16_glds_kernel<BM_, BN_, WM_, WN_, STAGES, A_MODE>;
    constexpr int smem = STAGES * (BM_ + BN_) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)p.tiles_m * p.tiles_n * (unsigned)p.split_k), dim3(WM_ * WN_ * 64), smem, st, p);
}

template <int BM_, int BN_, int WM_, int WN_, int STAGES>
static void launch_variant(const GldsParams& p, hipStream_t st) {
    if (p.a_mode == UC_A_CONV3X3) launch_variant_mode<BM_, BN_, WM_, WN_, STAGES, UC_A_CONV3X3>(p, st);
    else launch_variant_mode<BM_, BN_, WM_, WN_, STAGES, UC_A_DENSE>(p, st);
}

// variant: 0 = 128x128 (2x2 waves of 64x64), 1 = 256x128 (4x2 of 64x64), 2 = 256x256 (4x4 of 64x64)
int uc_launch_gemm_glds(const GldsParams& p, int variant, hipStream_t st) {
    switch (variant) {
        case 1: {
            static int deep1 = -1;
            if (deep1 < 0) { const char* e = getenv("UC_GEMM_SMALL_STAGES"); deep1 = e ? atoi(e) : 3; }
            const int64_t wgs = ceil_div64(p.M, 256) * ceil_div64(p.N, 128) * (p.split_k > 1 ? p.split_k : 1);
            if (deep1 == 3 && wgs <= 256) launch_variant<256, 128, 4, 2, 3>(p, st);   // latency regime, see below
            else launch_variant<256, 128, 4, 2, 2>(p, st);
            break;
        }
        case 2: launch_variant<256, 256, 4, 4, 2>(p, st); break;
        default: {
            // latency regime (fewer workgroups than CUs: every K-step waits for its own DMA): a 3-stage ring keeps two
            // stages in flight per workgroup
            static int deep = -1;
            if (deep < 0) { const char* e = getenv("UC_GEMM_SMALL_STAGES"); deep = e ? atoi(e) : 3; }
            const int64_t wgs = ceil_div64(p.M, 128) * ceil_div64(p.N, 128) * (p.split_k > 1 ? p.split_k : 1);
            if (deep == 3 && wgs <= 512) launch_variant<128, 128, 2, 2, 3>(p, st);
            else launch_variant<128, 128, 2, 2, 2>(p, st);
            break;
        }
    }
    return 0;
}
