#pragma once
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
#include "knobs.h"
#include <stdlib.h>

// How the epilogues see the kernel parameters: through the kernarg segment (scalar loads at the point of use), see the
// kernel's epilogue section.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((opencl_constant)) GldsParams& glds_pe_t;
#else
typedef const GldsParams& glds_pe_t;
#endif
#include <algorithm>
#include <type_traits>

// fp16 operand form (F16 = true): the same kernels with v_mfma_f32_16x16x32_f16 and fp16 stores — the prediction heads'
// "TF32-class" mode (10-bit mantissa like TF32, which is what the reference's fp32 heads run on under allow_tf32,
// libs/croco/blocks.py:15 / factory/dust3r.py:288-309), at the bf16 rate.  Only the EPI_ALL family is instantiated for it (3x3
// convolutions, 1x1 convolutions / ConvTranspose GEMMs: bias, ReLU / GELU, residuals, fused tail); RoPE / VT / LayerNorm options
// are rejected by the launcher.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 glds_half2_t __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ float4_t glds_mfma(bf16x8_t a, bf16x8_t b, float4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned glds_pack2(float lo, float hi) {
    if constexpr (F16) {
        const float2v_t v = {uc_sat_f16(lo), uc_sat_f16(hi)};      // (saturating: see common.h)
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, glds_half2_t));
    } else return pack_bf16x2(lo, hi);
}
__device__ __forceinline__ float4_t glds_unpack_f16x4(uint2 q) {
    const float2v_t a = __builtin_convertvector(__builtin_bit_cast(glds_half2_t, q.x), float2v_t);
    const float2v_t b = __builtin_convertvector(__builtin_bit_cast(glds_half2_t, q.y), float2v_t);
    return (float4_t){a.x, a.y, b.x, b.y};
}

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

// ReLU of eight bf16 values: a bf16 is negative exactly when its bit pattern is negative as an int16, so max(int16, 0) is the
// ReLU (-0 -> +0): one v_pk_max_i16 per register instead of shift / and / multiply / and-not — the fragment-load ReLU of the
// residual conv units sits in the K-loop, where vector instructions take their cycles from the matrix pipe.
typedef short glds_short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 glds_relu_bf16x8(uint4 v) {
    unsigned* q = reinterpret_cast<unsigned*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        q[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(glds_short2_t, q[i]), (glds_short2_t){0, 0}));
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

// saddr form: wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset
__device__ __forceinline__ void dma16_s_to_lds(unsigned voff, const void* sbase, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
        : "memory");
}

// chunk swizzle key of LDS row r: 128-B rows (r>>1)&7, 64-B rows 3*((r>>2)&1) — both make the ds_read_b128 fragment
// loads (lane = row, lane>>4 = k chunk) conflict-free within the hardware's 16-lane service groups
template <int BK_>
__device__ __forceinline__ int glds_swz(int r) { return BK_ == 64 ? ((r >> 1) & 7) : (((r >> 2) & 1) * 3); }

// Buffer-addressed form: source = descriptor base + voff + soff (bytes); a lane whose offset is outside the descriptor's
// range gets zeros written to its 16 LDS bytes.  s_nop 4 covers SGPR (descriptor / soffset / M0) write -> VMEM read.
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf_to_lds(unsigned voff, uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(srd), "s"(lds_byte_addr), "s"(soff)
        : "memory");
}

// ... under a wave-uniform execution mask (all lanes or none): a slot of a fixed per-wave DMA schedule that only some waves fill is
// issued with EXEC = 0 by the others — no branch in the instruction stream (a branch splits the MFMA stream into basic blocks),
// no work in the memory pipeline.
__device__ __forceinline__ void dma16_buf_to_lds_if(unsigned exec_half, unsigned voff, uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b32 %1, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_mov_b32 exec_lo, %6\n\t"
        "s_mov_b32 exec_hi, %6\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %2, %3, %5 offen lds\n\t"
        "s_mov_b64 exec, %0\n\t"
        "s_mov_b32 m0, %1"
        : "=&s"(keep_exec), "=&s"(keep)
        : "v"(voff), "s"(srd), "s"(lds_byte_addr), "s"(soff), "s"(exec_half)
        : "memory");
}

template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// Side panel of the eight-wave kernel (GldsParams::side_lds): 8 KiB of LDS behind the two-stage ring, filled by one DMA piece per wave
// before the first K-step — (mean, rstd) of the tile's 256 rows, column sums and bias of its 256 columns, RoPE positions of its rows.
constexpr int GLDS_SIDE_STATS = 0, GLDS_SIDE_COLSUM = 2048, GLDS_SIDE_BIAS = 3072, GLDS_SIDE_POS = 4096, GLDS_SIDE_BYTES = 8192;

// (mean, rstd) of row m for the folded-LayerNorm epilogues: finalized by uc_ln_stats_finalize, or merged here from the producer's
// block partials (small batches)
template <bool BATCH = false>
__device__ __forceinline__ float2 glds_ln_row_stats(glds_pe_t p, int64_t m) {
    if (p.ln_partial) return uc_ln_merge_row<BATCH>(p.ln_partial + m, p.M, p.ln_nblk, p.ln_eps);      // partials are [nblk][M]
    return p.ln_stats[m];
}

// Epilogue of V tiles that are written in the packed VT layout (un-swapped orientation).
template <int FA, bool LN = false, bool BATCH = false>
__device__ __forceinline__ void glds_epilogue_vt(glds_pe_t p, float4_t (&acc)[FA][4], int64_t wave_m, int64_t wave_n, int lane,
                                                 char* wbuf, const char* side = nullptr) {
    const int frow = lane & 15;
    const int g = lane >> 4;
    // Folded LayerNorm (LN): value = rstd[row] * (acc - mean[row] * colsum[channel]) + bias[channel], rows 16 i + 4 g + r,
    // channel 16 j + frow.  Lane l holds the statistics of row l (one coalesced load); the four rows of a fragment come
    // through ds_bpermute.  The correction is applied where a value is consumed: rewriting the 64 accumulators in place
    // first made the compiler park all of them in scratch.
    float csj[4] = {0.f, 0.f, 0.f, 0.f};
    float2 mine = make_float2(0.f, 1.f);
    static_assert(!LN || FA == 4, "folded LayerNorm: 64-row wave tiles");
    if constexpr (LN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) csj[j] = side ? reinterpret_cast<const float*>(side + GLDS_SIDE_COLSUM)[(int)(wave_n & 255) + 16 * j + frow]
                                                  : p.ln_colsum[wave_n + 16 * j + frow];
        mine = side ? reinterpret_cast<const float2*>(side + GLDS_SIDE_STATS)[(int)(wave_m & 255) + lane]
                    : glds_ln_row_stats<BATCH>(p, min(wave_m + lane, p.M - 1));
    }
    auto bias_of = [&](int j) __attribute__((always_inline)) -> float {
        if (side) return reinterpret_cast<const float*>(side + GLDS_SIDE_BIAS)[(int)(wave_n & 255) + 16 * j + frow];     // (side panel: bias present, launcher-checked)
        return p.bias ? p.bias[wave_n + 16 * j + frow] : 0.f;
    };
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
    auto row_stats = [&](int i) __attribute__((always_inline)) {
        if constexpr (LN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { mu[r] = __shfl(mine.x, 16 * i + 4 * g + r, 64); rs[r] = __shfl(mine.y, 16 * i + 4 * g + r, 64); }
        }
    };
    auto val = [&](int i, int j, int r, float bc) __attribute__((always_inline)) -> float {
        if constexpr (LN) return __builtin_fmaf(__builtin_fmaf(-mu[r], csj[j], acc[i][j][r]), rs[r], bc);
        else return acc[i][j][r] + bc;
    };
    if (FA == 4 && (p.vt_ntok & 63) == 0 && wave_m + 64 <= p.M && ((uintptr_t)p.vt_out & 15) == 0 && !UC_DBG(p, 16)) {
        // The wave's 64 tokens are one aligned 64-position group of one image: 64 channel rows x 128 contiguous bytes of VT.
        // Bounce [channel d][position] through the wave's LDS block (chunk c of row d at chunk c ^ (d & 7), 8-byte halves
        // swapped when d & 8) and store whole rows: 8 x dwordx4 instead of 16 x dwordx2 that touch 16 rows x 32 B each.
        const int head = (int)((wave_n - p.vt_col0) >> 6);
        const int nheads = (int)((p.N - p.vt_col0) >> 6);
        float bcj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bcj[j] = bias_of(j);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            row_stats(i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = 16 * j + frow;
                const uint2 pk = (uint2){pack_bf16x2(val(i, j, 0, bcj[j]), val(i, j, 1, bcj[j])), pack_bf16x2(val(i, j, 2, bcj[j]), val(i, j, 3, bcj[j]))};
                *reinterpret_cast<uint2*>(wbuf + d * 128 + (((2 * i + (g & 1)) ^ (d & 7)) << 4) + (((g >> 1) ^ ((d >> 3) & 1)) << 3)) = pk;
            }
        }
        const int b = (int)(wave_m / p.vt_ntok);
        const int tok0 = (int)(wave_m - (int64_t)b * p.vt_ntok);
        bf16_t* base = p.vt_out + ((int64_t)b * nheads + head) * 64 * (int64_t)p.vt_npad + tok0;
        const int crow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int d = 8 * ps + crow;
            uint4 v = *reinterpret_cast<const uint4*>(wbuf + d * 128 + (pch << 4));
            if (ps & 1) v = (uint4){v.z, v.w, v.x, v.y};
            *reinterpret_cast<uint4*>(base + (int64_t)d * p.vt_npad + ((pch ^ (d & 7)) << 3)) = v;
        }
        return;
    }
    {
        // acc[i][j][r]: token row m = wave_m + 16i + 4g + r, channel column wave_n + 16j + frow
        const int head = (int)((wave_n - p.vt_col0) >> 6);
        const int nheads = (int)((p.N - p.vt_col0) >> 6);
        // token counts that are multiples of 4 (round 6: 196 tokens of a 224 x 224 view; before: multiples of 16 only): the four rows
        // mb .. mb + 3 of a lane (mb is a multiple of 4) are one image's aligned token quad, i.e. four CONSECUTIVE positions of the
        // permuted 16-key group — one 8-byte store per channel, four lanes filling 32 contiguous bytes of a channel row, instead of
        // sixteen 2-byte stores (the 224 x 224 forward's QKV / KV GEMMs ran at 0.21-0.31 of peak on that path)
        const bool aligned = (p.vt_ntok & 3) == 0;
        float bcol[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bcol[j] = bias_of(j);
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int64_t mb = wave_m + 16 * i + 4 * g;
            row_stats(i);
            if (aligned) {
                if (mb >= p.M) continue;
                const int b = (int)(mb / p.vt_ntok);
                const int tok = (int)(mb % p.vt_ntok);
                const int qi = (tok & 15) >> 2;                      // the quad's index inside its 16-key group (uc_vt_perm)
                const int pos = (tok & ~15) + ((qi & 1) << 3) + ((qi >> 1) << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = 16 * j + frow;
                    bf16_t* dst = p.vt_out + (((int64_t)b * nheads + head) * 64 + d) * p.vt_npad + pos;
                    uint2 pk;
                    pk.x = pack_bf16x2(val(i, j, 0, bcol[j]), val(i, j, 1, bcol[j]));
                    pk.y = pack_bf16x2(val(i, j, 2, bcol[j]), val(i, j, 3, bcol[j]));
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
                            f32_to_bf16(val(i, j, r, bcol[j]));
                }
            }
        }
    }
}

// Epilogue of the swapped orientation (bias -> activation -> fused RoPE-2D -> residual(s) -> dact -> store).
//
// In the accumulator layout one store instruction touches 16 rows x 32-64 B, and the vector memory path retires about
// one row segment per 4 cycles per CU whatever its width: a 256x256 tile's store tail took 7 us (bf16 or fp32 output,
// tools/probes/store_tail.hip) against 1-2 us when every instruction covers whole 128-B lines.  So each wave bounces its
// 16x64 fragment rows through a private, XOR-swizzled 4-KiB LDS block (the stage buffers are free after the K-loop)
// and does all global traffic — pre-activation copy, residual reads, dact reads, the store — with a lane owning 4
// consecutive columns and 16 lanes covering one row (4 rows x 256 B of fp32 or 4 x 128 B of bf16 per instruction).
// Bias/activation move to that layout too; RoPE (partner channels live in one lane of the accumulator layout) is
// applied before the bounce.
//
// Run-time option tests must not sit inside per-element code: a `p.act` test per value compiled into a scalar compare
// and branch per ELEMENT (the epilogue then cost as much as six K-steps, half of it instruction fetch: the kernel was
// 40 k lines of ISA).  The two shapes that carry the forward are specialised at compile time —
//   glds_epilogue_bf16: bf16 output, no residual (qkv, fc1, kv projections, every convolution), ACT a template parameter,
//                       bf16 bounce;
//   glds_epilogue_fast<.., KIND 1>: fp32 output added to one or two fp32 residual streams (proj, fc2), fp32 bounce
//                       (its KIND 0 form is the fp32-bounce variant of the bf16 store, kept for A/B runs);
// everything else (split-K slabs, pre-activation copies, dact, bf16 residuals, partial column blocks, unaligned
// operands) takes a rolled generic drain whose option tests are per 4-column group.
// erf-GELU of the bf16-store epilogues without transcendentals: GELU(x) = x Phi(x), Phi(x) - 1/2 = x_c Q(x_c^2) with x_c = x clamped
// to +-4.5 and Q a degree-10 polynomial (least-squares fit in the Chebyshev basis over [0, 4.5^2], evaluated by Horner in
// t = 2 x_c^2 / 4.5^2 - 1).  |error| <= 1.2e-5 absolute inside the clamp, <= 3e-5 beyond it (Phi(-4.5) = 3.4e-6 instead of -> 0)
// — two orders below the bf16 output's own rounding.  15 packed-fp32-able operations per element instead of 16 scalar ones plus
// an exp2 and a reciprocal on the quarter-rate transcendental unit: the GELU epilogue of fc1 cost 7.9 us per 256x256 tile
// (as much as five K-steps), the VALU being the only unit at work in an epilogue.
// Explicit fused multiply-adds for the epilogue arithmetic: left to `-ffp-contract`, hipcc contracts each unrolled copy of an
// expression its own way, and a row's bf16 roundings would depend on where in a tile (or in a batch) the row sits.
__device__ __forceinline__ float4_t glds_fma4(float4_t a, float4_t b, float4_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float4_t glds_splat4(float v) { return (float4_t){v, v, v, v}; }

__device__ __forceinline__ float4_t glds_gelu4(float4_t x) {
    float4_t xc;
#pragma unroll
    for (int r = 0; r < 4; ++r) xc[r] = __builtin_amdgcn_fmed3f(x[r], -4.5f, 4.5f);
    const float4_t t = glds_fma4(xc * xc, glds_splat4(2.0f / 20.25f), glds_splat4(-1.0f));
    float4_t q = glds_fma4(t, glds_splat4(8.660091177e-04f), glds_splat4(-2.253821881e-03f));
    q = glds_fma4(q, t, glds_splat4(2.972107097e-03f));
    q = glds_fma4(q, t, glds_splat4(-5.424589433e-03f));
    q = glds_fma4(q, t, glds_splat4(1.124217992e-02f));
    q = glds_fma4(q, t, glds_splat4(-1.890690569e-02f));
    q = glds_fma4(q, t, glds_splat4(2.834482012e-02f));
    q = glds_fma4(q, t, glds_splat4(-4.013649033e-02f));
    q = glds_fma4(q, t, glds_splat4(5.469475207e-02f));
    q = glds_fma4(q, t, glds_splat4(-7.719214694e-02f));
    q = glds_fma4(q, t, glds_splat4(1.569050361e-01f));
    return x * glds_fma4(xc, q, glds_splat4(0.5f));
}

template <int ACT>
__device__ __forceinline__ float4_t glds_act4(float4_t v) {
    if constexpr (ACT == UC_ACT_GELU_ERF) return glds_gelu4(v);
    else if constexpr (ACT == UC_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); return v; }
    else return v;
}

template <int ACT>
__device__ __forceinline__ float glds_act_c(float v) {
    if constexpr (ACT == UC_ACT_GELU_ERF) return glds_gelu(v);
    else if constexpr (ACT == UC_ACT_RELU) return fmaxf(v, 0.f);
    else return v;
}

// LDS bounce block of one wave: 16 rows x 256 B, 16-B chunk c of row r stored at chunk c ^ r (conflict-free for the
// accumulator-layout ds_write_b128 and the row-contiguous ds_read_b128 alike)
__device__ __forceinline__ void glds_bounce_write(char* buf, int frow, int g, int j, float4_t v) {
    *reinterpret_cast<float4_t*>(buf + frow * 256 + (((4 * j + g) ^ frow) << 4)) = v;
}
__device__ __forceinline__ float4_t glds_bounce_read(const char* buf, int R, int cchunk) {
    return *reinterpret_cast<const float4_t*>(buf + R * 256 + ((cchunk ^ R) << 4));
}

// accumulator rows of fragment row-block i (+ bias and RoPE for mode 1) -> bounce block
template <int FA>
__device__ __forceinline__ void glds_stage_rows(glds_pe_t p, float4_t (&acc)[FA][4], int i, int mode, int64_t wave_m,
                                                int64_t wave_n, int frow, int g, char* buf) {
    if (mode == 1) {
        const int64_t m = min(wave_m + 16 * i + frow, p.M - 1);
        int py = (int)p.rope_pos[m * 2 + 0];
        int px = (int)p.rope_pos[m * 2 + 1];
        py = min(max(py, 0), p.rope_npos - 1);
        px = min(max(px, 0), p.rope_npos - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // fragment pair (2h, 2h+1) = channels d, d+16 of the y (h = 0) / x (h = 1) half
            const float4_t* tb = reinterpret_cast<const float4_t*>(p.rope_table + (h ? px : py) * 16 + 4 * g);
            const float4_t c0 = tb[0], c1 = tb[1];                               // (cos,sin) x 4
            const float cs[4] = {c0.x, c0.z, c1.x, c1.z}, sn[4] = {c0.y, c0.w, c1.y, c1.w};
            float4_t bu = (float4_t){0.f, 0.f, 0.f, 0.f}, bw = bu;
            if (p.bias) {                  // rope tiles are whole 64-column heads inside N (launcher-checked), bias 16-B aligned or scalar
                const float* bp = p.bias + wave_n + 32 * h + 4 * g;
                bu = (float4_t){bp[0], bp[1], bp[2], bp[3]};
                bw = (float4_t){bp[16], bp[17], bp[18], bp[19]};
            }
            float4_t ou, ow;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = acc[i][2 * h][r] + bu[r], w = acc[i][2 * h + 1][r] + bw[r];   // no activation with RoPE (launcher-checked)
                ou[r] = u * cs[r] - w * sn[r];
                ow[r] = w * cs[r] + u * sn[r];
            }
            glds_bounce_write(buf, frow, g, 2 * h, ou);
            glds_bounce_write(buf, frow, g, 2 * h + 1, ow);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds_bounce_write(buf, frow, g, j, acc[i][j]);
    }
}

template <int FA, int ACT, int KIND, bool NT = false>
__device__ __forceinline__ void glds_epilogue_fast(glds_pe_t p, float4_t (&acc)[FA][4], int mode, int64_t wave_m,
                                                   int64_t wave_n, int lane, char* wbuf) {
    const int frow = lane & 15, g = lane >> 4;
    // drain layout.  KIND 0 (bf16 out): a lane owns 8 consecutive columns, 8 lanes cover a row, an instruction stores 8
    // rows x 128 B as dwordx4.  KIND 1 (fp32): 4 columns per lane, 16 lanes per row, 4 rows x 256 B per instruction.
    constexpr int LPR = KIND == 0 ? 8 : 16;              // lanes per row
    constexpr int RPP = 64 / LPR;                        // rows per pass
    constexpr int NPS = 16 / RPP;                        // passes per 16-row block
    const int crow = lane / LPR, cc = lane % LPR;
    const int64_t nb = wave_n + (KIND == 0 ? 8 : 4) * cc;
    float4_t bias4 = (float4_t){0.f, 0.f, 0.f, 0.f}, bias4b = bias4;
    if (mode != 1 && p.bias) {
        bias4 = *reinterpret_cast<const float4_t*>(p.bias + nb);
        if constexpr (KIND == 0) bias4b = *reinterpret_cast<const float4_t*>(p.bias + nb + 4);
    }
    const int rows_left = (int)min((int64_t)(16 * FA), p.M - wave_m) - crow;      // row 16i + RPP*ps + crow exists iff 16i + RPP*ps < rows_left
    constexpr int ESZ = KIND == 0 ? 2 : 4;
    char* cp = (char*)p.C + ((wave_m + crow) * p.ldc + nb) * ESZ;
    const int64_t cstep = RPP * p.ldc * ESZ;
    const char* rp = nullptr; const char* rp2 = nullptr; int64_t rstep = 0;
    if constexpr (KIND == 1) {
        rp = (const char*)p.residual + ((wave_m + crow) * p.ldr + nb) * 4;
        rp2 = p.residual2 ? (const char*)p.residual2 + ((wave_m + crow) * p.ldr + nb) * 4 : nullptr;
        rstep = RPP * p.ldr * 4;
    }
    // the next row block is staged before the current one is drained: its LDS writes overlap this block's global traffic
    glds_stage_rows<FA>(p, acc, 0, mode, wave_m, wave_n, frow, g, wbuf);
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const char* buf = wbuf + (i & 1) * 4096;
        if (i + 1 < FA) glds_stage_rows<FA>(p, acc, i + 1, mode, wave_m, wave_n, frow, g, wbuf + ((i + 1) & 1) * 4096);
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int R = RPP * ps + crow;
            if constexpr (KIND == 0) {
                float4_t v = glds_bounce_read(buf, R, 2 * cc), w = glds_bounce_read(buf, R, 2 * cc + 1);
                if (16 * i + RPP * ps < rows_left) {
                    if (mode != 1) {
                        v += bias4; w += bias4b;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = glds_act_c<ACT>(v[r]); w[r] = glds_act_c<ACT>(w[r]); }
                    }
                    *reinterpret_cast<uint4*>(cp) = (uint4){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w), pack_bf16x2(w.x, w.y), pack_bf16x2(w.z, w.w)};
                }
            } else {
                float4_t v = glds_bounce_read(buf, R, cc);
                if (16 * i + RPP * ps < rows_left) {
                    if (mode != 1) v += bias4;
                    // NT (outputs of more than 128 MB, half the Infinity Cache): the residual stream is read once and written once per
                    // sub-layer — streaming it keeps the A / W panels of the K-loop in the L2s (+1.2 % on the forward); smaller
                    // outputs (the decoder's) stay cacheable, their consumer (LayerNorm) finds them on chip
                    if constexpr (NT) {
                        v += __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(rp));
                        if (rp2) v += __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(rp2));
                        __builtin_nontemporal_store(v, reinterpret_cast<float4_t*>(cp));
                    } else {
                        v += *reinterpret_cast<const float4_t*>(rp);
                        if (rp2) v += *reinterpret_cast<const float4_t*>(rp2);
                        *reinterpret_cast<float4_t*>(cp) = v;
                    }
                }
            }
            cp += cstep;
            if constexpr (KIND == 1) { rp += rstep; if (rp2) rp2 += rstep; }
        }
    }
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane
__device__ __forceinline__ float glds_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}

// fp32 output, optionally added to an fp32 residual stream (proj, fc2, patch / input embeddings; optionally a second addend).
// The output may alias the residual (in-place accumulate), so the compiler keeps every residual load behind the previous
// store: written as one load -> add -> store chain per 4-row pass that is 16 HBM round trips per wave, 21 us per 256x256
// tile — as long as the K-loop of a K = 1024 GEMM.  Here the residual of row block i + 1 is in flight (4 x 16 B per lane)
// while block i drains; each element is still read before it is written, which is all the in-place form needs.
//
// Producer half of the folded LayerNorm: with p.twin the stored rows are also written as bf16 (the A operand of the next
// GEMM), with p.stats_out every row's (sum, squared deviations from its own mean) over the wave's 64 columns is written
// — the row's 16 lanes reduce with DPP row rotations; uc_ln_stats_finalize merges the blocks into (mean, rstd).
//
// BS (bf16 residual stream, the reference's own policy under autocast): C and the residual(s) are bf16 — 2 + 2 bytes per element
// instead of 4 + 4 + 2; the stored rows ARE the next GEMM's A operand (no twin) and the statistics are those of the ROUNDED values
// the consumer will normalise.
template <int FA, bool NT, bool BS = false>
__device__ __forceinline__ void glds_epilogue_resid(glds_pe_t p, float4_t (&acc)[FA][4], int mode, int64_t wave_m,
                                                    int64_t wave_n, int lane, char* wbuf) {
    constexpr int ES = BS ? 2 : 4;                       // bytes per element of C and of the residual(s)
    const int frow = lane & 15, g = lane >> 4;
    const int crow = lane >> 4, cc = lane & 15;          // drain: 4 columns per lane, 16 lanes per row, 4 rows x 256 B per instruction
    const int64_t nb = wave_n + 4 * cc;
    float4_t bias4 = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (mode != 1 && p.bias) bias4 = *reinterpret_cast<const float4_t*>(p.bias + nb);
    const int rows_left = (int)min((int64_t)(16 * FA), p.M - wave_m) - crow;      // row 16i + 4ps + crow exists iff 16i + 4ps < rows_left
    // wave-uniform 64-bit bases + one 32-bit lane offset per matrix: per-lane 64-bit pointers for C, the residuals, the twin
    // and the statistics cost 10 registers this epilogue does not have (a spill reload between its stores waits for every
    // store issued before it)
    char* cbase = (char*)p.C + (wave_m * p.ldc + wave_n) * ES;
    const unsigned coff = (unsigned)(crow * (int)p.ldc + 4 * cc) * (unsigned)ES;
    const int64_t cstep = 4 * p.ldc * ES;
    // diagnostics (wrong results): dbg & 128 never reads the residual(s), dbg & 256 never writes the twin — together the HBM bytes of
    // a bf16 residual stream (4 per element instead of 10)
    const char* rbase = (p.residual && !UC_DBG(p, 128)) ? (const char*)p.residual + (wave_m * p.ldr + wave_n) * ES : nullptr;
    const char* rbase2 = (p.residual2 && !UC_DBG(p, 128)) ? (const char*)p.residual2 + (wave_m * p.ldr + wave_n) * ES : nullptr;
    const unsigned roff = (unsigned)(crow * (int)p.ldr + 4 * cc) * (unsigned)ES;
    const int64_t rstep = 4 * p.ldr * ES;
    char* tbase = (!BS && p.twin && !UC_DBG(p, 256)) ? (char*)p.twin + (wave_m * p.ldt + wave_n) * 2 : nullptr;
    const unsigned toff = (unsigned)(crow * (int)p.ldt + 4 * cc) * 2u;
    const int64_t tstep = 4 * p.ldt * 2;
    const int nblk = (int)(p.N >> 6);
    float2* sbase = p.stats_out ? p.stats_out + (wave_n >> 6) * p.M + wave_m : nullptr;      // [N/64][M]: block-major (ABI 11)
    (void)nblk;
    // ONE register set for the residual: pass ps of row block i + 1 is requested as soon as pass ps of block i has used
    // its value (each element is still read before the in-place store of its own row), so 4 loads per lane stay in flight.
    // (BS: the four bf16 values stay packed — 2 registers — until they are used; one residual only, see the launcher)
    typedef unsigned glds_u2_t __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<BS, glds_u2_t, float4_t>::type res_t;
    res_t res[4];
    auto load_res = [&](int i, int ps) __attribute__((always_inline)) {
        if constexpr (BS) {
            res[ps] = (glds_u2_t){0u, 0u};
            if (rbase && 16 * i + 4 * ps < rows_left) {
                const glds_u2_t* q = reinterpret_cast<const glds_u2_t*>(rbase + (4 * i + ps) * rstep + roff);
                if constexpr (NT) res[ps] = __builtin_nontemporal_load(q); else res[ps] = *q;
            }
        } else {
            res[ps] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (rbase && 16 * i + 4 * ps < rows_left) {
                const float4_t* q = reinterpret_cast<const float4_t*>(rbase + (4 * i + ps) * rstep + roff);
                if constexpr (NT) res[ps] = __builtin_nontemporal_load(q); else res[ps] = *q;
                if (rbase2) {
                    const float4_t* q2 = reinterpret_cast<const float4_t*>(rbase2 + (4 * i + ps) * rstep + roff);
                    if constexpr (NT) res[ps] += __builtin_nontemporal_load(q2); else res[ps] += *q2;
                }
            }
        }
    };
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) load_res(0, ps);
    glds_stage_rows<FA>(p, acc, 0, mode, wave_m, wave_n, frow, g, wbuf);
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const char* buf = wbuf + (i & 1) * 4096;
        if (i + 1 < FA) glds_stage_rows<FA>(p, acc, i + 1, mode, wave_m, wave_n, frow, g, wbuf + ((i + 1) & 1) * 4096);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            float4_t v = glds_bounce_read(buf, 4 * ps + crow, cc);
            if (mode != 1) v += bias4;
            if constexpr (BS) {
                v += (float4_t){__uint_as_float(res[ps].x << 16), __uint_as_float(res[ps].x & 0xffff0000u), __uint_as_float(res[ps].y << 16),
                                __uint_as_float(res[ps].y & 0xffff0000u)};
            } else v += res[ps];
            if (i + 1 < FA) load_res(i + 1, ps);
            const bool row_ok = 16 * i + 4 * ps < rows_left;
            if constexpr (BS) {
                const glds_u2_t pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
                if (row_ok) {
                    glds_u2_t* cq = reinterpret_cast<glds_u2_t*>(cbase + (4 * i + ps) * cstep + coff);
                    if constexpr (NT) __builtin_nontemporal_store(pk, cq); else *cq = pk;
                }
                // the statistics below are those of the stored (rounded) row
                v = (float4_t){__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                               __uint_as_float(pk.y & 0xffff0000u)};
            } else if (row_ok) {
                // NT (outputs of more than 128 MB, half the Infinity Cache): the residual stream is read once and written once per
                // sub-layer — streaming it keeps the A / W panels of the K-loop in the L2s (+1.2 % on the forward); smaller
                // outputs (the decoder's) stay cacheable, their consumer finds them on chip
                float4_t* cq = reinterpret_cast<float4_t*>(cbase + (4 * i + ps) * cstep + coff);
                if constexpr (NT) __builtin_nontemporal_store(v, cq); else *cq = v;
                if (tbase) *reinterpret_cast<uint2*>(tbase + (4 * i + ps) * tstep + toff) = (uint2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
            }
            if (sbase && !UC_DBG(p, 512)) {   // all 64 lanes take part in the row reductions (rows past M carry finite garbage and are not stored)
                // (no FMA contraction here: hipcc contracts each unrolled copy of this block differently, and a row's statistics —
                // hence the bf16 roundings of everything downstream — must not depend on where in a tile the row sits)
#pragma clang fp contract(off)
                const float s = glds_row16_sum((v.x + v.y) + (v.z + v.w));
                const float mu = s * (1.f / 64.f);
                const float4_t dv = v - mu;
                const float q = glds_row16_sum((dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w));
                if (row_ok && cc == 0 && !UC_DBG(p, 1024)) sbase[16 * i + 4 * ps + crow] = make_float2(s, q);
            }
        }
    }
}

#ifndef GLDS_BS_RES_AHEAD
#define GLDS_BS_RES_AHEAD 2     // row blocks of residual in flight ahead of the one being drained (bf16-stream epilogue)
#endif
#ifndef GLDS_BS_RES_AHEAD_WIDE
#define GLDS_BS_RES_AHEAD_WIDE 2   // ... in the kernels whose waves own 128-row tiles (eight- and four-wave forms: 256 registers per lane)
#endif
// sum over the 8 lanes 8k .. 8k+7 (one row of the bf16-stream drain), result in every lane: xor 1, xor 2 inside the quad, then the
// mirrored lane of the 8-lane half row (which lies in the other quad)
__device__ __forceinline__ float glds_row8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    return v;
}

// bf16 residual stream (round 4 form): bf16 output = one rounding of acc + bias + bf16 residual, row statistics of the STORED rows.
// Round 3 ran this through the fp32 residual drain (4 columns per lane: 8-byte loads / stores, one scattered 8-byte statistics store
// per row and wave): measured on the encoder's proj GEMM (316 us; 236 with a plain bf16 store) 20 us were the statistics stores
// (1024 eight-byte stores to 1024 different lines per tile), 6 us the reductions, 36 us the residual read (at the HBM burst rate: all
// CUs reach their epilogues together) and 18 us the drain's own instructions.  Here: the bf16-store drain layout — 8 columns per lane,
// 8 lanes per row, 8 rows x 128 B per instruction: 16-byte residual loads and stores, half the instructions; the row sums over 8
// lanes (3 DPP steps instead of 4) of 8 in-lane values; and the statistics leave as ONE coalesced store per 64 rows: after each pass
// every lane of a row holds that row's pair, lane (crow, cc) keeps the pair of pass cc, so at the end lane (crow, cc) owns row
// 8 cc + crow and the 64 lanes write 512 contiguous bytes of the block-major statistics array [N/64][M] (ABI 11).
template <int FA, bool NT>
__device__ __forceinline__ void glds_epilogue_bs(glds_pe_t p, float4_t (&acc)[FA][4], int /*mode*/, int64_t wave_m, int64_t wave_n, int lane,
                                                 char* wbuf) {
    constexpr int mode = 0;      // (the launcher routes no RoPE / VT tile to this family: keeps the RoPE branch of glds_stage_rows out of the kernel)
    const int frow = lane & 15, g = lane >> 4;
    const int crow = lane >> 3, cc = lane & 7;
    const int64_t nb = wave_n + 8 * cc;
    float4_t bias4 = (float4_t){0.f, 0.f, 0.f, 0.f}, bias4b = bias4;
    if (mode != 1 && p.bias) {
        bias4 = *reinterpret_cast<const float4_t*>(p.bias + nb);
        bias4b = *reinterpret_cast<const float4_t*>(p.bias + nb + 4);
    }
    const int rows_total = (int)min((int64_t)(16 * FA), p.M - wave_m);
    const int rows_left = rows_total - crow;              // row 16i + 8ps + crow exists iff 16i + 8ps < rows_left
    // wave-uniform 64-bit bases + one 32-bit lane offset per matrix (see glds_epilogue_resid)
    char* cbase = (char*)p.C + (wave_m * p.ldc + wave_n) * 2;
    const unsigned coff = (unsigned)(crow * (int)p.ldc + 8 * cc) * 2u;
    const int64_t cstep = 8 * p.ldc * 2;
    const char* rbase = (p.residual && !UC_DBG(p, 128)) ? (const char*)p.residual + (wave_m * p.ldr + wave_n) * 2 : nullptr;
    const unsigned roff = (unsigned)(crow * (int)p.ldr + 8 * cc) * 2u;
    const int64_t rstep = 8 * p.ldr * 2;
    float2* sbase = (p.stats_out && !UC_DBG(p, 512)) ? p.stats_out + (wave_n >> 6) * p.M + wave_m : nullptr;
    // the residual of the next AHEAD row blocks (2 x 16 bytes per lane each) is in flight while block i drains
    constexpr int AHEAD = FA > 4 ? GLDS_BS_RES_AHEAD_WIDE : GLDS_BS_RES_AHEAD;
    uint4_t res[AHEAD][2];
    auto load_res = [&](int i, int ps) __attribute__((always_inline)) {
        res[i % AHEAD][ps] = (uint4_t){0u, 0u, 0u, 0u};
        if (rbase && 16 * i + 8 * ps < rows_left) {
            const uint4_t* q = reinterpret_cast<const uint4_t*>(rbase + (2 * i + ps) * rstep + roff);
            if constexpr (NT) res[i % AHEAD][ps] = __builtin_nontemporal_load(q); else res[i % AHEAD][ps] = *q;
        }
    };
    auto lo = [](unsigned u) __attribute__((always_inline)) { return __uint_as_float(u << 16); };
    auto hi = [](unsigned u) __attribute__((always_inline)) { return __uint_as_float(u & 0xffff0000u); };
    float keep_s[FA / 4], keep_q[FA / 4];
#pragma unroll
    for (int h = 0; h < FA / 4; ++h) keep_s[h] = keep_q[h] = 0.f;
#pragma unroll
    for (int i = 0; i < AHEAD && i < FA; ++i) { load_res(i, 0); load_res(i, 1); }
    glds_stage_rows<FA>(p, acc, 0, mode, wave_m, wave_n, frow, g, wbuf);
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const char* buf = wbuf + (i & 1) * 4096;
        if (i + 1 < FA) glds_stage_rows<FA>(p, acc, i + 1, mode, wave_m, wave_n, frow, g, wbuf + ((i + 1) & 1) * 4096);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int R = 8 * ps + crow;
            float4_t v = glds_bounce_read(buf, R, 2 * cc), w = glds_bounce_read(buf, R, 2 * cc + 1);
            if (mode != 1) { v += bias4; w += bias4b; }
            const uint4_t rr = res[i % AHEAD][ps];
            v += (float4_t){lo(rr.x), hi(rr.x), lo(rr.y), hi(rr.y)};
            w += (float4_t){lo(rr.z), hi(rr.z), lo(rr.w), hi(rr.w)};
            if (i + AHEAD < FA) load_res(i + AHEAD, ps);
            const bool row_ok = 16 * i + 8 * ps < rows_left;
            const uint4_t pk = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w), pack_bf16x2(w.x, w.y), pack_bf16x2(w.z, w.w)};
            if (row_ok) {
                uint4_t* cq = reinterpret_cast<uint4_t*>(cbase + (2 * i + ps) * cstep + coff);
                if constexpr (NT) __builtin_nontemporal_store(pk, cq); else *cq = pk;
            }
            if (sbase) {   // statistics of the stored (rounded) row; all 64 lanes take part (rows past M carry finite garbage, not stored)
                // (no FMA contraction: a row's statistics must not depend on where in a tile the row sits — see glds_epilogue_resid)
#pragma clang fp contract(off)
                const float x0 = lo(pk.x), x1 = hi(pk.x), x2 = lo(pk.y), x3 = hi(pk.y), x4 = lo(pk.z), x5 = hi(pk.z), x6 = lo(pk.w), x7 = hi(pk.w);
                const float s = glds_row8_sum(((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)));
                const float mu = s * (1.f / 64.f);
                const float d0 = x0 - mu, d1 = x1 - mu, d2 = x2 - mu, d3 = x3 - mu, d4 = x4 - mu, d5 = x5 - mu, d6 = x6 - mu, d7 = x7 - mu;
                const float q = glds_row8_sum(((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7)));
                const bool mine = cc == ((2 * i + ps) & 7);
                keep_s[(2 * i + ps) >> 3] = mine ? s : keep_s[(2 * i + ps) >> 3];
                keep_q[(2 * i + ps) >> 3] = mine ? q : keep_q[(2 * i + ps) >> 3];
            }
        }
    }
    if (sbase && !UC_DBG(p, 1024)) {
#pragma unroll
        for (int h = 0; h < FA / 4; ++h) {
            const int row = 64 * h + 8 * cc + crow;      // pass 8 h + cc covered rows 64 h + 8 cc + (0..7)
            if (row < rows_total) sbase[row] = make_float2(keep_s[h], keep_q[h]);
        }
    }
}

// 16-bit output + bias + one or two 16-bit residuals laid out like C (the residual conv units of the DPT head: conv2(...) + x (+ the
// other path), libs/croco/dpt_block.py:56-63) — the bf16-stream drain above without the statistics, for bf16 or fp16 storage: the
// accumulator rows bounce through LDS in fp32, a lane adds bias and the residuals of its 8 columns in fp32 (16-byte residual loads, the
// next two row blocks' in flight), ONE rounding, 16-byte stores.  Before round 4's last session these launches took the generic drain
// (scalar-ish 8-byte traffic): the same 128^2 256->256 convolution cost 3.48 ms with its two residuals and 2.17 ms without.
template <int FA, bool NT, bool F16>
__device__ __forceinline__ void glds_epilogue_res16(glds_pe_t p, float4_t (&acc)[FA][4], int64_t wave_m, int64_t wave_n, int lane, char* wbuf) {
    const int frow = lane & 15, g = lane >> 4;
    const int crow = lane >> 3, cc = lane & 7;
    const int64_t nb = wave_n + 8 * cc;
    float4_t bias4 = (float4_t){0.f, 0.f, 0.f, 0.f}, bias4b = bias4;
    if (p.bias) {
        bias4 = *reinterpret_cast<const float4_t*>(p.bias + nb);
        bias4b = *reinterpret_cast<const float4_t*>(p.bias + nb + 4);
    }
    const int rows_total = (int)min((int64_t)(16 * FA), p.M - wave_m);
    const int rows_left = rows_total - crow;              // row 16i + 8ps + crow exists iff 16i + 8ps < rows_left
    char* cbase = (char*)p.C + (wave_m * p.ldc + wave_n) * 2;
    const unsigned coff = (unsigned)(crow * (int)p.ldc + 8 * cc) * 2u;
    const int64_t cstep = 8 * p.ldc * 2;
    const char* rbase = (const char*)p.residual + (wave_m * p.ldr + wave_n) * 2;
    const char* r2base = p.residual2 ? (const char*)p.residual2 + (wave_m * p.ldr + wave_n) * 2 : nullptr;
    const unsigned roff = (unsigned)(crow * (int)p.ldr + 8 * cc) * 2u;
    const int64_t rstep = 8 * p.ldr * 2;
    constexpr int AHEAD = 2;
    uint4_t res[AHEAD][2], res2[AHEAD][2];
    auto load_res = [&](int i, int ps) __attribute__((always_inline)) {
        res[i % AHEAD][ps] = (uint4_t){0u, 0u, 0u, 0u};
        res2[i % AHEAD][ps] = (uint4_t){0u, 0u, 0u, 0u};
        if (16 * i + 8 * ps < rows_left) {
            const uint4_t* q = reinterpret_cast<const uint4_t*>(rbase + (2 * i + ps) * rstep + roff);
            if constexpr (NT) res[i % AHEAD][ps] = __builtin_nontemporal_load(q); else res[i % AHEAD][ps] = *q;
            if (r2base) {
                const uint4_t* q2 = reinterpret_cast<const uint4_t*>(r2base + (2 * i + ps) * rstep + roff);
                if constexpr (NT) res2[i % AHEAD][ps] = __builtin_nontemporal_load(q2); else res2[i % AHEAD][ps] = *q2;
            }
        }
    };
    auto add16 = [](float4_t& v, float4_t& w, uint4_t rr) __attribute__((always_inline)) {
        if constexpr (F16) {
            v += glds_unpack_f16x4((uint2){rr.x, rr.y});
            w += glds_unpack_f16x4((uint2){rr.z, rr.w});
        } else {
            v += (float4_t){__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xffff0000u), __uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xffff0000u)};
            w += (float4_t){__uint_as_float(rr.z << 16), __uint_as_float(rr.z & 0xffff0000u), __uint_as_float(rr.w << 16), __uint_as_float(rr.w & 0xffff0000u)};
        }
    };
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < AHEAD && i < FA; ++i) { load_res(i, 0); load_res(i, 1); }
    glds_stage_rows<FA>(p, acc, 0, 0, wave_m, wave_n, frow, g, wbuf);
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const char* buf = wbuf + (i & 1) * 4096;
        if (i + 1 < FA) glds_stage_rows<FA>(p, acc, i + 1, 0, wave_m, wave_n, frow, g, wbuf + ((i + 1) & 1) * 4096);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int R = 8 * ps + crow;
            float4_t v = glds_bounce_read(buf, R, 2 * cc), w = glds_bounce_read(buf, R, 2 * cc + 1);
            v += bias4; w += bias4b;
            add16(v, w, res[i % AHEAD][ps]);
            add16(v, w, res2[i % AHEAD][ps]);             // (zeros without a second residual)
            if (i + AHEAD < FA) load_res(i + AHEAD, ps);
            if constexpr (F16) amax = uc_amax(uc_amax(amax, uc_amax(uc_amax(fabsf(v.x), fabsf(v.y)), uc_amax(fabsf(v.z), fabsf(v.w)))),
                                              uc_amax(uc_amax(fabsf(w.x), fabsf(w.y)), uc_amax(fabsf(w.z), fabsf(w.w))));      // (NaN-propagating: common.h)
            const uint4_t pk = {glds_pack2<F16>(v.x, v.y), glds_pack2<F16>(v.z, v.w), glds_pack2<F16>(w.x, w.y), glds_pack2<F16>(w.z, w.w)};
            if (16 * i + 8 * ps < rows_left) {
                uint4_t* cq = reinterpret_cast<uint4_t*>(cbase + (2 * i + ps) * cstep + coff);
                if constexpr (NT) __builtin_nontemporal_store(pk, cq); else *cq = pk;
            }
        }
    }
    if constexpr (F16) {
        if (p.sat_flag && __any(!(amax <= UC_F16_MAX)) && lane == 0) atomicOr(p.sat_flag, 1);
    }
}

// LN: the folded-LayerNorm form — the accumulator holds x . W'^T of the RAW rows; row statistics and the column sums of W'
// turn it into LN(x) . W^T:  rstd[m] * (acc - mean[m] * colsum[n]) + bias[n].
template <int FA, int ACT, bool NT = false, bool LN = false, bool F16 = false>
__device__ __forceinline__ void glds_epilogue_bf16(glds_pe_t p, float4_t (&acc)[FA][4], int mode, int64_t wave_m,
                                                   int64_t wave_n, int lane, char* wbuf, const char* side = nullptr) {
    const int frow = lane & 15, g = lane >> 4;
    // Folded LayerNorm (LN): value = rstd[row] * (acc - mean[row] * colsum[col]) + bias[col], rows 16 i + frow, columns
    // 16 j + 4 g + r.  Lane l holds the statistics of row l of the wave tile (one coalesced load); a row block's pair comes
    // through ds_bpermute when the block is staged.  Applied where a value is consumed, never as an in-place pass over acc.
    // The LN form keeps its two column vectors (column sums, bias) of the wave's 64 columns in LDS behind the bounce
    // blocks and reads 16 bytes of each per fragment: held in registers (32 of them) next to the accumulators they spilled,
    // and a spill reload between global stores waits for every store before it (vmcnt counts stores, in order).
    float2 mine[FA / 4];
    float* colbuf = reinterpret_cast<float*>(wbuf + 4096);
    const float* biasbuf = colbuf + 64;
    if constexpr (LN) {
        if (side) {      // staged before the K-loop (side panel): no global load at the head of the epilogue
            colbuf = const_cast<float*>(reinterpret_cast<const float*>(side + GLDS_SIDE_COLSUM)) + (int)(wave_n & 255);
            biasbuf = reinterpret_cast<const float*>(side + GLDS_SIDE_BIAS) + (int)(wave_n & 255);
#pragma unroll
            for (int q = 0; q < FA / 4; ++q) mine[q] = reinterpret_cast<const float2*>(side + GLDS_SIDE_STATS)[(int)(wave_m & 255) + 64 * q + lane];
        } else {
            colbuf[lane] = p.ln_colsum[wave_n + lane];
            colbuf[64 + lane] = p.bias ? p.bias[wave_n + lane] : 0.f;
#pragma unroll
            for (int q = 0; q < FA / 4; ++q) mine[q] = glds_ln_row_stats<(FA > 4)>(p, min(wave_m + 64 * q + lane, p.M - 1));
        }
    }
    float st_mu = 0.f, st_rs = 1.f;
    auto row_stats = [&](int i) __attribute__((always_inline)) {
        if constexpr (LN) { st_mu = __shfl(mine[i / 4].x, 16 * (i & 3) + frow, 64); st_rs = __shfl(mine[i / 4].y, 16 * (i & 3) + frow, 64); }
    };
    float4_t b4[4];
    if constexpr (!LN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b4[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4[j] = *reinterpret_cast<const float4_t*>(p.bias + wave_n + 16 * j + 4 * g);
        }
    }
    auto val4 = [&](int i, int j) __attribute__((always_inline)) -> float4_t {   // pre-activation values of fragment (i, j)
        if constexpr (LN) {
            const float4_t cs = *reinterpret_cast<const float4_t*>(colbuf + 16 * j + 4 * g);
            const float4_t bb = *reinterpret_cast<const float4_t*>(biasbuf + 16 * j + 4 * g);
            return glds_fma4(glds_fma4(glds_splat4(-st_mu), cs, acc[i][j]), glds_splat4(st_rs), bb);
        } else return acc[i][j] + b4[j];
    };
    // RoPE tiles: no loads inside the per-row-block code.  The table form (positions -> table address -> cos/sin, per row
    // block) was four dependent load chains per wave and cost 7 us per tile; here the positions of all row blocks are read up
    // front and the rotation angles go through the hardware sin/cos (argument in turns, v_fract first): 16 transcendentals
    // per row block per lane.  Channel 4g + r of a quarter has frequency F0 * base^(-(4g + r) / 16) (kernels.cu:36-81).
    unsigned pyx[FA];            // (y | x << 16) of row 16 i + frow: positions below 65536 (launcher-checked table size)
    float turn0 = 0.f;           // turns per unit position of channel 4g; channel 4g + r: turn0 * rope_ratio^r
    if (mode == 1) {
        turn0 = p.rope_turn0 * __builtin_amdgcn_exp2f((float)(4 * g) * p.rope_l2ratio);
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int64_t m = min(wave_m + 16 * i + frow, p.M - 1);
            const longlong2 yx = (LN && side) ? *reinterpret_cast<const longlong2*>(side + GLDS_SIDE_POS + ((int)(wave_m & 255) + 16 * i + frow) * 16)
                                              : *reinterpret_cast<const longlong2*>(p.rope_pos + m * 2);
            pyx[i] = ((unsigned)min(max((int)yx.x, 0), 65535)) | ((unsigned)min(max((int)yx.y, 0), 65535) << 16);
        }
    }
    const int wr_off = frow * 128 + (((g & 1) ^ (frow >> 3)) << 3);       // + ((2j + (g>>1)) ^ (frow & 7)) << 4
    float amax = 0.f;            // fp16 stores: largest magnitude this lane packs (values beyond +-65504 saturate and raise p.sat_flag)
    auto stage = [&](int i, char* buf) {
        row_stats(i);
        if (mode == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {      // fragment pair (2h, 2h+1) = channels d, d+16 of the y (h = 0) / x (h = 1) half
                const float pa = (float)(h ? (pyx[i] >> 16) : (pyx[i] & 0xffffu));
                const float4_t uu = val4(i, 2 * h), ww = val4(i, 2 * h + 1);
                float ou[4], ow[4];
                float turn = turn0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tfrac = __builtin_amdgcn_fractf(pa * turn);
                    turn *= p.rope_ratio;
                    const float cs = __builtin_amdgcn_cosf(tfrac), sn = __builtin_amdgcn_sinf(tfrac);
                    const float u = uu[r], w = ww[r];   // no activation with RoPE (launcher-checked)
                    ou[r] = __builtin_fmaf(u, cs, -(w * sn));
                    ow[r] = __builtin_fmaf(w, cs, u * sn);
                }
                *reinterpret_cast<uint2*>(buf + wr_off + (((4 * h + (g >> 1)) ^ (frow & 7)) << 4)) = (uint2){glds_pack2<F16>(ou[0], ou[1]), glds_pack2<F16>(ou[2], ou[3])};
                *reinterpret_cast<uint2*>(buf + wr_off + (((4 * h + 2 + (g >> 1)) ^ (frow & 7)) << 4)) = (uint2){glds_pack2<F16>(ow[0], ow[1]), glds_pack2<F16>(ow[2], ow[3])};
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4_t v = glds_act4<ACT>(val4(i, j));
                if constexpr (F16) amax = uc_amax(uc_amax(amax, uc_amax(fabsf(v.x), fabsf(v.y))), uc_amax(fabsf(v.z), fabsf(v.w)));
                *reinterpret_cast<uint2*>(buf + wr_off + (((2 * j + (g >> 1)) ^ (frow & 7)) << 4)) = (uint2){glds_pack2<F16>(v.x, v.y), glds_pack2<F16>(v.z, v.w)};
            }
        }
    };
    const int crow = lane >> 3, pch = lane & 7;              // drain: row 8*ps + crow, physical chunk pch
    const int rows_left = (int)min((int64_t)(16 * FA), p.M - wave_m) - crow;
    char* cp = (char*)p.C + ((wave_m + crow) * p.ldc + wave_n) * 2;
    const int64_t cstep = 8 * p.ldc * 2;
    // Plain form: row block i + 1 is staged before block i is drained (its LDS writes overlap the global stores).  Folded-
    // LayerNorm form: 32 more registers (column sums, statistics) are live while staging, and a spill reload between global
    // stores waits for ALL of them (vmcnt counts stores, in order): the epilogue took 16 us instead of 6.  It stages all four
    // row blocks first — the wave's 8-KiB LDS block holds them as bf16 — and drains with no accumulator left alive.
    constexpr bool ALL_FIRST = false;   // (measured: staging all four row blocks first costs the LDS-write / store overlap: +2.9 us per tile)
    if constexpr (ALL_FIRST) {
#pragma unroll
        for (int i = 0; i < FA; ++i) stage(i, wbuf + i * 2048);
    } else {
        stage(0, wbuf);
    }
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const char* buf = ALL_FIRST ? wbuf + i * 2048 : wbuf + (i & 1) * 2048;
        if constexpr (!ALL_FIRST) {
            if (i + 1 < FA) stage(i + 1, wbuf + ((i + 1) & 1) * 2048);
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int R = 8 * ps + crow;
            uint4 v = *reinterpret_cast<const uint4*>(buf + R * 128 + (pch << 4));
            if (ps) v = (uint4){v.z, v.w, v.x, v.y};         // rows 8..15 store their halves swapped
            if (16 * i + 8 * ps < rows_left) {
                if constexpr (NT) {
                    const uint4_t vv = (uint4_t){v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<uint4_t*>(cp + ((pch ^ (R & 7)) << 4)));
                } else {
                    *reinterpret_cast<uint4*>(cp + ((pch ^ (R & 7)) << 4)) = v;
                }
            }
            cp += cstep;
        }
    }
    if constexpr (F16) {
        if (p.sat_flag && __any(!(amax <= UC_F16_MAX)) && lane == 0) atomicOr(p.sat_flag, 1);
    }
}

// Training forms of the bf16 store (mode 0 tiles, no residual): PRE also stores the value BEFORE the activation (fc1's
// pre-activation, saved for the backward), DACT multiplies the result by act'(u) of a saved pre-activation u laid out like C
// (the activation backward fused into the data-gradient GEMM).  Same bounce / whole-row stores as glds_epilogue_bf16; in the
// generic drain these two cost 60 % on top of the K = 1024 GEMMs that carry them (12 % of a training step).
template <int FA, int ACT, bool PRE, bool DACT>
__device__ __forceinline__ void glds_epilogue_bf16_train(glds_pe_t p, float4_t (&acc)[FA][4], int64_t wave_m, int64_t wave_n, int lane,
                                                         char* wbuf) {
    const int frow = lane & 15, g = lane >> 4;
    float4_t b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b4[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4[j] = *reinterpret_cast<const float4_t*>(p.bias + wave_n + 16 * j + 4 * g);
    }
    const int wr_off = frow * 128 + (((g & 1) ^ (frow >> 3)) << 3);
    auto stage = [&](int i, char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4_t pre = acc[i][j] + b4[j];
            const int off = wr_off + (((2 * j + (g >> 1)) ^ (frow & 7)) << 4);
            if constexpr (PRE) *reinterpret_cast<uint2*>(buf + 4096 + off) = (uint2){pack_bf16x2(pre.x, pre.y), pack_bf16x2(pre.z, pre.w)};
            const float4_t v = glds_act4<ACT>(pre);
            *reinterpret_cast<uint2*>(buf + off) = (uint2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
        }
    };
    const int crow = lane >> 3, pch = lane & 7;
    const int rows_left = (int)min((int64_t)(16 * FA), p.M - wave_m) - crow;
    const int64_t row0 = ((wave_m + crow) * p.ldc + wave_n) * 2;           // byte offset of (row, column block) in C / preact / dact_u
    const int64_t cstep = 8 * p.ldc * 2;
    stage(0, wbuf);
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const char* buf = wbuf + (i & 1) * 2048;
        if (i + 1 < FA) stage(i + 1, wbuf + ((i + 1) & 1) * 2048);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int R = 8 * ps + crow;
            const int64_t off = row0 + (2 * i + ps) * cstep + ((pch ^ (R & 7)) << 4);
            uint4 v = *reinterpret_cast<const uint4*>(buf + R * 128 + (pch << 4));
            if (ps) v = (uint4){v.z, v.w, v.x, v.y};         // rows 8..15 keep their halves swapped in the bounce block
            if (16 * i + 8 * ps < rows_left) {
                if constexpr (DACT) {
                    const uint4 u = *reinterpret_cast<const uint4*>((const char*)p.dact_u + off);
                    const unsigned vv[4] = {v.x, v.y, v.z, v.w}, uu[4] = {u.x, u.y, u.z, u.w};
                    unsigned oo[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v0 = __uint_as_float(vv[q] << 16), v1 = __uint_as_float(vv[q] & 0xffff0000u);
                        const float u0 = __uint_as_float(uu[q] << 16), u1 = __uint_as_float(uu[q] & 0xffff0000u);
                        oo[q] = pack_bf16x2(v0 * glds_dact(u0, p.dact_act), v1 * glds_dact(u1, p.dact_act));
                    }
                    v = (uint4){oo[0], oo[1], oo[2], oo[3]};
                }
                *reinterpret_cast<uint4*>((char*)p.C + off) = v;
                if constexpr (PRE) {
                    uint4 w = *reinterpret_cast<const uint4*>(buf + 4096 + R * 128 + (pch << 4));
                    if (ps) w = (uint4){w.z, w.w, w.x, w.y};
                    *reinterpret_cast<uint4*>((char*)p.preact + off) = w;
                }
            }
        }
    }
}

// 4 values of row m, columns nb..nb+3 of a [*, ld] matrix of dtype dt: vector access when `full`, else per element inside N
__device__ __forceinline__ float4_t glds_load4(const void* base, int dt, int64_t idx, bool full, int64_t nb, int64_t N) {
    float4_t v = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (full) {
        if (dt == UC_F32) v = *reinterpret_cast<const float4_t*>((const float*)base + idx);
        else if (dt == UC_F16) v = glds_unpack_f16x4(*reinterpret_cast<const uint2*>((const bf16_t*)base + idx));
        else {
            const uint2 q = *reinterpret_cast<const uint2*>((const bf16_t*)base + idx);
            v = (float4_t){__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
        }
    } else {
        for (int r = 0; r < 4; ++r)
            if (nb + r < N) v[r] = dt == UC_F32 ? ((const float*)base)[idx + r] : (dt == UC_F16 ? f16_to_f32(((const bf16_t*)base)[idx + r]) : bf16_to_f32(((const bf16_t*)base)[idx + r]));
    }
    return v;
}
__device__ __forceinline__ void glds_store4(void* base, int dt, int64_t idx, bool full, int64_t nb, int64_t N, float4_t v) {
    if (full) {
        if (dt == UC_F32) *reinterpret_cast<float4_t*>((float*)base + idx) = v;
        else if (dt == UC_F16) *reinterpret_cast<uint2*>((bf16_t*)base + idx) = (uint2){glds_pack2<true>(v.x, v.y), glds_pack2<true>(v.z, v.w)};
        else *reinterpret_cast<uint2*>((bf16_t*)base + idx) = (uint2){pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    } else {
        for (int r = 0; r < 4; ++r) {
            if (nb + r >= N) continue;
            if (dt == UC_F32) ((float*)base)[idx + r] = v[r];
            else if (dt == UC_F16) ((bf16_t*)base)[idx + r] = f32_to_f16(v[r]);
            else ((bf16_t*)base)[idx + r] = f32_to_bf16(v[r]);
        }
    }
}

template <int FA>
__device__ __forceinline__ void glds_epilogue_generic(glds_pe_t p, float4_t (&acc)[FA][4], int mode, int64_t wave_m,
                                                      int64_t wave_n, int lane, int ksplit, char* wbuf) {
    const int frow = lane & 15, g = lane >> 4;
    const int crow = lane >> 4, cchunk = lane & 15;
    const int64_t nb = wave_n + 4 * cchunk;
    const bool full = p.vec_ok && nb + 3 < p.N;
    float4_t bias4 = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (mode != 1 && p.bias && p.split_k <= 1) bias4 = glds_load4(p.bias, UC_F32, nb, full, nb, p.N);
    float* slab = (float*)p.C + (int64_t)ksplit * p.M * p.ldc;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        char* buf = wbuf + (i & 1) * 4096;
        glds_stage_rows<FA>(p, acc, i, mode, wave_m, wave_n, frow, g, buf);
#pragma unroll 1
        for (int ps = 0; ps < 4; ++ps) {
            const int R = 4 * ps + crow;
            float4_t v = glds_bounce_read(buf, R, cchunk);
            const int64_t m = wave_m + 16 * i + R;
            if (m >= p.M || nb >= p.N) continue;
            const int64_t ci = m * p.ldc + nb;
            if (p.split_k > 1) { glds_store4(slab, UC_F32, ci, full, nb, p.N, v); continue; }
            if (mode != 1) {
                v += bias4;
                if (p.preact) glds_store4(p.preact, p.out_dtype, ci, full, nb, p.N, v);
                if (p.act == UC_ACT_GELU_ERF) { for (int r = 0; r < 4; ++r) v[r] = glds_gelu(v[r]); }
                else if (p.act == UC_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); }
            }
            if (p.residual) {
                v += glds_load4(p.residual, p.res_dtype, m * p.ldr + nb, full, nb, p.N);
                if (p.residual2) v += glds_load4(p.residual2, p.res_dtype, m * p.ldr + nb, full, nb, p.N);
            }
            if (p.dact_u) {   // fused activation backward: out = v * act'(u)
                const float4_t u = glds_load4(p.dact_u, UC_BF16, ci, full, nb, p.N);
                if (p.dact_act == UC_ACT_RELU) { for (int r = 0; r < 4; ++r) v[r] = u[r] > 0.f ? v[r] : 0.f; }
                else { for (int r = 0; r < 4; ++r) v[r] *= glds_dact(u[r], UC_ACT_GELU_ERF); }
            }
            amax = uc_amax(uc_amax(amax, uc_amax(fabsf(v.x), fabsf(v.y))), uc_amax(fabsf(v.z), fabsf(v.w)));
            glds_store4(p.C, p.out_dtype, ci, full, nb, p.N, v);
        }
    }
    // fp16 outputs saturate at +-65504 (glds_store4): tell the caller when they did
    if (p.out_dtype == UC_F16 && p.sat_flag && __any(!(amax <= UC_F16_MAX)) && lane == 0) atomicOr(p.sat_flag, 1);
}

enum { GLDS_EPI_ALL = 0, GLDS_EPI_BF16 = 1, GLDS_EPI_F32 = 2, GLDS_EPI_BS = 3, GLDS_EPI_RES16 = 4 };

// Fused narrow tail of a 128-wide tile (two wave columns of 64): out4[m][o] = tail_b[o] + sum_n act(acc[m][n] + bias[n]) * tail_w[o][n].
// The DPT regressor's conv3x3 -> ReLU -> Conv2d(128 -> 4, 1x1): the 128-channel map is never stored.  Per wave: its 64 columns of
// tail_w (as float4 over the four outputs) and of the bias sit in its LDS block; a lane multiplies its 16 values of a row by
// them, the four lanes that share a row (lane bits 4, 5) add up through two xor-shuffles, the two wave columns meet through
// LDS, and the left wave column stores 64 rows x 16 bytes — one contiguous KiB.
template <int FA, int ACT>
__device__ __forceinline__ void glds_epilogue_tail4(glds_pe_t p, float4_t (&acc)[FA][4], int64_t wave_m, int64_t wave_n, int lane,
                                                    int wave, char* smem) {
    static_assert(FA == 4, "the fused tail is written for 64-row wave tiles");
    const int frow = lane & 15, g = lane >> 4;
    char* wbuf = smem + wave * 8192;
    float4_t* wl = reinterpret_cast<float4_t*>(wbuf);              // [64 columns] x (o = 0..3)
    float* bl = reinterpret_cast<float*>(wbuf + 1024);            // [64] bias
    float4_t* xl = reinterpret_cast<float4_t*>(wbuf + 2048);       // [64 rows] partial sums of this wave column
    {
        const int64_t n = wave_n + lane;
        wl[lane] = (float4_t){p.tail_w[n], p.tail_w[p.N + n], p.tail_w[2 * p.N + n], p.tail_w[3 * p.N + n]};
        bl[lane] = p.bias ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        float4_t s = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4_t b4 = *reinterpret_cast<const float4_t*>(bl + 16 * j + 4 * g);
            const float4_t v = glds_act4<ACT>(acc[i][j] + b4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s += v[r] * wl[16 * j + 4 * g + r];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            s[o] += __shfl_xor(s[o], 16, 64);
            s[o] += __shfl_xor(s[o], 32, 64);
        }
        if (g == 0) xl[16 * i + frow] = s;
    }
    __syncthreads();                                             // both wave columns of every row block have written their partials
    if ((wave & 1) == 0) {                                       // wave = 2 * wave_row + wave_column in every 128-wide tile variant
        const float4_t* other = reinterpret_cast<const float4_t*>(smem + (wave + 1) * 8192 + 2048);
        float4_t o = xl[lane] + other[lane];
        if (p.tail_b) o += (float4_t){p.tail_b[0], p.tail_b[1], p.tail_b[2], p.tail_b[3]};
        if (wave_m + lane < p.M) *reinterpret_cast<float4_t*>(p.tail_out + (wave_m + lane) * 4) = o;
    }
}

// Epilogue dispatch shared by the 16-wave and the 8-wave kernels: picks the drain of this wave's 64-column block from the family
// compiled into the instantiation (EPI) and the wave's mode (0 plain, 1 RoPE, 2 packed-VT).
template <int FA, int A_MODE, int EPI, bool F16 = false>
__device__ __forceinline__ void glds_epilogue_dispatch(glds_pe_t pe, float4_t (&acc)[FA][4], int mode, int64_t wave_m, int64_t wave_n,
                                                       int tid, int wave, int ksplit, char* smem, const char* side = nullptr) {
    static_assert(!F16 || EPI == GLDS_EPI_ALL || EPI == GLDS_EPI_RES16, "the fp16 operand form exists in the EPI_ALL and RES16 families only");
    constexpr int OUT16 = F16 ? UC_F16 : UC_BF16;        // the 16-bit storage dtype of this instantiation
    if (wave_n >= pe.N) return;
    {
        // a fresh definition of the lane id: keeps the compiler from hoisting the epilogue's per-lane address math above
        // the K-loop, where it spilled loop-carried registers of the 128-VGPR kernels
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        char* wbuf = smem + wave * 8192;
        // (the launcher routes a descriptor to the BF16 / F32 family only when every tile of it takes that family's epilogue)
        const bool plain = pe.vec_ok && wave_n + 64 <= pe.N && pe.split_k <= 1 && !pe.preact && !pe.dact_u && !UC_DBG(pe, 16);
        auto bf16_family = [&]() __attribute__((always_inline)) {
            const bool nt = pe.nt_out & (mode == 1 ? 4 : 2);
            if (!F16 && A_MODE == UC_A_DENSE && (pe.ln_stats || pe.ln_partial)) {   // folded LayerNorm
                if constexpr (A_MODE == UC_A_DENSE && !F16) {
                    if (nt) {
                        if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_bf16<FA, UC_ACT_GELU_ERF, true, true>(pe, acc, mode, wave_m, wave_n, lane, wbuf, side);
                        else glds_epilogue_bf16<FA, UC_ACT_NONE, true, true>(pe, acc, mode, wave_m, wave_n, lane, wbuf, side);
                    } else {
                        if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_bf16<FA, UC_ACT_GELU_ERF, false, true>(pe, acc, mode, wave_m, wave_n, lane, wbuf, side);
                        else glds_epilogue_bf16<FA, UC_ACT_NONE, false, true>(pe, acc, mode, wave_m, wave_n, lane, wbuf, side);
                    }
                }
            } else if (nt) {
                if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_bf16<FA, UC_ACT_GELU_ERF, true, false, F16>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
                else if (pe.act == UC_ACT_RELU) glds_epilogue_bf16<FA, UC_ACT_RELU, true, false, F16>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
                else glds_epilogue_bf16<FA, UC_ACT_NONE, true, false, F16>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
            } else {
                if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_bf16<FA, UC_ACT_GELU_ERF, false, false, F16>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
                else if (pe.act == UC_ACT_RELU) glds_epilogue_bf16<FA, UC_ACT_RELU, false, false, F16>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
                else glds_epilogue_bf16<FA, UC_ACT_NONE, false, false, F16>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
            }
        };
        auto f32_family = [&]() __attribute__((always_inline)) {
            if (pe.nt_out & 1) glds_epilogue_resid<FA, true>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
            else glds_epilogue_resid<FA, false>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
        };
        if constexpr (EPI == GLDS_EPI_BF16) {
            if (mode == 2) {
                // (a 128-row wave tile drains as two 64-row halves: the packed-VT fast path is one aligned 64-token group)
                if constexpr (FA == 4) {
                    if (A_MODE == UC_A_DENSE && (pe.ln_stats || pe.ln_partial)) glds_epilogue_vt<4, A_MODE == UC_A_DENSE>(pe, acc, wave_m, wave_n, lane, wbuf);
                    else glds_epilogue_vt<4>(pe, acc, wave_m, wave_n, lane, wbuf);
                } else {
#pragma unroll
                    for (int h = 0; h < FA / 4; ++h) {
                        float4_t half[4][4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) half[i][j] = acc[4 * h + i][j];
                        if (A_MODE == UC_A_DENSE && (pe.ln_stats || pe.ln_partial)) glds_epilogue_vt<4, A_MODE == UC_A_DENSE, true>(pe, half, wave_m + 64 * h, wave_n, lane, wbuf, side);
                        else glds_epilogue_vt<4>(pe, half, wave_m + 64 * h, wave_n, lane, wbuf);
                    }
                }
            } else bf16_family();
        } else if constexpr (EPI == GLDS_EPI_F32) {
            f32_family();
        } else if constexpr (EPI == GLDS_EPI_RES16) {       // convolution with 16-bit residual(s) laid out like its 16-bit output (launcher: glds_res16_ok)
            if (pe.nt_out & 2) glds_epilogue_res16<FA, true, F16>(pe, acc, wave_m, wave_n, lane, wbuf);
            else glds_epilogue_res16<FA, false, F16>(pe, acc, wave_m, wave_n, lane, wbuf);
        } else if constexpr (EPI == GLDS_EPI_BS) {          // bf16 residual stream: bf16 output + bf16 residual(s) + row statistics
            if (pe.nt_out & 1) glds_epilogue_bs<FA, true>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
            else glds_epilogue_bs<FA, false>(pe, acc, mode, wave_m, wave_n, lane, wbuf);
        } else {
            if constexpr (A_MODE != UC_A_DENSE && FA == 4) {
                if (pe.tail_out) {      // (launcher: N == 128 on a 128-wide tile, every wave of the workgroup arrives here)
                    if (pe.act == UC_ACT_RELU) glds_epilogue_tail4<FA, UC_ACT_RELU>(pe, acc, wave_m, wave_n, lane, wave, smem);
                    else if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_tail4<FA, UC_ACT_GELU_ERF>(pe, acc, wave_m, wave_n, lane, wave, smem);
                    else glds_epilogue_tail4<FA, UC_ACT_NONE>(pe, acc, wave_m, wave_n, lane, wave, smem);
                    return;
                }
            } else if constexpr (A_MODE != UC_A_DENSE && FA == 8) {
                if (pe.tail_out) {      // a 128-row wave tile (eight-wave conv kernel, 4 x 2 waves) drains as two 64-row halves
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float4_t half[4][4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) half[i][j] = acc[4 * h + i][j];
                        if (h) __syncthreads();     // the left wave columns have read the first half's partial sums
                        if (pe.act == UC_ACT_RELU) glds_epilogue_tail4<4, UC_ACT_RELU>(pe, half, wave_m + 64 * h, wave_n, lane, wave, smem);
                        else if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_tail4<4, UC_ACT_GELU_ERF>(pe, half, wave_m + 64 * h, wave_n, lane, wave, smem);
                        else glds_epilogue_tail4<4, UC_ACT_NONE>(pe, half, wave_m + 64 * h, wave_n, lane, wave, smem);
                    }
                    return;
                }
            }
            if (!F16 && mode == 2) {
                if constexpr (F16) { }
                else if constexpr (FA == 4) glds_epilogue_vt<4>(pe, acc, wave_m, wave_n, lane, wbuf);
                else {
#pragma unroll
                    for (int h = 0; h < FA / 4; ++h) {
                        float4_t half[4][4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) half[i][j] = acc[4 * h + i][j];
                        glds_epilogue_vt<4>(pe, half, wave_m + 64 * h, wave_n, lane, wbuf);
                    }
                }
            } else if (plain && pe.out_dtype == OUT16 && !pe.residual) bf16_family();

            else if (!F16 && mode == 0 && pe.vec_ok && wave_n + 64 <= pe.N && pe.split_k <= 1 && pe.out_dtype == UC_BF16 && !pe.residual &&
                     !UC_DBG(pe, 16) && !pe.ln_stats && !pe.ln_partial && (pe.preact != nullptr) != (pe.dact_u != nullptr)) {
                // training: fc1 with its pre-activation copy / a data-gradient GEMM with the activation backward fused
                if (pe.preact) {
                    if (pe.act == UC_ACT_GELU_ERF) glds_epilogue_bf16_train<FA, UC_ACT_GELU_ERF, true, false>(pe, acc, wave_m, wave_n, lane, wbuf);
                    else if (pe.act == UC_ACT_RELU) glds_epilogue_bf16_train<FA, UC_ACT_RELU, true, false>(pe, acc, wave_m, wave_n, lane, wbuf);
                    else glds_epilogue_bf16_train<FA, UC_ACT_NONE, true, false>(pe, acc, wave_m, wave_n, lane, wbuf);
                } else if (pe.act == UC_ACT_NONE) glds_epilogue_bf16_train<FA, UC_ACT_NONE, false, true>(pe, acc, wave_m, wave_n, lane, wbuf);
                else glds_epilogue_generic<FA>(pe, acc, mode, wave_m, wave_n, lane, ksplit, wbuf);
            }
            else if (plain && pe.out_dtype == UC_F32 && (!pe.residual || pe.res_dtype == UC_F32) && pe.act == UC_ACT_NONE) f32_family();
            else glds_epilogue_generic<FA>(pe, acc, mode, wave_m, wave_n, lane, ksplit, wbuf);
        }
    }
}

// BK_ = 64 (128-B LDS rows) or 32 (64-B rows: half the LDS per stage, so two 8-wave workgroups share a CU and one's
// prologue/epilogue runs under the other's K-loop); WGS_PER_CU is the co-residency the register budget is sized for.
// EPI selects the epilogue family compiled into an instantiation (the launcher picks the instantiation from the descriptor):
//   GLDS_EPI_BF16: bf16 stores without residual — plain / activation / RoPE tiles, VT tiles, folded LayerNorm (qkv, fc1, q / kv
//                  projections, convolutions);   GLDS_EPI_F32: fp32 output (+ fp32 residuals, bf16 twin, row statistics: proj,
//                  fc2, embeddings);   GLDS_EPI_BS: the same sub-layers on a bf16 residual stream (bf16 output + bf16 residuals +
//                  row statistics);   GLDS_EPI_ALL: the generic drain next to the fast families (everything else).
// One family per kernel keeps the register allocation of the 128-VGPR K-loop out of reach of epilogue code it never runs:
// with all of them inlined into one function, every option added to one epilogue spilled DMA pointers inside the K-loop.

template <int BM_, int BN_, int WAVES_M, int WAVES_N, int STAGES, int A_MODE, int BK_ = 64, int WGS_PER_CU = 1, int EPI = GLDS_EPI_ALL, bool F16 = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, WGS_PER_CU * WAVES_M * WAVES_N / 4) void gemm_bf16_glds_kernel(GldsParams p) {
    static_assert(BN_ / WAVES_N == 64, "a wave owns 64 output columns (one 64-wide head)");
    constexpr int WTM = BM_ / WAVES_M;
    constexpr int FA = WTM / 16;          // A-row fragments per wave (4 or 8)
    static_assert(WTM % 16 == 0 && (FA == 4 || FA == 8), "wave tile rows");
    constexpr int NW = WAVES_M * WAVES_N;
    static_assert(BK_ == 64 || BK_ == 32, "K-step");
    constexpr int ROWB = BK_ * 2;         // bytes per LDS row
    constexpr int CPR = ROWB / 16;        // 16-byte chunks per row (8 or 4)
    constexpr int RPI = 1024 / ROWB;      // rows per 1-KiB DMA instruction (8 or 16)
    constexpr int STAGE_BYTES = (BM_ + BN_) * ROWB;
    constexpr int NI = (BM_ + BN_) / RPI; // 1-KiB DMA instructions per stage
    constexpr int PER = NI / NW;          // per wave
    static_assert(NI % NW == 0, "DMA instructions must split evenly over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (UC_TRACE(p)) tr0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int nwg = p.tiles_m * p.tiles_n;
    constexpr bool FUSE2_OK = BM_ == 128 && BN_ == 128;   // fused two-way K split (fuse_split2): this tile only (dense and conv, bf16 and fp16)
    const bool fuse2 = FUSE2_OK && p.fuse_split2;
    const int ksplit = fuse2 ? ((int)blockIdx.x >= nwg ? 1 : 0)
                             : (p.split_k > 1 ? (int)uc_div(blockIdx.x, p.dNwg) : 0);   // split-K slice
    const int t = glds_xcd_remap((int)blockIdx.x - ksplit * nwg, nwg);
    // Tile order inside an XCD's run: groups of GM row panels swept column by column, so the ~32 tiles an XCD runs
    // concurrently form a GM x (32/GM) block that shares GM A-panels and 32/GM W-panels in its L2 (a plain row-major
    // order shares 2 A-panels but streams ALL of W through every pair of row panels: 2.4x algorithmic fetch traffic).
    int tm, tn;
    {
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int grp = (int)uc_div((unsigned)t, p.dPerGroup), within = t - grp * per_group;
        const int first_m = grp * GM;
        const bool last = p.tiles_m - first_m < GM;                        // the ragged last group has tiles_m % GM row panels
        const int gsz = last ? p.tiles_m - first_m : GM;
        tn = (int)uc_div((unsigned)within, last ? p.dGmLast : p.dGm);
        tm = first_m + within - tn * gsz;
    }
    if (p.stagger > 0 && blockIdx.x < 256u * WGS_PER_CU) {
        // Experiment (UC_GEMM_STAGGER, off by default): de-phase the CUs.  Every tile of a launch costs the same, so all 256 CUs
        // reach their epilogues together and the store / residual traffic hits HBM as one burst while the matrix pipes idle.
        // The first round of workgroups (one per CU) starts in 8 phase groups chosen by ROW PANEL, so the column tiles that
        // share an A panel through their XCD's L2 stay in step; the offsets persist down each CU's chain of tiles.
        const unsigned phase = (unsigned)tm & 7u;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)phase * (unsigned)p.stagger;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int64_t m0 = (int64_t)tm * BM_;
    const int64_t n0 = (int64_t)tn * BN_;
    const int64_t wave_m = m0 + wr * WTM;
    const int64_t wave_n = n0 + wc * 64;

    // ---- per-wave epilogue mode (wave-uniform) ----
    const bool is_vt = A_MODE == UC_A_DENSE && p.vt_col0 >= 0 && wave_n >= p.vt_col0;        // conv tiles take neither epilogue
    const bool is_rope = A_MODE == UC_A_DENSE && !is_vt && p.rope_cols > 0 && wave_n < p.rope_cols;
    const int mode = is_vt ? 2 : (is_rope ? 1 : 0);

    // ---- DMA sources: instruction I = wave*PER + q covers combined-tile rows [RPI*I, RPI*(I+1)) ----
    // Dense: one 64-bit source pointer per instruction, advanced by k0.
    // Conv: buffer-addressed DMA.  Two wave-uniform descriptors (the input window of this tile, shifted back by one image
    // row + one pixel so that tap (ky,kx) is a non-negative uniform soffset, and the tile's weight rows); per instruction a
    // 32-bit byte offset of the lane's row and a 9-bit mask of the taps that fall inside the image.  A tap in the zero
    // padding sets the lane's offset to 0xffffffff: out of the descriptor's range, and the hardware writes zeros to LDS
    // (tools/probes/buffer_lds.hip) — no 64-bit per-lane address math, no padding source, half the address registers.
    const bf16_t* src[A_MODE == UC_A_DENSE ? PER : 1];
    unsigned st0[A_MODE == UC_A_DENSE ? 1 : PER], st1[A_MODE == UC_A_DENSE ? 1 : PER];
    uint4_t srd_a = (uint4_t){0u, 0u, 0u, 0u}, srd_w = srd_a;
    if constexpr (A_MODE != UC_A_DENSE) {
        const int b0 = (int)uc_div((unsigned)min(m0, p.M - 1), p.dHWo);     // < 2^30 output pixels (launcher-checked)
        const unsigned long long pa = (unsigned long long)(p.A + ((int64_t)b0 * p.cH * p.cW - (p.cW + 1)) * p.cCin);
        const unsigned long long pw = (unsigned long long)(p.W + min(n0, p.N - 1) * p.K);
        srd_a = (uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
        srd_w = (uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pw), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pw >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int rr = (wave * PER + q) * RPI + lane / CPR;
            const int c = (lane % CPR) ^ glds_swz<BK_>(rr);
            if (rr < BM_) {
                const unsigned m = (unsigned)min(m0 + rr, p.M - 1);          // < 2^30 output pixels (launcher-checked)
                const unsigned mrow = uc_div(m, p.dWo);
                const int ox = (int)(m - mrow * (unsigned)p.cWo) * p.cStride;
                const unsigned b = uc_div(mrow, p.dHo);
                const int oy = (int)(mrow - b * (unsigned)p.cHo) * p.cStride;
                st0[q] = (unsigned)((((int64_t)((int)b - b0) * p.cH + oy) * p.cW + ox) * p.cCin + c * 8) * 2u;
                unsigned colmask = 0, mask = 0;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) colmask |= ((unsigned)(ox - 1 + kx) < (unsigned)p.cW ? 1u : 0u) << kx;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) mask |= ((unsigned)(oy - 1 + ky) < (unsigned)p.cH ? colmask : 0u) << (3 * ky);
                st1[q] = mask;
            } else {
                const int64_t n = min(n0 + (rr - BM_), p.N - 1);
                st0[q] = (unsigned)((n - min(n0, p.N - 1)) * p.K + c * 8) * 2u;
                st1[q] = 0x1ffu;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int rr = (wave * PER + q) * RPI + lane / CPR;
            const int c = (lane % CPR) ^ glds_swz<BK_>(rr);   // logical chunk stored at physical chunk (lane % CPR) of row rr
            if (rr < BM_) src[q] = p.A + min(m0 + rr, p.M - 1) * p.lda + c * 8;
            else src[q] = p.W + min(n0 + (rr - BM_), p.N - 1) * p.K + c * 8;
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;   // LDS byte address of the dynamic region
    auto issue_stage = [&](int stage, int64_t k0) {
        int tap = 0;
        unsigned soff_a = 0, soff_w = 0;
        if constexpr (A_MODE != UC_A_DENSE) {
            tap = (int)uc_div((unsigned)k0, p.dCin);
            const int ch0 = (int)k0 - tap * p.cCin;
            const int ky = tap / 3, kx = tap - 3 * ky;
            soff_a = (unsigned)(((ky * p.cW + kx) * p.cCin + ch0) * 2);
            soff_w = (unsigned)(k0 * 2);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE_BYTES + wave * (PER * 1024) + q * 1024));
            if constexpr (A_MODE == UC_A_DENSE) {
                dma16_to_lds(src[q] + k0, dst);
            } else {
                const bool is_a = (wave * PER + q) * RPI < BM_;   // wave-uniform: an instruction is all-A or all-W
                const unsigned vo = ((st1[q] >> tap) & 1u) ? st0[q] : 0xffffffffu;
                if (is_a && UC_DBG(p, 64) && tap != 0) continue;   // diagnostics (wrong results): A tiles staged for tap 0 only — what a halo-tiled form could save at most
                if (is_a) dma16_buf_to_lds(vo, srd_a, soff_a, dst);
                else dma16_buf_to_lds(vo, srd_w, soff_w, dst);
            }
        }
    };

    // ---- fragment addressing (identity row maps: conflict-free under the (row>>1)&7 chunk swizzle) ----
    // Row r of fragment i is wr*WTM + 16 i + frow (A) / BM + wc*64 + 16 j + frow (W): the swizzle key (r>>1)&7 only depends
    // on frow (all other terms are multiples of 16), and the row offsets are one base + compile-time multiples of 2 KiB —
    // two base registers and two swizzled chunk offsets (one per 32-wide K half) address all 8..12 fragment reads.
    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int f_sw = glds_swz<BK_>(frow);
    const int a_base = (wr * WTM + frow) * ROWB;
    const int w_base = (BM_ + wc * 64 + frow) * ROWB;
    const int ch_off[2] = {((0 * 4 + fk) ^ f_sw) << 4, ((1 * 4 + fk) ^ f_sw) << 4};   // [1] unused when BK_ == 32

    float4_t acc[FA][4];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // K range of this workgroup (whole K unless split_k > 1)
    const int nk_total = (int)(p.K / BK_);
    const int nsplit = fuse2 ? 2 : p.split_k;
    const int nk_per = (nk_total + nsplit - 1) / nsplit;
    const int kt0 = ksplit * nk_per;
    int nk = max(0, min(nk_per, nk_total - kt0));
    if UC_DBG(p, 8) nk = min(nk, 1);                      // diagnostics: one K-step only (launch + prologue + epilogue cost)
    const int64_t kbase = (int64_t)kt0 * BK_;
    // SWAP: first MFMA operand = W rows -> C^T fragments (lane owns 4 consecutive columns of one row);
    // !SWAP (VT tiles): first operand = A rows (lane owns 4 consecutive tokens of one channel).
    auto compute_stage = [&](const char* st, auto swap_tag, auto&& mid) {
        constexpr bool SWAP = decltype(swap_tag)::value;
#pragma unroll
        for (int ks = 0; ks < BK_ / 32; ++ks) {
            bf16x8_t af[FA];
#pragma unroll
            for (int i = 0; i < FA; ++i) {
                uint4 raw = *reinterpret_cast<const uint4*>(st + a_base + ch_off[ks] + i * 16 * ROWB);
                if constexpr (A_MODE != UC_A_DENSE) {
                    if (p.relu_a) raw = glds_relu_bf16x8(raw);   // uniform flag: ReLU of the DPT residual conv unit, applied on load
                }
                af[i] = __builtin_bit_cast(bf16x8_t, raw);
            }
            if constexpr (A_MODE == UC_A_DENSE) {
                bf16x8_t wf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(st + w_base + ch_off[ks] + j * 16 * ROWB);
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (SWAP) acc[i][j] = glds_mfma<F16>(wf[j], af[i], acc[i][j]);
                        else acc[i][j] = glds_mfma<F16>(af[i], wf[j], acc[i][j]);
                    }
            } else {
                // conv tiles carry more loop state (offsets, tap masks, two descriptors): W fragments are read one at a time
                // (20 instead of 32 fragment registers) so that nothing spills inside the K-loop
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(st + w_base + ch_off[ks] + j * 16 * ROWB);
#pragma unroll
                    for (int i = 0; i < FA; ++i) acc[i][j] = glds_mfma<F16>(wf, af[i], acc[i][j]);
                }
            }
            if (ks == 0) mid();
        }
    };
    auto main_loop = [&](auto swap_tag) {
        if constexpr (STAGES == 2) {
            // 2-stage ring: the DMA of step kt+1 is in flight while the MFMAs of step kt run.
            if (nk > 0) issue_stage(0, kbase);
            for (int kt = 0; kt < nk; ++kt) {
                wait_vmcnt<0>();                 // this wave's pieces of stage kt have landed
                if (!UC_DBG(p, 2)) __builtin_amdgcn_s_barrier();    // ... and everyone else's; every wave is done reading stage kt-1
                if (UC_TRACE(p) && kt == 0) tr1 = __builtin_amdgcn_s_memrealtime();
                asm volatile("" ::: "memory");
                // where the next stage's DMA is issued (same-box A/B): dense pieces cost one 64-bit add each and go first
                // (behind the first MFMA group they lost 0-5 %, split between the wave halves of a SIMD 2-8 %); conv pieces
                // carry the tap test, s_nop 4 and a descriptor select and go behind the wave's first 16 queued MFMAs (+5 % on
                // the 256-channel convs)
                if constexpr (A_MODE == UC_A_DENSE) {
                    if (kt + 1 < nk && !UC_DBG(p, 1)) issue_stage((kt + 1) & 1, kbase + (int64_t)(kt + 1) * BK_);
                    compute_stage(smem + (kt & 1) * STAGE_BYTES, swap_tag, [] {});
                } else {
                    compute_stage(smem + (kt & 1) * STAGE_BYTES, swap_tag, [&] {
                        if (kt + 1 < nk && !UC_DBG(p, 1)) issue_stage((kt + 1) & 1, kbase + (int64_t)(kt + 1) * BK_);
                    });
                }
            }
        } else {
            // 3-stage ring, DMA two K-steps ahead: the loads of step kt+1 stay in flight across the barrier of step kt.
            // (A 4-stage ring of the 128x128 tile — three steps ahead — was measured at 1 pair per batch, where every K-step is a wait
            // for its own DMA: 8.49 vs 8.57 ms per forward, within noise; not kept.)
            if (nk > 0) issue_stage(0, kbase);
            if (nk > 1) issue_stage(1, kbase + BK_);
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) wait_vmcnt<PER>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (UC_TRACE(p) && kt == 0) tr1 = __builtin_amdgcn_s_memrealtime();
                asm volatile("" ::: "memory");
                int nxt = cur + 2; if (nxt >= 3) nxt -= 3;
                if (kt + 2 < nk) issue_stage(nxt, kbase + (int64_t)(kt + 2) * BK_);
                compute_stage(smem + cur * STAGE_BYTES, swap_tag, [] {});
                cur = (cur == 2) ? 0 : cur + 1;
            }
        }
    };
    if constexpr (A_MODE == UC_A_DENSE) {
        if (mode == 2) main_loop(std::false_type{}); else main_loop(std::true_type{});
    } else {
        main_loop(std::true_type{});
    }

    if constexpr (FUSE2_OK) {
        if (fuse2) {
            // Fused two-way K split.  The producers are the LOW block ids: the dispatcher starts them first and they never wait, so a
            // consumer that spins always has a partner that is running or has finished — also when two such launches share the CUs
            // (the decoder's two view streams).  The hand-over buffer is UNCACHED device memory (hipDeviceMallocUncached: no L2 on
            // either side, so partners on different XCDs — whose L2s are not coherent — need no buffer_wbl2 / buffer_inv, which
            // write back / invalidate a whole L2 and cost more than the split saves: measured); the flag is polled with volatile
            // (L1-bypassing) loads; ordering by s_waitcnt + barrier.
            float* ws = p.fs_ws + (size_t)t * (128 * 128) + (size_t)tid * 4;
            volatile unsigned* flag = p.fs_flags + t;
            if (ksplit == 0) {
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4_t*>(ws + (size_t)(i * 4 + j) * (NW * 64 * 4)) = acc[i][j];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's partial sums are in memory
                __syncthreads();                                        // ... everyone's
                if (tid == 0) *flag = 1u;
                return;
            }
            if (tid == 0) {
                while (*flag != 1u) __builtin_amdgcn_s_sleep(4);
            }
            __syncthreads();
            asm volatile("" ::: "memory");
            {   // (plain loads, all 16 in flight: the buffer is uncached memory, and no line of it can sit in this CU's L1 — a volatile
                //  access would be waited for one by one, 16 round trips)
                float4_t part[FA][4];
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) part[i][j] = *reinterpret_cast<const float4_t*>(ws + (size_t)(i * 4 + j) * (NW * 64 * 4));
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += part[i][j];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) *flag = 0u;            // for the next launch on this stream / the next graph replay
        }
    }

    // The epilogue's parameters are re-read from the kernarg segment HERE.  Carried through the K-loop in SGPRs (some 60 of
    // them: pointers, leading dimensions, option words) they overflowed the scalar file, the overflow went to VGPR lanes, and
    // the 128-VGPR K-loop answered with scratch reloads of its DMA source pointers inside the loop.
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((opencl_constant)) GldsParams* kp =
        (const __attribute__((opencl_constant)) GldsParams*)__builtin_amdgcn_kernarg_segment_ptr();   // explicit arguments start at offset 0
    asm volatile("" : "+s"(kp)::"memory");
    glds_pe_t pe = *kp;
#else
    glds_pe_t pe = p;
#endif
    if UC_DBG(pe, 4) {                                     // diagnostics: no epilogue (keeps the accumulators live)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s == 12345.678f) reinterpret_cast<float*>(pe.C)[0] = s;
        return;
    }
    // every wave is done with the last stage: the ring becomes the epilogue's bounce space (8 KiB per wave)
    static_assert(STAGES * STAGE_BYTES >= NW * 8192, "bounce space");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (UC_TRACE(pe)) tr2 = __builtin_amdgcn_s_memrealtime();
    glds_epilogue_dispatch<FA, A_MODE, EPI, F16>(pe, acc, mode, wave_m, wave_n, tid, wave, ksplit, smem);
    if (UC_TRACE(pe)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* t = UC_TRACE(pe) + (size_t)blockIdx.x * 6;
            t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memrealtime(); t[4] = hw; t[5] = xcc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Eight-wave form of the dense 256x256x64 tile: 2 x 4 waves of 128x64 (two waves per SIMD, 256 registers each).
//
// The 16-wave kernel above has 128 registers per wave: accumulators (64) + one K-half of fragments (32) fill them, so every
// fragment is read from LDS right before its MFMAs — after each barrier the matrix pipes wait for the first ds_reads of all
// 16 waves, and a 64x64 wave tile moves 512 B of LDS per MFMA.  Here a wave owns 128x64: 12 fragment reads feed 32 MFMAs
// (384 B per MFMA), and the registers hold the NEXT 32-deep K-chunk's fragments while the current chunk's MFMAs run:
//
//   chunk 0 of stage s:  MFMAs on (a, w)   | ds_read chunk 1 of stage s   -> (a, wn)       (a is refilled row block by row block)
//   mid-step:            s_waitcnt + s_barrier: everyone has READ stage s, everyone's DMA of stage s+1 has landed
//                        DMA of stage s+2 -> buffer of stage s
//   chunk 1 of stage s:  MFMAs on (a, wn)  | ds_read chunk 0 of stage s+1 -> (a, w)
//
// One barrier per K-step as before, but every wave arrives at it with 32 MFMAs' worth of operands already in registers.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_glds8_kernel(GldsParams p) {
    constexpr int BM_ = 256, BN_ = 256, WAVES_N = 4, FA = 8, ROWB = 128, STAGE_BYTES = (BM_ + BN_) * ROWB, PER = 8, RPI = 8, CPR = 8;
    constexpr int A_MODE = UC_A_DENSE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (UC_TRACE(p)) tr0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int nwg = p.tiles_m * p.tiles_n;
    const int ksplit = p.split_k > 1 ? (int)uc_div(blockIdx.x, p.dNwg) : 0;
    const int t = glds_xcd_remap((int)blockIdx.x - ksplit * nwg, nwg);
    int tm, tn;
    {   // tile order: see the 16-wave kernel
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int grp = (int)uc_div((unsigned)t, p.dPerGroup), within = t - grp * per_group;
        const int first_m = grp * GM;
        const bool last = p.tiles_m - first_m < GM;
        const int gsz = last ? p.tiles_m - first_m : GM;
        tn = (int)uc_div((unsigned)within, last ? p.dGmLast : p.dGm);
        tm = first_m + within - tn * gsz;
    }
    if (p.stagger > 0 && blockIdx.x < 256u) {
        const unsigned phase = (unsigned)tm & 7u;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)phase * (unsigned)p.stagger;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int64_t m0 = (int64_t)tm * BM_;
    const int64_t n0 = (int64_t)tn * BN_;
    const int64_t wave_m = m0 + wr * 128;
    const int64_t wave_n = n0 + wc * 64;
    const bool is_vt = p.vt_col0 >= 0 && wave_n >= p.vt_col0;
    const bool is_rope = !is_vt && p.rope_cols > 0 && wave_n < p.rope_cols;
    const int mode = is_vt ? 2 : (is_rope ? 1 : 0);

    // DMA sources.  Instruction I = wave*8 + q covers combined-tile rows [8 I, 8 I + 8): waves 0-3 stage the A rows, waves 4-7 the
    // W rows, 64 rows each.  The 8-row groups of a wave are one uniform stride apart, so a piece is addressed as a wave-uniform
    // 64-bit base (SGPR pair, advanced on the scalar unit) + one of TWO per-lane byte offsets (the chunk swizzle key
    // (row >> 1) & 7 only depends on the parity of q) — 2 address registers instead of 16.  Row groups past the matrix end are
    // clamped to its last 8 rows as a group (launcher: M % 8 == 0, N % 8 == 0); their products are never stored.
    const bool stages_a = wave < 4;
    const int64_t ld_src = stages_a ? p.lda : p.K;
    const bf16_t* sbase[PER];
    {
        const bf16_t* mat = stages_a ? p.A : p.W;
        const int64_t row0 = (stages_a ? m0 : n0) + (wave & 3) * 64, lim = (stages_a ? p.M : p.N) - 8;
#pragma unroll
        for (int q = 0; q < PER; ++q) sbase[q] = mat + min(row0 + 8 * q, lim) * ld_src;
    }
    const unsigned voff_row = (unsigned)((lane >> 3) * (int)ld_src) * 2u;
    const unsigned voff2[2] = {voff_row + (unsigned)(((lane & 7) ^ (lane >> 4)) << 4), voff_row + (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) << 4)};
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    auto issue_piece = [&](int stage, int64_t k0, int q) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE_BYTES + wave * (PER * 1024) + q * 1024));
        dma16_s_to_lds(voff2[q & 1], sbase[q] + k0, dst);
    };
    auto issue_stage = [&](int stage, int64_t k0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) issue_piece(stage, k0, q);
    };
    // side panel (see GLDS_SIDE_*): one 1-KiB piece per wave, issued BEFORE the first stage's pieces — loads return in order, so the
    // first K-step's vmcnt(0) + barrier cover it.  Out-of-range rows / columns are clamped (their products are never stored).
    const bool side_on = EPI == GLDS_EPI_BF16 && p.side_lds != 0;
    if (EPI == GLDS_EPI_BF16 && side_on) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(2 * STAGE_BYTES + wave * 1024));
        if (wave < 2) {             // (mean, rstd) of rows 128 wave + 2 lane, + 1
            const int r = (int)min((int64_t)(128 * wave + 2 * lane), p.M - 2 - m0);
            dma16_s_to_lds((unsigned)r * 8u, p.ln_stats + m0, dst);
        } else if (wave == 2) {     // column sums of columns 4 lane .. + 3
            const int c = (int)min((int64_t)(4 * lane), p.N - 4 - n0);
            dma16_s_to_lds((unsigned)c * 4u, p.ln_colsum + n0, dst);
        } else if (wave == 3) {     // bias
            const int c = (int)min((int64_t)(4 * lane), p.N - 4 - n0);
            dma16_s_to_lds((unsigned)c * 4u, p.bias + n0, dst);
        } else if (p.rope_cols > 0) {   // (y, x) of row 64 (wave - 4) + lane
            const int r = (int)min((int64_t)(64 * (wave - 4) + lane), p.M - 1 - m0);
            dma16_s_to_lds((unsigned)r * 16u, p.rope_pos + m0 * 2, dst);
        }
    }

    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int f_sw = glds_swz<64>(frow);
    const int a_base = (wr * 128 + frow) * ROWB;
    const int w_base = (BM_ + wc * 64 + frow) * ROWB;
    const int ch_off[2] = {((0 * 4 + fk) ^ f_sw) << 4, ((1 * 4 + fk) ^ f_sw) << 4};

    float4_t acc[FA][4];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk_total = (int)(p.K / 64);
    const int nk_per = (nk_total + p.split_k - 1) / p.split_k;
    const int kt0 = ksplit * nk_per;
    int nk = max(0, min(nk_per, nk_total - kt0));
    if UC_DBG(p, 8) nk = min(nk, 1);
    const int64_t kbase = (int64_t)kt0 * 64;

    auto main_loop = [&](auto swap_tag) __attribute__((always_inline)) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        if (nk <= 0) return;
        bf16x8_t a[FA], w[4], wn[4];
        auto rd_a = [&](const char* st, int ks, int i) __attribute__((always_inline)) {
            return *reinterpret_cast<const bf16x8_t*>(st + a_base + ch_off[ks] + i * 16 * ROWB);
        };
        auto rd_w = [&](const char* st, int ks, int j) __attribute__((always_inline)) {
            return *reinterpret_cast<const bf16x8_t*>(st + w_base + ch_off[ks] + j * 16 * ROWB);
        };
        auto mma = [&](int i, int j, bf16x8_t av, bf16x8_t wv) __attribute__((always_inline)) {
            if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, av, acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wv, acc[i][j], 0, 0, 0);
        };
        issue_stage(0, kbase);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (UC_TRACE(p)) tr1 = __builtin_amdgcn_s_memrealtime();
        asm volatile("" ::: "memory");
        if (nk > 1) issue_stage(1, kbase + 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = rd_w(smem, 0, j);
#pragma unroll
        for (int i = 0; i < FA; ++i) a[i] = rd_a(smem, 0, i);
        // chunk 0 of a stage: MFMAs on (a, w) while chunk 1 of the same stage streams into (a, wn).  The scheduling fences keep
        // every refill of a[i] behind the MFMAs that read the old a[i]: hoisted, both generations are live and the loop spills.
        auto chunk0 = [&](const char* cur) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FA; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma(i, j, a[i], w[j]);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) {      // behind the first MFMA group: in front of it the group would wait for these four reads as well
#pragma unroll
                    for (int j = 0; j < 4; ++j) wn[j] = rd_w(cur, 1, j);
                }
                a[i] = rd_a(cur, 1, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // One K-step but the last: chunk 0, the mid-step synchronisation, chunk 1 with the next stage's first fragments streaming
        // in.  DMA (compile-time): the DMA of stage kt + 2 goes into the buffer just released, one 1-KiB piece behind each MFMA
        // group (issued as a block in front of chunk 1 its ~80 scalar instructions hold back both waves of a SIMD right after
        // the barrier).  The loop bodies are straight-line: with the last steps' DMA-free / prefetch-free forms inside one loop
        // the accumulators went through phi copies (90 of them parked in scratch) and every join cost an lgkmcnt(0).
        // (the loops are rotated so that their headers sit at the synchronisation point: hipcc drains lgkmcnt at a loop header
        // whatever is pending, and there the drain is wanted)
        auto step = [&](int kt, auto dma_tag) __attribute__((always_inline)) {
            constexpr bool DMA = decltype(dma_tag)::value;
            const char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            // every fragment of stage kt is in registers (lgkmcnt(0)), this wave's DMA pieces of the next stage have landed
            // (vmcnt(0)); after the barrier both hold for the whole workgroup
            if UC_DBG(p, 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // diagnostics: do not wait for the DMA (wrong results)
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int64_t k2 = kbase + (int64_t)(kt + 2) * 64;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = rd_w(nxt, 0, j);
#pragma unroll
            for (int i = 0; i < FA; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma(i, j, a[i], wn[j]);
                __builtin_amdgcn_sched_barrier(0);
                a[i] = rd_a(nxt, 0, i);
                if constexpr (DMA) { if (!UC_DBG(p, 1)) issue_piece(kt & 1, k2, i); }
                __builtin_amdgcn_sched_barrier(0);
            }
            chunk0(nxt);
        };
        chunk0(smem);
        for (int kt = 0; kt + 2 < nk; ++kt) step(kt, std::true_type{});
        if (nk > 1) step(nk - 2, std::false_type{});
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma(i, j, a[i], wn[j]);
    };
    if (mode == 2) main_loop(std::false_type{}); else main_loop(std::true_type{});

#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((opencl_constant)) GldsParams* kp =
        (const __attribute__((opencl_constant)) GldsParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp)::"memory");
    glds_pe_t pe = *kp;
#else
    glds_pe_t pe = p;
#endif
    if UC_DBG(pe, 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s == 12345.678f) reinterpret_cast<float*>(pe.C)[0] = s;
        return;
    }
    __builtin_amdgcn_s_barrier();          // every wave is done with the ring: it becomes the bounce space (8 KiB per wave)
    asm volatile("" ::: "memory");
    if (UC_TRACE(pe)) tr2 = __builtin_amdgcn_s_memrealtime();
    glds_epilogue_dispatch<FA, A_MODE, EPI>(pe, acc, mode, wave_m, wave_n, tid, wave, ksplit, smem,
                                            (EPI == GLDS_EPI_BF16 && pe.side_lds) ? smem + 2 * STAGE_BYTES : nullptr);
    if (UC_TRACE(pe)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* t = UC_TRACE(pe) + (size_t)blockIdx.x * 6;
            t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memrealtime(); t[4] = hw; t[5] = xcc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Four-wave form of the dense 256x256x64 tile: 2 x 2 waves of 128x128, ONE wave per SIMD, 256 accumulators per lane pinned in
// a0..a255 and a hand-scheduled K-loop (gemm_glds4_loop.inc, generated by gen/gen_glds4_loop.py) — the layout of the vendor's
// hand-written kernels: 32 ds_read_b128 feed 128 MFMAs per wave per K-step (256 B of LDS per MFMA against 512 B for the 64x64 wave
// tiles of the 16-wave kernel), so the same FLOPs cost less LDS traffic — less energy, on a part that runs this kernel at its power
// cap.  hipcc cannot schedule this form (accumulator copies and scratch in the loop), hence the inline-asm loop; the
// prologue (tile order, DMA source set-up) and the epilogues are the shared C++ ones: after the loop each wave drains its 128x128
// tile as two 128x64 halves through glds_epilogue_dispatch<8, ...>.
//
// Measured (round 3).  First schedule (all 16 pieces behind ONE mid-step barrier with vmcnt(0)): on par with the
// 16-wave kernel (8192^3: 1384-1400 vs 1343-1412 TFLOP/s) — a piece had 1000-2000 cycles to land, less than the loaded latency.  Second
// schedule (two barriers per step, counted vmcnt(16), a piece behind every fifth MFMA: gen/gen_glds4_loop.py `step`): 8192^3 1535
// (the vendor's hand-written kernel of the same design: 1543), K = 4096 shapes +5.6 % over the 16-wave kernel, K = 1024 shapes +-1 %
// (bound by the ~5.6 us a tile costs outside its K-loop).  Routed for K >= 2048 (UC_GEMM_4WAVE, UC_GEMM_4WAVE_MIN_K); gemm_variant 7.
//
// LDS image, stage ring, chunk swizzle and DMA piece addressing are those of the eight-wave kernel; waves 0-1 stage the A rows, waves
// 2-3 the W rows (128 rows = 16 pieces of 1 KiB each per stage).  The accumulators cross from the asm statement to the epilogue in
// the physical registers a0..a255 (read out by the v_accvgpr_read statements of UC_GLDS4_READ_HALF*): nothing between the loop and
// the last read-out may allocate an AGPR — tools/check_glds4_agprs.py verifies that on the compiled code (build.py runs it).
#include "gemm_glds4_loop.inc"

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_bf16_glds4_kernel(GldsParams p) {
    constexpr int BM_ = 256, BN_ = 256, ROWB = 128, A_MODE = UC_A_DENSE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int t = glds_xcd_remap((int)blockIdx.x, nwg);
    int tm, tn;
    {   // tile order: see the 16-wave kernel
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int grp = (int)uc_div((unsigned)t, p.dPerGroup), within = t - grp * per_group;
        const int first_m = grp * GM;
        const bool last = p.tiles_m - first_m < GM;
        const int gsz = last ? p.tiles_m - first_m : GM;
        tn = (int)uc_div((unsigned)within, last ? p.dGmLast : p.dGm);
        tm = first_m + within - tn * gsz;
    }
    if (p.stagger > 0 && blockIdx.x < 256u) {
        // de-phase the CUs by row panel (see the 16-wave kernel): the epilogues' store / residual traffic as a stream, not a burst per round
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)((unsigned)tm & 7u) * (unsigned)p.stagger;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int64_t m0 = (int64_t)tm * BM_;
    const int64_t n0 = (int64_t)tn * BN_;
    const int64_t wave_m = m0 + wr * 128;
    const int64_t wave_n = n0 + wc * 128;
    // (launcher: rope_cols and vt_col0 are multiples of 128, so a wave's two 64-column halves share a mode)
    const bool is_vt = p.vt_col0 >= 0 && wave_n >= p.vt_col0;
    const bool is_rope = !is_vt && p.rope_cols > 0 && wave_n < p.rope_cols;
    const int mode = is_vt ? 2 : (is_rope ? 1 : 0);

    // DMA slab of this wave: 128 rows of A (waves 0, 1) or of W (waves 2, 3), 16 pieces of 8 rows
    const bool stages_a = wave < 2;
    const bf16_t* mat = stages_a ? p.A : p.W;
    const unsigned pitch = (unsigned)((stages_a ? p.lda : p.K) * 2);                       // bytes (launcher: < 2^31)
    const unsigned row0 = (unsigned)((stages_a ? m0 : n0) + (wave & 1) * 128);
    const unsigned lim = (unsigned)((stages_a ? p.M : p.N) - 8);
    const unsigned long long mat_u = (unsigned long long)mat;
    const unsigned base_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mat_u);
    const unsigned base_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mat_u >> 32));
    const unsigned voff_row = (unsigned)(lane >> 3) * pitch;
    const unsigned voff0 = voff_row + (unsigned)(((lane & 7) ^ (lane >> 4)) << 4);
    const unsigned voff1 = voff_row + (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) << 4);
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned lds_dma = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)wave * 16384u));
    // fragment read addresses of stage 0 (K halves 0 / 1): row frow of the wave's block 0, chunk (4 ks + fk) ^ swizzle(frow)
    const int frow = lane & 15, fk = lane >> 4, f_sw = glds_swz<64>(frow);
    const unsigned ch0 = (unsigned)(((0 * 4 + fk) ^ f_sw) << 4), ch1 = (unsigned)(((1 * 4 + fk) ^ f_sw) << 4);
    const unsigned a_row = lds_base + (unsigned)((wr * 128 + frow) * ROWB), w_row = lds_base + (unsigned)((BM_ + wc * 128 + frow) * ROWB);
    const unsigned nk = (unsigned)(p.K / 64);
    if (mode == 2) {
        asm volatile(UC_GLDS4_LOOP_NOSWAP
                     :
                     : "v"(a_row + ch0), "v"(a_row + ch1), "v"(w_row + ch0), "v"(w_row + ch1), "v"(voff0), "v"(voff1), "s"(base_lo), "s"(base_hi),
                       "s"(row0), "s"(lim), "s"(pitch), "s"(lds_dma), "s"(nk)
                     : UC_GLDS4_CLOBBERS);
    } else {
        asm volatile(UC_GLDS4_LOOP_SWAP
                     :
                     : "v"(a_row + ch0), "v"(a_row + ch1), "v"(w_row + ch0), "v"(w_row + ch1), "v"(voff0), "v"(voff1), "s"(base_lo), "s"(base_hi),
                       "s"(row0), "s"(lim), "s"(pitch), "s"(lds_dma), "s"(nk)
                     : UC_GLDS4_CLOBBERS);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((opencl_constant)) GldsParams* kp =
        (const __attribute__((opencl_constant)) GldsParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp)::"memory");
    glds_pe_t pe = *kp;
#else
    glds_pe_t pe = p;
#endif
    __builtin_amdgcn_s_barrier();          // every wave is done with the ring: it becomes the bounce space (8 KiB per wave)
    asm volatile("" ::: "memory");
    {
        float4_t acc[8][4];
        UC_GLDS4_READ_HALF0(acc)
        glds_epilogue_dispatch<8, A_MODE, EPI>(pe, acc, mode, wave_m, wave_n, tid, wave, 0, smem);
    }
    {
        float4_t acc[8][4];
        UC_GLDS4_READ_HALF1(acc)
        glds_epilogue_dispatch<8, A_MODE, EPI>(pe, acc, mode, wave_m, wave_n + 64, tid, wave, 0, smem);
    }
}

template <int EPI>
static void launch_glds4(GldsParams p, hipStream_t st) {
    p.tiles_m = (int)ceil_div64(p.M, 256);
    p.tiles_n = (int)ceil_div64(p.N, 256);
    p.dNwg = uc_make_fastdiv((unsigned)(p.tiles_m * p.tiles_n));
    p.dPerGroup = uc_make_fastdiv((unsigned)(p.group_m * p.tiles_n));
    p.dGm = uc_make_fastdiv((unsigned)p.group_m);
    p.dGmLast = uc_make_fastdiv((unsigned)std::max(1, p.tiles_m % p.group_m));
    auto kfn = gemm_bf16_glds4_kernel<EPI>;
    constexpr int smem = 2 * 512 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
}

// what the four-wave kernel takes: whole 8-row groups, one mode per 128-column wave tile, no split-K, 32-bit row pitches
static inline bool glds4_ok(const GldsParams& p) {
    return p.a_mode == UC_A_DENSE && p.M % 8 == 0 && p.N % 8 == 0 && p.M >= 8 && p.N >= 8 && p.split_k <= 1 && p.K >= 64 &&
           (p.vt_col0 < 0 || p.vt_col0 % 128 == 0) && (p.rope_cols <= 0 || p.rope_cols % 128 == 0) && p.lda * 2 < ((int64_t)1 << 31) &&
           p.K * 2 < ((int64_t)1 << 31) && p.M < ((int64_t)1 << 31) && p.N < ((int64_t)1 << 31);
}

template <int EPI>
static void launch_glds8(GldsParams p, hipStream_t st) {
    p.tiles_m = (int)ceil_div64(p.M, 256);
    p.tiles_n = (int)ceil_div64(p.N, 256);
    p.dNwg = uc_make_fastdiv((unsigned)(p.tiles_m * p.tiles_n));
    p.dPerGroup = uc_make_fastdiv((unsigned)(p.group_m * p.tiles_n));
    p.dGm = uc_make_fastdiv((unsigned)p.group_m);
    p.dGmLast = uc_make_fastdiv((unsigned)std::max(1, p.tiles_m % p.group_m));
    auto kfn = gemm_bf16_glds8_kernel<EPI>;
    constexpr int smem = 2 * 512 * 128 + (EPI == GLDS_EPI_BF16 ? GLDS_SIDE_BYTES : 0);
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    p.side_lds = (EPI == GLDS_EPI_BF16 && uc_knobs().gemm_side_lds && p.ln_stats && !p.ln_partial && p.ln_colsum && p.bias && p.split_k <= 1 &&
                  !(p.dbg & 16) && al16(p.ln_stats) && al16(p.ln_colsum) && al16(p.bias) &&
                  (p.rope_cols <= 0 || (p.rope_pos && al16(p.rope_pos)))) ? 1 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)p.tiles_m * p.tiles_n * (unsigned)p.split_k), dim3(512), smem, st, p);
}

template <int BM_, int BN_, int WM_, int WN_, int STAGES, int A_MODE, int BK_ = 64, int WGS_PER_CU = 1, int EPI = GLDS_EPI_ALL, bool F16 = false>
static void launch_variant_mode(GldsParams p, hipStream_t st) {
    p.tiles_m = (int)ceil_div64(p.M, BM_);
    p.tiles_n = (int)ceil_div64(p.N, BN_);
    p.dNwg = uc_make_fastdiv((unsigned)(p.tiles_m * p.tiles_n));
    p.dPerGroup = uc_make_fastdiv((unsigned)(p.group_m * p.tiles_n));
    p.dGm = uc_make_fastdiv((unsigned)p.group_m);
    p.dGmLast = uc_make_fastdiv((unsigned)std::max(1, p.tiles_m % p.group_m));
    auto kfn = gemm_bf16_glds_kernel<BM_, BN_, WM_, WN_, STAGES, A_MODE, BK_, WGS_PER_CU, EPI, F16>;
    constexpr int smem = STAGES * (BM_ + BN_) * BK_ * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    const unsigned slices = (BM_ == 128 && BN_ == 128 && p.fuse_split2) ? 2u : (unsigned)p.split_k;
    hipLaunchKernelGGL(kfn, dim3((unsigned)p.tiles_m * p.tiles_n * slices), dim3(WM_ * WN_ * 64), smem, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 / pad 1 / stride 1 convolution on maps at least 64 wide, ROW-WALKING form (round 3).  The implicit-GEMM kernel above stages
// a 256-pixel x 64-channel A tile per (tap, channel chunk): every input pixel crosses the LDS-DMA path nine times.  In the NT
// layout an LDS row IS a pixel, so the horizontal taps kx are ROW OFFSETS of the fragment reads: here a stage of A is the slab
// of input pixels a tile needs for ONE kernel row ky and 64 channels — for each of the tile's R = 256 / seg image-row segments the
// seg + 2 pixels ox0 - 1 .. ox0 + seg of input row oy + ky - 1 (zeros outside the image, through the descriptor's range check) — and
// the three taps (ky, 0..2) are three MFMA passes over it, each with its own 128 x 64 weight tile.  A crosses the DMA path three
// times instead of nine, with no per-tap masks; the weight tiles are what they were.
//   tile 256 pixels x 128 output channels, 8 waves of 64 x 64 (the accumulator layout of the <256, 128, 4, 2, ..> variants: the
//   shared epilogues apply unchanged); K order (ky, channel chunk, kx).
//   LDS: two slabs of 320 rows x 128 B (5 pieces per wave, the last ones may lie outside the slab: zeros) + a ring of three
//   weight tiles of 16 KiB = 128 KiB.  Unit = one (ky, chunk, kx): its weight tile, and with kx == 0 the slab; DMA two units ahead;
//   a wave waits for its own pieces of the NEXT unit only: vmcnt(2), vmcnt(2), vmcnt(7) round the three taps.
template <int EPI, bool F16>
__global__ __launch_bounds__(512, 2) void conv3x3_rows_kernel(GldsParams p) {
    constexpr int BM_ = 256, BN_ = 128, ROWB = 128, SLAB_ROWS = 320, SLAB_BYTES = SLAB_ROWS * ROWB, WT_BYTES = BN_ * ROWB;
    constexpr int A_MODE = UC_A_CONV3X3, FA = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int t = glds_xcd_remap((int)blockIdx.x, nwg);
    int tm, tn;
    {   // tile order: see the 16-wave kernel
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int grp = (int)uc_div((unsigned)t, p.dPerGroup), within = t - grp * per_group;
        const int first_m = grp * GM;
        const bool last = p.tiles_m - first_m < GM;
        const int gsz = last ? p.tiles_m - first_m : GM;
        tn = (int)uc_div((unsigned)within, last ? p.dGmLast : p.dGm);
        tm = first_m + within - tn * gsz;
    }
    const int64_t m0 = (int64_t)tm * BM_, n0 = (int64_t)tn * BN_;
    const int64_t wave_m = m0 + wr * 64, wave_n = n0 + wc * 64;

    // ---- tile geometry (wave-uniform): R row segments of seg pixels, first at (image row id rowid0 = b * H + oy0, ox0) ----
    const int W_ = p.cW, H_ = p.cH, Cin = p.cCin;
    const int seg = min(BM_, W_), R = BM_ / seg, segp = seg + 2;
    const unsigned rowid0 = uc_div((unsigned)m0, p.dWo);
    const int ox0 = (int)((unsigned)m0 - rowid0 * (unsigned)W_);
    const unsigned b0 = uc_div(rowid0, p.dHo);
    const int oy0 = (int)(rowid0 - b0 * (unsigned)H_);
    // descriptors: the slab's source window shifted back by one image row + one pixel (tap row ky and the chunk are a uniform
    // non-negative soffset), and the tile's weight rows
    const unsigned long long pa = (unsigned long long)(p.A + (((int64_t)rowid0 - 1) * W_ + (ox0 - 1)) * Cin);
    const unsigned long long pw = (unsigned long long)(p.W + n0 * p.K);
    const uint4_t srd_a = (uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
    const uint4_t srd_w = (uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pw), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pw >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
    // slab pieces of this wave: piece n = wave + 8 q covers slab rows 8 n .. 8 n + 7; lane -> row 8 n + lane / 8, physical chunk lane % 8
    unsigned sl_off[5], sl_mask[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int row = (wave + 8 * q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ glds_swz<64>(row);
        const int rr = row / segp, xs = row - rr * segp;             // (division by a wave-uniform value, once per tile)
        const int ix = ox0 - 1 + xs;
        sl_off[q] = (unsigned)((((int64_t)rr * W_ + xs) * Cin + c * 8) * 2);
        unsigned mask = 0;
        if (rr < R && ix >= 0 && ix < W_) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) mask |= ((unsigned)(oy0 + rr + ky - 1) < (unsigned)H_ ? 1u : 0u) << ky;
        }
        sl_mask[q] = mask;
    }
    unsigned w_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ glds_swz<64>(row);
        w_off[q] = (unsigned)(((int64_t)row * p.K + c * 8) * 2);
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned lds_w = lds_base + 2u * SLAB_BYTES;
    const int nch = Cin / 64;
    const int nunit = 9 * nch;
    auto issue_unit = [&](int u) {                     // u = (ky * nch + chunk) * 3 + kx, wave-uniform
        const int sstep = u / 3, kx = u - 3 * sstep;
        const int ky = sstep / nch, ch = sstep - ky * nch;
        if (kx == 0) {
            const unsigned soff = (unsigned)((((int64_t)ky * W_) * Cin + ch * 64) * 2);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const unsigned vo = ((sl_mask[q] >> ky) & 1u) ? sl_off[q] : 0xffffffffu;
                dma16_buf_to_lds(vo, srd_a, soff, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((sstep & 1) * SLAB_BYTES + (wave + 8 * q) * 1024)));
            }
        }
        const unsigned soff_w = (unsigned)((((ky * 3 + kx) * Cin) + ch * 64) * 2);
        const int slot = u % 3;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            dma16_buf_to_lds(w_off[q], srd_w, soff_w, __builtin_amdgcn_readfirstlane(lds_w + (unsigned)(slot * WT_BYTES + (wave * 2 + q) * 1024)));
    };

    // ---- fragment addressing: A fragment i of this wave = tile pixels wr * 64 + 16 i + frow = slab row base_i + frow, base_i =
    //      m_i + 2 (m_i / seg) (two halo pixels per row segment before it), + kx for the tap; W fragments as in the main kernel ----
    const int frow = lane & 15, fk = lane >> 4;
    int a_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mi = wr * 64 + 16 * i;
        a_row[i] = mi + 2 * (mi / seg) + frow;
    }
    const int w_sw = glds_swz<64>(frow);
    const int w_base = (wc * 64 + frow) * ROWB;
    const int w_ch[2] = {((0 * 4 + fk) ^ w_sw) << 4, ((1 * 4 + fk) ^ w_sw) << 4};

    float4_t acc[FA][4];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    auto compute_unit = [&](const char* slab, const char* wt, int kx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[FA], wf[4];
#pragma unroll
            for (int i = 0; i < FA; ++i) {
                const int row = a_row[i] + kx;
                uint4 raw = *reinterpret_cast<const uint4*>(slab + row * ROWB + (((ks * 4 + fk) ^ glds_swz<64>(row)) << 4));
                if (p.relu_a) raw = glds_relu_bf16x8(raw);
                af[i] = __builtin_bit_cast(bf16x8_t, raw);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(wt + w_base + w_ch[ks] + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = glds_mfma<F16>(wf[j], af[i], acc[i][j]);
        }
    };

    issue_unit(0);
    if (nunit > 1) issue_unit(1);
    for (int sstep = 0; sstep < 3 * nch; ++sstep) {
        const char* slab = smem + (sstep & 1) * SLAB_BYTES;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int u = sstep * 3 + kx;
            // this wave's pieces of unit u have landed when only unit u + 1's are outstanding: 2 weight pieces (+ 5 slab pieces if
            // it opens a new slab)
            if (u + 1 >= nunit) wait_vmcnt<0>();
            else if (kx == 2) wait_vmcnt<7>();
            else wait_vmcnt<2>();
            __builtin_amdgcn_s_barrier();          // ... everyone's have; and every wave is done with unit u - 1 (its ring slot is unit u + 2's)
            asm volatile("" ::: "memory");
            if (u + 2 < nunit) issue_unit(u + 2);
            compute_unit(slab, smem + 2 * SLAB_BYTES + (u % 3) * WT_BYTES, kx);
        }
    }

#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((opencl_constant)) GldsParams* kp =
        (const __attribute__((opencl_constant)) GldsParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp)::"memory");
    glds_pe_t pe = *kp;
#else
    glds_pe_t pe = p;
#endif
    __syncthreads();     // every wave is done with the slabs and the ring: they become the epilogue's bounce space (8 KiB per wave)
    glds_epilogue_dispatch<FA, A_MODE, EPI, F16>(pe, acc, 0, wave_m, wave_n, tid, wave, 0, smem);
}

// shapes the row-walking conv kernel takes: stride 1, whole 64-channel chunks, 128-column tiles, maps 64 .. wide whose rows tile 256
// pixels exactly (a tile = whole row segments of one image), 32-bit source windows
static inline bool conv_rows_ok(const GldsParams& p) {
    if (p.a_mode != UC_A_CONV3X3 || p.cStride != 1 || p.cCin % 64 != 0 || p.N % 128 != 0 || p.split_k > 1) return false;
    const int W = p.cW, H = p.cH;
    if (W < 64 || !(W % 256 == 0 || 256 % W == 0)) return false;
    const int R = W >= 256 ? 1 : 256 / W;
    if (H % R != 0 || p.M % 256 != 0) return false;
    return ((int64_t)(R + 3) * W + 4) * p.cCin * 2 < ((int64_t)1 << 31) && p.N * p.K * 2 < ((int64_t)1 << 31) && p.M < ((int64_t)1 << 30);
}

template <int EPI, bool F16>
static void launch_conv_rows(GldsParams p, hipStream_t st) {
    p.tiles_m = (int)(p.M / 256);
    p.tiles_n = (int)(p.N / 128);
    p.dNwg = uc_make_fastdiv((unsigned)(p.tiles_m * p.tiles_n));
    p.dPerGroup = uc_make_fastdiv((unsigned)(p.group_m * p.tiles_n));
    p.dGm = uc_make_fastdiv((unsigned)p.group_m);
    p.dGmLast = uc_make_fastdiv((unsigned)std::max(1, p.tiles_m % p.group_m));
    auto kfn = conv3x3_rows_kernel<EPI, F16>;
    constexpr int smem = 2 * 320 * 128 + 3 * 128 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)p.tiles_m * p.tiles_n), dim3(512), smem, st, p);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Eight-wave row-walking 3x3 convolution: 512 pixels x 128 output channels, 4 x 2 waves of 128 x 64 (the eight-wave GEMM's wave tile:
// 12 fragment reads feed 32 MFMAs, 384 B of LDS per MFMA against 512 B for the 64 x 64 wave tiles above), walked in 32-channel chunks.
//
//   super-step (ky, 32-channel chunk) = three units kx = 0, 1, 2 of 32 MFMAs per wave each.  Staged per super-step: the slab of input
//   pixels of kernel row ky (R row segments of seg + 2 pixels, 64-B rows: 33 pieces of 1 KiB) and the three taps' weight tiles
//   (128 x 32: 8 pieces each) — every input pixel crosses the LDS-DMA path three times, every weight tile once per 512 pixels
//   (57 KiB per super-step, 684 KiB per 512 x 128 tile at 128 input channels; the 256-pixel kernel above: 493 KiB per 256 pixels).
//   Two buffers of each (2 x 33 + 2 x 24 KiB = 114 KiB), one workgroup per CU, two waves per SIMD with 256 registers:
//
//   unit 0:  MFMAs on (a, w)   | ds_read unit 1 -> (a, wn)      (a is refilled row block by row block behind its MFMAs)
//   unit 1:  MFMAs on (a, wn)  | ds_read unit 2 -> (a, w)
//   unit 2, first four MFMA groups on (a, w)
//   s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: everyone has READ this super-step's buffers, everyone's DMA of the next one has landed —
//            with 256 matrix-pipe cycles per wave queued, so the pipe works through the barrier's skew
//   unit 2, last four groups   | ds_read unit 0 of the next super-step -> (a, wn)
//   DMA of the super-step after the next -> the buffers just released: 7-8 pieces per wave (4-5 of the slab, 3 of the weights), one
//            behind every other MFMA group from the synchronisation to the middle of the next unit 1 (the global -> LDS path takes ~40
//            cycles per piece and CU: 57 pieces are most of a super-step; >= 1.5 units of flight each)
//
//   One barrier per 96 MFMAs of a wave.  The weight registers alternate roles from one super-step to the next (three units): the loop
//   body is a PAIR of super-steps (launcher: Cin % 64 == 0), straight-line.  Measured against the first form (synchronisation at the
//   head of unit 2, all pieces inside unit 2, a zero-writing dump piece in the odd slot of seven waves): +1-3 % (fp16 operands), level (bf16).
//   64-B LDS rows with the chunk key 3 ((row >> 2) & 1): conflict-free ds_read_b128 fragments at every row offset (the taps shift
//   the 16 rows of a fragment by kx, row segments by two halo pixels each).
//   Accumulator layout = the eight-wave GEMM's (FA = 8): the shared epilogues apply; the fused 1x1 tail drains a 128-row wave tile
//   as two 64-row halves.
//
//   FLAT (round 6): maps whose rows do NOT tile 512 pixels (148 / 296 / 592-wide: the DINOv2-518 head; 56 / 112 / 224: 224 x 224 pairs; any
//   pixel count).  A tile is 512 CONSECUTIVE output pixels of the flattened (image, y, x) order — whole or partial rows, across
//   image borders, the last tile masked by the epilogues' row bound — and the slab of kernel row ky is the 514 consecutive INPUT pixels
//   m0 + (ky - 1) W - 1 ... : tap (ky, kx) of output pixel o is input pixel o + (ky - 1) W + (kx - 1) of the same flat order, so the
//   fragment of tap kx is again the slab shifted by kx rows, with no per-segment halo.  What the flat order gets wrong is the PADDING:
//   vertically (the pixel W back / ahead of an image's first / last row belongs to the neighbouring image) the DMA zero-fills a slab
//   row whose OWN pixel (y', x') has y' - (ky - 1) outside the map — every use of that row by an existing output with the tap inside
//   its row is then right; horizontally (kx = 0 at x = 0, kx = 2 at x = W - 1 read the neighbouring ROW's edge pixel, which other
//   outputs need as it is) the fragment's lanes of those pixels are zeroed in registers.  Which of a wave's 128 pixels sit in column 0 /
//   W - 1 is wave-uniform: two 128-bit maps in 8 SGPRs (16 lane masks in SGPR pairs is more scalar state than the kernel has room for:
//   the allocator then hands an inline-asm "s" operand a VGPR); per fragment of the kx = 0 and kx = 2 units the 16-bit slice is
//   replicated over the four 16-lane groups into VCC (4 scalar instructions) and four v_cndmask_b32 zero the lanes — about what
//   the ReLU-on-load form already spends there.  (volatile: hoisted out of the loop the sixteen masks would be live SGPR pairs again.)
template <int SLICE>
__device__ __forceinline__ bf16x8_t glds_zero_edge_lanes(bf16x8_t v, unsigned map_dword) {
    uint4 q = __builtin_bit_cast(uint4, v);
    unsigned t;
    asm volatile(
        "s_bfe_u32 %4, %5, %6\n\t"
        "s_mul_i32 %4, %4, 0x10001\n\t"
        "s_mov_b32 vcc_lo, %4\n\t"
        "s_mov_b32 vcc_hi, %4\n\t"
        "v_cndmask_b32_e64 %0, %0, 0, vcc\n\t"
        "v_cndmask_b32_e64 %1, %1, 0, vcc\n\t"
        "v_cndmask_b32_e64 %2, %2, 0, vcc\n\t"
        "v_cndmask_b32_e64 %3, %3, 0, vcc"
        : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w), "=&s"(t)
        : "s"(map_dword), "n"((16 * SLICE) | (16 << 16))
        : "vcc");
    return __builtin_bit_cast(bf16x8_t, q);
}

template <int EPI, bool F16, bool RELU_A, bool FLAT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_rows8_kernel(GldsParams p) {
    constexpr int BM_ = 512, BN_ = 128, ROWB = 64, NPIECE = 33, SLAB_BYTES = NPIECE * 1024, WT_BYTES = 3 * BN_ * ROWB;
    constexpr int WT0 = 0, SLAB0 = 2 * WT_BYTES;              // LDS image: weights[2] | slab[2]
    constexpr int A_MODE = UC_A_CONV3X3, FA = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int t = glds_xcd_remap((int)blockIdx.x, nwg);
    int tm, tn;
    {   // tile order: see the 16-wave kernel
        const int GM = p.group_m;
        const int per_group = GM * p.tiles_n;
        const int grp = (int)uc_div((unsigned)t, p.dPerGroup), within = t - grp * per_group;
        const int first_m = grp * GM;
        const bool last = p.tiles_m - first_m < GM;
        const int gsz = last ? p.tiles_m - first_m : GM;
        tn = (int)uc_div((unsigned)within, last ? p.dGmLast : p.dGm);
        tm = first_m + within - tn * gsz;
    }
    const int64_t m0 = (int64_t)tm * BM_, n0 = (int64_t)tn * BN_;
    const int64_t wave_m = m0 + wr * 128, wave_n = n0 + wc * 64;

    // ---- tile geometry (wave-uniform): R row segments of seg pixels, first at (image row id rowid0 = b * H + oy0, ox0) ----
    const int W_ = p.cW, H_ = p.cH, Cin = p.cCin;
    const int seg = FLAT ? BM_ : min(BM_, W_), R = BM_ / seg, segp = seg + 2;
    const unsigned rowid0 = uc_div((unsigned)m0, p.dWo);
    const int ox0 = (int)((unsigned)m0 - rowid0 * (unsigned)W_);
    const unsigned b0 = uc_div(rowid0, p.dHo);
    const int oy0 = (int)(rowid0 - b0 * (unsigned)H_);
    // slab row 0 of kernel row 0: segmented form = pixel (row above the tile's first, one left of its first column); flat form = flat
    // pixel m0 - W - 1.  (Either may lie before the tensor: lanes that would read there are masked — the descriptor base is only an origin.)
    const unsigned long long pa = FLAT ? (unsigned long long)(p.A + ((int64_t)m0 - 1 - W_) * Cin)
                                       : (unsigned long long)(p.A + (((int64_t)rowid0 - 1) * W_ + (ox0 - 1)) * Cin);
    const unsigned long long pw = (unsigned long long)(p.W + n0 * p.K);
    const uint4_t srd_a = (uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
    const uint4_t srd_w = (uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pw), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pw >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
    // slab pieces of this wave: piece n = wave + 8 q covers slab rows 16 n .. 16 n + 15; lane -> row 16 n + lane / 4, physical chunk lane % 4
    // (piece 32 = q 4 of wave 0 only).  sl_mask: bit 3 q + ky = the lane's pixel exists for tap row ky
    unsigned sl_off[5], sl_mask = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int row = (wave + 8 * q) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ glds_swz<32>(row);
        if constexpr (FLAT) {
            sl_off[q] = (__umul24((unsigned)row, (unsigned)Cin) + (unsigned)c * 8u) * 2u;
            asm volatile("" : "+v"(sl_off[q]));
            const int g = (int)m0 - 1 + row;                          // the slab row's own pixel (kernel row 1), flat; M < 2^30 (launcher)
            if (row < BM_ + 2 && g >= 0 && (int64_t)g < p.M && (q < 4 || wave == 0)) {
                const unsigned rid = uc_div((unsigned)g, p.dWo);      // image row id b * H + y
                const int yg = (int)(rid - uc_div(rid, p.dHo) * (unsigned)H_);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) sl_mask |= ((unsigned)(yg + ky - 1) < (unsigned)H_ ? 1u : 0u) << (3 * q + ky);
            }
        } else {
            const int rr = row / segp, xs = row - rr * segp;             // (division by a wave-uniform value, once per tile)
            const int ix = ox0 - 1 + xs;
            // (launcher: 32-bit windows, pixel and channel counts below 2^24: 24-bit multiplies — the 32-bit product of a v_mad_u64_u32
            //  lives in a register PAIR for the whole loop)
            sl_off[q] = (__umul24(__umul24((unsigned)rr, (unsigned)W_) + (unsigned)xs, (unsigned)Cin) + (unsigned)c * 8u) * 2u;
            asm volatile("" : "+v"(sl_off[q]));
            if (rr < R && ix >= 0 && ix < W_ && (q < 4 || wave == 0)) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) sl_mask |= ((unsigned)(oy0 + rr + ky - 1) < (unsigned)H_ ? 1u : 0u) << (3 * q + ky);
            }
        }
    }
    unsigned w_off;
    {
        const int row = wave * 16 + (lane >> 2);
        const int c = (lane & 3) ^ glds_swz<32>(row);
        w_off = (__umul24((unsigned)row, (unsigned)p.K) + (unsigned)c * 8u) * 2u;
        asm volatile("" : "+v"(w_off));
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    // both halves of the fifth slot's EXEC.  (Scalar by construction: as `readfirstlane(wave == 0 ? -1 : 0)` hipcc drops the readfirstlane
    //  of a value it knows to be uniform, may still SELECT it on the vector unit, and then hands the VGPR to the asm's "s" operand.)
    unsigned wave0_exec;
    asm volatile("s_cmp_eq_u32 %1, 0\n\ts_cselect_b32 %0, -1, 0" : "=s"(wave0_exec) : "s"(wave) : "scc");
    const int nch = Cin / 32;
    const int S = 3 * nch;                                   // super-steps, even (launcher)
    // piece j = 0..7 of super-step s -> buffer pair b: slab pieces q = j (j < 5), weight pieces of tap kx = j - 5
    auto issue_piece = [&](int ky, int ch, int b, int j) __attribute__((always_inline)) {
        if (j < 5) {
            // (piece 32 is wave 0's; the other waves issue their fifth slot with EXEC = 0: a branch here would split the MFMA stream into
            //  basic blocks, each join draining lgkmcnt — and accumulators spilled)
            const unsigned soff = (unsigned)((((int64_t)ky * W_) * Cin + ch * 32) * 2);
            const unsigned vo = ((sl_mask >> (3 * j + ky)) & 1u) ? sl_off[j] : 0xffffffffu;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(SLAB0 + b * SLAB_BYTES + (wave + 8 * j) * 1024));
            if (j == 4) dma16_buf_to_lds_if(wave0_exec, vo, srd_a, soff, dst);      // piece 32: wave 0's
            else dma16_buf_to_lds(vo, srd_a, soff, dst);
        } else {
            const int kx = j - 5;
            const unsigned soff_w = (unsigned)((((ky * 3 + kx) * Cin) + ch * 32) * 2);
            dma16_buf_to_lds(w_off, srd_w, soff_w, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(WT0 + b * WT_BYTES + kx * (BN_ * ROWB) + wave * 1024)));
        }
    };

    // ---- fragment addressing.  A fragment i of this wave = tile pixels wr * 128 + 16 i + frow = slab row mi + 2 (mi / seg) + frow + kx
    //      (two halo pixels per row segment before it); a wave's 128 pixels lie in one segment (seg >= 128): one base per tap, 16 i rows
    //      further = 1024 i bytes and the same chunk key.  W fragments: tile rows wc * 64 + 16 j + frow of the tap's 8-KiB tile ----
    const int frow = lane & 15, fk = lane >> 4;
    int a_off[3];
    {
        const int mi = wr * 128;
        const int base = mi + 2 * (mi / seg) + frow;             // (flat form: seg = 512, no halo rows between segments)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) a_off[kx] = SLAB0 + (base + kx) * ROWB + ((fk ^ glds_swz<32>(base + kx)) << 4);
    }
    // flat form: which of the wave's 128 pixels sit in column 0 / W - 1 (bit 16 i + frow of a 128-bit map; W >= 16: launcher)
    unsigned edge0[4] = {0u, 0u, 0u, 0u}, edge2[4] = {0u, 0u, 0u, 0u};
    if constexpr (FLAT) {
        const unsigned tp = (unsigned)m0 + (unsigned)(wr * 128 + frow);
        int x = (int)(tp - uc_div(tp, p.dWo) * (unsigned)W_);
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            edge0[i >> 1] |= ((unsigned)__ballot(x == 0) & 0xffffu) << (16 * (i & 1));
            edge2[i >> 1] |= ((unsigned)__ballot(x == W_ - 1) & 0xffffu) << (16 * (i & 1));
            x += 16;
            if (x >= W_) x -= W_;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            edge0[d] = (unsigned)__builtin_amdgcn_readfirstlane((int)edge0[d]);
            edge2[d] = (unsigned)__builtin_amdgcn_readfirstlane((int)edge2[d]);
        }
    }
    const int w_lane = WT0 + (wc * 64 + frow) * ROWB + ((fk ^ glds_swz<32>(frow)) << 4);

    float4_t acc[FA][4];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    bf16x8_t a[FA], w[4], wn[4];
    auto rd_a = [&](int b, int kx, int i) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8_t*>(smem + a_off[kx] + (b * SLAB_BYTES + i * 16 * ROWB));
    };
    auto rd_w = [&](int b, int kx, int j) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8_t*>(smem + w_lane + (b * WT_BYTES + kx * (BN_ * ROWB) + j * 16 * ROWB));
    };
    auto mma_row = [&](int i, bf16x8_t (&wv)[4], auto ckx) __attribute__((always_inline)) {      // ckx: the tap column these MFMAs belong to
        bf16x8_t av = a[i];
        if constexpr (RELU_A) av = __builtin_bit_cast(bf16x8_t, glds_relu_bf16x8(__builtin_bit_cast(uint4, av)));
        if constexpr (FLAT && decltype(ckx)::value == 0) av = (i & 1) ? glds_zero_edge_lanes<1>(av, edge0[i >> 1]) : glds_zero_edge_lanes<0>(av, edge0[i >> 1]);
        if constexpr (FLAT && decltype(ckx)::value == 2) av = (i & 1) ? glds_zero_edge_lanes<1>(av, edge2[i >> 1]) : glds_zero_edge_lanes<0>(av, edge2[i >> 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = glds_mfma<F16>(wv[j], av, acc[i][j]);
    };
    constexpr std::integral_constant<int, 0> KX0{};
    constexpr std::integral_constant<int, 1> KX1{};
    constexpr std::integral_constant<int, 2> KX2{};
    // one unit: MFMAs on (a, wcur) while the next unit's fragments stream into (a, wnext); the scheduling fences keep every refill of
    // a[i] behind the MFMAs that read the old a[i] (hoisted, both generations are live and the loop spills)
    auto unit = [&](bf16x8_t (&wcur)[4], bf16x8_t (&wnext)[4], int nb, int nkx, auto dma) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            if (nkx == 1) mma_row(i, wcur, KX0); else mma_row(i, wcur, KX1);      // (nkx = the tap column being LOADED: the MFMAs are one behind)
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wnext[j] = rd_w(nb, nkx, j);
            }
            a[i] = rd_a(nb, nkx, i);
            dma(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto no_dma = [](int) __attribute__((always_inline)) {};
    auto sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // prologue: super-steps 0 and 1 -> buffers 0 and 1 back to back (both are free; loads return in order: this wave's pieces of
    // super-step 0 have landed, the second batch keeps flying under units 0 and 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_piece(0, 0, 0, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_piece(0, 1, 1, j);          // (nch >= 2: launcher)
    wait_vmcnt<7>();      // (waves 1-7 have 7 + 7 pieces in flight, wave 0 8 + 8: at most 7 outstanding = the first batch has landed)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = rd_w(0, 0, j);
#pragma unroll
    for (int i = 0; i < FA; ++i) a[i] = rd_a(0, 0, i);
    unit(w, wn, 0, 1, no_dma);
    unit(wn, w, 0, 2, no_dma);
    // The loop is rotated so that its header sits at the synchronisation point (hipcc drains lgkmcnt at a loop header whatever is
    // pending, and there the drain is wanted).
    int ky2 = 0, ch2 = 2;                                         // (ky, chunk) of super-step s + 2, carried along
    if (ch2 >= nch) { ch2 = 0; ky2 = 1; }
    auto advance = [&]() __attribute__((always_inline)) { if (++ch2 == nch) { ch2 = 0; ++ky2; } };
    {
        // The synchronisation sits in the MIDDLE of unit 2: its first four MFMA groups (256 matrix-pipe cycles per wave, operands long in
        // registers) are queued when a wave reaches the s_waitcnt + barrier, so the pipe works through the barrier's skew instead of
        // draining behind the last fragment read; behind it the second half reads the next super-step's first fragments from the other
        // buffers; the pieces of super-step s + 2 follow one behind every other MFMA group: slots 0-1 in the rest of unit 2, 2-5 in the next
        // unit 0, 6-7 in the first half of unit 1 (>= 1.5 units of flight to the next synchronisation each).
        auto unit2a = [&](bf16x8_t (&wcur)[4]) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) mma_row(i, wcur, KX2);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto unit2b = [&](bf16x8_t (&wcur)[4], bf16x8_t (&wnext)[4], int nb, auto dma) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) wnext[j] = rd_w(nb, 0, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = rd_a(nb, 0, i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 4; i < FA; ++i) {
                mma_row(i, wcur, KX2);
                __builtin_amdgcn_sched_barrier(0);
                a[i] = rd_a(nb, 0, i);
                dma(i - 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // DMA schedule of super-step s + 2 (8 slots per wave), one piece behind every other MFMA group from the synchronisation on:
        // slots 0-1 in the rest of unit 2, 2-5 in unit 0, 6-7 in the first half of unit 1 — the global -> LDS path takes ~40 cycles
        // per piece and CU (tools/probes/dma_seg.hip): 57 pieces are 2300 of a super-step's 3072 matrix-pipe cycles, so they are spread
        // over as much of it as the buffers' release (the synchronisation) and the pieces' flight (>= 1.5 units) allow
        unit2a(w);
        for (int s = 0; s + 2 < S; s += 2) {
            sync();
            unit2b(w, wn, 1, [&](int g) __attribute__((always_inline)) { if (g & 1) issue_piece(ky2, ch2, 0, g >> 1); });                      // rest of unit 2 of s
            unit(wn, w, 1, 1, [&](int i) __attribute__((always_inline)) { if (i & 1) issue_piece(ky2, ch2, 0, 2 + (i >> 1)); });             // unit 0 of s + 1
            unit(w, wn, 1, 2, [&](int i) __attribute__((always_inline)) { if ((i & 1) && i < 4) issue_piece(ky2, ch2, 0, 6 + (i >> 1)); });  // unit 1
            advance();
            unit2a(wn);
            sync();
            unit2b(wn, w, 0, [&](int g) __attribute__((always_inline)) { if (g & 1) issue_piece(ky2, ch2, 1, g >> 1); });
            unit(w, wn, 0, 1, [&](int i) __attribute__((always_inline)) { if (i & 1) issue_piece(ky2, ch2, 1, 2 + (i >> 1)); });
            unit(wn, w, 0, 2, [&](int i) __attribute__((always_inline)) { if ((i & 1) && i < 4) issue_piece(ky2, ch2, 1, 6 + (i >> 1)); });
            advance();
            unit2a(w);
        }
        sync();
        unit2b(w, wn, 1, no_dma);
    }
    unit(wn, w, 1, 1, no_dma);
    unit(w, wn, 1, 2, no_dma);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FA; ++i) mma_row(i, wn, KX2);

#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((opencl_constant)) GldsParams* kp =
        (const __attribute__((opencl_constant)) GldsParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp)::"memory");
    glds_pe_t pe = *kp;
#else
    glds_pe_t pe = p;
#endif
    __syncthreads();     // every wave is done with the buffers: they become the epilogue's bounce space (8 KiB per wave)
    glds_epilogue_dispatch<FA, A_MODE, EPI, F16>(pe, acc, 0, wave_m, wave_n, tid, wave, 0, smem);
}

// shapes the eight-wave row-walking kernel takes: stride 1, whole 64-channel chunks (an even number of 32-channel super-steps), 128-column
// tiles, maps 128 .. wide whose rows tile 512 pixels exactly (a tile = whole row segments of one image), 32-bit source windows
// returns 0: not a shape of this kernel; 1: the segmented form (rows tile 512 pixels: no register masks); 2: the flat form (round 6: any
// map at least 16 wide, any pixel count — tiles of 512 consecutive pixels, the last one masked)
static inline int conv_rows8_ok(const GldsParams& p) {
    if (p.a_mode != UC_A_CONV3X3 || p.cStride != 1 || p.cCin % 64 != 0 || p.N % 128 != 0 || p.split_k > 1) return 0;
    const int W = p.cW, H = p.cH;
    if (p.N * p.K * 2 >= ((int64_t)1 << 31) || p.M >= ((int64_t)1 << 30)) return 0;
    if (W >= 128 && (W % 512 == 0 || 512 % W == 0)) {
        const int R = W >= 512 ? 1 : 512 / W;
        if (H % R == 0 && p.M % 512 == 0 && ((int64_t)(R + 3) * W + 4) * p.cCin * 2 < ((int64_t)1 << 31)) return 1;
    }
    if (W < 16 || g_uc_conv_rows_flat.load(std::memory_order_relaxed) == 0) return 0;
    return ((int64_t)(530 + 2 * W) * p.cCin * 2 < ((int64_t)1 << 31)) ? 2 : 0;
}

// where the eight-wave form is routed by default: everywhere its shape rules allow — measured ahead of both other forms on every
// DPT-head shape, bf16 and fp16, with and without ReLU on load (512^2 128->128 +16-19 %, 256^2 256->128 +24 %
// over the 256-pixel row kernel, 256 output channels +8-12 % over the 256x256 implicit-GEMM tile)
static inline bool conv_rows8_wins(const GldsParams&) { return true; }

template <int EPI, bool F16>
static void launch_conv_rows8(GldsParams p, hipStream_t st) {
    const bool flat = conv_rows8_ok(p) == 2;
    p.tiles_m = (int)((p.M + 511) / 512);
    p.tiles_n = (int)(p.N / 128);
    p.dNwg = uc_make_fastdiv((unsigned)(p.tiles_m * p.tiles_n));
    p.dPerGroup = uc_make_fastdiv((unsigned)(p.group_m * p.tiles_n));
    p.dGm = uc_make_fastdiv((unsigned)p.group_m);
    p.dGmLast = uc_make_fastdiv((unsigned)std::max(1, p.tiles_m % p.group_m));
    constexpr int smem = 2 * 33 * 1024 + 2 * 3 * 128 * 64;
    auto launch = [&](auto kfn, bool& attr_set) {
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            attr_set = true;
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)p.tiles_m * p.tiles_n), dim3(512), smem, st, p);
    };
    static bool set0 = false, set1 = false, set2 = false, set3 = false;
    if constexpr (EPI == GLDS_EPI_RES16) {      // (glds_launch_conv_res16: no ReLU on load)
        if (flat) launch(conv3x3_rows8_kernel<EPI, F16, false, true>, set2);
        else launch(conv3x3_rows8_kernel<EPI, F16, false>, set0);
    } else if (flat) {
        if (p.relu_a) launch(conv3x3_rows8_kernel<EPI, F16, true, true>, set3);
        else launch(conv3x3_rows8_kernel<EPI, F16, false, true>, set2);
    } else if (p.relu_a) launch(conv3x3_rows8_kernel<EPI, F16, true>, set1);
    else launch(conv3x3_rows8_kernel<EPI, F16, false>, set0);
}

// A convolution whose every tile takes the 16-bit residual epilogue (glds_epilogue_res16): 16-bit output, one or two residuals of the
// same dtype laid out like it, bias or none, no activation — the residual conv units' second convolution.  Its own epilogue family
// (one kernel per family: inlined next to the other drains it cost the fused-tail launches 5 % through the register allocation).
static inline bool glds_res16_ok(const GldsParams& p) {
    const int out16 = p.f16 ? UC_F16 : UC_BF16;
    return p.a_mode == UC_A_CONV3X3 && p.vec_ok && p.N % 64 == 0 && p.split_k <= 1 && !p.preact && !p.dact_u && !UC_DBG(p, 16) && p.out_dtype == out16 &&
           p.residual && p.res_dtype == out16 && p.act == UC_ACT_NONE && !p.relu_a && !p.tail_out && !p.ln_stats && !p.ln_partial && !p.stats_out && !p.twin &&
           (p.ldr & 7) == 0 && p.vt_col0 < 0 && p.rope_cols <= 0;
}
// the eight-wave row kernel where the default routing takes it, else the 256x256 implicit-GEMM tile; false: not a shape of this family's kernels
template <bool F16>
static bool glds_launch_conv_res16(const GldsParams& p, int variant, hipStream_t st) {
    if (!glds_res16_ok(p)) return false;
    const int rows_mode = g_uc_conv_rows.load(std::memory_order_relaxed);
    if (conv_rows8_ok(p) && (rows_mode == 3 || (rows_mode == 1 && conv_rows8_wins(p) && ((p.M + 511) / 512) * (p.N / 128) >= 256))) {
        launch_conv_rows8<GLDS_EPI_RES16, F16>(p, st);
        return true;
    }
    // (not the launches the 256-pixel row kernel takes: glds_launch_variants' rule)
    const bool rows256 = rows_mode > 0 && rows_mode < 3 && conv_rows_ok(p) && (rows_mode >= 2 || (p.N == 128 && p.cCin >= 256)) && (p.M / 256) * (p.N / 128) >= 256;
    if (variant == 2 && !rows256) {
        launch_variant_mode<256, 256, 4, 4, 2, UC_A_CONV3X3, 64, 1, GLDS_EPI_RES16, F16>(p, st);
        return true;
    }
    return false;
}

// Tile variants of one (A_MODE, EPI) pair: 0 = 128x128 (2x2 waves of 64x64), 1 = 256x128 (4x2), 2 = 256x256 (4x4), 3 = 256x128x32 with
// two co-resident workgroups per CU.
template <int A_MODE, int EPI, bool F16 = false>
static void glds_launch_variants(const GldsParams& p, int variant, hipStream_t st) {
    const int deep = uc_knobs().gemm_small_stages;
    const int64_t sk = p.split_k > 1 ? p.split_k : 1;
    if constexpr (A_MODE == UC_A_CONV3X3) {
        // Where it wins (conv_rows 1): 128 output channels and >= 256 input channels — 256^2 256 -> 128: 850 -> 959 TFLOP/s (fp16 834 -> 926).
        // With 128 input channels it is level (873 -> 891, fp16 875 -> 864), with 256 output channels the 256x256 tile of the
        // implicit-GEMM kernel streams half the weights per MFMA and stays ahead (1016 / 1106 vs 1000 / 1003 at 128^2 / 64^2): the weight
        // tiles, which this form does not reduce, are what a conv tile's LDS-DMA traffic mostly is.  Fewer tiles than CUs: the
        // latency-regime variants below.
        const int rows_mode = g_uc_conv_rows.load(std::memory_order_relaxed);
        // conv_rows 3: the eight-wave 512-pixel form wherever the shape allows; 1 (default): where it wins
        // (fewer tiles than CUs: the latency-regime variants below — unless forced: conv_rows 3 makes the kernel choice, and with it the
        //  summation order, independent of the batch size)
        if (conv_rows8_ok(p) && (rows_mode == 3 || (rows_mode == 1 && conv_rows8_wins(p) && ((p.M + 511) / 512) * (p.N / 128) >= 256))) {
            launch_conv_rows8<EPI, F16>(p, st);
            return;
        }
        if (rows_mode > 0 && rows_mode < 3 && conv_rows_ok(p) && (rows_mode >= 2 || (p.N == 128 && p.cCin >= 256)) && (p.M / 256) * (p.N / 128) >= 256) {
            launch_conv_rows<EPI, F16>(p, st);
            return;
        }
    }
    switch (variant) {
        case 1:
            // latency regime (fewer workgroups than CUs: every K-step waits for its own DMA): a 3-stage ring keeps two stages in flight
            if (deep == 3 && ceil_div64(p.M, 256) * ceil_div64(p.N, 128) * sk <= 256) launch_variant_mode<256, 128, 4, 2, 3, A_MODE, 64, 1, EPI, F16>(p, st);
            else launch_variant_mode<256, 128, 4, 2, 2, A_MODE, 64, 1, EPI, F16>(p, st);
            break;
        case 2: launch_variant_mode<256, 256, 4, 4, 2, A_MODE, 64, 1, EPI, F16>(p, st); break;
        case 3: launch_variant_mode<256, 128, 4, 2, 3, A_MODE, 32, 2, EPI, F16>(p, st); break;
        case 6:   // 256x256x64 with eight waves of 128x64 and register-resident next-chunk fragments (dense only)
            // (its DMA addresses row groups of 8 uniformly: matrices whose last group is partial stay on the 16-wave kernel)
            if (!F16 && A_MODE == UC_A_DENSE && p.M % 8 == 0 && p.N % 8 == 0) {
                if constexpr (A_MODE == UC_A_DENSE && !F16) launch_glds8<EPI>(p, st);
            } else launch_variant_mode<256, 256, 4, 4, 2, A_MODE, 64, 1, EPI, F16>(p, st);
            break;
        case 7:   // 256x256x64 with four waves of 128x128, accumulators in AGPRs, hand-scheduled K-loop (dense only)
            if (!F16 && glds4_ok(p)) {
                if constexpr (A_MODE == UC_A_DENSE && !F16) launch_glds4<EPI>(p, st);
            } else launch_variant_mode<256, 256, 4, 4, 2, A_MODE, 64, 1, EPI, F16>(p, st);
            break;
        case 4:   // 128x64 (two waves): the latency regime's small tile — launches whose 128x128 tiles cover at most half the CUs
            if constexpr (A_MODE == UC_A_DENSE && !F16) { launch_variant_mode<128, 64, 2, 1, 3, A_MODE, 64, 1, EPI, F16>(p, st); break; }
            [[fallthrough]];
        default:
            if (deep == 3 && ceil_div64(p.M, 128) * ceil_div64(p.N, 128) * sk <= 512) launch_variant_mode<128, 128, 2, 2, 3, A_MODE, 64, 1, EPI, F16>(p, st);
            else launch_variant_mode<128, 128, 2, 2, 2, A_MODE, 64, 1, EPI, F16>(p, st);
            break;
    }
}
