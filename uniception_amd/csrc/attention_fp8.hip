// FP8 (OCP e4m3) flash attention forward for gfx950 — BASELINE config 5's "fp8 MFMA attention path".
//
// Same structure as attn_bf16_kernel (4 waves x 32 queries, 64-key tiles, swapped product S^T = K Q^T so a query's scores
// live in one lane pair, online softmax with deferred rescale, O^T += V^T P^T), but both products run on the block-scaled
// K=64 instruction  v_mfma_scale_f32_32x32x64_f8f6f4  with unit scales (the only fp8 MFMA that runs at twice the bf16 rate
// on gfx950): ONE MFMA per 32x32 score block (head_dim 64 is a single K step) and one per 32-channel output block per
// 64-key tile — 4 matrix instructions per tile instead of 16.
//   * Q and K arrive in bf16 (RoPE already applied by the QKV GEMM); Q is converted to e4m3 once per workgroup, K while it is
//     staged into LDS (v_cvt_pk_fp8_f32), so HBM traffic equals the bf16 kernel's.
//   * P is converted with a x64 offset (P <= 2^2 under the rescale threshold -> <= 256 < 448 = e4m3 max; the offset keeps
//     small probabilities out of the subnormal range) and divided out in the final normalisation.
//   * V comes pre-packed by uc_vt_pack_fp8: transposed per head, e4m3, keys permuted inside every 64-key group so that the
//     accumulator register order of P^T IS the MFMA k-slot order: position 32*hh + 16*kb + r holds key
//     32*kb + (r&3) + 8*(r>>2) + 4*hh.  Pad positions are zero, so tail tiles need no masking of V.
// Operand layout of the instruction (probed, tools/probes/f8_layout.hip): byte e of lane l is A[row = l&31][k] resp.
// B[k][col = l&31] with ANY k assignment shared by A and B across the two lane halves; C/D as every 32x32 MFMA.
#include "common.h"

typedef int v8i_t __attribute__((ext_vector_type(8)));

struct AttnF8Params {
    const bf16_t* Q;
    const bf16_t* K;
    const unsigned char* VT8;
    bf16_t* O;
    int B, H, Nq, Nk, npad;
    int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, o_sb, o_sn, o_sh;
    float scale;
};

#define F8_TILE (64 * 64)          // bytes of one 64-row fp8 tile
#define F8_ONE 0x7f7f7f7f          // E8M0 scale bytes = 2^0
#define F8_POFF 64.0f              // offset applied to P before the e4m3 conversion
#define F8_RESCALE_LOG2 2.0f       // deferred-rescale threshold: P <= 2^2

__device__ __forceinline__ int f8_swz(int row, int chunk16) { return row * 64 + ((chunk16 ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ unsigned cvt4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// 8 bf16 (one 16-byte chunk) -> 8 e4m3 bytes
__device__ __forceinline__ uint2 bf16x8_to_fp8x8(uint4 v) {
    const unsigned* u = reinterpret_cast<const unsigned*>(&v);
    uint2 r;
    r.x = cvt4_fp8(__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u));
    r.y = cvt4_fp8(__uint_as_float(u[2] << 16), __uint_as_float(u[2] & 0xffff0000u), __uint_as_float(u[3] << 16), __uint_as_float(u[3] & 0xffff0000u));
    return r;
}

__global__ __launch_bounds__(256) void attn_fp8_kernel(AttnF8Params p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * F8_TILE];   // 2 stages x (K8 tile + VT8 tile)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    // 1-D grid; query tiles of one (batch, head) get ids that differ by multiples of 8 = one XCD's L2 (see attn_bf16_dma_kernel)
    const int nq = (p.Nq + 127) / 128, nbh = p.B * p.H;
    int qt, bh;
    {
        const int w = blockIdx.x;
        const int per_group = 8 * nq;
        const int grp = w / per_group, within = w - grp * per_group;
        if ((grp + 1) * 8 <= nbh) { bh = grp * 8 + (within & 7); qt = within >> 3; }
        else { const int rem = w - (nbh / 8) * 8 * nq; bh = (nbh / 8) * 8 + rem / nq; qt = rem % nq; }
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * 128 + wave * 32;

    const bf16_t* Qb = p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* Kb = p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const unsigned char* VTb = p.VT8 + ((int64_t)b * p.H + h) * 64 * (int64_t)p.npad;

    // ---- Q^T operand: lane (q = l31, half hi) holds channels 32*hi .. +31 as 32 e4m3 bytes ----
    v8i_t qf;
    {
        int q = q0 + l31;
        if (q >= p.Nq) q = p.Nq - 1;
        const bf16_t* qp = Qb + (int64_t)q * p.q_sn + hi * 32;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const uint2 f = bf16x8_to_fp8x8(*reinterpret_cast<const uint4*>(qp + 8 * c4));
            qf[2 * c4] = (int)f.x;
            qf[2 * c4 + 1] = (int)f.y;
        }
    }

    // ---- staging: K rows sr, sr+32 (chunk cc of 8 channels, converted to 8 bytes); VT8 row tid>>2, 16-byte chunk tid&3 ----
    const int cc = tid & 7, sr = tid >> 3;
    const int vrow = tid >> 2, vch = tid & 3;
    uint4 rk0, rk1, rv;
    const int kw_off0 = f8_swz(sr, cc >> 1) + ((cc & 1) << 3);
    const int kw_off1 = f8_swz(sr + 32, cc >> 1) + ((cc & 1) << 3);
    const int vw_off = f8_swz(vrow, vch);
#define F8_STAGE_LOAD(k0_)                                                                                  \
    do {                                                                                                    \
        int ka_ = (k0_) + sr, kb_ = (k0_) + sr + 32;                                                        \
        if (ka_ >= p.Nk) ka_ = p.Nk - 1;                                                                    \
        if (kb_ >= p.Nk) kb_ = p.Nk - 1;                                                                    \
        rk0 = *reinterpret_cast<const uint4*>(Kb + (int64_t)ka_ * p.k_sn + cc * 8);                        \
        rk1 = *reinterpret_cast<const uint4*>(Kb + (int64_t)kb_ * p.k_sn + cc * 8);                        \
        rv = *reinterpret_cast<const uint4*>(VTb + (int64_t)vrow * p.npad + (k0_) + vch * 16);              \
    } while (0)
#define F8_STAGE_WRITE(buf_)                                                                                \
    do {                                                                                                    \
        char* sk_ = smem + (buf_) * 2 * F8_TILE;                                                            \
        *reinterpret_cast<uint2*>(sk_ + kw_off0) = bf16x8_to_fp8x8(rk0);                                    \
        *reinterpret_cast<uint2*>(sk_ + kw_off1) = bf16x8_to_fp8x8(rk1);                                    \
        *reinterpret_cast<uint4*>(sk_ + F8_TILE + vw_off) = rv;                                             \
    } while (0)

    // fragment reads: row l31 (+32 for the second block), bytes 32*hi .. +31 = 16-byte chunks 2hi, 2hi+1
    const int r_off0 = f8_swz(l31, 2 * hi), r_off1 = f8_swz(l31, 2 * hi + 1);   // rows r and r+32 share the swizzle key
    auto frag = [&](const char* tile, int blk) -> v8i_t {
        const uint4 a = *reinterpret_cast<const uint4*>(tile + r_off0 + blk * (32 * 64));
        const uint4 c = *reinterpret_cast<const uint4*>(tile + r_off1 + blk * (32 * 64));
        v8i_t f;
        f[0] = (int)a.x; f[1] = (int)a.y; f[2] = (int)a.z; f[3] = (int)a.w;
        f[4] = (int)c.x; f[5] = (int)c.y; f[6] = (int)c.z; f[7] = (int)c.w;
        return f;
    };

    float16_t o[2];
    o[0] = (float16_t)(0.f);
    o[1] = (float16_t)(0.f);
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale * 1.44269504088896340736f;

    const int nt = (p.Nk + 63) / 64;
    F8_STAGE_LOAD(0);
    F8_STAGE_WRITE(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int k0 = t * 64;
        if (t + 1 < nt) F8_STAGE_LOAD(k0 + 64);
        const char* sk = smem + buf * 2 * F8_TILE;
        const char* sv = sk + F8_TILE;

        float16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = (float16_t)(0.f);
            s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(sk, kb), qf, s[kb], 0, 0, 0, F8_ONE, 0, F8_ONE);
        }
        if (k0 + 64 > p.Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Nk) s[kb][r] = -1e30f;
                }
        }
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const bool grow = (mt - m_run) * c > F8_RESCALE_LOG2;
        if (__any(grow)) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        v8i_t pf;   // P^T operand: byte 16*kb + r of lane (q, hi) = key 32*kb + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(fmaf(s[kb][4 * w + j], c, -mc));
                    psum += e[j];
                }
                pf[4 * kb + w] = (int)cvt4_fp8(e[0] * F8_POFF, e[1] * F8_POFF, e[2] * F8_POFF, e[3] * F8_POFF);
            }
        l_run += psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(sv, db), pf, o[db], 0, 0, 0, F8_ONE, 0, F8_ONE);

        if (t + 1 < nt) F8_STAGE_WRITE(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / (l_tot * F8_POFF);
    const int q = q0 + l31;
    if (q < p.Nq) {
        bf16_t* op = p.O + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * hi;
                uint2 pk;
                pk.x = pack_bf16x2(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv);
                pk.y = pack_bf16x2(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + d) = pk;
            }
    }
}

// ---------------------------------------------------------------------------------------
// attn_fp8_dma_kernel — the same products with BOTH tiles pre-packed in e4m3 (K8 from uc_k_pack_fp8, VT8 from
// uc_vt_pack_fp8) and staged by buffer-addressed LDS-DMA: no conversion or staging registers inside the key loop (the
// kernel above converts every K tile once per query tile, 8x redundantly at N = 1024), Nk % 64 == 0, outputs through an
// LDS bounce into whole-row stores, XCD-aware workgroup order.
// ---------------------------------------------------------------------------------------
typedef unsigned f8_uint4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void f8_dma16(unsigned voff, f8_uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
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
__device__ __forceinline__ f8_uint4_t f8_make_srd(const void* base) {
    const unsigned long long pa = (unsigned long long)base;
    return (f8_uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
}

struct AttnF8DmaParams {
    const bf16_t* Q;
    const unsigned char* K8;
    const unsigned char* VT8;
    bf16_t* O;
    int B, H, Nq, Nk, npad;
    int64_t q_sb, q_sn, q_sh, o_sb, o_sn, o_sh;
    float scale;
};

__global__ __launch_bounds__(256, 4) void attn_fp8_dma_kernel(AttnF8DmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * F8_TILE];   // 2 stages x (K8 tile + VT8 tile); later 4 KiB of output bounce per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int nq = (p.Nq + 127) / 128, nbh = p.B * p.H;
    int qt, bh;
    {
        const int w = blockIdx.x;
        const int per_group = 8 * nq;
        const int grp = w / per_group, within = w - grp * per_group;
        if ((grp + 1) * 8 <= nbh) { bh = grp * 8 + (within & 7); qt = within >> 3; }
        else { const int rem = w - (nbh / 8) * 8 * nq; bh = (nbh / 8) * 8 + rem / nq; qt = rem % nq; }
    }
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * 128 + wave * 32;
    const bf16_t* Qb = p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;

    // ---- DMA assignment: a 64x64-byte tile is 4 instructions of 16 rows; waves 0,1 carry the K8 tile, waves 2,3 the VT8 tile ----
    const bool is_v = wave >= 2;
    const f8_uint4_t srd = is_v ? f8_make_srd(p.VT8 + (int64_t)bh * 64 * (int64_t)p.npad) : f8_make_srd(p.K8 + (int64_t)bh * (int64_t)p.npad * 64);
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = ((wave & 1) * 2 + i) * 16 + (lane >> 2);
        const int cch = (lane & 3) ^ ((row >> 2) & 3);          // logical 16-byte chunk stored at physical chunk lane&3 of row `row`
        voff[i] = is_v ? (unsigned)(row * p.npad + cch * 16) : (unsigned)(row * 64 + cch * 16);
    }
    const unsigned tstep = is_v ? 64u : 4096u;                    // bytes between key tiles
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    auto issue_tile = [&](int t, int buf) {
        const unsigned dst = lds0 + (unsigned)(buf * 2 * F8_TILE + (is_v ? F8_TILE : 0) + (wave & 1) * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) f8_dma16(voff[i], srd, (unsigned)t * tstep, __builtin_amdgcn_readfirstlane(dst + i * 1024));
    };
    issue_tile(0, 0);

    // ---- Q^T operand: lane (q = l31, half hi) holds channels 32*hi .. +31 as 32 e4m3 bytes ----
    v8i_t qf;
    {
        int q = q0 + l31;
        if (q >= p.Nq) q = p.Nq - 1;
        const bf16_t* qp = Qb + (int64_t)q * p.q_sn + hi * 32;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const uint2 f = bf16x8_to_fp8x8(*reinterpret_cast<const uint4*>(qp + 8 * c4));
            qf[2 * c4] = (int)f.x;
            qf[2 * c4 + 1] = (int)f.y;
        }
    }
    const int r_off0 = f8_swz(l31, 2 * hi), r_off1 = f8_swz(l31, 2 * hi + 1);
    auto frag = [&](const char* tile, int blk) -> v8i_t {
        const uint4 a = *reinterpret_cast<const uint4*>(tile + r_off0 + blk * (32 * 64));
        const uint4 c = *reinterpret_cast<const uint4*>(tile + r_off1 + blk * (32 * 64));
        v8i_t f;
        f[0] = (int)a.x; f[1] = (int)a.y; f[2] = (int)a.z; f[3] = (int)a.w;
        f[4] = (int)c.x; f[5] = (int)c.y; f[6] = (int)c.z; f[7] = (int)c.w;
        return f;
    };

    float16_t o[2];
    o[0] = (float16_t)(0.f);
    o[1] = (float16_t)(0.f);
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale * 1.44269504088896340736f;
    const int nt = p.Nk / 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
        const char* sk = smem + buf * 2 * F8_TILE;
        const char* sv = sk + F8_TILE;
        float16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = (float16_t)(0.f);
            s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(sk, kb), qf, s[kb], 0, 0, 0, F8_ONE, 0, F8_ONE);
        }
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const bool grow = (mt - m_run) * c > F8_RESCALE_LOG2;
        if (__any(grow)) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        const float mc = m_run * c;
        float psum = 0.f;
        v8i_t pf;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(fmaf(s[kb][4 * w + j], c, -mc));
                    psum += e[j];
                }
                pf[4 * kb + w] = (int)cvt4_fp8(e[0] * F8_POFF, e[1] * F8_POFF, e[2] * F8_POFF, e[3] * F8_POFF);
            }
        l_run += psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(sv, db), pf, o[db], 0, 0, 0, F8_ONE, 0, F8_ONE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l_tot * F8_POFF);
    char* ob = smem + wave * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint2 pk;
            pk.x = pack_bf16x2(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv);
            pk.y = pack_bf16x2(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(ob + l31 * 128 + (((4 * db + g4) ^ (l31 & 7)) << 4) + hi * 8) = pk;
        }
    bf16_t* obase = p.O + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int R = 8 * ps + (lane >> 3);
        const int chunk = (lane & 7) ^ (R & 7);
        const uint4 v = *reinterpret_cast<const uint4*>(ob + R * 128 + ((lane & 7) << 4));
        if (q0 + R < p.Nq) *reinterpret_cast<uint4*>(obase + (int64_t)(q0 + R) * p.o_sn + chunk * 8) = v;
    }
}

extern "C" int uc_attention_fwd_fp8_k8(const void* Q, const void* K8, const void* VT8, void* O, int B, int H, int Nq, int Nk,
                                       int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh,
                                       float scale, uc_stream_t stream) {
    UC_REQUIRE(Q && K8 && VT8 && O, "uc_attention_fwd_fp8_k8: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "uc_attention_fwd_fp8_k8: bad shape");
    UC_REQUIRE(Nk % 64 == 0, "uc_attention_fwd_fp8_k8: Nk must be a multiple of 64 (use uc_attention_fwd_fp8 for ragged key counts)");
    UC_REQUIRE((int64_t)((Nq + 127) / 128) * H * B < ((int64_t)1 << 31) && (int64_t)64 * Nk < ((int64_t)1 << 31), "uc_attention_fwd_fp8_k8: problem too large");
    UC_REQUIRE(q_sn % 8 == 0 && q_sh % 8 == 0 && q_sb % 8 == 0, "uc_attention_fwd_fp8_k8: Q strides must be multiples of 8 elements");
    UC_REQUIRE(o_sn % 8 == 0 && o_sh % 8 == 0 && o_sb % 8 == 0 && (uintptr_t)O % 16 == 0 && (uintptr_t)Q % 16 == 0 &&
                   (uintptr_t)K8 % 16 == 0 && (uintptr_t)VT8 % 16 == 0, "uc_attention_fwd_fp8_k8: O strides must be multiples of 8 elements, pointers 16-byte aligned");
    AttnF8DmaParams p;
    p.Q = (const bf16_t*)Q; p.K8 = (const unsigned char*)K8; p.VT8 = (const unsigned char*)VT8; p.O = (bf16_t*)O;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.npad = Nk;
    p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh; p.scale = scale;
    hipLaunchKernelGGL(attn_fp8_dma_kernel, dim3((unsigned)(((Nq + 127) / 128) * H * B)), dim3(256), 0, (hipStream_t)stream, p);
    UC_CHECK_LAUNCH("uc_attention_fwd_fp8_k8");
    return UC_OK;
}

// row-major bf16 K (strided view) -> e4m3 rows [B,H,Npad,64], zero padded
__global__ __launch_bounds__(256) void k_pack_fp8_kernel(const bf16_t* __restrict__ K, unsigned char* __restrict__ K8, int H, int Nk, int npad,
                                                         int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t total_chunks) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 8-channel chunk per thread
    if (idx >= total_chunks) return;
    const int c8 = (int)(idx & 7);
    const int64_t row = idx >> 3;                                     // (b*H + h)*npad + n
    const int n = (int)(row % npad);
    const int64_t bhh = row / npad;
    const int h = (int)(bhh % H);
    const int64_t b = bhh / H;
    uint2 o = make_uint2(0u, 0u);
    if (n < Nk) o = bf16x8_to_fp8x8(*reinterpret_cast<const uint4*>(K + b * k_sb + (int64_t)n * k_sn + (int64_t)h * k_sh + c8 * 8));
    *reinterpret_cast<uint2*>(K8 + row * 64 + c8 * 8) = o;
}

extern "C" int uc_k_pack_fp8(const void* K, void* K8, int B, int H, int Nk, int64_t k_sb, int64_t k_sn, int64_t k_sh, uc_stream_t stream) {
    UC_REQUIRE(K && K8, "uc_k_pack_fp8: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nk > 0, "uc_k_pack_fp8: bad shape");
    UC_REQUIRE(k_sn % 8 == 0 && k_sh % 8 == 0 && k_sb % 8 == 0 && (uintptr_t)K % 16 == 0 && (uintptr_t)K8 % 8 == 0,
               "uc_k_pack_fp8: K strides must be multiples of 8 elements");
    const int npad = (Nk + 63) / 64 * 64;
    const int64_t total = (int64_t)B * H * npad * 8;
    hipLaunchKernelGGL(k_pack_fp8_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)K,
                       (unsigned char*)K8, H, Nk, npad, k_sb, k_sn, k_sh, total);
    UC_CHECK_LAUNCH("uc_k_pack_fp8");
    return UC_OK;
}

extern "C" int uc_attention_fwd_fp8(const void* Q, const void* K, const void* VT8, void* O, int B, int H, int Nq, int Nk,
                                    int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                                    int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, uc_stream_t stream) {
    UC_REQUIRE(Q && K && VT8 && O, "uc_attention_fwd_fp8: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && B <= 65535 && H <= 65535, "uc_attention_fwd_fp8: bad shape");
    UC_REQUIRE(q_sn % 8 == 0 && k_sn % 8 == 0 && q_sh % 8 == 0 && k_sh % 8 == 0 && q_sb % 8 == 0 && k_sb % 8 == 0,
               "uc_attention_fwd_fp8: Q/K strides must be multiples of 8 elements");
    UC_REQUIRE(o_sn % 4 == 0 && o_sh % 4 == 0 && o_sb % 4 == 0, "uc_attention_fwd_fp8: O strides must be multiples of 4 elements");
    AttnF8Params p;
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.VT8 = (const unsigned char*)VT8; p.O = (bf16_t*)O;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.npad = (Nk + 63) / 64 * 64;
    p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
    p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh; p.scale = scale;
    hipLaunchKernelGGL(attn_fp8_kernel, dim3((unsigned)(((Nq + 127) / 128) * H * B)), dim3(256), 0, (hipStream_t)stream, p);
    UC_CHECK_LAUNCH("uc_attention_fwd_fp8");
    return UC_OK;
}

// ---------------------------------------------------------------------------------------
// row-major bf16 V -> e4m3 V^T in the k-slot order of the kernel above, zero padded
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vt_pack_fp8_kernel(const bf16_t* __restrict__ V, unsigned char* __restrict__ VT8, int H, int Nk,
                                                          int npad, int64_t v_sb, int64_t v_sn, int64_t v_sh) {
    __shared__ float tile[64][65];   // [key][d]
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
    const bf16_t* vb = V + (int64_t)b * v_sb + (int64_t)h * v_sh;
    for (int idx = threadIdx.x; idx < 64 * 8; idx += 256) {
        const int key = idx >> 3, c8 = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k0 + key < Nk) v = *reinterpret_cast<const uint4*>(vb + (int64_t)(k0 + key) * v_sn + c8 * 8);
        const unsigned* u = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[key][c8 * 8 + 2 * e] = __uint_as_float(u[e] << 16);
            tile[key][c8 * 8 + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
        }
    }
    __syncthreads();
    unsigned* out = reinterpret_cast<unsigned*>(VT8 + ((int64_t)b * H + h) * 64 * (int64_t)npad + k0);
    for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
        const int d = idx >> 4, w = idx & 15;          // dword w of row d: positions 4w .. 4w+3
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = 4 * w + j;
            const int hh = pos >> 5, kb = (pos >> 4) & 1, r = pos & 15;
            e[j] = tile[32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh][d];
        }
        out[((int64_t)d * npad) / 4 + w] = cvt4_fp8(e[0], e[1], e[2], e[3]);
    }
}

extern "C" int uc_vt_pack_fp8(const void* V, void* VT8, int B, int H, int Nk, int D, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                              uc_stream_t stream) {
    UC_REQUIRE(V && VT8, "uc_vt_pack_fp8: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nk > 0 && D == 64, "uc_vt_pack_fp8: head_dim must be 64");
    UC_REQUIRE(v_sn % 8 == 0 && v_sh % 8 == 0 && v_sb % 8 == 0, "uc_vt_pack_fp8: V strides must be multiples of 8 elements");
    const int npad = (Nk + 63) / 64 * 64;
    hipLaunchKernelGGL(vt_pack_fp8_kernel, dim3(npad / 64, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)V,
                       (unsigned char*)VT8, H, Nk, npad, v_sb, v_sn, v_sh);
    UC_CHECK_LAUNCH("uc_vt_pack_fp8");
    return UC_OK;
}
