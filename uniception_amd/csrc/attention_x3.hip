// fp32-class scaled-dot-product attention on the matrix pipe (head_dim 64): engine.precision("bf16x3").
//
// The verification-grade mode (north star: outputs within 1e-3 of the reference) used to run softmax(QK^T)V in the one-thread-per-
// query fp32 VALU kernel (attn_f32_kernel) — 7.8x slower than the bf16 forward as a whole.  Here both products run as THREE bf16
// MFMA products of split operands, x = hi + lo with hi = bf16(x), lo = bf16(x - hi):
//     S  = Kh.Qh + Kh.Ql + Kl.Qh            (the dropped Kl.Ql term is 2^-16 relative)
//     O += Vh.Ph + Vh.Pl + Vl.Ph            P = exp2(S - m) in fp32, split the same way
// fp32 accumulation, fp32 softmax statistics (row sums of the UNROUNDED P): ~1e-5 relative on the output, against 4e-3 for the
// bf16 kernel.  Same tiling as attn_bf16_dma_kernel (attention.hip): 4 waves x 32 queries, swapped product S^T = K.Q^T, 64-key
// tiles of Kh / Kl / VTh / VTl staged by buffer-addressed LDS-DMA in a two-stage ring, outputs bounced through LDS into whole-row
// stores.  The operands come from a split pass (uc_attention_fwd_x3 runs it into the caller's workspace): Q is multiplied by
// scale * log2(e) in fp32 BEFORE the split, so the scores arrive in the exp2 domain and the softmax needs no multiply.
// Reference sites: libs/croco/blocks.py:123-125, models/utils/transformer_blocks.py:244-246, 373-375 (F.scaled_dot_product_attention).
#include "common.h"

typedef __bf16 x3_bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned x3_uint4_t __attribute__((ext_vector_type(4)));

#define X3_TILE_BYTES (64 * 128)   // 64 rows x 128 B (one bf16 tile: 64 keys x 64 channels, or 64 channels x 64 key positions)

__device__ __forceinline__ int x3_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int x3_key_of_pos(int pp) { const int hi = pp >> 3, j = pp & 7; return (j & 3) + 8 * (j >> 2) + 4 * hi; }

__device__ __forceinline__ void x3_dma16(unsigned voff, x3_uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
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
__device__ __forceinline__ x3_uint4_t x3_make_srd(const void* base, unsigned bytes) {
    const unsigned long long pa = (unsigned long long)base;
    return (x3_uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}

// ---------------------------------------------------------------------------------------------------------------------------
// split passes.  rows: fp32 [B, N, H, 64] strided (unit channel stride) -> hi, lo bf16 [B, N, H, 64] contiguous, x * mul first.
// ---------------------------------------------------------------------------------------------------------------------------
// With pos != nullptr the rows are rotated by RoPE-2D on the way (pos [B, N, 2] int64, table [npos][16] (cos, sin) of uc_rope_table: the
// arithmetic of uc_rope2d, channels [0,16) = u_y, [16,32) = v_y, [32,48) = u_x, [48,64) = v_x, u' = u cos - v sin, v' = v cos + u sin):
// the rotated q / k never make a pass through memory of their own.
__global__ __launch_bounds__(256) void x3_split_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int N, int H,
                                                            int64_t sb, int64_t sn, int64_t sh, float mul, int64_t n_items,
                                                            const int64_t* __restrict__ pos, const float2* __restrict__ table, int npos) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i & 7);
        const int64_t row = i >> 3;                 // (b * N + n) * H + h
        const int h = (int)(row % H);
        const int64_t bn = row / H;
        const int n = (int)(bn % N);
        const int64_t b = bn / N;
        const float* base = x + b * sb + (int64_t)n * sn + (int64_t)h * sh;
        const float4_t* src = reinterpret_cast<const float4_t*>(base + c8 * 8);
        const float4_t u = src[0], w = src[1];
        float v[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
        if (pos) {
            const int axis = c8 >> 2, is_v = (c8 >> 1) & 1, i0 = (c8 & 1) * 8;
            const float4_t* psrc = reinterpret_cast<const float4_t*>(base + (is_v ? c8 - 2 : c8 + 2) * 8);     // the pair's other half
            const float4_t pu = psrc[0], pw = psrc[1];
            const float o[8] = {pu.x, pu.y, pu.z, pu.w, pw.x, pw.y, pw.z, pw.w};
            int64_t p = pos[bn * 2 + axis];
            p = p < 0 ? 0 : (p >= npos ? npos - 1 : p);          // (the launcher's caller guarantees the range; no wild reads if it does not)
            const float2* cs = table + p * 16 + i0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 c = cs[k];
                v[k] = is_v ? v[k] * c.x + o[k] * c.y : v[k] * c.x - o[k] * c.y;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= mul;
        unsigned hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hh[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
            ll[k] = pack_bf16x2(v[2 * k] - __uint_as_float(hh[k] << 16), v[2 * k + 1] - __uint_as_float(hh[k] & 0xffff0000u));
        }
        *reinterpret_cast<uint4*>(hi + row * 64 + c8 * 8) = (uint4){hh[0], hh[1], hh[2], hh[3]};
        *reinterpret_cast<uint4*>(lo + row * 64 + c8 * 8) = (uint4){ll[0], ll[1], ll[2], ll[3]};
    }
}

// V fp32 [B, Nk, H, 64] strided -> VT hi, lo [B, H, 64, npad] in the packed key order of uc_vt_pack (zeros past Nk)
__global__ __launch_bounds__(256) void x3_vt_pack_kernel(const float* __restrict__ V, bf16_t* __restrict__ VTh, bf16_t* __restrict__ VTl, int H, int Nk,
                                                         int npad, int64_t v_sb, int64_t v_sn, int64_t v_sh) {
    __shared__ __attribute__((aligned(16))) float tile[64][64 + 4];   // [key][d]
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
    const float* vb = V + (int64_t)b * v_sb + (int64_t)h * v_sh;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = threadIdx.x + it * 256;          // 1024 chunks of 4 floats: key = idx >> 4, channels (idx & 15) * 4 .. +3
        const int key = idx >> 4, c4 = idx & 15;
        float4_t v = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (k0 + key < Nk) v = *reinterpret_cast<const float4_t*>(vb + (int64_t)(k0 + key) * v_sn + c4 * 4);
        *reinterpret_cast<float4_t*>(&tile[key][c4 * 4]) = v;
    }
    __syncthreads();
    const int64_t base = ((int64_t)b * H + h) * 64 * (int64_t)npad + k0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = threadIdx.x + it * 256;          // 512 chunks: channel d = idx >> 3, positions (idx & 7) * 8 .. +7
        const int d = idx >> 3, p8 = (idx & 7) * 8;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = p8 + j;
            e[j] = tile[(pos & ~15) + x3_key_of_pos(pos & 15)][d];
        }
        unsigned hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hh[k] = pack_bf16x2(e[2 * k], e[2 * k + 1]);
            ll[k] = pack_bf16x2(e[2 * k] - __uint_as_float(hh[k] << 16), e[2 * k + 1] - __uint_as_float(hh[k] & 0xffff0000u));
        }
        *reinterpret_cast<uint4*>(VTh + base + (int64_t)d * npad + p8) = (uint4){hh[0], hh[1], hh[2], hh[3]};
        *reinterpret_cast<uint4*>(VTl + base + (int64_t)d * npad + p8) = (uint4){ll[0], ll[1], ll[2], ll[3]};
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct X3Params {
    const bf16_t *Qh, *Ql, *Kh, *Kl, *VTh, *VTl;   // split operands, contiguous [B, N, H, 64] / [B, H, 64, npad]
    float* O;
    int B, H, Nq, Nk, npad;
    int64_t o_sb, o_sn, o_sh;
    float* lse;                                     // optional [B, H, Nq]: natural-log sum-exp of the scaled scores
    uc_fastdiv dGroup, dNq, dH;
};

__global__ __launch_bounds__(256, 2) void attn_x3_kernel(X3Params p) {
    // 2 stages x (Kh, Kl, VTh, VTl); the Q rows (hi, lo: 2 x 4 KiB per wave) arrive in the second stage before the ring needs it
    __shared__ __attribute__((aligned(16))) char smem[8 * X3_TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    constexpr int QT = 128;
    const int nq = (p.Nq + QT - 1) / QT;
    const int nbh = p.B * p.H;
    int qt, bh;
    {   // query tiles of one (batch, head) meet on one XCD (see attn_bf16_dma_kernel)
        const int w = blockIdx.x;
        const int per_group = 8 * nq;
        const int grp = (int)uc_div((unsigned)w, p.dGroup), within = w - grp * per_group;
        if ((grp + 1) * 8 <= nbh) { bh = grp * 8 + (within & 7); qt = within >> 3; }
        else {
            const int rem = w - (nbh >> 3) * 8 * nq, rb = (int)uc_div((unsigned)rem, p.dNq);
            bh = (nbh >> 3) * 8 + rb; qt = rem - rb * nq;
        }
    }
    const int b = (int)uc_div((unsigned)bh, p.dH), h = bh - b * p.H;
    const int q0 = qt * QT + wave * 32;
    const int64_t row_stride = (int64_t)p.H * 64;         // elements between tokens of the contiguous split operands
    const int64_t kbase = ((int64_t)b * p.Nk * p.H + h) * 64, qbase = ((int64_t)b * p.Nq * p.H + h) * 64;
    const int64_t vbase = ((int64_t)b * p.H + h) * 64 * (int64_t)p.npad;

    const unsigned k_bytes = (unsigned)((((int64_t)p.Nk - 1) * row_stride + 64) * 2);    // key rows >= Nk read as zeros
    const x3_uint4_t srd_kh = x3_make_srd(p.Kh + kbase, k_bytes), srd_kl = x3_make_srd(p.Kl + kbase, k_bytes);
    const x3_uint4_t srd_vh = x3_make_srd(p.VTh + vbase, 0xffffff00u), srd_vl = x3_make_srd(p.VTl + vbase, 0xffffff00u);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    unsigned voff_k[2], voff_v[2];       // a tile is 8 instructions of 8 rows x 128 B; wave w issues instructions 2w, 2w + 1 of all four tiles
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = (wave * 2 + i) * 8 + (lane >> 3);
        const int cch = (lane & 7) ^ ((rr >> 1) & 7);
        voff_k[i] = (unsigned)(((int64_t)rr * row_stride + cch * 8) * 2);
        voff_v[i] = (unsigned)(((int64_t)rr * p.npad + cch * 8) * 2);
    }
    const unsigned kstep = (unsigned)(64 * row_stride * 2);
    auto issue_tile = [&](int t, int buf) {
        const unsigned dst = lds0 + (unsigned)(buf * 4 * X3_TILE_BYTES + wave * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x3_dma16(voff_k[i], srd_kh, (unsigned)t * kstep, __builtin_amdgcn_readfirstlane(dst + i * 1024));
            x3_dma16(voff_k[i], srd_kl, (unsigned)t * kstep, __builtin_amdgcn_readfirstlane(dst + X3_TILE_BYTES + i * 1024));
            x3_dma16(voff_v[i], srd_vh, (unsigned)t * 128u, __builtin_amdgcn_readfirstlane(dst + 2 * X3_TILE_BYTES + i * 1024));
            x3_dma16(voff_v[i], srd_vl, (unsigned)t * 128u, __builtin_amdgcn_readfirstlane(dst + 3 * X3_TILE_BYTES + i * 1024));
        }
    };
    {   // Q rows of this wave (hi, lo) into the second stage
        const int64_t q_rows = min((int64_t)32, (int64_t)p.Nq - q0);
        const unsigned q_bytes = q_rows > 0 ? (unsigned)(((q_rows - 1) * row_stride + 64) * 2) : 0u;
        const x3_uint4_t srd_qh = x3_make_srd(p.Qh + qbase + (int64_t)q0 * row_stride, q_bytes);
        const x3_uint4_t srd_ql = x3_make_srd(p.Ql + qbase + (int64_t)q0 * row_stride, q_bytes);
        const unsigned dstq = lds0 + (unsigned)(4 * X3_TILE_BYTES + wave * 8192);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const int cch = (lane & 7) ^ ((rr >> 1) & 7);
            const unsigned vo = (unsigned)(((int64_t)rr * row_stride + cch * 8) * 2);
            x3_dma16(vo, srd_qh, 0u, __builtin_amdgcn_readfirstlane(dstq + i * 1024));
            x3_dma16(vo, srd_ql, 0u, __builtin_amdgcn_readfirstlane(dstq + 4096 + i * 1024));
        }
    }
    int r_off[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = x3_swz(l31, 2 * st + hi);

    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    x3_bf16x8_t qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *reinterpret_cast<const x3_bf16x8_t*>(smem + 4 * X3_TILE_BYTES + wave * 8192 + r_off[s]);
        ql[s] = *reinterpret_cast<const x3_bf16x8_t*>(smem + 4 * X3_TILE_BYTES + wave * 8192 + 4096 + r_off[s]);
    }
    __syncthreads();

    float16_t o[2];
    o[0] = (float16_t)(0.f);
    o[1] = (float16_t)(0.f);
    float m_run = -1e30f;      // running max, in the exp2 domain (Q carries scale * log2 e)
    float l_run = 0.f;
    const int nt = (p.Nk + 63) / 64;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
        const char* skh = smem + buf * 4 * X3_TILE_BYTES;
        const char* skl = skh + X3_TILE_BYTES;
        const char* svh = skh + 2 * X3_TILE_BYTES;
        const char* svl = skh + 3 * X3_TILE_BYTES;

        float16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = (float16_t)(0.f);
#pragma unroll
            for (int st = 0; st < 4; ++st) {      // the two small terms first, the large one last
                const x3_bf16x8_t kh = *reinterpret_cast<const x3_bf16x8_t*>(skh + r_off[st] + kb * (32 * 128));
                const x3_bf16x8_t kl = *reinterpret_cast<const x3_bf16x8_t*>(skl + r_off[st] + kb * (32 * 128));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[st], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[st], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[st], s[kb], 0, 0, 0);
            }
        }
        if (t == nt - 1 && (p.Nk & 63)) {       // ragged last tile: keys >= Nk (zero K rows, zero VT pads) leave the softmax
            const int k0 = t * 64;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) s[kb][r] = -1e30f;
        }
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        // deferred rescale as in the bf16 kernel: the old running max is kept while no query of the wave grew by more than 2^8 (P <= 256;
        // the hi / lo split of P is floating point, so its RELATIVE accuracy does not depend on the scale), the O / l rescale then
        // runs on a wave-uniform branch
        if (__any(mt - m_run > 8.0f)) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        float psum = 0.f;
        x3_bf16x8_t ph[4], pl[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(s[kb][hf * 8 + j] - m_run);
                    psum += e[j];
                }
                union { x3_bf16x8_t v; unsigned u[4]; } a, c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a.u[j] = pack_bf16x2(e[2 * j], e[2 * j + 1]);
                    c.u[j] = pack_bf16x2(e[2 * j] - __uint_as_float(a.u[j] << 16), e[2 * j + 1] - __uint_as_float(a.u[j] & 0xffff0000u));
                }
                ph[kb * 2 + hf] = a.v;
                pl[kb * 2 + hf] = c.v;
            }
        }
        l_run += psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const x3_bf16x8_t vh = *reinterpret_cast<const x3_bf16x8_t*>(svh + r_off[g] + db * (32 * 128));
                const x3_bf16x8_t vl = *reinterpret_cast<const x3_bf16x8_t*>(svl + r_off[g] + db * (32 * 128));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[g], o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[g], o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[g], o[db], 0, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (p.lse && q < p.Nq && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run * 0.69314718055994530942f + logf(l_tot);
    // bounce the wave's 32 x 64 fp32 outputs through its private 8 KiB of the ring: every store instruction writes 4 whole 256-B rows
    char* ob = smem + wave * 8192;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4_t v = (float4_t){o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv, o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv};
            const int chunk = 8 * db + 2 * g4 + hi;           // channels 32 db + 8 g4 + 4 hi .. + 3
            *reinterpret_cast<float4_t*>(ob + l31 * 256 + ((chunk ^ (l31 & 15)) << 4)) = v;
        }
    float* obase = p.O + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int R = 4 * ps + (lane >> 4);
        const int chunk = (lane & 15) ^ (R & 15);
        const float4_t v = *reinterpret_cast<const float4_t*>(ob + R * 256 + ((lane & 15) << 4));
        if (q0 + R < p.Nq) *reinterpret_cast<float4_t*>(obase + (int64_t)(q0 + R) * p.o_sn + chunk * 4) = v;
    }
}

extern "C" int64_t uc_attention_x3_workspace_bytes(int B, int H, int Nq, int Nk) {
    const int64_t npad = ((int64_t)Nk + 63) / 64 * 64;
    return 2 * 2 * ((int64_t)B * Nq * H * 64 + (int64_t)B * Nk * H * 64 + (int64_t)B * H * 64 * npad) + 256;
}

extern "C" int uc_attention_fwd_x3(const float* Q, const float* K, const float* V, float* O, void* workspace, int B, int H, int Nq, int Nk,
                                   int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                                   int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, float* lse,
                                   const int64_t* q_pos, const int64_t* k_pos, const float* rope_table, int rope_npos, uc_stream_t stream) {
    UC_REQUIRE(Q && K && V && O && workspace, "uc_attention_fwd_x3: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "uc_attention_fwd_x3: bad shape");
    UC_REQUIRE((!q_pos && !k_pos) || (q_pos && k_pos && rope_table && rope_npos > 0), "uc_attention_fwd_x3: RoPE needs both position arrays and the table");
    UC_REQUIRE(q_sb % 4 == 0 && q_sn % 4 == 0 && q_sh % 4 == 0 && k_sb % 4 == 0 && k_sn % 4 == 0 && k_sh % 4 == 0 && v_sb % 4 == 0 && v_sn % 4 == 0 &&
                   v_sh % 4 == 0 && o_sb % 4 == 0 && o_sn % 4 == 0 && o_sh % 4 == 0,
               "uc_attention_fwd_x3: strides must be multiples of 4 elements");
    UC_REQUIRE((uintptr_t)Q % 16 == 0 && (uintptr_t)K % 16 == 0 && (uintptr_t)V % 16 == 0 && (uintptr_t)O % 16 == 0 && (uintptr_t)workspace % 16 == 0,
               "uc_attention_fwd_x3: 16-byte alignment");
    const int npad = (Nk + 63) / 64 * 64;
    // 32-bit byte offsets of the LDS-DMA inside one (batch, head)'s rows
    UC_REQUIRE((int64_t)Nk * H * 64 * 2 < ((int64_t)1 << 31) && (int64_t)Nq * H * 64 * 2 < ((int64_t)1 << 31) && (int64_t)64 * npad * 2 < ((int64_t)1 << 31),
               "uc_attention_fwd_x3: sequence too long for 32-bit tile offsets");
    const int nqt = (Nq + 127) / 128;
    UC_REQUIRE((int64_t)nqt * H * B < ((int64_t)1 << 31), "uc_attention_fwd_x3: grid too large");
    hipStream_t st = (hipStream_t)stream;
    bf16_t* w = (bf16_t*)workspace;
    const int64_t nQ = (int64_t)B * Nq * H * 64, nK = (int64_t)B * Nk * H * 64, nV = (int64_t)B * H * 64 * npad;
    bf16_t *Qh = w, *Ql = Qh + nQ, *Kh = Ql + nQ, *Kl = Kh + nK, *VTh = Kl + nK, *VTl = VTh + nV;
    const int64_t iq = nQ / 8, ik = nK / 8;
    const unsigned gq = (unsigned)std::min<int64_t>((iq + 255) / 256, 65535 * 16), gk = (unsigned)std::min<int64_t>((ik + 255) / 256, 65535 * 16);
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(gq), dim3(256), 0, st, Q, Qh, Ql, Nq, H, q_sb, q_sn, q_sh, scale * 1.44269504088896340736f, iq,
                       q_pos, (const float2*)rope_table, rope_npos);
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(gk), dim3(256), 0, st, K, Kh, Kl, Nk, H, k_sb, k_sn, k_sh, 1.0f, ik, k_pos, (const float2*)rope_table, rope_npos);
    UC_REQUIRE(H <= 65535 && B <= 65535, "uc_attention_fwd_x3: B and H must fit a grid dimension");
    hipLaunchKernelGGL(x3_vt_pack_kernel, dim3(npad / 64, H, B), dim3(256), 0, st, V, VTh, VTl, H, Nk, npad, v_sb, v_sn, v_sh);
    X3Params p;
    p.Qh = Qh; p.Ql = Ql; p.Kh = Kh; p.Kl = Kl; p.VTh = VTh; p.VTl = VTl; p.O = O;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.npad = npad; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh; p.lse = lse;
    p.dGroup = uc_make_fastdiv((unsigned)(8 * nqt)); p.dNq = uc_make_fastdiv((unsigned)nqt); p.dH = uc_make_fastdiv((unsigned)H);
    hipLaunchKernelGGL(attn_x3_kernel, dim3((unsigned)(nqt * H * B)), dim3(256), 0, st, p);
    UC_CHECK_LAUNCH("uc_attention_fwd_x3");
    return UC_OK;
}
