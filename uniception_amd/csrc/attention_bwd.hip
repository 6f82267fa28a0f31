// Attention backward for gfx950 (bf16, head_dim 64, v_mfma_f32_32x32x16_bf16), flash-style recomputation.
//
//   S = scale Q K^T,  P = exp(S - LSE),  dV = P^T dO,  dP = dO V^T,  D = rowsum(dO * O),
//   dS = P * (dP - D),  dQ = scale dS K,  dK = scale dS^T Q.
//
// Two kernels, no atomics, every operand read from ROW-MAJOR tiles:
//  * attn_bwd_dq_kernel  — a workgroup owns 128 queries (4 waves x 32) and streams 64-key K/V tiles.  Swapped products
//    keep a query in one lane pair: S^T = K Q^T, dP^T = V dO^T (A operands = K / V rows via ds_read_b128, B operands =
//    Q^T / dO^T in registers); dS^T in accumulator layout is the B operand of  dQ^T += K^T dS^T.
//  * attn_bwd_dkv_kernel — a workgroup owns 128 keys and streams 64-query Q/dO tiles.  Un-swapped products keep a KEY in
//    one lane pair: S = Q K^T, dP = dO V^T; P and dS in accumulator layout are the B operands of  dV^T += dO^T P  and
//    dK^T += Q^T dS.
// The transposed A operands (K^T, Q^T, dO^T: 8 consecutive keys/queries of one channel per lane) come out of the SAME
// row-major LDS tiles through gfx950's transposing LDS read (ds_read_b64_tr_b16): a 16-lane group hands in the addresses of
// four rows and receives, per lane, one column of that 4x16 block.  The rows need not be adjacent, so they are picked to
// match the accumulator register order of P / dS ((r&3) + 8*(r>>2) + 4*hi): rows base..base+3 and base+8..base+11 —
// no packed transposes, no permutation pass.
// LDS tiles are 64 rows x 128 B with the (row>>1)&7 chunk swizzle (conflict-free ds_read_b128; the transposing reads see
// 2-way conflicts between rows r and r+2, which the MFMA/VALU work hides), single-buffered, 2-3 workgroups per CU.
#include "common.h"
#include "knobs.h"
#include <math.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* ab_lds_ptr_t;
typedef __attribute__((address_space(3))) bf16x4_t* ab_lds_v4_t;

struct AttnBwdParams {
    const bf16_t *Q, *K, *V, *O, *dO;
    const float* LSE;
    float* delta;
    float* aux;                 // [B H][2][nq_pad]: -lse * log2(e) (absent queries: -1e30) | -delta (0): the dK / dV kernel's accumulator start values
    int nq_pad;                 // Nq rounded up to 128
    bf16_t *dQ, *dK, *dV;
    int B, H, Nq, Nk;
    int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
    int64_t dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh;
    float scale;
    // inverse RoPE-2D of dQ / dK in the kernels' epilogues (round 4; NULL positions: gradients of the ROTATED q / k, as before)
    const int64_t* rope_qpos;   // [B * Nq][2] (y, x)
    const int64_t* rope_kpos;   // [B * Nk][2]
    float rope_turn0;           // F0 / (2 pi): rotation per unit position of channel 0, in turns
    float rope_ratio;           // base^(-1/16)
    unsigned long long* dbg;    // probe builds (-DB64_TIMING): per-workgroup cycle stamps
    UcDropout drop;             // attention dropout (uc_attention_bwd_drop; thr 0 elsewhere): the forward's mask, re-evaluated
};

#define TB (64 * 128)   // bytes of one 64-row tile

__device__ __forceinline__ int bswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Transposed MFMA A operand out of a row-major [64 rows][64 channels] swizzled tile: for the 32-channel block `cb` the lane
// (channel = 32 cb + (lane & 31), k half = lane >> 5) receives rows  slab16*16 + 4*(lane>>5) + {0,1,2,3, 8,9,10,11}.
__device__ __forceinline__ bf16x8_t tr_operand(const char* tile, int slab16, int cb, int lane) {
    const int jj = lane & 15;
    const int d0 = cb * 32 + (((lane >> 4) & 1) << 4);            // first channel of this 16-lane group's block
    const int piece = jj & 3;
    const int chunk = (d0 >> 3) + (piece >> 1);
    const int row0 = slab16 * 16 + 4 * (lane >> 5) + (jj >> 2);
    const int row1 = row0 + 8;
    const char* a0 = tile + row0 * 128 + ((chunk ^ ((row0 >> 1) & 7)) << 4) + ((piece & 1) << 3);
    const char* a1 = tile + row1 * 128 + ((chunk ^ ((row1 >> 1) & 7)) << 4) + ((piece & 1) << 3);
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ab_lds_v4_t)(ab_lds_ptr_t)a0);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ab_lds_v4_t)(ab_lds_ptr_t)a1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Asynchronous global -> LDS copy of one [64 rows][64 bf16] tile in the swizzled image (global_load_lds_dwordx4, 1 KiB per
// wave-instruction: lane -> row 8 n + (lane >> 3), physical chunk lane & 7 <- logical chunk (lane & 7) ^ ((row >> 1) & 7)).
// Wave w issues instructions 2w and 2w+1; rows beyond n_rows re-read the last valid row (their scores are masked).
__device__ __forceinline__ void ab_dma16(const void* gsrc, unsigned lds_byte_addr) {
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

__device__ __forceinline__ void dma_tile(const bf16_t* base, int64_t row_stride, int row0, int n_rows, unsigned lds_tile, int wave,
                                         int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = wave * 2 + q;
        const int rr = n * 8 + (lane >> 3);
        int row = row0 + rr;
        if (row >= n_rows) row = n_rows - 1;
        const int c = (lane & 7) ^ ((rr >> 1) & 7);
        ab_dma16(base + (int64_t)row * row_stride + c * 8, __builtin_amdgcn_readfirstlane(lds_tile + (unsigned)(n * 1024)));
    }
}

// Inverse RoPE-2D on a gradient held in the kernels' accumulator layout: g[db][4 g4 + r] = channel 32 db + 8 g4 + 4 hi + r of one token
// (db = 0: y half, db = 1: x half; inside a half the pair (u, v) = channels (c, c + 16), c = 8 g4 + 4 hi + r with g4 in {0, 1}: the same
// lane).  The forward rotated (u, v) by angle pos * F0 * base^(-c/16) (kernels.cu:36-81); the gradient goes back by the opposite
// angle: du = du' cos + dv' sin, dv = dv' cos - du' sin.  Angles through the hardware sin / cos on fract(pos * turn), like the forward's
// GEMM epilogue.  Replaces one uc_rope2d pass over dq and one over dk per attention (144 launches per training step).
__device__ __forceinline__ void ab_rope_inverse(float16_t (&g)[2], const int64_t* pos2, int hi, float turn0, float ratio) {
    const float py = (float)(int)pos2[0], px = (float)(int)pos2[1];
    float t4 = turn0;                                   // turn of channel 4 hi (hi = 1: x ratio^4)
    if (hi) { const float r2 = ratio * ratio; t4 *= r2 * r2; }
    const float r8 = (ratio * ratio) * (ratio * ratio) * (ratio * ratio) * (ratio * ratio);
#pragma unroll
    for (int g4 = 0; g4 < 2; ++g4) {
        float turn = g4 ? t4 * r8 : t4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const float tf = __builtin_amdgcn_fractf((db ? px : py) * turn);
                const float cs = __builtin_amdgcn_cosf(tf), sn = __builtin_amdgcn_sinf(tf);
                const float u = g[db][g4 * 4 + r], v = g[db][(g4 + 2) * 4 + r];
                g[db][g4 * 4 + r] = __builtin_fmaf(u, cs, v * sn);
                g[db][(g4 + 2) * 4 + r] = __builtin_fmaf(v, cs, -(u * sn));
            }
            turn *= ratio;
        }
    }
}

// delta[b,h,q] = sum_d dO[q,d] * O[q,d]
__global__ void attn_delta_kernel(AttnBwdParams p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.H * p.Nq;
    if (idx >= total) return;
    const int q = (int)(idx % p.Nq);
    const int h = (int)((idx / p.Nq) % p.H);
    const int b = (int)(idx / ((int64_t)p.Nq * p.H));
    const bf16_t* o = p.O + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
    const bf16_t* d = p.dO + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 a = *reinterpret_cast<const uint4*>(o + c * 8);
        const uint4 g = *reinterpret_cast<const uint4*>(d + c * 8);
        const unsigned* au = reinterpret_cast<const unsigned*>(&a);
        const unsigned* gu = reinterpret_cast<const unsigned*>(&g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s += __uint_as_float(au[e] << 16) * __uint_as_float(gu[e] << 16);
            s += __uint_as_float(au[e] & 0xffff0000u) * __uint_as_float(gu[e] & 0xffff0000u);
        }
    }
    p.delta[idx] = s;
}

// =================================================================================================================
// dQ
// =================================================================================================================
// 1-D grid -> (tile, batch*head): workgroup w runs on XCD w % 8 (round-robin dispatch), so the nt tiles that share one
// (batch, head)'s operands get ids that differ by multiples of 8 and meet in ONE XCD's L2 (see attn_bf16_dma_kernel)
__device__ __forceinline__ void ab_xcd_order(int w, int nt, int nbh, int& tile, int& bh) {
    const int per_group = 8 * nt;
    const int grp = w / per_group, within = w - grp * per_group;
    if ((grp + 1) * 8 <= nbh) { bh = grp * 8 + (within & 7); tile = within >> 3; }
    else { const int rem = w - (nbh / 8) * 8 * nt; bh = (nbh / 8) * 8 + rem / nt; tile = rem % nt; }
}

// DROP (attention dropout, uc_attention_bwd_drop): the forward computed O = P' V with P' = P o mask / (1 - p).  dV = P'^T dO;
// dP' = dO V^T; dP = dP' o mask / (1 - p); dS = P o (dP - delta) with delta = rowsum(dP o P) = rowsum(dP' o P') = rowsum(dO o O) as
// without dropout.  The accumulator chain that starts at -delta holds dP' - delta: dS = P ((acc + delta) m - delta), m = mask / (1 - p).
template <bool DROP = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem_all[4 * TB];   // 2 stages x (K rows | V rows), filled by LDS-DMA
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    int tile_, bh_;
    ab_xcd_order((int)blockIdx.x, (p.Nq + 127) / 128, p.B * p.H, tile_, bh_);
    const int b = bh_ / p.H, h = bh_ - b * p.H;
    const int q0 = tile_ * 128 + wave * 32;
    int q = q0 + l31;
    const bool q_ok = q < p.Nq;
    if (!q_ok) q = p.Nq - 1;

    const bf16_t* Kb = p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* Vb = p.V + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;

    bf16x8_t qf[4], dof[4];
    {
        const bf16_t* qp = p.Q + (int64_t)b * p.q_sb + (int64_t)q * p.q_sn + (int64_t)h * p.q_sh + hi * 8;
        const bf16_t* dp = p.dO + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh + hi * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * s);
            dof[s] = *reinterpret_cast<const bf16x8_t*>(dp + 16 * s);
        }
    }
    const float c = p.scale * 1.44269504088896340736f;
    const float lse2 = p.LSE[((int64_t)b * p.H + h) * p.Nq + q] * 1.44269504088896340736f;
    // Softmax diet (round 5, the forward's attention_p64.h): the wave's STATIONARY operand Q is pre-multiplied by scale * log2(e)
    // (once per workgroup; re-rounded to bf16: scores carry ~2^-9 |q| |k| scale of noise, below P's own bf16 rounding for the scores
    // that matter), the score chain starts at -lse2 and the dP chain at -delta (C operand of the first MFMA: two register blocks
    // holding the lane's query scalars) — per element P = exp2(acc), dS = P * acc': exp + mul + half a conversion instead of
    // fma + exp + sub + mul + half a conversion.
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        union { bf16x8_t v; unsigned u[4]; } a;
        a.v = qf[s];
#pragma unroll
        for (int j = 0; j < 4; ++j) a.u[j] = pack_bf16x2(__uint_as_float(a.u[j] << 16) * c, __uint_as_float(a.u[j] & 0xffff0000u) * c);
        qf[s] = a.v;
    }
    // delta[q] = sum_d dO[q,d] * O[q,d] (round 4: computed HERE — the lane pair of a query already holds its dO row as dof; one more row
    // load and a cross-half add replace the stand-alone pass of rounds 1-3) and left in p.delta for the dK / dV kernel, which runs next
    float dlt = 0.f;
    {
        const bf16_t* op = p.O + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh + hi * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(op + 16 * s);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt = fmaf((float)of[j], (float)dof[s][j], dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        if (hi == 0) {      // (every query slot up to nq_pad gets its pair: the 64-key dK / dV kernel reads whole 64-query tiles of them)
            float* ax = p.aux + ((int64_t)b * p.H + h) * 2 * p.nq_pad + (q0 + l31);
            ax[0] = q_ok ? -lse2 : -1e30f;
            ax[p.nq_pad] = q_ok ? -dlt : 0.f;
        }
    }

    unsigned dk1 = 0, dk2 = 0;
    if constexpr (DROP) uc_drop_keys(p.drop, (unsigned)(b * p.H + h), dk1, dk2);
    float16_t dq[2];
    dq[0] = (float16_t)(0.f);
    dq[1] = (float16_t)(0.f);
    float16_t neg_lse, neg_dlt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { neg_lse[r] = -lse2; neg_dlt[r] = -dlt; }
    asm volatile("" : "+v"(neg_lse), "+v"(neg_dlt));      // opaque: a splat hipcc recognises is re-materialised (16 v_mov) per use

    int r_off[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = bswz(l31, 2 * st + hi);
    const unsigned lds0 = (unsigned)(size_t)(ab_lds_ptr_t)smem_all;

    const int nt = (p.Nk + 63) / 64;
    dma_tile(Kb, p.k_sn, 0, p.Nk, lds0, wave, lane);
    dma_tile(Vb, p.v_sn, 0, p.Nk, lds0 + TB, wave, lane);
    for (int t = 0; t < nt; ++t) {
        const int k0 = t * 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // tile t landed for every wave; everyone is done reading the other buffer
        asm volatile("" ::: "memory");
        if (t + 1 < nt) {
            dma_tile(Kb, p.k_sn, k0 + 64, p.Nk, lds0 + ((t + 1) & 1) * 2 * TB, wave, lane);
            dma_tile(Vb, p.v_sn, k0 + 64, p.Nk, lds0 + ((t + 1) & 1) * 2 * TB + TB, wave, lane);
        }
        const char* smem = smem_all + (t & 1) * 2 * TB;
        const bool tail = k0 + 64 > p.Nk;

        bf16x8_t dsf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float16_t s = neg_lse, dp = neg_dlt;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(smem + r_off[st] + kb * (32 * 128));
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(smem + TB + r_off[st] + kb * (32 * 128));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);          // scale log2(e) Q K^T - lse2
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[st], dp, 0, 0, 0);       // dO V^T - delta
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = hf * 8 + j;
                    float pv = __builtin_amdgcn_exp2f(s[r]);
                    if (tail && k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) pv = 0.f;   // ragged last tile only (uniform test first)
                    if constexpr (DROP) {
                        const unsigned key = (unsigned)(k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                        const float m = uc_drop_hash(dk1, dk2, (unsigned)(q0 + l31), key) >= p.drop.thr ? p.drop.keep_scale : 0.f;
                        e[j] = pv * __builtin_fmaf(dp[r] + dlt, m, -dlt);
                    } else
                    e[j] = pv * dp[r];
                }
                union { bf16x8_t v; unsigned u[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf16x2(e[2 * j], e[2 * j + 1]);
                dsf[kb * 2 + hf] = pk.v;
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // K^T[d-block db, 16-key slab g]: clamped duplicate rows of a tail tile meet dS = 0
                dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(smem, g, db, lane), dsf[g], dq[db], 0, 0, 0);
            }
    }
    if (p.rope_qpos) ab_rope_inverse(dq, p.rope_qpos + ((int64_t)b * p.Nq + q) * 2, hi, p.rope_turn0, p.rope_ratio);     // (linear: the scale below commutes)
    if (q_ok) {
        bf16_t* op = p.dQ + (int64_t)b * p.dq_sb + (int64_t)q * p.dq_sn + (int64_t)h * p.dq_sh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * hi;
                uint2 pk;
                pk.x = pack_bf16x2(dq[db][g4 * 4 + 0] * p.scale, dq[db][g4 * 4 + 1] * p.scale);
                pk.y = pack_bf16x2(dq[db][g4 * 4 + 2] * p.scale, dq[db][g4 * 4 + 3] * p.scale);
                *reinterpret_cast<uint2*>(op + d) = pk;
            }
    }
}

// =================================================================================================================
// dK, dV
// =================================================================================================================
template <bool DROP = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem_all[4 * TB + 1024];   // 2 stages x (Q rows | dO rows) + 2 x (lse2[64] delta[64])
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    int tile_, bh_;
    ab_xcd_order((int)blockIdx.x, (p.Nk + 127) / 128, p.B * p.H, tile_, bh_);
    const int b = bh_ / p.H, h = bh_ - b * p.H;
    const int key0 = tile_ * 128 + wave * 32;
    int key = key0 + l31;
    const bool key_ok = key < p.Nk;
    if (!key_ok) key = p.Nk - 1;

    const bf16_t* Qb = p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* dOb = p.dO + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
    const float* ax_b = p.aux + ((int64_t)b * p.H + h) * 2 * p.nq_pad;      // written by the dQ kernel
    float* s_aux = reinterpret_cast<float*>(smem_all + 4 * TB);   // [stage][lse2 64 | delta 64]

    bf16x8_t kf[4], vf[4];   // B operands: lane key = l31, channels 16s + 8hi .. +7
    {
        const bf16_t* kp = p.K + (int64_t)b * p.k_sb + (int64_t)key * p.k_sn + (int64_t)h * p.k_sh + hi * 8;
        const bf16_t* vp = p.V + (int64_t)b * p.v_sb + (int64_t)key * p.v_sn + (int64_t)h * p.v_sh + hi * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8_t*>(kp + 16 * s);
            vf[s] = *reinterpret_cast<const bf16x8_t*>(vp + 16 * s);
        }
    }
    const float c = p.scale * 1.44269504088896340736f;
    // softmax diet (see the dQ kernel): K, this wave's stationary operand, carries scale * log2(e); the per-QUERY scalars -lse2 and
    // -delta are the C operands of the first MFMAs — here one value per accumulator ROW, read as they lie in LDS
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        union { bf16x8_t v; unsigned u[4]; } a;
        a.v = kf[s];
#pragma unroll
        for (int j = 0; j < 4; ++j) a.u[j] = pack_bf16x2(__uint_as_float(a.u[j] << 16) * c, __uint_as_float(a.u[j] & 0xffff0000u) * c);
        kf[s] = a.v;
    }
    float16_t dk[2], dv[2];
    dk[0] = (float16_t)(0.f); dk[1] = (float16_t)(0.f);
    dv[0] = (float16_t)(0.f); dv[1] = (float16_t)(0.f);
    unsigned dk1 = 0, dk2 = 0;
    if constexpr (DROP) uc_drop_keys(p.drop, (unsigned)(b * p.H + h), dk1, dk2);

    int r_off[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = bswz(l31, 2 * st + hi);
    const unsigned lds0 = (unsigned)(size_t)(ab_lds_ptr_t)smem_all;

    const int nt = (p.Nq + 63) / 64;
    auto stage_aux = [&](int stage, int q0) {   // per-query scalars of the tile: plain loads, 64 threads
        if (tid < 64) {
            const int q = q0 + tid;
            s_aux[stage * 128 + tid] = (q < p.Nq) ? ax_b[q] : -1e30f;       // (negated: accumulator start values) absent query: P = exp2(-1e30) = 0
            s_aux[stage * 128 + 64 + tid] = (q < p.Nq) ? ax_b[p.nq_pad + q] : 0.f;
        }
    };
    dma_tile(Qb, p.q_sn, 0, p.Nq, lds0, wave, lane);
    dma_tile(dOb, p.o_sn, 0, p.Nq, lds0 + TB, wave, lane);
    stage_aux(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int q0 = t * 64;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("" ::: "memory");
        if (t + 1 < nt) {
            dma_tile(Qb, p.q_sn, q0 + 64, p.Nq, lds0 + ((t + 1) & 1) * 2 * TB, wave, lane);
            dma_tile(dOb, p.o_sn, q0 + 64, p.Nq, lds0 + ((t + 1) & 1) * 2 * TB + TB, wave, lane);
            stage_aux((t + 1) & 1, q0 + 64);
        }
        const char* smem = smem_all + (t & 1) * 2 * TB;
        const float* s_lse = s_aux + (t & 1) * 128;
        const float* s_dl = s_lse + 64;

#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // accumulator row r = query qb*32 + (r & 3) + 8 (r >> 2) + 4 hi: registers 4k .. 4k+3 = one 16-byte LDS read
            float16_t s, dp, ndl;       // (ndl: the rows' -delta once more, for the dropout form of dS)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4_t l4 = *reinterpret_cast<const float4_t*>(s_lse + qb * 32 + 8 * k4 + 4 * hi);
                const float4_t d4 = *reinterpret_cast<const float4_t*>(s_dl + qb * 32 + 8 * k4 + 4 * hi);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { s[4 * k4 + jj] = l4[jj]; dp[4 * k4 + jj] = d4[jj]; if constexpr (DROP) ndl[4 * k4 + jj] = d4[jj]; }
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(smem + r_off[st] + qb * (32 * 128));
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(smem + TB + r_off[st] + qb * (32 * 128));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[st], s, 0, 0, 0);     // S[q rows, key cols]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[st], dp, 0, 0, 0);   // dP[q rows, key cols]
            }
            bf16x8_t pfr[2], dsfr[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float pe[8], de[8];
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8) {
                    const int r = hf * 8 + j8;
                    const float pv = __builtin_amdgcn_exp2f(s[r]);
                    if constexpr (DROP) {
                        const unsigned qq = (unsigned)(q0 + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                        const float m = uc_drop_hash(dk1, dk2, qq, (unsigned)(key0 + l31)) >= p.drop.thr ? p.drop.keep_scale : 0.f;
                        pe[j8] = pv * m;                                             // P' for dV
                        de[j8] = pv * __builtin_fmaf(dp[r] - ndl[r], m, ndl[r]);     // P ((acc + delta) m - delta), ndl = -delta
                    } else {
                    pe[j8] = pv;
                    de[j8] = pv * dp[r];
                    }
                }
                union { bf16x8_t v; unsigned u[4]; } a, d2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a.u[j] = pack_bf16x2(pe[2 * j], pe[2 * j + 1]);
                    d2.u[j] = pack_bf16x2(de[2 * j], de[2 * j + 1]);
                }
                pfr[hf] = a.v;
                dsfr[hf] = d2.v;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int g = qb * 2 + hf;   // 16-query slab inside the tile -> chunks 2g, 2g+1
                    // queries beyond Nq are clamped duplicates in the tiles; their P / dS columns are exactly 0
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(smem + TB, g, db, lane), pfr[hf], dv[db], 0, 0, 0);   // dV^T[d, key]
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(smem, g, db, lane), dsfr[hf], dk[db], 0, 0, 0);       // dK^T[d, key]
                }
        }
    }
    if (p.rope_kpos) ab_rope_inverse(dk, p.rope_kpos + ((int64_t)b * p.Nk + key) * 2, hi, p.rope_turn0, p.rope_ratio);
    if (key_ok) {
        bf16_t* kp = p.dK + (int64_t)b * p.dk_sb + (int64_t)key * p.dk_sn + (int64_t)h * p.dk_sh;
        bf16_t* vp = p.dV + (int64_t)b * p.dv_sb + (int64_t)key * p.dv_sn + (int64_t)h * p.dv_sh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * hi;
                uint2 a, v;
                a.x = pack_bf16x2(dk[db][g4 * 4 + 0] * p.scale, dk[db][g4 * 4 + 1] * p.scale);
                a.y = pack_bf16x2(dk[db][g4 * 4 + 2] * p.scale, dk[db][g4 * 4 + 3] * p.scale);
                v.x = pack_bf16x2(dv[db][g4 * 4 + 0], dv[db][g4 * 4 + 1]);
                v.y = pack_bf16x2(dv[db][g4 * 4 + 2], dv[db][g4 * 4 + 3]);
                *reinterpret_cast<uint2*>(kp + d) = a;
                *reinterpret_cast<uint2*>(vp + d) = v;
            }
    }
}

#include "attention_bwd64.h"

static int attention_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                              void* dQ, void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int64_t q_sb,
                              int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                              int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh,
                              int64_t dk_sb, int64_t dk_sn, int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale,
                              const int64_t* rope_qpos, const int64_t* rope_kpos, float rope_base, float rope_f0, const UcDropout& drop,
                              uc_stream_t stream) {
    UC_REQUIRE(Q && K && V && O && dO && LSE && dQ && dK && dV && delta, "uc_attention_bwd: null pointer");
    UC_REQUIRE((rope_qpos == nullptr) == (rope_kpos == nullptr), "uc_attention_bwd: rope_qpos and rope_kpos go together");
    UC_REQUIRE(!rope_qpos || (rope_base > 0.f && rope_f0 != 0.f), "uc_attention_bwd: the inverse RoPE needs base > 0 and F0 != 0");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && B <= 65535 && H <= 65535, "uc_attention_bwd: bad shape");
    UC_REQUIRE(q_sn % 8 == 0 && k_sn % 8 == 0 && v_sn % 8 == 0 && o_sn % 8 == 0 && q_sh % 8 == 0 && k_sh % 8 == 0 && v_sh % 8 == 0 &&
                   o_sh % 8 == 0 && q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0,
               "uc_attention_bwd: input strides must be multiples of 8 elements");
    UC_REQUIRE(dq_sn % 4 == 0 && dk_sn % 4 == 0 && dv_sn % 4 == 0 && dq_sh % 4 == 0 && dk_sh % 4 == 0 && dv_sh % 4 == 0 &&
                   dq_sb % 4 == 0 && dk_sb % 4 == 0 && dv_sb % 4 == 0,
               "uc_attention_bwd: output strides must be multiples of 4 elements");
    AttnBwdParams p;
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (const bf16_t*)O; p.dO = (const bf16_t*)dO;
    p.LSE = LSE; p.delta = delta; p.aux = delta;
    p.nq_pad = (Nq + 127) / 128 * 128;
    p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
    p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh; p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh;
    p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh; p.dq_sb = dq_sb; p.dq_sn = dq_sn; p.dq_sh = dq_sh; p.dk_sb = dk_sb; p.dk_sn = dk_sn;
    p.dk_sh = dk_sh; p.dv_sb = dv_sb; p.dv_sn = dv_sn; p.dv_sh = dv_sh; p.scale = scale;
    p.rope_qpos = rope_qpos; p.rope_kpos = rope_kpos; p.dbg = nullptr;
    p.rope_turn0 = rope_qpos ? (float)((double)rope_f0 / 6.283185307179586476925) : 0.f;
    p.rope_ratio = rope_qpos ? (float)pow((double)rope_base, -1.0 / 16.0) : 1.f;
    p.drop = drop;
    hipStream_t st = (hipStream_t)stream;
    if (drop.thr) {      // attention dropout: the 32-row kernels with the forward's mask re-evaluated per element
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3((unsigned)(((Nq + 127) / 128) * H * B)), dim3(256), 0, st, p);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3((unsigned)(((Nk + 127) / 128) * H * B)), dim3(256), 0, st, p);
        UC_CHECK_LAUNCH("uc_attention_bwd_drop");
        return UC_OK;
    }
    const int64_t total = (int64_t)B * H * Nq;
    (void)total;      // (delta is computed by the dQ kernel since round 4; attn_delta_kernel remains for reference)
    (void)uc_knobs();
    const int k64 = g_uc_attn_bwd64.load();
    const int64_t v_ext = ((int64_t)(B - 1) * v_sb + (int64_t)(H - 1) * v_sh + (int64_t)(Nk - 1) * v_sn + 64) * 2;
    // 64 queries per wave (attention_bwd64.h) when a 256-query workgroup is mostly real queries; the 32-query kernel otherwise
    const bool dq64 = Nk > 64 && v_ext < (int64_t)0xffffffffll && (k64 == 2 || (k64 == 1 && Nq >= 192 && ((Nq + 255) / 256) * 256 * 3 <= Nq * 4));
    if (dq64) {
        const int64_t items = (int64_t)((Nq + 255) / 256) * H * B;
        const int grid = (int)min((int64_t)(uc_num_cus() / 8 * 8), (items + 7) / 8 * 8);
        hipLaunchKernelGGL(attn_bwd_dq64_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
    } else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3((unsigned)(((Nq + 127) / 128) * H * B)), dim3(256), 0, st, p);
    // 64 keys per wave (attention_bwd64.h) when a 256-key workgroup is mostly real keys; the 32-key kernel otherwise
    // (its Q / dO / scratch descriptors span the whole tensors: 32-bit byte offsets)
    const int64_t q_ext = ((int64_t)(B - 1) * q_sb + (int64_t)(H - 1) * q_sh + (int64_t)(Nq - 1) * q_sn + 64) * 2;
    const int64_t o_ext = ((int64_t)(B - 1) * o_sb + (int64_t)(H - 1) * o_sh + (int64_t)(Nq - 1) * o_sn + 64) * 2;
    const bool fits32 = q_ext < (int64_t)0xffffffffll && o_ext < (int64_t)0xffffffffll && (int64_t)B * H * 2 * p.nq_pad * 4 < (int64_t)0xffffffffll;
    // query rows past Nq of a (batch, head) are read through those descriptors and must be FINITE (their P is exp2(-1e30 + s) = 0): true
    // when the next batch's rows follow directly (or the tensor ends: zero fill), not when a strided view leaves a gap of foreign memory
    // between batches — such views take the 32-key kernel, whose loads are bounded per (batch, head)
    const bool rows_follow = B == 1 || (q_sb == (int64_t)Nq * q_sn && o_sb == (int64_t)Nq * o_sn);
    const bool use64 = Nq > 64 && fits32 && rows_follow && (k64 == 2 || (k64 == 1 && Nk >= 192 && ((Nk + 255) / 256) * 256 * 3 <= Nk * 4));
    if (use64) {      // persistent: one workgroup per CU, workgroup g on XCD g % 8
        const int64_t items = (int64_t)((Nk + 255) / 256) * H * B;
        const int grid = (int)min((int64_t)(uc_num_cus() / 8 * 8), (items + 7) / 8 * 8);
        hipLaunchKernelGGL(attn_bwd_dkv64_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
    }
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3((unsigned)(((Nk + 127) / 128) * H * B)), dim3(256), 0, st, p);
    UC_CHECK_LAUNCH("uc_attention_bwd");
    return UC_OK;
}

extern "C" int uc_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                void* dQ, void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int64_t q_sb,
                                int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                                int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh,
                                int64_t dk_sb, int64_t dk_sn, int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale,
                                const int64_t* rope_qpos, const int64_t* rope_kpos, float rope_base, float rope_f0, uc_stream_t stream) {
    return attention_bwd_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Nq, Nk, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh,
                              o_sb, o_sn, o_sh, dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh, scale, rope_qpos, rope_kpos,
                              rope_base, rope_f0, uc_make_dropout(0.f, 0ull), stream);
}

// Backward of uc_attention_fwd_drop: uc_attention_bwd's arguments + the forward's (drop_p, seed).  O is the forward's (dropped) output,
// LSE that of the undropped scores.
extern "C" int uc_attention_bwd_drop(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                     void* dQ, void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int64_t q_sb,
                                     int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                                     int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh,
                                     int64_t dk_sb, int64_t dk_sn, int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale,
                                     const int64_t* rope_qpos, const int64_t* rope_kpos, float rope_base, float rope_f0, float drop_p,
                                     unsigned long long seed, uc_stream_t stream) {
    UC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "uc_attention_bwd_drop: drop_p must be in [0, 1) (got %g)", (double)drop_p);
    return attention_bwd_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Nq, Nk, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh,
                              o_sb, o_sn, o_sh, dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh, scale, rope_qpos, rope_kpos,
                              rope_base, rope_f0, uc_make_dropout(drop_p, seed), stream);
}

// =================================================================================================================
// fp32 verification-mode backward (head_dim <= 64): one thread per query row (dQ) / per key row (dK, dV), the other
// side streamed through LDS in 32-row tiles; plain fp32 FMA chains and expf, no atomics.
// =================================================================================================================
struct AttnBwdF32Params {
    const float *Q, *K, *V, *O, *dO, *LSE;
    float *dQ, *dK, *dV, *delta;
    int B, H, Nq, Nk, D;
    int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
    int64_t dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh;
    float scale;
    UcDropout drop;      // thr 0: none (a run-time test in these verification kernels)
};

#define BF_T 32

__global__ void attn_delta_f32_kernel(AttnBwdF32Params p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.H * p.Nq;
    if (idx >= total) return;
    const int q = (int)(idx % p.Nq);
    const int h = (int)((idx / p.Nq) % p.H);
    const int b = (int)(idx / ((int64_t)p.Nq * p.H));
    const float* o = p.O + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
    const float* d = p.dO + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
    float s = 0.f;
    for (int c = 0; c < p.D; ++c) s = fmaf(o[c], d[c], s);
    p.delta[idx] = s;
}

template <int DMAX>
__global__ __launch_bounds__(128) void attn_bwd_dq_f32_kernel(AttnBwdF32Params p) {
    __shared__ float Ks[BF_T][DMAX];
    __shared__ float Vs[BF_T][DMAX];
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * 128 + threadIdx.x;
    const int D = p.D;
    const bool active = q < p.Nq;
    const int qc = active ? q : p.Nq - 1;
    const float* qp = p.Q + (int64_t)b * p.q_sb + (int64_t)qc * p.q_sn + (int64_t)h * p.q_sh;
    const float* dp_ = p.dO + (int64_t)b * p.o_sb + (int64_t)qc * p.o_sn + (int64_t)h * p.o_sh;
    const float* Kb = p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const float* Vb = p.V + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    float qr[DMAX], dor[DMAX], acc[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        qr[d] = d < D ? qp[d] : 0.f;
        dor[d] = d < D ? dp_[d] : 0.f;
        acc[d] = 0.f;
    }
    const float lse = p.LSE[((int64_t)b * p.H + h) * p.Nq + qc];
    const float dlt = p.delta[((int64_t)b * p.H + h) * p.Nq + qc];
    unsigned dk1 = 0, dk2 = 0;
    if (p.drop.thr) uc_drop_keys(p.drop, (unsigned)(b * p.H + h), dk1, dk2);
    for (int k0 = 0; k0 < p.Nk; k0 += BF_T) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < BF_T * DMAX; idx += blockDim.x) {
            const int kk = idx / DMAX, d = idx % DMAX;
            const bool ok = (k0 + kk < p.Nk) && d < D;
            Ks[kk][d] = ok ? Kb[(int64_t)(k0 + kk) * p.k_sn + d] : 0.f;
            Vs[kk][d] = ok ? Vb[(int64_t)(k0 + kk) * p.v_sn + d] : 0.f;
        }
        __syncthreads();
        for (int kk = 0; kk < BF_T; ++kk) {
            if (k0 + kk >= p.Nk) break;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                s = fmaf(qr[d], Ks[kk][d], s);
                dp = fmaf(dor[d], Vs[kk][d], dp);
            }
            const float pw = expf(s * p.scale - lse);
            if (p.drop.thr) dp *= uc_drop_hash(dk1, dk2, (unsigned)qc, (unsigned)(k0 + kk)) >= p.drop.thr ? p.drop.keep_scale : 0.f;   // dP = dP' o mask / (1 - p)
            const float ds = pw * (dp - dlt);
#pragma unroll
            for (int d = 0; d < DMAX; ++d) acc[d] = fmaf(ds, Ks[kk][d], acc[d]);
        }
    }
    if (active) {
        float* op = p.dQ + (int64_t)b * p.dq_sb + (int64_t)q * p.dq_sn + (int64_t)h * p.dq_sh;
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            if (d < D) op[d] = acc[d] * p.scale;
    }
}

template <int DMAX>
__global__ __launch_bounds__(128) void attn_bwd_dkv_f32_kernel(AttnBwdF32Params p) {
    __shared__ float Qs[BF_T][DMAX];
    __shared__ float Ds[BF_T][DMAX];
    __shared__ float Ls[BF_T], Dl[BF_T];
    const int b = blockIdx.z, h = blockIdx.y;
    const int key = blockIdx.x * 128 + threadIdx.x;
    const int D = p.D;
    const bool active = key < p.Nk;
    const int kc = active ? key : p.Nk - 1;
    const float* kp = p.K + (int64_t)b * p.k_sb + (int64_t)kc * p.k_sn + (int64_t)h * p.k_sh;
    const float* vp = p.V + (int64_t)b * p.v_sb + (int64_t)kc * p.v_sn + (int64_t)h * p.v_sh;
    const float* Qb = p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* dOb = p.dO + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
    float kr[DMAX], vr[DMAX], dk[DMAX], dv[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        kr[d] = d < D ? kp[d] : 0.f;
        vr[d] = d < D ? vp[d] : 0.f;
        dk[d] = 0.f;
        dv[d] = 0.f;
    }
    unsigned dk1 = 0, dk2 = 0;
    if (p.drop.thr) uc_drop_keys(p.drop, (unsigned)(b * p.H + h), dk1, dk2);
    for (int q0 = 0; q0 < p.Nq; q0 += BF_T) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < BF_T * DMAX; idx += blockDim.x) {
            const int qq = idx / DMAX, d = idx % DMAX;
            const bool ok = (q0 + qq < p.Nq) && d < D;
            Qs[qq][d] = ok ? Qb[(int64_t)(q0 + qq) * p.q_sn + d] : 0.f;
            Ds[qq][d] = ok ? dOb[(int64_t)(q0 + qq) * p.o_sn + d] : 0.f;
        }
        if (threadIdx.x < BF_T) {
            const int qq = q0 + threadIdx.x;
            Ls[threadIdx.x] = qq < p.Nq ? p.LSE[((int64_t)b * p.H + h) * p.Nq + qq] : 0.f;
            Dl[threadIdx.x] = qq < p.Nq ? p.delta[((int64_t)b * p.H + h) * p.Nq + qq] : 0.f;
        }
        __syncthreads();
        for (int qq = 0; qq < BF_T; ++qq) {
            if (q0 + qq >= p.Nq) break;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                s = fmaf(Qs[qq][d], kr[d], s);
                dp = fmaf(Ds[qq][d], vr[d], dp);
            }
            const float pw = expf(s * p.scale - Ls[qq]);
            float pdrop = pw;                                     // P' = P o mask / (1 - p) for dV
            if (p.drop.thr) {
                const float m = uc_drop_hash(dk1, dk2, (unsigned)(q0 + qq), (unsigned)kc) >= p.drop.thr ? p.drop.keep_scale : 0.f;
                pdrop *= m;
                dp *= m;
            }
            const float ds = pw * (dp - Dl[qq]);
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                dv[d] = fmaf(pdrop, Ds[qq][d], dv[d]);
                dk[d] = fmaf(ds, Qs[qq][d], dk[d]);
            }
        }
    }
    if (active) {
        float* okp = p.dK + (int64_t)b * p.dk_sb + (int64_t)key * p.dk_sn + (int64_t)h * p.dk_sh;
        float* ovp = p.dV + (int64_t)b * p.dv_sb + (int64_t)key * p.dv_sn + (int64_t)h * p.dv_sh;
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            if (d < D) {
                okp[d] = dk[d] * p.scale;
                ovp[d] = dv[d];
            }
    }
}

static int attention_bwd_f32_impl(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                  const float* LSE, float* dQ, float* dK, float* dV, float* delta, int B, int H, int Nq,
                                  int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn,
                                  int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn,
                                  int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb, int64_t dk_sn,
                                  int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, const UcDropout& drop,
                                  uc_stream_t stream) {
    UC_REQUIRE(Q && K && V && O && dO && LSE && dQ && dK && dV && delta, "uc_attention_bwd_f32: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && D > 0 && D <= 64 && B <= 65535 && H <= 65535, "uc_attention_bwd_f32: bad shape (head_dim <= 64)");
    AttnBwdF32Params p;
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.dO = dO; p.LSE = LSE; p.dQ = dQ; p.dK = dK; p.dV = dV; p.delta = delta;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D;
    p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh; p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh;
    p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh; p.dq_sb = dq_sb; p.dq_sn = dq_sn; p.dq_sh = dq_sh; p.dk_sb = dk_sb; p.dk_sn = dk_sn;
    p.dk_sh = dk_sh; p.dv_sb = dv_sb; p.dv_sn = dv_sn; p.dv_sh = dv_sh; p.scale = scale;
    p.drop = drop;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * Nq;
    hipLaunchKernelGGL(attn_delta_f32_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, p);
    if (D <= 32) {
        hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<32>), dim3((Nq + 127) / 128, H, B), dim3(128), 0, st, p);
        hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<32>), dim3((Nk + 127) / 128, H, B), dim3(128), 0, st, p);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<64>), dim3((Nq + 127) / 128, H, B), dim3(128), 0, st, p);
        hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<64>), dim3((Nk + 127) / 128, H, B), dim3(128), 0, st, p);
    }
    UC_CHECK_LAUNCH("uc_attention_bwd_f32");
    return UC_OK;
}

extern "C" int uc_attention_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                    const float* LSE, float* dQ, float* dK, float* dV, float* delta, int B, int H, int Nq,
                                    int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn,
                                    int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn,
                                    int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb, int64_t dk_sn,
                                    int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, uc_stream_t stream) {
    return attention_bwd_f32_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Nq, Nk, D, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh,
                                  o_sb, o_sn, o_sh, dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh, scale,
                                  uc_make_dropout(0.f, 0ull), stream);
}

// fp32 backward of uc_attention_fwd_drop (verification mode): uc_attention_bwd_f32's arguments + the forward's (drop_p, seed)
extern "C" int uc_attention_bwd_f32_drop(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                         const float* LSE, float* dQ, float* dK, float* dV, float* delta, int B, int H, int Nq,
                                         int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn,
                                         int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn,
                                         int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, int64_t dk_sb, int64_t dk_sn,
                                         int64_t dk_sh, int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, float drop_p,
                                         unsigned long long seed, uc_stream_t stream) {
    UC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "uc_attention_bwd_f32_drop: drop_p must be in [0, 1) (got %g)", (double)drop_p);
    return attention_bwd_f32_impl(Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Nq, Nk, D, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh,
                                  o_sb, o_sn, o_sh, dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh, scale,
                                  uc_make_dropout(drop_p, seed), stream);
}
