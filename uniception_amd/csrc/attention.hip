// Fused scaled-dot-product attention (no mask, no dropout) for gfx950.
//
// attn_bf16_kernel — flash-style, head_dim 64, v_mfma_f32_32x32x16_bf16.
//   Workgroup = 4 wavefronts = 128 query rows of one (batch, head); each wave owns 32 queries.
//   Per 64-key tile:   S^T[key,q] = K . Q^T   (A = K rows from LDS, B = Q^T held in registers)
//   The swapped product puts a query's whole score column in ONE lane pair (lane, lane^32):
//   row max / row sum are 31 in-register ops + one cross-half exchange; the running max,
//   the rescale factor and the normaliser are lane-local scalars.
//   P^T (bf16) in the S^T accumulator layout is directly the B operand of  O^T[d,q] += V^T . P^T
//   when the key order inside each 16-key group is permuted the same way on the V side — which
//   is what the "VT" layout (uc_hip.h) bakes into memory, so the A operand V^T[d, 8 keys] is one
//   16-byte LDS read.  K and VT tiles are staged global -> VGPR -> LDS one tile ahead
//   (double-buffered, XOR-swizzled 128-byte rows, one barrier per tile).
//
// attn_f32_kernel — verification mode: one thread per query row, fp32 FMA chains, expf.
#include "common.h"
#include "knobs.h"
#include "attention_p64.h"
#include <stdlib.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct AttnParams {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    int B, H, Nq, Nk, D;
    int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
    int npad;     // VT row length
    float scale;
    float* lse;   // optional [B,H,Nq]: log-sum-exp of the scaled scores (saved for the backward)
    uc_fastdiv dGroup, dNq, dH;   // exact fast division by 8*nq, nq, H (workgroup -> (query tile, batch, head) in the DMA kernel)
    int prio_young;               // eight-wave workgroups: s_setprio 1 for waves 4-7 (the younger half loses every VALU arbitration to the older one)
    UcDropout drop;               // attention dropout (uc_attention_fwd_drop; thr 0 elsewhere)
};

#define KV_TILE 64
#define ATT_TILE_BYTES (KV_TILE * 128)  // 64 rows x 128 B

__device__ __forceinline__ int att_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// key offset inside a 16-key group for VT position pp (inverse of uc_vt_perm)
__device__ __forceinline__ int vt_key_of_pos(int pp) {
    const int hi = pp >> 3, j = pp & 7;
    return (j & 3) + 8 * (j >> 2) + 4 * hi;
}

// the body of attn_bf16_kernel for the 128 queries from q0w of (batch b, head h): also what attn_bf16_fixup_kernel recomputes a
// flagged block of the persistent kernel with (attention_p64.h)
// DROP: attention dropout (training with attn_drop > 0): the row sum — the softmax's normalisation — takes every probability, the
// P^T operand of O^T += V^T P^T only the kept ones, scaled by 1 / (1 - p): O = (P o mask / (1 - p)) V, the reference's
// attn_drop(softmax(...)) @ v.  The LSE is that of the undropped scores.
template <bool DROP = false>
__device__ __forceinline__ void attn_bf16_body(const AttnParams& p, const int b, const int h, const int q0w, char* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int q0 = q0w + wave * 32;

    const bf16_t* Qb = (const bf16_t*)p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* Kb = (const bf16_t*)p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* VTb = (const bf16_t*)p.V + ((int64_t)b * p.H + h) * 64 * (int64_t)p.npad;

    // ---- Q^T fragments (B operand): lane (q = l31, hi) holds Q[q][16s + 8hi .. +7], s = 0..3 ----
    bf16x8_t qf[4];
    {
        int q = q0 + l31;
        if (q >= p.Nq) q = p.Nq - 1;  // clamp; rows beyond Nq are never stored
        const bf16_t* qp = Qb + (int64_t)q * p.q_sn + hi * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * s);
    }

    // ---- staging assignment: 512 16-byte chunks per tile, 2 per thread ----
    const int cc = tid & 7;     // chunk column
    const int sr = tid >> 3;    // row 0..31 (+32)
    uint4 rk0, rk1, rv0, rv1;   // named (not arrays) so they stay in VGPRs
    auto load_k = [&](int k0, int row) -> uint4 {
        int key = k0 + row;
        if (key >= p.Nk) key = p.Nk - 1;  // clamped keys are masked in the softmax
        return *reinterpret_cast<const uint4*>(Kb + (int64_t)key * p.k_sn + cc * 8);
    };
    auto load_v = [&](int k0, int row) -> uint4 {
        // VT tile: row = channel d, 8 key positions [k0 + 8cc, +8)
        uint4 v = *reinterpret_cast<const uint4*>(VTb + (int64_t)row * p.npad + k0 + cc * 8);
        if (k0 + KV_TILE > p.Nk) {  // tail tile: zero positions whose key is out of range (0 * garbage guard)
            const int gbase = k0 + ((cc * 8) & ~15);
            const int pbase = (cc * 8) & 15;
            unsigned m0 = 0xffffffffu, m1 = 0xffffffffu, m2 = 0xffffffffu, m3 = 0xffffffffu;
            if (gbase + vt_key_of_pos(pbase + 0) >= p.Nk) m0 &= 0xffff0000u;
            if (gbase + vt_key_of_pos(pbase + 1) >= p.Nk) m0 &= 0x0000ffffu;
            if (gbase + vt_key_of_pos(pbase + 2) >= p.Nk) m1 &= 0xffff0000u;
            if (gbase + vt_key_of_pos(pbase + 3) >= p.Nk) m1 &= 0x0000ffffu;
            if (gbase + vt_key_of_pos(pbase + 4) >= p.Nk) m2 &= 0xffff0000u;
            if (gbase + vt_key_of_pos(pbase + 5) >= p.Nk) m2 &= 0x0000ffffu;
            if (gbase + vt_key_of_pos(pbase + 6) >= p.Nk) m3 &= 0xffff0000u;
            if (gbase + vt_key_of_pos(pbase + 7) >= p.Nk) m3 &= 0x0000ffffu;
            v.x &= m0; v.y &= m1; v.z &= m2; v.w &= m3;
        }
        return v;
    };
#define ATT_STAGE_LOAD(k0_)                 \
    do {                                    \
        rk0 = load_k((k0_), sr);            \
        rk1 = load_k((k0_), sr + 32);       \
        rv0 = load_v((k0_), sr);            \
        rv1 = load_v((k0_), sr + 32);       \
    } while (0)
    // loop-invariant LDS byte offsets: rows r and r+32 share the swizzle key, so one offset + an immediate serves both
    const int w_off = att_swz(sr, cc);
#define ATT_STAGE_WRITE(buf_)                                                          \
    do {                                                                               \
        char* sk_ = smem + (buf_) * 2 * ATT_TILE_BYTES + w_off;                        \
        *reinterpret_cast<uint4*>(sk_) = rk0;                                          \
        *reinterpret_cast<uint4*>(sk_ + 32 * 128) = rk1;                               \
        *reinterpret_cast<uint4*>(sk_ + ATT_TILE_BYTES) = rv0;                         \
        *reinterpret_cast<uint4*>(sk_ + ATT_TILE_BYTES + 32 * 128) = rv1;              \
    } while (0)
    int r_off[4];   // fragment read offsets: row l31 (+32 via immediate), chunk 2*st+hi, st = 0..3 (K and VT tiles alike)
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = att_swz(l31, 2 * st + hi);

    float16_t o[2];   // O^T accumulators: d-block x (16 regs): d = 32*db + (r&3) + 8*(r>>2) + 4*hi, q = l31
    o[0] = (float16_t)(0.f);
    o[1] = (float16_t)(0.f);
    float m_run = -1e30f;   // running max of raw scores (shared by the lane pair)
    float l_run = 0.f;      // lane-partial running sum
    const float c = p.scale * 1.44269504088896340736f;  // scale * log2(e)
    unsigned dk1 = 0, dk2 = 0;
    if constexpr (DROP) uc_drop_keys(p.drop, (unsigned)(b * p.H + h), dk1, dk2);

    const int nt = (p.Nk + KV_TILE - 1) / KV_TILE;
    ATT_STAGE_LOAD(0);
    ATT_STAGE_WRITE(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int k0 = t * KV_TILE;
        if (t + 1 < nt) ATT_STAGE_LOAD(k0 + KV_TILE);
        const char* sk = smem + buf * 2 * ATT_TILE_BYTES;
        const char* sv = sk + ATT_TILE_BYTES;

        // ---- S^T = K . Q^T : two 32-key blocks, 4 k-steps (16 channels) each ----
        float16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = (float16_t)(0.f);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + r_off[st] + kb * (32 * 128));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kb], 0, 0, 0);
            }
        }
        // ---- mask the tail ----
        if (k0 + KV_TILE > p.Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Nk) s[kb][r] = -1e30f;
                }
        }
        // ---- online softmax (per query = per lane pair) ----
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        // Deferred rescale: keep the old running max while no query of the wave grew by more than 2^RESCALE_LOG2 in the
        // exp2 domain (P stays <= 2^8, exact in the final O/l ratio); the O/l rescale then runs on a wave-uniform branch.
        const bool grow = (mt - m_run) * c > 8.0f;
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
        bf16x8_t pf[4];   // P^T B-operand fragments: slab g = 2*kb + half, 8 key slots per lane
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(fmaf(s[kb][hf * 8 + j], c, -mc));
                    psum += e[j];
                    if constexpr (DROP) {
                        const int r = hf * 8 + j;
                        const unsigned key = (unsigned)(k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
                        e[j] = uc_drop_hash(dk1, dk2, (unsigned)(q0 + l31), key) >= p.drop.thr ? e[j] * p.drop.keep_scale : 0.f;
                    }
                }
                union { bf16x8_t v; unsigned u[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf16x2(e[2 * j], e[2 * j + 1]);
                pf[kb * 2 + hf] = pk.v;
            }
        }
        l_run += psum;

        // ---- O^T += V^T . P^T : two 32-channel blocks x four 16-key slabs ----
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sv + r_off[g] + db * (32 * 128));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g], o[db], 0, 0, 0);
            }

        if (t + 1 < nt) ATT_STAGE_WRITE(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds 4 consecutive channels per (db, r>>2) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (p.lse && q < p.Nq && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run * p.scale + logf(l_tot);
    if (q < p.Nq) {
        bf16_t* op = (bf16_t*)p.O + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
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

__global__ __launch_bounds__(256) void attn_bf16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * ATT_TILE_BYTES];  // 2 stages x (K tile + VT tile)
    attn_bf16_body(p, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x * 128, smem);
}
__global__ __launch_bounds__(256) void attn_bf16_drop_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * ATT_TILE_BYTES];
    attn_bf16_body<true>(p, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x * 128, smem);
}

// ---------------------------------------------------------------------------------------
// attn_bf16_fixup_kernel — launched behind attn_bf16_p64_kernel: that kernel takes every exponential against the row maximum of an
// item's FIRST 32 keys and flags a 64-query block whose row sums left [2^-100, 2^100] (a score ~69 nats above / below that
// maximum: P or O may have over- / underflowed) with a sentinel in the block's first output word.  One thread per block reads
// that word; a flagged block is recomputed by the whole workgroup with the exact online softmax above (the 128 queries from the
// block's first one: the second half re-does an unflagged neighbour, or finds it flagged and is simply ahead of its own scan).
// Flags are rare by construction; the scan is one 4-byte load per 64 x 64 outputs.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bf16_fixup_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * ATT_TILE_BYTES];
    __shared__ int s_list[256];
    __shared__ int s_cnt;
    const int nblk = (p.Nq + 63) >> 6;
    const long long total = (long long)p.B * p.H * nblk;
    for (long long c0 = (long long)blockIdx.x * 256; c0 < total; c0 += (long long)gridDim.x * 256) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const long long id = c0 + threadIdx.x;
        if (id < total) {
            const int blk = (int)(id % nblk), bh = (int)(id / nblk), b = bh / p.H, h = bh - b * p.H;
            const unsigned* w = reinterpret_cast<const unsigned*>((const bf16_t*)p.O + (int64_t)b * p.o_sb + (int64_t)(blk * 64) * p.o_sn + (int64_t)h * p.o_sh);
            if (__builtin_nontemporal_load(w) == P64_SENTINEL) s_list[atomicAdd(&s_cnt, 1)] = (int)threadIdx.x;
        }
        __syncthreads();
        const int n = s_cnt;
        for (int i = 0; i < n; ++i) {
            const long long fid = c0 + s_list[i];
            const int blk = (int)(fid % nblk), bh = (int)(fid / nblk), b = bh / p.H, h = bh - b * p.H;
            __syncthreads();                              // (the body's LDS ring and s_list: the previous body is done)
            attn_bf16_body(p, b, h, blk * 64, smem);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// attn_bf16_dma_kernel — the same algorithm (the VT pad positions of a ragged last tile must hold zeros: uc_vt_pack and
// ops.vt_buffer guarantee it) with the K and
// VT tiles staged by buffer-addressed LDS-DMA instead of global -> VGPR -> LDS.  The register-staged kernel needs 184
// registers (152 VGPR + the score accumulators pushed into 32 AGPRs: 2 waves per SIMD, 64 v_accvgpr moves per key tile);
// without the 16 staging registers, the per-chunk tail masks of V and the per-tile address math this one fits the 128-register budget
// of 4 waves per SIMD.  Descriptors: K rows / VT rows of this (batch, head); the per-lane byte offsets are loop
// invariant, the tile advance is a wave-uniform soffset.
// ---------------------------------------------------------------------------------------
typedef unsigned att_uint4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void att_dma16(unsigned voff, att_uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
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
__device__ __forceinline__ att_uint4_t att_make_srd(const void* base) {
    const unsigned long long pa = (unsigned long long)base;
    return (att_uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)), 0xffffff00u, 0x00020000u};
}

// What bounds it (round 2, tools/scratch/probe_attn_anatomy.py + tools/pmc_attn.sh, B = 64, H = 16, N = 1024): the costs of the exp2
// work and of the PV products are ADDITIVE (-18 % without the exps, -20 % without the PV MFMAs, -2 % without the per-tile
// barrier; static priorities or a start stagger per workgroup slot: nothing).  SQ_ACTIVE_INST_VALU — which includes a matrix
// instruction for its whole 32 cycles — is 81-86 % of the wave-resident time of a SIMD: matrix and vector instructions of the
// co-resident waves take turns, they do not overlap.  Per key tile and wave that is 512 cycles of MFMA + ~350 of VALU (three
// quarters of it the 32 exp2): at head_dim 64 the matrix pipe cannot be more than ~60 % busy, the kernel has it at 48 %.
// A form software-pipelined INSIDE a wave (S(t+1) MFMAs issued between the exps of tile t, PV(t) per 16-key chunk as soon
// as its P exists, fragment reads two MFMAs ahead, separate K / VT rings; 160 registers, three waves per SIMD) was built,
// verified and measured the same per-SIMD time (412 vs 412 us): removed again.  Packed fp32 (v_pk_fma_f32 for the scale,
// v_pk_add_f32 for the row sums: 31 instructions instead of 63) measured 5-7 % SLOWER (394 -> 414-421 us): not kept.
// NW waves of 32 queries share every K / VT tile: NW = 8 (256 queries per workgroup) stages each tile once for twice the
// queries of NW = 4 — half the global->LDS traffic per query — at the same 16 waves per CU (two workgroups of 48 KiB).
template <int NW, int DBG = 0>
__global__ __launch_bounds__(NW * 64, 4) void attn_bf16_dma_kernel(AttnParams p) {
    // 2 stages x (K tile + VT tile); the Q rows of waves 4..7 (NW = 8) arrive in a third region
    __shared__ __attribute__((aligned(16))) char smem[(NW == 8 ? 6 : 4) * ATT_TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    // 1-D grid.  Workgroup w runs on XCD w % 8 (round-robin dispatch): the nq query tiles that share one (batch, head)'s
    // K/VT rows are given ids that differ by multiples of 8, so that they meet in ONE XCD's L2 instead of pulling the
    // same K/VT through all eight.
    constexpr int QT = 32 * NW;                 // queries per workgroup
    const int nq = (p.Nq + QT - 1) / QT;
    const int nbh = p.B * p.H;
    int qt, bh;
    {
        const int w = blockIdx.x;
        const int per_group = 8 * nq;
        const int grp = (int)uc_div((unsigned)w, p.dGroup), within = w - grp * per_group;
        if ((grp + 1) * 8 <= nbh) { bh = grp * 8 + (within & 7); qt = within >> 3; }
        else {   // last, partial group: plain order
            const int rem = w - (nbh >> 3) * 8 * nq, rb = (int)uc_div((unsigned)rem, p.dNq);
            bh = (nbh >> 3) * 8 + rb; qt = rem - rb * nq;
        }
    }
    const int b = (int)uc_div((unsigned)bh, p.dH), h = bh - b * p.H;
    const int q0 = qt * QT + wave * 32;
    if constexpr (DBG & 8) {    // experiment: static priority by (approximate) workgroup slot on the CU
        const unsigned slot = (blockIdx.x >> 8) & 3u;
        if (slot == 0) __builtin_amdgcn_s_setprio(0);
        else if (slot == 1) __builtin_amdgcn_s_setprio(1);
        else if (slot == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
    if constexpr (DBG & 16) {   // experiment: de-phase the first round of workgroups by a quarter of a key tile each
        if (blockIdx.x < 1024u) {
            const unsigned slot = (blockIdx.x >> 8) & 3u;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)slot * 15u) __builtin_amdgcn_s_sleep(4);
        }
    }

    const bf16_t* Qb = (const bf16_t*)p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* Kb = (const bf16_t*)p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* VTb = (const bf16_t*)p.V + ((int64_t)b * p.H + h) * 64 * (int64_t)p.npad;

    // ---- DMA assignment: a tile is 8 instructions of 8 rows x 128 B; wave w issues instructions PW*w .. PW*w + PW - 1 of both tiles ----
    constexpr int PW = 8 / NW;                  // 2 (four waves) or 1 (eight waves)
    att_uint4_t srd_k = att_make_srd(Kb);
    srd_k.z = (unsigned)__builtin_amdgcn_readfirstlane((int)((((int64_t)p.Nk - 1) * p.k_sn + 64) * 2));   // key rows >= Nk read as zeros
    const att_uint4_t srd_v = att_make_srd(VTb);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    unsigned voff_k[PW], voff_v[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int rr = (wave * PW + i) * 8 + (lane >> 3);
        const int cch = (lane & 7) ^ ((rr >> 1) & 7);         // logical chunk stored at physical chunk lane&7 of row rr
        voff_k[i] = (unsigned)(((int64_t)rr * p.k_sn + cch * 8) * 2);
        voff_v[i] = (unsigned)(((int64_t)rr * p.npad + cch * 8) * 2);
    }
    const unsigned kstep = (unsigned)(KV_TILE * p.k_sn * 2);   // bytes between key tiles of K
    auto issue_tile = [&](int t, int buf) {
        const unsigned dst = lds0 + (unsigned)(buf * 2 * ATT_TILE_BYTES + wave * (PW * 1024));
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            att_dma16(voff_k[i], srd_k, (unsigned)t * kstep, __builtin_amdgcn_readfirstlane(dst + i * 1024));
            att_dma16(voff_v[i], srd_v, (unsigned)t * (KV_TILE * 2), __builtin_amdgcn_readfirstlane(dst + ATT_TILE_BYTES + i * 1024));
        }
    };

    // ---- Q: the wave's 32 query rows (128 B each) arrive by DMA too — 4 instructions of 8 whole rows instead of 4 loads
    //      that touch 32 rows x 32 B each (the vector memory path pays per row segment) — into the second stage buffer,
    //      which the K/VT ring only needs from tile 1 on.  Rows beyond Nq fall outside the descriptor: zeros. ----
    {
        const unsigned long long qa = (unsigned long long)(Qb + (int64_t)q0 * p.q_sn);
        const int64_t q_rows = min((int64_t)32, (int64_t)p.Nq - q0);
        att_uint4_t srd_q = att_make_srd((const void*)qa);
        srd_q.z = (unsigned)__builtin_amdgcn_readfirstlane((int)(q_rows > 0 ? ((q_rows - 1) * p.q_sn + 64) * 2 : 0));
        const unsigned dstq = lds0 + (unsigned)(2 * ATT_TILE_BYTES + wave * 4096);     // waves 4..7 land behind the second stage
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const int cch = (lane & 7) ^ ((rr >> 1) & 7);
            att_dma16((unsigned)(((int64_t)rr * p.q_sn + cch * 8) * 2), srd_q, 0u, __builtin_amdgcn_readfirstlane(dstq + i * 1024));
        }
    }
    int r_off[4];   // fragment read offsets: row l31 (+32 via immediate), chunk 2*st+hi, st = 0..3 (K and VT tiles alike)
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = att_swz(l31, 2 * st + hi);

    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Q^T fragments (B operand): lane (q = l31, hi) holds Q[q][16s + 8hi .. +7], s = 0..3
    bf16x8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(smem + 2 * ATT_TILE_BYTES + wave * 4096 + r_off[s]);
    __syncthreads();     // every wave holds its Q fragments before tile 1 overwrites the buffer

    float16_t o[2];
    o[0] = (float16_t)(0.f);
    o[1] = (float16_t)(0.f);
    float m_run = -1e30f;
    float l_run = 0.f;
    const float c = p.scale * 1.44269504088896340736f;  // scale * log2(e)
    if constexpr (NW == 8) {
        // static priority for the second-dispatched half (MI355X_MICROARCH.md "Two waves per SIMD", item 4): one s_setprio before the
        // loop, no per-segment flips (`wave` is a readfirstlane value: the branch is scalar, s_setprio ignores EXEC)
        if (p.prio_young && wave >= 4) __builtin_amdgcn_s_setprio(1);
    }

    const int nt = (p.Nk + KV_TILE - 1) / KV_TILE;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
        const char* sk = smem + buf * 2 * ATT_TILE_BYTES;
        const char* sv = sk + ATT_TILE_BYTES;

        float16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = (float16_t)(0.f);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + r_off[st] + kb * (32 * 128));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kb], 0, 0, 0);
            }
        }
        if (t == nt - 1 && (p.Nk & (KV_TILE - 1))) {    // ragged last tile: keys >= Nk (zero K rows, zero VT pads) leave the softmax
            const int k0 = t * KV_TILE;
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
        const bool grow = (mt - m_run) * c > 8.0f;     // deferred rescale, see attn_bf16_kernel
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
        bf16x8_t pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = (DBG & 2) ? s[kb][hf * 8 + j] : __builtin_amdgcn_exp2f(fmaf(s[kb][hf * 8 + j], c, -mc));
                    psum += e[j];
                }
                union { bf16x8_t v; unsigned u[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf16x2(e[2 * j], e[2 * j + 1]);
                pf[kb * 2 + hf] = pk.v;
            }
        }
        l_run += psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sv + r_off[g] + db * (32 * 128));
                if constexpr (!(DBG & 4)) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g], o[db], 0, 0, 0);
                else o[db][g] += (float)vf[0] * (float)pf[g][0];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t+1 have landed
        if constexpr (!(DBG & 1)) __syncthreads();                    // ... everyone's; and every wave is done reading tile t
    }

    // ---- normalise; bounce the wave's 32 x 64 outputs through its private 4 KiB of the (now free) ring so that every
    //      store instruction writes 8 whole 128-B rows instead of 32 rows x 16 B ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l_tot);
    const int q = q0 + l31;
    if (p.lse && q < p.Nq && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run * p.scale + logf(l_tot);
    char* ob = smem + wave * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint2 pk;       // channels 32db + 8g4 + 4hi .. +3 of query l31 = half of 16-B chunk 4db + g4
            pk.x = pack_bf16x2(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv);
            pk.y = pack_bf16x2(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(ob + l31 * 128 + (((4 * db + g4) ^ (l31 & 7)) << 4) + hi * 8) = pk;
        }
    bf16_t* obase = (bf16_t*)p.O + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int R = 8 * ps + (lane >> 3);
        const int chunk = (lane & 7) ^ (R & 7);
        const uint4 v = *reinterpret_cast<const uint4*>(ob + R * 128 + ((lane & 7) << 4));
        if (q0 + R < p.Nq) *reinterpret_cast<uint4*>(obase + (int64_t)(q0 + R) * p.o_sn + chunk * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------
// attn_bf16_rs_kernel — the same arithmetic as attn_bf16_dma_kernel<8>, ROLE-SPLIT (round 3): the loop is cut into segments that
// are either matrix work (PV of the previous key tile + QK^T of the next one: 16 MFMAs and their fragment reads) or vector work
// (the softmax of one tile: max, exp2, sums, conversions), every segment ends in a workgroup barrier, and the two halves of the
// workgroup run the SAME segment sequence one segment apart — while waves 0-3 are in a matrix segment, waves 4-7 (their partners
// on the four SIMDs: a workgroup's waves go to the SIMDs cyclically) are in a vector segment, and vice versa.
//
// Why: with every wave of a workgroup in the same phase (one barrier per tile), the matrix pipe of a SIMD sees both of its waves'
// MFMAs together and then neither — round 2 measured the exp2 work (-18 %) and the PV products (-20 %) as ADDITIVE costs and the
// pipe 41-48 % busy.  Here a SIMD's two waves of a workgroup alternate by construction (MI355X_MICROARCH.md "Two waves per SIMD":
// the pairing that nets is matrix beside vector); per tile and wave the matrix segment is 512 cycles and the vector segment ~350.
//
// Segment s = 0 .. 2 nt + 1 (nt key tiles); half h (0: waves 0-3, 1: waves 4-7) executes local segment ls = s - h:
//     ls = 2 t      matrix:  O += V(t-1) P(t-1)   (t >= 1),   S = K(t) Q^T   (t < nt)
//     ls = 2 t + 1  vector:  softmax of tile t: running max (deferred rescale), P(t) = exp2(..) as bf16 fragments, row sums
// LDS ring as before (two K buffers, two VT buffers): K(t) is read in segments 2t, 2t+1 and VT(t) in 2t+2, 2t+3; K(t+1) and VT(t)
// are issued at the start of segment 2t (their buffers were last read in segment 2t-1) and waited for before the barrier that
// ends segment 2t+1.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 4) void attn_bf16_rs_kernel(AttnParams p) {
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) char smem[6 * ATT_TILE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    constexpr int QT = 32 * NW;
    const int nq = (p.Nq + QT - 1) / QT;
    const int nbh = p.B * p.H;
    int qt, bh;
    {
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

    const bf16_t* Qb = (const bf16_t*)p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const bf16_t* Kb = (const bf16_t*)p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const bf16_t* VTb = (const bf16_t*)p.V + ((int64_t)b * p.H + h) * 64 * (int64_t)p.npad;

    // DMA: a tile is 8 instructions of 8 rows x 128 B; wave w issues instruction w of the K tile and of the VT tile
    att_uint4_t srd_k = att_make_srd(Kb);
    srd_k.z = (unsigned)__builtin_amdgcn_readfirstlane((int)((((int64_t)p.Nk - 1) * p.k_sn + 64) * 2));   // key rows >= Nk read as zeros
    const att_uint4_t srd_v = att_make_srd(VTb);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    unsigned voff_k, voff_v;
    {
        const int rr = wave * 8 + (lane >> 3);
        const int cch = (lane & 7) ^ ((rr >> 1) & 7);
        voff_k = (unsigned)(((int64_t)rr * p.k_sn + cch * 8) * 2);
        voff_v = (unsigned)(((int64_t)rr * p.npad + cch * 8) * 2);
    }
    const unsigned kstep = (unsigned)(KV_TILE * p.k_sn * 2);
    auto issue_k = [&](int t) {
        att_dma16(voff_k, srd_k, (unsigned)t * kstep, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((t & 1) * 2 * ATT_TILE_BYTES + wave * 1024)));
    };
    auto issue_v = [&](int t) {
        att_dma16(voff_v, srd_v, (unsigned)t * (KV_TILE * 2),
                  __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((t & 1) * 2 * ATT_TILE_BYTES + ATT_TILE_BYTES + wave * 1024)));
    };
    {   // Q rows of this wave into the second stage (waves 0-3) / the third region (waves 4-7)
        const unsigned long long qa = (unsigned long long)(Qb + (int64_t)q0 * p.q_sn);
        const int64_t q_rows = min((int64_t)32, (int64_t)p.Nq - q0);
        att_uint4_t srd_q = att_make_srd((const void*)qa);
        srd_q.z = (unsigned)__builtin_amdgcn_readfirstlane((int)(q_rows > 0 ? ((q_rows - 1) * p.q_sn + 64) * 2 : 0));
        const unsigned dstq = lds0 + (unsigned)(2 * ATT_TILE_BYTES + wave * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const int cch = (lane & 7) ^ ((rr >> 1) & 7);
            att_dma16((unsigned)(((int64_t)rr * p.q_sn + cch * 8) * 2), srd_q, 0u, __builtin_amdgcn_readfirstlane(dstq + i * 1024));
        }
    }
    int r_off[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) r_off[st] = att_swz(l31, 2 * st + hi);

    issue_k(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(smem + 2 * ATT_TILE_BYTES + wave * 4096 + r_off[s]);
    __syncthreads();     // every wave holds its Q fragments before K(1) / VT(1) overwrite the buffer

    float16_t o[2], sc[2];
    o[0] = (float16_t)(0.f);
    o[1] = (float16_t)(0.f);
    sc[0] = (float16_t)(0.f);
    sc[1] = (float16_t)(0.f);
    bf16x8_t pf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) pf[g] = (bf16x8_t)(0.f);
    float m_run = -1e30f;
    float l_run = 0.f;
    const float c = p.scale * 1.44269504088896340736f;
    const int nt = (p.Nk + KV_TILE - 1) / KV_TILE;
    auto issue = [&](int t) __attribute__((always_inline)) {       // start of even segment 2t: VT(t) and K(t+1) set out
        if (t < nt) issue_v(t);
        if (t + 1 < nt) issue_k(t + 1);
    };
    auto matrix = [&](int t) __attribute__((always_inline)) {      // O += VT(t-1) P(t-1); S = K(t) Q^T
        if (t >= 1) {
            const char* sv = smem + ((t - 1) & 1) * 2 * ATT_TILE_BYTES + ATT_TILE_BYTES;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sv + r_off[g] + db * (32 * 128));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g], o[db], 0, 0, 0);
                }
        }
        if (t < nt) {
            const char* sk = smem + (t & 1) * 2 * ATT_TILE_BYTES;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                sc[kb] = (float16_t)(0.f);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + r_off[st] + kb * (32 * 128));
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], sc[kb], 0, 0, 0);
                }
            }
        }
    };
    auto vector = [&](int t) __attribute__((always_inline)) {      // softmax of tile t: sc -> pf
        if (t == nt - 1 && (p.Nk & (KV_TILE - 1))) {
            const int k0 = t * KV_TILE;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) sc[kb][r] = -1e30f;
        }
        float mt = sc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sc[kb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const bool grow = (mt - m_run) * c > 8.0f;     // deferred rescale, see attn_bf16_kernel (PV of tile t-1 is complete)
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
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(fmaf(sc[kb][hf * 8 + j], c, -mc));
                    psum += e[j];
                }
                union { bf16x8_t v; unsigned u[4]; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.u[j] = pack_bf16x2(e[2 * j], e[2 * j + 1]);
                pf[kb * 2 + hf] = pk.v;
            }
        }
        l_run += psum;
    };
    // two loops with the same barrier sequence (2 nt + 2 segments), one segment apart: a single loop with a role switch keeps the
    // union of both roles' registers alive across its back edge (468 bytes of scratch at the 128-register budget)
    const bool seg_prio = p.prio_young == 2;     // experiment: priority 1 while in a matrix segment (cdna guide T5)
    if (half == 0) {
        for (int t = 0; t <= nt; ++t) {
            issue(t);
            if (seg_prio) __builtin_amdgcn_s_setprio(1);
            matrix(t);                                            // segment 2t
            if (seg_prio) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (t < nt) vector(t);                                // segment 2t + 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // what was issued at the start of segment 2t has landed
            __syncthreads();
        }
    } else {
        for (int t = 0; t <= nt; ++t) {
            issue(t);
            if (t >= 1) vector(t - 1);                            // segment 2t
            __syncthreads();
            if (seg_prio) __builtin_amdgcn_s_setprio(1);
            matrix(t);                                            // segment 2t + 1
            if (seg_prio) __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l_tot);
    const int q = q0 + l31;
    if (p.lse && q < p.Nq && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run * p.scale + logf(l_tot);
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
    bf16_t* obase = (bf16_t*)p.O + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int R = 8 * ps + (lane >> 3);
        const int chunk = (lane & 7) ^ (R & 7);
        const uint4 v = *reinterpret_cast<const uint4*>(ob + R * 128 + ((lane & 7) << 4));
        if (q0 + R < p.Nq) *reinterpret_cast<uint4*>(obase + (int64_t)(q0 + R) * p.o_sn + chunk * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------
// row-major V -> VT packing (for callers that did not get VT from the GEMM epilogue)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vt_pack_kernel(const bf16_t* __restrict__ V, bf16_t* __restrict__ VT, int H,
                                                      int Nk, int D, int npad, int64_t v_sb, int64_t v_sn,
                                                      int64_t v_sh) {
    // block: (key tile of 64, head, batch); LDS transpose of a [64 keys][D] tile
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);  // [64][D+2]
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
    const int ld = D + 2;
    const bf16_t* vb = V + (int64_t)b * v_sb + (int64_t)h * v_sh;
    for (int idx = threadIdx.x; idx < 64 * D; idx += blockDim.x) {
        const int key = idx / D, d = idx % D;
        tile[key * ld + d] = (k0 + key < Nk) ? vb[(int64_t)(k0 + key) * v_sn + d] : (bf16_t)0;
    }
    __syncthreads();
    bf16_t* out = VT + ((int64_t)b * H + h) * D * (int64_t)npad + k0;
    for (int idx = threadIdx.x; idx < 64 * D; idx += blockDim.x) {
        const int pos = idx & 63, d = idx >> 6;
        const int key = (pos & ~15) + vt_key_of_pos(pos & 15);
        out[(int64_t)d * npad + pos] = tile[key * ld + d];
    }
}

// D == 64, 16-byte accesses on both sides: 8-channel chunks in, 8-position chunks out (the generic kernel above moves single
// bf16 values and divides per element)
__global__ __launch_bounds__(256) void vt_pack64_kernel(const bf16_t* __restrict__ V, bf16_t* __restrict__ VT, int H, int Nk,
                                                        int npad, int64_t v_sb, int64_t v_sn, int64_t v_sh) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][64 + 8];   // [key][d], 144-byte rows
    const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 64;
    const bf16_t* vb = V + (int64_t)b * v_sb + (int64_t)h * v_sh;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = threadIdx.x + it * 256;          // 512 chunks: key = idx >> 3, 8 channels from (idx & 7) * 8
        const int key = idx >> 3, c8 = idx & 7;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (k0 + key < Nk) v = *reinterpret_cast<const uint4*>(vb + (int64_t)(k0 + key) * v_sn + c8 * 8);
        *reinterpret_cast<uint4*>(&tile[key][c8 * 8]) = v;
    }
    __syncthreads();
    bf16_t* out = VT + ((int64_t)b * H + h) * 64 * (int64_t)npad + k0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = threadIdx.x + it * 256;          // 512 chunks: channel d = idx >> 3, positions (idx & 7) * 8 .. +7
        const int d = idx >> 3, p8 = (idx & 7) * 8;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = p8 + j;
            e[j] = tile[(pos & ~15) + vt_key_of_pos(pos & 15)][d];
        }
        uint4 o;
        o.x = (unsigned)e[0] | ((unsigned)e[1] << 16); o.y = (unsigned)e[2] | ((unsigned)e[3] << 16);
        o.z = (unsigned)e[4] | ((unsigned)e[5] << 16); o.w = (unsigned)e[6] | ((unsigned)e[7] << 16);
        *reinterpret_cast<uint4*>(out + (int64_t)d * npad + p8) = o;
    }
}

extern "C" int uc_vt_pack(const void* V, void* VT, int B, int H, int Nk, int D, int64_t v_sb, int64_t v_sn,
                          int64_t v_sh, uc_stream_t stream) {
    UC_REQUIRE(V && VT, "uc_vt_pack: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nk > 0 && D > 0 && D <= 256, "uc_vt_pack: bad shape");
    const int npad = (Nk + 63) / 64 * 64;
    if (D == 64 && v_sb % 8 == 0 && v_sn % 8 == 0 && v_sh % 8 == 0 && (uintptr_t)V % 16 == 0 && (uintptr_t)VT % 16 == 0)
        hipLaunchKernelGGL(vt_pack64_kernel, dim3(npad / 64, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)V, (bf16_t*)VT, H, Nk,
                           npad, v_sb, v_sn, v_sh);
    else
        hipLaunchKernelGGL(vt_pack_kernel, dim3(npad / 64, H, B), dim3(256), (size_t)64 * (D + 2) * 2, (hipStream_t)stream,
                           (const bf16_t*)V, (bf16_t*)VT, H, Nk, D, npad, v_sb, v_sn, v_sh);
    UC_CHECK_LAUNCH("uc_vt_pack");
    return UC_OK;
}

// ---------------------------------------------------------------------------------------
// fp32 verification kernel: one thread per query, 32-key tiles of K and V in LDS (broadcast reads)
// ---------------------------------------------------------------------------------------
#define F32_KT 32

template <int DMAX, bool DROP = false>
__global__ __launch_bounds__(128) void attn_f32_kernel(AttnParams p) {
    __shared__ float Ks[F32_KT][DMAX];
    __shared__ float Vs[F32_KT][DMAX];
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * 128 + threadIdx.x;
    const int D = p.D;
    const float* Qb = (const float*)p.Q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    const float* Kb = (const float*)p.K + (int64_t)b * p.k_sb + (int64_t)h * p.k_sh;
    const float* Vb = (const float*)p.V + (int64_t)b * p.v_sb + (int64_t)h * p.v_sh;
    const bool active = q < p.Nq;
    float qr[DMAX], acc[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        qr[d] = (active && d < D) ? Qb[(int64_t)q * p.q_sn + d] : 0.f;
        acc[d] = 0.f;
    }
    float m_run = -INFINITY, l_run = 0.f;
    unsigned dk1 = 0, dk2 = 0;
    if constexpr (DROP) uc_drop_keys(p.drop, (unsigned)(b * p.H + h), dk1, dk2);
    for (int k0 = 0; k0 < p.Nk; k0 += F32_KT) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < F32_KT * DMAX; idx += blockDim.x) {
            const int kk = idx / DMAX, d = idx % DMAX;
            const bool ok = (k0 + kk < p.Nk) && d < D;
            Ks[kk][d] = ok ? Kb[(int64_t)(k0 + kk) * p.k_sn + d] : 0.f;
            Vs[kk][d] = ok ? Vb[(int64_t)(k0 + kk) * p.v_sn + d] : 0.f;
        }
        __syncthreads();
        float s[F32_KT];
        float mt = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < F32_KT; ++kk) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) a = fmaf(qr[d], Ks[kk][d], a);
            a *= p.scale;
            s[kk] = (k0 + kk < p.Nk) ? a : -INFINITY;
            mt = fmaxf(mt, s[kk]);
        }
        const float m_new = fmaxf(m_run, mt);
        const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) acc[d] *= alpha;
#pragma unroll
        for (int kk = 0; kk < F32_KT; ++kk) {
            float pw = (k0 + kk < p.Nk) ? expf(s[kk] - m_new) : 0.f;
            l_run += pw;
            if constexpr (DROP) pw = uc_drop_hash(dk1, dk2, (unsigned)q, (unsigned)(k0 + kk)) >= p.drop.thr ? pw * p.drop.keep_scale : 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) acc[d] = fmaf(pw, Vs[kk][d], acc[d]);
        }
        m_run = m_new;
    }
    if (active) {
        if (p.lse) p.lse[((int64_t)b * p.H + h) * p.Nq + q] = m_run + logf(l_run);
        float* op = (float*)p.O + (int64_t)b * p.o_sb + (int64_t)q * p.o_sn + (int64_t)h * p.o_sh;
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            if (d < D) op[d] = acc[d] * inv;
    }
}

// keep mask of uc_attention_fwd_drop / uc_attention_bwd_drop as bytes [B, H, Nq, Nk] (1 = kept): what a reference implementation
// multiplies the probabilities with (tests), evaluated by the same function as the kernels
__global__ void attn_drop_mask_kernel(unsigned char* mask, int H, int Nq, int Nk, int64_t total, UcDropout d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % Nk);
    const int q = (int)((idx / Nk) % Nq);
    const unsigned bh = (unsigned)(idx / ((int64_t)Nk * Nq));
    unsigned k1, k2;
    uc_drop_keys(d, bh, k1, k2);
    mask[idx] = uc_drop_hash(k1, k2, (unsigned)q, (unsigned)k) >= d.thr ? 1 : 0;
}

extern "C" int uc_attention_drop_mask(void* mask, int B, int H, int Nq, int Nk, float drop_p, unsigned long long seed, uc_stream_t stream) {
    UC_REQUIRE(mask && B > 0 && H > 0 && Nq > 0 && Nk > 0, "uc_attention_drop_mask: bad arguments");
    UC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "uc_attention_drop_mask: drop_p must be in [0, 1) (got %g)", (double)drop_p);
    const int64_t total = (int64_t)B * H * Nq * Nk;
    hipLaunchKernelGGL(attn_drop_mask_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned char*)mask, H, Nq, Nk, total, uc_make_dropout(drop_p, seed));
    UC_CHECK_LAUNCH("uc_attention_drop_mask");
    return UC_OK;
}

// Attention forward with dropout of the probabilities (training, attn_drop > 0): the argument list of uc_attention_fwd + (drop_p, seed).
// bf16: the register-staged 128-query kernel with the mask applied between the softmax and the second product; fp32: the verification
// kernel.  drop_p == 0 is uc_attention_fwd.
extern "C" int uc_attention_fwd_drop(const void* Q, const void* K, const void* V, void* O, int dtype, int v_layout,
                                     int B, int H, int Nq, int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh,
                                     int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                                     int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, float* lse, float drop_p,
                                     unsigned long long seed, uc_stream_t stream) {
    UC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "uc_attention_fwd_drop: drop_p must be in [0, 1) (got %g)", (double)drop_p);
    if (drop_p == 0.f)
        return uc_attention_fwd(Q, K, V, O, dtype, v_layout, B, H, Nq, Nk, D, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh,
                                scale, lse, stream);
    UC_REQUIRE(Q && K && V && O, "uc_attention_fwd_drop: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && D > 0, "uc_attention_fwd_drop: bad shape");
    UC_REQUIRE(H <= 65535 && B <= 65535, "uc_attention_fwd_drop: B and H must fit a grid dimension");
    AttnParams p;
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D;
    p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
    p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
    p.npad = (Nk + 63) / 64 * 64;
    p.scale = scale;
    p.lse = lse;
    p.prio_young = 0;
    p.drop = uc_make_dropout(drop_p, seed);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_BF16) {
        UC_REQUIRE(D == 64, "uc_attention_fwd_drop(bf16): head_dim must be 64 (got %d)", D);
        UC_REQUIRE(v_layout == UC_V_PACKED_T, "uc_attention_fwd_drop(bf16): V must be in the packed VT layout (uc_vt_pack)");
        UC_REQUIRE(q_sb % 8 == 0 && q_sn % 8 == 0 && q_sh % 8 == 0 && k_sb % 8 == 0 && k_sn % 8 == 0 && k_sh % 8 == 0,
                   "uc_attention_fwd_drop(bf16): Q/K strides must be multiples of 8 elements");
        UC_REQUIRE(o_sb % 4 == 0 && o_sn % 4 == 0 && o_sh % 4 == 0, "uc_attention_fwd_drop(bf16): O strides must be multiples of 4");
        UC_REQUIRE(((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0) && ((uintptr_t)O % 8 == 0),
                   "uc_attention_fwd_drop(bf16): pointer alignment");
        hipLaunchKernelGGL(attn_bf16_drop_kernel, dim3((Nq + 127) / 128, H, B), dim3(256), 0, st, p);
    } else if (dtype == UC_F32) {
        UC_REQUIRE(v_layout == UC_V_ROWMAJOR, "uc_attention_fwd_drop(f32): V must be row-major");
        UC_REQUIRE(D <= 64, "uc_attention_fwd_drop(f32): head_dim must be <= 64 (got %d)", D);
        const dim3 grid((Nq + 127) / 128, H, B);
        if (D <= 32) hipLaunchKernelGGL((attn_f32_kernel<32, true>), grid, dim3(128), 0, st, p);
        else hipLaunchKernelGGL((attn_f32_kernel<64, true>), grid, dim3(128), 0, st, p);
    } else {
        uc_set_error("uc_attention_fwd_drop: unsupported dtype %d", dtype);
        return UC_ERR_BAD_ARG;
    }
    UC_CHECK_LAUNCH("uc_attention_fwd_drop");
    return UC_OK;
}

extern "C" int uc_attention_fwd(const void* Q, const void* K, const void* V, void* O, int dtype, int v_layout,
                                int B, int H, int Nq, int Nk, int D, int64_t q_sb, int64_t q_sn, int64_t q_sh,
                                int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh,
                                int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale, float* lse, uc_stream_t stream) {
    UC_REQUIRE(Q && K && V && O, "uc_attention_fwd: null pointer");
    UC_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && D > 0, "uc_attention_fwd: bad shape");
    UC_REQUIRE(H <= 65535 && B <= 65535, "uc_attention_fwd: B and H must fit a grid dimension");
    AttnParams p;
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D;
    p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
    p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
    p.npad = (Nk + 63) / 64 * 64;
    p.scale = scale;
    p.lse = lse;
    p.prio_young = uc_knobs().attn_prio;
    p.drop = uc_make_dropout(0.f, 0ull);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_BF16) {
        UC_REQUIRE(D == 64, "uc_attention_fwd(bf16): head_dim must be 64 (got %d)", D);
        UC_REQUIRE(v_layout == UC_V_PACKED_T, "uc_attention_fwd(bf16): V must be in the packed VT layout (uc_vt_pack)");
        UC_REQUIRE(q_sb % 8 == 0 && q_sn % 8 == 0 && q_sh % 8 == 0 && k_sb % 8 == 0 && k_sn % 8 == 0 && k_sh % 8 == 0,
                   "uc_attention_fwd(bf16): Q/K strides must be multiples of 8 elements");
        UC_REQUIRE(o_sb % 4 == 0 && o_sn % 4 == 0 && o_sh % 4 == 0, "uc_attention_fwd(bf16): O strides must be multiples of 4");
        UC_REQUIRE(((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0) && ((uintptr_t)O % 8 == 0),
                   "uc_attention_fwd(bf16): pointer alignment");
        // eight waves per workgroup (256 queries share each K / VT tile) when that does not add a mostly empty query tile
        const int nw_env = uc_knobs().attn_nw;
        const int waste8 = (Nq + 255) / 256 * 256 - Nq, waste4 = (Nq + 127) / 128 * 128 - Nq;
        // (a launch whose 256-query tiles would not give every CU two workgroups takes 128-query tiles: one pair of 512 x 512 views is
        //  128 tiles of 256 queries — half the chip idle — or 256 of 128)
        const bool few8 = (int64_t)((Nq + 255) / 256) * H * B < 2 * (int64_t)uc_num_cus();
        const int nw = nw_env == 4 || nw_env == 8 ? nw_env : ((Nq >= 256 && waste8 - waste4 < 64 && !few8) ? 8 : 4);
        const int qtile = 32 * nw, nqt = (Nq + qtile - 1) / qtile;
        p.dGroup = uc_make_fastdiv((unsigned)(8 * nqt)); p.dNq = uc_make_fastdiv((unsigned)nqt); p.dH = uc_make_fastdiv((unsigned)H);
        const int use_dma = uc_knobs().attn_dma;
        // DMA-staged kernel: whole 64-key tiles, 32-bit byte offsets inside one (batch, head)'s K rows / VT rows
        const bool dma_ok = use_dma && (int64_t)nqt * H * B < ((int64_t)1 << 31) && (uintptr_t)O % 16 == 0 && o_sb % 8 == 0 && o_sn % 8 == 0 && o_sh % 8 == 0 && (int64_t)32 * q_sn * 2 < ((int64_t)1 << 31) && (int64_t)Nk * k_sn * 2 < ((int64_t)1 << 31) && (int64_t)64 * p.npad * 2 < ((int64_t)1 << 31);
#ifdef UC_DIAG
        const int dbg = uc_knobs().attn_dbg;   // diag build only (results are wrong): 1 no per-tile barrier, 2 no exp (P = S), 4 no PV MFMAs
        if (dma_ok && nw == 4 && dbg) {
            const dim3 g((unsigned)(nqt * H * B));
            if (dbg == 1) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 1>), g, dim3(256), 0, st, p);
            else if (dbg == 2) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 2>), g, dim3(256), 0, st, p);
            else if (dbg == 3) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 3>), g, dim3(256), 0, st, p);
            else if (dbg == 4) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 4>), g, dim3(256), 0, st, p);
            else if (dbg == 5) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 5>), g, dim3(256), 0, st, p);
            else if (dbg == 8) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 8>), g, dim3(256), 0, st, p);
            else if (dbg == 16) hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 16>), g, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((attn_bf16_dma_kernel<4, 24>), g, dim3(256), 0, st, p);
        } else
#endif
        // persistent 64-queries-per-wave kernel (attention_p64.h) + its fix-up scan: 32-bit DMA offsets as above, at least two key
        // tiles, and (policy) enough (batch, head, 256-query tile) items — one per workgroup of the 2 x 256 (measured break-even: 384 items
        // at 1024 keys, tools/bench_attention_ab.py; 256 items when an item is 64 key tiles long) —, a query count
        // that does not leave a quarter of the last tile empty.  (A wanted log-sum-exp is no obstacle: the kernel rounds scale * log2(e) * Q
        // to bf16, so its scores and LSE carry ~2^-9 of |q| |k| scale — and the backward's dQ kernel rounds Q the same way and recomputes
        // exactly these scores, the dK / dV kernel rounds K instead: the same noise class either way.)
        const int p64_mode = g_uc_attn_p64.load(std::memory_order_relaxed);
        const int64_t p64_items = (int64_t)B * H * ((Nq + 255) / 256);
        const int waste256 = (Nq + 255) / 256 * 256 - Nq;
        const bool p64_ok = dma_ok && p64_mode && Nk > 64 && p64_items < ((int64_t)1 << 28) && (int64_t)64 * q_sn * 2 < ((int64_t)1 << 31) && (int64_t)64 * o_sn * 2 < ((int64_t)1 << 31) &&
                            (p64_mode == 2 || ((p64_items >= 512 || (p64_items >= 256 && Nk >= 4096)) && waste256 * 4 <= Nq));
        if (p64_ok && !g_uc_attn_rs.load(std::memory_order_relaxed)) {
            AttnP64Params pp;
            pp.Q = Q; pp.K = K; pp.V = V; pp.O = O; pp.lse = lse; pp.B = B; pp.H = H; pp.Nq = Nq; pp.Nk = Nk;
            pp.q_sb = q_sb; pp.q_sn = q_sn; pp.q_sh = q_sh; pp.k_sb = k_sb; pp.k_sn = k_sn; pp.k_sh = k_sh; pp.o_sb = o_sb; pp.o_sn = o_sn; pp.o_sh = o_sh;
            pp.npad = p.npad; pp.c = scale * 1.44269504088896340736f; pp.q_prescaled = 0;
            pp.nq = (Nq + 255) / 256; pp.dNq = uc_make_fastdiv((unsigned)pp.nq); pp.dH = uc_make_fastdiv((unsigned)H);
            pp.dbg = nullptr;
            const int per_cu = 2, ncu = uc_num_cus();
            int grid = per_cu * ncu / 8 * 8;
            if (grid < 8) grid = 8;
            const int64_t items8 = (p64_items + 7) / 8 * 8;
            if ((int64_t)grid > items8) grid = (int)items8;
            if (Nk & 63) hipLaunchKernelGGL(attn_bf16_p64_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, pp);
            else hipLaunchKernelGGL(attn_bf16_p64_kernel<false>, dim3((unsigned)grid), dim3(256), 0, st, pp);
            UC_CHECK_LAUNCH("uc_attention_fwd (persistent kernel)");
            const int64_t nblocks = (int64_t)B * H * ((Nq + 63) / 64);
            int fgrid = (int)((nblocks + 255) / 256 < ncu ? (nblocks + 255) / 256 : ncu);
            hipLaunchKernelGGL(attn_bf16_fixup_kernel, dim3((unsigned)fgrid), dim3(256), 0, st, p);
        } else
        if (dma_ok && nw == 8 && g_uc_attn_rs.load(std::memory_order_relaxed)) hipLaunchKernelGGL(attn_bf16_rs_kernel, dim3((unsigned)(nqt * H * B)), dim3(512), 0, st, p);
        else if (dma_ok && nw == 8) hipLaunchKernelGGL(attn_bf16_dma_kernel<8>, dim3((unsigned)(nqt * H * B)), dim3(512), 0, st, p);
        else if (dma_ok) hipLaunchKernelGGL(attn_bf16_dma_kernel<4>, dim3((unsigned)(nqt * H * B)), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(attn_bf16_kernel, dim3((Nq + 127) / 128, H, B), dim3(256), 0, st, p);
    } else if (dtype == UC_F32) {
        UC_REQUIRE(v_layout == UC_V_ROWMAJOR, "uc_attention_fwd(f32): V must be row-major");
        UC_REQUIRE(D <= 64, "uc_attention_fwd(f32): head_dim must be <= 64 (got %d)", D);
        const dim3 grid((Nq + 127) / 128, H, B);
        if (D <= 32) hipLaunchKernelGGL((attn_f32_kernel<32>), grid, dim3(128), 0, st, p);
        else hipLaunchKernelGGL((attn_f32_kernel<64>), grid, dim3(128), 0, st, p);
    } else {
        uc_set_error("uc_attention_fwd: unsupported dtype %d", dtype);
        return UC_ERR_BAD_ARG;
    }
    UC_CHECK_LAUNCH("uc_attention_fwd");
    return UC_OK;
}
