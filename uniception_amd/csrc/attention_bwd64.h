// attn_bwd_dkv64_kernel — dK / dV of the attention backward with 64 keys per wave (round 5; included by attention_bwd.hip).
//
// Why.  attn_bwd_dkv_kernel gives a wave 32 keys: per 64-query tile it issues 32 MFMAs and reads 32 KiB of LDS fragments for them
// (the Q and dO rows once as A operands of S / dP, once transposed for dK^T / dV^T) — 1 KiB per MFMA, which is the LDS port's whole
// bandwidth at the MFMA peak, with the transposing reads' 2-way conflicts on top; the compiler's schedule runs the softmax between the
// two MFMA groups of a query block.  Here (the forward's attention_p64.h, applied to the backward):
//   * a wave owns 64 keys (two 32-key blocks), so every fragment read feeds two MFMAs: 0.5 KiB per MFMA; its dK^T / dV^T accumulators
//     (128 registers) live in AGPRs, the stationary K / V fragments (64), the scores and dP (64) and two generations of P / dS (64) in
//     VGPRs — 350 registers, hence ONE wave per SIMD (four waves, 256 keys per workgroup, one workgroup per CU) and a schedule in
//     which the wave's own vector work runs under its own MFMAs;
//   * hand-placed slots (inline-asm MFMAs, one per slot, `sched_barrier` between slots): per 32-query block
//       phase A: 16 slots  S[kb] = Q K^T - lse2,  dP[kb] = dO V^T - delta      (fragment reads of the Q / dO rows beside them)
//       phase B: 16 slots  dV^T += dO^T P,  dK^T += Q^T dS  of the PREVIOUS block, and beside them this block's softmax:
//                per slot two exponentials, two products, two conversions (P = exp2(S), dS = P dP);
//   * the per-query scalars -lse2 and -delta are the accumulators' start values (the C operand of a chain's first MFMA), read from
//     LDS once per block; they come from a scratch array the dQ kernel (which runs first) writes: [B H][2][Nq rounded up to 128], absent
//     queries at -1e30 / 0 so that a ragged last tile needs no mask (P = exp2(-1e30) = 0);
//   * Q / dO tiles (and the 128 scalars) stream through a three-slot LDS ring by LDS-DMA, ONE barrier per 64-query tile: tile t+1 is
//     made visible and tile t-1's slot released at the same point (between the tile's two query blocks), the pieces of tile t+2 are
//     issued behind it and waited for a whole tile later.
//   * persistent workgroups (one per CU) walk a list of (batch, head, 256-key tile) items dealt so that the key tiles of one (batch,
//     head) — which stream the same Q / dO rows — run on ONE XCD; the tile stream runs on ACROSS item seams (the next item's first
//     tiles are in flight under the current item's last ones), the next item's K / V rows wait in an LDS staging area from the
//     item's start, so a seam costs its arithmetic (read-out, 128 accumulator clears, 16 fragment reads), not three HBM round trips.
// Key rows beyond Nk read as zeros and are not stored; waves wholly beyond Nk compute along (barriers) and store nothing.
// Needs Nq > 64 (two query tiles: the lookahead of two tiles then stays inside the next item).
#pragma once
#include "attention_p64.h"

#define B64_TILE 8192
#define B64_AUX_OFF (2 * B64_TILE)               // neg_lse2[64] | neg_delta[64] | 512 B the idle waves' pieces land in
#define B64_SLOT (2 * B64_TILE + 1024)
#define B64_STAGE_OFF (3 * B64_SLOT)              // per wave: the NEXT item's 64 K rows | 64 V rows (swizzled tile images), 16 KiB
#define B64_BOUNCE_OFF (B64_STAGE_OFF + 4 * 2 * B64_TILE)      // per wave: 8 KiB for the read-out (whole 128-byte rows per store)
#define B64_LDS_BYTES (B64_BOUNCE_OFF + 4 * B64_TILE)           // 147 KiB: one workgroup per CU

__device__ __forceinline__ void b64_mfma_acc_a(float16_t& d, p64_bf16x8_t a, p64_bf16x8_t b) {       // accumulator in an AGPR block
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// score chains: accumulator in VGPRs, the STATIONARY operand (K / V fragments) in accumulator registers (gfx90a+: an MFMA's A / B
// operands may come from either file) — 64 VGPRs the vector work needs
__device__ __forceinline__ void b64_mfma_first_b(float16_t& d, p64_bf16x8_t a, p64_bf16x8_t b, const float16_t& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void b64_mfma_acc_b(float16_t& d, p64_bf16x8_t a, p64_bf16x8_t b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ float b64_mul(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef float b64_float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ b64_float2_t b64_pk_mul(b64_float2_t a, b64_float2_t b) {
    b64_float2_t r;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// LDS-DMA pieces with M0 declared clobbered (one wave per SIMD: every instruction is four cycles of the wave's issue time, and the
// save / restore pair of p64_dma16 is two of five)
__device__ __forceinline__ void b64_dma16(unsigned voff, p64_uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" : : "v"(voff), "s"(srd), "s"(lds_byte_addr), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ void b64_dma4(unsigned voff, p64_uint4_t srd, unsigned soff, unsigned lds_byte_addr) {      // 64 lanes x 4 B
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %3 offen lds" : : "v"(voff), "s"(srd), "s"(lds_byte_addr), "s"(soff) : "memory", "m0");
}
// the two transposing reads of one transposed A operand (see tr_operand): addresses a0 / a1 are the lane's, `imm` the slab offset
template <int IMM>
__device__ __forceinline__ p64_bf16x8_t b64_tr(const char* a0, const char* a1) {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ab_lds_v4_t)(ab_lds_ptr_t)(a0 + IMM));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((ab_lds_v4_t)(ab_lds_ptr_t)(a1 + IMM));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

#ifdef B64_TIMING     // probe builds: wave 0 of every workgroup accumulates s_memtime deltas per code section into p.dbg[8 g + section]
#define B64_TS(i) do { if (wave == 0) { P64_PIN(); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); t_acc[i] += n_ - t_last; t_last = n_; P64_PIN(); } } while (0)
#else
#define B64_TS(i) do { } while (0)
#endif

__global__ __launch_bounds__(256, 1) void attn_bwd_dkv64_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[B64_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(ab_lds_ptr_t)smem;
    const int nt = (p.Nq + 63) >> 6;
#ifdef B64_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_last, r_begin = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- this workgroup's item list: XCD x = g % 8 owns the (batch, head) pairs bh = 8 k + x; its items j = k * nkt + key tile are dealt
    //      round-robin to the XCD's workgroups ----
    const int g = blockIdx.x, xcd = g & 7, slot_w = g >> 3, nslots = (int)(gridDim.x >> 3);
    const int nbh = p.B * p.H, nkt = (p.Nk + 255) >> 8;
    const int items_x = ((nbh - xcd + 7) >> 3) * nkt;
    if (slot_w >= items_x) return;          // (uniform per workgroup)
    // (descriptors span the WHOLE tensors — the launcher checks that they fit 32-bit byte offsets — and an item is three byte offsets into
    // them: a query tile past Nq then reads the next batch's rows instead of zeros, finite values that meet P = exp2(-1e30) = 0)
    struct Item { unsigned oq, oo, oa; unsigned long long kb, vb; int b, h, key0; };
    auto make_item = [&](int j) -> Item {
        Item it;
        const int kq = j / nkt, kt = j - kq * nkt;
        const int bh = kq * 8 + xcd;
        it.b = bh / p.H;
        it.h = bh - it.b * p.H;
        it.key0 = __builtin_amdgcn_readfirstlane(kt * 256 + wave * 64);
        // (readfirstlane: the integer division above runs on the vector unit, and everything derived from it would stay there)
        it.b = __builtin_amdgcn_readfirstlane(it.b);
        it.h = __builtin_amdgcn_readfirstlane(it.h);
        it.oq = (unsigned)(((int64_t)it.b * p.q_sb + (int64_t)it.h * p.q_sh) * 2);
        it.oo = (unsigned)(((int64_t)it.b * p.o_sb + (int64_t)it.h * p.o_sh) * 2);
        it.oa = (unsigned)((int64_t)(it.b * p.H + it.h) * 2 * p.nq_pad * 4);
        it.kb = (unsigned long long)(p.K + (int64_t)it.b * p.k_sb + (int64_t)it.h * p.k_sh);
        it.vb = (unsigned long long)(p.V + (int64_t)it.b * p.v_sb + (int64_t)it.h * p.v_sh);
        return it;
    };

    // ---- the stream: per tile and wave 2 pieces of Q rows, 2 of dO rows (rows 16 w .. 16 w + 15), 1 of the scalars ----
    const p64_uint4_t srd_q = p64_make_srd(p.Q, (unsigned)((((int64_t)p.B - 1) * p.q_sb + ((int64_t)p.H - 1) * p.q_sh + ((int64_t)p.Nq - 1) * p.q_sn + 64) * 2));
    const p64_uint4_t srd_o = p64_make_srd(p.dO, (unsigned)((((int64_t)p.B - 1) * p.o_sb + ((int64_t)p.H - 1) * p.o_sh + ((int64_t)p.Nq - 1) * p.o_sn + 64) * 2));
    const p64_uint4_t srd_a = p64_make_srd(p.aux, (unsigned)((int64_t)nbh * 2 * p.nq_pad * 4));
    unsigned voff_q[2], voff_o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = i * 8 + (lane >> 3);
        const int cch = (lane & 7) ^ ((rr >> 1) & 7);
        voff_q[i] = (unsigned)(((int64_t)rr * p.q_sn + cch * 8) * 2);
        voff_o[i] = (unsigned)(((int64_t)rr * p.o_sn + cch * 8) * 2);
    }
    const unsigned voff_a = (unsigned)lane * 4u;
    const unsigned qstep = (unsigned)(64 * p.q_sn * 2), ostep = (unsigned)(64 * p.o_sn * 2);
    const unsigned q16 = (unsigned)(16 * p.q_sn * 2) * (unsigned)wave, o16 = (unsigned)(16 * p.o_sn * 2) * (unsigned)wave;
    const unsigned a_so = (unsigned)((wave & 1) * p.nq_pad * 4);
    struct Pieces { unsigned so_q, so_o, so_a, dst; };
    auto prep = [&](const Item& it, int t, int s) -> Pieces {
        Pieces pc;
        pc.so_q = __builtin_amdgcn_readfirstlane(it.oq + (unsigned)t * qstep + q16);
        pc.so_o = __builtin_amdgcn_readfirstlane(it.oo + (unsigned)t * ostep + o16);
        pc.so_a = __builtin_amdgcn_readfirstlane(it.oa + a_so + (unsigned)t * 256u);
        pc.dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(s * B64_SLOT) + (unsigned)wave * 2048u);
        return pc;
    };
    auto issue_piece = [&](const Pieces& pc, auto n_tag) __attribute__((always_inline)) {
        constexpr int n = decltype(n_tag)::value;
        if constexpr (n == 0) b64_dma16(voff_q[0], srd_q, pc.so_q, pc.dst);
        else if constexpr (n == 1) b64_dma16(voff_q[1], srd_q, pc.so_q, pc.dst + 1024);
        else if constexpr (n == 2) b64_dma16(voff_o[0], srd_o, pc.so_o, pc.dst + B64_TILE);
        else if constexpr (n == 3) b64_dma16(voff_o[1], srd_o, pc.so_o, pc.dst + B64_TILE + 1024);
        else b64_dma4(voff_a, srd_a, pc.so_a, pc.dst - (unsigned)wave * 2048u + B64_AUX_OFF + (unsigned)wave * 256u);
    };
    auto issue_all = [&](const Pieces& pc) __attribute__((always_inline)) {
        issue_piece(pc, P64Int<0>()); issue_piece(pc, P64Int<1>()); issue_piece(pc, P64Int<2>()); issue_piece(pc, P64Int<3>());
        issue_piece(pc, P64Int<4>());
    };
    // the wave's 64 K rows and 64 V rows of an item into its staging area (16 pieces of 8 rows; key rows >= Nk read as zeros)
    auto issue_stage = [&](const Item& it) {
        const p64_uint4_t srd_k = p64_make_srd((const void*)it.kb, (unsigned)((((int64_t)p.Nk - 1) * p.k_sn + 64) * 2));
        const p64_uint4_t srd_v = p64_make_srd((const void*)it.vb, (unsigned)((((int64_t)p.Nk - 1) * p.v_sn + 64) * 2));
        const unsigned dst = lds0 + (unsigned)(B64_STAGE_OFF + wave * 2 * B64_TILE);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const int cch = (lane & 7) ^ ((rr >> 1) & 7);
            const unsigned vk = (unsigned)(((int64_t)rr * p.k_sn + cch * 8) * 2), vv = (unsigned)(((int64_t)rr * p.v_sn + cch * 8) * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p64_dma16(vk, srd_k, (unsigned)((int64_t)(it.key0 + 16 * k) * p.k_sn * 2), __builtin_amdgcn_readfirstlane(dst + k * 2048 + i * 1024));
                p64_dma16(vv, srd_v, (unsigned)((int64_t)(it.key0 + 16 * k) * p.v_sn * 2), __builtin_amdgcn_readfirstlane(dst + B64_TILE + k * 2048 + i * 1024));
            }
        }
    };

    int j = slot_w;
    Item cur = make_item(j);
    Item nxt = cur;
    bool has_next = j + nslots < items_x;
    if (has_next) nxt = make_item(j + nslots);
    issue_stage(cur);
    issue_all(prep(cur, 0, 0));
    issue_all(prep(cur, 1, 1));
    // ring slot 2 is read as "the tile before tile 0" by the first phase B (beside P = dS = 0): finite data
    {
        uint4* z = reinterpret_cast<uint4*>(smem + 2 * B64_SLOT);
#pragma unroll
        for (int i = 0; i < 2 * B64_TILE / 16 / 256; ++i) z[i * 256 + tid] = make_uint4(0u, 0u, 0u, 0u);
    }

    p64_bf16x8_t kf[2][4], vf[2][4];       // stationary B operands: lane key = kb * 32 + l31, channels 16 st + 8 hi .. + 7; K carries scale * log2(e)
    float16_t dk[2][2], dv[2][2];          // [channel block db][key block kb], transposed: row = channel, column = key
    float16_t S[2], dP[2];                 // [key block]: rows = the block's 32 queries
    p64_bf16x8_t P[2][2][2], dS[2][2][2];  // [generation][key block][16-query slab]
    p64_bf16x8_t FA[8], F[4];              // row fragments of phase A, ring of transposed fragments of phase B (plans: the phases' comments)
    float16_t L, Dl;                       // the next block's start values
    float e0 = 0.f, e1 = 0.f;              // the exponentials in flight between an "exp" slot and the next ("fin") slot
    b64_float2_t dd = {0.f, 0.f};          // ... and the pair's products, converted in the exponential slot after

    // lane addresses inside the ring (advanced with the slots): row fragments, transposed fragments, start values
    const char* ka[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) ka[st] = smem + bswz(l31, 2 * st + hi);
    const char* tr0[2];
    const char* tr1[2];
    {
        const int jj = lane & 15, piece = jj & 3;
        const int row0 = 4 * (lane >> 5) + (jj >> 2), row1 = row0 + 8;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int chunk = ((db * 32 + (((lane >> 4) & 1) << 4)) >> 3) + (piece >> 1);
            tr0[db] = smem + 2 * B64_SLOT + row0 * 128 + ((chunk ^ ((row0 >> 1) & 7)) << 4) + ((piece & 1) << 3);      // (starts at ring slot 2)
            tr1[db] = smem + 2 * B64_SLOT + row1 * 128 + ((chunk ^ ((row1 >> 1) & 7)) << 4) + ((piece & 1) << 3);
        }
    }
    const char* cad = smem + B64_AUX_OFF + hi * 16;          // start values of the CURRENT tile: + which * 256 + qb * 128 + k4 * 32
    int s_rd = 0;

    // start values of a query block (base = its tile's scalars + qb * 128): accumulator row r = 4 k4 + i is query 8 k4 + 4 hi + i of the
    // block.  They are read ONCE per block into L / Dl (both key blocks' chains take them as the C operand of their first MFMA): read
    // into the score registers themselves (16 reads per block instead of 8, all of them in phase B beside its 16 transposing reads) the
    // LDS port, not the matrix pipe, set the pace — 4 waves x 280 port cycles per phase B against 16 MFMAs x 32.
    auto init_quad = [&](const char* base, auto k4_tag) __attribute__((always_inline)) {
        constexpr int k4 = decltype(k4_tag)::value;
        const float4_t l4 = *reinterpret_cast<const float4_t*>(base + (k4 & 3) * 32 + (k4 >> 2) * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) { if constexpr (k4 < 4) L[4 * k4 + i] = l4[i]; else Dl[4 * (k4 - 4) + i] = l4[i]; }
    };
    // the wave's stationary operands out of its staging area (landed), the accumulators and "block -1" cleared
    auto load_stationary = [&]() __attribute__((always_inline)) {
        const char* sk = smem + B64_STAGE_OFF + wave * 2 * B64_TILE;
        const float c = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                union { p64_bf16x8_t v; unsigned u[4]; } a;
                a.v = *reinterpret_cast<const p64_bf16x8_t*>(sk + kb * 4096 + bswz(l31, 2 * st + hi));
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) a.u[q4] = pack_bf16x2(__uint_as_float(a.u[q4] << 16) * c, __uint_as_float(a.u[q4] & 0xffff0000u) * c);
                kf[kb][st] = a.v;
                vf[kb][st] = *reinterpret_cast<const p64_bf16x8_t*>(sk + B64_TILE + kb * 4096 + bswz(l31, 2 * st + hi));
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) { dk[i][k] = (float16_t)(0.f); dv[i][k] = (float16_t)(0.f); }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                P[1][kb][hf] = (p64_bf16x8_t)(0.f);       // generation 1 = "block -1"
                dS[1][kb][hf] = (p64_bf16x8_t)(0.f);
            }
        // the item's first phase A finishes "block -1"'s softmax (pairs 11..15, key block 1): exp2(-1e30) = 0, 0 * 0 = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[1][r] = -1e30f; dP[1][r] = 0.f; }
        e0 = 0.f; e1 = 0.f;
        dd = (b64_float2_t){0.f, 0.f};
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_stationary();
    if (has_next) issue_stage(nxt);        // (behind the fragment reads above: the compiler's lgkmcnt wait for them precedes their first use)
    // block 0's start values and first fragments
    init_quad(cad, P64Int<0>()); init_quad(cad, P64Int<1>()); init_quad(cad, P64Int<2>()); init_quad(cad, P64Int<3>());
    init_quad(cad, P64Int<4>()); init_quad(cad, P64Int<5>()); init_quad(cad, P64Int<6>()); init_quad(cad, P64Int<7>());
    FA[0] = *reinterpret_cast<const p64_bf16x8_t*>(ka[0]);
    FA[1] = *reinterpret_cast<const p64_bf16x8_t*>(ka[0] + B64_TILE);
    FA[2] = *reinterpret_cast<const p64_bf16x8_t*>(ka[1]);
    FA[3] = *reinterpret_cast<const p64_bf16x8_t*>(ka[1] + B64_TILE);
    FA[4] = *reinterpret_cast<const p64_bf16x8_t*>(ka[2]);
    FA[5] = *reinterpret_cast<const p64_bf16x8_t*>(ka[2] + B64_TILE);

    // The schedule of one 32-query block j (slots g = 0 .. 39 counted from its phase A; one MFMA per slot):
    //   phase A, g = 0..15   g < 8: S[0] / dP[0] (key block 0: even / odd slots, chunk pair st = g >> 1), g >= 8: S[1] / dP[1]; the chains' first
    //                        MFMAs take the block's start values L / Dl as C.  Row fragments FA[2 st] (Q rows), FA[2 st + 1] (dO rows): each
    //                        feeds slots 2 st (+1) and 8 + 2 st (+1); FA[0..5] were read by the previous phase B (slots 10..15), FA[6..7] in g = 0, 1
    //                        (six slots ahead of their first use: four were not enough — the row fragments' waits cost 17 % of the kernel).
    //   phase B, g = 16..31  dV^T / dK^T of block j - 1 (fragment plan below) — unchanged by block j.
    //   the block's softmax  runs at HALF density from g = 9 to g = 39 — under phase A's second half, phase B and the NEXT block's first 8
    //                        slots: pair q (key block q >> 3, accumulator rows 2 (q & 7), + 1) has its two exponentials in slot 9 + 2 q and its
    //                        two products and two conversions in the slot after (pairs 14, 15: slots 37, 38).  Key block 0's scores are
    //                        complete at g = 7 and die at the next block's g = 0 (32); key block 1's are complete at g = 15 and die at 40.
    //                        All 32 slots of a block then carry ~19 cycles of vector work beside their MFMA (32) instead of 16 slots carrying 45.
    // Phase B fragments b0..b7: pair p = 2 hf + db: b(2 p) = dO^T, b(2 p + 1) = Q^T of (16-query slab hf of the previous block, channel block
    // db); used in slots 4 p + {0, 1} and 4 p + {2, 3}; fragment m sits in F[m & 3]; loads b0 A12, b1 A13, b2 A15, b3 B0, b4 B2, b5 B4, b6 B6, b7 B8.
    // (the product's conversion rides in the NEXT exponential slot: behind the packed multiply it costs a wait state, and with one wave
    // per SIMD every instruction — an s_nop too — is four cycles of issue time)
#ifndef B64_PKMUL
#define B64_PKMUL 0     // (v_pk_mul_f32 is no bargain here: two v_mul_f32 are 8 % faster on the whole kernel — measured, tools/probes/run_b64.sh)
#endif
#ifndef B64_DEFER
#define B64_DEFER 1
#endif
#ifndef B64_DBG
#define B64_DBG 0       // probe builds, wrong results: 1 no softmax arithmetic, 2 no start-value reads, 4 no transposing reads, 8 no row-fragment reads in the phases, 16 no DMA pieces in the loop
#endif
    auto sm_cvt_ds = [&](auto qp_tag, auto gen_tag) __attribute__((always_inline)) {
        constexpr int qp = decltype(qp_tag)::value, kbp = qp >> 3, ip = qp & 7, G = decltype(gen_tag)::value;
        union { p64_bf16x8_t v; unsigned u[4]; } d;
        d.v = dS[G][kbp][ip >> 2];
        d.u[ip & 3] = p64_cvt_pk(dd.x, dd.y);
        dS[G][kbp][ip >> 2] = d.v;
    };
    auto sm_exp = [&](auto q_tag, auto gen_tag) __attribute__((always_inline)) {       // gen: generation of pair q - 1 (pair 15 of the previous block for q = 0)
        constexpr int q = decltype(q_tag)::value, kb = q >> 3, i = q & 7;
        if constexpr (B64_DBG & 1) return;
        // (conversion first: a vector instruction right behind a transcendental one costs a wait state)
        if constexpr (B64_DEFER) sm_cvt_ds(P64Int<((q + 15) & 15)>(), gen_tag);
        e0 = p64_exp2(S[kb][2 * i]);
        e1 = p64_exp2(S[kb][2 * i + 1]);
    };
    auto sm_fin = [&](auto q_tag, auto gen_tag) __attribute__((always_inline)) {
        constexpr int q = decltype(q_tag)::value, kb = q >> 3, i = q & 7, G = decltype(gen_tag)::value;
        if constexpr (B64_DBG & 1) return;
        if constexpr (B64_PKMUL) dd = b64_pk_mul((b64_float2_t){e0, e1}, (b64_float2_t){dP[kb][2 * i], dP[kb][2 * i + 1]});
        else { dd.x = b64_mul(e0, dP[kb][2 * i]); dd.y = b64_mul(e1, dP[kb][2 * i + 1]); }
        union { p64_bf16x8_t v; unsigned u[4]; } a;
        a.v = P[G][kb][i >> 2];
        a.u[i & 3] = p64_cvt_pk(e0, e1);
        P[G][kb][i >> 2] = a.v;
        if constexpr (!B64_DEFER) sm_cvt_ds(q_tag, gen_tag);
    };
    // ---- phase A of query block qb of the current tile; beside it: the tail of the previous block's softmax (slots 0..7, generation
    //      qb ^ 1), the head of this block's (slots 9..15, generation qb), from slot 10 the NEXT block's start values (nb_init: the first
    //      MFMAs of key block 1 have read L / Dl in slots 8, 9), in the odd block of a tile the five pieces of tile t + 2 ----
    auto phase_a = [&](auto qb_tag, const char* nb_init, const Pieces& pc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qb_tag)::value, pq = qb ^ 1;       // pq: the previous block's position in ITS tile (which tr0 / tr1 point at)
        auto slot = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value, st = (k & 7) >> 1, kb = k >> 3;
            if constexpr (!(B64_DBG & 8) && k < 2) FA[6 + k] = *reinterpret_cast<const p64_bf16x8_t*>(ka[3] + qb * 4096 + (k & 1) * B64_TILE);
            if constexpr (!(B64_DBG & 4) && k == 12) F[0] = b64_tr<B64_TILE + 4096 * pq>(tr0[0], tr1[0]);          // b0: dO^T, slab 2 pq, channel block 0
            if constexpr (!(B64_DBG & 4) && k == 13) F[1] = b64_tr<4096 * pq>(tr0[0], tr1[0]);                     // b1: Q^T,  slab 2 pq, channel block 0
            if constexpr (!(B64_DBG & 4) && k == 15) F[2] = b64_tr<B64_TILE + 4096 * pq>(tr0[1], tr1[1]);          // b2: dO^T, slab 2 pq, channel block 1
            if constexpr (qb == 1) {
                if constexpr (!(B64_DBG & 16) && k == 1) issue_piece(pc, P64Int<0>());
                if constexpr (!(B64_DBG & 16) && k == 2) issue_piece(pc, P64Int<1>());
                if constexpr (!(B64_DBG & 16) && k == 4) issue_piece(pc, P64Int<2>());
                if constexpr (!(B64_DBG & 16) && k == 5) issue_piece(pc, P64Int<3>());
                if constexpr (!(B64_DBG & 16) && k == 6) issue_piece(pc, P64Int<4>());
            }
            if constexpr ((k & 7) == 0) b64_mfma_first_b(S[kb], FA[0], kf[kb][0], L);
            else if constexpr ((k & 7) == 1) b64_mfma_first_b(dP[kb], FA[1], vf[kb][0], Dl);
            else if constexpr ((k & 1) == 0) b64_mfma_acc_b(S[kb], FA[2 * st], kf[kb][st]);
            else b64_mfma_acc_b(dP[kb], FA[2 * st + 1], vf[kb][st]);
            // ---- vector work ----
            if constexpr (k == 0) sm_fin(P64Int<11>(), P64Int<qb ^ 1>());
            if constexpr (k == 1) sm_exp(P64Int<12>(), P64Int<qb ^ 1>());
            if constexpr (k == 2) sm_fin(P64Int<12>(), P64Int<qb ^ 1>());
            if constexpr (k == 3) sm_exp(P64Int<13>(), P64Int<qb ^ 1>());
            if constexpr (k == 4) sm_fin(P64Int<13>(), P64Int<qb ^ 1>());
            if constexpr (k == 5) sm_exp(P64Int<14>(), P64Int<qb ^ 1>());
            if constexpr (k == 6) { sm_fin(P64Int<14>(), P64Int<qb ^ 1>()); sm_exp(P64Int<15>(), P64Int<qb ^ 1>()); }
            if constexpr (k == 7) sm_fin(P64Int<15>(), P64Int<qb ^ 1>());
            if constexpr (k == 9) sm_exp(P64Int<0>(), P64Int<qb ^ 1>());              // (converts the previous block's last product)
            if constexpr (k >= 11 && (k & 1) == 1) sm_exp(P64Int<((k - 9) >> 1)>(), P64Int<qb>());
            if constexpr (k >= 10 && (k & 1) == 0) sm_fin(P64Int<((k - 10) >> 1)>(), P64Int<qb>());
            // ---- the next block's start values ----
            if constexpr (!(B64_DBG & 2) && k >= 10 && k <= 13) init_quad(nb_init, P64Int<k - 10>());
            if constexpr (!(B64_DBG & 2) && k == 14) { init_quad(nb_init, P64Int<4>()); init_quad(nb_init, P64Int<5>()); }
            if constexpr (!(B64_DBG & 2) && k == 15) { init_quad(nb_init, P64Int<6>()); init_quad(nb_init, P64Int<7>()); }
            P64_PIN();
        };
        slot(P64Int<0>()); slot(P64Int<1>()); slot(P64Int<2>()); slot(P64Int<3>());
        slot(P64Int<4>()); slot(P64Int<5>()); slot(P64Int<6>()); slot(P64Int<7>());
        slot(P64Int<8>()); slot(P64Int<9>()); slot(P64Int<10>()); slot(P64Int<11>());
        slot(P64Int<12>()); slot(P64Int<13>()); slot(P64Int<14>()); slot(P64Int<15>());
    };
    // ---- phase B behind phase A of block (t, ODD): dV^T / dK^T of the previous block (generation ODD ^ 1 of P / dS; tile tr0 / tr1 point
    //      at, slabs 2 pq, 2 pq + 1) beside the middle of this block's softmax (generation ODD); the next block's first row fragments come
    //      from nb_rows ----
    auto phase_b = [&](auto odd_tag, int nb_rows) __attribute__((always_inline)) {
        constexpr int ODD = decltype(odd_tag)::value, GW = ODD, GR = ODD ^ 1, pq = ODD ^ 1;
        auto slot = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value, pp = k >> 2, c = k & 3, hf = pp >> 1, db = pp & 1;
            if constexpr (!(B64_DBG & 4) && k == 0) F[3] = b64_tr<4096 * pq>(tr0[1], tr1[1]);                          // b3: Q^T  slab 2 pq,     channel block 1
            if constexpr (!(B64_DBG & 4) && k == 2) F[0] = b64_tr<B64_TILE + 4096 * pq + 2048>(tr0[0], tr1[0]);        // b4: dO^T slab 2 pq + 1, channel block 0
            if constexpr (!(B64_DBG & 4) && k == 4) F[1] = b64_tr<4096 * pq + 2048>(tr0[0], tr1[0]);                   // b5: Q^T
            if constexpr (!(B64_DBG & 4) && k == 6) F[2] = b64_tr<B64_TILE + 4096 * pq + 2048>(tr0[1], tr1[1]);        // b6: dO^T slab 2 pq + 1, channel block 1
            if constexpr (!(B64_DBG & 4) && k == 8) F[3] = b64_tr<4096 * pq + 2048>(tr0[1], tr1[1]);                   // b7: Q^T
            if constexpr (!(B64_DBG & 8) && k >= 10) FA[k - 10] = *reinterpret_cast<const p64_bf16x8_t*>(ka[(k - 10) >> 1] + nb_rows + ((k - 10) & 1) * B64_TILE);
            if constexpr (c == 0) b64_mfma_acc_a(dv[db][0], F[(2 * pp) & 3], P[GR][0][hf]);
            else if constexpr (c == 1) b64_mfma_acc_a(dv[db][1], F[(2 * pp) & 3], P[GR][1][hf]);
            else if constexpr (c == 2) b64_mfma_acc_a(dk[db][0], F[(2 * pp + 1) & 3], dS[GR][0][hf]);
            else b64_mfma_acc_a(dk[db][1], F[(2 * pp + 1) & 3], dS[GR][1][hf]);
            if constexpr (k == 0) sm_fin(P64Int<3>(), P64Int<GW>());
            else if constexpr (k & 1) sm_exp(P64Int<((k + 7) >> 1)>(), P64Int<GW>());
            else sm_fin(P64Int<((k + 6) >> 1)>(), P64Int<GW>());
            P64_PIN();
        };
        slot(P64Int<0>()); slot(P64Int<1>()); slot(P64Int<2>()); slot(P64Int<3>());
        slot(P64Int<4>()); slot(P64Int<5>()); slot(P64Int<6>()); slot(P64Int<7>());
        slot(P64Int<8>()); slot(P64Int<9>()); slot(P64Int<10>()); slot(P64Int<11>());
        slot(P64Int<12>()); slot(P64Int<13>()); slot(P64Int<14>()); slot(P64Int<15>());
    };
    auto advance_tr = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < 2; ++db) { tr0[db] += step; tr1[db] += step; }
    };

    B64_TS(6);
    // ================================================================== the item loop ============================================
    for (;;) {
        // ---- the tile loop: A(2t) B(2t) | barrier | A(2t+1) B(2t+1); the stream runs on into the next item ----
        for (int t = 0; t < nt; ++t) {
            const int nstep = s_rd == 2 ? -2 * B64_SLOT : B64_SLOT;        // ring step current -> next tile
            const int pstep = s_rd == 0 ? -2 * B64_SLOT : B64_SLOT;        // ring step previous -> current tile
            const int s_w = __builtin_amdgcn_readfirstlane(s_rd == 0 ? 2 : s_rd - 1);      // the previous tile's slot: tile t + 2 of the stream lands there
            // (no next item: a harmless re-read into a slot nobody reads again)
            // (one prep on selected scalars: three inlined preps behind branches cost the wave's only issue stream ~300 cycles per tile)
            const bool in_cur = t + 2 < nt, use_nxt = !in_cur && has_next;
            Item src = cur;
            src.oq = use_nxt ? nxt.oq : cur.oq;
            src.oo = use_nxt ? nxt.oo : cur.oo;
            src.oa = use_nxt ? nxt.oa : cur.oa;
            const Pieces pc = prep(src, in_cur ? t + 2 : (use_nxt ? t + 2 - nt : nt - 1), s_w);
            B64_TS(0);
            phase_a(P64Int<0>(), cad + 128, pc);
            B64_TS(1);
            phase_b(P64Int<0>(), 4096);
            B64_TS(2);
            advance_tr(pstep);
            // every wave is done with tile t - 1 (its last reads were phase B above), and its own pieces of tile t + 1 — issued a whole tile
            // ago — have landed: behind the barrier tile t + 1 is visible and the slot of tile t - 1 may be overwritten
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            B64_TS(3);
            phase_a(P64Int<1>(), cad + nstep, pc);
            B64_TS(4);
            phase_b(P64Int<1>(), nstep);
            B64_TS(5);
#pragma unroll
            for (int st = 0; st < 4; ++st) ka[st] += nstep;
            cad += nstep;
            s_rd = s_rd == 2 ? 0 : s_rd + 1;
        }
        B64_TS(0);
        // ---- the tail of the last block's softmax (pairs 11..15; in the stream it runs under the next block's first slots), then its
        //      dV^T / dK^T (tile nt - 1, slabs 2, 3; generation 1; tr0 / tr1 were advanced to that tile in its iteration) ----
        asm volatile("s_nop 0");
        sm_fin(P64Int<11>(), P64Int<1>());
        sm_exp(P64Int<12>(), P64Int<1>()); asm volatile("s_nop 0"); sm_fin(P64Int<12>(), P64Int<1>());
        sm_exp(P64Int<13>(), P64Int<1>()); asm volatile("s_nop 0"); sm_fin(P64Int<13>(), P64Int<1>());
        sm_exp(P64Int<14>(), P64Int<1>()); asm volatile("s_nop 0"); sm_fin(P64Int<14>(), P64Int<1>());
        sm_exp(P64Int<15>(), P64Int<1>()); asm volatile("s_nop 0"); sm_fin(P64Int<15>(), P64Int<1>());
        if constexpr (B64_DEFER) {
            asm volatile("s_nop 1");
            sm_cvt_ds(P64Int<15>(), P64Int<1>());
        }
        p64_mfma_settle();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const p64_bf16x8_t ta = hf ? b64_tr<B64_TILE + 4096 + 2048>(tr0[db], tr1[db]) : b64_tr<B64_TILE + 4096>(tr0[db], tr1[db]);
                const p64_bf16x8_t tq = hf ? b64_tr<4096 + 2048>(tr0[db], tr1[db]) : b64_tr<4096>(tr0[db], tr1[db]);
                b64_mfma_acc_a(dv[db][0], ta, P[1][0][hf]);
                b64_mfma_acc_a(dv[db][1], ta, P[1][1][hf]);
                b64_mfma_acc_a(dk[db][0], tq, dS[1][0][hf]);
                b64_mfma_acc_a(dk[db][1], tq, dS[1][1][hf]);
            }
        p64_mfma_settle();

        // ---- read-out: inverse RoPE on dK, scale, bf16; bounced through the wave's 8 KiB so that a store instruction writes 8 whole
        //      128-byte rows (a per-lane store at a row stride touches 32 lines, and the store path pays per line) ----
        {
            char* ob = smem + B64_BOUNCE_OFF + wave * B64_TILE;
            const int64_t rows = min((int64_t)64, (int64_t)p.Nk - cur.key0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {                               // m = 0: dK, 1: dV
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float16_t gk[2] = {m ? dv[0][kb] : dk[0][kb], m ? dv[1][kb] : dk[1][kb]};
                    if (m == 0 && p.rope_kpos) {
                        const int key = cur.key0 + kb * 32 + l31, kc = key < p.Nk ? key : p.Nk - 1;
                        ab_rope_inverse(gk, p.rope_kpos + ((int64_t)cur.b * p.Nk + kc) * 2, hi, p.rope_turn0, p.rope_ratio);
                    }
                    const float sc = m ? 1.f : p.scale;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            uint2 pk;       // channels 32 db + 8 g4 + 4 hi .. + 3 of key l31 = half of 16-byte chunk 4 db + g4
                            pk.x = pack_bf16x2(gk[db][g4 * 4 + 0] * sc, gk[db][g4 * 4 + 1] * sc);
                            pk.y = pack_bf16x2(gk[db][g4 * 4 + 2] * sc, gk[db][g4 * 4 + 3] * sc);
                            *reinterpret_cast<uint2*>(ob + kb * 4096 + l31 * 128 + (((4 * db + g4) ^ (l31 & 7)) << 4) + hi * 8) = pk;
                        }
                }
                uint4 rw[8];
                {
                    const unsigned oa = lds0 + (unsigned)(B64_BOUNCE_OFF + wave * B64_TILE) + (unsigned)(lane >> 3) * 128u + ((unsigned)(lane & 7) << 4);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                                 "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(rw[0]), "=&v"(rw[1]), "=&v"(rw[2]), "=&v"(rw[3]), "=&v"(rw[4]), "=&v"(rw[5]), "=&v"(rw[6]), "=&v"(rw[7])
                                 : "v"(oa) : "memory");
                }
                const bf16_t* gw = m ? p.dV + (int64_t)cur.b * p.dv_sb + (int64_t)cur.h * p.dv_sh + (int64_t)cur.key0 * p.dv_sn
                                     : p.dK + (int64_t)cur.b * p.dk_sb + (int64_t)cur.h * p.dk_sh + (int64_t)cur.key0 * p.dk_sn;
                const int64_t sn = m ? p.dv_sn : p.dk_sn;
                const unsigned gbytes = rows > 0 ? (unsigned)(((rows - 1) * sn + 64) * 2) : 0u;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, (int)gbytes, 0x00020000);
                const int R = lane >> 3;
                const unsigned vo = (unsigned)(((int64_t)R * sn + (((lane & 7) ^ (R & 7)) * 8)) * 2);
                const unsigned o8 = (unsigned)(8 * sn * 2);
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 d = {rw[ps].x, rw[ps].y, rw[ps].z, rw[ps].w};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)vo, (int)(ps * o8), 0);
                }
            }
        }
        B64_TS(7);
        if (!has_next) break;
        // ---- the seam: the next item's K / V rows have been in the staging area since this item's start ----
        j += nslots;
        cur = nxt;
        has_next = j + nslots < items_x;
        if (has_next) nxt = make_item(j + nslots);
        // (the staging pieces were issued at the previous item's start and have been waited for by its tile loop's vmcnt(0))
        load_stationary();
        if (has_next) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area has been read
            issue_stage(nxt);
        }
        B64_TS(6);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef B64_TIMING
    if (tid == 0 && p.dbg) {
        for (int i = 0; i < 8; ++i) p.dbg[(size_t)blockIdx.x * 8 + i] = t_acc[i];
        p.dbg[65536 + 4 * blockIdx.x] = t_begin;
        p.dbg[65536 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        p.dbg[65536 + 4 * blockIdx.x + 2] = r_begin;
        p.dbg[65536 + 4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// =====================================================================================================================================
// attn_bwd_dq64_kernel — dQ with 64 queries per wave: the same construction turned around.  A wave owns 64 queries (two 32-query blocks);
// K / V tiles of 64 keys stream through the ring; per 32-key granule
//   phase A: 16 slots  S^T[qb] = K Q^T - lse2,  dP^T[qb] = V dO^T - delta   (swapped products: a query is a lane; the per-query scalars are
//                      register blocks that hold the lane's value 16 times — built once per item)
//   phase B:  8 slots  dQ^T[db][qb] += K^T dS^T of the PREVIOUS granule (K^T by transposing reads of the K tile)
//   beside them the granule's softmax, one pair per slot from slot 9 of phase A to slot 2 of the next granule's: P = exp2(S^T),
//   dS^T = P dP^T -> bf16 (P itself is not needed).
// 48 MFMAs and 16 row + 16 transposing fragment reads per tile and wave (the 32-query kernel: 24 and 16 + 16).  The wave's stationary
// operands (Q pre-multiplied by scale * log2(e), dO: B operands in accumulator registers), its dQ^T accumulators (accumulator registers)
// and its per-query scalars come out of an LDS staging area that holds the NEXT item's Q / dO / O rows and LSE from the item's start;
// delta = rowsum(dO * O) is computed there and left, with -lse2, in the scratch array for the dK / dV kernel.
// Key rows beyond Nk read as ZEROS (per-(batch, head) descriptors): whatever their scores, they add K^T dS^T = 0 — no mask.
#define D64_SLOT (2 * B64_TILE)
#define D64_STAGE_OFF (3 * D64_SLOT)
#define D64_STAGE_W (3 * B64_TILE + 256)         // per wave: Q rows | dO rows | O rows (the read-out's bounce area afterwards) | LSE[64]
#define D64_LDS_BYTES (D64_STAGE_OFF + 4 * D64_STAGE_W)       // 145 KiB

__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[D64_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(ab_lds_ptr_t)smem;
    const int nt = (p.Nk + 63) >> 6;

    const int g = blockIdx.x, xcd = g & 7, slot_w = g >> 3, nslots = (int)(gridDim.x >> 3);
    const int nbh = p.B * p.H, nqt = (p.Nq + 255) >> 8;
    const int items_x = ((nbh - xcd + 7) >> 3) * nqt;
    if (slot_w >= items_x) return;          // (uniform per workgroup)
    struct Item { unsigned long long kb; unsigned ov; int b, h, q0; };
    auto make_item = [&](int j) -> Item {
        Item it;
        const int kq = j / nqt, qt = j - kq * nqt;
        const int bh = kq * 8 + xcd;
        it.b = bh / p.H;
        it.h = bh - it.b * p.H;
        it.b = __builtin_amdgcn_readfirstlane(it.b);
        it.h = __builtin_amdgcn_readfirstlane(it.h);
        it.q0 = __builtin_amdgcn_readfirstlane(qt * 256 + wave * 64);
        it.kb = (unsigned long long)(p.K + (int64_t)it.b * p.k_sb + (int64_t)it.h * p.k_sh);
        it.ov = (unsigned)(((int64_t)it.b * p.v_sb + (int64_t)it.h * p.v_sh) * 2);
        return it;
    };

    // ---- the stream: per tile and wave 2 pieces of K rows, 2 of V rows (rows 16 w .. 16 w + 15) ----
    const unsigned kbytes = (unsigned)((((int64_t)p.Nk - 1) * p.k_sn + 64) * 2);        // key rows >= Nk read as zeros
    const p64_uint4_t srd_v = p64_make_srd(p.V, (unsigned)((((int64_t)p.B - 1) * p.v_sb + ((int64_t)p.H - 1) * p.v_sh + ((int64_t)p.Nk - 1) * p.v_sn + 64) * 2));
    unsigned voff_k[2], voff_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = i * 8 + (lane >> 3);
        const int cch = (lane & 7) ^ ((rr >> 1) & 7);
        voff_k[i] = (unsigned)(((int64_t)rr * p.k_sn + cch * 8) * 2);
        voff_v[i] = (unsigned)(((int64_t)rr * p.v_sn + cch * 8) * 2);
    }
    const unsigned kstep = (unsigned)(64 * p.k_sn * 2), vstep = (unsigned)(64 * p.v_sn * 2);
    const unsigned k16 = (unsigned)(16 * p.k_sn * 2) * (unsigned)wave, v16 = (unsigned)(16 * p.v_sn * 2) * (unsigned)wave;
    struct Pieces { p64_uint4_t srd_k; unsigned so_k, so_v, dst; };
    auto prep = [&](const Item& it, int t, int s) -> Pieces {
        Pieces pc;
        pc.srd_k = p64_make_srd((const void*)it.kb, kbytes);
        pc.so_k = __builtin_amdgcn_readfirstlane((unsigned)t * kstep + k16);
        pc.so_v = __builtin_amdgcn_readfirstlane(it.ov + (unsigned)t * vstep + v16);
        pc.dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(s * D64_SLOT) + (unsigned)wave * 2048u);
        return pc;
    };
    auto issue_piece = [&](const Pieces& pc, auto n_tag) __attribute__((always_inline)) {
        constexpr int n = decltype(n_tag)::value;
        if constexpr (n == 0) b64_dma16(voff_k[0], pc.srd_k, pc.so_k, pc.dst);
        else if constexpr (n == 1) b64_dma16(voff_k[1], pc.srd_k, pc.so_k, pc.dst + 1024);
        else if constexpr (n == 2) b64_dma16(voff_v[0], srd_v, pc.so_v, pc.dst + B64_TILE);
        else b64_dma16(voff_v[1], srd_v, pc.so_v, pc.dst + B64_TILE + 1024);
    };
    auto issue_all = [&](const Pieces& pc) __attribute__((always_inline)) {
        issue_piece(pc, P64Int<0>()); issue_piece(pc, P64Int<1>()); issue_piece(pc, P64Int<2>()); issue_piece(pc, P64Int<3>());
    };
    // the wave's 64 Q / dO / O rows and its 64 LSE values of an item into its staging area (query rows >= Nq read as zeros)
    auto issue_stage = [&](const Item& it) {
        const int64_t rows = min((int64_t)64, (int64_t)p.Nq - it.q0);
        const bf16_t* qb = p.Q + (int64_t)it.b * p.q_sb + (int64_t)it.h * p.q_sh + (int64_t)it.q0 * p.q_sn;
        const bf16_t* gb = p.dO + (int64_t)it.b * p.o_sb + (int64_t)it.h * p.o_sh + (int64_t)it.q0 * p.o_sn;
        const bf16_t* ob = p.O + (int64_t)it.b * p.o_sb + (int64_t)it.h * p.o_sh + (int64_t)it.q0 * p.o_sn;
        const float* lb = p.LSE + ((int64_t)it.b * p.H + it.h) * p.Nq + it.q0;
        const p64_uint4_t srd_q = p64_make_srd(qb, rows > 0 ? (unsigned)(((rows - 1) * p.q_sn + 64) * 2) : 0u);
        const p64_uint4_t srd_g = p64_make_srd(gb, rows > 0 ? (unsigned)(((rows - 1) * p.o_sn + 64) * 2) : 0u);
        const p64_uint4_t srd_o = p64_make_srd(ob, rows > 0 ? (unsigned)(((rows - 1) * p.o_sn + 64) * 2) : 0u);
        const p64_uint4_t srd_l = p64_make_srd(lb, rows > 0 ? (unsigned)(rows * 4) : 0u);
        const unsigned dst = lds0 + (unsigned)(D64_STAGE_OFF + wave * D64_STAGE_W);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const int cch = (lane & 7) ^ ((rr >> 1) & 7);
            const unsigned vq = (unsigned)(((int64_t)rr * p.q_sn + cch * 8) * 2), vo = (unsigned)(((int64_t)rr * p.o_sn + cch * 8) * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p64_dma16(vq, srd_q, (unsigned)((int64_t)(16 * k) * p.q_sn * 2), __builtin_amdgcn_readfirstlane(dst + k * 2048 + i * 1024));
                p64_dma16(vo, srd_g, (unsigned)((int64_t)(16 * k) * p.o_sn * 2), __builtin_amdgcn_readfirstlane(dst + B64_TILE + k * 2048 + i * 1024));
                p64_dma16(vo, srd_o, (unsigned)((int64_t)(16 * k) * p.o_sn * 2), __builtin_amdgcn_readfirstlane(dst + 2 * B64_TILE + k * 2048 + i * 1024));
            }
        }
        b64_dma4((unsigned)lane * 4u, srd_l, 0u, __builtin_amdgcn_readfirstlane(dst + 3 * B64_TILE));
    };

    int j = slot_w;
    Item cur = make_item(j);
    Item nxt = cur;
    bool has_next = j + nslots < items_x;
    if (has_next) nxt = make_item(j + nslots);
    issue_stage(cur);
    issue_all(prep(cur, 0, 0));
    issue_all(prep(cur, nt > 1 ? 1 : 0, 1));
    {   // ring slot 2 is read as "the tile before tile 0" by the first phase B (beside dS = 0): finite data
        uint4* z = reinterpret_cast<uint4*>(smem + 2 * D64_SLOT);
#pragma unroll
        for (int i = 0; i < B64_TILE / 16 / 256; ++i) z[i * 256 + tid] = make_uint4(0u, 0u, 0u, 0u);
    }

    p64_bf16x8_t qf[2][4], dof[2][4];      // stationary B operands: lane query = qb * 32 + l31, channels 16 st + 8 hi .. + 7; Q carries scale * log2(e)
    float16_t dq[2][2];                    // [channel block db][query block qb], transposed: row = channel, column = query
    float16_t S[2], dP[2];                 // [query block]: rows = the granule's 32 keys, column = the lane's query
    float16_t neg_lse[2], neg_dlt[2];      // the chains' start values: the lane's scalars, 16 times
    p64_bf16x8_t dS[2][2][2];              // [generation][query block][16-key slab]
    p64_bf16x8_t FA[8], F[4];
    float e0 = 0.f, e1 = 0.f, d0 = 0.f, d1 = 0.f;

    const char* ka[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) ka[st] = smem + bswz(l31, 2 * st + hi);
    const char* tr0[2];
    const char* tr1[2];
    {
        const int jj = lane & 15, piece = jj & 3;
        const int row0 = 4 * (lane >> 5) + (jj >> 2), row1 = row0 + 8;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int chunk = ((db * 32 + (((lane >> 4) & 1) << 4)) >> 3) + (piece >> 1);
            tr0[db] = smem + 2 * D64_SLOT + row0 * 128 + ((chunk ^ ((row0 >> 1) & 7)) << 4) + ((piece & 1) << 3);      // (starts at ring slot 2)
            tr1[db] = smem + 2 * D64_SLOT + row1 * 128 + ((chunk ^ ((row1 >> 1) & 7)) << 4) + ((piece & 1) << 3);
        }
    }
    int s_rd = 0;

    // the wave's stationary operands and scalars out of its staging area (landed); delta and -lse2 into the scratch array
    auto load_stationary = [&](const Item& it) __attribute__((always_inline)) {
        const char* sq = smem + D64_STAGE_OFF + wave * D64_STAGE_W;
        const float c = p.scale * 1.44269504088896340736f;
        float* ax = p.aux + ((int64_t)it.b * p.H + it.h) * 2 * p.nq_pad;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float dlt = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                union { p64_bf16x8_t v; unsigned u[4]; } a;
                a.v = *reinterpret_cast<const p64_bf16x8_t*>(sq + qb * 4096 + bswz(l31, 2 * st + hi));
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) a.u[q4] = pack_bf16x2(__uint_as_float(a.u[q4] << 16) * c, __uint_as_float(a.u[q4] & 0xffff0000u) * c);
                qf[qb][st] = a.v;
                const p64_bf16x8_t gf = *reinterpret_cast<const p64_bf16x8_t*>(sq + B64_TILE + qb * 4096 + bswz(l31, 2 * st + hi));
                const p64_bf16x8_t of = *reinterpret_cast<const p64_bf16x8_t*>(sq + 2 * B64_TILE + qb * 4096 + bswz(l31, 2 * st + hi));
                dof[qb][st] = gf;
#pragma unroll
                for (int e = 0; e < 8; ++e) dlt = fmaf((float)of[e], (float)gf[e], dlt);
            }
            dlt += __shfl_xor(dlt, 32, 64);
            const float lse2 = *reinterpret_cast<const float*>(sq + 3 * B64_TILE + (qb * 32 + l31) * 4) * 1.44269504088896340736f;
            const int q = it.q0 + qb * 32 + l31;
            if (hi == 0 && q < p.nq_pad) {      // (every query slot up to nq_pad gets its pair: the dK / dV kernel reads whole 64-query tiles of them)
                ax[q] = q < p.Nq ? -lse2 : -1e30f;
                ax[p.nq_pad + q] = q < p.Nq ? -dlt : 0.f;
            }
            float16_t nl, nd;
#pragma unroll
            for (int r = 0; r < 16; ++r) { nl[r] = -lse2; nd[r] = -dlt; }
            asm volatile("" : "+v"(nl), "+v"(nd));      // opaque: a splat hipcc recognises is re-materialised (16 v_mov) per use
            neg_lse[qb] = nl;
            neg_dlt[qb] = nd;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) dq[i][k] = (float16_t)(0.f);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) dS[1][qb][hf] = (p64_bf16x8_t)(0.f);       // generation 1 = "granule -1"
        // the item's first phase A finishes "granule -1"'s softmax (pairs 13..15, query block 1): exp2(-1e30) = 0, 0 * 0 = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[1][r] = -1e30f; dP[1][r] = 0.f; }
        e0 = 0.f; e1 = 0.f; d0 = 0.f; d1 = 0.f;
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_stationary(cur);
    if (has_next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_stage(nxt);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) FA[m] = *reinterpret_cast<const p64_bf16x8_t*>(ka[m >> 1] + (m & 1) * B64_TILE);

    // softmax pair q of a granule: query block q >> 3, accumulator rows 2 (q & 7), + 1 (keys).  A slot runs, in this order, the conversion of
    // pair C's products (computed a slot earlier), the products of pair F (exponentials a slot earlier), the exponentials of pair E.
    auto sm = [&](auto c_tag, auto f_tag, auto e_tag, auto gen_tag) __attribute__((always_inline)) {
        constexpr int C = decltype(c_tag)::value, Fq = decltype(f_tag)::value, E = decltype(e_tag)::value, G = decltype(gen_tag)::value;
        if constexpr (B64_DBG & 1) return;
        if constexpr (C >= 0) {
            union { p64_bf16x8_t v; unsigned u[4]; } d;
            d.v = dS[G][C >> 3][(C & 7) >> 2];
            d.u[C & 3] = p64_cvt_pk(d0, d1);
            dS[G][C >> 3][(C & 7) >> 2] = d.v;
        }
        if constexpr (Fq >= 0) {
            d0 = b64_mul(e0, dP[Fq >> 3][2 * (Fq & 7)]);
            d1 = b64_mul(e1, dP[Fq >> 3][2 * (Fq & 7) + 1]);
        }
        if constexpr (E >= 0) {
            e0 = p64_exp2(S[E >> 3][2 * (E & 7)]);
            e1 = p64_exp2(S[E >> 3][2 * (E & 7) + 1]);
        }
    };
    // ---- phase A of key granule kb of the current tile: slot k: query block k >> 3, chunk pair st = (k & 7) >> 1, even: S^T, odd: dP^T;
    //      row fragments FA[2 st] (K rows), FA[2 st + 1] (V rows) were read by the previous phase B.  Beside it: the tail of the previous
    //      granule's softmax (slots 0..3), the head of this one's (from slot 9), the first transposed fragments of phase B (slots 11, 13, 15),
    //      in the odd granule of a tile the four pieces of tile t + 2 (slots 4..7) ----
    auto phase_a = [&](auto kb_tag, const Pieces& pc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_tag)::value, pq = kb ^ 1;
        typedef P64Int<-1> N_;
        auto slot = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value, st = (k & 7) >> 1, qb = k >> 3;
            if constexpr (!(B64_DBG & 4) && k == 11) F[0] = b64_tr<4096 * pq>(tr0[0], tr1[0]);            // slab 2 pq, channel block 0
            if constexpr (!(B64_DBG & 4) && k == 13) F[1] = b64_tr<4096 * pq>(tr0[1], tr1[1]);            // slab 2 pq, channel block 1
            if constexpr (!(B64_DBG & 4) && k == 15) F[2] = b64_tr<4096 * pq + 2048>(tr0[0], tr1[0]);     // slab 2 pq + 1, channel block 0
            if constexpr (kb == 1 && !(B64_DBG & 16)) {
                if constexpr (k == 4) issue_piece(pc, P64Int<0>());
                if constexpr (k == 5) issue_piece(pc, P64Int<1>());
                if constexpr (k == 6) issue_piece(pc, P64Int<2>());
                if constexpr (k == 7) issue_piece(pc, P64Int<3>());
            }
            if constexpr ((k & 7) == 0) b64_mfma_first_b(S[qb], FA[0], qf[qb][0], neg_lse[qb]);
            else if constexpr ((k & 7) == 1) b64_mfma_first_b(dP[qb], FA[1], dof[qb][0], neg_dlt[qb]);
            else if constexpr ((k & 1) == 0) b64_mfma_acc_b(S[qb], FA[2 * st], qf[qb][st]);
            else b64_mfma_acc_b(dP[qb], FA[2 * st + 1], dof[qb][st]);
            if constexpr (k == 0) sm(P64Int<12>(), P64Int<13>(), P64Int<14>(), P64Int<kb ^ 1>());
            if constexpr (k == 1) sm(P64Int<13>(), P64Int<14>(), P64Int<15>(), P64Int<kb ^ 1>());
            if constexpr (k == 2) sm(P64Int<14>(), P64Int<15>(), N_(), P64Int<kb ^ 1>());
            if constexpr (k == 3) sm(P64Int<15>(), N_(), N_(), P64Int<kb ^ 1>());
            if constexpr (k == 9) sm(N_(), N_(), P64Int<0>(), P64Int<kb>());
            if constexpr (k == 10) sm(N_(), P64Int<0>(), P64Int<1>(), P64Int<kb>());
            if constexpr (k >= 11) sm(P64Int<k - 11>(), P64Int<k - 10>(), P64Int<k - 9>(), P64Int<kb>());
            P64_PIN();
        };
        slot(P64Int<0>()); slot(P64Int<1>()); slot(P64Int<2>()); slot(P64Int<3>());
        slot(P64Int<4>()); slot(P64Int<5>()); slot(P64Int<6>()); slot(P64Int<7>());
        slot(P64Int<8>()); slot(P64Int<9>()); slot(P64Int<10>()); slot(P64Int<11>());
        slot(P64Int<12>()); slot(P64Int<13>()); slot(P64Int<14>()); slot(P64Int<15>());
    };
    // ---- phase B behind phase A of granule kb: dQ^T += K^T dS^T of the previous granule (generation kb ^ 1; the tile tr0 / tr1 point at,
    //      slabs 2 pq, 2 pq + 1): slot k: fragment k >> 1 = (slab hf, channel block db), query block k & 1; the next granule's row fragments,
    //      one per slot, from nb_rows ----
    auto phase_b = [&](auto kb_tag, int nb_rows) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_tag)::value, GR = kb ^ 1, pq = kb ^ 1;
        typedef P64Int<-1> N_;
        auto slot = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value, pp = k >> 1, hf = pp >> 1, db = pp & 1, qb = k & 1;
            if constexpr (!(B64_DBG & 4) && k == 1) F[3] = b64_tr<4096 * pq + 2048>(tr0[1], tr1[1]);      // slab 2 pq + 1, channel block 1
            if constexpr (!(B64_DBG & 8)) FA[k] = *reinterpret_cast<const p64_bf16x8_t*>(ka[k >> 1] + nb_rows + (k & 1) * B64_TILE);
            b64_mfma_acc_a(dq[db][qb], F[pp], dS[GR][qb][hf]);
            if constexpr (k == 0) sm(P64Int<5>(), P64Int<6>(), P64Int<7>(), P64Int<kb>());
            if constexpr (k == 1) sm(P64Int<6>(), P64Int<7>(), N_(), P64Int<kb>());
            if constexpr (k == 2) sm(P64Int<7>(), N_(), P64Int<8>(), P64Int<kb>());
            if constexpr (k == 3) sm(N_(), P64Int<8>(), P64Int<9>(), P64Int<kb>());
            if constexpr (k >= 4) sm(P64Int<k + 4>(), P64Int<k + 5>(), P64Int<k + 6>(), P64Int<kb>());
            P64_PIN();
        };
        slot(P64Int<0>()); slot(P64Int<1>()); slot(P64Int<2>()); slot(P64Int<3>());
        slot(P64Int<4>()); slot(P64Int<5>()); slot(P64Int<6>()); slot(P64Int<7>());
    };
    auto advance_tr = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < 2; ++db) { tr0[db] += step; tr1[db] += step; }
    };

    for (;;) {
        for (int t = 0; t < nt; ++t) {
            const int nstep = s_rd == 2 ? -2 * D64_SLOT : D64_SLOT;        // ring step current -> next tile
            const int pstep = s_rd == 0 ? -2 * D64_SLOT : D64_SLOT;        // ring step previous -> current tile
            const int s_w = __builtin_amdgcn_readfirstlane(s_rd == 0 ? 2 : s_rd - 1);
            const Pieces pc = t + 2 < nt ? prep(cur, t + 2, s_w) : (has_next ? prep(nxt, t + 2 - nt, s_w) : prep(cur, nt - 1, s_w));
            phase_a(P64Int<0>(), pc);
            phase_b(P64Int<0>(), 4096);
            advance_tr(pstep);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            phase_a(P64Int<1>(), pc);
            phase_b(P64Int<1>(), nstep);
#pragma unroll
            for (int st = 0; st < 4; ++st) ka[st] += nstep;
            s_rd = s_rd == 2 ? 0 : s_rd + 1;
        }
        // ---- the tail of the last granule's softmax (generation 1), then its dQ^T (tile nt - 1, slabs 2, 3) ----
        {
            typedef P64Int<-1> N_;
            asm volatile("s_nop 0");
            sm(P64Int<12>(), P64Int<13>(), P64Int<14>(), P64Int<1>()); asm volatile("s_nop 0");
            sm(P64Int<13>(), P64Int<14>(), P64Int<15>(), P64Int<1>()); asm volatile("s_nop 0");
            sm(P64Int<14>(), P64Int<15>(), N_(), P64Int<1>()); asm volatile("s_nop 0");
            sm(P64Int<15>(), N_(), N_(), P64Int<1>());
        }
        p64_mfma_settle();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const p64_bf16x8_t tk = hf ? b64_tr<4096 + 2048>(tr0[db], tr1[db]) : b64_tr<4096>(tr0[db], tr1[db]);
                b64_mfma_acc_a(dq[db][0], tk, dS[1][0][hf]);
                b64_mfma_acc_a(dq[db][1], tk, dS[1][1][hf]);
            }
        p64_mfma_settle();

        // ---- read-out: inverse RoPE, scale, bf16, bounced through the O rows of the staging area (delta of the NEXT item, which needs
        //      them, is taken first: load_stationary below reads only Q and dO ... so the next item's O rows must be consumed before) ----
        // (order: the next item's stationary operands are loaded BEFORE the bounce overwrites its O rows; the read-out values wait in registers)
        float16_t out[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float16_t gq[2] = {dq[0][qb], dq[1][qb]};
            if (p.rope_qpos) {
                const int q = cur.q0 + qb * 32 + l31, qc = q < p.Nq ? q : p.Nq - 1;
                ab_rope_inverse(gq, p.rope_qpos + ((int64_t)cur.b * p.Nq + qc) * 2, hi, p.rope_turn0, p.rope_ratio);
            }
            out[0][qb] = gq[0];
            out[1][qb] = gq[1];
        }
        const Item done = cur;
        const bool more = has_next;
        if (more) {
            j += nslots;
            cur = nxt;
            has_next = j + nslots < items_x;
            if (has_next) nxt = make_item(j + nslots);
            load_stationary(cur);                                   // (its staging pieces were issued an item ago and waited for by the tile loop's vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staging area has been read
        }
        {
            char* ob = smem + D64_STAGE_OFF + wave * D64_STAGE_W + 2 * B64_TILE;
            const int64_t rows = min((int64_t)64, (int64_t)p.Nq - done.q0);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        uint2 pk;
                        pk.x = pack_bf16x2(out[db][qb][g4 * 4 + 0] * p.scale, out[db][qb][g4 * 4 + 1] * p.scale);
                        pk.y = pack_bf16x2(out[db][qb][g4 * 4 + 2] * p.scale, out[db][qb][g4 * 4 + 3] * p.scale);
                        *reinterpret_cast<uint2*>(ob + qb * 4096 + l31 * 128 + (((4 * db + g4) ^ (l31 & 7)) << 4) + hi * 8) = pk;
                    }
            uint4 rw[8];
            {
                const unsigned oa = lds0 + (unsigned)(D64_STAGE_OFF + wave * D64_STAGE_W + 2 * B64_TILE) + (unsigned)(lane >> 3) * 128u + ((unsigned)(lane & 7) << 4);
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                             "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(rw[0]), "=&v"(rw[1]), "=&v"(rw[2]), "=&v"(rw[3]), "=&v"(rw[4]), "=&v"(rw[5]), "=&v"(rw[6]), "=&v"(rw[7])
                             : "v"(oa) : "memory");
            }
            const bf16_t* gw = p.dQ + (int64_t)done.b * p.dq_sb + (int64_t)done.h * p.dq_sh + (int64_t)done.q0 * p.dq_sn;
            const unsigned gbytes = rows > 0 ? (unsigned)(((rows - 1) * p.dq_sn + 64) * 2) : 0u;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, (int)gbytes, 0x00020000);
            const int R = lane >> 3;
            const unsigned vo = (unsigned)(((int64_t)R * p.dq_sn + (((lane & 7) ^ (R & 7)) * 8)) * 2);
            const unsigned o8 = (unsigned)(8 * p.dq_sn * 2);
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 d = {rw[ps].x, rw[ps].y, rw[ps].z, rw[ps].w};
                __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)vo, (int)(ps * o8), 0);
            }
        }
        if (!more) break;
        if (has_next) issue_stage(nxt);         // (behind the bounce reads: lgkmcnt(0) inside the asm above)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
