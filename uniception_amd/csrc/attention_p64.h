// attn_bf16_p64_kernel — persistent flash attention forward for head_dim 64 on gfx950 (round 5).
//
// Why another kernel.  The eight-wave kernel of attention.hip gives a wave 32 queries: per 64-key tile it issues 16 MFMAs
// (512 matrix-pipe cycles) and ~135 vector instructions (a row maximum, a scale-and-subtract, an exponential, a row sum and
// half a conversion per score), reads 16 LDS fragments for them, runs the two kinds one after the other, and pays the whole
// prologue / epilogue of a workgroup (Q, first K / VT tile from HBM; output store) every 16 tiles: 0.36 of the bf16 peak.
// This kernel changes the four things its anatomy priced:
//
//  * a wave owns 64 queries (two 32-query blocks A, B), so every K / VT fragment read from LDS feeds two MFMAs (16 reads per 32
//    MFMAs instead of 16 per 16); four waves per workgroup, two workgroups per CU = TWO WAVES PER SIMD at 256 registers each
//    (tools/probes/valu_rates.hip: the loop's instruction mix issues at 37 cycles per MFMA from one wave per SIMD and at 28 per
//    MFMA from two) — O (64) and the Q fragments (32) in named accumulator registers, scores / P / -m in VGPRs;
//  * the softmax diet — per score ONE v_exp_f32, one add and half a conversion: Q is pre-multiplied by scale * log2(e), the
//    score accumulators START at -m (the C operand of the first MFMA of a chain is a register block holding -m), and there is
//    no row maximum in the steady state: the running maximum is the maximum of the item's FIRST 32 keys, every later
//    exponential is taken against it (P up to 2^100 is as exact in bf16 as P <= 1, and the sums are fp32), and the row sum tells
//    at the end of the item whether some score outgrew it by more than 2^100 (~69 nats: then P or O may have overflowed).  Such a
//    64-query block gets a sentinel in its first output word and attn_bf16_fixup_kernel (attention.hip), launched behind every
//    call, recomputes it with the exact online softmax of attn_bf16_kernel — nothing else ever touches O inside the loop;
//  * software pipelining at 32-key granularity INSIDE the wave: region h issues QK^T of half-tile h+1, the exponentials of
//    half-tile h and PV of half-tile h-1 — 16 MFMAs beside ~80 vector instructions and 8 fragment reads per region, all in one
//    basic block, so the matrix pipe works under the vector work of the same wave (tools/probes/mfma_valu.hip: <= 5 vector
//    instructions per MFMA gap are free);
//  * persistent workgroups: a workgroup walks a list of (batch, head, query tile) items; K / VT tiles stream through a
//    three-slot LDS ring two iterations ahead ACROSS item seams (the next item's first tiles and its Q rows are in flight
//    under the current item's last tiles), so the per-item fixed cost is the epilogue's arithmetic, not three HBM round trips.
//
// Items are dealt so that the query tiles of one (batch, head) run on ONE XCD (workgroup g runs on XCD g % 8): K / VT of a
// head are pulled through one L2.
//
// Layouts are attention.hip's: Q, K [B, N, H, 64] bf16 by strides; V in the packed VT layout ([B, H, 64, npad], key order
// permuted inside 16-key groups, pads zero); S^T = K Q^T (swapped product) so that a query's scores sit in one lane pair.
#pragma once
#include "common.h"

typedef __bf16 p64_bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned p64_uint4_t __attribute__((ext_vector_type(4)));

struct AttnP64Params {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    float* lse;          // optional [B, H, Nq]
    int B, H, Nq, Nk;
    int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, o_sb, o_sn, o_sh;
    int npad;            // VT row length (Nk rounded up to 64)
    float c;             // scale * log2(e)
    int q_prescaled;     // Q already carries c (producer GEMM epilogue): no in-kernel multiply
    int nq;              // query tiles (256 queries) per (batch, head)
    uc_fastdiv dNq, dH;  // division by nq, H
    unsigned long long* dbg;   // probe builds: per-workgroup cycle stamps
};

#define P64_TILE 8192            // one K or VT tile: 64 rows x 128 B
#define P64_RING 3
#define P64_K_OFF 0
#define P64_V_OFF (P64_RING * P64_TILE)
#define P64_Q_OFF (2 * P64_RING * P64_TILE)             // 4 waves x 8 KiB of Q rows (next item)
#define P64_LDS_BYTES (P64_Q_OFF + 4 * P64_TILE)        // 80 KiB: two workgroups per CU

__device__ __forceinline__ int p64_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void p64_dma16(unsigned voff, p64_uint4_t srd, unsigned soff, unsigned lds_byte_addr) {
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
__device__ __forceinline__ p64_uint4_t p64_make_srd(const void* base, unsigned bytes) {
    const unsigned long long pa = (unsigned long long)base;
    return (p64_uint4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pa),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(pa >> 32) & 0xffffu)),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}

struct P64TagF { static constexpr bool value = false; };
struct P64TagT { static constexpr bool value = true; };
template <int I> struct P64Int { static constexpr int value = I; };
#define P64_L_MAX 1.2676506e30f        // 2^100: a row sum beyond this (inf and NaN included) ...
#define P64_L_MIN 7.8886091e-31f       // ... or below 2^-100 flags the wave's 64-query block for recomputation
#define P64_SENTINEL 0x7fc57fc5u       // two bf16 NaNs with a payload no arithmetic produces: "recompute this block"

// Register plan (256 VGPRs per lane, two waves per SIMD, NO accumulator registers: a kernel that uses any AGPR gets its budget split
// 128 / 128 by hipcc, and the scores, P and -m alone — which the vector unit reads and writes — are 112): O 64, scores 64, P 32, Q
// fragments 32, -m 16, fragment ring 16.  Every MFMA goes through inline asm (placed by hand; with builtins hipcc decides their
// order and, above 256 registers, their register file).  NOTHING but the MFMAs touches O between its zeroing and its read-out: a
// path that modifies it and re-joins the loop makes the allocator give O two homes and copy all 64 registers between them in every
// iteration.  What hipcc does not do for asm MFMAs (cdna_hip_programming.md 5.7): wait states between the last MFMA of a chain
// and the first VALU access to its result — the main loop places 8 PV MFMAs (>= 256 cycles) between them by construction, the
// other sites carry p64_mfma_settle().
__device__ __forceinline__ void p64_mfma_first(float16_t& d, p64_bf16x8_t a, p64_bf16x8_t b, const float16_t& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void p64_mfma_first0(float16_t& d, p64_bf16x8_t a, p64_bf16x8_t b) {       // C = 0: no zeroing pass
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void p64_mfma_acc(float16_t& d, p64_bf16x8_t a, p64_bf16x8_t b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// pinned single-instruction vector helpers of the hand-placed regions
__device__ __forceinline__ void p64_add(float& acc, float e) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc) : "v"(e)); }
__device__ __forceinline__ unsigned p64_cvt_pk(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float p64_exp2(float x) {
    float r;
    asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// ... of a score whose key may lie beyond Nk: P = 0 unless CST < lim (lim = Nk - first key of the lane's half-tile)
template <int CST>
__device__ __forceinline__ float p64_exp2_masked(float x, int lim) {
    float r;
    asm volatile("v_exp_f32 %0, %1\n\tv_cmp_lt_i32 vcc, %3, %2\n\ts_nop 0\n\tv_cndmask_b32 %0, 0, %0, vcc" : "=&v"(r) : "v"(x), "v"(lim), "i"(CST) : "vcc");
    return r;
}
#define P64_PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void p64_mfma_settle() {      // an asm MFMA's result -> VALU reader / writer: 8-pass XDL, 12 wait states
    P64_PIN();
    asm volatile("s_nop 7\n\ts_nop 4");
    P64_PIN();
}
#ifdef P64_TIMING     // probe builds: wave 0 of every workgroup accumulates s_memtime deltas per code section into p.dbg[8 g + section]
#define P64_TS(i) do { if (wave == 0) { P64_PIN(); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); t_acc[i] += n_ - t_last; t_last = n_; P64_PIN(); } } while (0)
#else
#define P64_TS(i) do { } while (0)
#endif

// RAGGED: the launch has a ragged last key tile (Nk % 64 != 0) — the two half-tiles of the last tile take the masked softmax.
// Needs Nk > 64 (two key tiles: the lookahead of two iterations then stays inside the next item).
template <bool RAGGED>
__global__ __launch_bounds__(256, 2) void attn_bf16_p64_kernel(AttnP64Params p) {
    __shared__ __attribute__((aligned(1024))) char smem[P64_LDS_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lane-derived constants are RE-DERIVED where the first / last iteration of an item needs them (fresh_lane: opaque to hipcc):
    // hoisted to kernel entry they are spilled around the main loop, and a spill reload carries a compiler-inserted vmcnt(0) that
    // waits for every DMA piece in flight
    auto fresh_lane = [&]() __attribute__((always_inline)) -> int {
        int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return l;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;

    // ---- this workgroup's item list: XCD x = g % 8 owns the (batch, head) pairs bh = 8 k + x; its items j = k * nq + qt are dealt
    //      round-robin to the XCD's workgroups ----
    const int g = blockIdx.x, xcd = g & 7, slot_w = g >> 3, nslots = (int)(gridDim.x >> 3);
    const int nbh = p.B * p.H;
    const int items_x = ((nbh - xcd + 7) >> 3) * p.nq;     // items of this XCD
    const int nt = (p.Nk + 63) >> 6;

    // per-lane DMA offsets: a piece is 8 rows x 128 B; row rr = 8 * piece + (lane >> 3); LOGICAL chunk stored at physical chunk lane & 7
    unsigned voff_k[2], voff_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = i * 8 + (lane >> 3);                 // pieces of equal parity share the swizzle term
        const int cch = (lane & 7) ^ ((rr >> 1) & 7);
        voff_k[i] = (unsigned)(((int64_t)rr * p.k_sn + cch * 8) * 2);
        voff_v[i] = (unsigned)(((int64_t)rr * p.npad + cch * 8) * 2);
    }
    const unsigned kstep = (unsigned)(64 * p.k_sn * 2);    // bytes between key tiles of K
    const unsigned k16 = (unsigned)(16 * p.k_sn * 2), v16 = (unsigned)(16 * p.npad * 2), q16 = (unsigned)(16 * p.q_sn * 2);
    const unsigned kbytes = (unsigned)((((int64_t)p.Nk - 1) * p.k_sn + 64) * 2);     // key rows >= Nk read as zeros
    const unsigned vbytes = (unsigned)((int64_t)64 * p.npad * 2);

    struct Item { unsigned long long kb, vb, qb; unsigned qbytes; int b, h, q0; };      // (scalars: the descriptors are rebuilt per issue)
    auto make_item = [&](int j) -> Item {
        Item it;
        const int kq = (int)uc_div((unsigned)j, p.dNq), qt = j - kq * p.nq;
        const int bh = kq * 8 + xcd;
        it.b = (int)uc_div((unsigned)bh, p.dH);
        it.h = bh - it.b * p.H;
        it.q0 = qt * 256 + wave * 64;
        it.kb = (unsigned long long)((const bf16_t*)p.K + (int64_t)it.b * p.k_sb + (int64_t)it.h * p.k_sh);
        it.vb = (unsigned long long)((const bf16_t*)p.V + ((int64_t)it.b * p.H + it.h) * 64 * (int64_t)p.npad);
        it.qb = (unsigned long long)((const bf16_t*)p.Q + (int64_t)it.b * p.q_sb + (int64_t)it.h * p.q_sh + (int64_t)it.q0 * p.q_sn);
        const int64_t q_rows = min((int64_t)64, (int64_t)p.Nq - it.q0);
        it.qbytes = q_rows > 0 ? (unsigned)(((q_rows - 1) * p.q_sn + 64) * 2) : 0u;
        return it;
    };
    // the four DMA pieces of one iteration's tiles (K tile tk, VT tile tv of an item into ring slot s): wave w carries pieces 2w, 2w+1
    // of each tile (rows 16w .. 16w+15).  Prepared as scalars at the head of an iteration, issued one per slot inside the regions.
    struct Pieces { p64_uint4_t srd_k, srd_v; unsigned so_k, so_v, dst_k, dst_v; };
    auto prep = [&](const Item& it, int i, int s) -> Pieces {       // iteration i of an item reads K(i + 1) (i <= nt - 2) and VT(i) (i >= 0)
        Pieces pc;
        const int tk = i + 1 <= nt - 1 ? i + 1 : 0, tv = i >= 0 ? i : 0;     // (dummy tiles keep the piece count per iteration at four)
        pc.srd_k = p64_make_srd((const void*)it.kb, kbytes);
        pc.srd_v = p64_make_srd((const void*)it.vb, vbytes);
        pc.so_k = (unsigned)tk * kstep + (unsigned)wave * k16;
        pc.so_v = (unsigned)tv * 128u + (unsigned)wave * v16;
        pc.dst_k = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(P64_K_OFF + s * P64_TILE + wave * 2048));
        pc.dst_v = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(P64_V_OFF + s * P64_TILE + wave * 2048));
        return pc;
    };
    auto issue_piece = [&](const Pieces& pc, auto n_tag) __attribute__((always_inline)) {
        constexpr int n = decltype(n_tag)::value;
        if constexpr (n == 0) p64_dma16(voff_k[0], pc.srd_k, pc.so_k, pc.dst_k);
        else if constexpr (n == 1) p64_dma16(voff_k[1], pc.srd_k, pc.so_k, pc.dst_k + 1024);
        else if constexpr (n == 2) p64_dma16(voff_v[0], pc.srd_v, pc.so_v, pc.dst_v);
        else p64_dma16(voff_v[1], pc.srd_v, pc.so_v, pc.dst_v + 1024);
    };
    auto issue_all = [&](const Pieces& pc) __attribute__((always_inline)) {
        issue_piece(pc, P64Int<0>()); issue_piece(pc, P64Int<1>()); issue_piece(pc, P64Int<2>()); issue_piece(pc, P64Int<3>());
    };
    auto issue_q = [&](const Item& it) {        // the wave's own 64 rows: 8 pieces into its private 8 KiB
        const unsigned dst = lds0 + (unsigned)(P64_Q_OFF + wave * P64_TILE);
        const p64_uint4_t srd = p64_make_srd((const void*)it.qb, it.qbytes);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = i * 8 + (lane >> 3);
            const unsigned vq = (unsigned)(((int64_t)rr * p.q_sn + ((lane & 7) ^ ((rr >> 1) & 7)) * 8) * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) p64_dma16(vq, srd, (unsigned)k * q16, __builtin_amdgcn_readfirstlane(dst + k * 2048 + i * 1024));
        }
    };

    if (slot_w >= items_x) return;          // (uniform per workgroup: no barrier is skipped by part of one)

    // fragment read offsets INSIDE the current ring slot: row l31 (+32 rows via +4096), chunk 2*st+hi (K rows, VT rows and Q rows
    // alike) + s_rd * P64_TILE; advanced with s_rd
    int ka[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) ka[st] = p64_swz(lane & 31, 2 * st + (lane >> 5));

    // ---- warm-up of the stream: Q and the tiles of iterations -1, 0 of the first item ----
    int j = slot_w;
    Item cur = make_item(j);
    Item nxt = cur;
    bool has_next = j + nslots < items_x;
    if (has_next) nxt = make_item(j + nslots);
    int s_rd = 0;            // ring slot of the current iteration; the lookahead writes s_rd + 2
#ifdef P64_TIMING
    unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_last, r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    issue_q(cur);
    issue_all(prep(cur, -1, 0));
    issue_all(prep(cur, 0, 1));
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // Q and ring slot 0 have landed (this wave's pieces)
    __syncthreads();

    float16_t S[2][2], negm;       // ONE stale maximum per lane for its two queries (l31 of block A and of block B)
    float16_t o[2][2];             // [query block][channel block]
    p64_bf16x8_t qf[2][4];
    p64_bf16x8_t P[2][2][2];       // [key block][query block][16-key slab]
    p64_bf16x8_t F[4];             // fragment ring of the main loop
    float m2, l[2];

    // ================================================================== building blocks ==========================================
    // compiler-scheduled forms (first and last iteration of an item)
    auto qk_plain = [&](int kb, auto zero_tag) __attribute__((always_inline)) {      // zero_tag: scores started at 0 (no maximum yet)
        constexpr bool ZERO = decltype(zero_tag)::value;
        const char* sk = smem + P64_K_OFF + kb * 4096;
        const p64_bf16x8_t k0 = *reinterpret_cast<const p64_bf16x8_t*>(sk + ka[0]);
        const p64_bf16x8_t k1 = *reinterpret_cast<const p64_bf16x8_t*>(sk + ka[1]);
        const p64_bf16x8_t k2 = *reinterpret_cast<const p64_bf16x8_t*>(sk + ka[2]);
        const p64_bf16x8_t k3 = *reinterpret_cast<const p64_bf16x8_t*>(sk + ka[3]);
        if constexpr (ZERO) {
            p64_mfma_first0(S[kb][0], k0, qf[0][0]);
            p64_mfma_first0(S[kb][1], k0, qf[1][0]);
        } else {
            p64_mfma_first(S[kb][0], k0, qf[0][0], negm);
            p64_mfma_first(S[kb][1], k0, qf[1][0], negm);
        }
        p64_mfma_acc(S[kb][0], k1, qf[0][1]);
        p64_mfma_acc(S[kb][1], k1, qf[1][1]);
        p64_mfma_acc(S[kb][0], k2, qf[0][2]);
        p64_mfma_acc(S[kb][1], k2, qf[1][2]);
        p64_mfma_acc(S[kb][0], k3, qf[0][3]);
        p64_mfma_acc(S[kb][1], k3, qf[1][3]);
    };
    auto pv_plain = [&](auto kb_tag) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_tag)::value;
        const char* sv = smem + P64_V_OFF;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const p64_bf16x8_t v0 = *reinterpret_cast<const p64_bf16x8_t*>(sv + ka[2 * kb + hf]);
            const p64_bf16x8_t v1 = *reinterpret_cast<const p64_bf16x8_t*>(sv + ka[2 * kb + hf] + 4096);
            p64_mfma_acc(o[0][0], v0, P[kb][0][hf]);
            p64_mfma_acc(o[1][0], v0, P[kb][1][hf]);
            p64_mfma_acc(o[0][1], v1, P[kb][0][hf]);
            p64_mfma_acc(o[1][1], v1, P[kb][1][hf]);
        }
    };
    // P = exp2(S[kb] - d) for both query blocks (SHIFT: d != 0), row sums into l
    auto softmax_plain = [&](int kb, int t, float d, int hi, auto shift_tag, auto mask_tag) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_tag)::value, SHIFT = decltype(shift_tag)::value;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float e[16];
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                e[r] = __builtin_amdgcn_exp2f(SHIFT ? S[kb][qb][r] - d : S[kb][qb][r]);
                if constexpr (MASK) e[r] = (t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi < p.Nk) ? e[r] : 0.f;
                if (r & 1) a1 += e[r]; else a0 += e[r];
            }
            l[qb] += a0 + a1;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                union { p64_bf16x8_t v; unsigned u[4]; } pk;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) pk.u[q4] = pack_bf16x2(e[hf * 8 + 2 * q4], e[hf * 8 + 2 * q4 + 1]);
                P[kb][qb][hf] = pk.v;
            }
        }
    };

    auto advance_slot = [&]() __attribute__((always_inline)) {
        const bool wrap = s_rd + 1 >= P64_RING;
        const int step = wrap ? -(P64_RING - 1) * P64_TILE : P64_TILE;
        s_rd = wrap ? 0 : s_rd + 1;
#pragma unroll
        for (int st = 0; st < 4; ++st) ka[st] += step;
    };
    auto wait_barrier = [&](bool q_issued) __attribute__((always_inline)) {
        if (q_issued) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
    };
    auto end_iteration = [&](bool q_issued) __attribute__((always_inline)) {
        wait_barrier(q_issued);
        advance_slot();
    };

    // One REGION of the main loop, hand-placed: 16 slots of { one MFMA, two exponentials, two row-sum adds, one conversion } with
    // the 8 fragment reads four slots ahead of their first use and the iteration's four DMA pieces in slots 9 / 11 (K pieces in
    // region A, VT pieces in region B); sched_barrier(0) after every slot keeps the order.
    //   slots 0-7   S[KQ][qb]  = K(KQ, st) Q(qb)^T (- m)      fragments 0-3 = K rows of key block KQ, chunk pair st
    //   slots 8-15  O[qb][db] += VT(KQ: slab hf, channels db) P[KQ][qb][hf]   fragments 4-7
    //   beside them  softmax of S[KQ ^ 1] -> P[KQ ^ 1]  (two scores per slot)
    // fragment f of the region lives in F[f & 3]: read in slot 2f - 4 (fragments 0, 1 of region B by region A: slots 12, 14; those
    // of region A right behind the barrier that made the tile visible, ahead of the iteration's scalar work), used in slots 2f, 2f + 1.
    auto region = [&](auto kq_tag, auto mask_tag, auto first_tag, int t_sm, const Pieces& pc, bool qi, int nstep) __attribute__((always_inline)) {
        constexpr int KQ = decltype(kq_tag)::value, KS = KQ ^ 1;
        constexpr bool MASK = decltype(mask_tag)::value, FIRST = decltype(first_tag)::value;     // FIRST: the item's first PV (O starts at 0)
        float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        float e0 = 0.f, e1 = 0.f;                                 // the slot's two exponentials, consumed by the next slot
        const int lim = MASK ? p.Nk - (t_sm * 64 + KS * 32 + 4 * (fresh_lane() >> 5)) : 0;
        auto slot = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            // ---- fragment read for two slots x 2 ahead ----
            if constexpr ((k & 1) == 0) {
                constexpr int f = (k >> 1) + 2;                   // fragment index, 2 .. 9 (8, 9 = region B's 0, 1)
                if constexpr (f < 4) F[f & 3] = *reinterpret_cast<const p64_bf16x8_t*>(smem + P64_K_OFF + KQ * 4096 + ka[f]);
                else if constexpr (f < 8) F[f & 3] = *reinterpret_cast<const p64_bf16x8_t*>(smem + P64_V_OFF + ((f - 4) & 1) * 4096 + ka[2 * KQ + ((f - 4) >> 1)]);
                else if constexpr (KQ == 0) F[f & 3] = *reinterpret_cast<const p64_bf16x8_t*>(smem + P64_K_OFF + 4096 + ka[f - 8]);
                else {
                    // region B, slots 12 / 14: the iteration's last LDS read of the current ring slot was slot 10 — the end-of-iteration
                    // wait + barrier sits HERE, and region A's first fragments of the NEXT tile are read under the last four MFMAs
                    if constexpr (k == 12) wait_barrier(qi);
                    F[f & 3] = *reinterpret_cast<const p64_bf16x8_t*>(smem + P64_K_OFF + nstep + ka[f - 8]);
                }
            }
            if constexpr (k == 9) issue_piece(pc, P64Int<2 * KQ>());
            if constexpr (k == 11) issue_piece(pc, P64Int<2 * KQ + 1>());
            // ---- the slot's MFMA ----
            if constexpr (k < 8) {
                constexpr int st = k >> 1, qb = k & 1;
                if constexpr (st == 0) p64_mfma_first(S[KQ][qb], F[st & 3], qf[qb][0], negm);
                else p64_mfma_acc(S[KQ][qb], F[st & 3], qf[qb][st]);
            } else {
                constexpr int kk = k - 8, f = 4 + (kk >> 1), hf = kk >> 2, db = (kk >> 1) & 1, qb = kk & 1;
                if constexpr (FIRST && hf == 0) p64_mfma_first0(o[qb][db], F[f & 3], P[KQ][qb][hf]);
                else p64_mfma_acc(o[qb][db], F[f & 3], P[KQ][qb][hf]);
            }
            // ---- vector work: finish the previous slot's pair, start this slot's (every instruction pinned: an SLP-packed or sunk
            //      add / conversion leaves the slots and runs with no MFMA beside it) ----
            if constexpr (k > 0) {
                constexpr int kp = k - 1, qbp = kp >> 3, idx = kp & 7;
                p64_add(acc[qbp][0], e0);
                p64_add(acc[qbp][1], e1);
                union { p64_bf16x8_t v; unsigned u[4]; } pk;
                pk.v = P[KS][qbp][idx >> 2];
                pk.u[idx & 3] = p64_cvt_pk(e0, e1);
                P[KS][qbp][idx >> 2] = pk.v;
            }
            {
                constexpr int qb = k >> 3, r0 = 2 * (k & 7);
                if constexpr (MASK) {
                    e0 = p64_exp2_masked<(r0 & 3) + 8 * (r0 >> 2)>(S[KS][qb][r0], lim);
                    e1 = p64_exp2_masked<((r0 + 1) & 3) + 8 * ((r0 + 1) >> 2)>(S[KS][qb][r0 + 1], lim);
                } else {
                    e0 = p64_exp2(S[KS][qb][r0]);
                    e1 = p64_exp2(S[KS][qb][r0 + 1]);
                }
            }
            P64_PIN();
        };
        slot(P64Int<0>()); slot(P64Int<1>()); slot(P64Int<2>()); slot(P64Int<3>());
        slot(P64Int<4>()); slot(P64Int<5>()); slot(P64Int<6>()); slot(P64Int<7>());
        slot(P64Int<8>()); slot(P64Int<9>()); slot(P64Int<10>()); slot(P64Int<11>());
        slot(P64Int<12>()); slot(P64Int<13>()); slot(P64Int<14>()); slot(P64Int<15>());
        {   // the last pair
            asm volatile("s_nop 0");                               // transcendental -> use
            p64_add(acc[1][0], e0);
            p64_add(acc[1][1], e1);
            union { p64_bf16x8_t v; unsigned u[4]; } pk;
            pk.v = P[KS][1][1];
            pk.u[3] = p64_cvt_pk(e0, e1);
            P[KS][1][1] = pk.v;
        }
        l[0] += acc[0][0] + acc[0][1];
        l[1] += acc[1][0] + acc[1][1];
        P64_PIN();
    };

    // the lookahead of iteration i: the tiles of iteration i + 2 (this item's, or the next item's first ones) as prepared pieces; the
    // next item's Q rows are issued on the spot (q_issued: 12 pieces in flight behind this iteration instead of 4)
    auto lookahead = [&](int i, bool& q_issued) __attribute__((always_inline)) -> Pieces {
        const int li = i + 2, s = s_rd + 2 >= P64_RING ? s_rd + 2 - P64_RING : s_rd + 2;
        q_issued = false;
        if (li <= nt - 1) return prep(cur, li, s);
        if (li == nt && has_next) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (two-tile items: this item's Q fragment reads have returned)
            issue_q(nxt);
            q_issued = true;
        }
        return prep(has_next ? nxt : cur, li - nt - 1, s);          // (no next item: harmless re-reads of this one)
    };
    // the wave's Q fragments out of its private Q rows (landed): the lane's rows l31 and 32 + l31, pre-multiplied by scale * log2(e)
    auto load_qf = [&]() __attribute__((always_inline)) {
        const char* sq = smem + P64_Q_OFF + wave * P64_TILE - s_rd * P64_TILE;      // (ka carries the ring slot)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int st = 0; st < 4; ++st) qf[qb][st] = *reinterpret_cast<const p64_bf16x8_t*>(sq + ka[st] + qb * 4096);
        if (!p.q_prescaled) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    union { p64_bf16x8_t v; unsigned u[4]; } a;
                    a.v = qf[qb][st];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        a.u[q4] = pack_bf16x2(__uint_as_float(a.u[q4] << 16) * p.c, __uint_as_float(a.u[q4] & 0xffff0000u) * p.c);
                    qf[qb][st] = a.v;
                }
        }
    };
    load_qf();

    // ================================================================== the item loop ============================================
    for (;;) {
        // ---- iteration -1: scores of tile 0; the first softmax (SETS the running maximum: the maximum over the item's first 32
        //      keys of the lane's two rows).  The Q fragments are in registers: loaded above / by the previous item's last iteration ----
        {
            bool qi;
            const Pieces pc = lookahead(-1, qi);
            l[0] = 0.f;
            l[1] = 0.f;
            P64_TS(8);
            asm volatile("s_nop 1");                               // VALU-written Q fragments -> MFMA operands
            qk_plain(0, P64TagT());                                // raw scores of the first 32 keys
            p64_mfma_settle();
            P64_TS(9);
            float mt = fmaxf(S[0][0][0], S[0][1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, fmaxf(S[0][0][r], S[0][1][r]));
            {   // the other half of the rows' keys sits in lane ^ 32
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));            // tile 0 is never the ragged one (Nk > 64): no mask
            }
            m2 = mt;
            {
                float16_t nm;
#pragma unroll
                for (int r = 0; r < 16; ++r) nm[r] = -mt;
                asm volatile("" : "+v"(nm));                      // opaque: a splat the compiler recognises is re-materialised per use
                negm = nm;
            }
            P64_TS(10);
            asm volatile("s_nop 1");                               // VALU-written -m -> MFMA operand
            qk_plain(1, P64TagF());
            softmax_plain(0, 0, mt, 0, P64TagT(), P64TagF());
            P64_TS(11);
            issue_all(pc);
            end_iteration(qi);
            F[0] = *reinterpret_cast<const p64_bf16x8_t*>(smem + P64_K_OFF + ka[0]);      // region A's first fragments (later: by region B)
            F[1] = *reinterpret_cast<const p64_bf16x8_t*>(smem + P64_K_OFF + ka[1]);
        }
        // ---- iterations 0 .. nt-2: region 2i+1 = { QK^T(i+1, kb0), softmax(i, kb1), PV(i, kb0) }, region 2i+2 = { QK^T(i+1, kb1),
        //      softmax(i+1, kb0), PV(i, kb1) }; ring slot s_rd holds K(i+1) and VT(i) ----
        P64_TS(4);
        auto body = [&](int i, auto mask_tag) __attribute__((always_inline)) {
            bool qi;
            const Pieces pc = lookahead(i, qi);
            const int nstep = s_rd + 1 >= P64_RING ? -(P64_RING - 1) * P64_TILE : P64_TILE;
            P64_TS(0);
            if (i == 0) region(P64Int<0>(), P64TagF(), P64TagT(), i, pc, qi, nstep);       // (softmax of (i, kb1): tile i < nt - 1 is whole)
            else region(P64Int<0>(), P64TagF(), P64TagF(), i, pc, qi, nstep);
            P64_TS(1);
            region(P64Int<1>(), mask_tag, P64TagF(), i + 1, pc, qi, nstep);               // (carries the iteration's wait + barrier)
            P64_TS(2);
            advance_slot();
            P64_TS(3);
        };
        // (the ragged launch runs its last iteration — the one whose second region takes the last tile's first half — as a copy
        // behind the loop: a masked and an unmasked region B side by side INSIDE the loop cost the allocator ~170 spills)
        const int n_plain = RAGGED ? nt - 2 : nt - 1;
        for (int i = 0; i < n_plain; ++i) body(i, P64TagF());
        if constexpr (RAGGED) body(nt - 2, P64TagT());
        // ---- iteration nt-1 (tail): PV(nt-1, kb0) beside softmax(nt-1, kb1); PV(nt-1, kb1); the next item's Q fragments; the outputs ----
        {
            bool qi;
            const Pieces pc = lookahead(nt - 1, qi);
            const int ln = fresh_lane(), hi = ln >> 5, l31 = ln & 31;
            p64_mfma_settle();
            pv_plain(P64Int<0>());
            if (RAGGED) softmax_plain(1, nt - 1, 0.f, hi, P64TagF(), P64TagT());
            else softmax_plain(1, nt - 1, 0.f, hi, P64TagF(), P64TagF());
            pv_plain(P64Int<1>());
            if (has_next) {
                // the next item's Q rows were issued an iteration ago, ahead of that iteration's 4 pieces: in order, they have landed
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                load_qf();                                          // (this item's are dead: the last score product is behind us)
            }
            P64_TS(12);
            // ---- normalise; bounce the wave's 64 x 64 outputs through its private Q rows (free now) so that a store instruction writes 8
            //      whole 128-byte rows: a per-lane store at a row stride touches 32 lines and the store path pays per line ----
            float inv[2];
            bool bad = false;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l[qb]), __float_as_uint(l[qb]), false, false);
                const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                inv[qb] = __builtin_amdgcn_rcpf(l_tot);
                bad = bad || !(l_tot <= P64_L_MAX && l_tot >= P64_L_MIN);
                const int q = cur.q0 + qb * 32 + l31;
                if (p.lse && q < p.Nq && hi == 0)
                    p.lse[((int64_t)cur.b * p.H + cur.h) * p.Nq + q] = m2 * 0.69314718055994530942f + logf(l_tot);
            }
            const bool flag = __any(bad);                          // some row of the wave left the range of its stale maximum: the block is recomputed
            p64_mfma_settle();                                     // the last PV MFMAs -> their first readers
            P64_TS(13);
            char* ob = smem + P64_Q_OFF + wave * P64_TILE;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the Q fragment reads above have returned
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        uint2 pk;       // channels 32db + 8g4 + 4hi .. +3 of query l31 = half of 16-B chunk 4db + g4
                        pk.x = pack_bf16x2(o[qb][db][g4 * 4 + 0] * inv[qb], o[qb][db][g4 * 4 + 1] * inv[qb]);
                        pk.y = pack_bf16x2(o[qb][db][g4 * 4 + 2] * inv[qb], o[qb][db][g4 * 4 + 3] * inv[qb]);
                        *reinterpret_cast<uint2*>(ob + qb * 4096 + l31 * 128 + (((4 * db + g4) ^ (l31 & 7)) << 4) + hi * 8) = pk;
                    }
            // all eight row groups out of LDS first (asm: hipcc splits a uint4 load it has to patch into dword pairs), then eight
            // buffer stores against a descriptor that ends with the wave's last valid row — no exec-masked branch per store
            uint4 rw[8];
            {
                const unsigned oa = (unsigned)(P64_Q_OFF + wave * P64_TILE) + (unsigned)(ln >> 3) * 128u + ((unsigned)(ln & 7) << 4);
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                             "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(rw[0]), "=&v"(rw[1]), "=&v"(rw[2]), "=&v"(rw[3]), "=&v"(rw[4]), "=&v"(rw[5]), "=&v"(rw[6]), "=&v"(rw[7])
                             : "v"(oa + lds0) : "memory");
            }
            if (flag && ln == 0) rw[0].x = P64_SENTINEL;                // (lane 0 of group 0: channels 0..7 of the block's first row)
            {
                const bf16_t* ow = (const bf16_t*)p.O + (int64_t)cur.b * p.o_sb + (int64_t)cur.h * p.o_sh + (int64_t)cur.q0 * p.o_sn;
                const int64_t o_rows = min((int64_t)64, (int64_t)p.Nq - cur.q0);
                const unsigned obytes = o_rows > 0 ? (unsigned)(((o_rows - 1) * p.o_sn + 64) * 2) : 0u;
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ow, 0, (int)obytes, 0x00020000);
                const int R = ln >> 3;
                const unsigned vo = (unsigned)(((int64_t)R * p.o_sn + (((ln & 7) ^ (R & 7)) * 8)) * 2);
                const unsigned o8 = (unsigned)(8 * p.o_sn * 2);
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 d = {rw[ps].x, rw[ps].y, rw[ps].z, rw[ps].w};
                    __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, (int)vo, (int)(ps * o8), 0);
                }
            }
            issue_all(pc);              // (behind the stores: a spill reload above would make hipcc wait for everything in flight)
            // the seam: everything the next item's first iteration reads has to have landed
            P64_TS(5);
            // (vmcnt retires in order: the 8 stores and this iteration's 4 pieces may stay in flight)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __syncthreads();
            advance_slot();
            P64_TS(6);
        }
        if (!has_next) break;
        j += nslots;
        cur = nxt;
        has_next = j + nslots < items_x;
        if (has_next) nxt = make_item(j + nslots);
        P64_TS(7);
    }
#ifdef P64_TIMING
    if (tid == 0 && p.dbg) {
        for (int i = 0; i < 16; ++i) p.dbg[(size_t)g * 16 + i] = t_acc[i];
        p.dbg[8192 + 4 * g] = t_begin;
        p.dbg[8192 + 4 * g + 1] = __builtin_amdgcn_s_memtime();
        p.dbg[8192 + 4 * g + 2] = r_begin;
        p.dbg[8192 + 4 * g + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
