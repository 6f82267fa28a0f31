// "TN" bf16 GEMM for gfx950 — the weight-gradient contraction over tokens / pixels, without transposing either operand:
//
//   C[I,J] (fp32) = sum_t A[t,i] * B[t,j]        A = dY [T,I] row-major,  B = X [T,J] row-major  (dW = dY^T X)
//   conv mode:     B[t,j] = act(x[b, oy*s-1+ky, ox*s-1+kx, c]),  t = (b,oy,ox), j = (ky*3+kx)*Cin + c   (implicit im2col)
//
// Both operands have the REDUCTION index t on the slow axis, so an MFMA operand (8 consecutive k per lane) is a column
// walk.  The tiles are staged [64 t][256 i|j] (row = 512 B) by the same lane-linear LDS-DMA as the NT kernel and read
// with gfx950's transposing LDS load (ds_read_b64_tr_b16): inside a 16-lane group lane q hands in the address of 4
// consecutive bf16 of row (q>>2) — together a [4 t][16 i] block — and receives column q of it, i.e. 4 consecutive t for
// its own i.  Two such reads (t, t+4) make one 16x16x32 operand.
// Bank layout: a wave-instruction is served in two 32-lane halves; a half reads rows {r..r+3} and {r+8..r+11} of one
// 32-byte column block, so the 32-byte block index is XOR-ed with s(r) = (r&3) | ((r>>3)&1)<<2 — eight distinct
// 8-bank groups.  As in the NT kernel the permutation is applied to the DMA's per-lane SOURCE address.
// Split-K over t: slice s stores its partial product to the slab C + s*I*J (plain stores; see uc_splitk_reduce).
#include "common.h"
#include "knobs.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* tn_lds_ptr_t;

struct TnParams {
    const bf16_t* A;     // [T, I], leading dim lda
    int64_t lda;
    const bf16_t* B;     // dense: [T, J] leading dim ldb; conv: NHWC image [cB, cH, cW, cCin]
    int64_t ldb;
    int64_t T, I, J;
    int conv;            // 0 dense, 1 implicit im2col of a 3x3 / pad 1 conv
    int cB, cH, cW, cCin, cStride, cHo, cWo, relu_b;
    uc_fastdiv dWo, dHo, dCin;   // exact fast division by cWo, cHo, cCin
    float* C;            // [split_k][I, J]
    float* colsum;       // optional: sum_t A[t,i] (the bias gradient) from the tj == 0 tiles: [split_k][I] slabs, or
    int colsum_atomic;   //           with colsum_atomic one [I] buffer that every K slice adds to atomically (+=)
    int split_k;
    int tiles_i, tiles_j;
};

__device__ uint4 g_tn_zero[2];   // 16 zero bytes (+ slack): DMA source for rows beyond T and for the conv's zero padding

__device__ __forceinline__ void tn_dma16(const void* gsrc, unsigned lds_byte_addr) {
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

__device__ __forceinline__ int tn_swz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__device__ __forceinline__ bf16x4_t tn_relu4(bf16x4_t v) {
    uint2 u = __builtin_bit_cast(uint2, v);
    unsigned* q = reinterpret_cast<unsigned*>(&u);
    typedef short tn_short2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 2; ++i)      // negative bf16 <=> negative int16: one v_pk_max_i16 per register
        q[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(tn_short2_t, q[i]), (tn_short2_t){0, 0}));
    return __builtin_bit_cast(bf16x4_t, u);
}

#define TN_BN 256

// BM_ x 256 x 64 workgroup tile (BM_ = 256: 4x4 waves, BM_ = 128: 2x4 waves — for I <= 128, e.g. the 128-channel convs of the
// DPT regressor, where a 256-row tile would be half empty); every wave owns 64 x 64 outputs.
template <int BM_, bool CONV>
__global__ __launch_bounds__(BM_ * 4) void gemm_tn_kernel(TnParams p) {
    constexpr int NW = BM_ / 64 * 4;                 // waves per workgroup
    constexpr int A_ROW = BM_ * 2;                   // bytes per t-row of the A tile (512 / 256); B rows are 512 B
    constexpr int A_TILE = 64 * A_ROW;
    constexpr int STAGE = A_TILE + 64 * TN_BN * 2;   // A tile then B tile
    constexpr int NI_A = A_TILE / 1024;              // 1-KiB DMA instructions of the A tile (32 / 16)
    constexpr int PER = (NI_A + 32) / NW;            // per wave (4 / 6)
    constexpr int RA = 1024 / A_ROW;                 // A rows per DMA instruction (2 / 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;              // (BM_/64) x 4 waves, 64 x 64 outputs each

    const int nwg = p.tiles_i * p.tiles_j;
    const int ksplit = (int)blockIdx.x / nwg;
    const int tile = (int)blockIdx.x - ksplit * nwg;
    const int ti = tile / p.tiles_j, tj = tile - ti * p.tiles_j;
    const int64_t i0 = (int64_t)ti * BM_, j0 = (int64_t)tj * TN_BN;

    // ---- DMA plan: NI_A + 32 one-KiB instructions per stage, PER per wave.  Instruction n < NI_A covers RA rows of the A
    //      tile (lane -> row n*RA + lane / (64/RA), physical 16-byte chunk lane % (64/RA)); the others 2 rows of the B tile.
    //      Per lane and instruction only the (clamped) logical column is kept; rows and conv taps are re-derived. ----
    const int d_ra = (int)((unsigned)lane / (64u / RA)), d_rb = (int)((unsigned)lane >> 5);   // lane's row inside an A / B instruction
#define TN_DROW(n_) ((n_) < NI_A ? (n_) * RA + d_ra : 2 * ((n_) - NI_A) + d_rb)
    int d_col[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int n = wave * PER + q;
        const int r = TN_DROW(n);
        const int pc = n < NI_A ? (int)((unsigned)lane % (64u / RA)) : (lane & 31);
        const int lc = ((((pc >> 1) ^ tn_swz(r)) << 1) | (pc & 1));      // logical 16-byte chunk held at physical chunk pc
        const int64_t lim = (n < NI_A) ? p.I : p.J;
        int64_t col = ((n < NI_A) ? i0 : j0) + lc * 8;
        if (col + 8 > lim) col = lim - 8;                                // duplicate a valid chunk; those outputs are never stored
        d_col[q] = (int)col;
    }
    const unsigned lds_base = (unsigned)(size_t)(tn_lds_ptr_t)smem;
    auto issue_stage = [&](int stage, int64_t t0) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int n = wave * PER + q;                                 // wave-uniform
            const int64_t t = t0 + __builtin_amdgcn_readfirstlane(n < NI_A ? n * RA : 2 * (n - NI_A)) + (n < NI_A ? d_ra : d_rb);
            const void* g = g_tn_zero;
            if (t < p.T) {
                if (n < NI_A) {
                    g = p.A + t * p.lda + d_col[q];
                } else if constexpr (!CONV) {
                    g = p.B + t * p.ldb + d_col[q];
                } else {
                    const int tap = (int)uc_div((unsigned)d_col[q], p.dCin);
                    const int c = d_col[q] - tap * p.cCin;
                    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;     // tap / 3 for tap in 0..8
                    const unsigned tt = (unsigned)t;                        // pixel counts fit 31 bits (checked by the launcher)
                    const unsigned row = uc_div(tt, p.dWo);
                    const int ox = (int)(tt - row * (unsigned)p.cWo);
                    const int b = (int)uc_div(row, p.dHo);
                    const int oy = (int)(row - (unsigned)b * (unsigned)p.cHo);
                    const int iy = oy * p.cStride - 1 + ky, ix = ox * p.cStride - 1 + kx;
                    if (iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW)
                        g = p.B + (((int64_t)b * p.cH + iy) * p.cW + ix) * p.cCin + c;
                }
            }
            tn_dma16(g, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + n * 1024)));
        }
    };

    // ---- transposing fragment reads: lane (g = lane>>4, q = lane&15); k-step ks, half h:
    //      row r = 32ks + 8g + 4h + (q>>2), 32-byte block blk (16 columns), piece (q&3)*8 bytes.
    //      s(r) = (q>>2) | (g&1)<<2 for every (ks,h), so one swizzled offset per fragment + immediates ----
    const int fg = lane >> 4, fq = lane & 15;
    const int f_sw = (fq >> 2) | ((fg & 1) << 2);
    const int f_row = 8 * fg + (fq >> 2);
    int a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = f_row * A_ROW + (fq & 3) * 8 + (((wr * 4 + i) ^ f_sw) << 5);
        b_off[i] = f_row * 512 + (fq & 3) * 8 + (((wc * 4 + i) ^ f_sw) << 5) + A_TILE;
    }

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk_total = (int)((p.T + 63) / 64);
    const int nk_per = (nk_total + p.split_k - 1) / p.split_k;
    const int kt0 = ksplit * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const int64_t tbase = (int64_t)kt0 * 64;

    typedef __attribute__((address_space(3))) bf16x4_t* lds_v4_t;
#define TN_FRAG(dst_, st_, off_, ks_, ROWB_, relu_)                                                                                 \
    do {                                                                                                                            \
        bf16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(tn_lds_ptr_t)((st_) + (off_) + (ks_) * (32 * (ROWB_))));                 \
        bf16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(tn_lds_ptr_t)((st_) + (off_) + (ks_) * (32 * (ROWB_)) + 4 * (ROWB_)));   \
        if (relu_) { lo_ = tn_relu4(lo_); hi_ = tn_relu4(hi_); }                                                                     \
        (dst_) = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                                                          \
    } while (0)

    // fused bias gradient: the waves of the first column tile that own A rows (wc == 0) also add up their A fragments
    // (v_dot2 against ones: 4 per fragment) — sum_t A[t,i] costs no extra pass over dY
    const bool do_colsum = p.colsum != nullptr && tj == 0 && wc == 0;   // wave-uniform
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const bf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};

    if (nk > 0) issue_stage(0, tbase);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue_stage((kt + 1) & 1, tbase + (int64_t)(kt + 1) * 64);
        const char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) TN_FRAG(af[i], st, a_off[i], ks, A_ROW, false);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 0, 1), ones2, csum[i], false);
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 2, 3), ones2, csum[i], false);
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 4, 5), ones2, csum[i], false);
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 6, 7), ones2, csum[i], false);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) TN_FRAG(bf[j], st, b_off[j], ks, 512, CONV && p.relu_b != 0);
            // swapped operands: D[row = j][col = i] -> a lane owns 4 consecutive j of one i (16-byte stores)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    if (do_colsum) {   // lanes fq, fq+16, fq+32, fq+48 hold the four k-groups of column i: fold them, lane group 0 stores
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int64_t ii = i0 + wr * 64 + 16 * i + fq;
            if (fg == 0 && ii < p.I) {
                if (p.colsum_atomic) unsafeAtomicAdd(p.colsum + ii, v);      // split_k adds per column: negligible contention
                else p.colsum[(int64_t)ksplit * p.I + ii] = v;
            }
        }
    }

    // ---- epilogue: lane (col = lane&15 -> i, rows 4*(lane>>4)+r -> j) ----
    float* slab = p.C + (int64_t)ksplit * p.I * p.J;
    const bool vec = (p.J % 4 == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t ii = i0 + wr * 64 + 16 * i + fq;
        if (ii >= p.I) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t jj = j0 + wc * 64 + 16 * j + 4 * fg;
            if (vec && jj + 3 < p.J) {
                *reinterpret_cast<float4_t*>(slab + ii * p.J + jj) = acc[i][j];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (jj + r < p.J) slab[ii * p.J + jj + r] = acc[i][j][r];
            }
        }
    }
}

// shapes the row-walking conv weight-gradient kernel takes (everything else: the implicit-im2col kernel)
static inline bool uc_conv_dw_rows_ok(int64_t Cout, int H, int W, int Cin, int stride) {
    return uc_knobs().conv_dw_rows && stride == 1 && W % 64 == 0 && Cin % 128 == 0 && Cout % 128 == 0 && H > 0;
}

extern "C" int uc_gemm_tn_conv_tiles(int64_t Cout, int conv_H, int conv_W, int conv_Cin, int conv_stride) {
    if (uc_conv_dw_rows_ok(Cout, conv_H, conv_W, conv_Cin, conv_stride)) return (int)(3 * (Cout / 128) * (conv_Cin / 128));
    const int64_t J = 9 * (int64_t)conv_Cin;
    return (int)(ceil_div64(Cout, Cout <= 128 ? 128 : 256) * ceil_div64(J, TN_BN));
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 / pad 1 / stride 1 conv on maps whose width is a multiple of 64 — the DPT head's big convs — WITHOUT
// the implicit im2col of the kernel above.  There a stage is 64 pixels x 256 im2col columns: every 16 bytes of it are located by
// a per-lane tap / pixel decode, the same input pixel is fetched nine times, and a wave gets 32 MFMAs per 6 LDS-DMA pieces (0.22 of the
// MFMA peak on the 128-channel convs of the regressor).  Here a workgroup owns ONE kernel row ky, 128 output channels (i) and 128 input
// channels (c), and walks 64-pixel segments of image rows: a stage is the segment's dY rows [64][128 i] and the 66 input pixels
// ox0 - 1 .. ox0 + 64 of input row oy + ky - 1 [66][128 c] — both plain row runs of NHWC tensors, 33 pieces — and the three taps kx
// are three MFMA passes over the SAME staged pixels, read one LDS row further along each time (the reduction index is the slow axis
// of the tile, so a pixel shift is a row offset): 48 MFMAs per wave and stage against 4 pieces.
//   grid = split_k x (3 ky) x (Cout / 128) x (Cin / 128);  C slab layout and the fused bias gradient are uc_gemm_tn's.
struct CdwParams {
    const bf16_t* A;     // dY [T, Cout], leading dim lda
    int64_t lda;
    const bf16_t* X;     // NHWC input [B, H, W, Cin]
    int H, W, Cin, relu_b;
    int64_t I, J;        // Cout, 9 * Cin
    int64_t nseg;        // B * H * (W / 64) segments of 64 output pixels
    uc_fastdiv dSpr, dH; // exact fast division by W / 64 and by H
    float* C;
    float* colsum;
    int colsum_atomic, split_k, tiles_i, tiles_c;
};

__global__ __launch_bounds__(512) void conv_dw_rows_kernel(CdwParams p) {
    constexpr int ROWB = 256, A_TILE = 64 * ROWB, B_ROWS = 68, STAGE = A_TILE + B_ROWS * ROWB, NST = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // 2 x 4 waves: 64 output channels x 32 input channels each, x 3 taps
    const int ntile = 3 * p.tiles_i * p.tiles_c;
    const int ksplit = (int)blockIdx.x / ntile;
    int tile = (int)blockIdx.x - ksplit * ntile;
    const int ky = tile % 3; tile /= 3;
    const int tc = tile % p.tiles_c, ti = tile / p.tiles_c;
    const int64_t i0 = (int64_t)ti * 128, c0 = (int64_t)tc * 128;

    const int64_t per = (p.nseg + p.split_k - 1) / p.split_k;
    const int64_t s0 = (int64_t)ksplit * per;
    const int nk = (int)max((int64_t)0, min(per, p.nseg - s0));

    // ---- DMA plan: piece n covers 4 rows of 256 B; lane -> row n * 4 + (lane >> 4), physical 16-byte chunk lane & 15, which holds
    //      the logical chunk with its 32-byte block index XOR-ed by tn_swz(row) (the permutation sits on the SOURCE side).
    //      A pieces 0..15 (wave w: 2w, 2w+1), X pieces 0..16 (wave w: 2w, 2w+1; wave 0 also 16 = rows 64..67, of which 64, 65 count) ----
    const int d_r = lane >> 4, d_pc = lane & 15;
    auto src_col = [&](int row) { return ((((d_pc >> 1) ^ tn_swz(row)) << 1) | (d_pc & 1)) * 8; };
    const unsigned lds_base = (unsigned)(size_t)(tn_lds_ptr_t)smem;
    auto issue_stage = [&](int stage, int64_t seg) {
        // segment -> (image row id = b * H + oy, ox0): wave-uniform
        const unsigned sg = (unsigned)seg;
        const unsigned rowid = uc_div(sg, p.dSpr);
        const int ox0 = (int)(sg - rowid * (unsigned)(p.W >> 6)) * 64;
        const unsigned b = uc_div(rowid, p.dH);
        const int oy = (int)(rowid - b * (unsigned)p.H);
        const int iy = oy + ky - 1;
        const bool row_ok = iy >= 0 && iy < p.H;
        const int64_t t0 = (int64_t)rowid * p.W + ox0;                                   // first output pixel of the segment
        const bf16_t* xrow = p.X + (((int64_t)b * p.H + iy) * p.W + (ox0 - 1)) * p.Cin + c0;   // input pixel ox0 - 1 of row iy (may be outside)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = wave * 2 + q;
            const int r = n * 4 + d_r;
            tn_dma16(p.A + (t0 + r) * p.lda + i0 + src_col(r), __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + n * 1024)));
        }
        const int nx = wave == 0 ? 3 : 2;
        for (int q = 0; q < nx; ++q) {
            const int n = q < 2 ? wave * 2 + q : 16;
            const int r = n * 4 + d_r;
            const int ix = ox0 - 1 + r;
            const void* g = g_tn_zero;
            if (row_ok && r < 66 && ix >= 0 && ix < p.W) g = xrow + (int64_t)r * p.Cin + src_col(r);
            tn_dma16(g, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + A_TILE + n * 1024)));
        }
    };

    // ---- transposing fragment reads (see gemm_tn_kernel): lane (fg = lane >> 4, fq = lane & 15) reads row 32 ks + 8 fg + 4 h + (fq >> 2)
    //      (+ kx for the input pixels), 8 bytes at (fq & 3) * 8 of the 32-byte block (16 columns) it owns a column of ----
    const int fg = lane >> 4, fq = lane & 15;
    const int f_row = 8 * fg + (fq >> 2);
    int a_off[4], b_off[3][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = f_row * ROWB + (fq & 3) * 8 + (((wr * 4 + i) ^ tn_swz(f_row)) << 5);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = f_row + 4 * h + kx;
                b_off[kx][h][j] = A_TILE + row * ROWB + (fq & 3) * 8 + (((wc * 2 + j) ^ tn_swz(row)) << 5);
            }

    float4_t acc[3][4][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[kx][i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((address_space(3))) bf16x4_t* lds_v4_t;
    const bool do_colsum = p.colsum != nullptr && ky == 1 && tc == 0 && wc == 0;   // wave-uniform
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const bf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};
    const bool relu = p.relu_b != 0;

    // 3-stage ring, DMA two segments ahead; a wave waits for its own pieces of the older stage only (4 or 5 pieces per stage)
    if (nk > 0) issue_stage(0, s0);
    if (nk > 1) issue_stage(1, s0 + 1);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int nxt = cur + 2; if (nxt >= NST) nxt -= NST;
        if (kt + 2 < nk) issue_stage(nxt, s0 + kt + 2);
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) TN_FRAG(af[i], st, a_off[i], ks, ROWB, false);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 0, 1), ones2, csum[i], false);
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 2, 3), ones2, csum[i], false);
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 4, 5), ones2, csum[i], false);
                    csum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(af[i], af[i], 6, 7), ones2, csum[i], false);
                }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8_t bf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bf16x4_t lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(tn_lds_ptr_t)(st + b_off[kx][0][j] + ks * (32 * ROWB)));
                    bf16x4_t hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_t)(tn_lds_ptr_t)(st + b_off[kx][1][j] + ks * (32 * ROWB)));
                    if (relu) { lo_ = tn_relu4(lo_); hi_ = tn_relu4(hi_); }
                    bf[j] = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[kx][i][j], 0, 0, 0);
            }
        }
        cur = (cur == NST - 1) ? 0 : cur + 1;
    }

    if (do_colsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int64_t ii = i0 + wr * 64 + 16 * i + fq;
            if (fg == 0) {
                if (p.colsum_atomic) unsafeAtomicAdd(p.colsum + ii, v);
                else p.colsum[(int64_t)ksplit * p.I + ii] = v;
            }
        }
    }
    // ---- epilogue: lane (col = fq -> output channel, rows 4 fg + r -> input channel) of tap (ky, kx) ----
    float* slab = p.C + (int64_t)ksplit * p.I * p.J;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ii = i0 + wr * 64 + 16 * i + fq;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t jj = (int64_t)(ky * 3 + kx) * p.Cin + c0 + wc * 32 + 16 * j + 4 * fg;
                *reinterpret_cast<float4_t*>(slab + ii * p.J + jj) = acc[kx][i][j];
            }
        }
}

extern "C" int uc_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t T, int64_t I, int64_t J, int conv_B,
                          int conv_H, int conv_W, int conv_Cin, int conv_stride, int relu_b, float* C, float* colsum_a,
                          int colsum_atomic, int split_k, uc_stream_t stream) {
    UC_REQUIRE(A && B && C, "uc_gemm_tn: null pointer");
    UC_REQUIRE(T > 0 && I >= 8 && J >= 8 && I % 8 == 0 && J % 8 == 0 && lda % 8 == 0, "uc_gemm_tn: I, J and lda must be multiples of 8");
    UC_REQUIRE(split_k >= 1 && split_k <= 1024, "uc_gemm_tn: bad split_k");
    UC_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), "uc_gemm_tn: operands must be 16-byte aligned");
    TnParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.T = T; p.I = I; p.J = J;
    p.conv = conv_B > 0 ? 1 : 0;
    p.cB = conv_B; p.cH = conv_H; p.cW = conv_W; p.cCin = conv_Cin; p.cStride = conv_stride; p.relu_b = relu_b ? 1 : 0;
    p.cHo = p.cWo = 0;
    if (p.conv) {
        UC_REQUIRE(conv_H > 0 && conv_W > 0 && conv_Cin > 0 && conv_Cin % 8 == 0 && conv_stride > 0, "uc_gemm_tn: bad conv geometry (Cin must be a multiple of 8)");
        p.cHo = (conv_H - 1) / conv_stride + 1;
        p.cWo = (conv_W - 1) / conv_stride + 1;
        UC_REQUIRE(T == (int64_t)conv_B * p.cHo * p.cWo && J == 9 * (int64_t)conv_Cin, "uc_gemm_tn: conv shape mismatch");
        UC_REQUIRE(T < (int64_t)1 << 31, "uc_gemm_tn: too many pixels");
        p.dWo = uc_make_fastdiv((unsigned)p.cWo); p.dHo = uc_make_fastdiv((unsigned)p.cHo); p.dCin = uc_make_fastdiv((unsigned)conv_Cin);
    } else {
        UC_REQUIRE(ldb % 8 == 0 && ldb >= J && !relu_b, "uc_gemm_tn: ldb must be a multiple of 8");
    }
    p.C = C; p.colsum = colsum_a; p.colsum_atomic = colsum_atomic ? 1 : 0; p.split_k = split_k;
    hipStream_t st = (hipStream_t)stream;
    if (p.conv && uc_conv_dw_rows_ok(I, conv_H, conv_W, conv_Cin, conv_stride) && lda % 8 == 0) {
        // stride-1 convs on maps a multiple of 64 wide with whole 128-channel tiles: one kernel row per workgroup, taps as row shifts
        CdwParams c;
        c.A = (const bf16_t*)A; c.lda = lda; c.X = (const bf16_t*)B; c.H = conv_H; c.W = conv_W; c.Cin = conv_Cin; c.relu_b = relu_b ? 1 : 0;
        c.I = I; c.J = J; c.nseg = (int64_t)conv_B * conv_H * (conv_W / 64);
        UC_REQUIRE(c.nseg < (int64_t)1 << 31, "uc_gemm_tn: too many pixels");
        c.dSpr = uc_make_fastdiv((unsigned)(conv_W / 64)); c.dH = uc_make_fastdiv((unsigned)conv_H);
        c.C = C; c.colsum = colsum_a; c.colsum_atomic = colsum_atomic ? 1 : 0; c.split_k = split_k;
        c.tiles_i = (int)(I / 128); c.tiles_c = conv_Cin / 128;
        constexpr int SMC = 3 * (64 * 256 + 68 * 256);
        static bool cattr = false;
        if (!cattr) {
            (void)hipFuncSetAttribute((const void*)conv_dw_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMC);
            cattr = true;
        }
        hipLaunchKernelGGL(conv_dw_rows_kernel, dim3((unsigned)(3 * c.tiles_i * c.tiles_c) * (unsigned)split_k), dim3(512), SMC, st, c);
        UC_CHECK_LAUNCH("uc_gemm_tn(conv rows)");
        return UC_OK;
    }
    const bool narrow = I <= 128;                   // half-height tile for the 128-row products (no half-empty MFMA tiles)
    p.tiles_i = (int)ceil_div64(I, narrow ? 128 : 256);
    p.tiles_j = (int)ceil_div64(J, TN_BN);
    constexpr int SM256 = 2 * (64 * 256 * 2 + 64 * TN_BN * 2), SM128 = 2 * (64 * 128 * 2 + 64 * TN_BN * 2);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SM256);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM256);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SM128);
        (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SM128);
        attr_set = true;
    }
    const dim3 grid((unsigned)p.tiles_i * p.tiles_j * (unsigned)split_k);
    if (narrow) {
        if (p.conv) hipLaunchKernelGGL((gemm_tn_kernel<128, true>), grid, dim3(512), SM128, st, p);
        else hipLaunchKernelGGL((gemm_tn_kernel<128, false>), grid, dim3(512), SM128, st, p);
    } else {
        if (p.conv) hipLaunchKernelGGL((gemm_tn_kernel<256, true>), grid, dim3(1024), SM256, st, p);
        else hipLaunchKernelGGL((gemm_tn_kernel<256, false>), grid, dim3(1024), SM256, st, p);
    }
    UC_CHECK_LAUNCH("uc_gemm_tn");
    return UC_OK;
}
