// GEMM family for gfx950:  C[M,N] = epilogue( A_op[M,K] . W[N,K]^T )
//
//  * gemm_bf16_kernel  — bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//      128x128x64 workgroup tile, 4 wavefronts (2x2), each wave a 64x64 sub-tile = 4x4 MFMA fragments.
//      A/W tiles are staged global -> VGPR -> LDS (16 B per lane, issued one K-step ahead so the HBM
//      latency hides under the MFMAs of the current step), LDS double-buffered, one barrier per K-step.
//      LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 so the four
//      16-lane groups of a ds_read_b128 fragment load hit 16 distinct 4-bank slots (conflict-free).
//      The A loader is a functor: dense rows, or an implicit-GEMM gather for 3x3 convolutions over an
//      NHWC image (zero padding by predication), optionally applying ReLU on the fly.
//  * gemm_f32_kernel   — fp32 operands, one k-ordered fmaf chain per output (bit-comparable with a
//      scalar fp32 reference up to summation order inside the reference BLAS); used by the fp32
//      verification mode.  64x64x16 tile, 4x4 outputs per thread.
//
// The epilogue (bias -> activation -> fused RoPE-2D -> residual -> store) is shared.
// With the 16x16 MFMA C layout (col = lane&15, row = 4*(lane>>4)+reg) and a 64-column wave tile
// aligned to a 64-wide head, the RoPE partner of channel d (< 16) is channel d+16 of the same
// row: fragment ni and ni+1 of the SAME lane and register, so the rotation needs no cross-lane traffic.
#include "common.h"
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>
#include "gemm_glds.h"
#include "knobs.h"
#include <stdlib.h>
#include <algorithm>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct GemmParams {
    const void* A;
    int64_t lda;
    const void* W;
    int64_t M, N, K;
    int relu_a;
    int cB, cH, cW, cCin, cStride, cHo, cWo;
    const float* bias;
    int act;
    const void* residual;
    const void* residual2;
    int res_dtype;
    int64_t ldr;
    int64_t rope_cols;
    const int64_t* rope_pos;
    const float2* rope_table;
    int rope_npos;
    int64_t vt_col0;
    bf16_t* vt_out;
    int vt_ntok, vt_npad;
    void* C;
    int out_dtype;
    int64_t ldc;
    int tiles_m, tiles_n;
    void* preact;   // fp32 kernel only (the bf16 path carries it in GldsParams)
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == UC_ACT_GELU_ERF) return gelu_erf(v);
    if (act == UC_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

__device__ __forceinline__ float load_res(const void* r, int dtype, int64_t idx) {
    return dtype == UC_F32 ? ((const float*)r)[idx] : (dtype == UC_F16 ? f16_to_f32(((const bf16_t*)r)[idx]) : bf16_to_f32(((const bf16_t*)r)[idx]));
}

__device__ __forceinline__ void store_out(void* c, int dtype, int64_t idx, float v) {
    if (dtype == UC_F32) ((float*)c)[idx] = v;
    else if (dtype == UC_F16) ((bf16_t*)c)[idx] = f32_to_f16(v);
    else ((bf16_t*)c)[idx] = f32_to_bf16(v);
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each XCD a
// contiguous run of tiles so tiles that share an A row-panel / W column-panel share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// =======================================================================================
// bf16 MFMA kernel
// =======================================================================================
#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
// LDS: 2 stages x (A 128x64 + W 128x64) bf16 = 2 x 32 KiB
#define TILE_BYTES (BM * BK * 2)

__device__ __forceinline__ int swz_off(int row, int chunk) {  // byte offset inside a 128-row x 128-B tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ uint4 relu_bf16x8(uint4 v) {
    unsigned* p = reinterpret_cast<unsigned*>(&v);
    typedef short short2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i)      // negative bf16 <=> negative int16: one v_pk_max_i16 per register
        p[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2_t, p[i]), (short2_t){0, 0}));
    return v;
}

// F16: fp16 operands (v_mfma_f32_16x16x32_f16) and fp16 outputs / residuals — the fallback of the heads' TF32-class mode for channel
// counts the direct-to-LDS kernels do not take (small test models); bias / activation / residuals only.
typedef _Float16 gemm_f16x8_t __attribute__((ext_vector_type(8)));
template <int A_MODE, bool F16 = false>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int nwg = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    // tile order inside an XCD run: n fastest, so a run shares A row panels and sweeps W
    const int tm = t / p.tiles_n, tn = t % p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int64_t n0 = (int64_t)tn * BN;

    // ---- staging assignment: this thread moves chunk column cc of rows (tid>>3) + 32*i, i<4 ----
    const int cc = tid & 7;
    const int r0 = tid >> 3;
    const bf16_t* Ab = (const bf16_t*)p.A;
    const bf16_t* Wb = (const bf16_t*)p.W;

    int64_t a_row_off[4];   // dense: row*lda.  conv: ((b*H + oy*s-1)*W + ox*s-1) * Cin (may be "negative": guarded by iy/ix tests)
    int a_iy0[4], a_ix0[4]; // conv: top-left input coordinate of the 3x3 window
    const bf16_t* w_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m0 + r0 + 32 * i;
        if (m >= p.M) m = p.M - 1;  // clamp: rows beyond M are computed on valid data and never stored
        if (A_MODE == UC_A_DENSE) {
            a_row_off[i] = m * p.lda;
            a_iy0[i] = a_ix0[i] = 0;
        } else {
            const int ox = (int)(m % p.cWo);
            const int oy = (int)((m / p.cWo) % p.cHo);
            const int b = (int)(m / ((int64_t)p.cWo * p.cHo));
            a_iy0[i] = oy * p.cStride - 1;
            a_ix0[i] = ox * p.cStride - 1;
            a_row_off[i] = (int64_t)b * p.cH * p.cW;  // image base in pixels
        }
        int64_t n = n0 + r0 + 32 * i;
        if (n >= p.N) n = p.N - 1;
        w_row[i] = Wb + n * p.K;
    }

    uint4 ra[4], rw[4];
    auto stage_load = [&](int64_t k0) {
        const int64_t kk = k0 + cc * 8;
        const bool kin = kk < p.K;
        if (A_MODE == UC_A_DENSE) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ra[i] = kin ? *reinterpret_cast<const uint4*>(Ab + a_row_off[i] + kk) : make_uint4(0, 0, 0, 0);
        } else {
            const int tap = (int)(kk / p.cCin);
            const int ch = (int)(kk % p.cCin);
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                const bool ok = kin && iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW;
                ra[i] = ok ? *reinterpret_cast<const uint4*>(Ab + (a_row_off[i] + (int64_t)iy * p.cW + ix) * p.cCin + ch)
                           : make_uint4(0, 0, 0, 0);
            }
        }
        if (p.relu_a) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = relu_bf16x8(ra[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rw[i] = kin ? *reinterpret_cast<const uint4*>(w_row[i] + kk) : make_uint4(0, 0, 0, 0);
    };
    auto stage_write = [&](int buf) {
        char* sa = smem + buf * 2 * TILE_BYTES;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sa + swz_off(row, cc)) = ra[i];
            *reinterpret_cast<uint4*>(sw + swz_off(row, cc)) = rw[i];
        }
    };

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (int)((p.K + BK - 1) / BK);
    stage_load(0);
    stage_write(0);
    __syncthreads();

    const int frow = lane & 15;   // fragment row (A) / column (W) inside a 16-wide fragment
    const int fk = lane >> 4;     // k-group: 8 consecutive k per lane

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage_load((int64_t)(kt + 1) * BK);
        const char* sa = smem + buf * 2 * TILE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + fk;
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8_t*>(sa + swz_off(row, chunk));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + j * 16 + frow;
                wf[j] = *reinterpret_cast<const bf16x8_t*>(sw + swz_off(row, chunk));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gemm_f16x8_t, af[i]), __builtin_bit_cast(gemm_f16x8_t, wf[j]), acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ------------------------------- epilogue -------------------------------------------
    const int64_t wave_m = m0 + wr * 64;
    const int64_t wave_n = n0 + wc * 64;
    const int ecol = lane & 15;
    const int erow = (lane >> 4) * 4;
    const bool do_rope = p.rope_cols > 0 && wave_n < p.rope_cols;

    float bcol[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t n = wave_n + j * 16 + ecol;
        bcol[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
    if (p.vt_col0 >= 0 && wave_n >= p.vt_col0) {
        // V columns: write transposed + key-permuted ("VT", see uc_hip.h) so the attention kernel's PV
        // operand is a plain 16-byte LDS read.  A lane owns 4 consecutive tokens (reg 0..3) of one channel.
        if (wave_n >= p.N) return;
        const int head = (int)((wave_n - p.vt_col0) >> 6);
        const int nheads = (int)((p.N - p.vt_col0) >> 6);
        const bool aligned = (p.vt_ntok & 15) == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t mb = wave_m + i * 16 + erow;   // first of this lane's 4 rows
            if (aligned) {
                if (mb >= p.M) continue;
                const int b = (int)(mb / p.vt_ntok);
                const int tok = (int)(mb % p.vt_ntok);
                const int g = lane >> 4;
                const int pos = (tok & ~15) + ((g & 1) << 3) + ((g >> 1) << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = j * 16 + ecol;
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
                    for (int j = 0; j < 4; ++j) {
                        const int d = j * 16 + ecol;
                        p.vt_out[(((int64_t)b * nheads + head) * 64 + d) * p.vt_npad + pos] =
                            f32_to_bf16(acc[i][j][r] + bcol[j]);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = wave_m + i * 16 + erow + r;
            if (m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = apply_act(acc[i][j][r] + bcol[j], p.act);
            if (do_rope) {
                int py = (int)p.rope_pos[m * 2 + 0];
                int px = (int)p.rope_pos[m * 2 + 1];
                py = min(max(py, 0), p.rope_npos - 1);
                px = min(max(px, 0), p.rope_npos - 1);
                const float2 cy = p.rope_table[py * 16 + ecol];
                const float2 cx = p.rope_table[px * 16 + ecol];
                const float u0 = v[0], v0 = v[1], u1 = v[2], v1 = v[3];
                v[0] = u0 * cy.x - v0 * cy.y;
                v[1] = v0 * cy.x + u0 * cy.y;
                v[2] = u1 * cx.x - v1 * cx.y;
                v[3] = v1 * cx.x + u1 * cx.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = wave_n + j * 16 + ecol;
                if (n >= p.N) continue;
                float o = v[j];
                if (p.residual) o += load_res(p.residual, p.res_dtype, m * p.ldr + n);
                if (p.residual2) o += load_res(p.residual2, p.res_dtype, m * p.ldr + n);
                store_out(p.C, p.out_dtype, m * p.ldc + n, o);
            }
        }
    }
}

// =======================================================================================
// fp32 kernel (verification mode): exact fp32 products, k-ascending fmaf chain.
// =======================================================================================
#define FBM 64
#define FBN 64
#define FBK 16

template <int A_MODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    __shared__ float As[FBK][FBM + 4];
    __shared__ float Ws[FBK][FBN + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
    const int64_t m0 = (int64_t)tm * FBM, n0 = (int64_t)tn * FBN;
    const float* Af = (const float*)p.A;
    const float* Wf = (const float*)p.W;

    // staging: thread loads A[m0 + (tid>>2)][k0 + (tid&3)*4 .. +3] and the same for W
    const int lrow = tid >> 2;
    const int lk = (tid & 3) * 4;
    int64_t am = m0 + lrow;
    const bool am_ok = am < p.M;
    if (!am_ok) am = p.M - 1;
    int64_t wn = n0 + lrow;
    const bool wn_ok = wn < p.N;
    if (!wn_ok) wn = p.N - 1;
    int a_iy0 = 0, a_ix0 = 0;
    int64_t a_base = 0;
    if (A_MODE == UC_A_DENSE) {
        a_base = am * p.lda;
    } else {
        const int ox = (int)(am % p.cWo);
        const int oy = (int)((am / p.cWo) % p.cHo);
        const int b = (int)(am / ((int64_t)p.cWo * p.cHo));
        a_iy0 = oy * p.cStride - 1;
        a_ix0 = ox * p.cStride - 1;
        a_base = (int64_t)b * p.cH * p.cW;
    }

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int64_t k0 = 0; k0 < p.K; k0 += FBK) {
        float av[4], wv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t kk = k0 + lk + e;
            float a = 0.f, w = 0.f;
            if (kk < p.K) {
                if (A_MODE == UC_A_DENSE) {
                    a = Af[a_base + kk];
                } else {
                    const int tap = (int)(kk / p.cCin), ch = (int)(kk % p.cCin);
                    const int iy = a_iy0 + tap / 3, ix = a_ix0 + tap % 3;
                    if (iy >= 0 && iy < p.cH && ix >= 0 && ix < p.cW)
                        a = Af[(a_base + (int64_t)iy * p.cW + ix) * p.cCin + ch];
                }
                if (p.relu_a) a = fmaxf(a, 0.f);
                w = Wf[wn * p.K + kk];
            }
            av[e] = a;
            wv[e] = w;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[lk + e][lrow] = av[e];
            Ws[lk + e][lrow] = wv[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FBK; ++k) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = Ws[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float v = acc[i][j] + (p.bias ? p.bias[n] : 0.f);
            if (p.preact) store_out(p.preact, p.out_dtype, m * p.ldc + n, v);
            v = apply_act(v, p.act);
            if (p.residual) v += load_res(p.residual, p.res_dtype, m * p.ldr + n);
            if (p.residual2) v += load_res(p.residual2, p.res_dtype, m * p.ldr + n);
            store_out(p.C, p.out_dtype, m * p.ldc + n, v);
        }
    }
}

// Hand-over buffer of the small-M path (fuse_split2): caller-provided through uc_gemm_desc.fuse_ws (ABI 11) — 128 tiles x 128 x 128 fp32
// partial sums followed by 128 flag words.  UNCACHED device memory: the partners of a split tile may run on different XCDs (separate,
// non-coherent L2s).  Round 3 kept a lazily grown, never-freed pool of such buffers in here, keyed by (device, stream, capture id):
// allocation and a synchronous memset inside a launch path, state a C caller could neither size nor release — removed.
static constexpr int64_t UC_FUSE_TILES = 128;
static constexpr int64_t UC_FUSE_WS_FLOATS = UC_FUSE_TILES * 128 * 128;
extern "C" int64_t uc_gemm_fuse_ws_bytes(void) { return UC_FUSE_WS_FLOATS * (int64_t)sizeof(float) + UC_FUSE_TILES * (int64_t)sizeof(unsigned); }

extern "C" int uc_gemm(const uc_gemm_desc* d, uc_stream_t stream) {
    UC_REQUIRE(d, "uc_gemm: null descriptor");
    UC_REQUIRE(d->A && d->W && (d->C || d->tail_out), "uc_gemm: null operand pointer");
    UC_REQUIRE(d->M >= 0 && d->N > 0 && d->K > 0, "uc_gemm: bad shape M=%lld N=%lld K=%lld", (long long)d->M,
               (long long)d->N, (long long)d->K);
    UC_REQUIRE(d->a_mode == UC_A_DENSE || d->a_mode == UC_A_CONV3X3, "uc_gemm: bad a_mode %d", d->a_mode);
    const bool f16 = d->compute_dtype == UC_F16;   // fp16 operands (TF32-class heads): the bf16 MFMA path with the f16 instruction
    const int lo16 = f16 ? UC_F16 : UC_BF16;        // the 16-bit storage dtype that goes with the operands
    UC_REQUIRE(d->out_dtype == UC_F32 || d->out_dtype == lo16, "uc_gemm: bad out_dtype %d for compute dtype %d", d->out_dtype, d->compute_dtype);
    UC_REQUIRE(d->act >= UC_ACT_NONE && d->act <= UC_ACT_RELU, "uc_gemm: bad act %d", d->act);
    UC_REQUIRE(d->residual || !d->residual2, "uc_gemm: residual2 without residual");
    if (d->residual)
        UC_REQUIRE(d->res_dtype == UC_F32 || d->res_dtype == lo16, "uc_gemm: bad res_dtype %d for compute dtype %d", d->res_dtype, d->compute_dtype);
    if (d->M == 0) return UC_OK;

    GemmParams p;
    p.A = d->A; p.lda = d->lda; p.W = d->W; p.M = d->M; p.N = d->N; p.K = d->K; p.relu_a = d->relu_a;
    p.cB = d->conv_B; p.cH = d->conv_H; p.cW = d->conv_W; p.cCin = d->conv_Cin; p.cStride = d->conv_stride;
    p.cHo = d->conv_Ho; p.cWo = d->conv_Wo;
    p.bias = d->bias; p.act = d->act; p.residual = d->residual; p.residual2 = d->residual2; p.res_dtype = d->res_dtype; p.ldr = d->ldr;
    p.rope_cols = d->rope_cols; p.rope_pos = d->rope_pos; p.rope_table = (const float2*)d->rope_table;
    p.rope_npos = d->rope_npos; p.vt_col0 = d->vt_col0; p.vt_out = (bf16_t*)d->vt_out; p.vt_ntok = d->vt_ntok;
    p.vt_npad = d->vt_npad; p.C = d->C; p.out_dtype = d->out_dtype; p.ldc = d->ldc; p.preact = nullptr;

    if (d->a_mode == UC_A_CONV3X3) {
        UC_REQUIRE(d->conv_B > 0 && d->conv_H > 0 && d->conv_W > 0 && d->conv_Cin > 0 && d->conv_stride > 0,
                   "uc_gemm: bad conv geometry");
        UC_REQUIRE(d->conv_Ho == (d->conv_H - 1) / d->conv_stride + 1 && d->conv_Wo == (d->conv_W - 1) / d->conv_stride + 1,
                   "uc_gemm: conv output size must be floor((H-1)/s)+1 (3x3, pad 1)");
        UC_REQUIRE(d->M == (int64_t)d->conv_B * d->conv_Ho * d->conv_Wo, "uc_gemm: conv M mismatch");
        UC_REQUIRE(d->K == (int64_t)9 * d->conv_Cin, "uc_gemm: conv K must be 9*Cin");
    } else {
        UC_REQUIRE(d->lda >= d->K, "uc_gemm: lda < K");
    }
    if (!d->tail_out) UC_REQUIRE(d->ldc >= (d->vt_col0 >= 0 ? d->vt_col0 : d->N), "uc_gemm: ldc smaller than the columns written to C");
    if (d->residual) UC_REQUIRE(d->ldr >= d->N, "uc_gemm: ldr < N");

    hipStream_t st = (hipStream_t)stream;
    if (d->compute_dtype == UC_BF16 || f16) {
        if (f16)
            UC_REQUIRE(d->rope_cols <= 0 && d->vt_col0 < 0 && !d->ln_stats && !d->ln_colsum && !d->twin_out && !d->stats_out && d->split_k <= 1 &&
                           !d->preact_out && !d->dact_u,
                       "uc_gemm(f16): the fp16 operand form takes bias / activation / residual(s) / the fused tail only (prediction heads)");
        UC_REQUIRE(d->K % 8 == 0, "uc_gemm(bf16): K must be a multiple of 8 (got %lld)", (long long)d->K);
        UC_REQUIRE(((uintptr_t)d->A % 16 == 0) && ((uintptr_t)d->W % 16 == 0), "uc_gemm(bf16): A/W must be 16-byte aligned");
        if (d->a_mode == UC_A_DENSE) UC_REQUIRE(d->lda % 8 == 0, "uc_gemm(bf16): lda must be a multiple of 8");
        else UC_REQUIRE(d->conv_Cin % 8 == 0, "uc_gemm(bf16): conv Cin must be a multiple of 8");
        if (d->rope_cols > 0) {
            UC_REQUIRE(d->rope_cols % 64 == 0 && d->rope_cols <= d->N, "uc_gemm: rope_cols must be a multiple of the 64-wide head and <= N");
            UC_REQUIRE(d->rope_pos && d->rope_table && d->rope_npos > 0, "uc_gemm: rope needs positions and table");
            UC_REQUIRE(d->rope_base > 0.f && d->rope_f0 != 0.f, "uc_gemm: rope needs the base and F0 the table was built with");
            UC_REQUIRE(d->act == UC_ACT_NONE, "uc_gemm: rope epilogue cannot be combined with an activation");
        }
        if (d->vt_col0 >= 0) {
            UC_REQUIRE(d->vt_col0 % 64 == 0 && (d->N - d->vt_col0) % 64 == 0 && d->vt_col0 < d->N,
                       "uc_gemm: vt_col0 and N must delimit whole 64-wide heads");
            UC_REQUIRE(d->vt_out && d->vt_ntok > 0 && d->M % d->vt_ntok == 0, "uc_gemm: vt epilogue needs vt_out and M %% vt_ntok == 0");
            UC_REQUIRE(d->vt_npad % 64 == 0 && d->vt_npad >= d->vt_ntok, "uc_gemm: vt_npad must be roundup(vt_ntok,64)");
            UC_REQUIRE(d->act == UC_ACT_NONE && !d->residual, "uc_gemm: vt epilogue cannot be combined with act/residual");
            UC_REQUIRE(d->rope_cols <= d->vt_col0, "uc_gemm: rope columns overlap vt columns");
            UC_REQUIRE((uintptr_t)d->vt_out % 8 == 0, "uc_gemm: vt_out must be 8-byte aligned");
        }
        // dense operands with K % 64 == 0 take the direct-to-LDS kernel (gemm_glds.hip)
        // (initial value from UC_GEMM_VARIANT, switchable at run time through uc_tuning_set: tests and micro-benchmarks run every
        // tile variant inside one process)
        const UcKnobs& knobs = uc_knobs();
        const int forced_variant = g_uc_gemm_variant.load(std::memory_order_relaxed);   // -3: automatic, -1: register-staged kernel, 0..3, 6: glds tile variants
        if (d->split_k > 1) {
            UC_REQUIRE(d->a_mode == UC_A_DENSE && d->K % 64 == 0 && !d->relu_a, "uc_gemm: split_k needs a dense operand with K %% 64 == 0");
            UC_REQUIRE(d->out_dtype == UC_F32 && !d->bias && d->act == UC_ACT_NONE && !d->residual && d->rope_cols <= 0 && d->vt_col0 < 0 && !d->preact_out,
                       "uc_gemm: split_k accumulates raw fp32 partial products only (no epilogue options)");
            UC_REQUIRE(d->split_k <= 1024, "uc_gemm: split_k too large");
        }
        if (d->preact_out) UC_REQUIRE(d->a_mode == UC_A_DENSE && d->K % 64 == 0 && !d->relu_a && d->vt_col0 < 0 && d->rope_cols <= 0,
                                      "uc_gemm: preact_out is implemented on the dense direct-to-LDS kernel only");
        if (d->dact_u) {
            UC_REQUIRE(d->dact_act == UC_ACT_GELU_ERF || d->dact_act == UC_ACT_RELU, "uc_gemm: bad dact_act %d", d->dact_act);
            UC_REQUIRE(d->out_dtype == UC_BF16 && d->split_k <= 1 && d->vt_col0 < 0 && d->rope_cols <= 0 && !d->preact_out &&
                           ((d->a_mode == UC_A_DENSE && d->K % 64 == 0 && !d->relu_a) || (d->a_mode == UC_A_CONV3X3 && d->conv_Cin % 32 == 0)),
                       "uc_gemm: dact_u needs the bf16 direct-to-LDS kernels, bf16 output and a plain epilogue");
        }
        // (fp16 operands have no register-staged fallback kernel: K % 32 == 0 — the DPT's 96-channel ConvTranspose GEMM — takes the
        // 32-deep tile of the direct-to-LDS kernel instead)
        const bool dense32 = f16 && d->a_mode == UC_A_DENSE && d->K % 64 != 0 && d->K % 32 == 0 && !d->relu_a;
        const bool glds_dense = d->a_mode == UC_A_DENSE && (d->K % 64 == 0 || dense32) && !d->relu_a;
        if (d->ln_stats || d->ln_colsum) {
            UC_REQUIRE(d->ln_stats && d->ln_colsum, "uc_gemm: the folded LayerNorm needs both ln_stats and ln_colsum");
            UC_REQUIRE(d->ln_nblk == 0 || (d->ln_nblk > 0 && d->K == (int64_t)64 * d->ln_nblk && d->ln_eps > 0.f),
                       "uc_gemm: ln_nblk > 0 (block partials instead of finalized statistics) needs K == 64 * ln_nblk and ln_eps > 0");
            UC_REQUIRE(glds_dense && d->out_dtype == UC_BF16 && !d->residual && !d->preact_out && !d->dact_u && d->split_k <= 1 &&
                           d->N % 64 == 0 && (d->act == UC_ACT_NONE || d->act == UC_ACT_GELU_ERF),
                       "uc_gemm: ln_stats needs the dense direct-to-LDS kernel (K %% 64 == 0), bf16 output, N %% 64 == 0 and a plain epilogue");
            UC_REQUIRE((uintptr_t)d->ln_stats % 8 == 0 && (uintptr_t)d->ln_colsum % 16 == 0 && (uintptr_t)d->C % 16 == 0 && d->ldc % 8 == 0 &&
                           (!d->bias || (uintptr_t)d->bias % 16 == 0), "uc_gemm: ln_stats / ln_colsum / C / bias alignment");
        }
        if (d->twin_out || d->stats_out) {
            const bool f32_stream = d->out_dtype == UC_F32 && (!d->residual || d->res_dtype == UC_F32);
            const bool bf16_stream = d->out_dtype == UC_BF16 && (!d->residual || d->res_dtype == UC_BF16) && !d->residual2 && !d->twin_out;   // C is its own twin
            UC_REQUIRE(glds_dense && (f32_stream || bf16_stream) && d->act == UC_ACT_NONE && !d->preact_out && !d->dact_u && d->split_k <= 1 &&
                           d->vt_col0 < 0 && d->rope_cols <= 0 && d->N % 64 == 0,
                       "uc_gemm: twin_out / stats_out need the dense direct-to-LDS kernel, an fp32 stream (fp32 output + fp32 residual) or a bf16 "
                       "stream (bf16 output + ONE bf16 residual, no twin), N %% 64 == 0");
            UC_REQUIRE((uintptr_t)d->C % 16 == 0 && d->ldc % 8 == 0 && (!d->bias || (uintptr_t)d->bias % 16 == 0) &&
                           (!d->residual || ((uintptr_t)d->residual % 16 == 0 && d->ldr % 4 == 0 && (!d->residual2 || (uintptr_t)d->residual2 % 16 == 0))),
                       "uc_gemm: twin_out / stats_out need 16-byte aligned C / bias / residual");
            if (d->twin_out) UC_REQUIRE((uintptr_t)d->twin_out % 8 == 0 && d->ldt % 4 == 0 && d->ldt >= d->N, "uc_gemm: twin_out alignment / ldt");
            if (d->stats_out) UC_REQUIRE((uintptr_t)d->stats_out % 8 == 0, "uc_gemm: stats_out alignment");
        }
        if (d->tail_out) {
            UC_REQUIRE(d->tail_w && d->N == 128 && d->a_mode == UC_A_CONV3X3,
                       "uc_gemm: the fused tail needs tail_w, a 3x3 convolution and N == 128 (one 128-wide tile holds a pixel's channels)");
            UC_REQUIRE(!d->residual && d->rope_cols <= 0 && d->vt_col0 < 0 && d->split_k <= 1 && !d->preact_out && !d->dact_u && !d->ln_stats &&
                           !d->twin_out && !d->stats_out, "uc_gemm: the fused tail takes bias + activation only");
            UC_REQUIRE((uintptr_t)d->tail_out % 16 == 0 && (uintptr_t)d->tail_w % 4 == 0 && (!d->tail_b || (uintptr_t)d->tail_b % 4 == 0),
                       "uc_gemm: tail_out must be 16-byte aligned");
        }
        // the conv DMA addresses a tile's input window (the images its 256 output rows touch) with 32-bit byte offsets
        const int64_t conv_window_bytes = d->a_mode == UC_A_CONV3X3
            ? (256 / std::max<int64_t>(1, (int64_t)d->conv_Ho * d->conv_Wo) + 2) * (int64_t)d->conv_H * d->conv_W * d->conv_Cin * 2 : 0;
        // Cin % 64 == 0: any tile; Cin % 32 == 0 (the DPT's 96-channel reassemble stage): the 32-deep K-step tile only
        const bool glds_conv = d->a_mode == UC_A_CONV3X3 && d->conv_Cin % 32 == 0 &&
                               (int64_t)d->conv_B * d->conv_H * d->conv_W < (int64_t)1 << 30 && conv_window_bytes < (int64_t)1 << 31 &&
                               (int64_t)256 * d->K * 2 < (int64_t)1 << 31;
        if ((glds_dense || glds_conv) && forced_variant != -1) {
            GldsParams g;
            g.A = (const bf16_t*)d->A; g.lda = d->lda; g.W = (const bf16_t*)d->W; g.M = d->M; g.N = d->N; g.K = d->K;
            g.bias = d->bias; g.act = d->act; g.residual = d->residual; g.residual2 = d->residual2; g.res_dtype = d->res_dtype;
            g.ldr = d->ldr; g.rope_cols = d->rope_cols; g.rope_pos = d->rope_pos; g.rope_table = (const float2*)d->rope_table;
            g.rope_npos = d->rope_npos;
            g.rope_turn0 = d->rope_cols > 0 ? (float)((double)d->rope_f0 / 6.283185307179586476925) : 0.f;
            g.rope_ratio = d->rope_cols > 0 ? (float)pow((double)d->rope_base, -1.0 / 16.0) : 1.f;
            g.rope_l2ratio = d->rope_cols > 0 ? (float)(-log2((double)d->rope_base) / 16.0) : 0.f;
            g.vt_col0 = d->vt_col0; g.vt_out = (bf16_t*)d->vt_out; g.vt_ntok = d->vt_ntok;
            g.vt_npad = d->vt_npad; g.C = d->C; g.out_dtype = d->out_dtype; g.ldc = d->ldc; g.tiles_m = g.tiles_n = 0;
            const bool c_ok = ((uintptr_t)d->C % 16 == 0) && (d->ldc % 8 == 0);
            const bool b_ok = !d->bias || ((uintptr_t)d->bias % 16 == 0);
            const bool r_ok = !d->residual || (((uintptr_t)d->residual % 16 == 0) && (d->ldr % 4 == 0) &&
                                               (!d->residual2 || (uintptr_t)d->residual2 % 16 == 0));
            const bool x_ok = (!d->preact_out || (uintptr_t)d->preact_out % 16 == 0) && (!d->dact_u || (uintptr_t)d->dact_u % 8 == 0);
            g.vec_ok = (c_ok && b_ok && r_ok && x_ok) ? 1 : 0;
            g.preact = d->preact_out; g.split_k = d->split_k > 1 ? d->split_k : 1;
            g.ln_stats = d->ln_nblk > 0 ? nullptr : (const float2*)d->ln_stats; g.ln_colsum = d->ln_colsum;
            g.ln_partial = d->ln_nblk > 0 ? (const float2*)d->ln_stats : nullptr; g.ln_nblk = d->ln_nblk; g.ln_eps = d->ln_eps;
            g.twin = (bf16_t*)d->twin_out; g.ldt = d->ldt; g.stats_out = (float2*)d->stats_out;
            g.tail_w = d->tail_w; g.tail_b = d->tail_b; g.tail_out = d->tail_out;
            if (d->tail_out) g.vec_ok = 0;     // never one of the single-family kernels
            g.dact_u = (const bf16_t*)d->dact_u; g.dact_act = d->dact_act;
            g.group_m = knobs.gemm_group_m;
#ifdef UC_DIAG
            g.dbg = knobs.gemm_dbg;     // (diag build only; the release build has no code behind these bits)
#else
            g.dbg = 0;
#endif
            g.f16 = f16 ? 1 : 0;
            g.sat_flag = f16 ? d->sat_flag : nullptr;
            g.a_mode = d->a_mode; g.relu_a = d->relu_a; g.cH = d->conv_H; g.cW = d->conv_W; g.cCin = d->conv_Cin;
            g.cStride = d->conv_stride; g.cHo = d->conv_Ho; g.cWo = d->conv_Wo;
            if (d->a_mode == UC_A_CONV3X3) {
                g.dWo = uc_make_fastdiv((unsigned)d->conv_Wo); g.dHo = uc_make_fastdiv((unsigned)d->conv_Ho);
                g.dHWo = uc_make_fastdiv((unsigned)d->conv_Ho * (unsigned)d->conv_Wo); g.dCin = uc_make_fastdiv((unsigned)d->conv_Cin);
            }
            int variant = forced_variant;
            if ((d->a_mode == UC_A_CONV3X3 && d->conv_Cin % 64 != 0) || dense32) variant = 3;
            else if (variant < 0) {
                // tile choice: the 256x256 tile (16 waves) has the best steady state (least LDS fill per flop) but needs
                // enough tiles to cover the 256 CUs; smaller problems fall back to 256x128 / 128x128 tiles.
                const int64_t t256 = ceil_div64(d->M, 256) * ceil_div64(d->N, 256);
                const int64_t t256x128 = ceil_div64(d->M, 256) * ceil_div64(d->N, 128);
                const int64_t sk = d->split_k > 1 ? d->split_k : 1;
                variant = t256 * sk >= 192 ? 2 : (t256x128 * sk >= 160 ? 1 : 0);
                if (d->a_mode == UC_A_DENSE) {
                    // Dense launches of a few rounds of tiles (the batch sweep's 2 - 16 pairs): what matters is how many ROUNDS of
                    // workgroups a tile size needs on the 256 CUs, times what a round of that tile costs — measured per round at
                    // K = 768 / 1024 (tools/scratch/bench_midsize_variants.py): 128x128 ~13 / 19 us, 256x128 ~16 / 23 us, 256x256 ~22 / 29 us,
                    // i.e. 1 : 1.25 : 1.7.  The thresholds above missed the quantisation: 144 tiles of 256x256 beat 288 of 256x128 by
                    // 38 % (decoder qkv at 4 pairs), 144 of 256x128 beat 288 of 128x128 by 47 % (at 2 pairs).
                    const int64_t cus = uc_num_cus();
                    const int64_t t128 = ceil_div64(d->M, 128) * ceil_div64(d->N, 128);
                    const double c0 = (double)ceil_div64(t128 * sk, cus), c1 = 1.25 * (double)ceil_div64(t256x128 * sk, cus),
                                 c2 = 1.7 * (double)ceil_div64(t256 * sk, cus);
                    variant = (c2 <= c1 && c2 <= c0) ? 2 : (c1 <= c0 ? 1 : 0);
                }
                // a 256-wide tile whose last column block is at most half full wastes a 128-column slab of MFMA work per row
                // panel (N = 128: half of every tile): take the 256x128 tile there
                const int64_t waste256 = ceil_div64(d->N, 256) * 256 - d->N, waste128 = ceil_div64(d->N, 128) * 128 - d->N;
                if (variant == 2 && waste256 - waste128 >= 128 && t256x128 * sk >= 160) variant = 1;
                // 256x128 tiles with enough workgroups for two per CU: the 32-deep K-step form (72 KiB of LDS, two co-resident
                // 8-wave workgroups) keeps 16 waves on a CU where the 64-deep form (96 KiB) leaves 8
                if (variant == 1 && t256x128 * sk >= 512 && knobs.gemm_coresident) variant = 3;
            }
            if (d->tail_out && (variant == 2 || variant == 6)) variant = 1;    // the tail needs a tile that spans all 128 columns with two wave columns
            // Small-M path: a launch (dense or 3x3 conv) whose 128x128 tiles cover at most half the CUs is a chain of K / 64 dependent steps of
            // ~0.7 us on each of them (neither a smaller tile nor a deeper ring shortens it: measured) — split K in two across twice
            // the workgroups, hand-over inside the kernel (fuse_split2).  UC_GEMM_SMALLM / tuning knob small_m_split = smallest K it is taken for
            // (0: never — the sum over K is then one chain whatever the batch size, and a pair's bits do not depend on its batch).
            g.fuse_split2 = 0; g.fs_ws = nullptr; g.fs_flags = nullptr;
            const int small_m_k = g_uc_small_m_split.load(std::memory_order_relaxed);
            if (forced_variant < 0 && variant == 0 && d->split_k <= 1 && !d->tail_out && small_m_k > 0 &&
                d->K >= small_m_k && d->K % 128 == 0 &&
                2 * ceil_div64(d->M, 128) * ceil_div64(d->N, 128) <= uc_num_cus() &&
                ceil_div64(d->M, 128) * ceil_div64(d->N, 128) <= UC_FUSE_TILES && d->fuse_ws) {
                UC_REQUIRE((uintptr_t)d->fuse_ws % 256 == 0, "uc_gemm: fuse_ws must be 256-byte aligned");
                g.fuse_split2 = 1; g.fs_ws = (float*)d->fuse_ws; g.fs_flags = (unsigned*)((float*)d->fuse_ws + UC_FUSE_WS_FLOATS);
            }
            { const int nt = knobs.gemm_nt;
              const int64_t out_bytes = d->M * d->N * (d->out_dtype == UC_F32 ? 4 : 2);
              g.nt_out = out_bytes > ((int64_t)128 << 20) ? (nt >= 0 ? nt : 7) : 0; }   // bit 0: fp32 residual stream, 1: bf16 outputs, 2: bf16 RoPE (q, k) tiles
            g.stagger = g_uc_gemm_stagger.load(std::memory_order_relaxed);   // -1: the launcher's default policy
            g.trace = nullptr;
#ifdef UC_DIAG
            const int trace_on = knobs.gemm_trace;
            static unsigned long long* trace_buf = nullptr;
            const size_t trace_cap = 1 << 16;
            if (trace_on) {
                if (!trace_buf) (void)hipMalloc((void**)&trace_buf, trace_cap * 6 * sizeof(unsigned long long));
                g.trace = trace_buf;
            }
#endif
            uc_launch_gemm_glds(g, variant, st, forced_variant < 0);
            UC_CHECK_LAUNCH("uc_gemm(glds)");
#ifdef UC_DIAG
            if (trace_on) {   // diagnostics only (diag build): per-CU timeline statistics of this launch to stderr
                (void)hipStreamSynchronize(st);
                const int bm = variant >= 1 ? 256 : 128, bn = (variant == 2 || variant == 6) ? 256 : 128;
                size_t nwg = (size_t)ceil_div64(d->M, bm) * ceil_div64(d->N, bn) * (size_t)g.split_k;
                if (nwg > trace_cap) nwg = trace_cap;
                unsigned long long* h = (unsigned long long*)malloc(nwg * 6 * sizeof(unsigned long long));
                (void)hipMemcpy(h, trace_buf, nwg * 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                double pro = 0, loop = 0, epi = 0; unsigned long long tmin = ~0ull, tmax = 0;
                for (size_t i = 0; i < nwg; ++i) {
                    pro += (double)(h[i*6+1] - h[i*6+0]); loop += (double)(h[i*6+2] - h[i*6+1]); epi += (double)(h[i*6+3] - h[i*6+2]);
                    if (h[i*6+0] < tmin) tmin = h[i*6+0];
                    if (h[i*6+3] > tmax) tmax = h[i*6+3];
                }
                // per-CU gaps: sort WGs by (xcc, hw id cu/se bits) then start time
                struct Rec { unsigned long long key, s, e; };
                Rec* r = (Rec*)malloc(nwg * sizeof(Rec));
                for (size_t i = 0; i < nwg; ++i) {
                    const unsigned hw = (unsigned)h[i*6+4];
                    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    r[i].key = ((h[i*6+5] & 0xf) << 12) | (se << 8) | (sh << 4) | cu; r[i].s = h[i*6+0]; r[i].e = h[i*6+3];
                }
                qsort(r, nwg, sizeof(Rec), [](const void* a, const void* b) -> int {
                    const Rec* x = (const Rec*)a; const Rec* y = (const Rec*)b;
                    if (x->key != y->key) return x->key < y->key ? -1 : 1;
                    return x->s < y->s ? -1 : (x->s > y->s ? 1 : 0); });
                double gap = 0; size_t ngap = 0, ncu = 0; double overlap = 0;
                for (size_t i = 0; i < nwg; ++i) {
                    if (i == 0 || r[i].key != r[i-1].key) { ++ncu; continue; }
                    const double gp = (double)r[i].s - (double)r[i-1].e;
                    if (gp >= 0) { gap += gp; ++ngap; } else overlap += 1;
                }
                fprintf(stderr, "[uc_gemm trace] M=%lld N=%lld K=%lld variant=%d wgs=%zu cus=%zu span=%.1f us | per WG: prologue %.2f us, loop %.2f us, epilogue %.2f us | same-CU gap %.2f us (n=%zu, overlapping pairs %.0f)\n",
                        (long long)d->M, (long long)d->N, (long long)d->K, variant, nwg, ncu, (tmax - tmin) * 0.01,
                        pro / nwg * 0.01, loop / nwg * 0.01, epi / nwg * 0.01, ngap ? gap / ngap * 0.01 : 0.0, ngap, overlap);
                free(h); free(r);
            }
#endif
            return UC_OK;
        }
        UC_REQUIRE(!d->ln_stats && !d->twin_out && !d->stats_out, "uc_gemm: the LayerNorm fusion options need the direct-to-LDS kernel (forced off?)");
        UC_REQUIRE(!d->tail_out, "uc_gemm: the fused tail needs the direct-to-LDS kernel (dense K %% 64 == 0 / conv Cin %% 32 == 0)");
        p.tiles_m = (int)ceil_div64(d->M, BM);
        p.tiles_n = (int)ceil_div64(d->N, BN);
        const unsigned grid = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
        const size_t smem = 4 * TILE_BYTES;
        if (f16 && d->a_mode == UC_A_DENSE)
            hipLaunchKernelGGL((gemm_bf16_kernel<UC_A_DENSE, true>), dim3(grid), dim3(GEMM_THREADS), smem, st, p);
        else if (f16)
            hipLaunchKernelGGL((gemm_bf16_kernel<UC_A_CONV3X3, true>), dim3(grid), dim3(GEMM_THREADS), smem, st, p);
        else if (d->a_mode == UC_A_DENSE)
            hipLaunchKernelGGL((gemm_bf16_kernel<UC_A_DENSE>), dim3(grid), dim3(GEMM_THREADS), smem, st, p);
        else
            hipLaunchKernelGGL((gemm_bf16_kernel<UC_A_CONV3X3>), dim3(grid), dim3(GEMM_THREADS), smem, st, p);
    } else if (d->compute_dtype == UC_F32) {
        UC_REQUIRE(!d->ln_stats && !d->twin_out && !d->stats_out && !d->tail_out, "uc_gemm(f32): the LayerNorm fusion options and the fused tail are bf16-path only");
        if (d->split_k > 1 || d->dact_u) {
            uc_set_error("uc_gemm(f32): split_k / dact_u are only implemented for the bf16 MFMA path");
            return UC_ERR_UNSUPPORTED;
        }
        p.preact = d->preact_out;
        if (d->vt_col0 >= 0) {
            uc_set_error("uc_gemm(f32): vt epilogue is only implemented for the bf16 MFMA path");
            return UC_ERR_UNSUPPORTED;
        }
        if (d->rope_cols > 0) {
            uc_set_error("uc_gemm(f32): fused RoPE epilogue is only implemented for the bf16 MFMA path; call uc_rope2d");
            return UC_ERR_UNSUPPORTED;
        }
        p.tiles_m = (int)ceil_div64(d->M, FBM);
        p.tiles_n = (int)ceil_div64(d->N, FBN);
        const unsigned grid = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
        if (d->a_mode == UC_A_DENSE)
            hipLaunchKernelGGL((gemm_f32_kernel<UC_A_DENSE>), dim3(grid), dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<UC_A_CONV3X3>), dim3(grid), dim3(256), 0, st, p);
    } else {
        uc_set_error("uc_gemm: unsupported compute dtype %d", d->compute_dtype);
        return UC_ERR_BAD_ARG;
    }
    UC_CHECK_LAUNCH("uc_gemm");
    return UC_OK;
}
