// Internal helpers shared by the gfx950 kernels of libuc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/uc_hip.h"

// ---------------------------------------------------------------------------------------
// error plumbing (thread-local text, negative status codes)
// ---------------------------------------------------------------------------------------
void uc_set_error(const char* fmt, ...);

#define UC_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            uc_set_error(__VA_ARGS__);        \
            return UC_ERR_BAD_ARG;            \
        }                                     \
    } while (0)

#define UC_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            uc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return UC_ERR_LAUNCH;                                                 \
        }                                                                         \
    } while (0)

// ---------------------------------------------------------------------------------------
// bf16 / f16 as raw 16-bit storage.  bf16 -> f32 is a shift; f32 -> bf16 rounds to nearest even
// (same as torch's c10::BFloat16).
// ---------------------------------------------------------------------------------------
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// f32 -> bf16 through the gfx950 hardware conversion (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved);
// a hand-rolled integer rounding costs ~6 VALU + a NaN branch per element and was the top VALU item of the
// attention softmax.
typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
typedef float float2v_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const float2v_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v_t));
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float f16_to_f32(unsigned short v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
// fp16 stores SATURATE at +-65504 (round 4): a plain cast turns anything beyond into inf, and the next layer turns inf into NaN.  The
// TF32-class heads keep TF32's mantissa in fp16 operands but not its exponent; saturation keeps an out-of-range map finite, and the
// producers that can overflow (GEMM / conv epilogues, conversions into fp16) raise a caller-provided flag when they hit the limit
// (uc_gemm_desc.sat_flag, uc_convert's sat_flag), so the host can fall back to a wider head format.
// NaN is NOT a range problem and must not be hidden: v_med3_f32 returns a finite bound for a NaN input, so the NaN is passed through
// (the stored fp16 is NaN), and the detectors use the NaN-PROPAGATING maximum (uc_amax: v_maximum3_f32, IEEE 754-2019 maximum) — a
// NaN anywhere makes `!(amax <= UC_F16_MAX)` true, the flag trips and the host leaves the fp16 policy like for any overflow.
#define UC_F16_MAX 65504.0f
__device__ __forceinline__ float uc_sat_f16(float f) { return f != f ? f : __builtin_amdgcn_fmed3f(f, -UC_F16_MAX, UC_F16_MAX); }
__device__ __forceinline__ float uc_amax(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ unsigned short f32_to_f16(float f) {
    _Float16 h = (_Float16)uc_sat_f16(f);
    unsigned short v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}

// element type tags so kernels can be written once
struct F32Tag {
    typedef float storage;
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
struct BF16Tag {
    typedef bf16_t storage;
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};
struct F16Tag {
    typedef unsigned short storage;
    static __device__ __forceinline__ float load(const unsigned short* p) { return f16_to_f32(*p); }
    static __device__ __forceinline__ void store(unsigned short* p, float v) { *p = f32_to_f16(v); }
};

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- exact division of n < 2^31 by a run-time constant d >= 1 without the ~28-instruction software divide -------------
// magic m = ceil(2^(31 + ceil(log2 d)) / d) fits 32 bits and floor(n / d) = (n * m) >> (31 + ceil(log2 d)) for every
// n < 2^31 (m*d - 2^p < d <= 2^(p-31)).  d == 1 is encoded as m == 0.  Host side fills the pair once per launch.
struct uc_fastdiv { unsigned m, s; };
static inline uc_fastdiv uc_make_fastdiv(unsigned d) {
    uc_fastdiv f; f.m = 0; f.s = 0;
    if (d <= 1) return f;
    int l = 0;
    while ((1u << l) < d) ++l;
    const int p = 31 + l;
    f.m = (unsigned)((((unsigned __int128)1 << p) + d - 1) / d);
    f.s = (unsigned)(p - 32);
    return f;
}
__device__ __forceinline__ unsigned uc_div(unsigned n, uc_fastdiv f) { return f.m ? (__umulhi(n, f.m) >> f.s) : n; }


// ---------------------------------------------------------------------------------------
// Row statistics of the folded LayerNorm from the producer GEMM's per-block partials: p[b] = (sum, squared deviations from the
// block mean) of 64-column block b of one row; Chan et al.: with block means mu_b and the row mean mu, M2 = sum_b [M2_b + 64
// (mu_b - mu)^2].  Returns (mean, 1 / sqrt(var_biased + eps)).  ONE definition for the stand-alone kernel
// (uc_ln_stats_finalize, large batches) and the consumer GEMM's epilogue (small batches: no launch): the 16 slot sums are added up
// in a fixed binary tree and nothing is contracted into FMAs, so both give the same bits — a pair's result does not depend on
// which of the two its batch size selects.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float uc_ln_tree16(const float (&v)[16]) {
#pragma clang fp contract(off)
    float a[8], b[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = v[q] + v[q + 8];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = a[q] + a[q + 4];
    return (b[0] + b[2]) + (b[1] + b[3]);
}
// p[b * stride] = block b of the row: stride 1 for a row-major [rows][nblk] array, `rows` for the block-major [nblk][rows] array the
// producer GEMMs write since ABI 11 (one coalesced statistics store per 64 rows instead of a scattered 8-byte store per row)
// BATCH: callers with 32 registers to spare at the call (the eight-wave GEMM's epilogue: 256 per lane) take the all-loads-first form.
template <bool BATCH = false>
__device__ __forceinline__ float2 uc_ln_merge_row(const float2* __restrict__ p, int64_t stride, int nblk, float eps) {
#pragma clang fp contract(off)
    float s[16];
    if (BATCH && nblk <= 16) {
        // rows of at most 1024 channels (every transformer of this path): the row's pairs are read ONCE, all loads in flight
        // together (the loops below are 2 x nblk dependent round trips when nblk is not a compile-time constant); same sums in
        // the same order, so the same bits
        float2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p[min(q, nblk - 1) * stride];          // (clamped, not predicated: no branch between the loads)
#pragma unroll
        for (int q = 0; q < 16; ++q) s[q] = q < nblk ? v[q].x : 0.f;
        const float cnt = 64.f * (float)nblk;
        const float mu = uc_ln_tree16(s) / cnt;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float d = v[q].x * (1.f / 64.f) - mu;
            s[q] = q < nblk ? v[q].y + (64.f * d) * d : 0.f;
        }
        return make_float2(mu, 1.0f / sqrtf(uc_ln_tree16(s) / cnt + eps));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        s[q] = 0.f;
        for (int b = q; b < nblk; b += 16) s[q] += p[b * stride].x;
    }
    const float cnt = 64.f * (float)nblk;
    const float mu = uc_ln_tree16(s) / cnt;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        s[q] = 0.f;
        for (int b = q; b < nblk; b += 16) {
            const float2 pb = p[b * stride];
            const float d = pb.x * (1.f / 64.f) - mu;
            s[q] += pb.y + (64.f * d) * d;
        }
    }
    return make_float2(mu, 1.0f / sqrtf(uc_ln_tree16(s) / cnt + eps));
}

// ---- attention dropout (attn_drop > 0 in training; reference: utils/transformer_blocks.py:198, 245, 251, 374, 380 — nn.Dropout on the
// softmax probabilities, or dropout_p of the fused SDPA) ----
// The keep decision of element (batch b, head h, query q, key k) is a COUNTER-BASED function of (seed, b * H + h, q, k): no mask is
// stored — the forward kernel, both backward kernels (which rebuild P) and uc_attention_drop_mask evaluate the same function, whatever
// their tile shapes (a lane of the forward / dQ kernels holds four consecutive KEYS of a query, a lane of the dK / dV kernel four
// consecutive QUERIES of a key: a generator that yields groups of outputs would fit one of them).  Two murmur3-style rounds over the
// two 32-bit words (q + key word 1, k + key word 2), then the murmur3 finaliser; the per-(seed, b, h) key words are mixed on the
// device from the 64-bit seed.  keep <=> hash >= thr, thr = round(p * 2^32): P(keep) = 1 - p to 2^-32.
struct UcDropout {
    unsigned thr;          // 0: no dropout
    unsigned seed_lo, seed_hi;
    float keep_scale;      // 1 / (1 - p)
};
__host__ __device__ __forceinline__ unsigned uc_fmix32(unsigned y) {
    y ^= y >> 16; y *= 0x85EBCA6Bu; y ^= y >> 13; y *= 0xC2B2AE35u; y ^= y >> 16;
    return y;
}
// the two key words of one (batch, head)
__host__ __device__ __forceinline__ void uc_drop_keys(const UcDropout& d, unsigned bh, unsigned& k1, unsigned& k2) {
    k1 = uc_fmix32(d.seed_lo ^ (bh * 0x9E3779B1u) ^ 0x3C6EF372u);
    k2 = uc_fmix32(d.seed_hi + bh * 0x7F4A7C15u + 0xDAA66D2Bu);
}
__host__ __device__ __forceinline__ unsigned uc_drop_hash(unsigned k1, unsigned k2, unsigned q, unsigned k) {
    unsigned x = (q + k1) * 0xCC9E2D51u;
    x = (x << 15) | (x >> 17);
    x *= 0x1B873593u;
    x ^= (k + k2) * 0x85EBCA6Bu;
    x = (x << 13) | (x >> 19);
    x = x * 5u + 0xE6546B64u;
    return uc_fmix32(x);
}
static inline UcDropout uc_make_dropout(float p, unsigned long long seed) {
    UcDropout d;
    double t = (double)p * 4294967296.0;
    d.thr = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)(t + 0.5));
    d.seed_lo = (unsigned)(seed & 0xFFFFFFFFull);
    d.seed_hi = (unsigned)(seed >> 32);
    d.keep_scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    return d;
}

// compute units of the current device (persistent kernels launch one workgroup per CU)
static inline int uc_num_cus() {
    static int n = [] {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }();
    return n;
}
