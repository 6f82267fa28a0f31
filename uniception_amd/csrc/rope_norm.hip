// RoPE-2D (curope drop-in) and LayerNorm kernels for gfx950.
// Both are HBM-bound streaming kernels: one pass over the data, 16-byte accesses per lane where
// the layout allows, trig hoisted out of the per-head loop (the reference recomputes nothing per
// head either: kernels.cu:47-50 computes cos/sin once per (token, channel) and loops over heads).
#include "common.h"
#include "knobs.h"
#include <stdlib.h>
#include <type_traits>

// ---------------------------------------------------------------------------------------
// RoPE-2D in place.  Reference arithmetic (curope/kernels.cu:36-81, curope.cpp:21-46):
//   Q = D/4; channels [0,Q)=u_y [Q,2Q)=v_y [2Q,3Q)=u_x [3Q,4Q)=v_x
//   ang = pos[axis] * (fwd / powf(base, i/Q));  u' = u cos - v sin;  v' = v cos + u sin
// Work split: a workgroup owns TOK consecutive tokens. Phase 1 fills an LDS table of
// (cos,sin) for (token, axis, i); phase 2 streams all heads of those tokens through it.
// ---------------------------------------------------------------------------------------
template <typename Tag, int VEC>
__global__ __launch_bounds__(256) void rope2d_kernel(typename Tag::storage* __restrict__ tok,
                                                     const int64_t* __restrict__ pos, int64_t BN,
                                                     int N, int H, int Q, int64_t sb, int64_t sn,
                                                     int64_t sh, float base, float fwd, int TOK) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* cs = reinterpret_cast<float2*>(smem_raw);  // [TOK][2][Q]
    const int64_t tok0 = (int64_t)blockIdx.x * TOK;
    const int ntok = (int)min((int64_t)TOK, BN - tok0);
    const int tab = ntok * 2 * Q;
    for (int t = threadIdx.x; t < tab; t += blockDim.x) {
        const int i = t % Q;
        const int axis = (t / Q) & 1;
        const int lt = t / (2 * Q);
        const float p = (float)pos[(tok0 + lt) * 2 + axis];
        const float inv_freq = fwd / powf(base, (float)i / (float)Q);
        const float ang = p * inv_freq;
        cs[t] = make_float2(cosf(ang), sinf(ang));
    }
    __syncthreads();
    const int qv = Q / VEC;                // vector groups per quarter
    const int per_tok = H * 2 * qv;        // work items per token
    const int items = ntok * per_tok;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int g = it % qv;
        const int axis = (it / qv) & 1;
        const int h = (it / (2 * qv)) % H;
        const int lt = it / per_tok;
        const int64_t t = tok0 + lt;
        const int64_t b = t / N, n = t % N;
        typename Tag::storage* pu = tok + b * sb + n * sn + (int64_t)h * sh + axis * 2 * Q + g * VEC;
        typename Tag::storage* pv = pu + Q;
        const float2* c = cs + (lt * 2 + axis) * Q + g * VEC;
        typedef typename Tag::storage vec_t __attribute__((ext_vector_type(VEC)));
        union { vec_t v; typename Tag::storage s[VEC]; } ru, rv;
        ru.v = *reinterpret_cast<const vec_t*>(pu);   // one 8/16-byte access per lane when VEC==4
        rv.v = *reinterpret_cast<const vec_t*>(pv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float u = Tag::load(&ru.s[e]), v = Tag::load(&rv.s[e]);
            const float2 k = c[e];
            Tag::store(&ru.s[e], u * k.x - v * k.y);
            Tag::store(&rv.s[e], v * k.x + u * k.y);
        }
        *reinterpret_cast<vec_t*>(pu) = ru.v;
        *reinterpret_cast<vec_t*>(pv) = rv.v;
    }
}

template <typename Tag>
static int launch_rope(void* tokens, const int64_t* pos, int B, int N, int H, int D, int64_t sb,
                       int64_t sn, int64_t sh, float base, float fwd, hipStream_t st) {
    const int Q = D / 4;
    const int64_t BN = (int64_t)B * N;
    const int TOK = 8;
    const size_t smem = (size_t)TOK * 2 * Q * sizeof(float2);
    const unsigned grid = (unsigned)ceil_div64(BN, TOK);
    typedef typename Tag::storage S;
    // vector width 4 needs Q%4==0 and every address 4-element aligned
    const bool vec4 = (Q % 4 == 0) && (sb % 4 == 0) && (sn % 4 == 0) && (sh % 4 == 0) &&
                      (((uintptr_t)tokens) % (4 * sizeof(S)) == 0);
    if (vec4)
        hipLaunchKernelGGL((rope2d_kernel<Tag, 4>), dim3(grid), dim3(256), smem, st, (S*)tokens, pos,
                           BN, N, H, Q, sb, sn, sh, base, fwd, TOK);
    else
        hipLaunchKernelGGL((rope2d_kernel<Tag, 1>), dim3(grid), dim3(256), smem, st, (S*)tokens, pos,
                           BN, N, H, Q, sb, sn, sh, base, fwd, TOK);
    return 0;
}

extern "C" int uc_rope2d(void* tokens, const int64_t* positions, int B, int N, int H, int D,
                         int64_t sb, int64_t sn, int64_t sh, float base, float fwd, int dtype,
                         uc_stream_t stream) {
    UC_REQUIRE(tokens && positions, "uc_rope2d: null pointer");
    UC_REQUIRE(B >= 0 && N >= 0 && H >= 0 && D > 0, "uc_rope2d: negative dimension");
    UC_REQUIRE(D % 4 == 0, "uc_rope2d: token dim must be multiple of 4 (got %d)", D);
    UC_REQUIRE(D <= 1024, "uc_rope2d: head dim %d too large", D);
    if ((int64_t)B * N * H == 0) return UC_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case UC_F32: launch_rope<F32Tag>(tokens, positions, B, N, H, D, sb, sn, sh, base, fwd, st); break;
        case UC_BF16: launch_rope<BF16Tag>(tokens, positions, B, N, H, D, sb, sn, sh, base, fwd, st); break;
        case UC_F16: launch_rope<F16Tag>(tokens, positions, B, N, H, D, sb, sn, sh, base, fwd, st); break;
        default: uc_set_error("uc_rope2d: unsupported dtype %d", dtype); return UC_ERR_BAD_ARG;
    }
    UC_CHECK_LAUNCH("uc_rope2d");
    return UC_OK;
}

__global__ void rope_table_kernel(float2* __restrict__ table, int npos, int Q, float base, float F0) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npos * Q) return;
    const int i = t % Q, p = t / Q;
    const float inv_freq = F0 / powf(base, (float)i / (float)Q);
    const float ang = (float)p * inv_freq;
    table[t] = make_float2(cosf(ang), sinf(ang));
}

extern "C" int uc_rope_table(float* table, int npos, int Q, float base, float F0, uc_stream_t stream) {
    UC_REQUIRE(table && npos > 0 && Q > 0, "uc_rope_table: bad argument");
    const int n = npos * Q;
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (float2*)table, npos, Q, base, F0);
    UC_CHECK_LAUNCH("uc_rope_table");
    return UC_OK;
}

// ---------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, row cached in registers (C <= 64*4*LN_MAXV), two-pass
// mean/variance in fp32, 16-byte loads and stores.  4 rows per 256-thread workgroup.
// ---------------------------------------------------------------------------------------
#define LN_MAXV 8

template <typename TI>
__device__ __forceinline__ float4_t ln_load4(const typename TI::storage* p);
template <>
__device__ __forceinline__ float4_t ln_load4<F32Tag>(const float* p) {
    return *reinterpret_cast<const float4_t*>(p);
}
template <>
__device__ __forceinline__ float4_t ln_load4<BF16Tag>(const bf16_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    float4_t v;
    v.x = __uint_as_float(r.x << 16);
    v.y = __uint_as_float(r.x & 0xffff0000u);
    v.z = __uint_as_float(r.y << 16);
    v.w = __uint_as_float(r.y & 0xffff0000u);
    return v;
}
template <typename TO>
__device__ __forceinline__ void ln_store4(typename TO::storage* p, float4_t v);
template <>
__device__ __forceinline__ void ln_store4<F32Tag>(float* p, float4_t v) {
    *reinterpret_cast<float4_t*>(p) = v;
}
template <>
__device__ __forceinline__ void ln_store4<BF16Tag>(bf16_t* p, float4_t v) {
    uint2 r;
    r.x = pack_bf16x2(v.x, v.y);
    r.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
}

// Exact-width variant: C == NV*256, no per-chunk predicates, so all NV 16-byte loads of a row are issued back to back
// (the predicated generic kernel below serialises them behind exec-mask branches: 2.8 TB/s vs 7+ TB/s cache-hot).
template <typename TI, typename TO, int NV, bool NT = false>
__global__ __launch_bounds__(256) void layernorm_exact_kernel(const typename TI::storage* __restrict__ x,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              typename TO::storage* __restrict__ y, bf16_t* __restrict__ twin,
                                                              int64_t rows, float eps) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const typename TI::storage* xr = x + row * C;
    float4_t v[NV], g[NV], bb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        // NT: an fp32 residual stream larger than half the Infinity Cache is read once here and once by the next residual
        // epilogue, two GEMMs later — streaming it leaves the caches to the GEMM operands
        if constexpr (NT && std::is_same<TI, F32Tag>::value) v[i] = __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(xr + (i * 64 + lane) * 4));
        else v[i] = ln_load4<TI>(xr + (i * 64 + lane) * 4);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const float4_t*>(gamma + (i * 64 + lane) * 4);
        bb[i] = *reinterpret_cast<const float4_t*>(beta + (i * 64 + lane) * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, d = v[i].z - mean, e = v[i].w - mean;
        q += (a * a + b * b) + (d * d + e * e);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / (float)C) + eps);
    typename TO::storage* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float4_t o;
        o.x = (v[i].x - mean) * rstd * g[i].x + bb[i].x;
        o.y = (v[i].y - mean) * rstd * g[i].y + bb[i].y;
        o.z = (v[i].z - mean) * rstd * g[i].z + bb[i].z;
        o.w = (v[i].w - mean) * rstd * g[i].w + bb[i].w;
        ln_store4<TO>(yr + (i * 64 + lane) * 4, o);
        if (twin) ln_store4<BF16Tag>(twin + row * C + (i * 64 + lane) * 4, o);   // bf16 copy for the consumers that take bf16 operands
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const typename TI::storage* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            typename TO::storage* __restrict__ y, bf16_t* __restrict__ twin,
                                                            int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const typename TI::storage* xr = x + row * C;
    float4_t v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            v[i] = ln_load4<TI>(xr + c);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            const float a = v[i].x - mean, b = v[i].y - mean, d = v[i].z - mean, e = v[i].w - mean;
            q += (a * a + b * b) + (d * d + e * e);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    typename TO::storage* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            const float4_t g = *reinterpret_cast<const float4_t*>(gamma + c);
            const float4_t bb = *reinterpret_cast<const float4_t*>(beta + c);
            float4_t o;
            o.x = (v[i].x - mean) * rstd * g.x + bb.x;
            o.y = (v[i].y - mean) * rstd * g.y + bb.y;
            o.z = (v[i].z - mean) * rstd * g.z + bb.z;
            o.w = (v[i].w - mean) * rstd * g.w + bb.w;
            ln_store4<TO>(yr + c, o);
            if (twin) ln_store4<BF16Tag>(twin + row * C + c, o);
        }
    }
}

// generic fallback: any C, scalar accesses, three passes over the (cache-resident) row
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_scalar_kernel(const typename TI::storage* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               typename TO::storage* __restrict__ y, bf16_t* __restrict__ twin,
                                                               int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const typename TI::storage* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += TI::load(xr + c);
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = TI::load(xr + c) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    typename TO::storage* yr = y + row * C;
    for (int c = lane; c < C; c += 64) {
        const float o = (TI::load(xr + c) - mean) * rstd * gamma[c] + beta[c];
        TO::store(yr + c, o);
        if (twin) BF16Tag::store(twin + row * C + c, o);
    }
}

template <typename TI, typename TO>
static void launch_ln(const void* x, const float* g, const float* b, void* y, bf16_t* twin, int64_t rows, int C,
                      float eps, hipStream_t st) {
    typedef typename TI::storage SI;
    typedef typename TO::storage SO;
    const unsigned grid = (unsigned)ceil_div64(rows, 4);
    const bool vec = (C % 4 == 0) && (C <= 64 * 4 * LN_MAXV) && ((uintptr_t)x % 16 == 0) &&
                     ((uintptr_t)y % 16 == 0) && ((uintptr_t)twin % 8 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)b % 16 == 0) &&
                     (((int64_t)C * sizeof(SI)) % (4 * sizeof(SI)) == 0);
    const bool exact = vec && (C % 256 == 0);
    const int nt_env = uc_knobs().ln_nt;
    const bool nt = nt_env >= 0 ? nt_env != 0 : (rows * (int64_t)C * (int64_t)sizeof(SI) > ((int64_t)128 << 20));
#define UC_LN_EXACT(NV_)                                                                                                          \
    do {                                                                                                                          \
        if (nt) hipLaunchKernelGGL((layernorm_exact_kernel<TI, TO, NV_, true>), dim3(grid), dim3(256), 0, st, (const SI*)x, g, b,  \
                                   (SO*)y, twin, rows, eps);                                                                      \
        else hipLaunchKernelGGL((layernorm_exact_kernel<TI, TO, NV_, false>), dim3(grid), dim3(256), 0, st, (const SI*)x, g, b,    \
                                (SO*)y, twin, rows, eps);                                                                         \
    } while (0)
    if (exact && C == 256) UC_LN_EXACT(1);
    else if (exact && C == 512) UC_LN_EXACT(2);
    else if (exact && C == 768) UC_LN_EXACT(3);
    else if (exact && C == 1024) UC_LN_EXACT(4);
    else if (exact && C == 1536) UC_LN_EXACT(6);
    else if (exact && C == 2048) UC_LN_EXACT(8);
#undef UC_LN_EXACT
    else if (vec)
        hipLaunchKernelGGL((layernorm_vec_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const SI*)x, g, b,
                           (SO*)y, twin, rows, C, eps);
    else
        hipLaunchKernelGGL((layernorm_scalar_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const SI*)x, g,
                           b, (SO*)y, twin, rows, C, eps);
}

extern "C" int uc_layernorm_twin(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                 void* y_twin_bf16, int64_t rows, int C, float eps, uc_stream_t stream) {
    UC_REQUIRE(x && gamma && beta && y, "uc_layernorm: null pointer");
    UC_REQUIRE(rows >= 0 && C > 0, "uc_layernorm: bad shape rows=%lld C=%d", (long long)rows, C);
    if (rows == 0) return UC_OK;
    hipStream_t st = (hipStream_t)stream;
    bf16_t* tw = (bf16_t*)y_twin_bf16;
    if (x_dtype == UC_F32 && y_dtype == UC_F32) launch_ln<F32Tag, F32Tag>(x, gamma, beta, y, tw, rows, C, eps, st);
    else if (x_dtype == UC_F32 && y_dtype == UC_BF16) launch_ln<F32Tag, BF16Tag>(x, gamma, beta, y, tw, rows, C, eps, st);
    else if (x_dtype == UC_BF16 && y_dtype == UC_BF16) launch_ln<BF16Tag, BF16Tag>(x, gamma, beta, y, tw, rows, C, eps, st);
    else if (x_dtype == UC_BF16 && y_dtype == UC_F32) launch_ln<BF16Tag, F32Tag>(x, gamma, beta, y, tw, rows, C, eps, st);
    else {
        uc_set_error("uc_layernorm: unsupported dtypes %d -> %d", x_dtype, y_dtype);
        return UC_ERR_BAD_ARG;
    }
    UC_CHECK_LAUNCH("uc_layernorm");
    return UC_OK;
}

extern "C" int uc_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                            int y_dtype, int64_t rows, int C, float eps, uc_stream_t stream) {
    return uc_layernorm_twin(x, x_dtype, gamma, beta, y, y_dtype, nullptr, rows, C, eps, stream);
}

// ---------------------------------------------------------------------------------------
// Row statistics of the folded LayerNorm: the producer GEMM's epilogue leaves, per row and 64-column block, the block sum
// and the squared deviations from the block mean; the blocks are merged pairwise-exactly (Chan et al.): with block means
// mu_b and the row mean mu,  M2 = sum_b [ M2_b + 64 (mu_b - mu)^2 ].  One thread per row (nblk <= 64 blocks of 8 bytes).
// ---------------------------------------------------------------------------------------
__global__ void ln_stats_finalize_kernel(const float2* __restrict__ partial, int64_t rows, int nblk, float eps, float2* __restrict__ out) {
    // one thread per row; the merge itself is uc_ln_merge_row (common.h),
    // shared with the consumer GEMM's epilogue
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) out[r] = uc_ln_merge_row<true>(partial + r, rows, nblk, eps);     // block-major partials [nblk][rows]: coalesced across the rows of a wave
}

extern "C" int uc_ln_stats_finalize(const float* partial, int64_t rows, int nblk, float eps, float* out, uc_stream_t stream) {
    UC_REQUIRE(partial && out && rows >= 0 && nblk > 0, "uc_ln_stats_finalize: bad argument");
    UC_REQUIRE((uintptr_t)partial % 8 == 0 && (uintptr_t)out % 8 == 0, "uc_ln_stats_finalize: 8-byte alignment");
    if (rows == 0) return UC_OK;
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)partial, rows, nblk, eps, (float2*)out);
    UC_CHECK_LAUNCH("uc_ln_stats_finalize");
    return UC_OK;
}
