// HBM-bound data-movement kernels of the DUSt3R path for gfx950: patch gather, layout/dtype
// conversion at the public BCHW boundary, bilinear resize (align_corners=True), the pixel scatters
// behind ConvTranspose2d(k=s) / pixel_shuffle, the pointmap+confidence adaptor and the 1x1 head conv.
// All are single-pass streaming kernels; channel-last (NHWC) tensors are moved 8 elements per lane
// (16 B bf16 / 2x16 B fp32).
#include "common.h"
#include <type_traits>
#include "knobs.h"

// ---- generic 8-element vector load/store with fp32 math in between --------------------------
struct V8 { float v[8]; };

template <typename Tag>
__device__ __forceinline__ V8 load8(const typename Tag::storage* p);
template <>
__device__ __forceinline__ V8 load8<F32Tag>(const float* p) {
    V8 r;
    const float4_t a = *reinterpret_cast<const float4_t*>(p);
    const float4_t b = *reinterpret_cast<const float4_t*>(p + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <>
__device__ __forceinline__ V8 load8<BF16Tag>(const bf16_t* p) {
    V8 r;
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
typedef _Float16 uc_half2_t __attribute__((ext_vector_type(2)));
typedef float uc_float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {       // round-to-nearest-even, like torch's Half
    const uc_float2_t v = {uc_sat_f16(lo), uc_sat_f16(hi)};          // (saturating: see common.h)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, uc_half2_t));
}
__device__ __forceinline__ void unpack_f16x2(unsigned u, float& lo, float& hi) {
    const uc_float2_t v = __builtin_convertvector(__builtin_bit_cast(uc_half2_t, u), uc_float2_t);
    lo = v.x; hi = v.y;
}
template <>
__device__ __forceinline__ V8 load8<F16Tag>(const unsigned short* p) {
    V8 r;
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    unpack_f16x2(u.x, r.v[0], r.v[1]); unpack_f16x2(u.y, r.v[2], r.v[3]);
    unpack_f16x2(u.z, r.v[4], r.v[5]); unpack_f16x2(u.w, r.v[6], r.v[7]);
    return r;
}
template <typename Tag>
__device__ __forceinline__ void store8(typename Tag::storage* p, const V8& r);
template <>
__device__ __forceinline__ void store8<F16Tag>(unsigned short* p, const V8& r) {
    uint4 u;
    u.x = pack_f16x2(r.v[0], r.v[1]); u.y = pack_f16x2(r.v[2], r.v[3]);
    u.z = pack_f16x2(r.v[4], r.v[5]); u.w = pack_f16x2(r.v[6], r.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
template <>
__device__ __forceinline__ void store8<F32Tag>(float* p, const V8& r) {
    *reinterpret_cast<float4_t*>(p) = (float4_t){r.v[0], r.v[1], r.v[2], r.v[3]};
    *reinterpret_cast<float4_t*>(p + 4) = (float4_t){r.v[4], r.v[5], r.v[6], r.v[7]};
}
template <>
__device__ __forceinline__ void store8<BF16Tag>(bf16_t* p, const V8& r) {
    uint4 u;
    u.x = pack_bf16x2(r.v[0], r.v[1]); u.y = pack_bf16x2(r.v[2], r.v[3]);
    u.z = pack_bf16x2(r.v[4], r.v[5]); u.w = pack_bf16x2(r.v[6], r.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}

#define EW_GRID(n_items) ((unsigned)min((int64_t)65536 * 4, ceil_div64((n_items), 256)))

// =======================================================================================
// patch gather: img fp32 NCHW -> cols [B*h*w, Cin*P*P], columns (c,u,v).  One work item = 4 pixels of a patch row.
// =======================================================================================
template <typename TO>
__global__ void patch_gather_kernel(const float* __restrict__ img, typename TO::storage* __restrict__ cols, int B,
                                    int Cin, int H, int W, int P, int64_t items) {
    const int h = H / P, w = W / P;
    const int P4 = P / 4;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        // item index order == output order: (token, c, u, v4)
        int64_t r = it;
        const int v4 = (int)(r % P4); r /= P4;
        const int u = (int)(r % P); r /= P;
        const int c = (int)(r % Cin); r /= Cin;
        const int j = (int)(r % w); r /= w;
        const int i = (int)(r % h);
        const int b = (int)(r / h);
        const float4_t px = *reinterpret_cast<const float4_t*>(
            img + (((int64_t)b * Cin + c) * H + (int64_t)i * P + u) * W + (int64_t)j * P + v4 * 4);
        typename TO::storage* o = cols + it * 4;
        TO::store(o + 0, px.x); TO::store(o + 1, px.y); TO::store(o + 2, px.z); TO::store(o + 3, px.w);
    }
}

// any patch size (e.g. 14): one element per work item, output-ordered
template <typename TO>
__global__ void patch_gather_scalar_kernel(const float* __restrict__ img, typename TO::storage* __restrict__ cols, int B,
                                           int Cin, int H, int W, int P, int64_t n) {
    const int h = H / P, w = W / P;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n; it += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = it;
        const int v = (int)(r % P); r /= P;
        const int u = (int)(r % P); r /= P;
        const int c = (int)(r % Cin); r /= Cin;
        const int j = (int)(r % w); r /= w;
        const int i = (int)(r % h);
        const int b = (int)(r / h);
        TO::store(cols + it, img[(((int64_t)b * Cin + c) * H + (int64_t)i * P + u) * W + (int64_t)j * P + v]);
    }
}

extern "C" int uc_patch_gather(const float* img, void* cols, int out_dtype, int B, int Cin, int H, int W, int P,
                               uc_stream_t stream) {
    UC_REQUIRE(img && cols, "uc_patch_gather: null pointer");
    UC_REQUIRE(B > 0 && Cin > 0 && P > 0 && H % P == 0 && W % P == 0, "uc_patch_gather: H,W must be multiples of the patch size");
    UC_REQUIRE(out_dtype == UC_F32 || out_dtype == UC_BF16, "uc_patch_gather: bad out_dtype %d", out_dtype);
    hipStream_t st = (hipStream_t)stream;
    const bool quad = (P % 4 == 0) && (W % 4 == 0) && ((uintptr_t)img % 16 == 0);   // 16-byte pixel quads
    if (!quad) {
        const int64_t n = (int64_t)B * Cin * H * W;
        if (out_dtype == UC_F32)
            hipLaunchKernelGGL((patch_gather_scalar_kernel<F32Tag>), dim3(EW_GRID(n)), dim3(256), 0, st, img, (float*)cols, B, Cin, H, W, P, n);
        else
            hipLaunchKernelGGL((patch_gather_scalar_kernel<BF16Tag>), dim3(EW_GRID(n)), dim3(256), 0, st, img, (bf16_t*)cols, B, Cin, H, W, P, n);
        UC_CHECK_LAUNCH("uc_patch_gather");
        return UC_OK;
    }
    const int64_t items = (int64_t)B * Cin * H * W / 4;
    if (out_dtype == UC_F32)
        hipLaunchKernelGGL((patch_gather_kernel<F32Tag>), dim3(EW_GRID(items)), dim3(256), 0, st, img, (float*)cols, B, Cin, H, W, P, items);
    else
        hipLaunchKernelGGL((patch_gather_kernel<BF16Tag>), dim3(EW_GRID(items)), dim3(256), 0, st, img, (bf16_t*)cols, B, Cin, H, W, P, items);
    UC_CHECK_LAUNCH("uc_patch_gather");
    return UC_OK;
}

// =======================================================================================
// NCHW <-> NHWC with dtype conversion: LDS-tiled transpose of [C, HW] <-> [HW, C] per batch image.
// =======================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose2d_kernel(const typename TI::storage* __restrict__ src,
                                                          typename TO::storage* __restrict__ dst, int R, int S) {
    // per batch (blockIdx.z): src [R][S] -> dst [S][R]; 32x32 tiles, 256 threads (32 x 8)
    __shared__ float tile[32][33];
    const int64_t boff = (int64_t)blockIdx.z * R * S;
    const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, s = s0 + tx;
        if (r < R && s < S) tile[ty + 8 * k][tx] = TI::load(src + boff + (int64_t)r * S + s);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = s0 + ty + 8 * k, r = r0 + tx;
        if (r < R && s < S) TO::store(dst + boff + (int64_t)s * R + r, tile[tx][ty + 8 * k]);
    }
}

template <typename TI, typename TO>
static void launch_transpose(const void* src, void* dst, int Bn, int R, int S, hipStream_t st) {
    hipLaunchKernelGGL((transpose2d_kernel<TI, TO>), dim3((S + 31) / 32, (R + 31) / 32, Bn), dim3(256), 0, st,
                       (const typename TI::storage*)src, (typename TO::storage*)dst, R, S);
}

static int dispatch_transpose(const char* name, const void* src, int sd, void* dst, int dd, int Bn, int R, int S,
                              hipStream_t st) {
    UC_REQUIRE(src && dst, "%s: null pointer", name);
    UC_REQUIRE(Bn > 0 && R > 0 && S > 0 && Bn <= 65535 && (R + 31) / 32 <= 65535, "%s: bad shape", name);
    if (sd == UC_F32 && dd == UC_F32) launch_transpose<F32Tag, F32Tag>(src, dst, Bn, R, S, st);
    else if (sd == UC_F32 && dd == UC_BF16) launch_transpose<F32Tag, BF16Tag>(src, dst, Bn, R, S, st);
    else if (sd == UC_BF16 && dd == UC_F32) launch_transpose<BF16Tag, F32Tag>(src, dst, Bn, R, S, st);
    else if (sd == UC_BF16 && dd == UC_BF16) launch_transpose<BF16Tag, BF16Tag>(src, dst, Bn, R, S, st);
    else if (sd == UC_F32 && dd == UC_F16) launch_transpose<F32Tag, F16Tag>(src, dst, Bn, R, S, st);
    else if (sd == UC_F16 && dd == UC_F32) launch_transpose<F16Tag, F32Tag>(src, dst, Bn, R, S, st);
    else if (sd == UC_BF16 && dd == UC_F16) launch_transpose<BF16Tag, F16Tag>(src, dst, Bn, R, S, st);
    else { uc_set_error("%s: unsupported dtypes %d -> %d", name, sd, dd); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH(name);
    return UC_OK;
}

extern "C" int uc_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C, int H, int W,
                               uc_stream_t stream) {
    return dispatch_transpose("uc_nchw_to_nhwc", src, src_dtype, dst, dst_dtype, B, C, H * W, (hipStream_t)stream);
}
extern "C" int uc_nhwc_to_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C, int H, int W,
                               uc_stream_t stream) {
    return dispatch_transpose("uc_nhwc_to_nchw", src, src_dtype, dst, dst_dtype, B, H * W, C, (hipStream_t)stream);
}

// 8 elements per work item (16-byte accesses on the 16-bit side, 2 x 16 on the fp32 side); the scalar form of rounds 1-3 ran at a
// quarter of the HBM rate.  Conversions INTO fp16 saturate and, with sat_flag, report values beyond +-65504 (see common.h).
template <typename TI, typename TO>
__global__ void convert_kernel(const typename TI::storage* __restrict__ s, typename TO::storage* __restrict__ d, int64_t n, int* __restrict__ sat_flag, int vec) {
    constexpr bool TO_F16 = sizeof(typename TO::storage) == 2 && !std::is_same<TO, BF16Tag>::value;
    const int64_t n8 = vec ? n >> 3 : 0;        // (buffers that are not 16-byte aligned: everything takes the scalar loop below)
    float amax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        if constexpr (sizeof(typename TI::storage) == 4) {
            const float4_t a = *reinterpret_cast<const float4_t*>(s + i * 8), b = *reinterpret_cast<const float4_t*>(s + i * 8 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            typename TI::storage raw[8];
            *reinterpret_cast<uint4*>(raw) = *reinterpret_cast<const uint4*>(s + i * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = TI::load(raw + k);
        }
        if constexpr (TO_F16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) amax = uc_amax(amax, fabsf(v[k]));      // (NaN-propagating: a NaN trips the flag too)
        }
        if constexpr (sizeof(typename TO::storage) == 4) {
            *reinterpret_cast<float4_t*>(d + i * 8) = (float4_t){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4_t*>(d + i * 8 + 4) = (float4_t){v[4], v[5], v[6], v[7]};
        } else {
            typename TO::storage raw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) TO::store(raw + k, v[k]);
            *reinterpret_cast<uint4*>(d + i * 8) = *reinterpret_cast<const uint4*>(raw);
        }
    }
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {   // ragged tail
        const float v = TI::load(s + i);
        if constexpr (TO_F16) amax = uc_amax(amax, fabsf(v));
        TO::store(d + i, v);
    }
    if constexpr (TO_F16) {
        if (sat_flag && !(amax <= UC_F16_MAX)) atomicOr(sat_flag, 1);
    }
}

extern "C" int uc_convert(const void* src, int sd, void* dst, int dd, int64_t n, int* sat_flag, uc_stream_t stream) {
    UC_REQUIRE(src && dst && n >= 0, "uc_convert: bad argument");
    if (n == 0) return UC_OK;
    const int vec = ((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(EW_GRID(vec ? (n + 7) / 8 : n)), b(256);
    if (sd == UC_F32 && dd == UC_BF16) hipLaunchKernelGGL((convert_kernel<F32Tag, BF16Tag>), g, b, 0, st, (const float*)src, (bf16_t*)dst, n, sat_flag, vec);
    else if (sd == UC_BF16 && dd == UC_F32) hipLaunchKernelGGL((convert_kernel<BF16Tag, F32Tag>), g, b, 0, st, (const bf16_t*)src, (float*)dst, n, sat_flag, vec);
    else if (sd == UC_F32 && dd == UC_F32) hipLaunchKernelGGL((convert_kernel<F32Tag, F32Tag>), g, b, 0, st, (const float*)src, (float*)dst, n, sat_flag, vec);
    else if (sd == UC_BF16 && dd == UC_BF16) hipLaunchKernelGGL((convert_kernel<BF16Tag, BF16Tag>), g, b, 0, st, (const bf16_t*)src, (bf16_t*)dst, n, sat_flag, vec);
    else if (sd == UC_F32 && dd == UC_F16) hipLaunchKernelGGL((convert_kernel<F32Tag, F16Tag>), g, b, 0, st, (const float*)src, (unsigned short*)dst, n, sat_flag, vec);
    else if (sd == UC_F16 && dd == UC_F32) hipLaunchKernelGGL((convert_kernel<F16Tag, F32Tag>), g, b, 0, st, (const unsigned short*)src, (float*)dst, n, sat_flag, vec);
    else if (sd == UC_BF16 && dd == UC_F16) hipLaunchKernelGGL((convert_kernel<BF16Tag, F16Tag>), g, b, 0, st, (const bf16_t*)src, (unsigned short*)dst, n, sat_flag, vec);
    else { uc_set_error("uc_convert: unsupported dtypes %d -> %d", sd, dd); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_convert");
    return UC_OK;
}

// =======================================================================================
// View positional encoding of the global / alternating multi-view transformers (info_sharing/global_attention_transformer.py
// :365-392): x[b, v*T + t, :] += pe[v, :] for the V views of T tokens each; rows past V*T (global extra tokens) are untouched.
// x fp32 [B, L, C] in place, pe fp32 [V, C].  One work item = 4 channels.
// =======================================================================================
__global__ void add_view_pe_kernel(float* __restrict__ x, const float* __restrict__ pe, int64_t B, int L, int T, int V, int C4) {
    const int64_t n = B * (int64_t)V * T * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int64_t r = i / C4;                 // (b, v, t)
        const int64_t bv = r / T;
        const int v = (int)(bv % V);
        const int64_t b = bv / V;
        float4_t* px = reinterpret_cast<float4_t*>(x + ((b * L + (int64_t)v * T + (r - bv * T)) * C4 + c) * 4);
        *px = *px + *reinterpret_cast<const float4_t*>(pe + ((int64_t)v * C4 + c) * 4);
    }
}

extern "C" int uc_add_view_pe(float* x, const float* pe, int64_t B, int L, int T, int V, int C, uc_stream_t stream) {
    UC_REQUIRE(x && pe && B >= 0 && L > 0 && T > 0 && V > 0 && (int64_t)V * T <= L && C > 0 && C % 4 == 0, "uc_add_view_pe: bad argument");
    UC_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)pe % 16 == 0, "uc_add_view_pe: 16-byte alignment");
    const int64_t n = B * (int64_t)V * T * (C / 4);
    if (n == 0) return UC_OK;
    hipLaunchKernelGGL(add_view_pe_kernel, dim3(EW_GRID(n)), dim3(256), 0, (hipStream_t)stream, x, pe, B, L, T, V, C / 4);
    UC_CHECK_LAUNCH("uc_add_view_pe");
    return UC_OK;
}

// =======================================================================================
// bf16x3 operand split: an fp32 row x[0:C] becomes the bf16 row [hi | hi | lo] (3C wide) with hi = bf16(x), lo = bf16(x - hi).
// Against weights laid out [Wh | Wl | Wh] the ordinary bf16 MFMA GEMM then accumulates xh.wh + xh.wl + xl.wh in fp32 — the
// three leading terms of the exact product (relative error ~2^-16 per term): fp32-class results at a third of the bf16
// matrix rate instead of the fp32 vector rate.  relu != 0 clamps x at zero first (the DPT residual units' ReLU-on-load cannot
// be applied to hi and lo separately).  One work item = 8 channels.
// =======================================================================================
__global__ void split_bf16x3_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t rows, int C8, int relu) {
    const int64_t n = rows * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C8;
        const int c = (int)(i - r * C8);
        const float4_t* src = reinterpret_cast<const float4_t*>(x + (r * C8 + c) * 8);
        float v[8];
        const float4_t a = src[0], b = src[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float p = v[2 * k], q = v[2 * k + 1];
            if (relu) { p = fmaxf(p, 0.f); q = fmaxf(q, 0.f); }
            const unsigned h = pack_bf16x2(p, q);
            hi[k] = h;
            lo[k] = pack_bf16x2(p - __uint_as_float(h << 16), q - __uint_as_float(h & 0xffff0000u));
        }
        bf16_t* dst = out + r * (int64_t)C8 * 24 + c * 8;
        const uint4 H = (uint4){hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<uint4*>(dst) = H;
        *reinterpret_cast<uint4*>(dst + (int64_t)C8 * 8) = H;
        *reinterpret_cast<uint4*>(dst + (int64_t)C8 * 16) = (uint4){lo[0], lo[1], lo[2], lo[3]};
    }
}

extern "C" int uc_split_bf16x3(const float* x, void* out, int64_t rows, int C, int relu, uc_stream_t stream) {
    UC_REQUIRE(x && out && rows >= 0 && C > 0 && C % 8 == 0, "uc_split_bf16x3: C must be a positive multiple of 8");
    UC_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "uc_split_bf16x3: 16-byte alignment");
    if (rows == 0) return UC_OK;
    const int64_t n = rows * (C / 8);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(EW_GRID(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, rows, C / 8, relu);
    UC_CHECK_LAUNCH("uc_split_bf16x3");
    return UC_OK;
}

// =======================================================================================
// bilinear resize, align_corners=True, NHWC.  src = dst * (in-1)/(out-1); separable weights.
// (torch: area_pixel_compute_scale -> (in-1)/(out-1) when out>1 else 0; index = scale*dst;
//  i0 = floor, i1 = min(i0+1, in-1), lambda = index - i0.)  One work item = 8 channels of one output pixel.
// =======================================================================================
template <typename Tag>
__global__ void bilinear_kernel(const typename Tag::storage* __restrict__ src, typename Tag::storage* __restrict__ dst,
                                int B, int Hi, int Wi, int C, int Ho, int Wo, int ch, int cw, float sy, float sx) {
    // grid (x: 256-thread pieces of one output row's cw * C/8 items, y: output row, z: image): no 64-bit index division
    // per item (four of them cost more than the interpolation itself and held the kernel at half the HBM rate)
    const unsigned C8 = (unsigned)C / 8u;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)cw * C8) return;
    const unsigned ox = t / C8, c8 = t - ox * C8;
    const int oy = blockIdx.y, b = blockIdx.z;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const typename Tag::storage* base = src + (int64_t)b * Hi * Wi * C + c8 * 8;
    const V8 p00 = load8<Tag>(base + ((int64_t)y0 * Wi + x0) * C);
    const V8 p01 = load8<Tag>(base + ((int64_t)y0 * Wi + x1) * C);
    const V8 p10 = load8<Tag>(base + ((int64_t)y1 * Wi + x0) * C);
    const V8 p11 = load8<Tag>(base + ((int64_t)y1 * Wi + x1) * C);
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        o.v[e] = hy * (hx * p00.v[e] + lx * p01.v[e]) + ly * (hx * p10.v[e] + lx * p11.v[e]);
    store8<Tag>(dst + (((int64_t)b * ch + oy) * cw) * C + (int64_t)t * 8, o);
}

// R output rows per work item (upsampling: vertical scale <= 1/2, so rows oy .. oy + R - 1 read at most the R/2 + 2 input rows
// ya .. ya + R/2 + 1): R + 4 loads for R outputs instead of 4 R, several times the bytes in flight per thread, the column
// index math once.  3.5 -> 4.1 (R = 2) -> 4.4-4.7 (R = 4; R = 8 is no better) TB/s of algorithmic traffic on the DPT heads' x2 resizes.
template <typename Tag, int R>
__global__ void bilinear_rows_kernel(const typename Tag::storage* __restrict__ src, typename Tag::storage* __restrict__ dst,
                                     int B, int Hi, int Wi, int C, int Ho, int Wo, int ch, int cw, float sy, float sx) {
    constexpr int RIN = R / 2 + 2;
    const unsigned C8 = (unsigned)C / 8u;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)cw * C8) return;
    const unsigned ox = t / C8, c8 = t - ox * C8;
    const int oy = R * blockIdx.y, b = blockIdx.z;
    const float fx = sx * (float)ox;
    const int x0 = (int)fx;
    const int x1 = min(x0 + 1, Wi - 1);
    const float lx = fx - (float)x0, hx = 1.f - lx;
    const int ya = (int)(sy * (float)oy);
    const typename Tag::storage* base = src + (int64_t)b * Hi * Wi * C + c8 * 8;
    float h[RIN][8];                               // horizontally interpolated input rows ya + i (clamped to the last row)
#pragma unroll
    for (int i = 0; i < RIN; ++i) {
        const int64_t y = min(ya + i, Hi - 1);
        const V8 p0 = load8<Tag>(base + (y * Wi + x0) * C), p1 = load8<Tag>(base + (y * Wi + x1) * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) h[i][e] = hx * p0.v[e] + lx * p1.v[e];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (oy + k >= ch) break;
        const float fy = sy * (float)(oy + k);
        const int y0 = (int)fy;
        const float ly = fy - (float)y0, hy = 1.f - ly;
        const int rel = y0 - ya;                   // 0 .. RIN - 2
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float top = h[0][e], bot = h[1][e];
#pragma unroll
            for (int i = 1; i < RIN - 1; ++i) { top = rel == i ? h[i][e] : top; bot = rel == i ? h[i + 1][e] : bot; }
            o.v[e] = hy * top + ly * bot;
        }
        store8<Tag>(dst + (((int64_t)b * ch + oy + k) * cw) * C + (int64_t)t * 8, o);
    }
}

extern "C" int uc_bilinear_nhwc(const void* src, void* dst, int dtype, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                int crop_h, int crop_w, uc_stream_t stream) {
    UC_REQUIRE(src && dst, "uc_bilinear_nhwc: null pointer");
    UC_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "uc_bilinear_nhwc: bad shape (C must be a multiple of 8)");
    UC_REQUIRE(crop_h > 0 && crop_h <= Ho && crop_w > 0 && crop_w <= Wo, "uc_bilinear_nhwc: bad crop");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    UC_REQUIRE(crop_h <= 65535 && B <= 65535 && (int64_t)crop_w * (C / 8) < ((int64_t)1 << 31), "uc_bilinear_nhwc: shape exceeds the launch grid");
    const dim3 grid((unsigned)(((int64_t)crop_w * (C / 8) + 255) / 256), (unsigned)crop_h, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    const int rows2 = uc_knobs().bilinear_rows2;   // rows per work item of the upsampling form: 4 (default), 2, 0 = one-row kernel
    // (bf16 only: the fp32 kernels are the verification path — its gradient fixtures sit on ReLU boundaries of the tiny test models,
    // where a 1e-7 change of the forward's rounding flips a mask and moves a small gradient tensor by 1e-3)
    if (rows2 && dtype == UC_F16 && sy <= 0.5f) {
        const dim3 gr(grid.x, (unsigned)((crop_h + 3) / 4), (unsigned)B);
        hipLaunchKernelGGL((bilinear_rows_kernel<F16Tag, 4>), gr, dim3(256), 0, st, (const unsigned short*)src, (unsigned short*)dst, B, Hi, Wi, C, Ho, Wo, crop_h, crop_w, sy, sx);
        UC_CHECK_LAUNCH("uc_bilinear_nhwc");
        return UC_OK;
    }
    if (rows2 && dtype == UC_BF16 && sy <= 0.5f) {
        const int R = rows2 == 4 ? 4 : 2;
        const dim3 gr(grid.x, (unsigned)((crop_h + R - 1) / R), (unsigned)B);
        if (R == 4) hipLaunchKernelGGL((bilinear_rows_kernel<BF16Tag, 4>), gr, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, Hi, Wi, C, Ho, Wo, crop_h, crop_w, sy, sx);
        else hipLaunchKernelGGL((bilinear_rows_kernel<BF16Tag, 2>), gr, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, Hi, Wi, C, Ho, Wo, crop_h, crop_w, sy, sx);
        UC_CHECK_LAUNCH("uc_bilinear_nhwc");
        return UC_OK;
    }
    if (dtype == UC_F32)
        hipLaunchKernelGGL((bilinear_kernel<F32Tag>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, B, Hi, Wi, C, Ho, Wo, crop_h, crop_w, sy, sx);
    else if (dtype == UC_BF16)
        hipLaunchKernelGGL((bilinear_kernel<BF16Tag>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, Hi, Wi, C, Ho, Wo, crop_h, crop_w, sy, sx);
    else if (dtype == UC_F16)
        hipLaunchKernelGGL((bilinear_kernel<F16Tag>), grid, dim3(256), 0, st, (const unsigned short*)src, (unsigned short*)dst, B, Hi, Wi, C, Ho, Wo, crop_h, crop_w, sy, sx);
    else { uc_set_error("uc_bilinear_nhwc: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_bilinear_nhwc");
    return UC_OK;
}

// =======================================================================================
// ConvTranspose2d(k=s) pixel scatter: src [B*h*w, k*k*Cout] (columns (u,v,o)) -> dst NHWC [B, k*h, k*w, Cout]
// =======================================================================================
template <typename Tag>
__global__ void convt_scatter_kernel(const typename Tag::storage* __restrict__ src, typename Tag::storage* __restrict__ dst,
                                     int B, int h, int w, int k, int Cout, int64_t items) {
    const int C8 = Cout / 8;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        // iterate in destination order
        int64_t r = it;
        const int c8 = (int)(r % C8); r /= C8;
        const int X = (int)(r % (k * w)); r /= (k * w);
        const int Y = (int)(r % (k * h));
        const int b = (int)(r / (k * h));
        const int i = Y / k, u = Y % k, j = X / k, v = X % k;
        const int64_t srow = ((int64_t)b * h + i) * w + j;
        const V8 x = load8<Tag>(src + srow * ((int64_t)k * k * Cout) + (int64_t)(u * k + v) * Cout + c8 * 8);
        store8<Tag>(dst + it * 8, x);
    }
}

extern "C" int uc_convt_scatter(const void* src, void* dst, int dtype, int B, int h, int w, int k, int Cout,
                                uc_stream_t stream) {
    UC_REQUIRE(src && dst, "uc_convt_scatter: null pointer");
    UC_REQUIRE(B > 0 && h > 0 && w > 0 && k > 0 && Cout > 0 && Cout % 8 == 0, "uc_convt_scatter: bad shape (Cout must be a multiple of 8)");
    const int64_t items = (int64_t)B * h * w * k * k * (Cout / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_F32)
        hipLaunchKernelGGL((convt_scatter_kernel<F32Tag>), dim3(EW_GRID(items)), dim3(256), 0, st, (const float*)src, (float*)dst, B, h, w, k, Cout, items);
    else if (dtype == UC_BF16)
        hipLaunchKernelGGL((convt_scatter_kernel<BF16Tag>), dim3(EW_GRID(items)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, h, w, k, Cout, items);
    else if (dtype == UC_F16)
        hipLaunchKernelGGL((convt_scatter_kernel<F16Tag>), dim3(EW_GRID(items)), dim3(256), 0, st, (const unsigned short*)src, (unsigned short*)dst, B, h, w, k, Cout, items);
    else { uc_set_error("uc_convt_scatter: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_convt_scatter");
    return UC_OK;
}

// =======================================================================================
// pixel_shuffle(P): src [B*h*w, Cout*P*P] (column c*P*P + u*P + v) -> dst fp32 NCHW [B, Cout, P*h, P*w]
// =======================================================================================
template <typename Tag>
__global__ void pixel_shuffle_kernel(const typename Tag::storage* __restrict__ src, float* __restrict__ dst, int B, int h,
                                     int w, int P, int Cout, int64_t n) {
    const int Wd = P * w, Hd = P * h;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n; it += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = it;
        const int X = (int)(r % Wd); r /= Wd;
        const int Y = (int)(r % Hd); r /= Hd;
        const int c = (int)(r % Cout);
        const int b = (int)(r / Cout);
        const int i = Y / P, u = Y % P, j = X / P, v = X % P;
        const int64_t srow = ((int64_t)b * h + i) * w + j;
        dst[it] = Tag::load(src + srow * ((int64_t)Cout * P * P) + (int64_t)c * P * P + u * P + v);
    }
}

extern "C" int uc_pixel_shuffle(const void* src, int src_dtype, float* dst, int B, int h, int w, int P, int Cout,
                                uc_stream_t stream) {
    UC_REQUIRE(src && dst && B > 0 && h > 0 && w > 0 && P > 0 && Cout > 0, "uc_pixel_shuffle: bad argument");
    const int64_t n = (int64_t)B * Cout * P * h * P * w;
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == UC_F32)
        hipLaunchKernelGGL((pixel_shuffle_kernel<F32Tag>), dim3(EW_GRID(n)), dim3(256), 0, st, (const float*)src, dst, B, h, w, P, Cout, n);
    else if (src_dtype == UC_BF16)
        hipLaunchKernelGGL((pixel_shuffle_kernel<BF16Tag>), dim3(EW_GRID(n)), dim3(256), 0, st, (const bf16_t*)src, dst, B, h, w, P, Cout, n);
    else if (src_dtype == UC_F16)
        hipLaunchKernelGGL((pixel_shuffle_kernel<F16Tag>), dim3(EW_GRID(n)), dim3(256), 0, st, (const unsigned short*)src, dst, B, h, w, P, Cout, n);
    else { uc_set_error("uc_pixel_shuffle: bad dtype %d", src_dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_pixel_shuffle");
    return UC_OK;
}

// =======================================================================================
// pointmap + confidence adaptor ("exp","exp") fused with the BCHW -> BHWC permute
// =======================================================================================
__global__ void pointmap_adaptor_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sp,
                                        float* __restrict__ pts, float* __restrict__ conf, int64_t HW, int64_t n,
                                        float vmin, float vspan) {
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = it / HW, pix = it % HW;
        const float* px = x + b * sb + pix * sp;
        const float X = px[0], Y = px[sc], Z = px[2 * sc], Cf = px[3 * sc];
        // torch.norm(dim=1): sqrt of the sum of squares
        const float d = sqrtf(X * X + Y * Y + Z * Z);
        const float s = expm1f(d) / fmaxf(d, 1e-8f);
        pts[it * 3 + 0] = X * s;  // (xyz / clip(d)) * expm1(d); rounding differs from the reference only in op order
        pts[it * 3 + 1] = Y * s;
        pts[it * 3 + 2] = Z * s;
        conf[it] = vmin + fminf(expf(Cf), vspan);
    }
}

extern "C" int uc_pointmap_adaptor(const float* x, int64_t x_sb, int64_t x_sc, int64_t x_sp, float* pts, float* conf,
                                   int B, int H, int W, float conf_vmin, float conf_vmax, uc_stream_t stream) {
    UC_REQUIRE(x && pts && conf && B > 0 && H > 0 && W > 0, "uc_pointmap_adaptor: bad argument");
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(pointmap_adaptor_kernel, dim3(EW_GRID(n)), dim3(256), 0, (hipStream_t)stream, x, x_sb, x_sc, x_sp,
                       pts, conf, (int64_t)H * W, n, conf_vmin, conf_vmax - conf_vmin);
    UC_CHECK_LAUNCH("uc_pointmap_adaptor");
    return UC_OK;
}

// =======================================================================================
// 1x1 conv Cin -> 4 (+bias) on NHWC features: one thread per pixel, weights in LDS (broadcast reads)
// =======================================================================================
template <typename Tag>
__global__ __launch_bounds__(256) void conv1x1_to4_kernel(const typename Tag::storage* __restrict__ feat,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int64_t npix, int Cin) {
    __shared__ float ws[4][256];
    for (int i = threadIdx.x; i < 4 * Cin; i += blockDim.x) ws[i / Cin][i % Cin] = w[i];
    __syncthreads();
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        float a0 = bias[0], a1 = bias[1], a2 = bias[2], a3 = bias[3];
        const typename Tag::storage* f = feat + pix * Cin;
        for (int c = 0; c < Cin; c += 8) {
            const V8 x = load8<Tag>(f + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0 = fmaf(x.v[e], ws[0][c + e], a0);
                a1 = fmaf(x.v[e], ws[1][c + e], a1);
                a2 = fmaf(x.v[e], ws[2][c + e], a2);
                a3 = fmaf(x.v[e], ws[3][c + e], a3);
            }
        }
        *reinterpret_cast<float4_t*>(out + pix * 4) = (float4_t){a0, a1, a2, a3};
    }
}

// Coalesced form for Cin = 8 * 2^k (<= 256): C8 = Cin/8 consecutive lanes share a pixel, each loads ONE 16-byte chunk (so a
// wave reads 1 KiB of consecutive memory per load instead of 64 scattered 16-byte pieces), keeps its 4x8 weights in
// registers and the 4 partial sums are folded across the C8 lanes with xor-shuffles.
template <typename Tag, int C8>
__global__ __launch_bounds__(256) void conv1x1_to4_coop_kernel(const typename Tag::storage* __restrict__ feat,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, int64_t npix) {
    constexpr int Cin = C8 * 8;
    constexpr int PPB = 256 / C8;                  // pixels per block-iteration
    const int chunk = threadIdx.x % C8, pl = threadIdx.x / C8;
    float wr[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[o][e] = w[o * Cin + chunk * 8 + e];
    const float4_t b4 = *reinterpret_cast<const float4_t*>(bias);
    for (int64_t pix = (int64_t)blockIdx.x * PPB + pl; pix < npix; pix += (int64_t)gridDim.x * PPB) {
        const V8 x = load8<Tag>(feat + pix * Cin + chunk * 8);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[0] = fmaf(x.v[e], wr[0][e], a[0]);
            a[1] = fmaf(x.v[e], wr[1][e], a[1]);
            a[2] = fmaf(x.v[e], wr[2][e], a[2]);
            a[3] = fmaf(x.v[e], wr[3][e], a[3]);
        }
#pragma unroll
        for (int off = C8 / 2; off > 0; off >>= 1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) a[o] += __shfl_xor(a[o], off, 64);
        }
        if (chunk == 0)
            *reinterpret_cast<float4_t*>(out + pix * 4) = (float4_t){a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w};
    }
}

template <typename Tag>
static bool launch_conv1x1_to4_coop(const void* feat, const float* w, const float* b, float* out, int64_t npix, int Cin, hipStream_t st) {
    const int C8 = Cin / 8;
#define UC_C1X1(C8_)                                                                                                         \
    case C8_: {                                                                                                              \
        const unsigned grid = (unsigned)min((int64_t)65536, ceil_div64(npix, 256 / C8_));                                    \
        hipLaunchKernelGGL((conv1x1_to4_coop_kernel<Tag, C8_>), dim3(grid), dim3(256), 0, st,                                \
                           (const typename Tag::storage*)feat, w, b, out, npix);                                             \
        return true;                                                                                                         \
    }
    switch (C8) {
        UC_C1X1(1) UC_C1X1(2) UC_C1X1(4) UC_C1X1(8) UC_C1X1(16) UC_C1X1(32)
        default: return false;
    }
#undef UC_C1X1
}

extern "C" int uc_conv1x1_to4(const void* feat, int dtype, const float* w, const float* b, float* out, int64_t npix,
                              int Cin, uc_stream_t stream) {
    UC_REQUIRE(feat && w && b && out && npix > 0, "uc_conv1x1_to4: bad argument");
    UC_REQUIRE(Cin > 0 && Cin <= 256 && Cin % 8 == 0, "uc_conv1x1_to4: Cin must be a multiple of 8 and <= 256 (got %d)", Cin);
    hipStream_t st = (hipStream_t)stream;
    if (((uintptr_t)b % 16 == 0) && ((dtype == UC_F32 && launch_conv1x1_to4_coop<F32Tag>(feat, w, b, out, npix, Cin, st)) ||
                                      (dtype == UC_BF16 && launch_conv1x1_to4_coop<BF16Tag>(feat, w, b, out, npix, Cin, st)) ||
                                      (dtype == UC_F16 && launch_conv1x1_to4_coop<F16Tag>(feat, w, b, out, npix, Cin, st)))) {
        UC_CHECK_LAUNCH("uc_conv1x1_to4");
        return UC_OK;
    }
    if (dtype == UC_F32)
        hipLaunchKernelGGL((conv1x1_to4_kernel<F32Tag>), dim3(EW_GRID(npix)), dim3(256), 0, st, (const float*)feat, w, b, out, npix, Cin);
    else if (dtype == UC_BF16)
        hipLaunchKernelGGL((conv1x1_to4_kernel<BF16Tag>), dim3(EW_GRID(npix)), dim3(256), 0, st, (const bf16_t*)feat, w, b, out, npix, Cin);
    else if (dtype == UC_F16)
        hipLaunchKernelGGL((conv1x1_to4_kernel<F16Tag>), dim3(EW_GRID(npix)), dim3(256), 0, st, (const unsigned short*)feat, w, b, out, npix, Cin);
    else { uc_set_error("uc_conv1x1_to4: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_conv1x1_to4");
    return UC_OK;
}

// =======================================================================================
// Token-sequence assembly of cls/register-token ViTs (DINOv2): [cls + pos0 | registers | patches + pos] and the
// reverse split.  fp32 residual stream, float4 per lane.
// =======================================================================================
__global__ void assemble_tokens_kernel(const float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ reg,
                                       const float* __restrict__ pos, float* __restrict__ out, int B, int hw, int R, int D4,
                                       int64_t items) {
    const int Nt = 1 + R + hw;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(it % D4);
        const int n = (int)((it / D4) % Nt);
        const int b = (int)(it / ((int64_t)D4 * Nt));
        float4_t v;
        if (n == 0) {
            const float4_t c = reinterpret_cast<const float4_t*>(cls)[d4], p = reinterpret_cast<const float4_t*>(pos)[d4];
            v = (float4_t){c.x + p.x, c.y + p.y, c.z + p.z, c.w + p.w};
        } else if (n <= R) {
            v = reinterpret_cast<const float4_t*>(reg)[(int64_t)(n - 1) * D4 + d4];
        } else {
            const int i = n - 1 - R;
            const float4_t t = reinterpret_cast<const float4_t*>(tok)[((int64_t)b * hw + i) * D4 + d4];
            const float4_t p = reinterpret_cast<const float4_t*>(pos)[(int64_t)(1 + i) * D4 + d4];
            v = (float4_t){t.x + p.x, t.y + p.y, t.z + p.z, t.w + p.w};
        }
        reinterpret_cast<float4_t*>(out)[it] = v;
    }
}

extern "C" int uc_assemble_tokens(const float* tok, const float* cls, const float* reg, const float* pos, float* out, int B,
                                  int hw, int R, int D, uc_stream_t stream) {
    UC_REQUIRE(tok && cls && pos && out && (R == 0 || reg), "uc_assemble_tokens: null pointer");
    UC_REQUIRE(B > 0 && hw > 0 && R >= 0 && D > 0 && D % 4 == 0, "uc_assemble_tokens: bad shape (D must be a multiple of 4)");
    const int64_t items = (int64_t)B * (1 + R + hw) * (D / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(EW_GRID(items)), dim3(256), 0, (hipStream_t)stream, tok, cls, reg, pos, out, B,
                       hw, R, D / 4, items);
    UC_CHECK_LAUNCH("uc_assemble_tokens");
    return UC_OK;
}

__global__ void token_slice_kernel(const float* __restrict__ src, float* __restrict__ dst, int Ns, int Nd, int src_off, int dst_off,
                                   int n, int D4, int64_t items) {
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(it % D4);
        const int i = (int)((it / D4) % n);
        const int b = (int)(it / ((int64_t)D4 * n));
        reinterpret_cast<float4_t*>(dst)[((int64_t)b * Nd + dst_off + i) * D4 + d4] =
            reinterpret_cast<const float4_t*>(src)[((int64_t)b * Ns + src_off + i) * D4 + d4];
    }
}

extern "C" int uc_token_slice(const float* src, float* dst, int B, int Ns, int Nd, int src_off, int dst_off, int n, int D,
                              uc_stream_t stream) {
    UC_REQUIRE(src && dst && B > 0 && n > 0 && D > 0 && D % 4 == 0, "uc_token_slice: bad argument (D must be a multiple of 4)");
    UC_REQUIRE(src_off >= 0 && dst_off >= 0 && src_off + n <= Ns && dst_off + n <= Nd, "uc_token_slice: slice out of range");
    const int64_t items = (int64_t)B * n * (D / 4);
    hipLaunchKernelGGL(token_slice_kernel, dim3(EW_GRID(items)), dim3(256), 0, (hipStream_t)stream, src, dst, Ns, Nd, src_off, dst_off,
                       n, D / 4, items);
    UC_CHECK_LAUNCH("uc_token_slice");
    return UC_OK;
}


// =======================================================================================
// Adaptor "channel programs" (reference: prediction_heads/adaptors.py:25-2300).  Every adaptor of the reference splits the
// decoded channels, applies a small per-pixel transform to each group and concatenates the results: here the whole
// composition is ONE pass — a list of up to UC_ADAPTOR_MAX_SEGS segments {op, input channels, output channels, parameters,
// clip} evaluated per pixel.  x: fp32 BCHW-shaped (strides sb, sc, sw; rows dense), out: fp32 NHWC [B,H,W,Cout].
// =======================================================================================
struct AdaptorProgram { uc_adaptor_seg seg[UC_ADAPTOR_MAX_SEGS]; int nseg; };

__device__ __forceinline__ float ad_clip(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ void adaptor_program_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sw, float* __restrict__ out,
                                       int64_t npix, int HW, int W, int Cout, AdaptorProgram prog) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int pix = (int)(i - b * HW);
        const float* px = x + b * sb + (int64_t)pix * sw;
        float* po = out + i * Cout;
        for (int s = 0; s < prog.nseg; ++s) {
            const uc_adaptor_seg g = prog.seg[s];
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < g.n) v[k] = px[(int64_t)(g.c0 + k) * sc];
            float* o = po + g.o0;
            switch (g.op) {
                case UC_AD_ELEM:            // linear / square / exp per channel, then clip
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < g.n) { const float t = g.mode == 1 ? v[k] * v[k] : (g.mode == 2 ? expf(v[k]) : v[k]); o[k] = ad_clip(t, g.vmin, g.vmax); }
                    break;
                case UC_AD_NORM: {          // direction x f(distance): f = d^2 ("square") | expm1(d) ("exp"), then clip
                    float d2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < g.n) d2 += v[k] * v[k];
                    const float d = sqrtf(d2);
                    const float f = (g.mode == 1 ? d * d : expm1f(d)) / fmaxf(d, 1e-8f);
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < g.n) o[k] = ad_clip(v[k] * f, g.vmin, g.vmax);
                    break;
                }
                case UC_AD_ZEXP: {          // (x e^z, y e^z, e^z), then clip
                    const float z = expf(v[2]);
                    o[0] = ad_clip(v[0] * z, g.vmin, g.vmax); o[1] = ad_clip(v[1] * z, g.vmin, g.vmax); o[2] = ad_clip(z, g.vmin, g.vmax);
                    break;
                }
                case UC_AD_DIR: {           // clip, optional clamp of the last channel from below (flag 1), then unit norm (flag 2) or last channel = 1 (flag 4)
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < g.n) v[k] = ad_clip(v[k], g.vmin, g.vmax);
                    if (g.flags & 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k == g.n - 1) v[k] = fmaxf(v[k], g.p[0]);
                    }
                    float sc_ = 1.f;
                    if (g.flags & 2) {
                        float d2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k < g.n) d2 += v[k] * v[k];
                        sc_ = 1.f / fmaxf(sqrtf(d2), 1e-8f);
                    } else if (g.flags & 4) {
                        float last = 1.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k == g.n - 1) last = v[k];
                        sc_ = 1.f / last;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < g.n) o[k] = v[k] * sc_;
                    break;
                }
                case UC_AD_CONF_EXP:        // vmin + min(exp(x), vmax - vmin)
                    o[0] = g.vmin + fminf(expf(v[0]), g.vmax - g.vmin);
                    break;
                case UC_AD_CONF_SIGMOID:    // sigmoid(x) (vmax - vmin) + vmin
                    o[0] = (1.f / (1.f + expf(-v[0]))) * (g.vmax - g.vmin) + g.vmin;
                    break;
                case UC_AD_MASK:            // (logits, sigmoid(logits))
                    o[0] = v[0]; o[1] = 1.f / (1.f + expf(-v[0]));
                    break;
                case UC_AD_FLOW:            // x std + mean per channel (already scaled to the output shape by the caller)
                    o[0] = v[0] * g.p[0] + g.p[1]; o[1] = v[1] * g.p[2] + g.p[3];
                    break;
                case UC_AD_FLOWCOORD: {     // 0.5 (x + 1) (W, H) + 0.5 - (pixel centre)
                    const int yy = pix / W, xx = pix - yy * W;
                    o[0] = 0.5f * (v[0] + 1.f) * g.p[0] + 0.5f - ((float)xx + 0.5f);
                    o[1] = 0.5f * (v[1] + 1.f) * g.p[1] + 0.5f - ((float)yy + 0.5f);
                    break;
                }
                case UC_AD_COV2D: {         // (c1, c2, s) -> covariance (3), log det (1), inverse covariance (3); p[0] = offset of c1, c2
                    const float c1 = v[0] + g.p[0], c2 = v[1] + g.p[0];
                    const float th = tanhf(v[2]);
                    const float de = 0.5f * (c1 + c2);
                    const float om = 1.f - th * th + 1e-8f;
                    const float ic = 1.f / om;
                    o[0] = expf(c1); o[1] = expf(c2); o[2] = th * expf(de);
                    o[3] = c1 + c2 + logf(om);
                    o[4] = ic * expf(-c1); o[5] = ic * expf(-c2); o[6] = -ic * th * expf(-de);
                    break;
                }
                default: break;
            }
        }
    }
}

extern "C" int uc_adaptor_program(const float* x, int64_t sb, int64_t sc, int64_t sw, float* out, int B, int H, int W, int Cout,
                                  const uc_adaptor_seg* segs, int nseg, uc_stream_t stream) {
    UC_REQUIRE(x && out && segs && B > 0 && H > 0 && W > 0 && Cout > 0, "uc_adaptor_program: bad argument");
    UC_REQUIRE(nseg > 0 && nseg <= UC_ADAPTOR_MAX_SEGS, "uc_adaptor_program: 1..%d segments", UC_ADAPTOR_MAX_SEGS);
    AdaptorProgram prog;
    prog.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        const uc_adaptor_seg& g = segs[s];
        const int nout = g.op == UC_AD_MASK ? 2 : (g.op == UC_AD_COV2D ? 7 : g.n);
        UC_REQUIRE(g.op >= UC_AD_ELEM && g.op <= UC_AD_COV2D && g.n >= 1 && g.n <= 4 && g.c0 >= 0 && g.o0 >= 0 && g.o0 + nout <= Cout,
                   "uc_adaptor_program: bad segment %d", s);
        UC_REQUIRE(((g.op != UC_AD_ZEXP && g.op != UC_AD_COV2D) || g.n == 3) && ((g.op != UC_AD_FLOW && g.op != UC_AD_FLOWCOORD) || g.n == 2) &&
                       ((g.op != UC_AD_CONF_EXP && g.op != UC_AD_CONF_SIGMOID && g.op != UC_AD_MASK) || g.n == 1),
                   "uc_adaptor_program: segment %d has the wrong channel count for its op", s);
        prog.seg[s] = g;
    }
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(adaptor_program_kernel, dim3(EW_GRID(npix)), dim3(256), 0, (hipStream_t)stream, x, sb, sc, sw, out, npix, H * W, W, Cout, prog);
    UC_CHECK_LAUNCH("uc_adaptor_program");
    return UC_OK;
}

// Gradient of adaptor_program_kernel with respect to x for an arbitrary downstream loss: dout fp32 NHWC [B,H,W,Cout] -> dx with the
// strides of x.  The masks follow torch's autograd for the reference's composition of ops: clip / clamp pass the gradient where
// the unclipped value lies inside [vmin, vmax] (bounds included), norm() has gradient 0 at the origin.  Input channels no segment
// reads (bit clear in `covered`) get a zero gradient.
__device__ __forceinline__ float ad_pass(float t, float lo, float hi) { return (t >= lo && t <= hi) ? 1.f : 0.f; }

__global__ void adaptor_program_bwd_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sw, const float* __restrict__ dout,
                                           float* __restrict__ dx, int64_t npix, int HW, int Cin, int Cout, unsigned long long covered,
                                           AdaptorProgram prog) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int pix = (int)(i - b * HW);
        const int64_t base = b * sb + (int64_t)pix * sw;
        const float* px = x + base;
        float* pd = dx + base;
        const float* pg = dout + i * Cout;
        for (int c = 0; c < Cin; ++c)
            if (!((covered >> c) & 1ull)) pd[(int64_t)c * sc] = 0.f;
        for (int s = 0; s < prog.nseg; ++s) {
            const uc_adaptor_seg sg = prog.seg[s];
            float v[4] = {0.f, 0.f, 0.f, 0.f}, g[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
            const int nout = sg.op == UC_AD_MASK ? 2 : (sg.op == UC_AD_COV2D ? 7 : sg.n);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < sg.n) v[k] = px[(int64_t)(sg.c0 + k) * sc];
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < nout) g[k] = pg[sg.o0 + k];
            switch (sg.op) {
                case UC_AD_ELEM:
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float e = expf(v[k]);
                        const float t = sg.mode == 1 ? v[k] * v[k] : (sg.mode == 2 ? e : v[k]);
                        const float dt = sg.mode == 1 ? 2.f * v[k] : (sg.mode == 2 ? e : 1.f);
                        d[k] = g[k] * ad_pass(t, sg.vmin, sg.vmax) * dt;
                    }
                    break;
                case UC_AD_NORM: {          // y = v f(r) / max(r, 1e-8): dv = s g' + v (v . g') (f'(r) r - f(r)) / r^3
                    float r2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < sg.n) r2 += v[k] * v[k];
                    const float r = sqrtf(r2), rc = fmaxf(r, 1e-8f);
                    const float em1 = expm1f(r);
                    const float f = sg.mode == 1 ? r * r : em1;
                    const float sl = f / rc;
                    // (f' r - f) / r^2: 1 for r^2; for expm1 the series below 1e-2 (the quotient cancels there)
                    const float q = sg.mode == 1 ? 1.f : (r > 1e-2f ? ((em1 + 1.f) * r - em1) / (r * r) : 0.5f + r * (1.f / 3.f));
                    float dot = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < sg.n) { g[k] *= ad_pass(v[k] * sl, sg.vmin, sg.vmax); dot += v[k] * g[k]; }
                    const float kk = (r >= 1e-8f) ? dot * q / rc : 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = sl * g[k] + v[k] * kk;
                    break;
                }
                case UC_AD_ZEXP: {
                    const float z = expf(v[2]);
                    const float g0 = g[0] * ad_pass(v[0] * z, sg.vmin, sg.vmax), g1 = g[1] * ad_pass(v[1] * z, sg.vmin, sg.vmax);
                    const float g2 = g[2] * ad_pass(z, sg.vmin, sg.vmax);
                    d[0] = g0 * z; d[1] = g1 * z; d[2] = (g0 * v[0] + g1 * v[1] + g2) * z;
                    break;
                }
                case UC_AD_DIR: {
                    float c[4], m[4];          // clipped values and the pass mask of clip (+ clamp of the last channel)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { m[k] = ad_pass(v[k], sg.vmin, sg.vmax); c[k] = ad_clip(v[k], sg.vmin, sg.vmax); }
                    if (sg.flags & 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k == sg.n - 1) { m[k] *= (c[k] >= sg.p[0]) ? 1.f : 0.f; c[k] = fmaxf(c[k], sg.p[0]); }
                    }
                    float dc[4] = {g[0], g[1], g[2], g[3]};
                    if (sg.flags & 2) {          // y = c / max(|c|, 1e-8)
                        float r2 = 0.f, dot = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k < sg.n) { r2 += c[k] * c[k]; dot += c[k] * g[k]; }
                        const float r = sqrtf(r2), rc = fmaxf(r, 1e-8f);
                        const float kk = (r >= 1e-8f) ? dot / (rc * rc * r) : 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) dc[k] = g[k] / rc - c[k] * kk;
                    } else if (sg.flags & 4) {   // y = c / c_last
                        float last = 1.f, dot = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { if (k == sg.n - 1) last = c[k]; if (k < sg.n) dot += c[k] * g[k]; }
                        const float il = 1.f / last;
#pragma unroll
                        for (int k = 0; k < 4; ++k) dc[k] = g[k] * il - (k == sg.n - 1 ? dot * il * il : 0.f);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = dc[k] * m[k];
                    break;
                }
                case UC_AD_CONF_EXP: {
                    const float e = expf(v[0]);
                    d[0] = (e <= sg.vmax - sg.vmin) ? g[0] * e : 0.f;
                    break;
                }
                case UC_AD_CONF_SIGMOID: {
                    const float sgm = 1.f / (1.f + expf(-v[0]));
                    d[0] = g[0] * (sg.vmax - sg.vmin) * sgm * (1.f - sgm);
                    break;
                }
                case UC_AD_MASK: {
                    const float sgm = 1.f / (1.f + expf(-v[0]));
                    d[0] = g[0] + g[1] * sgm * (1.f - sgm);
                    break;
                }
                case UC_AD_FLOW:
                    d[0] = g[0] * sg.p[0]; d[1] = g[1] * sg.p[2];
                    break;
                case UC_AD_FLOWCOORD:
                    d[0] = 0.5f * sg.p[0] * g[0]; d[1] = 0.5f * sg.p[1] * g[1];
                    break;
                case UC_AD_COV2D: {
                    const float c1 = v[0] + sg.p[0], c2 = v[1] + sg.p[0];
                    const float th = tanhf(v[2]);
                    const float de = 0.5f * (c1 + c2);
                    const float ic = 1.f / (1.f - th * th + 1e-8f);
                    const float e1 = expf(c1), e2 = expf(c2), ed = expf(de), n1 = expf(-c1), n2 = expf(-c2), nd = expf(-de);
                    const float o2 = th * ed, o4 = ic * n1, o5 = ic * n2, o6 = -ic * th * nd;
                    const float dic = 2.f * th * ic * ic;          // d ic / d th
                    d[0] = g[0] * e1 + 0.5f * g[2] * o2 + g[3] - g[4] * o4 - 0.5f * g[6] * o6;
                    d[1] = g[1] * e2 + 0.5f * g[2] * o2 + g[3] - g[5] * o5 - 0.5f * g[6] * o6;
                    const float dth = g[2] * ed - g[3] * 2.f * th * ic + g[4] * dic * n1 + g[5] * dic * n2 - g[6] * nd * (ic + th * dic);
                    d[2] = dth * (1.f - th * th);
                    break;
                }
                default: break;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < sg.n) pd[(int64_t)(sg.c0 + k) * sc] = d[k];
        }
    }
}

extern "C" int uc_adaptor_program_bwd(const float* x, int64_t sb, int64_t sc, int64_t sw, const float* dout, float* dx, int B, int H,
                                      int W, int Cin, int Cout, const uc_adaptor_seg* segs, int nseg, uc_stream_t stream) {
    UC_REQUIRE(x && dout && dx && segs && B > 0 && H > 0 && W > 0 && Cout > 0, "uc_adaptor_program_bwd: bad argument");
    UC_REQUIRE(Cin > 0 && Cin <= 64, "uc_adaptor_program_bwd: 1..64 input channels");
    UC_REQUIRE(nseg > 0 && nseg <= UC_ADAPTOR_MAX_SEGS, "uc_adaptor_program_bwd: 1..%d segments", UC_ADAPTOR_MAX_SEGS);
    AdaptorProgram prog;
    prog.nseg = nseg;
    unsigned long long covered = 0ull;
    for (int s = 0; s < nseg; ++s) {
        const uc_adaptor_seg& g = segs[s];
        const int nout = g.op == UC_AD_MASK ? 2 : (g.op == UC_AD_COV2D ? 7 : g.n);
        UC_REQUIRE(g.op >= UC_AD_ELEM && g.op <= UC_AD_COV2D && g.n >= 1 && g.n <= 4 && g.c0 >= 0 && g.c0 + g.n <= Cin && g.o0 >= 0 &&
                       g.o0 + nout <= Cout, "uc_adaptor_program_bwd: bad segment %d", s);
        UC_REQUIRE(((g.op != UC_AD_ZEXP && g.op != UC_AD_COV2D) || g.n == 3) && ((g.op != UC_AD_FLOW && g.op != UC_AD_FLOWCOORD) || g.n == 2) &&
                       ((g.op != UC_AD_CONF_EXP && g.op != UC_AD_CONF_SIGMOID && g.op != UC_AD_MASK) || g.n == 1),
                   "uc_adaptor_program_bwd: segment %d has the wrong channel count for its op", s);
        const unsigned long long bits = ((1ull << g.n) - 1ull) << g.c0;
        UC_REQUIRE(!(covered & bits), "uc_adaptor_program_bwd: segment %d reads a channel another segment reads (gradients would have to be summed)", s);
        covered |= bits;
        prog.seg[s] = g;
    }
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(adaptor_program_bwd_kernel, dim3(EW_GRID(npix)), dim3(256), 0, (hipStream_t)stream, x, sb, sc, sw, dout, dx, npix,
                       H * W, Cin, Cout, covered, prog);
    UC_CHECK_LAUNCH("uc_adaptor_program_bwd");
    return UC_OK;
}
