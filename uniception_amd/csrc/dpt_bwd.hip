// Backward-only data-movement kernels of the DPT head (gfx950): the adjoint of the align_corners bilinear resize, the
// inverse pixel scatter of ConvTranspose2d(k=s), the transposed im2col that feeds the 3x3 weight-gradient GEMM, the
// zero-stuffing that turns a stride-2 conv's data gradient into a stride-1 one, and the backward of the 4-channel head conv.
// All NHWC; 8 channels (16 B bf16) per lane where the channel axis is contiguous.
#include "common.h"

struct V8b { float v[8]; };
template <typename Tag> __device__ __forceinline__ V8b ld8(const typename Tag::storage* p);
template <> __device__ __forceinline__ V8b ld8<F32Tag>(const float* p) {
    V8b r;
    const float4_t a = *reinterpret_cast<const float4_t*>(p);
    const float4_t b = *reinterpret_cast<const float4_t*>(p + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <> __device__ __forceinline__ V8b ld8<BF16Tag>(const bf16_t* p) {
    V8b r;
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
template <typename Tag> __device__ __forceinline__ void st8(typename Tag::storage* p, const V8b& r);
template <> __device__ __forceinline__ void st8<F32Tag>(float* p, const V8b& r) {
    *reinterpret_cast<float4_t*>(p) = (float4_t){r.v[0], r.v[1], r.v[2], r.v[3]};
    *reinterpret_cast<float4_t*>(p + 4) = (float4_t){r.v[4], r.v[5], r.v[6], r.v[7]};
}
template <> __device__ __forceinline__ void st8<BF16Tag>(bf16_t* p, const V8b& r) {
    uint4 u;
    u.x = pack_bf16x2(r.v[0], r.v[1]); u.y = pack_bf16x2(r.v[2], r.v[3]);
    u.z = pack_bf16x2(r.v[4], r.v[5]); u.w = pack_bf16x2(r.v[6], r.v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}

#define DB_GRID(n_items) ((unsigned)min((int64_t)65536 * 4, ceil_div64((n_items), 256)))

// =================================================================================================================
// bilinear backward (exact adjoint of bilinear_kernel in elementwise.hip, including its index clamps): gather form, one
// work item = 8 channels of one INPUT pixel, looping over the output pixels whose two taps can touch it.
// =================================================================================================================
__device__ __forceinline__ void tap_range(int i, float s, int n_out, int& lo, int& hi) {
    if (s <= 0.f) { lo = 0; hi = n_out - 1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / s) - 1);
    hi = min(n_out - 1, (int)ceilf((float)(i + 1) / s) + 1);
}

template <typename Tag>
__global__ void bilinear_bwd_kernel(const typename Tag::storage* __restrict__ dy, typename Tag::storage* __restrict__ dx, int B,
                                    int Hi, int Wi, int C, int ch, int cw, float sy, float sx, int64_t items) {
    const int C8 = C / 8;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = it;
        const int c8 = (int)(r % C8); r /= C8;
        const int xi = (int)(r % Wi); r /= Wi;
        const int yi = (int)(r % Hi);
        const int b = (int)(r / Hi);
        int ylo, yhi, xlo, xhi;
        tap_range(yi, sy, ch, ylo, yhi);
        tap_range(xi, sx, cw, xlo, xhi);
        V8b acc;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc.v[e] = 0.f;
        const typename Tag::storage* base = dy + (int64_t)b * ch * cw * C + c8 * 8;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float fy = sy * (float)oy;
            const int y0 = (int)fy, y1 = min(y0 + 1, Hi - 1);
            const float ly = fy - (float)y0;
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float fx = sx * (float)ox;
                const int x0 = (int)fx, x1 = min(x0 + 1, Wi - 1);
                const float lx = fx - (float)x0;
                const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
                if (wx == 0.f) continue;
                const V8b g = ld8<Tag>(base + ((int64_t)oy * cw + ox) * C);
                const float w = wy * wx;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc.v[e] = fmaf(w, g.v[e], acc.v[e]);
            }
        }
        st8<Tag>(dx + it * 8, acc);
    }
}

extern "C" int uc_bilinear_nhwc_bwd(const void* dy, void* dx, int dtype, int B, int Hi, int Wi, int C, int Ho, int Wo, int crop_h,
                                    int crop_w, uc_stream_t stream) {
    UC_REQUIRE(dy && dx, "uc_bilinear_nhwc_bwd: null pointer");
    UC_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "uc_bilinear_nhwc_bwd: bad shape (C must be a multiple of 8)");
    UC_REQUIRE(crop_h > 0 && crop_h <= Ho && crop_w > 0 && crop_w <= Wo, "uc_bilinear_nhwc_bwd: bad crop");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const int64_t items = (int64_t)B * Hi * Wi * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_F32)
        hipLaunchKernelGGL((bilinear_bwd_kernel<F32Tag>), dim3(DB_GRID(items)), dim3(256), 0, st, (const float*)dy, (float*)dx, B, Hi, Wi, C, crop_h, crop_w, sy, sx, items);
    else if (dtype == UC_BF16)
        hipLaunchKernelGGL((bilinear_bwd_kernel<BF16Tag>), dim3(DB_GRID(items)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, Hi, Wi, C, crop_h, crop_w, sy, sx, items);
    else { uc_set_error("uc_bilinear_nhwc_bwd: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_bilinear_nhwc_bwd");
    return UC_OK;
}

// =================================================================================================================
// inverse of uc_convt_scatter: src NHWC [B, k*h, k*w, Cout] -> dst [B*h*w, k*k*Cout] (columns (u,v,o))
// =================================================================================================================
template <typename Tag>
__global__ void convt_gather_kernel(const typename Tag::storage* __restrict__ src, typename Tag::storage* __restrict__ dst, int B,
                                    int h, int w, int k, int Cout, int64_t items) {
    const int C8 = Cout / 8;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = it;   // source (image) order, so reads are coalesced and each destination element is written once
        const int c8 = (int)(r % C8); r /= C8;
        const int X = (int)(r % (k * w)); r /= (k * w);
        const int Y = (int)(r % (k * h));
        const int b = (int)(r / (k * h));
        const int i = Y / k, u = Y % k, j = X / k, v = X % k;
        const int64_t drow = ((int64_t)b * h + i) * w + j;
        st8<Tag>(dst + drow * ((int64_t)k * k * Cout) + (int64_t)(u * k + v) * Cout + c8 * 8, ld8<Tag>(src + it * 8));
    }
}

extern "C" int uc_convt_gather(const void* src, void* dst, int dtype, int B, int h, int w, int k, int Cout, uc_stream_t stream) {
    UC_REQUIRE(src && dst && B > 0 && h > 0 && w > 0 && k > 0 && Cout > 0 && Cout % 8 == 0, "uc_convt_gather: bad argument (Cout must be a multiple of 8)");
    const int64_t items = (int64_t)B * h * k * w * k * (Cout / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_F32) hipLaunchKernelGGL((convt_gather_kernel<F32Tag>), dim3(DB_GRID(items)), dim3(256), 0, st, (const float*)src, (float*)dst, B, h, w, k, Cout, items);
    else if (dtype == UC_BF16) hipLaunchKernelGGL((convt_gather_kernel<BF16Tag>), dim3(DB_GRID(items)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, h, w, k, Cout, items);
    else { uc_set_error("uc_convt_gather: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_convt_gather");
    return UC_OK;
}

// =================================================================================================================
// transposed im2col for the 3x3 (pad 1) weight gradient: dst[(tap*Cin + c), p] = act(x[b, oy*s-1+ky, ox*s-1+kx, c]),
// p = (b*Ho + oy)*Wo + ox, zero outside the image and for p in [npix, ld).  64x64 tiles through LDS per tap.
// =================================================================================================================
template <typename Tag>
__global__ __launch_bounds__(256) void im2col_t_kernel(const typename Tag::storage* __restrict__ x, typename Tag::storage* __restrict__ dst,
                                                        int B, int H, int W, int Cin, int stride, int Ho, int Wo, int relu,
                                                        int64_t npix, int64_t ld) {
    __shared__ float tile[64][65];
    const int tap = blockIdx.z;
    const int ky = tap / 3, kx = tap % 3;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t p = p0 + ty + 4 * k;
        const int c = c0 + tx;
        float v = 0.f;
        if (p < npix && c < Cin) {
            const int ox = (int)(p % Wo);
            const int oy = (int)((p / Wo) % Ho);
            const int b = (int)(p / ((int64_t)Wo * Ho));
            const int iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = Tag::load(x + (((int64_t)b * H + iy) * W + ix) * Cin + c);
                if (relu) v = fmaxf(v, 0.f);
            }
        }
        tile[ty + 4 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k;
        const int64_t p = p0 + tx;
        if (c < Cin && p < ld) Tag::store(dst + ((int64_t)tap * Cin + c) * ld + p, tile[tx][ty + 4 * k]);
    }
}

extern "C" int uc_im2col_t(const void* x, void* dst, int dtype, int B, int H, int W, int Cin, int stride, int relu, int64_t ld,
                           uc_stream_t stream) {
    UC_REQUIRE(x && dst && B > 0 && H > 0 && W > 0 && Cin > 0 && stride > 0, "uc_im2col_t: bad argument");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t npix = (int64_t)B * Ho * Wo;
    UC_REQUIRE(ld >= npix && ld < npix + 64, "uc_im2col_t: ld must be in [npix, npix+64)");
    dim3 grid((unsigned)ceil_div64(ld, 64), (unsigned)((Cin + 63) / 64), 9);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_F32) hipLaunchKernelGGL((im2col_t_kernel<F32Tag>), grid, dim3(256), 0, st, (const float*)x, (float*)dst, B, H, W, Cin, stride, Ho, Wo, relu, npix, ld);
    else if (dtype == UC_BF16) hipLaunchKernelGGL((im2col_t_kernel<BF16Tag>), grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)dst, B, H, W, Cin, stride, Ho, Wo, relu, npix, ld);
    else { uc_set_error("uc_im2col_t: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_im2col_t");
    return UC_OK;
}

// =================================================================================================================
// zero-stuffing: dst[b, y, x, :] = src[b, y/s, x/s, :] when y and x are multiples of s (and inside src), else 0
// =================================================================================================================
template <typename Tag>
__global__ void dilate_kernel(const typename Tag::storage* __restrict__ src, typename Tag::storage* __restrict__ dst, int B, int h,
                              int w, int H, int W, int C, int s, int64_t items) {
    const int C8 = C / 8;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = it;
        const int c8 = (int)(r % C8); r /= C8;
        const int X = (int)(r % W); r /= W;
        const int Y = (int)(r % H);
        const int b = (int)(r / H);
        V8b v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = 0.f;
        if (Y % s == 0 && X % s == 0 && Y / s < h && X / s < w)
            v = ld8<Tag>(src + ((((int64_t)b * h + Y / s) * w + X / s) * C) + c8 * 8);
        st8<Tag>(dst + it * 8, v);
    }
}

extern "C" int uc_dilate_nhwc(const void* src, void* dst, int dtype, int B, int h, int w, int H, int W, int C, int stride,
                              uc_stream_t stream) {
    UC_REQUIRE(src && dst && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && stride > 0, "uc_dilate_nhwc: bad argument");
    UC_REQUIRE((h - 1) * stride < H && (w - 1) * stride < W, "uc_dilate_nhwc: source does not fit the dilated grid");
    const int64_t items = (int64_t)B * H * W * (C / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_F32) hipLaunchKernelGGL((dilate_kernel<F32Tag>), dim3(DB_GRID(items)), dim3(256), 0, st, (const float*)src, (float*)dst, B, h, w, H, W, C, stride, items);
    else if (dtype == UC_BF16) hipLaunchKernelGGL((dilate_kernel<BF16Tag>), dim3(DB_GRID(items)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, h, w, H, W, C, stride, items);
    else { uc_set_error("uc_dilate_nhwc: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_dilate_nhwc");
    return UC_OK;
}

// =================================================================================================================
// backward of uc_conv1x1_to4:  out[p,o] = b[o] + sum_c feat[p,c] w[o,c]
//   dfeat[p,c] = sum_o dout[p,o] w[o,c];  dw[o,c] += sum_p dout[p,o] feat[p,c];  db[o] += sum_p dout[p,o]
// thread = (pixel lane, 8-channel chunk); per-thread 4x8 partial dw, reduced over the block's pixel lanes in LDS.
// =================================================================================================================
template <typename Tag>
__global__ __launch_bounds__(256) void conv1x1_to4_bwd_kernel(const typename Tag::storage* __restrict__ feat, const float* __restrict__ w,
                                                               const float* __restrict__ dout, typename Tag::storage* __restrict__ dfeat,
                                                               float* __restrict__ dw, float* __restrict__ db, int64_t npix, int Cin, int relu_mask) {
    __shared__ float red[256][33];
    const int C8 = Cin / 8;
    const int lanes = 256 / C8;              // pixel lanes per block
    const int chunk = threadIdx.x % C8, pl = threadIdx.x / C8;
    const bool live = pl < lanes;
    float wr[4][8], acc[4][8], bacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wr[o][e] = w[o * Cin + chunk * 8 + e];
            acc[o][e] = 0.f;
        }
    if (live) {
        for (int64_t p = (int64_t)blockIdx.x * lanes + pl; p < npix; p += (int64_t)gridDim.x * lanes) {
            const float4_t g = *reinterpret_cast<const float4_t*>(dout + p * 4);
            const float gv[4] = {g.x, g.y, g.z, g.w};
            const V8b f = ld8<Tag>(feat + p * Cin + chunk * 8);
            V8b d;
#pragma unroll
            for (int e = 0; e < 8; ++e) d.v[e] = gv[0] * wr[0][e] + gv[1] * wr[1][e] + gv[2] * wr[2][e] + gv[3] * wr[3][e];
            if (relu_mask) {      // feat is the OUTPUT of a ReLU: its backward rides here (the stand-alone mask pass re-read both maps)
#pragma unroll
                for (int e = 0; e < 8; ++e) d.v[e] = f.v[e] > 0.f ? d.v[e] : 0.f;
            }
            st8<Tag>(dfeat + p * Cin + chunk * 8, d);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(gv[o], f.v[e], acc[o][e]);
                if (chunk == 0) bacc[o] += gv[o];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x][o * 8 + e] = live ? acc[o][e] : 0.f;
    __syncthreads();
    if (threadIdx.x < C8) {   // pixel lane 0 of every chunk folds the other lanes
#pragma unroll 1
        for (int i = 0; i < 32; ++i) {
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[l * C8 + threadIdx.x][i];
            unsafeAtomicAdd(dw + (i >> 3) * Cin + threadIdx.x * 8 + (i & 7), s);
        }
    }
    // bias gradient: chunk-0 threads hold the partial sums
    __syncthreads();
    if (chunk == 0) {
#pragma unroll
        for (int o = 0; o < 4; ++o) red[pl][o] = live ? bacc[o] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[l][threadIdx.x];
        unsafeAtomicAdd(db + threadIdx.x, s);
    }
}

extern "C" int uc_conv1x1_to4_bwd(const void* feat, int dtype, const float* w, const float* dout, void* dfeat, float* dw, float* db,
                                  int64_t npix, int Cin, int relu_mask, uc_stream_t stream) {
    UC_REQUIRE(feat && w && dout && dfeat && dw && db, "uc_conv1x1_to4_bwd: null pointer");
    UC_REQUIRE(npix > 0 && Cin > 0 && Cin <= 256 && Cin % 8 == 0, "uc_conv1x1_to4_bwd: Cin must be a multiple of 8 and <= 256");
    const int lanes = 256 / (Cin / 8);
    const unsigned grid = (unsigned)min((int64_t)2048, ceil_div64(npix, lanes));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UC_F32) hipLaunchKernelGGL((conv1x1_to4_bwd_kernel<F32Tag>), dim3(grid), dim3(256), 0, st, (const float*)feat, w, dout, (float*)dfeat, dw, db, npix, Cin, relu_mask);
    else if (dtype == UC_BF16) hipLaunchKernelGGL((conv1x1_to4_bwd_kernel<BF16Tag>), dim3(grid), dim3(256), 0, st, (const bf16_t*)feat, w, dout, (bf16_t*)dfeat, dw, db, npix, Cin, relu_mask);
    else { uc_set_error("uc_conv1x1_to4_bwd: bad dtype %d", dtype); return UC_ERR_BAD_ARG; }
    UC_CHECK_LAUNCH("uc_conv1x1_to4_bwd");
    return UC_OK;
}
