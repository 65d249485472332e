// Element-wise / resampling kernels of the FAN heat-map regressor forward (eval mode), NHWC bf16, gfx950.
//
// Replaces, around the MFMA convolutions of conv_gemm.hip / the 6-channel stem of conv_wgrad_stem.hip, the glue of
// reference Face-DeId/core/wing.py:178-260: bilinear input resize + x*0.5+0.5 + CoordConv channels (:241-244,:78-118),
// avg_pool2d 2x2 (:64,:213), nearest x2 up-sampling + add (:73-75), the channel concatenation + residual of ConvBlock
// (:171-175), and the heat-map head: per-group channel sums, bilinear x4 (align_corners=True), clamp (:246-251).
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

typedef unsigned short bf16_t;
__device__ __forceinline__ bf16_t f2bf_f(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ void load8f(const bf16_t* p, float (&f)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void store8f(bf16_t* p, const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf_f(f[2 * i]) | ((unsigned)f2bf_f(f[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void store8f(float* p, const float (&f)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

// out [B,6,S,S] f32: channels 0..2 = bilinear(x [B,3,Hin,Win], align_corners=False) * 0.5 + 0.5, 3..5 = coords [3,S,S]
__global__ __launch_bounds__(256) void fan_input_kernel(const float* __restrict__ x, const float* __restrict__ coords,
                                                        float* __restrict__ out, int B, int Hin, int Win, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)B * 6 * S * S;
    if (i >= tot) return;
    const int w = (int)(i % S), h = (int)((i / S) % S), c = (int)((i / ((long)S * S)) % 6), b = (int)(i / ((long)S * S * 6));
    if (c >= 3) { out[i] = coords[((long)(c - 3) * S + h) * S + w]; return; }
    const float sh = (float)Hin / (float)S, sw = (float)Win / (float)S;
    const float fy = fmaxf(((float)h + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf(((float)w + 0.5f) * sw - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = x + ((long)b * 3 + c) * Hin * Win;
    const float v = (1.f - ly) * ((1.f - lx) * p[(long)y0 * Win + x0] + lx * p[(long)y0 * Win + x1]) +
                    ly * ((1.f - lx) * p[(long)y1 * Win + x0] + lx * p[(long)y1 * Win + x1]);
    out[i] = v * 0.5f + 0.5f;
}

// [B,H,W,C] bf16 -> [B,H/2,W/2,C] bf16
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W,
                                                            int C) {
    const int Ho = H / 2, Wo = W / 2, c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * Ho * Wo * c8n) return;
    const int c0 = (int)(i % c8n) * 8, wo = (int)((i / c8n) % Wo), ho = (int)((i / ((long)c8n * Wo)) % Ho);
    const int b = (int)(i / ((long)c8n * Wo * Ho));
    float a[8], acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            load8f(x + (((long)b * H + 2 * ho + dy) * W + 2 * wo + dx) * C + c0, a);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += a[k];
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= 0.25f;
    store8f(y + i * 8, acc);
}

// out [B,H,W,C] = up1 + nearest_x2(low [B,H/2,W/2,C])
template <typename T>
__global__ __launch_bounds__(256) void upsample2_add_kernel(const T* __restrict__ up1, const T* __restrict__ low,
                                                            T* __restrict__ out, int B, int H, int W, int C) {
    const int c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * c8n) return;
    const int c0 = (int)(i % c8n) * 8, w = (int)((i / c8n) % W), h = (int)((i / ((long)c8n * W)) % H);
    const int b = (int)(i / ((long)c8n * W * H));
    float a[8], l[8];
    load8f(up1 + i * 8, a);
    load8f(low + (((long)b * (H / 2) + h / 2) * (W / 2) + w / 2) * C + c0, l);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += l[k];
    store8f(out + i * 8, a);
}

// out [M][n1+n2+n3] = cat(o1[M][:n1] (stride s1), o2[M][:n2] (stride s2), o3[M][:n3] (stride s3)) + res [M][n1+n2+n3]
template <typename T>
__global__ __launch_bounds__(256) void concat3_add_kernel(const T* __restrict__ o1, const T* __restrict__ o2,
                                                          const T* __restrict__ o3, const T* __restrict__ res,
                                                          T* __restrict__ out, long M, int n1, int n2, int n3, int s1,
                                                          int s2, int s3) {
    const int C = n1 + n2 + n3, c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * c8n) return;
    const int c0 = (int)(i % c8n) * 8;
    const long m = i / c8n;
    const T* src = (c0 < n1) ? o1 + m * s1 + c0 : (c0 < n1 + n2) ? o2 + m * s2 + (c0 - n1) : o3 + m * s3 + (c0 - n1 - n2);
    float a[8], r[8];
    load8f(src, a);
    load8f(res + i * 8, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += r[k];
    store8f(out + i * 8, a);
}

// l0 raw [M][ldr] f32 accumulators (+ bias [nch]) -> raw_out [B][nch][HW] f32 (optional) and sums [B][2][HW] f32 over channels
// [0, split) and [split, nsum)
__global__ __launch_bounds__(256) void fan_head_sum_kernel(const float* __restrict__ raw, const float* __restrict__ bias,
                                                           float* __restrict__ raw_out, float* __restrict__ sums, long M, int HW,
                                                           int ldr, int nch, int split, int nsum) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const long b = m / HW, px = m % HW;
    float s0 = 0.f, s1 = 0.f;
    for (int c = 0; c < nch; ++c) {
        const float v = raw[m * ldr + c] + bias[c];
        if (raw_out) raw_out[(b * nch + c) * HW + px] = v;
        if (c < split) s0 += v;
        else if (c < nsum) s1 += v;
    }
    sums[(b * 2 + 0) * HW + px] = s0;
    sums[(b * 2 + 1) * HW + px] = s1;
}

// in [n][S][S] f32 -> out [n][S*f][S*f] f32, bilinear align_corners=True, clamp [0,1]
__global__ __launch_bounds__(256) void bilinear_up_clamp_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int S,
                                                                int f) {
    const int So = S * f;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * So * So) return;
    const int w = (int)(i % So), h = (int)((i / So) % So);
    const long m = i / ((long)So * So);
    const float sc = (float)(S - 1) / (float)(So - 1);
    const float fy = (float)h * sc, fx = (float)w * sc;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, S - 1), x1 = min(x0 + 1, S - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = in + m * S * S;
    const float v = (1.f - ly) * ((1.f - lx) * p[y0 * S + x0] + lx * p[y0 * S + x1]) + ly * ((1.f - lx) * p[y1 * S + x0] + lx * p[y1 * S + x1]);
    out[i] = fminf(fmaxf(v, 0.f), 1.f);
}

// ----------------------------------------------------------------------------- fp32-accurate convolutions on the bf16 MFMA path
// y = act(x * scale + shift) in f32, written as THREE bf16 channel groups [hi | lo | hi] (hi = bf16(y), lo = bf16(y - hi)), zero
// padded to Cp channels.  A convolution over that tensor with the weight groups [W_hi | W_hi | W_lo] accumulates
// y_hi W_hi + y_lo W_hi + y_hi W_lo in the MFMA's f32 accumulators: the product error drops from 2^-9 to about 2^-16
// relative, which is what the 1e-3 parity bar needs through FAN's ~100 layers (the reference runs the regressor in fp32).
// C % 8 == 0: one thread = 8 channels of one row (two float4 loads, three 16-byte stores), pad chunks zero-filled
__global__ __launch_bounds__(256) void bn_act_split3_vec_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                                bf16_t* __restrict__ y, long rows, int C, int Cp, int relu, int ldx) {
    const int cpr = C / 8 + (Cp - 3 * C) / 8;                 // work items per row: data chunks, then pad chunks
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cpr) return;
    const long row = i / cpr;
    const int k = (int)(i % cpr);
    bf16_t* yr = y + row * Cp;
    if (k >= C / 8) {
        *reinterpret_cast<uint4*>(yr + 3 * C + (k - C / 8) * 8) = make_uint4(0, 0, 0, 0);
        return;
    }
    const int c0 = k * 8;
    const float4 u = *reinterpret_cast<const float4*>(x + row * ldx + c0), v = *reinterpret_cast<const float4*>(x + row * ldx + c0 + 4);
    float t[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w}, hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (coef) t[j] = t[j] * coef[c0 + j] + coef[C + c0 + j];
        if (relu) t[j] = fmaxf(t[j], 0.f);
        hi[j] = __builtin_bit_cast(float, (unsigned)f2bf_f(t[j]) << 16);
        lo[j] = t[j] - hi[j];
    }
    store8f(yr + c0, hi);
    store8f(yr + C + c0, lo);
    store8f(yr + 2 * C + c0, hi);
}

__global__ __launch_bounds__(256) void bn_act_split3_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                            bf16_t* __restrict__ y, long rows, int C, int Cp, int relu, int ldx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Cp) return;
    const long row = i / Cp;
    const int c = (int)(i % Cp);
    float v = 0.f;
    const int grp = c / C, cc = c % C;
    if (c < 3 * C) {
        float t = x[row * ldx + cc];
        if (coef) t = t * coef[cc] + coef[C + cc];
        if (relu) t = fmaxf(t, 0.f);
        const float hi = __builtin_bit_cast(float, (unsigned)f2bf_f(t) << 16);
        v = grp == 1 ? t - hi : hi;
    }
    y[i] = f2bf_f(v);
}

// one thread per 8 output columns of a source row: writes that chunk of the three stacked copies
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long rows, int C,
                                                          int Cp, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = Cp / 8;
    if (i >= rows * cpr) return;
    const long r = i / cpr;
    const int c0 = (int)(i % cpr) * 8;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + 2 * k + e;
            v[e] = c < C ? x[r * ldx + c] : 0.f;
        }
        const unsigned h0 = __builtin_bit_cast(unsigned short, (__bf16)v[0]), h1 = __builtin_bit_cast(unsigned short, (__bf16)v[1]);
        const float f0 = __builtin_bit_cast(float, h0 << 16), f1 = __builtin_bit_cast(float, h1 << 16);
        const unsigned l0 = __builtin_bit_cast(unsigned short, (__bf16)(v[0] - f0)), l1 = __builtin_bit_cast(unsigned short, (__bf16)(v[1] - f1));
        hi[k] = h0 | (h1 << 16);
        lo[k] = l0 | (l1 << 16);
    }
    const uint4 H = make_uint4(hi[0], hi[1], hi[2], hi[3]), Lw = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    uint4* o = reinterpret_cast<uint4*>(y + r * Cp + c0);
    const long plane = rows * (long)Cp / 8;                     // uint4 elements per stacked copy
    o[0] = H;
    o[plane] = mode == 0 ? Lw : H;
    o[2 * plane] = mode == 0 ? H : Lw;
}

}  // namespace ppv

using namespace ppv;

// ---- bilinear resize of NCHW f32 planes with autograd (round 6: FAN.get_heatmap_train's two F.interpolate calls, wing.py:264,270).
// torch's arithmetic (aten UpSample.h area_pixel_compute_source_index): align_corners ? scale * dst, scale = (in - 1) / (out - 1)
//                                                                                    : max(scale * (dst + 0.5) - 0.5, 0), scale = in / out
__device__ __forceinline__ void bilin_src(int dst, float scale, int align, int in, int& i0, int& i1, float& l1) {
    float s = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = min((int)s, in - 1);
    i1 = min(i0 + 1, in - 1);
    l1 = s - (float)i0;
}
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int Hi, int Wi,
                                                           int Ho, int Wo, float sh, float sw, int align) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
    const long pl = i / ((long)Wo * Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(oy, sh, align, Hi, y0, y1, ly);
    bilin_src(ox, sw, align, Wi, x0, x1, lx);
    const float* p = x + pl * Hi * Wi;
    const float hy = 1.f - ly, hx = 1.f - lx;
    y[i] = hy * (hx * p[(long)y0 * Wi + x0] + lx * p[(long)y0 * Wi + x1]) + ly * (hx * p[(long)y1 * Wi + x0] + lx * p[(long)y1 * Wi + x1]);
}
// adjoint as a GATHER (deterministic, no atomics): input pixel (iy, ix) collects every output pixel whose two taps per axis include it;
// the candidate output range per axis is bracketed from the inverse map (+-2) and each candidate re-derives its own taps
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, long planes, int Hi, int Wi,
                                                           int Ho, int Wo, float sh, float sw, int align) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * Hi * Wi) return;
    const int ix = (int)(i % Wi), iy = (int)((i / Wi) % Hi);
    const long pl = i / ((long)Wi * Hi);
    auto range = [&](int idx, float scale, int n_out, int& lo, int& hi) {
        // outputs with source position in (idx - 1, idx + 1): dst in ((idx - 1) / scale, (idx + 1) / scale) up to the half-pixel shift
        const float inv = scale > 0.f ? 1.f / scale : (float)n_out;
        lo = max(0, (int)floorf(((float)idx - 1.f) * inv) - 2);
        hi = min(n_out - 1, (int)ceilf(((float)idx + 1.f) * inv) + 2);
        if (scale <= 0.f) { lo = 0; hi = n_out - 1; }
    };
    int oy_lo, oy_hi, ox_lo, ox_hi;
    range(iy, sh, Ho, oy_lo, oy_hi);
    range(ix, sw, Wo, ox_lo, ox_hi);
    const float* g = gy + pl * Ho * Wo;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly;
        bilin_src(oy, sh, align, Hi, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            int x0, x1;
            float lx;
            bilin_src(ox, sw, align, Wi, x0, x1, lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            if (wx != 0.f) row += wx * g[(long)oy * Wo + ox];
        }
        acc += wy * row;
    }
    gx[i] = acc;
}

extern "C" {

static void bilin_scales(int Hi, int Wi, int Ho, int Wo, int align, float* sh, float* sw) {
    *sh = align ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f) : (float)Hi / (float)Ho;
    *sw = align ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f) : (float)Wi / (float)Wo;
}
int ppv_bilinear_resize_fwd(const float* x, float* y, long planes, int Hi, int Wi, int Ho, int Wo, int align_corners, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (planes < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return PPV_ERR_BAD_SIZE;
    float sh, sw;
    bilin_scales(Hi, Wi, Ho, Wo, align_corners, &sh, &sw);
    const long tot = planes * Ho * Wo;
    bilinear_fwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(x, y, planes, Hi, Wi, Ho, Wo, sh, sw, align_corners ? 1 : 0);
    return ppv_last_error();
}
int ppv_bilinear_resize_bwd(const float* gy, float* gx, long planes, int Hi, int Wi, int Ho, int Wo, int align_corners, hipStream_t stream) {
    if (!gy || !gx) return PPV_ERR_NULL;
    if (planes < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return PPV_ERR_BAD_SIZE;
    float sh, sw;
    bilin_scales(Hi, Wi, Ho, Wo, align_corners, &sh, &sw);
    const long tot = planes * Hi * Wi;
    bilinear_bwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(gy, gx, planes, Hi, Wi, Ho, Wo, sh, sw, align_corners ? 1 : 0);
    return ppv_last_error();
}

int ppv_fan_input(const float* x, const float* coords, float* out, int B, int Hin, int Win, int S, hipStream_t stream) {
    if (!x || !coords || !out) return PPV_ERR_NULL;
    const long tot = (long)B * 6 * S * S;
    fan_input_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(x, coords, out, B, Hin, Win, S);
    return ppv_last_error();
}

// f32 = 0: bf16 tensors, 1: f32 tensors (fp32-accurate FAN mode)
int ppv_avgpool2_nhwc(const void* x, void* y, int B, int H, int W, int C, int f32, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (H % 2 || W % 2 || C % 8) return PPV_ERR_BAD_SIZE;
    const long tot = (long)B * (H / 2) * (W / 2) * (C / 8);
    if (f32) avgpool2_nhwc_kernel<float><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const float*)x, (float*)y, B, H, W, C);
    else avgpool2_nhwc_kernel<bf16_t><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)y, B, H, W, C);
    return ppv_last_error();
}

int ppv_upsample2_add(const void* up1, const void* low, void* out, int B, int H, int W, int C, int f32, hipStream_t stream) {
    if (!up1 || !low || !out) return PPV_ERR_NULL;
    if (H % 2 || W % 2 || C % 8) return PPV_ERR_BAD_SIZE;
    const long tot = (long)B * H * W * (C / 8);
    if (f32) upsample2_add_kernel<float><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const float*)up1, (const float*)low, (float*)out, B, H, W, C);
    else upsample2_add_kernel<bf16_t><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const bf16_t*)up1, (const bf16_t*)low, (bf16_t*)out, B, H, W, C);
    return ppv_last_error();
}

int ppv_concat3_add(const void* o1, const void* o2, const void* o3, const void* res, void* out, long M, int n1, int n2, int n3,
                    int s1, int s2, int s3, int f32, hipStream_t stream) {
    if (!o1 || !o2 || !o3 || !res || !out) return PPV_ERR_NULL;
    if (n1 % 8 || n2 % 8 || n3 % 8) return PPV_ERR_BAD_SIZE;
    const long tot = M * ((n1 + n2 + n3) / 8);
    if (f32) concat3_add_kernel<float><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const float*)o1, (const float*)o2, (const float*)o3,
                                                                                         (const float*)res, (float*)out, M, n1, n2, n3, s1, s2, s3);
    else concat3_add_kernel<bf16_t><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const bf16_t*)o1, (const bf16_t*)o2, (const bf16_t*)o3,
                                                                                      (const bf16_t*)res, (bf16_t*)out, M, n1, n2, n3, s1, s2, s3);
    return ppv_last_error();
}

int ppv_fan_head(const void* raw, const float* bias, float* raw_out, float* sums, float* heat, int B, int S, int ldr, int nch,
                 int split, int nsum, int up, hipStream_t stream) {
    if (!raw || !bias || !sums || !heat) return PPV_ERR_NULL;
    const long M = (long)B * S * S;
    fan_head_sum_kernel<<<(unsigned)((M + 255) / 256), 256, 0, stream>>>((const float*)raw, bias, raw_out, sums, M, S * S, ldr, nch, split, nsum);
    const long tot = (long)B * 2 * S * up * S * up;
    bilinear_up_clamp_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(sums, heat, (long)B * 2, S, up);
    return ppv_last_error();
}

// x [rows][ldx] f32 (first C columns used) -> y [3 * rows][Cp] bf16, the bf16 split stacked along the ROWS (the reduction axis of a
// transposed product g^T h): mode 0 = [hi; lo; hi], mode 1 = [hi; hi; lo], columns >= C zero.  g^T h ~ y0(g)^T y1(h) with f32
// accumulation = g_hi^T h_hi + g_lo^T h_hi + g_hi^T h_lo (product error ~2^-16): feeds ppv_conv_wgrad with the decoder's batched
// weight gradients (Image_Caption/models.py:199-214 under autograd).
int ppv_split3_rows(const float* x, long ldx, void* y, long rows, int C, int Cp, int mode, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (rows < 1 || C < 1 || Cp < C || Cp % 8 || ldx < C || mode < 0 || mode > 1) return PPV_ERR_BAD_SIZE;
    const long items = rows * (Cp / 8);
    ppv::split3_rows_kernel<<<(unsigned)((items + 255) / 256), 256, 0, stream>>>(x, ldx, (ppv::bf16_t*)y, rows, C, Cp, mode);
    return ppv_last_error();
}

// x [rows][ldx] f32 (first C columns used) -> y [rows][Cp] bf16 = [hi | lo | hi | 0...] of act(x * coef[0][c] + coef[1][c]) (coef may be null = identity);
// Cp >= 3 * C (zero padded).  See bn_act_split3_kernel: feeds ppv_conv_gemm with [W_hi | W_hi | W_lo] weights.
int ppv_bn_act_split3(const float* x, const float* coef, void* y, long rows, int C, int Cp, int relu, int ldx, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (rows < 1 || C < 1 || Cp < 3 * C || ldx < C) return PPV_ERR_BAD_SIZE;
    const long tot = rows * Cp;
    if (C % 8 == 0 && Cp % 8 == 0 && ldx % 4 == 0) {
        const long items = rows * (C / 8 + (Cp - 3 * C) / 8);
        ppv::bn_act_split3_vec_kernel<<<(unsigned)((items + 255) / 256), 256, 0, stream>>>(x, coef, (ppv::bf16_t*)y, rows, C, Cp, relu, ldx);
    } else {
        ppv::bn_act_split3_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(x, coef, (ppv::bf16_t*)y, rows, C, Cp, relu, ldx);
    }
    return ppv_last_error();
}

}  // extern "C"
