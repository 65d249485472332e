// SSIM loss (Image_Caption/pytorch_ssim/__init__.py:20-40, used as camera_loss = 'SSIM', train.py:172-173):
// five 11x11 Gaussian-window moments (zero padding) of two images, the SSIM map and its mean -- fused into one kernel per
// direction.  The window is separable (outer product of one normalised 1-D Gaussian, sigma 1.5), so a 32x32 output tile is
// two 11-tap passes over an LDS halo tile instead of five 121-tap depthwise convolutions; nothing but the images is read and
// nothing but the per-image sums (forward) or the image gradient (backward) is written.
//   forward : sums[b] += sum over the planes of image b of the SSIM map            (f64 atomics, one per workgroup)
//   backward: d/d img2 for upstream weight gs[b] per map pixel of image b:
//             dL/dy_p = conv(a)_p + 2 y_p conv(b)_p + x_p conv(c)_p,  a = dS/dmu2, b = dS/dE[yy], c = dS/dE[xy]  (times gs)
//             (the map is symmetric in its arguments: d/d img1 is the same kernel with the images swapped)
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

struct SsimWin { float w[11]; };
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

// grid (ceil(W/32), ceil(H/32), B*C), 256 threads
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, double* __restrict__ sums,
                                                       int C, int H, int W, SsimWin win) {
    __shared__ float sx[42][43], sy[42][43];
    __shared__ float h5[5][42][32];
    __shared__ float s4[4];
    const int tid = threadIdx.x, tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
    const long plane = (long)blockIdx.z * H * W;
    for (int i = tid; i < 42 * 42; i += 256) {
        const int r = i / 42, c = i % 42, gy = ty0 - 5 + r, gx = tx0 - 5 + c;
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        sx[r][c] = ok ? x[plane + (long)gy * W + gx] : 0.f;
        sy[r][c] = ok ? y[plane + (long)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 42 * 32; i += 256) {
        const int r = i / 32, c = i % 32;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float u = sx[r][c + k], v = sy[r][c + k], wk = win.w[k];
            a += wk * u; b += wk * v; aa += wk * u * u; bb += wk * v * v; ab += wk * u * v;
        }
        h5[0][r][c] = a; h5[1][r][c] = b; h5[2][r][c] = aa; h5[3][r][c] = bb; h5[4][r][c] = ab;
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = tid; i < 32 * 32; i += 256) {
        const int r = i / 32, c = i % 32;
        if (ty0 + r >= H || tx0 + c >= W) continue;
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float wk = win.w[k];
            m1 += wk * h5[0][r + k][c]; m2 += wk * h5[1][r + k][c]; s11 += wk * h5[2][r + k][c];
            s22 += wk * h5[3][r + k][c]; s12 += wk * h5[4][r + k][c];
        }
        const float v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cv = s12 - m1 * m2;
        acc += ((2.f * m1 * m2 + SSIM_C1) * (2.f * cv + SSIM_C2)) / ((m1 * m1 + m2 * m2 + SSIM_C1) * (v1 + v2 + SSIM_C2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) s4[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) atomicAdd(&sums[blockIdx.z / C], (double)(s4[0] + s4[1] + s4[2] + s4[3]));
}

// grid as above; dynamic LDS (see ppv_ssim_bwd): d/d y
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ gs, float* __restrict__ dy, int C, int H, int W,
                                                       SsimWin win) {
    extern __shared__ float sm[];
    float (*sx)[53] = reinterpret_cast<float (*)[53]>(sm);                       // [52][53]
    float (*sy)[53] = reinterpret_cast<float (*)[53]>(sm + 52 * 53);             // [52][53]
    float (*h5)[52][42] = reinterpret_cast<float (*)[52][42]>(sm + 2 * 52 * 53); // [5][52][42]
    float (*abc)[42][43] = reinterpret_cast<float (*)[42][43]>(sm + 2 * 52 * 53 + 5 * 52 * 42);          // [3][42][43]
    float (*hab)[42][32] = reinterpret_cast<float (*)[42][32]>(sm + 2 * 52 * 53 + 5 * 52 * 42 + 3 * 42 * 43);   // [3][42][32]
    const int tid = threadIdx.x, tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
    const long plane = (long)blockIdx.z * H * W;
    const float g = gs[blockIdx.z / C];
    for (int i = tid; i < 52 * 52; i += 256) {
        const int r = i / 52, c = i % 52, gy = ty0 - 10 + r, gx = tx0 - 10 + c;
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        sx[r][c] = ok ? x[plane + (long)gy * W + gx] : 0.f;
        sy[r][c] = ok ? y[plane + (long)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 52 * 42; i += 256) {
        const int r = i / 42, c = i % 42;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float u = sx[r][c + k], v = sy[r][c + k], wk = win.w[k];
            a += wk * u; b += wk * v; aa += wk * u * u; bb += wk * v * v; ab += wk * u * v;
        }
        h5[0][r][c] = a; h5[1][r][c] = b; h5[2][r][c] = aa; h5[3][r][c] = bb; h5[4][r][c] = ab;
    }
    __syncthreads();
    for (int i = tid; i < 42 * 42; i += 256) {                 // SSIM-map pixels (ty0 - 5 + r, tx0 - 5 + c)
        const int r = i / 42, c = i % 42, py = ty0 - 5 + r, px = tx0 - 5 + c;
        float da = 0.f, db = 0.f, dc = 0.f;
        if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) {
            float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float wk = win.w[k];
                m1 += wk * h5[0][r + k][c]; m2 += wk * h5[1][r + k][c]; s11 += wk * h5[2][r + k][c];
                s22 += wk * h5[3][r + k][c]; s12 += wk * h5[4][r + k][c];
            }
            const float v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cv = s12 - m1 * m2;
            const float A1 = 2.f * m1 * m2 + SSIM_C1, A2 = 2.f * cv + SSIM_C2, B1 = m1 * m1 + m2 * m2 + SSIM_C1, B2 = v1 + v2 + SSIM_C2;
            const float inv = 1.f / (B1 * B2), S = A1 * A2 * inv;
            da = g * ((2.f * m1 * (A2 - A1)) * inv - S * (2.f * m2 / B1 - 2.f * m2 / B2));
            db = g * (-S / B2);
            dc = g * (2.f * A1 * inv);
        }
        abc[0][r][c] = da; abc[1][r][c] = db; abc[2][r][c] = dc;
    }
    __syncthreads();
    for (int i = tid; i < 42 * 32; i += 256) {
        const int r = i / 32, c = i % 32;
        float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float wk = win.w[k];
            a += wk * abc[0][r][c + k]; b += wk * abc[1][r][c + k]; cc += wk * abc[2][r][c + k];
        }
        hab[0][r][c] = a; hab[1][r][c] = b; hab[2][r][c] = cc;
    }
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += 256) {
        const int r = i / 32, c = i % 32, py = ty0 + r, px = tx0 + c;
        if (py >= H || px >= W) continue;
        float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float wk = win.w[k];
            a += wk * hab[0][r + k][c]; b += wk * hab[1][r + k][c]; cc += wk * hab[2][r + k][c];
        }
        dy[plane + (long)py * W + px] = a + 2.f * sy[r + 10][c + 10] * b + sx[r + 10][c + 10] * cc;
    }
}

}  // namespace ppv

extern "C" {

// sums [B] f64 PRE-ZEROED += per-image sum of the SSIM map of img1, img2 [B][C][H][W] f32; win: the 11 normalised taps
int ppv_ssim_fwd(const float* img1, const float* img2, double* sums, const float* win, int B, int C, int H, int W, hipStream_t stream) {
    if (!img1 || !img2 || !sums || !win) return PPV_ERR_NULL;
    if (B < 1 || C < 1 || H < 1 || W < 1) return PPV_ERR_BAD_SIZE;
    ppv::SsimWin w;
    for (int k = 0; k < 11; ++k) w.w[k] = win[k];
    ppv::ssim_fwd_kernel<<<dim3((W + 31) / 32, (H + 31) / 32, B * C), 256, 0, stream>>>(img1, img2, sums, C, H, W, w);
    return ppv_last_error();
}

// d_img2 [B][C][H][W] = gradient of sum_b gscale[b] * (sum of image b's SSIM map) w.r.t. img2 (swap img1 / img2 for d_img1).
// win is a HOST array of 11 floats; gscale [B] f32 on the device.
int ppv_ssim_bwd(const float* img1, const float* img2, const float* gscale, float* d_img2, const float* win, int B, int C, int H, int W,
                 hipStream_t stream) {
    if (!img1 || !img2 || !gscale || !d_img2 || !win) return PPV_ERR_NULL;
    if (B < 1 || C < 1 || H < 1 || W < 1) return PPV_ERR_BAD_SIZE;
    ppv::SsimWin w;
    for (int k = 0; k < 11; ++k) w.w[k] = win[k];
    constexpr int lds = (2 * 52 * 53 + 5 * 52 * 42 + 3 * 42 * 43 + 3 * 42 * 32) * 4;
    static PpvDevOnce attr_once;
    if (attr_once.need()) { PPV_ATTR(hipFuncSetAttribute((const void*)ppv::ssim_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr_once.done(); }
    ppv::ssim_bwd_kernel<<<dim3((W + 31) / 32, (H + 31) / 32, B * C), 256, lds, stream>>>(img1, img2, gscale, d_img2, C, H, W, w);
    return ppv_last_error();
}

}  // extern "C"
