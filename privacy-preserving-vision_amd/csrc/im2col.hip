// Patch rows for the weight gradient of convolutions with very few input channels (the 3-channel image convolutions of the StarGAN-v2
// blocks, Face-DeId/core/model.py:12-53 autograd): x [B,H,W,C] f32 NHWC -> rows [B*Ho*Wo][Kp] bf16 with column (r * S + s) * C + c =
// x[b, ho * stride - pad + r, wo * stride - pad + s, c] (0 outside the image, 0 in the padding columns K .. Kp-1), in one of the two
// terms of the bf16 split: part 0 = hi = bf16(x), part 1 = mid = bf16(x - hi), part 2 = lo = bf16(x - hi - mid)
// (the third term of the six-product form, nn_ops.conv2d_f32(exact=True)).  The R x S x C gradient then is a 1x1 weight gradient with
// Kp "channels" on the MFMA kernel (ppv_conv_wgrad): the same K-padding idea as the trunk's 7x7 stem (conv_wgrad_stem.hip).
#include <hip/hip_runtime.h>
#include "ppv_common.h"
#include "ppv_hip.h"

namespace {
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ __launch_bounds__(256) void im2col_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, int B, int H, int W,
                                                           int C, int Ho, int Wo, int R, int S, int stride, int pad, int Kp, int part) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)B * Ho * Wo * Kp;
    if (i >= tot) return;
    const int k = (int)(i % Kp);
    const long m = i / Kp;
    float v = 0.f;
    if (k < R * S * C) {
        const int c = k % C, rs = k / C, s = rs % S, r = rs / S;
        const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
        const int h = ho * stride - pad + r, w = wo * stride - pad + s;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = x[(((long)b * H + h) * W + w) * C + c];
    }
    const unsigned short hi = f2bf(v);
    const float r1 = v - bf2f(hi);
    out[i] = part == 0 ? hi : part == 1 ? f2bf(r1) : f2bf(r1 - bf2f(f2bf(r1)));
}

// channel-padded halves of the bf16 split of an NHWC f32 tensor: x [rows][C] -> out [rows][Cp], part as above
__global__ __launch_bounds__(256) void pad_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, long rows, int C, int Cp,
                                                        int part) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Cp) return;
    const int c = (int)(i % Cp);
    const float v = c < C ? x[(i / Cp) * C + c] : 0.f;
    const unsigned short hi = f2bf(v);
    const float r1 = v - bf2f(hi);
    out[i] = part == 0 ? hi : part == 1 ? f2bf(r1) : f2bf(r1 - bf2f(f2bf(r1)));
}
}  // namespace

extern "C" {

int ppv_im2col_split(const float* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int Kp,
                     int part, hipStream_t stream) {
    if (!x || !out) return PPV_ERR_NULL;
    if (B < 1 || C < 1 || R < 1 || S < 1 || Kp < R * S * C || stride < 1 || (part < 0 || part > 2)) return PPV_ERR_BAD_SIZE;
    const long tot = (long)B * Ho * Wo * Kp;
    im2col_split_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(x, (unsigned short*)out, B, H, W, C, Ho, Wo, R, S, stride, pad, Kp, part);
    return ppv_last_error();
}

int ppv_pad_split(const float* x, void* out, long rows, int C, int Cp, int part, hipStream_t stream) {
    if (!x || !out) return PPV_ERR_NULL;
    if (rows < 1 || C < 1 || Cp < C || part < 0 || part > 2) return PPV_ERR_BAD_SIZE;
    pad_split_kernel<<<(unsigned)((rows * Cp + 255) / 256), 256, 0, stream>>>(x, (unsigned short*)out, rows, C, Cp, part);
    return ppv_last_error();
}

}  // extern "C"
