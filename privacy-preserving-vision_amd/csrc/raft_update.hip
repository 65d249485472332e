// Gate arithmetic of RAFT's convolutional GRUs (reference Face-DeId/RAFT/core/update.py:33-77 SepConvGRU, :16-31 ConvGRU), gfx950.
// The six convolutions run on the MFMA implicit-GEMM kernel (ppv_conv_gemm_rect, fp32-accurate three-term bf16 split as in
// ppv_amd.fan); these two element-wise kernels sit between them, NHWC f32:
//   gru_zr : zr [M][2 Ch] = conv_{z|r}(hx) without bias  ->  z = sigmoid(. + b_z),  rh = sigmoid(. + b_r) * h
//   gru_out: q [M][Ch] = conv_q([r h, x]) without bias   ->  h' = (1 - z) h + z tanh(q + b_q)
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

__global__ __launch_bounds__(256) void gru_zr_kernel(const float* __restrict__ zr, int ldzr, const float* __restrict__ bias,
                                                     const float* __restrict__ h, float* __restrict__ z, float* __restrict__ rh,
                                                     long rows, int Ch) {
    const int c4n = Ch / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const float4 vz = *reinterpret_cast<const float4*>(zr + row * ldzr + c), vr = *reinterpret_cast<const float4*>(zr + row * ldzr + Ch + c);
    const float4 bz = *reinterpret_cast<const float4*>(bias + c), br = *reinterpret_cast<const float4*>(bias + Ch + c);
    const float4 hv = *reinterpret_cast<const float4*>(h + row * Ch + c);
    *reinterpret_cast<float4*>(z + row * Ch + c) =
        make_float4(sigmoid_f(vz.x + bz.x), sigmoid_f(vz.y + bz.y), sigmoid_f(vz.z + bz.z), sigmoid_f(vz.w + bz.w));
    *reinterpret_cast<float4*>(rh + row * Ch + c) = make_float4(sigmoid_f(vr.x + br.x) * hv.x, sigmoid_f(vr.y + br.y) * hv.y,
                                                                sigmoid_f(vr.z + br.z) * hv.z, sigmoid_f(vr.w + br.w) * hv.w);
}

__global__ __launch_bounds__(256) void gru_out_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ bias,
                                                      const float* __restrict__ z, const float* __restrict__ h,
                                                      float* __restrict__ hn, long rows, int Ch) {
    const int c4n = Ch / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const float4 vq = *reinterpret_cast<const float4*>(q + row * ldq + c), bq = *reinterpret_cast<const float4*>(bias + c);
    const float4 zv = *reinterpret_cast<const float4*>(z + row * Ch + c), hv = *reinterpret_cast<const float4*>(h + row * Ch + c);
    auto upd = [](float zz, float hh, float qq) { return (1.0f - zz) * hh + zz * tanhf(qq); };
    *reinterpret_cast<float4*>(hn + row * Ch + c) = make_float4(upd(zv.x, hv.x, vq.x + bq.x), upd(zv.y, hv.y, vq.y + bq.y),
                                                                upd(zv.z, hv.z, vq.z + bq.z), upd(zv.w, hv.w, vq.w + bq.w));
}

}  // namespace ppv

using namespace ppv;

extern "C" {

// zr [rows][ldzr] f32: columns 0..Ch-1 = z pre-activation, Ch..2Ch-1 = r pre-activation (conv outputs, no bias); bias [2 Ch];
// h [rows][Ch] f32 -> z [rows][Ch], rh [rows][Ch].  Ch % 4 == 0.
int ppv_gru_zr(const float* zr, int ldzr, const float* bias, const float* h, float* z, float* rh, long rows, int Ch, hipStream_t stream) {
    if (!zr || !bias || !h || !z || !rh) return PPV_ERR_NULL;
    if (Ch % 4 || ldzr < 2 * Ch || ldzr % 4 || rows < 1) return PPV_ERR_BAD_SIZE;
    gru_zr_kernel<<<(unsigned)((rows * (Ch / 4) + 255) / 256), 256, 0, stream>>>(zr, ldzr, bias, h, z, rh, rows, Ch);
    return ppv_last_error();
}

// q [rows][ldq] f32 (conv output, no bias), bias [Ch], z, h [rows][Ch] -> hn = (1 - z) h + z tanh(q + bias)
int ppv_gru_out(const float* q, int ldq, const float* bias, const float* z, const float* h, float* hn, long rows, int Ch, hipStream_t stream) {
    if (!q || !bias || !z || !h || !hn) return PPV_ERR_NULL;
    if (Ch % 4 || ldq < Ch || ldq % 4 || rows < 1) return PPV_ERR_BAD_SIZE;
    gru_out_kernel<<<(unsigned)((rows * (Ch / 4) + 255) / 256), 256, 0, stream>>>(q, ldq, bias, z, h, hn, rows, Ch);
    return ppv_last_error();
}

}  // extern "C"
