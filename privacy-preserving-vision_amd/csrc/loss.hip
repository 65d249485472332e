// Camera MSE loss of the training harness, fused (Image_Caption/train.py:284-288: `loss_cam = 1 - criterion_camera(imgs, sensor)` with
// criterion_camera = nn.MSELoss(), train.py:170-171) -- HBM-bound, two launches per step instead of the four torch passes
// (aten::mse_loss elementwise + mean, mse_loss_backward, autograd's add_ of the two gradients that reach `sensor`).
//
//   forward : out[0] = mean((a - b)^2).  One pass over both tensors, float4 loads (four in flight per lane), f32 per-lane sums, f64 from
//             the wave up.  DETERMINISTIC: every workgroup writes its partial, a one-workgroup launch adds them in index order.
//   backward: g_b = g_in + k * (b - a)   with k = coef * gscalar[0]  (gscalar: the DEVICE scalar autograd hands to the loss; no host
//             sync), g_in = the gradient that reached `b` through its other consumer (the encoder) or null.  One pass: 2-3 reads + 1
//             write per element instead of (2 reads + 1 write) + (2 reads + 1 write).  Optional g_a = -k * (b - a).
#include <hip/hip_runtime.h>
#include "ppv_common.h"
#include "ppv_hip.h"

namespace ppv {

constexpr int MSE_NT = 256, MSE_MAX_WG = 8192;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(MSE_NT) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, double* __restrict__ partial) {
    const long n4 = n >> 2, stride = (long)gridDim.x * MSE_NT;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float acc = 0.f;
    long i = (long)blockIdx.x * MSE_NT + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {              // four independent 16-byte loads of each tensor in flight
        float4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = a4[i + u * stride]; y[u] = b4[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d0 = x[u].x - y[u].x, d1 = x[u].y - y[u].y, d2 = x[u].z - y[u].z, d3 = x[u].w - y[u].w;
            acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    for (; i < n4; i += stride) {
        const float4 x = a4[i], y = b4[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {             // tail (n not a multiple of 4)
        const float d = a[(n4 << 2) + threadIdx.x] - b[(n4 << 2) + threadIdx.x];
        acc += d * d;
    }
    __shared__ double s_w[MSE_NT / 64];
    const double w = wave_sum_f64((double)acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < MSE_NT / 64; ++k) t += s_w[k];
        partial[blockIdx.x] = t;
    }
}

// The partials in index order (one workgroup).  A launch of its own: the first version let the last workgroup to arrive do this behind a
// device-scope fence + ticket, and 2048 workgroups each executing `__threadfence()` (an L2 write-back + invalidate on this multi-L2 part)
// while the others were still streaming took the pass to 130 us for 201 MB (1.5 TB/s; the backward pass, no fences, moves 400 MB in 65 us).
__global__ __launch_bounds__(MSE_NT) void mse_final_kernel(const double* __restrict__ partial, int np, float* __restrict__ out, double inv_n) {
    double t = 0.0;
    for (int k = threadIdx.x; k < np; k += MSE_NT) t += partial[k];            // fixed order per lane
    t = wave_sum_f64(t);
    __shared__ double s_w[MSE_NT / 64];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < MSE_NT / 64; ++k) s += s_w[k];
        out[0] = (float)(s * inv_n);
    }
}

template <bool HAS_IN, bool WANT_A>
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ g_in, const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gscalar, float coef, float* __restrict__ g_b,
                                                      float* __restrict__ g_a, long n) {
    const float k = coef * gscalar[0];
    const long n4 = n >> 2, stride = (long)gridDim.x * 256;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    const float4* i4 = reinterpret_cast<const float4*>(g_in);
    float4* o4 = reinterpret_cast<float4*>(g_b);
    float4* oa4 = reinterpret_cast<float4*>(g_a);
    auto one = [&](float4 x, float4 y, float4 gi, long at) {
        float4 d = {k * (y.x - x.x), k * (y.y - x.y), k * (y.z - x.z), k * (y.w - x.w)};
        if (WANT_A) oa4[at] = float4{-d.x, -d.y, -d.z, -d.w};
        if (HAS_IN) { d.x += gi.x; d.y += gi.y; d.z += gi.z; d.w += gi.w; }
        o4[at] = d;
    };
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 x[4], y[4], gi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x[u] = a4[i + u * stride];
            y[u] = b4[i + u * stride];
            if (HAS_IN) gi[u] = i4[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(x[u], y[u], gi[u], i + u * stride);
    }
    for (; i < n4; i += stride) one(a4[i], b4[i], HAS_IN ? i4[i] : float4{0.f, 0.f, 0.f, 0.f}, i);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long j = (n4 << 2) + threadIdx.x;
        const float d = k * (b[j] - a[j]);
        if (WANT_A) g_a[j] = -d;
        g_b[j] = HAS_IN ? g_in[j] + d : d;
    }
}

}  // namespace ppv

extern "C" {

size_t ppv_mse_workspace_bytes(void) { return (size_t)ppv::MSE_MAX_WG * sizeof(double) + 256; }

int ppv_mse_fwd(const float* a, const float* b, long n, void* workspace, float* out, hipStream_t stream) {
    if (!a || !b || !workspace || !out) return PPV_ERR_NULL;
    if (n < 1 || ((size_t)a % 16) || ((size_t)b % 16) || ((size_t)workspace % 16)) return PPV_ERR_BAD_SIZE;
    long wg = ((n >> 2) + ppv::MSE_NT * 4 - 1) / (ppv::MSE_NT * 4);              // >= four float4 per lane where the tensor is large enough
    wg = wg < 1 ? 1 : (wg > ppv::MSE_MAX_WG ? ppv::MSE_MAX_WG : wg);
    double* partial = (double*)workspace;
    ppv::mse_fwd_kernel<<<(unsigned)wg, ppv::MSE_NT, 0, stream>>>(a, b, n, partial);
    ppv::mse_final_kernel<<<1, ppv::MSE_NT, 0, stream>>>(partial, (int)wg, out, 1.0 / (double)n);
    return ppv_last_error();
}

int ppv_mse_bwd(const float* g_in, const float* a, const float* b, const float* gscalar, float coef, float* g_b, float* g_a, long n,
                hipStream_t stream) {
    if (!a || !b || !gscalar || !g_b) return PPV_ERR_NULL;
    if (n < 1 || ((size_t)a % 16) || ((size_t)b % 16) || ((size_t)g_b % 16) || (g_in && ((size_t)g_in % 16)) || (g_a && ((size_t)g_a % 16)))
        return PPV_ERR_BAD_SIZE;
    long wg = ((n >> 2) + 256 * 4 - 1) / (256 * 4);
    wg = wg < 1 ? 1 : (wg > 8192 ? 8192 : wg);
    const unsigned g = (unsigned)wg;
    if (g_in && g_a) ppv::mse_bwd_kernel<true, true><<<g, 256, 0, stream>>>(g_in, a, b, gscalar, coef, g_b, g_a, n);
    else if (g_in) ppv::mse_bwd_kernel<true, false><<<g, 256, 0, stream>>>(g_in, a, b, gscalar, coef, g_b, g_a, n);
    else if (g_a) ppv::mse_bwd_kernel<false, true><<<g, 256, 0, stream>>>(g_in, a, b, gscalar, coef, g_b, g_a, n);
    else ppv::mse_bwd_kernel<false, false><<<g, 256, 0, stream>>>(g_in, a, b, gscalar, coef, g_b, g_a, n);
    return ppv_last_error();
}

}  // extern "C"
