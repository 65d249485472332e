// The element-wise side of the fp32 trunk (Encoder(precision="fp32"), round 6): train-mode BatchNorm2d (+ residual) (+ ReLU), the stem's
// max-pool and the adaptive average pool on NHWC float32 activations, forward and backward, gfx950.  Reference: the torchvision ResNet-101
// behind Image_Caption/models.py:17-41 trained in fp32 (train.py:245: batch statistics, biased variance for the normalisation, unbiased
// for the running estimate, eps inside the square root; eval mode: the running statistics).  All HBM-bound, float4 accesses, and
// DETERMINISTIC: statistics are per-row-block partial sums in f64, added in block order by the finalize kernels (no atomics).
//   forward : bnf32_partial (sum x, sum x^2 per channel and row block) -> bnf32_finalize (coef [4][C] = scale, shift, mean, invstd;
//             running statistics) -> bnf32_apply (y = act(x * scale + shift + res))
//   backward: bnf32_bwd_partial (sum gt, sum gt * xhat; gt = g * (y > 0) behind a ReLU) -> bnf32_bwd_finalize (d beta, d gamma) ->
//             bnf32_bwd_apply (g_x = scale * (gt - d beta / n - xhat * d gamma / n); the residual branch receives gt)
#include <hip/hip_runtime.h>
#include "ppv_common.h"
#include "ppv_hip.h"

namespace ppv {

typedef unsigned short bf16_t_;
constexpr int BF_CB = 64;          // channels per workgroup (16 float4 lanes)
constexpr int BF_MAXBLK = 256;     // row blocks (partials per channel)

__host__ __device__ inline int bf_rows_per_block(long rows) {
    long r = (rows + BF_MAXBLK - 1) / BF_MAXBLK;
    r = (r + 15) / 16 * 16;
    return (int)(r < 16 ? 16 : r);
}

// partial [nblk][2][C] f64.  grid (nblk, C / 64), 256 threads = 16 row lanes x 16 channel quads
template <bool BWD, bool RELU>
__global__ __launch_bounds__(256) void bnf32_partial_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ coef, double* __restrict__ partial, long rows, int C,
                                                            int rpb) {
    __shared__ double s_a[16][BF_CB + 1], s_b[16][BF_CB + 1];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c0 = blockIdx.y * BF_CB + cq * 4;
    const long r0 = (long)blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    // forward: sums of d = x - K with K = row 0 of the channel (a value within a few standard deviations of the mean): the variance comes
    // out of E[d^2] - E[d]^2 without the cancellation of E[x^2] - E[x]^2 when |mean| >> std (a 16-sample layer-4 BatchNorm behind a ReLU)
    float4 mean = *reinterpret_cast<const float4*>(x + c0), inv = {1.f, 1.f, 1.f, 1.f};
    if (BWD) {
        mean = *reinterpret_cast<const float4*>(coef + 2 * C + c0);
        inv = *reinterpret_cast<const float4*>(coef + 3 * C + c0);
    }
    const float kk[4] = {mean.x, mean.y, mean.z, mean.w};
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (long rb = r0 + rl; rb < r1; rb += 16 * 8) {          // eight rows per lane in f32, then into the f64 running sums
        float fa[4] = {0.f, 0.f, 0.f, 0.f}, fb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long r = rb + (long)u * 16;
            if (r < r1) {
                const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c0);
                const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
                if (!BWD) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float d = xx[k] - kk[k]; fa[k] += d; fb[k] = __builtin_fmaf(d, d, fb[k]); }
                } else {
                    const float4 gv = *reinterpret_cast<const float4*>(g + r * C + c0);
                    float gg[4] = {gv.x, gv.y, gv.z, gv.w};
                    if (RELU) {
                        const float4 yv = *reinterpret_cast<const float4*>(y + r * C + c0);
                        const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) gg[k] = yy[k] > 0.f ? gg[k] : 0.f;
                    }
                    const float mm[4] = {mean.x, mean.y, mean.z, mean.w}, ii[4] = {inv.x, inv.y, inv.z, inv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { fa[k] += gg[k]; fb[k] = __builtin_fmaf(gg[k], (xx[k] - mm[k]) * ii[k], fb[k]); }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] += (double)fa[k]; b[k] += (double)fb[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_a[rl][cq * 4 + k] = a[k]; s_b[rl][cq * 4 + k] = b[k]; }
    __syncthreads();
    if (threadIdx.x < 2 * BF_CB) {
        const int which = threadIdx.x / BF_CB, c = threadIdx.x % BF_CB;
        double t = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += which ? s_b[i][c] : s_a[i][c];          // fixed order
        partial[((long)blockIdx.x * 2 + which) * C + blockIdx.y * BF_CB + c] = t;
    }
}

// totals of nblk partial rows in block order: thread (c, grp) adds blocks grp, grp + 4, ...; the four groups meet in LDS in fixed order
__device__ __forceinline__ void bf_totals(const double* __restrict__ partial, int nblk, int C, int c, double& s, double& q, double (*sm)[2][BF_CB]) {
    const int cl = threadIdx.x & (BF_CB - 1), grp = threadIdx.x >> 6;
    double a = 0, b = 0;
    for (int t = grp; t < nblk; t += 4) { a += partial[((long)t * 2) * C + c]; b += partial[((long)t * 2 + 1) * C + c]; }
    sm[grp][0][cl] = a; sm[grp][1][cl] = b;
    __syncthreads();
    s = sm[0][0][cl] + sm[1][0][cl] + sm[2][0][cl] + sm[3][0][cl];
    q = sm[0][1][cl] + sm[1][1][cl] + sm[2][1][cl] + sm[3][1][cl];
}

// grid C / 64, 256 threads.  train: coefficients from the batch; !train: from the running statistics (partial unused)
__global__ __launch_bounds__(256) void bnf32_finalize_kernel(const float* __restrict__ x, const double* __restrict__ partial, int nblk, double count, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                             float momentum, float eps, float* __restrict__ coef, int C, int train) {
    __shared__ double sm[4][2][BF_CB];
    const int c = blockIdx.x * BF_CB + (threadIdx.x & (BF_CB - 1));
    double mean, var;
    if (train) {
        double s, q;
        bf_totals(partial, nblk, C, c, s, q, sm);
        const double md = s / count;                              // mean of d = x - K
        var = q / count - md * md;
        if (var < 0) var = 0;
        mean = (double)x[c] + md;                                 // K = row 0
    } else {
        mean = (double)run_mean[c];
        var = (double)run_var[c];
    }
    if (threadIdx.x >= BF_CB) return;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    coef[c] = sc;
    coef[C + c] = beta[c] - (float)mean * sc;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = invstd;
    if (train && run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        const double unb = count > 1 ? var * count / (count - 1.0) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// sums [2][C] = (d beta, d gamma); also into dgamma / dbeta (may be null)
__global__ __launch_bounds__(256) void bnf32_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ sums,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
    __shared__ double sm[4][2][BF_CB];
    const int c = blockIdx.x * BF_CB + (threadIdx.x & (BF_CB - 1));
    double s, q;
    bf_totals(partial, nblk, C, c, s, q, sm);
    if (threadIdx.x >= BF_CB) return;
    sums[c] = (float)s;
    sums[C + c] = (float)q;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bnf32_apply_kernel(const float* __restrict__ x, const float* __restrict__ coef, const float* __restrict__ res,
                                                          float* __restrict__ y, long n4, int C) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i * 4) % C);
        const float4 xv = *reinterpret_cast<const float4*>(x + i * 4);
        const float4 sc = *reinterpret_cast<const float4*>(coef + c), sh = *reinterpret_cast<const float4*>(coef + C + c);
        float4 o = {__builtin_fmaf(xv.x, sc.x, sh.x), __builtin_fmaf(xv.y, sc.y, sh.y), __builtin_fmaf(xv.z, sc.z, sh.z), __builtin_fmaf(xv.w, sc.w, sh.w)};
        if (RES) {
            const float4 r = *reinterpret_cast<const float4*>(res + i * 4);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(y + i * 4) = o;
    }
}

// inv_n = 1 / rows (train) or 0 (eval: running statistics are constants)
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bnf32_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ x,
                                                              const float* __restrict__ coef, const float* __restrict__ sums, float inv_n,
                                                              float* __restrict__ gx, float* __restrict__ gres, long n4, int C) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i * 4) % C);
        const float4 gv = *reinterpret_cast<const float4*>(g + i * 4), xv = *reinterpret_cast<const float4*>(x + i * 4);
        float gg[4] = {gv.x, gv.y, gv.z, gv.w};
        if (RELU) {
            const float4 yv = *reinterpret_cast<const float4*>(y + i * 4);
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) gg[k] = yy[k] > 0.f ? gg[k] : 0.f;
        }
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xx[k] - coef[2 * C + c + k]) * coef[3 * C + c + k];
            o[k] = coef[c + k] * (gg[k] - sums[c + k] * inv_n - xh * (sums[C + c + k] * inv_n));
        }
        *reinterpret_cast<float4*>(gx + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
        if (RES) *reinterpret_cast<float4*>(gres + i * 4) = make_float4(gg[0], gg[1], gg[2], gg[3]);
    }
}

// ---- max-pool 3 x 3, stride 2, padding 1 (resnet.3), NHWC f32.  arg [B,Ho,Wo,C] u8: window offset (3 r + s) of the FIRST maximum in
// row-major scan order (torch.nn.MaxPool2d's choice: the gradient goes there)
__global__ __launch_bounds__(256) void maxpool_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ arg,
                                                              int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c4n = C / 4;
    const long n = (long)B * Ho * Wo * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cq = (int)(i % c4n);
        const long p = i / c4n;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned a[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int h = 2 * ho - 1 + r, w = 2 * wo - 1 + s;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * C + cq * 4);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (vv[k] > m[k] || vv[k] != vv[k]) { m[k] = vv[k]; a[k] = 3 * r + s; }
                }
            }
        *reinterpret_cast<float4*>(y + p * C + cq * 4) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uchar4*>(arg + p * C + cq * 4) = make_uchar4((unsigned char)a[0], (unsigned char)a[1], (unsigned char)a[2], (unsigned char)a[3]);
    }
}

// gather form (deterministic): input pixel (h, w) collects the outputs whose window covers it and whose arg points at it
__global__ __launch_bounds__(256) void maxpool_f32_bwd_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ arg,
                                                              float* __restrict__ gx, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c4n = C / 4;
    const long n = (long)B * H * W * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cq = (int)(i % c4n);
        const long p = i / c4n;
        const int w = (int)(p % W), h = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = h / 2; ho <= (h + 1) / 2; ++ho)               // 2 ho - 1 <= h <= 2 ho + 1
            for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
                if (ho >= Ho || wo >= Wo) continue;
                const unsigned me = 3 * (h - (2 * ho - 1)) + (w - (2 * wo - 1));
                const long q = (((long)b * Ho + ho) * Wo + wo) * C + cq * 4;
                const uchar4 a = *reinterpret_cast<const uchar4*>(arg + q);
                const float4 g = *reinterpret_cast<const float4*>(gy + q);
                if (a.x == me) o[0] += g.x;
                if (a.y == me) o[1] += g.y;
                if (a.z == me) o[2] += g.z;
                if (a.w == me) o[3] += g.w;
            }
        *reinterpret_cast<float4*>(gx + p * C + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- AdaptiveAvgPool2d((E, E)) (models.py:27,39), NHWC f32: window of output i = [floor(i h / E), ceil((i + 1) h / E))
__global__ __launch_bounds__(256) void adaptive_pool_f32_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int E) {
    const int c4n = C / 4;
    const long n = (long)B * E * E * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cq = (int)(i % c4n);
        const long p = i / c4n;
        const int ox = (int)(p % E), oy = (int)((p / E) % E), b = (int)(p / ((long)E * E));
        const int y0 = (oy * H) / E, y1 = ((oy + 1) * H + E - 1) / E, x0 = (ox * W) / E, x1 = ((ox + 1) * W + E - 1) / E;
        float4 s = {0.f, 0.f, 0.f, 0.f};
        for (int h = y0; h < y1; ++h)
            for (int w = x0; w < x1; ++w) {
                const float4 v = *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * C + cq * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        const float k = 1.f / (float)((y1 - y0) * (x1 - x0));
        *reinterpret_cast<float4*>(y + p * C + cq * 4) = make_float4(s.x * k, s.y * k, s.z * k, s.w * k);
    }
}

__global__ __launch_bounds__(256) void adaptive_pool_f32_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B, int H, int W, int C, int E) {
    const int c4n = C / 4;
    const long n = (long)B * H * W * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cq = (int)(i % c4n);
        const long p = i / c4n;
        const int w = (int)(p % W), h = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        // outputs whose window contains h: floor(o H / E) <= h < ceil((o + 1) H / E)
        const int oy0 = (h * E) / H, ox0 = (w * E) / W;
        float4 s = {0.f, 0.f, 0.f, 0.f};
        for (int oy = max(oy0 - 1, 0); oy < E && (oy * H) / E <= h; ++oy) {
            const int y0 = (oy * H) / E, y1 = ((oy + 1) * H + E - 1) / E;
            if (h < y0 || h >= y1) continue;
            for (int ox = max(ox0 - 1, 0); ox < E && (ox * W) / E <= w; ++ox) {
                const int x0 = (ox * W) / E, x1 = ((ox + 1) * W + E - 1) / E;
                if (w < x0 || w >= x1) continue;
                const float k = 1.f / (float)((y1 - y0) * (x1 - x0));
                const float4 g = *reinterpret_cast<const float4*>(gy + (((long)b * E + oy) * E + ox) * C + cq * 4);
                s.x += g.x * k; s.y += g.y * k; s.z += g.z * k; s.w += g.w * k;
            }
        }
        *reinterpret_cast<float4*>(gx + p * C + cq * 4) = s;
    }
}

// ---- operand of the f32-level convolution: x [rows][C] f32 -> y [rows][Cp] bf16 = [h | m | h | l | h | m | 0 ...] of the three-way split
// x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)); against the K-concatenated filter [H | H | M | H | L | M] ONE MFMA
// convolution then accumulates the six products that matter (h H, m H, h M, l H, h L, m M: ~2^-24 of the product, f32 level).
// One thread per four source channels (or four padding columns).  C % 4 == 0, Cp % 4 == 0, Cp >= 6 C.
__global__ __launch_bounds__(256) void split6_kernel(const float* __restrict__ x, bf16_t_* __restrict__ y, long rows, int C, int Cp) {
    const int per_row = C / 4 + (Cp - 6 * C) / 4;
    const long n = rows * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / per_row;
        const int j = (int)(i % per_row);
        bf16_t_* yr = y + r * Cp;
        if (j >= C / 4) {
            *reinterpret_cast<uint2*>(yr + 6 * C + (j - C / 4) * 4) = make_uint2(0u, 0u);
            continue;
        }
        const float4 v = *reinterpret_cast<const float4*>(x + r * C + j * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], m[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = __builtin_bit_cast(unsigned short, (__bf16)vv[k]);
            const float r1 = vv[k] - __builtin_bit_cast(float, (unsigned)h[k] << 16);
            m[k] = __builtin_bit_cast(unsigned short, (__bf16)r1);
            const float r2 = r1 - __builtin_bit_cast(float, (unsigned)m[k] << 16);
            l[k] = __builtin_bit_cast(unsigned short, (__bf16)r2);
        }
        const uint2 ph = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        const uint2 pm = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
        const uint2 pl = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
        bf16_t_* o = yr + j * 4;
        *reinterpret_cast<uint2*>(o) = ph;
        *reinterpret_cast<uint2*>(o + C) = pm;
        *reinterpret_cast<uint2*>(o + 2 * C) = ph;
        *reinterpret_cast<uint2*>(o + 3 * C) = pl;
        *reinterpret_cast<uint2*>(o + 4 * C) = ph;
        *reinterpret_cast<uint2*>(o + 5 * C) = pm;
    }
}

static inline unsigned bf_grid(long n) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace ppv

using namespace ppv;

extern "C" {

size_t ppv_bn_f32_workspace_bytes(int C) { return (size_t)BF_MAXBLK * 2 * (size_t)C * sizeof(double); }

int ppv_bn_f32_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                   const float* res, float* y, float* coef, void* workspace, long rows, int C, int relu, int train, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || !coef || !workspace) return PPV_ERR_NULL;
    if (rows < 1 || C < BF_CB || C % BF_CB) return PPV_ERR_BAD_SIZE;
    if (!train && (!run_mean || !run_var)) return PPV_ERR_NULL;
    if (run_mean && !run_var) return PPV_ERR_NULL;
    const int rpb = bf_rows_per_block(rows), nblk = (int)((rows + rpb - 1) / rpb);
    if (train) bnf32_partial_kernel<false, false><<<dim3(nblk, C / BF_CB), 256, 0, stream>>>(x, nullptr, nullptr, nullptr, (double*)workspace, rows, C, rpb);
    bnf32_finalize_kernel<<<C / BF_CB, 256, 0, stream>>>(x, (const double*)workspace, nblk, (double)rows, gamma, beta, run_mean, run_var, momentum, eps, coef, C, train ? 1 : 0);
    const long n4 = rows * C / 4;
    const unsigned g = bf_grid(n4);
    if (res && relu) bnf32_apply_kernel<true, true><<<g, 256, 0, stream>>>(x, coef, res, y, n4, C);
    else if (res) bnf32_apply_kernel<true, false><<<g, 256, 0, stream>>>(x, coef, res, y, n4, C);
    else if (relu) bnf32_apply_kernel<false, true><<<g, 256, 0, stream>>>(x, coef, res, y, n4, C);
    else bnf32_apply_kernel<false, false><<<g, 256, 0, stream>>>(x, coef, res, y, n4, C);
    return ppv_last_error();
}

int ppv_bn_f32_bwd(const float* g, const float* y, const float* x, const float* coef, float* gx, float* gres, float* dgamma, float* dbeta,
                   float* sums, void* workspace, long rows, int C, int relu, int train, hipStream_t stream) {
    if (!g || !x || !coef || !gx || !sums || !workspace || (relu && !y)) return PPV_ERR_NULL;
    if (rows < 1 || C < BF_CB || C % BF_CB) return PPV_ERR_BAD_SIZE;
    const int rpb = bf_rows_per_block(rows), nblk = (int)((rows + rpb - 1) / rpb);
    if (relu) bnf32_partial_kernel<true, true><<<dim3(nblk, C / BF_CB), 256, 0, stream>>>(x, g, y, coef, (double*)workspace, rows, C, rpb);
    else bnf32_partial_kernel<true, false><<<dim3(nblk, C / BF_CB), 256, 0, stream>>>(x, g, y, coef, (double*)workspace, rows, C, rpb);
    bnf32_bwd_finalize_kernel<<<C / BF_CB, 256, 0, stream>>>((const double*)workspace, nblk, sums, dgamma, dbeta, C);
    const long n4 = rows * C / 4;
    const unsigned gr = bf_grid(n4);
    const float inv_n = train ? (float)(1.0 / (double)rows) : 0.f;
    if (gres && relu) bnf32_bwd_apply_kernel<true, true><<<gr, 256, 0, stream>>>(g, y, x, coef, sums, inv_n, gx, gres, n4, C);
    else if (gres) bnf32_bwd_apply_kernel<true, false><<<gr, 256, 0, stream>>>(g, y, x, coef, sums, inv_n, gx, gres, n4, C);
    else if (relu) bnf32_bwd_apply_kernel<false, true><<<gr, 256, 0, stream>>>(g, y, x, coef, sums, inv_n, gx, gres, n4, C);
    else bnf32_bwd_apply_kernel<false, false><<<gr, 256, 0, stream>>>(g, y, x, coef, sums, inv_n, gx, gres, n4, C);
    return ppv_last_error();
}

int ppv_split6_rows(const float* x, void* y, long rows, int C, int Cp, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (rows < 1 || C < 4 || C % 4 || Cp % 4 || Cp < 6 * C) return PPV_ERR_BAD_SIZE;
    split6_kernel<<<bf_grid(rows * (C / 4 + (Cp - 6 * C) / 4)), 256, 0, stream>>>(x, (bf16_t_*)y, rows, C, Cp);
    return ppv_last_error();
}

int ppv_maxpool_f32_fwd(const float* x, float* y, void* arg, int B, int H, int W, int C, hipStream_t stream) {
    if (!x || !y || !arg) return PPV_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || C % 4) return PPV_ERR_BAD_SIZE;
    maxpool_f32_fwd_kernel<<<bf_grid((long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4)), 256, 0, stream>>>(x, y, (unsigned char*)arg, B, H, W, C);
    return ppv_last_error();
}

int ppv_maxpool_f32_bwd(const float* gy, const void* arg, float* gx, int B, int H, int W, int C, hipStream_t stream) {
    if (!gy || !arg || !gx) return PPV_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || C % 4) return PPV_ERR_BAD_SIZE;
    maxpool_f32_bwd_kernel<<<bf_grid((long)B * H * W * (C / 4)), 256, 0, stream>>>(gy, (const unsigned char*)arg, gx, B, H, W, C);
    return ppv_last_error();
}

int ppv_adaptive_pool_f32_fwd(const float* x, float* y, int B, int H, int W, int C, int E, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || E < 1 || C % 4) return PPV_ERR_BAD_SIZE;
    adaptive_pool_f32_fwd_kernel<<<bf_grid((long)B * E * E * (C / 4)), 256, 0, stream>>>(x, y, B, H, W, C, E);
    return ppv_last_error();
}

int ppv_adaptive_pool_f32_bwd(const float* gy, float* gx, int B, int H, int W, int C, int E, hipStream_t stream) {
    if (!gy || !gx) return PPV_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || E < 1 || C % 4) return PPV_ERR_BAD_SIZE;
    adaptive_pool_f32_bwd_kernel<<<bf_grid((long)B * H * W * (C / 4)), 256, 0, stream>>>(gy, gx, B, H, W, C, E);
    return ppv_last_error();
}

}  // extern "C"
