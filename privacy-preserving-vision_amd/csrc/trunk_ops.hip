// Element-wise / reduction kernels of the ResNet-101 trunk around the MFMA convolutions, NHWC bf16, gfx950.
//
// Train-mode BatchNorm2d + ReLU (+ residual) forward and backward (batch statistics in ALL 104 BNs,
// eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased for the running estimate), the stem's fused
// BN + ReLU + MaxPool 3x3/2, and the Encoder's AdaptiveAvgPool2d(36) on an 8x8 map (models.py:27,39-40; output is
// already NHWC so the reference's permute disappears).  All HBM-bound: 16-byte accesses, fp32 math.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>
#include "ppv_common.h"
#include "bn_coef.h"

namespace ppv {

typedef unsigned short bf16_t;
__device__ __forceinline__ bf16_t f2bf_(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f_(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

struct bf8 { bf16_t v[8]; };
__device__ __forceinline__ void load8(const bf16_t* p, float (&f)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void unpack8_(const uint4 u, float (&f)[8]) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void unpack8u(const uint4 u, float (&f)[8]) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
#ifndef PPV_NT_ELT
#define PPV_NT_ELT 0       // 1: element-wise outputs (activations, gradients) stored non-temporally (A/B build; measured neutral to -0.1 ms)
#endif
__device__ __forceinline__ void store8(bf16_t* p, const float (&f)[8]) {
    typedef unsigned nt_u32x4_t __attribute__((ext_vector_type(4)));
    nt_u32x4_t w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf_(f[2 * i]) | ((unsigned)f2bf_(f[2 * i + 1]) << 16);
#if PPV_NT_ELT
    __builtin_nontemporal_store(w, reinterpret_cast<nt_u32x4_t*>(p));
#else
    *reinterpret_cast<nt_u32x4_t*>(p) = w;
#endif
}

// ----------------------------------------------------------------------------- BN statistics -> coefficients
// part [T][2][C] (sum, sumsq of the bf16 conv output) -> coef [4][C] = scale, shift, mean, invstd; running stats updated.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int T, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           float momentum, float eps, float* __restrict__ coef, int C) {
    __shared__ double s_s[16][64], s_q[16][64];
    const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0, q = 0;
    for (int t = tg; t < T; t += 16) {
        s += (double)part[((long)t * 2 + 0) * C + c];
        q += (double)part[((long)t * 2 + 1) * C + c];
    }
    s_s[tg][cl] = s; s_q[tg][cl] = q;
    __syncthreads();
    if (tg == 0) {
#pragma unroll
        for (int i = 1; i < 16; ++i) { s += s_s[i][cl]; q += s_q[i][cl]; }
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        coef[c] = sc;
        coef[C + c] = beta[c] - (float)mean * sc;
        coef[2 * C + c] = (float)mean;
        coef[3 * C + c] = invstd;
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
            const double unb = count > 1 ? var * count / (count - 1.0) : var;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
        }
    }
}

// ----------------------------------------------------------------------------- BN apply (+ residual) (+ ReLU)
// y = act(x*s1 + t1 + res),  res = 0 | r | r*s2 + t2
template <int RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_kernel(const bf16_t* __restrict__ x, const float* __restrict__ coef1,
                                                     const bf16_t* __restrict__ r, const float* __restrict__ coef2,
                                                     bf16_t* __restrict__ y, unsigned char* __restrict__ pos_bits, long n8, int C,
                                                     long res_mod8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float xv[8], rv[8], o[8];
    load8(x + i * 8, xv);
    if (RES) load8(r + (res_mod8 ? i % res_mod8 : i) * 8, rv);      // res_mod8 > 0: residual broadcast over the batch
    const float4 sa = *reinterpret_cast<const float4*>(coef1 + c0), sb = *reinterpret_cast<const float4*>(coef1 + c0 + 4);
    const float4 ta = *reinterpret_cast<const float4*>(coef1 + C + c0), tb = *reinterpret_cast<const float4*>(coef1 + C + c0 + 4);
    const float s1[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
    const float t1[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = xv[k] * s1[k] + t1[k];
        if (RES == 1) v += rv[k];
        if (RES == 2) v += rv[k] * coef2[c0 + k] + coef2[C + c0 + k];
        o[k] = RELU ? fmaxf(v, 0.f) : v;
    }
    store8(y + i * 8, o);
    if (pos_bits) {                                   // bit k = (y[8 i + k] > 0): the ReLU mask at 1/16 of the tensor's bytes
        unsigned b = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) b |= (o[k] > 0.f ? 1u : 0u) << k;
        pos_bits[i] = (unsigned char)b;
    }
}

// ----------------------------------------------------------------------------- BN apply with the coefficients derived per thread (round 3)
// The convolution leaves its statistics in ONE row (stat_rows = 1: every tile adds its column sums to the same [2][C] floats), so the
// element-wise apply kernel can derive scale / shift for ITS eight channels from 16 floats + gamma / beta in a few instructions and
// the 4.7-us bn_finalize launch between every convolution and its apply pass (104 per step, pure dispatch latency) disappears.  The
// threads of the first row also write coef [4][C] (scale, shift, mean, invstd: what the backward pass reads) and the running
// statistics.  Totals in f32 as bn_finalize reads them, mean / variance in double, f32 rsq + one Newton step (<= 1 ulp of the f64 one).
// (ppv_bn_act_train, round 2, folded 32 partial rows per workgroup in a looped kernel and was slower than the two launches.)
// MEASURED in the step (encoder.py PPV_BN_FOLD_ACT=1): 1.3 % SLOWER than bn_finalize + bn_act (5217 vs 5285 images/s): kept opt-in.
template <int RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_fold_kernel(const bf16_t* __restrict__ x, const float* __restrict__ sums, double inv_count,
                                                          double unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                          float eps, float* __restrict__ coef, const bf16_t* __restrict__ r,
                                                          bf16_t* __restrict__ y, unsigned char* __restrict__ pos_bits, long n8, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float xv[8], rv[8], o[8];
    load8(x + i * 8, xv);
    if (RES) load8(r + i * 8, rv);
    const float4 sa = *reinterpret_cast<const float4*>(sums + c0), sb = *reinterpret_cast<const float4*>(sums + c0 + 4);
    const float4 qa = *reinterpret_cast<const float4*>(sums + C + c0), qb = *reinterpret_cast<const float4*>(sums + C + c0 + 4);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c0), gb = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    const float4 ba = *reinterpret_cast<const float4*>(beta + c0), bb = *reinterpret_cast<const float4*>(beta + c0 + 4);
    const float s_[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w}, q_[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const float g_[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w}, b_[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
    const bool writer = i * 8 < C;                     // the first row's threads publish the coefficients of their channels
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double mean = (double)s_[k] * inv_count;
        double var = (double)q_[k] * inv_count - mean * mean;
        if (var < 0) var = 0;
        const float ve = (float)(var + (double)eps);
        float invstd = __builtin_amdgcn_rsqf(ve);
        invstd = invstd * (1.5f - 0.5f * ve * invstd * invstd);
        const float sc = g_[k] * invstd, sh = b_[k] - (float)mean * sc;
        if (writer) {
            const int c = c0 + k;
            coef[c] = sc; coef[C + c] = sh; coef[2 * C + c] = (float)mean; coef[3 * C + c] = invstd;
            if (run_mean) {
                run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
                run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * unbias);
            }
        }
        float v = xv[k] * sc + sh;
        if (RES == 1) v += rv[k];
        o[k] = RELU ? fmaxf(v, 0.f) : v;
        bits |= (o[k] > 0.f ? 1u : 0u) << k;
    }
    store8(y + i * 8, o);
    if (pos_bits) pos_bits[i] = (unsigned char)bits;
}

// Same contract, coefficients shared by the workgroup (round 3, second form): a workgroup covers U * rpp whole rows (rpp = 256 / (C / 8)
// rows per pass).  Its 256 threads first request their U rows, then thread j derives scale / shift of channels j, j + 256, ... (CPT =
// C / 256 of them, one for C <= 256) from the T partial rows -- one channel's few scalars per thread instead of eight channels on a
// subset of the threads: 40 VGPRs, eight waves per SIMD like the plain kernel (the first form of this kernel kept eight channels'
// partial rows in registers: 197 VGPRs, two waves per SIMD, +4..7 us per launch).  Coefficients meet in LDS.
// C in {64, 128, ..., 2048} (C / 8 a power of two <= 256), T <= 2 * ... any (loop).
template <int RES, bool RELU, int U, int CPT>
__global__ __launch_bounds__(256) void bn_act_fold_wg_kernel(const bf16_t* __restrict__ x, const float* __restrict__ sums, double inv_count,
                                                             double unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                             float eps, float* __restrict__ coef, const bf16_t* __restrict__ r,
                                                             bf16_t* __restrict__ y, unsigned char* __restrict__ pos_bits, long rows, int C,
                                                             int tpr_log2, int T) {
    __shared__ float s_sc[256 * CPT], s_sh[256 * CPT];
    const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
    const int tc = threadIdx.x & (tpr - 1), tr = threadIdx.x >> tpr_log2;
    const int c0 = tc * 8;
    const long row0 = (long)blockIdx.x * (rpp * U) + tr;
    uint4 xr[U], rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long row = row0 + (long)u * rpp;
        if (row < rows) {
            xr[u] = *reinterpret_cast<const uint4*>(x + row * C + c0);
            if (RES) rr[u] = *reinterpret_cast<const uint4*>(r + row * C + c0);
        }
    }
    // statistics of this thread's channels: all loads first (they return right behind the rows), then the arithmetic
    constexpr int TR = 4;                              // partial rows held in registers (more rows: a second, dependent round of loads)
    float ps[CPT][TR], pq[CPT][TR], pg[CPT], pb[CPT];
    // unconditional loads from clamped indices, a scheduling fence, THEN the masks: written as `cond ? load : 0` every load sat in its own
    // branch with its own wait -- 2 T serial round trips in front of every workgroup's first store (round 5)
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int cc = min((int)threadIdx.x + j * 256, C - 1);
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const long o = ((long)min(t, T - 1) * 2) * C + cc;
            ps[j][t] = sums[o];
            pq[j][t] = sums[o + C];
        }
        pg[j] = gamma[cc];
        pb[j] = beta[cc];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const bool ok = (int)threadIdx.x + j * 256 < C;
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            ps[j][t] = (ok && t < T) ? ps[j][t] : 0.f;
            pq[j][t] = (ok && t < T) ? pq[j][t] : 0.f;
        }
        pg[j] = ok ? pg[j] : 0.f;
        pb[j] = ok ? pb[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = threadIdx.x + j * 256;
        double s_ = 0, q_ = 0;
#pragma unroll
        for (int t = 0; t < TR; ++t) { s_ += (double)ps[j][t]; q_ += (double)pq[j][t]; }
        for (int t = TR; t < T; ++t) { if (c < C) { s_ += (double)sums[((long)t * 2) * C + c]; q_ += (double)sums[((long)t * 2 + 1) * C + c]; } }
        const BnCoef k = bn_coef_pinned(s_, q_, inv_count, unbias, pg[j], pb[j], eps);      // bn_coef.h: the BNIN halo kernel derives the same bits
        const float sc = k.sc, sh = k.sh;
        if (c < C) {
            s_sc[c] = sc; s_sh[c] = sh;
            if (blockIdx.x == 0) {
                coef[c] = sc; coef[C + c] = sh; coef[2 * C + c] = k.mean; coef[3 * C + c] = k.invstd;
                if (run_mean) {
                    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * k.mean;
                    run_var[c] = (1.f - momentum) * run_var[c] + momentum * k.var_unbiased;
                }
            }
        }
    }
    __syncthreads();
    const float4 c_0 = *reinterpret_cast<const float4*>(s_sc + c0), c_1 = *reinterpret_cast<const float4*>(s_sc + c0 + 4);
    const float4 c_2 = *reinterpret_cast<const float4*>(s_sh + c0), c_3 = *reinterpret_cast<const float4*>(s_sh + c0 + 4);
    const float sc[8] = {c_0.x, c_0.y, c_0.z, c_0.w, c_1.x, c_1.y, c_1.z, c_1.w};
    const float sh[8] = {c_2.x, c_2.y, c_2.z, c_2.w, c_3.x, c_3.y, c_3.z, c_3.w};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long row = row0 + (long)u * rpp;
        if (row >= rows) break;
        const long e = row * C + c0;
        float xv[8], rv[8], o[8];
        unpack8_(xr[u], xv);
        if (RES) unpack8_(rr[u], rv);
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = __builtin_fmaf(xv[k], sc[k], sh[k]);          // pinned (bn_coef.h bn_relu_pinned): the BNIN halo kernel applies the same
            if (RES == 1) v += rv[k];
            o[k] = RELU ? fmaxf(v, 0.f) : v;
            bits |= (o[k] > 0.f ? 1u : 0u) << k;
        }
        store8(y + e, o);
        if (pos_bits) pos_bits[e >> 3] = (unsigned char)bits;
    }
}

// ----------------------------------------------------------------------------- train-mode BN: statistics -> coefficients -> apply, one launch
// bn_finalize + bn_act in one kernel (round 2): the coefficient launch was 4.8 us of dispatch latency per BatchNorm for ~1 us of
// work (104 launches per step, tools/micro/graph_chain.py: a dependent tiny kernel costs 4.2 us).  A workgroup owns a block of
// CB = min(C, 256) channels and walks rows rb, rb + RB, ...; its prologue folds the [T][2][C] partial sums of ITS channels (thread
// (tc, tr) takes rows tr, tr + rpp, ... of its eight channels, the partials meet in LDS, totals in double as bn_finalize_kernel) --
// 16 KB of L2-resident reads against >= 64 KB of payload.  Row block 0 also writes coef [4][C] (for the backward pass) and the
// running statistics.  RES == 2: the residual has its own BatchNorm (projection shortcut): both are folded here.
struct BnFold {
    const float* part; int T; double count;
    const float* gamma; const float* beta; float* run_mean; float* run_var; float momentum; float eps; float* coef;
};

__device__ __forceinline__ void bn_fold8(const BnFold& f, int C, int c0, int tc, int tr, int rpp, int tpr, bool writer, float* s_red,
                                         float (&sc)[8], float (&sh)[8]) {
    float ps[8], pq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ps[k] = pq[k] = 0.f;
    for (int t = tr; t < f.T; t += rpp) {
        const float4 a0 = *reinterpret_cast<const float4*>(f.part + ((long)t * 2 + 0) * C + c0), a1 = *reinterpret_cast<const float4*>(f.part + ((long)t * 2 + 0) * C + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(f.part + ((long)t * 2 + 1) * C + c0), b1 = *reinterpret_cast<const float4*>(f.part + ((long)t * 2 + 1) * C + c0 + 4);
        ps[0] += a0.x; ps[1] += a0.y; ps[2] += a0.z; ps[3] += a0.w; ps[4] += a1.x; ps[5] += a1.y; ps[6] += a1.z; ps[7] += a1.w;
        pq[0] += b0.x; pq[1] += b0.y; pq[2] += b0.z; pq[3] += b0.w; pq[4] += b1.x; pq[5] += b1.y; pq[6] += b1.z; pq[7] += b1.w;
    }
    __syncthreads();                                  // s_red may still be read by a previous fold
    float* mine = s_red + ((tr * tpr + tc) * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) { mine[k] = ps[k]; mine[8 + k] = pq[k]; }
    __syncthreads();
    double s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.0;
    for (int j = 0; j < rpp; ++j) {
        const float* o = s_red + ((j * tpr + tc) * 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] += (double)o[k]; q[k] += (double)o[8 + k]; }
    }
    // every workgroup repeats this for its channels: no f64 division / square root here (hundreds of instructions each) -- products
    // with 1 / count in double, then the f32 reciprocal square root refined by one Newton step (<= 1 ulp of the rounded f64 value)
    const double inv_n = 1.0 / f.count;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        const double mean = s[k] * inv_n;
        double var = q[k] * inv_n - mean * mean;
        if (var < 0) var = 0;
        const float ve = (float)(var + (double)f.eps);
        float invstd = __builtin_amdgcn_rsqf(ve);
        invstd = invstd * (1.5f - 0.5f * ve * invstd * invstd);
        sc[k] = f.gamma[c] * invstd;
        sh[k] = f.beta[c] - (float)mean * sc[k];
        if (writer) {
            f.coef[c] = sc[k];
            f.coef[C + c] = sh[k];
            f.coef[2 * C + c] = (float)mean;
            f.coef[3 * C + c] = invstd;
            if (f.run_mean) {
                f.run_mean[c] = (1.f - f.momentum) * f.run_mean[c] + f.momentum * (float)mean;
                const double unb = f.count > 1 ? var * f.count / (f.count - 1.0) : var;
                f.run_var[c] = (1.f - f.momentum) * f.run_var[c] + f.momentum * (float)unb;
            }
        }
    }
}

template <int RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_train_kernel(const bf16_t* __restrict__ x, BnFold f1, const bf16_t* __restrict__ r, BnFold f2,
                                                           bf16_t* __restrict__ y, unsigned char* __restrict__ pos_bits, long rows, int C,
                                                           int ncb) {
    __shared__ float s_red[256 * 16];
    const int CB = C < 256 ? C : 256;
    const int tpr = CB / 8, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const int cb = blockIdx.x % ncb, rb = blockIdx.x / ncb, RB = gridDim.x / ncb;
    const int c0 = cb * CB + tc * 8;
    float s1[8], t1[8], s2[8], t2[8];
    bn_fold8(f1, C, c0, tc, tr, rpp, tpr, rb == 0 && tr == 0, s_red, s1, t1);
    if (RES == 2) bn_fold8(f2, C, c0, tc, tr, rpp, tpr, rb == 0 && tr == 0, s_red, s2, t2);
    // eight rows in flight per thread (64 KB per CU at two workgroups per CU): few workgroups, because each one re-reads its
    // channels' whole [T][2][CB] partial-sum block (64 KB at T = 32, C = 256) -- 2048 workgroups made that 4x the tensor itself
    constexpr int U = 8;
    const long stride = (long)RB * rpp;
    for (long row0 = (long)rb * rpp + tr; row0 < rows; row0 += U * stride) {
        uint4 xr[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * stride;
            if (row < rows) {
                xr[u] = *reinterpret_cast<const uint4*>(x + row * C + c0);
                if (RES) rr[u] = *reinterpret_cast<const uint4*>(r + row * C + c0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * stride;
            if (row >= rows) break;
            const long e = row * C + c0;
            float xv[8], rv[8], o[8];
            unpack8_(xr[u], xv);
            if (RES) unpack8_(rr[u], rv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = xv[k] * s1[k] + t1[k];
                if (RES == 1) v += rv[k];
                if (RES == 2) v += rv[k] * s2[k] + t2[k];
                o[k] = RELU ? fmaxf(v, 0.f) : v;
            }
            store8(y + e, o);
            if (pos_bits) {
                unsigned b = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) b |= (o[k] > 0.f ? 1u : 0u) << k;
                pos_bits[e >> 3] = (unsigned char)b;
            }
        }
    }
}

// ----------------------------------------------------------------------------- BN backward
// pass 1: per-channel partial sums of g_pre and g_pre * x, g_pre = g_y * (y > 0) [RELU] ; part [32][2][C] pre-zeroed,
// block b adds into row b & fold_mask (31 with the separate coefficient launch, 7 with the fused apply)
// RELU: 0 none, 1 mask from the stored activation y, 2 mask recomputed as (x*scale + shift > 0) from coef (BN + ReLU
// without residual: saves reading y; the expression is the forward kernel's, so the mask is bit-identical)
template <int RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                            const bf16_t* __restrict__ x, const float* __restrict__ coef,
                                                            float* __restrict__ part, long rows, int C, int rows_per_blk,
                                                            int fold_mask) {
    __shared__ float s_red[256][17];
    const int tpr = C / 8;                         // threads per row
    const int rpp = 256 / tpr;                     // rows per pass
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long r0 = (long)blockIdx.x * rows_per_blk;
    const long r1 = min(rows, r0 + rows_per_blk);
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = b[k] = 0.f;
    float sc[8], sh[8];
    if (RELU == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = coef[tc * 8 + k]; sh[k] = coef[C + tc * 8 + k]; }
    }
    if (tr < rpp) {
#pragma unroll 4
        for (long row = r0 + tr; row < r1; row += rpp) {
            const long o = row * C + tc * 8;
            float g[8], xv[8], yv[8];
            load8(gy + o, g);
            load8(x + o, xv);
            if (RELU == 1) load8(y + o, yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (RELU == 2) yv[k] = xv[k] * sc[k] + sh[k];
                const float gp = (RELU && !(yv[k] > 0.f)) ? 0.f : g[k];
                a[k] += gp;
                b[k] += gp * xv[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { s_red[threadIdx.x][k] = a[k]; s_red[threadIdx.x][8 + k] = b[k]; }
    __syncthreads();
    // thread t < 2*C handles (which = t / C, channel = t % C)
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        const int which = t / C, c = t % C;
        const int tcc = c / 8, k = c % 8;
        float v = 0.f;
        for (int rr = 0; rr < rpp; ++rr) v += s_red[rr * tpr + tcc][which * 8 + k];
        atomicAdd(&part[((long)(blockIdx.x & fold_mask) * 2 + which) * C + c], v);
    }
}

// reduce the partials, emit per-channel backward coefficients kc [3][C] (g_x = kc0*g_pre + kc1*x + kc2) and, if
// requested, the affine-parameter gradients dgamma / dbeta (f32 [C]).
__global__ __launch_bounds__(1024) void bn_bwd_coef_kernel(const float* __restrict__ part, int nblk, double count,
                                                           const float* __restrict__ coef, float* __restrict__ kc,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
    __shared__ double s_a[16][64], s_b[16][64];
    const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double a = 0, b = 0;
    for (int t = tg; t < nblk; t += 16) {
        a += (double)part[((long)t * 2 + 0) * C + c];
        b += (double)part[((long)t * 2 + 1) * C + c];
    }
    s_a[tg][cl] = a; s_b[tg][cl] = b;
    __syncthreads();
    if (tg == 0) {
#pragma unroll
        for (int i = 1; i < 16; ++i) { a += s_a[i][cl]; b += s_b[i][cl]; }
        const double scale = coef[c], mean = coef[2 * C + c], invstd = coef[3 * C + c];
        const double dbeta_ = a;                               // sum g_pre
        const double dgam = invstd * (b - mean * a);           // sum g_pre * xhat
        // g_x = scale * (g_pre - dbeta/cnt - xhat * dgam/cnt),  xhat = (x - mean) * invstd
        kc[c] = (float)scale;
        kc[C + c] = (float)(-scale * invstd * dgam / count);
        kc[2 * C + c] = (float)(-scale * dbeta_ / count + scale * invstd * mean * dgam / count);
        if (dgamma) dgamma[c] = (float)dgam;
        if (dbeta) dbeta[c] = (float)dbeta_;
    }
}

// pass 2: g_x = kc0*g_pre + kc1*x + kc2 ; optionally also store g_pre (the residual branch's gradient)
template <int RELU, bool WRITE_GPRE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                           const bf16_t* __restrict__ x, const float* __restrict__ kc,
                                                           const float* __restrict__ coef, bf16_t* __restrict__ gx,
                                                           bf16_t* __restrict__ gpre, long n8, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float g[8], xv[8], yv[8], o[8], gp[8];
    load8(gy + i * 8, g);
    load8(x + i * 8, xv);
    if (RELU == 1) load8(y + i * 8, yv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (RELU == 2) yv[k] = xv[k] * coef[c0 + k] + coef[C + c0 + k];
        gp[k] = (RELU && !(yv[k] > 0.f)) ? 0.f : g[k];
        o[k] = kc[c0 + k] * gp[k] + kc[C + c0 + k] * xv[k] + kc[2 * C + c0 + k];
    }
    store8(gx + i * 8, o);
    if (WRITE_GPRE) store8(gpre + i * 8, gp);
}

// pass 2, coefficients in the prologue: every workgroup sums the 8 folded partial rows for its own channel slice
// (<= 256 channels, 16 KB of L2-resident f32) and derives kc itself, so the separate coefficient launch and its two
// kernel boundaries disappear.  grid (row slabs, C / CS), CS = min(C, 256); block = (256 / (CS/8)) rows x CS/8 lanes.
// SUMS2: the same gradient also feeds a second BatchNorm (the projection shortcut's, raw output x2): its backward sums (sum g,
// sum g * x2) are taken here into part2 [8][2][C] (pre-zeroed), one extra read of x2 instead of that BN's own reduce pass.
template <int RELU, bool WRITE_GPRE, bool SUMS2 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                                 const bf16_t* __restrict__ x, const float* __restrict__ part,
                                                                 const float* __restrict__ coef, double count,
                                                                 bf16_t* __restrict__ gx, bf16_t* __restrict__ gpre,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 long rows, int C, int rows_per_blk, int fold_rows,
                                                                 const bf16_t* __restrict__ x2, float* __restrict__ part2) {
    __shared__ float s_k[5][256];
    __shared__ float s_red2[SUMS2 ? 256 : 1][17];
    const int CS = C < 256 ? C : 256;
    const int cb = blockIdx.y * CS;
    if ((int)threadIdx.x < CS) {
        const int c = cb + threadIdx.x;
        double a = 0, b = 0;
#pragma unroll 8
        for (int t = 0; t < fold_rows; ++t) {
            a += (double)part[((long)t * 2 + 0) * C + c];
            b += (double)part[((long)t * 2 + 1) * C + c];
        }
        const double scale = coef[c], mean = coef[2 * C + c], invstd = coef[3 * C + c];
        const double dgam = invstd * (b - mean * a);
        s_k[0][threadIdx.x] = (float)scale;
        s_k[1][threadIdx.x] = (float)(-scale * invstd * dgam / count);
        s_k[2][threadIdx.x] = (float)(-scale * a / count + scale * invstd * mean * dgam / count);
        s_k[3][threadIdx.x] = (float)scale;
        s_k[4][threadIdx.x] = coef[C + c];
        if (blockIdx.x == 0) {
            if (dgamma) dgamma[c] = (float)dgam;
            if (dbeta) dbeta[c] = (float)a;
        }
    }
    __syncthreads();
    const int tpr = CS / 8, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    float k0[8], k1[8], k2[8], sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        k0[k] = s_k[0][tc * 8 + k]; k1[k] = s_k[1][tc * 8 + k]; k2[k] = s_k[2][tc * 8 + k];
        if (RELU == 2) { sc[k] = s_k[3][tc * 8 + k]; sh[k] = s_k[4][tc * 8 + k]; }
    }
    const long r0 = (long)blockIdx.x * rows_per_blk;
    const long r1 = min(rows, r0 + rows_per_blk);
    float a2[8], b2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a2[k] = b2[k] = 0.f;
#pragma unroll 4
    for (long row = r0 + tr; row < r1; row += rpp) {
        const long o_ = row * C + cb + tc * 8;
        float g[8], xv[8], yv[8], o[8], gp[8];
        load8(gy + o_, g);
        load8(x + o_, xv);
        if (RELU == 1) load8(y + o_, yv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (RELU == 2) yv[k] = xv[k] * sc[k] + sh[k];
            gp[k] = (RELU && !(yv[k] > 0.f)) ? 0.f : g[k];
            o[k] = k0[k] * gp[k] + k1[k] * xv[k] + k2[k];
        }
        store8(gx + o_, o);
        if (WRITE_GPRE) store8(gpre + o_, gp);
        if constexpr (SUMS2) {
            float x2v[8];
            load8(x2 + o_, x2v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { a2[k] += gp[k]; b2[k] += gp[k] * x2v[k]; }
        }
    }
    if constexpr (SUMS2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s_red2[threadIdx.x][k] = a2[k]; s_red2[threadIdx.x][8 + k] = b2[k]; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * CS; t += 256) {
            const int which = t / CS, c = t % CS;
            float v = 0.f;
            for (int rr = 0; rr < rpp; ++rr) v += s_red2[rr * tpr + c / 8][which * 8 + c % 8];
            atomicAdd(&part2[((long)(blockIdx.x & 7) * 2 + which) * C + cb + c], v);
        }
    }
}

// ----------------------------------------------------------------------------- stem: BN + ReLU + MaxPool 3x3/2 pad 1
// raw [B,H,W,C] bf16 -> pooled [B,H/2,W/2,C] bf16 (+ argmax tap 0..8, first maximum in scan order as torch)
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const bf16_t* __restrict__ x, const float* __restrict__ coef,
                                                              bf16_t* __restrict__ y, unsigned char* __restrict__ arg,
                                                              int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)B * Ho * Wo * c8n;
    if (i >= tot) return;
    const int c0 = (int)(i % c8n) * 8;
    const int wo = (int)((i / c8n) % Wo), ho = (int)((i / ((long)c8n * Wo)) % Ho), b = (int)(i / ((long)c8n * Wo * Ho));
    // The nine taps are requested TOGETHER from clamped coordinates and masked afterwards: behind a bounds branch per tap (round 1-4
    // form) every tap was its own load -> wait -> compare round trip, nine serial L2 / HBM latencies per thread (147 us for 335 MB).
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = coef[c0 + k]; sh[k] = coef[C + c0 + k]; }
    uint4 raw[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int h = 2 * ho - 1 + t / 3, w = 2 * wo - 1 + t % 3;
        ok[t] = ((unsigned)h < (unsigned)H) & ((unsigned)w < (unsigned)W);
        const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
        raw[t] = *reinterpret_cast<const uint4*>(x + (((long)b * H + hc) * W + wc) * C + c0);
    }
    float best[8];
    int bi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float xv[8];
        unpack8u(raw[t], xv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // the activation is stored as bf16 downstream: compare the ROUNDED values
            float v = bf2f_(f2bf_(fmaxf(xv[k] * sc[k] + sh[k], 0.f)));
            v = ok[t] ? v : -INFINITY;
            if (v > best[k]) { best[k] = v; bi[k] = t; }
        }
    }
    store8(y + i * 8, best);
    unsigned long long packed = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) packed |= (unsigned long long)(best[k] > 0.f ? bi[k] : 0xff) << (8 * k);   // 0xff: the window's maximum is 0 (no tap takes a gradient)
    *reinterpret_cast<unsigned long long*>(arg + i * 8) = packed;
}

// backward of the above down to the ReLU input: g_pre[b,h,w,c] = sum over the <= 4 windows that picked (h,w)
__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                               const unsigned char* __restrict__ arg,
                                                               bf16_t* __restrict__ gpre, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)B * H * W * c8n;
    if (i >= tot) return;
    const int c0 = (int)(i % c8n) * 8;
    const int w = (int)((i / c8n) % W), h = (int)((i / ((long)c8n * W)) % H), b = (int)(i / ((long)c8n * W * H));
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    // windows (ho,wo) with 2ho-1+r = h  ->  ho in {(h+1)/2 - 1 .. (h+1)/2}, r = h + 1 - 2 ho in [0,2]
    for (int ho = (h + 1) / 2 - 1; ho <= (h + 1) / 2; ++ho) {
        const int r = h + 1 - 2 * ho;
        if (ho < 0 || ho >= Ho || r < 0 || r > 2) continue;
        for (int wo = (w + 1) / 2 - 1; wo <= (w + 1) / 2; ++wo) {
            const int s = w + 1 - 2 * wo;
            if (wo < 0 || wo >= Wo || s < 0 || s > 2) continue;
            const long o = (((long)b * Ho + ho) * Wo + wo) * C + c0;
            const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(arg + o);
            float g[8], yv[8];
            load8(gy + o, g);
            load8(y + o, yv);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((int)((packed >> (8 * k)) & 0xff) == r * 3 + s && yv[k] > 0.f) acc[k] += g[k];
        }
    }
    store8(gpre + i * 8, acc);
}

// ----------------------------------------------------------------------------- stem backward in two passes instead of three (round 3)
// max-pool backward -> BatchNorm backward of the stem used to be maxpool_relu_bwd (writes the 268-MB pre-pool gradient), bn_bwd_reduce
// (reads it + the raw conv output) and the apply pass (reads both again, writes g_x): 1.8 GB.  The pre-pool gradient is a GATHER of
// the pooled gradient (<= 4 windows per pixel), so both BatchNorm passes can take it straight from the 67-MB pooled tensors: pass 1
// = the sums (sum g_pre, sum g_pre * x per channel, 8 partial rows), pass 2 = coefficients in the prologue + g_x.  1.1 GB.
// The <= 2 x 2 windows a pre-pool pixel can belong to are fetched TOGETHER from clamped indices and masked afterwards (a bounds branch
// per window made every window its own arg / g / y round trip: four serial latencies per pixel, 219 us for the stem's 700 MB).
__device__ __forceinline__ void maxpool_gather8(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                const unsigned char* __restrict__ arg, int b, int h, int w, int c0, int Ho, int Wo, int C,
                                                float (&acc)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    const int ho0 = (h + 1) / 2 - 1, wo0 = (w + 1) / 2 - 1;
    unsigned long long pk[4];
    uint4 gv[4];
    int tap[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ho = ho0 + (q >> 1), wo = wo0 + (q & 1);
        const int r = h + 1 - 2 * ho, s = w + 1 - 2 * wo;
        ok[q] = ((unsigned)ho < (unsigned)Ho) & ((unsigned)wo < (unsigned)Wo) & ((unsigned)r <= 2u) & ((unsigned)s <= 2u);
        tap[q] = r * 3 + s;
        const int hc = min(max(ho, 0), Ho - 1), wc = min(max(wo, 0), Wo - 1);
        const long o = (((long)b * Ho + hc) * Wo + wc) * C + c0;
        // lanes whose candidate is not a window of their pixel (2.25 of the 4 are, on average) do not fetch: the loads are predicated
        // per lane, the use below is arithmetic (no second branch for the compiler to sink the loads into)
        pk[q] = 0xffffffffffffffffull;
        gv[q] = make_uint4(0, 0, 0, 0);
        if (ok[q]) {                                           // (the pooled activation itself is not needed: a window whose maximum is 0
            pk[q] = *reinterpret_cast<const unsigned long long*>(arg + o);   //  carries the tap byte 0xff, bn_relu_maxpool_kernel)
            gv[q] = *reinterpret_cast<const uint4*>(gy + o);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float g[8];
        unpack8u(gv[q], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += ((int)((pk[q] >> (8 * k)) & 0xff) == tap[q]) ? g[k] : 0.f;   // (no fetch: tap byte 0xff)
    }
}

// APPLY = false: part [8][2][C] (pre-zeroed) += sums over this workgroup's pixels; APPLY = true: g_x = k0 g_pre + k1 x + k2 with the
// coefficients derived from part in the prologue (as bn_bwd_apply_fused_kernel), dgamma / dbeta by workgroup 0.  C <= 256, C % 8 == 0;
// a workgroup walks ppw consecutive pixels per thread row.
template <bool APPLY>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                             const unsigned char* __restrict__ arg, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ coef, double count, float* __restrict__ part,
                                                             bf16_t* __restrict__ gx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int B, int H, int W, int C, int ppw) {
    __shared__ float s_k[3][256];
    __shared__ float s_red[256][17];
    const int Ho = H / 2, Wo = W / 2, tpr = C / 8, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr, c0 = tc * 8;
    if (APPLY) {
        if ((int)threadIdx.x < C) {
            const int c = threadIdx.x;
            double a = 0, b = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) { a += (double)part[((long)t * 2 + 0) * C + c]; b += (double)part[((long)t * 2 + 1) * C + c]; }
            const double scale = coef[c], mean = coef[2 * C + c], invstd = coef[3 * C + c];
            const double dgam = invstd * (b - mean * a);
            s_k[0][c] = (float)scale;
            s_k[1][c] = (float)(-scale * invstd * dgam / count);
            s_k[2][c] = (float)(-scale * a / count + scale * invstd * mean * dgam / count);
            if (blockIdx.x == 0) {
                if (dgamma) dgamma[c] = (float)dgam;
                if (dbeta) dbeta[c] = (float)a;
            }
        }
        __syncthreads();
    }
    float k0[8], k1[8], k2[8], sa[8], sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sa[k] = sb[k] = 0.f;
        if (APPLY) { k0[k] = s_k[0][c0 + k]; k1[k] = s_k[1][c0 + k]; k2[k] = s_k[2][c0 + k]; }
    }
    const long npix = (long)B * H * W;
    const long p0 = (long)blockIdx.x * rpp * ppw;
    if (tr < rpp) {
        for (int it = 0; it < ppw; ++it) {
            const long pix = p0 + (long)it * rpp + tr;
            if (pix >= npix) break;
            const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
            float acc[8], xv[8];
            const uint4 xr = *reinterpret_cast<const uint4*>(x + pix * C + c0);   // in flight with the gather's loads
            maxpool_gather8(gy, y, arg, b, h, w, c0, Ho, Wo, C, acc);
            unpack8u(xr, xv);
            if (APPLY) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = k0[k] * acc[k] + k1[k] * xv[k] + k2[k];
                store8(gx + pix * C + c0, o);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { sa[k] += acc[k]; sb[k] += acc[k] * xv[k]; }
            }
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s_red[threadIdx.x][k] = sa[k]; s_red[threadIdx.x][8 + k] = sb[k]; }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * C; t += 256) {
            const int which = t / C, c = t % C;
            float v = 0.f;
            for (int rr = 0; rr < rpp; ++rr) v += s_red[rr * tpr + c / 8][which * 8 + c % 8];
            atomicAdd(&part[((long)(blockIdx.x & 7) * 2 + which) * C + c], v);
        }
    }
}

// The sums of the pass above taken over the POOLED tensors only.  A pooled element with y > 0 sends its gradient to exactly one
// pre-pool position (its argmax), so  sum_p g_pre[p] = sum_o g[o] [y[o] > 0]  and  sum_p g_pre[p] x[p] = sum_o g[o] [y[o] > 0] x[argmax(o)],
// and x at the argmax follows from the pooled activation itself: y = bf16(x * scale + shift) there, x = (y - shift) / scale (one bf16
// rounding of the BatchNorm output away from the stored x).  The reconstruction loses 2^-9 |y| / |gamma| of x-hat: harmless while
// |gamma| is comparable with |beta|, catastrophic for the near-dead channels of an ImageNet-pretrained bn1 (gamma down to 1e-8 next to
// beta ~ 0.1: y = bf16(beta + gamma x-hat) keeps nothing of x-hat).  Channels with |gamma| < |beta| / 8 (or no usable scale at all)
// therefore read x at the arg-max position of the raw tensor -- exact, a gather for those channels only (r3 advisor).
// 134 MB instead of 400 MB at B = 128 (the raw stem output is not read).  part [8][2][C] pre-zeroed.
__global__ __launch_bounds__(256) void pooled_bn_sums_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ y,
                                                             const unsigned char* __restrict__ arg, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ coef, float* __restrict__ part, int B, int H,
                                                             int W, int C, int ppw) {
    __shared__ float s_red[256][17];
    const int Ho = H / 2, Wo = W / 2, tpr = C / 8, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr, c0 = tc * 8;
    float isc[8], sh[8], sa[8], sb[8];
    bool exact[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float sc = coef[c0 + k], shf = coef[C + c0 + k], mean = coef[2 * C + c0 + k], istd = coef[3 * C + c0 + k];
        const float gamma = sc / istd, beta = shf + mean * sc;        // coef rows: scale = gamma invstd, shift = beta - mean scale, mean, invstd
        exact[k] = !(fabsf(sc) >= 1e-20f) || !(fabsf(gamma) >= 0.125f * fabsf(beta));
        isc[k] = exact[k] ? 0.f : 1.f / sc;
        sh[k] = shf;
        sa[k] = sb[k] = 0.f;
    }
    const long npool = (long)B * Ho * Wo;
    const long p0 = (long)blockIdx.x * rpp * ppw;
    if (tr < rpp) {
        for (int it0 = 0; it0 < ppw; it0 += 4) {                   // four pooled pixels in flight per thread
            uint4 gr[4], yr[4];
            long oo[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = p0 + (long)(it0 + u) * rpp + tr;
                oo[u] = (it0 + u < ppw && o < npool) ? o : -1;
                const long oc = oo[u] < 0 ? 0 : o;
                gr[u] = *reinterpret_cast<const uint4*>(gy + oc * C + c0);
                yr[u] = *reinterpret_cast<const uint4*>(y + oc * C + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
            const long o = oo[u];
            if (o < 0) continue;
            float g[8], yv[8];
            unpack8u(gr[u], g);
            unpack8u(yr[u], yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!(yv[k] > 0.f)) continue;
                float xv = (yv[k] - sh[k]) * isc[k];
                if (exact[k]) {                                   // near-dead channel: x from the raw tensor at the arg-max
                    const int wo = (int)(o % Wo), ho = (int)((o / Wo) % Ho), b = (int)(o / ((long)Wo * Ho));
                    const int a = arg[o * C + c0 + k], r = a / 3, s_ = a % 3;
                    const int h = 2 * ho - 1 + r, w = 2 * wo - 1 + s_;
                    xv = bf2f_(x[(((long)b * H + h) * W + w) * C + c0 + k]);
                }
                sa[k] += g[k];
                sb[k] += g[k] * xv;
            }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { s_red[threadIdx.x][k] = sa[k]; s_red[threadIdx.x][8 + k] = sb[k]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += 256) {
        const int which = t / C, c = t % C;
        float v = 0.f;
        for (int rr = 0; rr < rpp; ++rr) v += s_red[rr * tpr + c / 8][which * 8 + c % 8];
        atomicAdd(&part[((long)(blockIdx.x & 7) * 2 + which) * C + c], v);
    }
}

// ----------------------------------------------------------------------------- AdaptiveAvgPool2d(E) on [B,H,W,C]
// window of output i: [floor(i*H/E), ceil((i+1)*H/E))  (torch adaptive pooling)
template <typename TO>
__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const bf16_t* __restrict__ x, TO* __restrict__ y, int B, int H,
                                                                int W, int C, int E) {
    const int c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)B * E * E * c8n;
    if (i >= tot) return;
    const int c0 = (int)(i % c8n) * 8;
    const int ox = (int)((i / c8n) % E), oy = (int)((i / ((long)c8n * E)) % E), b = (int)(i / ((long)c8n * E * E));
    const int h0 = (oy * H) / E, h1 = ((oy + 1) * H + E - 1) / E, w0 = (ox * W) / E, w1 = ((ox + 1) * W + E - 1) / E;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int h = h0; h < h1; ++h)
        for (int w = w0; w < w1; ++w) {
            float v[8];
            load8(x + (((long)b * H + h) * W + w) * C + c0, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
    const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
    TO* o = y + i * 8;
    if constexpr (sizeof(TO) == 4) {
        // 1.36 GB at B = 128: streamed out once, far larger than the Infinity Cache -> non-temporal 16-byte stores
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        const f32x4_ lo = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
        const f32x4_ hi = {acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
        __builtin_nontemporal_store(lo, reinterpret_cast<f32x4_*>(o));
        __builtin_nontemporal_store(hi, reinterpret_cast<f32x4_*>(o) + 1);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (TO)(acc[k] * inv);
    }
}

template <typename TG>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const TG* __restrict__ gy, bf16_t* __restrict__ gx,
                                                                const bf16_t* __restrict__ mask_src, int B, int H, int W, int C,
                                                                int E) {
    const int c8n = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)B * H * W * c8n;
    if (i >= tot) return;
    const int c0 = (int)(i % c8n) * 8;
    const int w = (int)((i / c8n) % W), h = (int)((i / ((long)c8n * W)) % H), b = (int)(i / ((long)c8n * W * H));
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    // outputs whose window contains h: oy in [floor(h*E/H) - 1, ceil((h+1)*E/H)]
    const int oy_lo = max(0, (h * E) / H - 1), oy_hi = min(E - 1, ((h + 1) * E + H - 1) / H);
    const int ox_lo = max(0, (w * E) / W - 1), ox_hi = min(E - 1, ((w + 1) * E + W - 1) / W);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const int h0 = (oy * H) / E, h1 = ((oy + 1) * H + E - 1) / E;
        if (h < h0 || h >= h1) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int w0 = (ox * W) / E, w1 = ((ox + 1) * W + E - 1) / E;
            if (w < w0 || w >= w1) continue;
            const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
            const TG* g = gy + ((((long)b * E + oy) * E + ox) * C + c0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += (float)g[k] * inv;
        }
    }
    if (mask_src) {                                   // ReLU backward of the tensor that was pooled (y > 0)
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((short)mask_src[i * 8 + k] <= 0) acc[k] = 0.f;
    }
    store8(gx + i * 8, acc);
}

// ---- round 6: the same two passes with NO per-thread index arithmetic.  The kernels above decode (b, oy, ox, c) from a flat thread index
// (three integer divisions per thread, two more per window) for 32 bytes of output: the dense 1.36-GB f32 tensor left at 4.1 TB/s and its
// gradient came back at 4.6 where a plain store / load stream runs at 6-6.8 (tools/micro/write_pattern.hip).  Here a workgroup owns one
// output row (forward) / one input pixel (backward): the window bounds are wave-uniform, a lane owns four channels per 1024-channel
// chunk, and every wave store / load is one contiguous KB.  C % 4 == 0.
template <typename TO>
__global__ __launch_bounds__(256) void adaptive_pool_fwd_row_kernel(const bf16_t* __restrict__ x, TO* __restrict__ y, int H, int W, int C, int E) {
    const int oy = blockIdx.x, b = blockIdx.y;
    const int h0 = (oy * H) / E, h1 = ((oy + 1) * H + E - 1) / E;
    const bf16_t* xb = x + (long)b * H * W * C;
    TO* ob = y + ((long)b * E + oy) * E * C;
    // windows are one or two pixels wide / high (E >= H, W): the (up to) four taps are loaded unconditionally from clamped positions and
    // weighted 0 / 1 -- no data-dependent control flow, so the loads of several output pixels are in flight together
    const int hb = (h1 - h0 > 1) ? h0 + 1 : h0;
    const float kh = (h1 - h0 > 1) ? 1.f : 0.f;
#pragma unroll 4
    for (int ox = 0; ox < E; ++ox) {
        const int w0 = (ox * W) / E, w1 = ((ox + 1) * W + E - 1) / E;
        const int wb = (w1 - w0 > 1) ? w0 + 1 : w0;
        const float kw = (w1 - w0 > 1) ? 1.f : 0.f;
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll 2
        for (int c = threadIdx.x * 4; c < C; c += 1024) {
            const uint2 v00 = *reinterpret_cast<const uint2*>(xb + ((long)h0 * W + w0) * C + c);
            const uint2 v01 = *reinterpret_cast<const uint2*>(xb + ((long)h0 * W + wb) * C + c);
            const uint2 v10 = *reinterpret_cast<const uint2*>(xb + ((long)hb * W + w0) * C + c);
            const uint2 v11 = *reinterpret_cast<const uint2*>(xb + ((long)hb * W + wb) * C + c);
            auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
            auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
            const float k01 = kw, k10 = kh, k11 = kw * kh;
            float a0 = lo(v00.x) + k01 * lo(v01.x) + k10 * lo(v10.x) + k11 * lo(v11.x);
            float a1 = hi(v00.x) + k01 * hi(v01.x) + k10 * hi(v10.x) + k11 * hi(v11.x);
            float a2 = lo(v00.y) + k01 * lo(v01.y) + k10 * lo(v10.y) + k11 * lo(v11.y);
            float a3 = hi(v00.y) + k01 * hi(v01.y) + k10 * hi(v10.y) + k11 * hi(v11.y);
            TO* o = ob + (long)ox * C + c;
            if constexpr (sizeof(TO) == 4) {
                typedef float f32x4_ __attribute__((ext_vector_type(4)));
                const f32x4_ r = {a0 * inv, a1 * inv, a2 * inv, a3 * inv};
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4_*>(o));     // streamed out once, far larger than the Infinity Cache
            } else {
                o[0] = (TO)(a0 * inv); o[1] = (TO)(a1 * inv); o[2] = (TO)(a2 * inv); o[3] = (TO)(a3 * inv);
            }
        }
    }
}

template <typename TG>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_px_kernel(const TG* __restrict__ gy, bf16_t* __restrict__ gx,
                                                                   const bf16_t* __restrict__ mask_src, int H, int W, int C, int E) {
    const int w = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    // outputs whose window contains h / w, with the reciprocal of the window's extent (wave-uniform; at most MAXO per axis)
    constexpr int MAXO = 8;
    __shared__ int s_oy[MAXO], s_ox[MAXO];
    __shared__ float s_iy[MAXO], s_ix[MAXO];
    __shared__ int s_n[2];
    if (threadIdx.x == 0) {
        int ny = 0, nx = 0;
        for (int oy = max(0, (h * E) / H - 1); oy <= min(E - 1, ((h + 1) * E + H - 1) / H); ++oy) {
            const int h0 = (oy * H) / E, h1 = ((oy + 1) * H + E - 1) / E;
            if (h >= h0 && h < h1 && ny < MAXO) { s_oy[ny] = oy; s_iy[ny] = 1.f / (float)(h1 - h0); ++ny; }
        }
        for (int ox = max(0, (w * E) / W - 1); ox <= min(E - 1, ((w + 1) * E + W - 1) / W); ++ox) {
            const int w0 = (ox * W) / E, w1 = ((ox + 1) * W + E - 1) / E;
            if (w >= w0 && w < w1 && nx < MAXO) { s_ox[nx] = ox; s_ix[nx] = 1.f / (float)(w1 - w0); ++nx; }
        }
        s_n[0] = ny; s_n[1] = nx;
    }
    __syncthreads();
    const int ny = s_n[0], nx = s_n[1];
    const long pix = ((long)b * H + h) * W + w;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int i = 0; i < ny; ++i) {
            const TG* row = gy + (((long)b * E + s_oy[i]) * E) * C + c;
            const float iy = s_iy[i];
#pragma unroll
            for (int j = 0; j < MAXO; ++j) {                   // (unrolled: the loads of one output row are in flight together)
                if (j < nx) {
                    const TG* g = row + (long)s_ox[j] * C;
                    const float k = iy * s_ix[j];
                    a0 += (float)g[0] * k; a1 += (float)g[1] * k; a2 += (float)g[2] * k; a3 += (float)g[3] * k;
                }
            }
        }
        if (mask_src) {                                        // ReLU backward of the tensor that was pooled (y > 0)
            const uint2 m = *reinterpret_cast<const uint2*>(mask_src + pix * C + c);
            if ((short)(m.x & 0xffffu) <= 0) a0 = 0.f;
            if ((short)(m.x >> 16) <= 0) a1 = 0.f;
            if ((short)(m.y & 0xffffu) <= 0) a2 = 0.f;
            if ((short)(m.y >> 16) <= 0) a3 = 0.f;
        }
        uint2 o;
        o.x = (unsigned)f2bf_(a0) | ((unsigned)f2bf_(a1) << 16);
        o.y = (unsigned)f2bf_(a2) | ((unsigned)f2bf_(a3) << 16);
        *reinterpret_cast<uint2*>(gx + pix * C + c) = o;
    }
}

}  // namespace ppv

using namespace ppv;

static int env_int_(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

extern "C" {

int ppv_bn_finalize(const float* part, int T, double count, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float momentum, float eps, float* coef, int C, hipStream_t stream) {
    if (!part || !gamma || !beta || !coef) return PPV_ERR_NULL;
    if (C % 64) return PPV_ERR_BAD_SIZE;
    bn_finalize_kernel<<<C / 64, 1024, 0, stream>>>(part, T, count, gamma, beta, run_mean, run_var, momentum, eps, coef, C);
    return ppv_last_error();
}

// y = act(x*s1+t1 + res);  res_mode 0 none, 1 identity r, 2 r*s2+t2 (coef2); res_mod > 0: r holds res_mod elements and
// is broadcast (index modulo), e.g. a per-pixel constant map shared by the batch
// pos_bits (n / 8 bytes, may be null): bit k of byte i = (y[8 i + k] > 0), consumed by ppv_conv_gemm's mask_bits
int ppv_bn_act(const void* x, const float* coef1, const void* r, const float* coef2, void* y, void* pos_bits, long n, int C,
               int res_mode, int relu, long res_mod, hipStream_t stream) {
    if (!x || !coef1 || !y || (res_mode && !r) || (res_mode == 2 && !coef2)) return PPV_ERR_NULL;
    if (C % 8 || n % 8) return PPV_ERR_BAD_SIZE;
    const long n8 = n / 8;
    const unsigned gb = (unsigned)((n8 + 255) / 256);
    const bf16_t *xx = (const bf16_t*)x, *rr = (const bf16_t*)r;
    bf16_t* yy = (bf16_t*)y;
    unsigned char* pb = (unsigned char*)pos_bits;
    if (res_mode == 0 && relu) bn_act_kernel<0, true><<<gb, 256, 0, stream>>>(xx, coef1, rr, coef2, yy, pb, n8, C, res_mod / 8);
    else if (res_mode == 0) bn_act_kernel<0, false><<<gb, 256, 0, stream>>>(xx, coef1, rr, coef2, yy, pb, n8, C, res_mod / 8);
    else if (res_mode == 1 && relu) bn_act_kernel<1, true><<<gb, 256, 0, stream>>>(xx, coef1, rr, coef2, yy, pb, n8, C, res_mod / 8);
    else if (res_mode == 1) bn_act_kernel<1, false><<<gb, 256, 0, stream>>>(xx, coef1, rr, coef2, yy, pb, n8, C, res_mod / 8);
    else if (relu) bn_act_kernel<2, true><<<gb, 256, 0, stream>>>(xx, coef1, rr, coef2, yy, pb, n8, C, res_mod / 8);
    else bn_act_kernel<2, false><<<gb, 256, 0, stream>>>(xx, coef1, rr, coef2, yy, pb, n8, C, res_mod / 8);
    return ppv_last_error();
}

// Train-mode BatchNorm in one launch: statistics [T][2][C] (as ppv_conv_gemm leaves them) -> coefficients (coef [4][C] written for
// the backward pass, running statistics updated) -> y = act(x * s + t (+ res)).  res_mode 0 none, 1 identity r, 2 r normalised by
// its own BatchNorm (part2 ... coef2: the projection shortcut's).  Replaces ppv_bn_finalize + ppv_bn_act (reference:
// torch BatchNorm2d in training mode + ReLU (+ residual add) of torchvision's Bottleneck, Image_Caption/models.py:17-21, train.py:245).
// C % 64 == 0 (C <= 256) or C % 256 == 0; rows * C < 2^31 * 8.
// y = act(BatchNorm_train(x) (+ r)) with the statistics in ONE row sums [2][C] (sum, sum of squares of the stored conv output, as
// ppv_conv_gemm leaves them with stat_rows = 1): coefficients derived per thread, coef [4][C] and the running statistics written
// by the first row's threads.  res_mode 0 none, 1 identity residual r.  pos_bits as ppv_bn_act.  C % 8 == 0, rows * C % 8 == 0.
static int bn_act_fold_impl(const void* x, const float* sums, int T, double count, const float* gamma, const float* beta, float* run_mean,
                            float* run_var, float momentum, float eps, float* coef, const void* r, void* y, void* pos_bits, long n, int C,
                            int res_mode, int relu, hipStream_t stream);
int ppv_bn_act_fold(const void* x, const float* sums, double count, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float momentum, float eps, float* coef, const void* r, void* y, void* pos_bits, long n, int C,
                    int res_mode, int relu, hipStream_t stream) {
    return bn_act_fold_impl(x, sums, 1, count, gamma, beta, run_mean, run_var, momentum, eps, coef, r, y, pos_bits, n, C, res_mode, relu, stream);
}
// sums [T][2][C]: the statistics in T partial rows (ppv_conv_gemm with stat_rows = T); C / 8 must be a power of two <= 256 when T > 1
int ppv_bn_act_fold_rows(const void* x, const float* sums, int T, double count, const float* gamma, const float* beta, float* run_mean,
                         float* run_var, float momentum, float eps, float* coef, const void* r, void* y, void* pos_bits, long n, int C,
                         int res_mode, int relu, hipStream_t stream) {
    return bn_act_fold_impl(x, sums, T, count, gamma, beta, run_mean, run_var, momentum, eps, coef, r, y, pos_bits, n, C, res_mode, relu, stream);
}
static int bn_act_fold_impl(const void* x, const float* sums, int T, double count, const float* gamma, const float* beta, float* run_mean,
                            float* run_var, float momentum, float eps, float* coef, const void* r, void* y, void* pos_bits, long n, int C,
                            int res_mode, int relu, hipStream_t stream) {
    if (!x || !sums || !gamma || !beta || !coef || !y || (res_mode == 1 && !r)) return PPV_ERR_NULL;
    if (C % 8 || n % 8 || res_mode < 0 || res_mode > 1 || count < 1 || (run_mean && !run_var) || T < 1) return PPV_ERR_BAD_SIZE;
    if (T > 1 && !(C / 8 <= 256 && ((C / 8) & (C / 8 - 1)) == 0)) return PPV_ERR_BAD_SIZE;
    const long n8 = n / 8;
    const unsigned gb = (unsigned)((n8 + 255) / 256);
    const double inv = 1.0 / count, unb = count > 1 ? count / (count - 1.0) : 1.0;
    const int tpr = C / 8;
    static const int per_thread = getenv("PPV_BN_FOLD_THREAD") ? atoi(getenv("PPV_BN_FOLD_THREAD")) : 0;
    if ((!per_thread || T > 1) && tpr <= 256 && (tpr & (tpr - 1)) == 0) {           // workgroup-shared coefficients
        int lg = 0;
        while ((1 << lg) < tpr) ++lg;
        constexpr int U = 4;
        const long rows = n / C;
        const int rpp = 256 / tpr;
        const unsigned g2 = (unsigned)((rows + (long)rpp * U - 1) / ((long)rpp * U));
#define PPV_FOLDW2(RES_, RELU_, CPT_) bn_act_fold_wg_kernel<RES_, RELU_, U, CPT_><<<g2, 256, 0, stream>>>((const bf16_t*)x, sums, inv, unb, gamma, beta, \
        run_mean, run_var, momentum, eps, coef, (const bf16_t*)r, (bf16_t*)y, (unsigned char*)pos_bits, rows, C, lg, T)
#define PPV_FOLDW(RES_, RELU_) do { if (C <= 256) PPV_FOLDW2(RES_, RELU_, 1); else if (C <= 512) PPV_FOLDW2(RES_, RELU_, 2); \
        else if (C <= 1024) PPV_FOLDW2(RES_, RELU_, 4); else PPV_FOLDW2(RES_, RELU_, 8); } while (0)
        if (res_mode == 1 && relu) PPV_FOLDW(1, true);
        else if (res_mode == 1) PPV_FOLDW(1, false);
        else if (relu) PPV_FOLDW(0, true);
        else PPV_FOLDW(0, false);
#undef PPV_FOLDW2
#undef PPV_FOLDW
        return ppv_last_error();
    }
#define PPV_FOLD(RES_, RELU_) bn_act_fold_kernel<RES_, RELU_><<<gb, 256, 0, stream>>>((const bf16_t*)x, sums, inv, unb, gamma, beta, run_mean, run_var, \
        momentum, eps, coef, (const bf16_t*)r, (bf16_t*)y, (unsigned char*)pos_bits, n8, C)
    if (res_mode == 1 && relu) PPV_FOLD(1, true);
    else if (res_mode == 1) PPV_FOLD(1, false);
    else if (relu) PPV_FOLD(0, true);
    else PPV_FOLD(0, false);
#undef PPV_FOLD
    return ppv_last_error();
}

int ppv_bn_act_train(const void* x, const float* part, int T, double count, const float* gamma, const float* beta, float* run_mean,
                     float* run_var, float momentum, float eps, float* coef, const void* r, const float* part2, int T2,
                     const float* gamma2, const float* beta2, float* run_mean2, float* run_var2, float momentum2, float eps2,
                     float* coef2, void* y, void* pos_bits, long rows, int C, int res_mode, int relu, hipStream_t stream) {
    if (!x || !part || !gamma || !beta || !coef || !y || (res_mode && !r)) return PPV_ERR_NULL;
    if (res_mode == 2 && (!part2 || !gamma2 || !beta2 || !coef2)) return PPV_ERR_NULL;
    if (C % 64 || (C > 256 && C % 256) || T < 1 || (res_mode == 2 && T2 < 1)) return PPV_ERR_BAD_SIZE;
    BnFold f1{part, T, count, gamma, beta, run_mean, run_var, momentum, eps, coef};
    BnFold f2{part2, T2, count, gamma2, beta2, run_mean2, run_var2, momentum2, eps2, coef2};
    const int CB = C < 256 ? C : 256, ncb = C / CB, rpp = 256 / (CB / 8);
    long rbk = (rows + rpp - 1) / rpp;                          // row blocks if every workgroup took one pass
    const long want = (512 + ncb - 1) / ncb;                    // ~512 workgroups (2 per CU), eight rows in flight per thread
    if (rbk > want) rbk = want;
    if (rbk < 1) rbk = 1;
    const unsigned grid = (unsigned)(rbk * ncb);
    const bf16_t *xx = (const bf16_t*)x, *rr = (const bf16_t*)r;
    bf16_t* yy = (bf16_t*)y;
    unsigned char* pb = (unsigned char*)pos_bits;
#define PPV_BNT(RES_, RELU_) bn_act_train_kernel<RES_, RELU_><<<grid, 256, 0, stream>>>(xx, f1, rr, f2, yy, pb, rows, C, ncb)
    if (res_mode == 0 && relu) PPV_BNT(0, true);
    else if (res_mode == 0) PPV_BNT(0, false);
    else if (res_mode == 1 && relu) PPV_BNT(1, true);
    else if (res_mode == 1) PPV_BNT(1, false);
    else if (relu) PPV_BNT(2, true);
    else PPV_BNT(2, false);
#undef PPV_BNT
    return ppv_last_error();
}

int ppv_bn_bwd_blocks(long rows, int C) {
    const int rpp = 256 / (C / 8);
    long rpb = (long)rpp * 16;
    long nb = (rows + rpb - 1) / rpb;
    while (nb > 2048) { rpb *= 2; nb = (rows + rpb - 1) / rpb; }
    return (int)nb;
}

// Train-mode BN backward; relu: 0 none, 1 ReLU mask from the stored activation y, 2 mask recomputed from x and coef
// (BN + ReLU without residual; y may be null).  Writes g_x (bf16), optionally g_pre (bf16, may be
// null), dgamma / dbeta (f32 [C], may be null).  part: scratch >= 64 * C floats; part_prezeroed 0: zeroed here, 1: the caller
// zeroed it, 2: it already holds the [8][2][C] sums (ppv_conv_gemm_red took them while storing gy; relu must be 0).  kc: scratch 3*C.
}  // extern "C"
namespace ppv {
static thread_local hipEvent_t t_stop_event = nullptr;
// one-shot: the next fused BatchNorm-backward apply launch of this thread signals `e` from its own dispatch packet (hipExtLaunchKernel)
void bn_bwd_stop_event_once(hipEvent_t e) { t_stop_event = e; }
// true (and cleared) when the launch that was to take the event did not run the fused apply kernel: the caller records an event itself
bool bn_bwd_stop_event_unused() { const bool u = t_stop_event != nullptr; t_stop_event = nullptr; return u; }
}  // namespace ppv
using ppv::t_stop_event;
extern "C" {
static int bn_bwd_impl(const void* gy, const void* y, const void* x, const float* coef, double count, void* gx, void* gpre,
                       float* dgamma, float* dbeta, float* part, float* kc, long rows, int C, int relu, int part_prezeroed,
                       const void* x2, float* part2, hipStream_t stream) {
    if (!gy || !x || !coef || !gx || !part || !kc || (relu == 1 && !y)) return PPV_ERR_NULL;
    if (C % 64 || C > 2048) return PPV_ERR_BAD_SIZE;
    static const int fused = env_int_("PPV_BN_BWD_FUSED", 1);
    const bool fuse = fused && (C == 64 || C == 128 || C % 256 == 0);
    const int rpp = 256 / (C / 8);
    static const int red_iters = env_int_("PPV_BN_RED_ITERS", 8), fold_rows = env_int_("PPV_BN_FOLD", 8);
    long rpb = (long)rpp * (fuse ? red_iters : 16);
    long nb = (rows + rpb - 1) / rpb;
    while (nb > (fuse ? 1024 : 2048)) { rpb *= 2; nb = (rows + rpb - 1) / rpb; }
    const bf16_t *g = (const bf16_t*)gy, *yy = (const bf16_t*)y, *xx = (const bf16_t*)x;
    if (part_prezeroed == 2 && (!fuse || fold_rows != 8)) return PPV_ERR_BAD_SIZE;   // sums already taken by ppv_conv_gemm_red
    if (!part_prezeroed) (void)hipMemsetAsync(part, 0, sizeof(float) * 64 * C, stream);
    const int fold = fuse ? fold_rows - 1 : 31;
    if (part_prezeroed == 2) {}
    else if (relu == 2) bn_bwd_reduce_kernel<2><<<(unsigned)nb, 256, 0, stream>>>(g, yy, xx, coef, part, rows, C, (int)rpb, fold);
    else if (relu) bn_bwd_reduce_kernel<1><<<(unsigned)nb, 256, 0, stream>>>(g, yy, xx, coef, part, rows, C, (int)rpb, fold);
    else bn_bwd_reduce_kernel<0><<<(unsigned)nb, 256, 0, stream>>>(g, yy, xx, coef, part, rows, C, (int)rpb, fold);
    bf16_t *ox = (bf16_t*)gx, *op = (bf16_t*)gpre;
    if (fuse) {
        static const int iters = env_int_("PPV_BN_BWD_ITERS", 8);
        const int CS = C < 256 ? C : 256;
        const int arpp = 256 / (CS / 8);
        const long arpb = (long)arpp * iters;
        const dim3 grid((unsigned)((rows + arpb - 1) / arpb), (unsigned)(C / CS));
        // a stop event asked for by the executor (ppv::bn_bwd_stop_event_once): the kernel's OWN dispatch packet signals it -- a fork of the
        // weight-gradient stream then needs no event-record packet behind this launch on the main chain
        const hipEvent_t stop_ev = t_stop_event;
        t_stop_event = nullptr;
#define PPV_APPLY(R, G)                                                                                                                      \
    do {                                                                                                                                     \
        if (stop_ev)                                                                                                                         \
            hipExtLaunchKernelGGL((bn_bwd_apply_fused_kernel<R, G>), grid, dim3(256), 0, stream, nullptr, stop_ev, 0, g, yy, xx, (const float*)part, \
                                  coef, count, ox, op, dgamma, dbeta, rows, C, (int)arpb, fold_rows, (const bf16_t*)nullptr, (float*)nullptr); \
        else                                                                                                                                 \
            bn_bwd_apply_fused_kernel<R, G><<<grid, 256, 0, stream>>>(g, yy, xx, part, coef, count, ox, op, dgamma, dbeta, rows, C, (int)arpb, \
                                                                      fold_rows, nullptr, nullptr);                                           \
    } while (0)
        if (x2) {                                               // projection-shortcut sums ride along (relu 0, no g_pre copy)
            if (relu || gpre || !part2) return PPV_ERR_BAD_SIZE;
            if (stop_ev)
                hipExtLaunchKernelGGL((bn_bwd_apply_fused_kernel<0, false, true>), grid, dim3(256), 0, stream, nullptr, stop_ev, 0, g, yy, xx,
                                      (const float*)part, coef, count, ox, op, dgamma, dbeta, rows, C, (int)arpb, fold_rows, (const bf16_t*)x2, part2);
            else
                bn_bwd_apply_fused_kernel<0, false, true><<<grid, 256, 0, stream>>>(g, yy, xx, part, coef, count, ox, op, dgamma, dbeta, rows, C,
                                                                                 (int)arpb, fold_rows, (const bf16_t*)x2, part2);
        } else
        if (relu == 2 && gpre) PPV_APPLY(2, true);
        else if (relu == 2) PPV_APPLY(2, false);
        else if (relu && gpre) PPV_APPLY(1, true);
        else if (relu) PPV_APPLY(1, false);
        else if (gpre) PPV_APPLY(0, true);
        else PPV_APPLY(0, false);
#undef PPV_APPLY
        return ppv_last_error();
    }
    if (x2) return PPV_ERR_BAD_SIZE;                            // the ride-along sums exist in the fused apply only
    bn_bwd_coef_kernel<<<C / 64, 1024, 0, stream>>>(part, 32, count, coef, kc, dgamma, dbeta, C);
    const long n8 = rows * C / 8;
    const unsigned gb = (unsigned)((n8 + 255) / 256);
    if (relu == 2 && gpre) bn_bwd_apply_kernel<2, true><<<gb, 256, 0, stream>>>(g, yy, xx, kc, coef, ox, op, n8, C);
    else if (relu == 2) bn_bwd_apply_kernel<2, false><<<gb, 256, 0, stream>>>(g, yy, xx, kc, coef, ox, op, n8, C);
    else if (relu && gpre) bn_bwd_apply_kernel<1, true><<<gb, 256, 0, stream>>>(g, yy, xx, kc, coef, ox, op, n8, C);
    else if (relu) bn_bwd_apply_kernel<1, false><<<gb, 256, 0, stream>>>(g, yy, xx, kc, coef, ox, op, n8, C);
    else if (gpre) bn_bwd_apply_kernel<0, true><<<gb, 256, 0, stream>>>(g, yy, xx, kc, coef, ox, op, n8, C);
    else bn_bwd_apply_kernel<0, false><<<gb, 256, 0, stream>>>(g, yy, xx, kc, coef, ox, op, n8, C);
    return ppv_last_error();
}

int ppv_bn_bwd(const void* gy, const void* y, const void* x, const float* coef, double count, void* gx, void* gpre,
               float* dgamma, float* dbeta, float* part, float* kc, long rows, int C, int relu, int part_prezeroed,
               hipStream_t stream) {
    return bn_bwd_impl(gy, y, x, coef, count, gx, gpre, dgamma, dbeta, part, kc, rows, C, relu, part_prezeroed, nullptr, nullptr, stream);
}

// ppv_bn_bwd (relu = 0, no g_pre copy) that also takes the backward sums of a SECOND BatchNorm fed by the same gradient (the
// projection shortcut of a down-sampling bottleneck: x2 = its raw conv output) into part2 [8][2][C] (PRE-ZEROED); that BatchNorm's
// ppv_bn_bwd then runs with part_prezeroed = 2.  C in {64, 128} or C % 256 == 0.
int ppv_bn_bwd_sums2(const void* gy, const void* x, const float* coef, double count, void* gx, float* dgamma, float* dbeta,
                     float* part, float* kc, long rows, int C, int part_prezeroed, const void* x2, float* part2,
                     hipStream_t stream) {
    if (!x2 || !part2) return PPV_ERR_NULL;
    return bn_bwd_impl(gy, nullptr, x, coef, count, gx, nullptr, dgamma, dbeta, part, kc, rows, C, 0, part_prezeroed, x2, part2, stream);
}

int ppv_bn_relu_maxpool(const void* x, const float* coef, void* y, void* arg, int B, int H, int W, int C,
                        hipStream_t stream) {
    if (!x || !coef || !y || !arg) return PPV_ERR_NULL;
    if (C % 8 || H % 2 || W % 2) return PPV_ERR_BAD_SIZE;
    const long tot = (long)B * (H / 2) * (W / 2) * (C / 8);
    bn_relu_maxpool_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const bf16_t*)x, coef, (bf16_t*)y,
                                                                             (unsigned char*)arg, B, H, W, C);
    return ppv_last_error();
}

int ppv_maxpool_relu_bwd(const void* gy, const void* y, const void* arg, void* gpre, int B, int H, int W, int C,
                         hipStream_t stream) {
    if (!gy || !y || !arg || !gpre) return PPV_ERR_NULL;
    const long tot = (long)B * H * W * (C / 8);
    maxpool_relu_bwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>((const bf16_t*)gy, (const bf16_t*)y,
                                                                              (const unsigned char*)arg, (bf16_t*)gpre, B, H, W, C);
    return ppv_last_error();
}

// Backward of resnet.1-3 (BatchNorm2d(train) + ReLU + MaxPool 3x3/2) from the POOLED gradient: gy, y [B,H/2,W/2,C] bf16 (pooled
// gradient / output), arg [B,H/2,W/2,C] u8 (ppv_bn_relu_maxpool), x [B,H,W,C] bf16 raw convolution output, coef [4][C] of that
// BatchNorm -> gx [B,H,W,C] bf16 = gradient w.r.t. the raw convolution output; dgamma / dbeta f32 [C] (may be null).
// part: scratch >= 16 * C floats (zeroed here).  Equals ppv_maxpool_relu_bwd + ppv_bn_bwd(relu = 0) without the 268-MB intermediate.
int ppv_maxpool_bn_bwd(const void* gy, const void* y, const void* arg, const void* x, const float* coef, double count, void* gx,
                       float* dgamma, float* dbeta, float* part, int B, int H, int W, int C, hipStream_t stream) {
    if (!gy || !y || !arg || !x || !coef || !gx || !part) return PPV_ERR_NULL;
    if (C % 8 || C > 256 || 256 % (C / 8) || H % 2 || W % 2 || count < 1) return PPV_ERR_BAD_SIZE;
    (void)hipMemsetAsync(part, 0, sizeof(float) * 16 * C, stream);
    const int rpp = 256 / (C / 8), ppw = 8;
    const long npix = (long)B * H * W;
    const unsigned gb = (unsigned)((npix + (long)rpp * ppw - 1) / ((long)rpp * ppw));
    // sums from the pooled tensors alone (PPV_STEM_BWD_SUMS=raw: the pass over the raw tensor, 400 MB instead of 134 MB at B = 128)
    static const bool raw_sums = getenv("PPV_STEM_BWD_SUMS") && !strcmp(getenv("PPV_STEM_BWD_SUMS"), "raw");
    if (raw_sums)
        maxpool_bn_bwd_kernel<false><<<gb, 256, 0, stream>>>((const bf16_t*)gy, (const bf16_t*)y, (const unsigned char*)arg, (const bf16_t*)x, coef,
                                                            count, part, nullptr, nullptr, nullptr, B, H, W, C, ppw);
    else {
        const long npool = npix / 4;
        const unsigned gp = (unsigned)((npool + (long)rpp * ppw - 1) / ((long)rpp * ppw));
        pooled_bn_sums_kernel<<<gp, 256, 0, stream>>>((const bf16_t*)gy, (const bf16_t*)y, (const unsigned char*)arg, (const bf16_t*)x, coef, part,
                                                     B, H, W, C, ppw);
    }
    maxpool_bn_bwd_kernel<true><<<gb, 256, 0, stream>>>((const bf16_t*)gy, (const bf16_t*)y, (const unsigned char*)arg, (const bf16_t*)x, coef,
                                                       count, part, (bf16_t*)gx, dgamma, dbeta, B, H, W, C, ppw);
    return ppv_last_error();
}

int ppv_adaptive_pool_fwd(const void* x, void* y, int B, int H, int W, int C, int E, int out_f32, hipStream_t stream) {
    if (!x || !y) return PPV_ERR_NULL;
    if (C % 8) return PPV_ERR_BAD_SIZE;
    static const int px = env_int_("PPV_POOL_PX", 1);           // A/B: 0 = the flat-index kernels of rounds 1-5
    if (px && C % 4 == 0 && B <= 65535 && E <= 65535 && H <= E && W <= E) {          // (windows of one or two pixels per axis)
        if (out_f32) adaptive_pool_fwd_row_kernel<float><<<dim3(E, B), 256, 0, stream>>>((const bf16_t*)x, (float*)y, H, W, C, E);
        else adaptive_pool_fwd_row_kernel<__bf16><<<dim3(E, B), 256, 0, stream>>>((const bf16_t*)x, (__bf16*)y, H, W, C, E);
        return ppv_last_error();
    }
    const long tot = (long)B * E * E * (C / 8);
    const unsigned gb = (unsigned)((tot + 255) / 256);
    if (out_f32) adaptive_pool_fwd_kernel<float><<<gb, 256, 0, stream>>>((const bf16_t*)x, (float*)y, B, H, W, C, E);
    else adaptive_pool_fwd_kernel<__bf16><<<gb, 256, 0, stream>>>((const bf16_t*)x, (__bf16*)y, B, H, W, C, E);
    return ppv_last_error();
}

// mask_src (bf16 [B][H][W][C], may be null): the pooled tensor itself; lanes where it is <= 0 get a zero gradient.
int ppv_adaptive_pool_bwd(const void* gy, void* gx, const void* mask_src, int B, int H, int W, int C, int E, int g_f32,
                          hipStream_t stream) {
    if (!gy || !gx) return PPV_ERR_NULL;
    static const int px = env_int_("PPV_POOL_PX", 1);
    // (E <= 4 * min(H, W) + ...: at most eight windows per axis cover one input pixel)
    if (px && C % 4 == 0 && B <= 65535 && H <= 65535 && E <= 6 * H && E <= 6 * W) {
        if (g_f32) adaptive_pool_bwd_px_kernel<float><<<dim3(W, H, B), 256, 0, stream>>>((const float*)gy, (bf16_t*)gx, (const bf16_t*)mask_src, H, W, C, E);
        else adaptive_pool_bwd_px_kernel<__bf16><<<dim3(W, H, B), 256, 0, stream>>>((const __bf16*)gy, (bf16_t*)gx, (const bf16_t*)mask_src, H, W, C, E);
        return ppv_last_error();
    }
    const long tot = (long)B * H * W * (C / 8);
    const unsigned gb = (unsigned)((tot + 255) / 256);
    if (g_f32) adaptive_pool_bwd_kernel<float><<<gb, 256, 0, stream>>>((const float*)gy, (bf16_t*)gx, (const bf16_t*)mask_src, B, H, W, C, E);
    else adaptive_pool_bwd_kernel<__bf16><<<gb, 256, 0, stream>>>((const __bf16*)gy, (bf16_t*)gx, (const bf16_t*)mask_src, B, H, W, C, E);
    return ppv_last_error();
}

}  // extern "C"
