// Soft-attention LSTM caption decoder (Image_Caption/models.py:57-218): the per-time-step kernels.
//
// What is hoisted out of the time loop (result-identical, see decoder.py): att1 = encoder_att(encoder_out) is computed
// ONCE with the bf16 MFMA GEMM (ppv_conv_gemm as a 1x1 convolution) instead of once per step (models.py:83 via :207).
// Inside the loop every step is HBM-bound streaming over two per-image tables,
//     att1 [B][P][A] bf16   (scores)   and   encs [B][P][E] bf16   (context),
// so the kernels below are organised around whole-row 16-byte loads, one wave per pixel row, wave-shuffle reductions,
// and a grid of (images x pixel slabs) or (images x channel slabs) that fills 256 CUs at B = 128:
//
//   dec_prepare      f32 encoder_out --gather by sort order--> bf16 encs, + per-image mean (init_h / init_c input)
//   dec_score_fwd    e[b,p] = w_full . relu(att1[b,p,:] + att2[b,:])                      (models.py:85)
//   dec_ctx_fwd      alpha = softmax_p(e); awe = sum_p alpha_p encs[b,p,:]; x = sigmoid(f_beta(h)) * awe  (:86-87, :205-206)
//   lstm_fwd/bwd     the LSTMCell pointwise part (gate order i, f, g, o)                    (:207-210)
//   dec_ctx_bwd      d awe, d gate, d alpha_p = d awe . encs[b,p,:]
//   dec_score_bwd    softmax backward, relu mask recomputed from att1 + att2, f32 accumulation into d att1
//   dec_combine      d encoder_out[order[b]] = (context + score paths) + d mean / P
//
// The dense GEMMs of a step (h projections, LSTM gates, vocabulary scores) are plain library GEMMs on the host side.
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

typedef unsigned short bf16_t;

__device__ __forceinline__ float bflo(unsigned x) { return __builtin_bit_cast(float, x << 16); }
__device__ __forceinline__ float bfhi(unsigned x) { return __builtin_bit_cast(float, x & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }
// Wave-wide reductions on the DPP path: quad_perm xor 1 / xor 2, row_half_mirror, row_mirror leave every lane of a 16-lane row with
// the row's total, four v_readlane fold the rows.  (Written with __shfl_xor the six steps compile to ds_bpermute_b32: six DEPENDENT
// LDS round trips, ~0.7 us per reduction -- the compact score kernel does one per class, 28 per wave: 22 of its 29 us, round 5.)
#define PPV_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
    v += PPV_DPP(v, 0xB1);                                     // quad_perm [1,0,3,2]
    v += PPV_DPP(v, 0x4E);                                     // quad_perm [2,3,0,1]
    v += PPV_DPP(v, 0x141);                                    // row_half_mirror
    v += PPV_DPP(v, 0x140);                                    // row_mirror
    const int i = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, PPV_DPP(v, 0xB1));
    v = fmaxf(v, PPV_DPP(v, 0x4E));
    v = fmaxf(v, PPV_DPP(v, 0x141));
    v = fmaxf(v, PPV_DPP(v, 0x140));
    const int i = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48))));
}
// block-wide (256 threads) reductions through 4 LDS words
__device__ __forceinline__ float block_sum(float v, float* s4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}
__device__ __forceinline__ float block_max(float v, float* s4) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
}

// ----------------------------------------------------------------------------- prepare
// grid (B, ceil(P / PT)); encs[b][p][:] = bf16(enc[order[b]][p][:]); mean[b][:] += sum_p / P  (mean pre-zeroed)
__global__ __launch_bounds__(256) void dec_prepare_kernel(const float* __restrict__ enc, const long* __restrict__ order,
                                                          bf16_t* __restrict__ encs, float* __restrict__ mean, int P, int E,
                                                          int PT) {
    const int b = blockIdx.x, p0 = blockIdx.y * PT, p1 = min(p0 + PT, P);
    const float* src = enc + order[b] * (long)P * E;
    bf16_t* dst = encs + (long)b * P * E;
    const float invP = 1.f / (float)P;
    for (int c = threadIdx.x * 8; c < E; c += 2048) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = p0; p < p1; ++p) {
            const float4 u = *reinterpret_cast<const float4*>(src + (long)p * E + c);
            const float4 v = *reinterpret_cast<const float4*>(src + (long)p * E + c + 4);
            s[0] += u.x; s[1] += u.y; s[2] += u.z; s[3] += u.w; s[4] += v.x; s[5] += v.y; s[6] += v.z; s[7] += v.w;
            uint4 o;
            o.x = pack_bf2(u.x, u.y); o.y = pack_bf2(u.z, u.w); o.z = pack_bf2(v.x, v.y); o.w = pack_bf2(v.z, v.w);
            *reinterpret_cast<uint4*>(dst + (long)p * E + c) = o;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&mean[(long)b * E + c + k], s[k] * invP);
    }
}

// ----------------------------------------------------------------------------- attention scores
// grid (bt, ceil(P / PS)), 4 waves, one wave per pixel row.  hproj row b: [att2 (A) | gate_pre (E)], stride ldh.
__global__ __launch_bounds__(256) void dec_score_fwd_kernel(const bf16_t* __restrict__ att1, const float* __restrict__ hproj,
                                                            int ldh, const float* __restrict__ wfull,
                                                            float* __restrict__ ebuf, int P, int A, int PS) {
    extern __shared__ float sm[];
    float* sA2 = sm;
    float* sW = sm + A;
    const int b = blockIdx.x, p0 = blockIdx.y * PS, p1 = min(p0 + PS, P);
    for (int a = threadIdx.x; a < A; a += 256) {
        sA2[a] = hproj[(long)b * ldh + a];
        sW[a] = wfull[a];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // four pixel rows per wave per trip: their loads are issued together (4 x 1 KB in flight per wave)
    for (int pb = p0 + wave * 4; pb < p1; pb += 16) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int a0 = lane * 8; a0 < A; a0 += 512) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = min(pb + j, p1 - 1);
                v[j] = *reinterpret_cast<const uint4*>(att1 + ((long)b * P + p) * A + a0);
            }
            float a2[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { a2[k] = sA2[a0 + k]; w[k] = sW[a0 + k]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[j] += fmaxf(bflo(w4[k]) + a2[2 * k], 0.f) * w[2 * k];
                    acc[j] += fmaxf(bfhi(w4[k]) + a2[2 * k + 1], 0.f) * w[2 * k + 1];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = wave_sum(acc[j]);
            if (lane == 0 && pb + j < p1) ebuf[(long)b * P + pb + j] = r;
        }
    }
}

// ----------------------------------------------------------------------------- softmax + context + gate
// grid (bt, ceil(E / 256)): every workgroup redoes the (cheap) softmax of its image, then owns 256 context channels.
// alpha_out [bt][P] (written by channel slab 0), awe_save [bt][E] (ungated), xh[b*ldx + x_off + e] = gate * awe.
// SOFTMAX = false: ebuf already holds the P mixing coefficients of the image (compact path: P = cells, coefficients = beta).
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void dec_ctx_fwd_kernel(const bf16_t* __restrict__ encs, const float* __restrict__ ebuf,
                                                          const float* __restrict__ hproj, int ldh, int gate_off,
                                                          float* __restrict__ alpha_out, float* __restrict__ awe_save,
                                                          float* __restrict__ xh, int ldx, int x_off, int P, int E) {
    extern __shared__ float sm[];
    float* sAl = sm;                 // [P]
    float* sRed = sm + P;            // [8][256]
    __shared__ float s4[4];
    const int b = blockIdx.x, c0 = blockIdx.y * 256, tid = threadIdx.x;
    if (SOFTMAX) {
        float m = -3.4e38f;
        for (int p = tid; p < P; p += 256) m = fmaxf(m, ebuf[(long)b * P + p]);
        m = block_max(m, s4);
        float s = 0.f;
        for (int p = tid; p < P; p += 256) {
            const float v = __expf(ebuf[(long)b * P + p] - m);
            sAl[p] = v;
            s += v;
        }
        s = block_sum(s, s4);
        const float inv = 1.f / s;
        for (int p = tid; p < P; p += 256) {
            const float a = sAl[p] * inv;
            sAl[p] = a;
            if (blockIdx.y == 0) alpha_out[(long)b * P + p] = a;
        }
    } else {
        for (int p = tid; p < P; p += 256) sAl[p] = ebuf[(long)b * P + p];
    }
    __syncthreads();
    const int cg = tid & 31, pg = tid >> 5, c = c0 + cg * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < E) {
        const bf16_t* base = encs + (long)b * P * E + c;
        for (int p = pg; p < P; p += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + (long)p * E);
            const float a = sAl[p];
            acc[0] += a * bflo(v.x); acc[1] += a * bfhi(v.x); acc[2] += a * bflo(v.y); acc[3] += a * bfhi(v.y);
            acc[4] += a * bflo(v.z); acc[5] += a * bfhi(v.z); acc[6] += a * bflo(v.w); acc[7] += a * bfhi(v.w);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sRed[pg * 256 + cg * 8 + k] = acc[k];
    __syncthreads();
    const int e = c0 + tid;
    if (e < E) {
        float awe = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) awe += sRed[g * 256 + tid];
        awe_save[(long)b * E + e] = awe;
        xh[(long)b * ldx + x_off + e] = sigmoidf(hproj[(long)b * ldh + gate_off + e]) * awe;
    }
}

// ----------------------------------------------------------------------------- LSTM cell, pointwise part
// z [bt][4D] = W_ih x + b_ih + W_hh h + b_hh (i, f, g, o).  gates [bt][4D] keeps the ACTIVATED values for backward.
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* __restrict__ z, const float* __restrict__ c_prev,
                                                       float* __restrict__ gates, float* __restrict__ c_new,
                                                       float* __restrict__ h_a, int ld_a, float* __restrict__ h_b, int ld_b,
                                                       int bt, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bt * D) return;
    const int b = (int)(i / D), d = (int)(i % D);
    const float* zr = z + (long)b * 4 * D;
    const float gi = sigmoidf(zr[d]), gf = sigmoidf(zr[D + d]), gg = tanhf(zr[2 * D + d]), go = sigmoidf(zr[3 * D + d]);
    const float c = gf * c_prev[i] + gi * gg;
    const float h = go * tanhf(c);
    float* gr = gates + (long)b * 4 * D;
    gr[d] = gi; gr[D + d] = gf; gr[2 * D + d] = gg; gr[3 * D + d] = go;
    c_new[i] = c;
    h_a[(long)b * ld_a + d] = h;
    if (h_b) h_b[(long)b * ld_b + d] = h;
}

// dh [bt][D] (from the vocabulary scores + the next step), dc_in [bt][D] (from the next step) -> dz [bt][4D], dc_prev.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                       const float* __restrict__ c_new, const float* __restrict__ dh,
                                                       const float* __restrict__ dc_in, float* __restrict__ dz,
                                                       float* __restrict__ dc_prev, int bt, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)bt * D) return;
    const int b = (int)(i / D), d = (int)(i % D);
    const float* gr = gates + (long)b * 4 * D;
    const float gi = gr[d], gf = gr[D + d], gg = gr[2 * D + d], go = gr[3 * D + d];
    const float tc = tanhf(c_new[i]);
    const float dhv = dh[i];
    const float dc = dc_in[i] + dhv * go * (1.f - tc * tc);
    float* dr = dz + (long)b * 4 * D;
    dr[d] = dc * gg * gi * (1.f - gi);
    dr[D + d] = dc * c_prev[i] * gf * (1.f - gf);
    dr[2 * D + d] = dc * gi * (1.f - gg * gg);
    dr[3 * D + d] = dhv * tc * go * (1.f - go);
    dc_prev[i] = dc * gf;
}

// ----------------------------------------------------------------------------- context backward
// grid (bt, ceil(P / PS)).  dxh row b holds d(gated awe) at x_off.  Slab 0 also writes d gate_pre into dhproj and the
// ungated d awe (kept for the batched alpha^T . d awe GEMM after the time loop).
// dalpha[b][p] = d awe . encs[b][p][:] (+ the caller's gradient on the returned alphas)
__global__ __launch_bounds__(256) void dec_ctx_bwd_kernel(const bf16_t* __restrict__ encs, const float* __restrict__ dxh, int ldx,
                                                          int x_off, const float* __restrict__ hproj, int ldh, int gate_off,
                                                          const float* __restrict__ awe_save, const float* __restrict__ dalpha_in,
                                                          float* __restrict__ dhproj, float* __restrict__ dawe_out,
                                                          float* __restrict__ dalpha, int P, int E, int PS) {
    extern __shared__ float sD[];    // [E] d awe
    const int b = blockIdx.x, p0 = blockIdx.y * PS, p1 = min(p0 + PS, P);
    for (int e = threadIdx.x; e < E; e += 256) {
        const float g = sigmoidf(hproj[(long)b * ldh + gate_off + e]);
        const float dx = dxh[(long)b * ldx + x_off + e];
        const float da = dx * g;
        sD[e] = da;
        if (blockIdx.y == 0) {
            dhproj[(long)b * ldh + gate_off + e] = dx * awe_save[(long)b * E + e] * g * (1.f - g);
            dawe_out[(long)b * E + e] = da;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // four rows per wave at a time: their loads are in flight together (one row after the other was one L2 round trip per row and
    // 512-column piece: 33 us for the 64 cells of the compact path)
    for (int pb = p0 + wave; pb < p1; pb += 16) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e0 = lane * 8; e0 < E; e0 += 512) {
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = min(pb + 4 * i, p1 - 1);                         // (a clamped row is computed and dropped)
                v[i] = *reinterpret_cast<const uint4*>(encs + ((long)b * P + p) * E + e0);
            }
            const float* d = sD + e0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] += bflo(v[i].x) * d[0] + bfhi(v[i].x) * d[1] + bflo(v[i].y) * d[2] + bfhi(v[i].y) * d[3] + bflo(v[i].z) * d[4] +
                          bfhi(v[i].z) * d[5] + bflo(v[i].w) * d[6] + bfhi(v[i].w) * d[7];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = pb + 4 * i;
            const float a = wave_sum(acc[i]);
            if (lane == 0 && p < p1) dalpha[(long)b * P + p] = a + (dalpha_in ? dalpha_in[(long)b * P + p] : 0.f);
        }
    }
}

// ----------------------------------------------------------------------------- score backward
// grid (bt, ceil(P / PS)).  d e_p = alpha_p (d alpha_p - sum_q alpha_q d alpha_q);  pre = att1 + att2;
// d att1[b][p][a] += [pre > 0] d e_p w_a  (f32 read-modify-write);  d att2[b][a] += same summed over p (atomics into
// dhproj[:, 0:A], pre-zeroed);  d w_full[b][a] += d e_p relu(pre)  (atomics, per-image rows).  A <= 2048.
__global__ __launch_bounds__(256) void dec_score_bwd_kernel(const bf16_t* __restrict__ att1, const float* __restrict__ hproj, int ldh,
                                                            const float* __restrict__ wfull, const float* __restrict__ alpha,
                                                            const float* __restrict__ dalpha, float* __restrict__ datt1,
                                                            float* __restrict__ dhproj, float* __restrict__ dwfull, int P, int A,
                                                            int PS) {
    extern __shared__ float sm[];
    float* sA2 = sm;                 // [A]
    float* sW = sm + A;              // [A]
    float* sAcc = sm + 2 * A;        // [2][A] cross-wave accumulators (d att2, d w)
    __shared__ float s4[4];
    const int b = blockIdx.x, p0 = blockIdx.y * PS, p1 = min(p0 + PS, P), tid = threadIdx.x;
    for (int a = tid; a < A; a += 256) {
        sA2[a] = hproj[(long)b * ldh + a];
        sW[a] = wfull[a];
        sAcc[a] = 0.f;
        sAcc[A + a] = 0.f;
    }
    float dot = 0.f;
    for (int p = tid; p < P; p += 256) dot += alpha[(long)b * P + p] * dalpha[(long)b * P + p];
    dot = block_sum(dot, s4);        // (its barriers also publish sA2 / sW / sAcc)
    const int wave = tid >> 6, lane = tid & 63;
    float da2[4][8], dw[4][8];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int k = 0; k < 8; ++k) da2[ch][k] = dw[ch][k] = 0.f;
    for (int p = p0 + wave; p < p1; p += 4) {
        const float de = alpha[(long)b * P + p] * (dalpha[(long)b * P + p] - dot);
        const bf16_t* row = att1 + ((long)b * P + p) * A;
        float* drow = datt1 + ((long)b * P + p) * A;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int a0 = ch * 512 + lane * 8;
            if (a0 < A) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + a0);
                const float x[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
                float4 g0 = *reinterpret_cast<const float4*>(drow + a0), g1 = *reinterpret_cast<const float4*>(drow + a0 + 4);
                float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float pre = x[k] + sA2[a0 + k];
                    const float dp = pre > 0.f ? de * sW[a0 + k] : 0.f;
                    dw[ch][k] += de * fmaxf(pre, 0.f);
                    da2[ch][k] += dp;
                    g[k] += dp;
                }
                *reinterpret_cast<float4*>(drow + a0) = make_float4(g[0], g[1], g[2], g[3]);
                *reinterpret_cast<float4*>(drow + a0 + 4) = make_float4(g[4], g[5], g[6], g[7]);
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const int a0 = ch * 512 + lane * 8;
        if (a0 < A) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                atomicAdd(&sAcc[a0 + k], da2[ch][k]);
                atomicAdd(&sAcc[A + a0 + k], dw[ch][k]);
            }
        }
    }
    __syncthreads();
    for (int a = tid; a < A; a += 256) {
        atomicAdd(&dhproj[(long)b * ldh + a], sAcc[a]);
        atomicAdd(&dwfull[(long)b * A + a], sAcc[A + a]);        // per-image rows (a single shared row serialises 3456 adders)
    }
}

// ----------------------------------------------------------------------------- encoder gradient assembly
// out[order[b]][p][:] = acc[b][p][:] + dmean[b][:] / P     (un-sorts the batch; float4 granules)
__global__ __launch_bounds__(256) void dec_combine_kernel(const float* __restrict__ acc, const float* __restrict__ dmean,
                                                          const long* __restrict__ order, float* __restrict__ out, int P, int E) {
    const int b = blockIdx.y;
    const long per = (long)P * E / 4, i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const int e4 = (int)(i % (E / 4));
    const float invP = 1.f / (float)P;
    const float4 a = reinterpret_cast<const float4*>(acc + (long)b * P * E)[i];
    const float4 m = reinterpret_cast<const float4*>(dmean + (long)b * E)[e4];
    reinterpret_cast<float4*>(out + order[b] * (long)P * E)[i] =
        make_float4(a.x + m.x * invP, a.y + m.y * invP, a.z + m.z * invP, a.w + m.w * invP);
}

// out[order[b]][p][e] = part[b][p][e] + dmean[b][e] * (gamma ? gamma[p] : 1 / P) + sum_t alpha[t][b][p] * dawe[t][b][e]
// (score path + init_h/init_c path + context path in ONE pass over the 4 * B * P * E bytes; un-sorts the batch).
// grid (B, ceil(P / 48), E / 256); alpha [T][B][P], dawe [T][B][E]; LDS: [T][256] d awe + [T][48] alpha.
// General path: P pixels, gamma null, f32 output.  Compact path (round 5: replaces baddbmm_ + add_ + cast + index_put of decoder.py):
// the "pixels" are the C cells, alpha = beta [T][B][C], gamma [C] = each cell's share of the per-image mean, output in the cell map's
// dtype (OUT_BF16: one rounding of the f32 sum).
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void dec_enc_grad_kernel(const float* __restrict__ part, const float* __restrict__ dmean,
                                                           const float* __restrict__ alpha, const float* __restrict__ dawe,
                                                           const long* __restrict__ order, const float* __restrict__ gamma,
                                                           void* __restrict__ out, int B, int P, int E, int T) {
    extern __shared__ float sm[];
    float* sD = sm;                  // [T][256]
    float* sA = sm + T * 256;        // [T][48]
    const int b = blockIdx.x, p0 = blockIdx.y * 48, np = min(48, P - p0), c0 = blockIdx.z * 256, tid = threadIdx.x;
    for (int i = tid; i < T * 256; i += 256) sD[i] = dawe[((long)(i >> 8) * B + b) * E + c0 + (i & 255)];
    for (int i = tid; i < T * 48; i += 256) {
        const int t = i / 48, j = i % 48;
        sA[i] = j < np ? alpha[((long)t * B + b) * P + p0 + j] : 0.f;
    }
    __syncthreads();
    const int c4 = (tid & 63) * 4, pw = tid >> 6;
    const float invP = 1.f / (float)P;
    const float4 m = *reinterpret_cast<const float4*>(dmean + (long)b * E + c0 + c4);
    const long obase = order[b] * (long)P * E;
    for (int j = pw; j < np; j += 4) {
        const long off = (long)(p0 + j) * E + c0 + c4;
        const float gm = gamma ? gamma[p0 + j] : invP;
        float4 a = *reinterpret_cast<const float4*>(part + (long)b * P * E + off);
        a.x += m.x * gm; a.y += m.y * gm; a.z += m.z * gm; a.w += m.w * gm;
        for (int t = 0; t < T; ++t) {
            const float w = sA[t * 48 + j];
            const float4 d = *reinterpret_cast<const float4*>(sD + t * 256 + c4);
            a.x += w * d.x; a.y += w * d.y; a.z += w * d.z; a.w += w * d.w;
        }
        if constexpr (OUT_BF16) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
            const f32x2_ lo = {a.x, a.y}, hi = {a.z, a.w};
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + obase + off) =
                make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_)),
                           __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_)));
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + obase + off) = a;
        }
    }
}

// mean[b][e] = sum_c gamma[c] * cells[b][c][e]: the per-image mean of the pooled [P, E] tensor (models.py:143-145) taken on the C cells it
// was pooled from (gamma[c] = the share of cell c in that mean).  cells bf16 [B][C][E], one thread per 8 channels.  (Was
// torch.einsum("c,bce->be") on an f32 copy of the cells: a library bmm + a 33-MB conversion.)
__global__ __launch_bounds__(256) void decc_mean_kernel(const bf16_t* __restrict__ cells, const float* __restrict__ gamma,
                                                        float* __restrict__ mean, int B, int C, int E) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int e8 = E / 8;
    if (i >= (long)B * e8) return;
    const int b = (int)(i / e8), e = (int)(i % e8) * 8;
    const bf16_t* p = cells + (long)b * C * E + e;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + (long)c * E);
        const float g = gamma[c];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[2 * k] += g * __builtin_bit_cast(float, w[k] << 16);
            a[2 * k + 1] += g * __builtin_bit_cast(float, w[k] & 0xffff0000u);
        }
    }
    *reinterpret_cast<float4*>(mean + (long)b * E + e) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(mean + (long)b * E + e + 4) = make_float4(a[4], a[5], a[6], a[7]);
}

// ============================================================================= compact attention (pooled-map structure)
// The Encoder's output is AdaptiveAvgPool2d(36) of an 8 x 8 map (models.py:27,39): every one of the P = 1296 "pixels" is the
// mean of 1, 2 or 4 of the C = 64 cells, and only Q = 225 distinct (cell set) classes exist.  encoder_att, the mean and the
// weighted context are linear, so they commute with the pooling:
//     att1[b,p,:] = sum_k w_q att1c[b, cell_k(q), :],   awe = sum_c beta_c feat[b,c,:],   beta_c = sum_{q: c in q} mult_q alpha_q w_q
// with q = class(p).  A step then streams C*(A+E) instead of P*(A+E) values per image (20x fewer bytes), the tables stay
// L2-resident, and the result is the same function of the same numbers (f32 accumulation of bf16 cells).
// Tables (device, built on the host from the pooling geometry): cls_cells [Q][4] (-1 = unused), cls_w [Q] = 1 / #cells,
// cls_mult [Q] = pixels in the class, pix_class [P].
struct ClassTables {
    const int* cells;      // [Q][4]
    const float* w;        // [Q]
    const float* mult;     // [Q]
    const int* pix_class;  // [P]
};

// grid (bt), 512 threads (8 waves: only bt <= 128 workgroups exist, so each one is made as wide as the class loop allows); LDS: att1c[b] (C*A bf16) | att2 [A] | w_full [A] | e / alpha [Q] | beta [C]
__global__ __launch_bounds__(512) void decc_score_fwd_kernel(const bf16_t* __restrict__ att1c, const float* __restrict__ hproj,
                                                             int ldh, const float* __restrict__ wfull, ClassTables tb,
                                                             float* __restrict__ alpha_out, float* __restrict__ alq_out,
                                                             float* __restrict__ beta_out, int P, int Q, int C, int A) {
    extern __shared__ __attribute__((aligned(16))) char smc[];
    bf16_t* sT = reinterpret_cast<bf16_t*>(smc);                          // [C + 1][A]: row C is zero (the "no cell" slot of a class)
    float* sA2 = reinterpret_cast<float*>(smc + (size_t)(C + 1) * A * 2);
    float* sW = sA2 + A;
    float* sE = sW + A;
    float* sBeta = sE + Q;
    float* sWq = sBeta + C;                                               // [Q]   class tables staged once (dependent walks: see the
    int* sCells = reinterpret_cast<int*>(sWq + Q);                        // [Q][4]  backward kernel)
    __shared__ float s8[8];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint4* src = reinterpret_cast<const uint4*>(att1c + (long)b * C * A);
    // eight 16-byte loads in flight per thread (written as `sT[i] = src[i]` the loop compiled to load / wait / store per iteration: eight
    // serial L2 round trips for the 64-KB table, a third of the kernel)
    for (int i0 = tid; i0 < C * A / 8; i0 += 8 * 512) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[min(i0 + k * 512, C * A / 8 - 1)];
        __builtin_amdgcn_sched_barrier(0);                                 // (a clamped index re-writes the last chunk with its own value:
#pragma unroll                                                             //  no branch for the compiler to sink the loads into)
        for (int k = 0; k < 8; ++k) reinterpret_cast<uint4*>(sT)[min(i0 + k * 512, C * A / 8 - 1)] = v[k];
    }
    for (int a = tid; a < A; a += 512) { sA2[a] = hproj[(long)b * ldh + a]; sW[a] = wfull[a]; }
    for (int c = tid; c < C; c += 512) sBeta[c] = 0.f;
    for (int q = tid; q < Q; q += 512) sWq[q] = tb.w[q];
    for (int i = tid; i < 4 * Q; i += 512) { const int c = tb.cells[i]; sCells[i] = c < 0 ? C : c; }
    for (int a = tid; a < A / 2; a += 512) reinterpret_cast<unsigned*>(sT + (size_t)C * A)[a] = 0u;
    __syncthreads();
    // (the four member rows of a class are read UNCONDITIONALLY -- unused slots point at the zero row: behind a branch per slot each
    // read waited for the one before it, four LDS round trips per class and wave)
    // per 512-channel piece: this lane's att2 / w_full values stay in registers, a class = FOUR row reads issued together (as
    // written before -- read, wait, unpack per row, then eight dependent reads of att2 / w_full -- a class cost 12 serial LDS round
    // trips: 0.7 us, 20 of the kernel's 29 us), two classes per iteration
    for (int piece = 0; piece * 512 < A; ++piece) {                        // wave-uniform: the reductions below need every lane
        const bool live = piece * 512 + lane * 8 < A;                     // (A < 512: the lanes past the end carry zero weights)
        const int a0 = live ? piece * 512 + lane * 8 : 0;
        float a2[8], wv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a2[k] = sA2[a0 + k]; wv[k] = live ? sW[a0 + k] : 0.f; }
        for (int q0 = wave; q0 < Q; q0 += 16) {
            float acc[2];
            uint4 v[2][4];
            float wq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = min(q0 + 8 * u, Q - 1);
                wq[u] = sWq[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const uint4*>(sT + (long)sCells[q * 4 + k] * A + a0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pre[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pre[k] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 w = v[u][k];
                    pre[0] += bflo(w.x); pre[1] += bfhi(w.x); pre[2] += bflo(w.y); pre[3] += bfhi(w.y);
                    pre[4] += bflo(w.z); pre[5] += bfhi(w.z); pre[6] += bflo(w.w); pre[7] += bfhi(w.w);
                }
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) a += fmaxf(pre[k] * wq[u] + a2[k], 0.f) * wv[k];
                acc[u] = a;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + 8 * u;
                const float t = wave_sum(acc[u]);
                if (lane == 0 && q < Q) sE[q] = piece ? sE[q] + t : t;
            }
        }
    }
    __syncthreads();
    float m = -3.4e38f;
    for (int q = tid; q < Q; q += 512) m = fmaxf(m, sE[q]);
    m = wave_max(m);
    if (lane == 0) s8[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(fmaxf(s8[0], s8[1]), fmaxf(s8[2], s8[3])), fmaxf(fmaxf(s8[4], s8[5]), fmaxf(s8[6], s8[7])));
    __syncthreads();
    float z = 0.f;
    for (int q = tid; q < Q; q += 512) {
        const float v = __expf(sE[q] - m);
        sE[q] = v;
        z += v * tb.mult[q];
    }
    z = wave_sum(z);
    if (lane == 0) s8[wave] = z;
    __syncthreads();
    z = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    const float inv = 1.f / z;
    for (int q = tid; q < Q; q += 512) {
        const float al = sE[q] * inv;                               // alpha of EVERY pixel of class q
        sE[q] = al;
        alq_out[(long)b * Q + q] = al;
        const float bw = al * tb.mult[q] * sWq[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = sCells[q * 4 + k];
            if (c < C) atomicAdd(&sBeta[c], bw);
        }
    }
    __syncthreads();
    for (int p = tid; p < P; p += 512) alpha_out[(long)b * P + p] = sE[tb.pix_class[p]];
    for (int c = tid; c < C; c += 512) beta_out[(long)b * C + c] = sBeta[c];
}

// grid (bt), 1024 threads: ONE workgroup per image, wave w owns cells w, w + 16, ...  dfb [bt][C] = d awe . feat[b,c,:] (from
// dec_ctx_bwd_kernel run on the cells), galpha [bt][P] or null = the caller's gradient on the returned per-pixel alphas.
// Class-total softmax backward, then GATHER form of the relu / encoder_att backward: a cell sums w_q d pre_q over the <= 12 classes it
// belongs to (cell_cls [C][12], -1 padded) -- no atomics on the big accumulator, one read-modify-write of the cell's d att1c row; d att2 /
// d w_full are taken once per class (by the wave that owns the class's first cell), folded across the waves in LDS and stored once.
// (Round 5: was four 256-thread workgroups per image, each re-loading the image's 64-KB att1c table and redoing the 225-term prelude,
// with 2 x 512 f32 GLOBAL atomics per wave on rows shared by 16 adders: 68 us per step; the atomics were the cost.)
__global__ __launch_bounds__(1024) void decc_score_bwd_kernel(const bf16_t* __restrict__ att1c, const float* __restrict__ hproj,
                                                              int ldh, const float* __restrict__ wfull, ClassTables tb,
                                                              const int* __restrict__ cell_cls, const float* __restrict__ alq,
                                                              const float* __restrict__ dfb, const float* __restrict__ galpha,
                                                              float* __restrict__ datt1c, float* __restrict__ dhproj,
                                                              float* __restrict__ dwfull, int P, int Q, int C, int A) {
    extern __shared__ __attribute__((aligned(16))) char smc[];
    bf16_t* sT = reinterpret_cast<bf16_t*>(smc);                          // [C + 1][A] bf16: row C is zero (the "no cell" slot of a class)
    float* sA2 = reinterpret_cast<float*>(smc + (size_t)(C + 1) * A * 2); // [A]
    float* sW = sA2 + A;                                                  // [A]
    float* sDa2 = sW + A;                                                 // [A] d att2 of this image
    float* sDw = sDa2 + A;                                                // [A] d w_full of this image
    float* sDe = sDw + A;                                                 // [Q] class-total d e
    float* sGa = sDe + Q;                                                 // [Q]
    // the class tables, staged once: the loops below walk them with DEPENDENT indices (cell -> classes -> member cells), which from
    // global memory is one L2 round trip per step of the walk (~50 per wave: most of the kernel's 50 us)
    float* sWq = sGa + Q;                                                 // [Q]
    int* sCells = reinterpret_cast<int*>(sWq + Q);                        // [Q][4]
    int* sCls = sCells + 4 * Q;                                           // [C][12]
    __shared__ float s16[16];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint4* src = reinterpret_cast<const uint4*>(att1c + (long)b * C * A);
    for (int i0 = tid; i0 < C * A / 8; i0 += 4 * 1024) {                   // four loads in flight per thread (see the forward kernel)
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = src[min(i0 + k * 1024, C * A / 8 - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<uint4*>(sT)[min(i0 + k * 1024, C * A / 8 - 1)] = v[k];
    }
    for (int a = tid; a < A; a += 1024) { sA2[a] = hproj[(long)b * ldh + a]; sW[a] = wfull[a]; sDa2[a] = 0.f; sDw[a] = 0.f; }
    for (int q = tid; q < Q; q += 1024) { sGa[q] = 0.f; sWq[q] = tb.w[q]; }
    for (int i = tid; i < 4 * Q; i += 1024) { const int c = tb.cells[i]; sCells[i] = c < 0 ? C : c; }
    for (int i = tid; i < 12 * C; i += 1024) sCls[i] = cell_cls[i];
    for (int a = tid; a < A / 2; a += 1024) reinterpret_cast<unsigned*>(sT + (size_t)C * A)[a] = 0u;
    __syncthreads();
    if (galpha)
        for (int p = tid; p < P; p += 1024) atomicAdd(&sGa[tb.pix_class[p]], galpha[(long)b * P + p]);
    __syncthreads();
    float part = 0.f;
    for (int q = tid; q < Q; q += 1024) {
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = sCells[q * 4 + k];
            if (c < C) d += dfb[(long)b * C + c];
        }
        const float Dq = tb.mult[q] * sWq[q] * d + sGa[q];
        sGa[q] = Dq;
        part += alq[(long)b * Q + q] * Dq;
    }
    part = wave_sum(part);
    if (lane == 0) s16[wave] = part;
    __syncthreads();
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) S += s16[w];
    for (int q = tid; q < Q; q += 1024) sDe[q] = alq[(long)b * Q + q] * (sGa[q] - tb.mult[q] * S);
    __syncthreads();
    for (int a0 = lane * 8; a0 < A; a0 += 512) {
        float a2[8], wv[8], da2[8], dw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a2[k] = sA2[a0 + k]; wv[k] = sW[a0 + k]; da2[k] = dw[k] = 0.f; }
        for (int c = wave; c < C; c += 16) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float4* g = reinterpret_cast<float4*>(datt1c + ((long)b * C + c) * A + a0);
            float4 v0 = g[0], v1 = g[1];                                   // the row this cell accumulates into: requested before the class walk
            for (int j = 0; j < 12; ++j) {
                const int q = sCls[c * 12 + j];
                if (q < 0) break;                                          // wave-uniform
                const float wq = sWq[q], de = sDe[q];
                const bool first = sCells[q * 4] == c;                     // this wave accounts the class for d att2 / d w
                float pre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                uint4 v4[4];                                                // the four member rows in flight together (unused slots
#pragma unroll                                                              // read the zero row: no branch, no wait between the reads)
                for (int k = 0; k < 4; ++k) v4[k] = *reinterpret_cast<const uint4*>(sT + (long)sCells[q * 4 + k] * A + a0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 v = v4[k];
                    pre[0] += bflo(v.x); pre[1] += bfhi(v.x); pre[2] += bflo(v.y); pre[3] += bfhi(v.y);
                    pre[4] += bflo(v.z); pre[5] += bfhi(v.z); pre[6] += bflo(v.w); pre[7] += bfhi(v.w);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float pr = pre[k] * wq + a2[k];
                    const float dp = pr > 0.f ? de * wv[k] : 0.f;
                    acc[k] += wq * dp;
                    if (first) { da2[k] += dp; dw[k] += de * fmaxf(pr, 0.f); }
                }
            }
            v0.x += acc[0]; v0.y += acc[1]; v0.z += acc[2]; v0.w += acc[3];
            v1.x += acc[4]; v1.y += acc[5]; v1.z += acc[6]; v1.w += acc[7];
            g[0] = v0; g[1] = v1;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                      // 16 waves per address, LDS atomics
            atomicAdd(&sDa2[a0 + k], da2[k]);
            atomicAdd(&sDw[a0 + k], dw[k]);
        }
    }
    __syncthreads();
    for (int a = tid; a < A; a += 1024) {                                  // the only workgroup of this image: plain read-modify-write
        dhproj[(long)b * ldh + a] += sDa2[a];
        dwfull[(long)b * A + a] += sDw[a];
    }
}

}  // namespace ppv

using namespace ppv;

extern "C" {

static inline int dec_slab(int P) { return P <= 64 ? 16 : 48; }      // pixels (cells) per workgroup of the context kernels: four rows per wave

// models.py:181-183,151 : gather by the length sort, bf16 copy and per-image mean.  mean [B][E] PRE-ZEROED.  E % 8 == 0.
int ppv_dec_prepare(const float* enc, const long* order, void* encs, float* mean, int B, int P, int E, hipStream_t stream) {
    if (!enc || !order || !encs || !mean) return PPV_ERR_NULL;
    if (B < 1 || P < 1 || E < 8 || E % 8) return PPV_ERR_BAD_SIZE;
    const int PT = 16;
    dec_prepare_kernel<<<dim3(B, (P + PT - 1) / PT), 256, 0, stream>>>(enc, order, (bf16_t*)encs, mean, P, E, PT);
    return ppv_last_error();
}

// models.py:83-90,205-206 for the first bt (sorted) images of one time step.
//   att1 [B][P][A] bf16 (encoder_att WITHOUT bias), hproj [bt][ldh] f32 = [att2 + both biases (A) | f_beta(h) (E)],
//   wfull [A]; ebuf [bt][P] scratch; alpha_out [bt][P]; awe_save [bt][E]; xh row stride ldx, gated context at x_off.
int ppv_dec_attend_fwd(const void* att1, const void* encs, const float* hproj, int ldh, const float* wfull, float* ebuf,
                       float* alpha_out, float* awe_save, float* xh, int ldx, int x_off, int bt, int P, int A, int E,
                       hipStream_t stream) {
    if (!att1 || !encs || !hproj || !wfull || !ebuf || !alpha_out || !awe_save || !xh) return PPV_ERR_NULL;
    if (bt < 1 || A % 8 || E % 8 || A > 2048 || P > 8192 || ldh < A + E) return PPV_ERR_BAD_SIZE;
    const int PS = dec_slab(P);
    dec_score_fwd_kernel<<<dim3(bt, (P + PS - 1) / PS), 256, 2 * A * sizeof(float), stream>>>((const bf16_t*)att1, hproj, ldh, wfull,
                                                                                             ebuf, P, A, PS);
    dec_ctx_fwd_kernel<true><<<dim3(bt, (E + 255) / 256), 256, (P + 8 * 256) * sizeof(float), stream>>>(
        (const bf16_t*)encs, ebuf, hproj, ldh, A, alpha_out, awe_save, xh, ldx, x_off, P, E);
    return ppv_last_error();
}

// Adjoint of ppv_dec_attend_fwd for one step.  dxh: gradient of the LSTM input rows (gated context at x_off);
// dalpha_in [bt][P] or null; dhproj [bt][ldh] with its first A columns PRE-ZEROED (att2 part is accumulated, gate part
// written); dawe_out [bt][E]; dalpha [bt][P] scratch; datt1 [B][P][A] f32 ACCUMULATED; dwfull [B][A] per-image rows ACCUMULATED.
int ppv_dec_attend_bwd(const void* att1, const void* encs, const float* hproj, int ldh, const float* wfull, const float* alpha,
                       const float* awe_save, const float* dxh, int ldx, int x_off, const float* dalpha_in, float* dhproj,
                       float* dawe_out, float* dalpha, float* datt1, float* dwfull, int bt, int P, int A, int E,
                       hipStream_t stream) {
    if (!att1 || !encs || !hproj || !wfull || !alpha || !awe_save || !dxh || !dhproj || !dawe_out || !dalpha || !datt1 || !dwfull)
        return PPV_ERR_NULL;
    if (bt < 1 || A % 8 || E % 8 || A > 2048 || P > 8192 || ldh < A + E) return PPV_ERR_BAD_SIZE;
    const int PS = dec_slab(P);
    dec_ctx_bwd_kernel<<<dim3(bt, (P + PS - 1) / PS), 256, E * sizeof(float), stream>>>(
        (const bf16_t*)encs, dxh, ldx, x_off, hproj, ldh, A, awe_save, dalpha_in, dhproj, dawe_out, dalpha, P, E, PS);
    dec_score_bwd_kernel<<<dim3(bt, (P + PS - 1) / PS), 256, 4 * A * sizeof(float), stream>>>(
        (const bf16_t*)att1, hproj, ldh, wfull, alpha, dalpha, datt1, dhproj, dwfull, P, A, PS);
    return ppv_last_error();
}

// torch.nn.LSTMCell pointwise part (models.py:207-210).  h is written to h_a (row stride ld_a) and, if non-null, h_b.
int ppv_lstm_cell_fwd(const float* z, const float* c_prev, float* gates, float* c_new, float* h_a, int ld_a, float* h_b, int ld_b,
                      int bt, int D, hipStream_t stream) {
    if (!z || !c_prev || !gates || !c_new || !h_a) return PPV_ERR_NULL;
    if (bt < 1 || D < 1) return PPV_ERR_BAD_SIZE;
    lstm_fwd_kernel<<<(unsigned)(((long)bt * D + 255) / 256), 256, 0, stream>>>(z, c_prev, gates, c_new, h_a, ld_a, h_b, ld_b, bt, D);
    return ppv_last_error();
}

int ppv_lstm_cell_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh, const float* dc_in, float* dz,
                      float* dc_prev, int bt, int D, hipStream_t stream) {
    if (!gates || !c_prev || !c_new || !dh || !dc_in || !dz || !dc_prev) return PPV_ERR_NULL;
    if (bt < 1 || D < 1) return PPV_ERR_BAD_SIZE;
    lstm_bwd_kernel<<<(unsigned)(((long)bt * D + 255) / 256), 256, 0, stream>>>(gates, c_prev, c_new, dh, dc_in, dz, dc_prev, bt, D);
    return ppv_last_error();
}

// d encoder_out [B][P][E] f32 in the caller's (unsorted) batch order.  E % 4 == 0.
int ppv_dec_combine(const float* acc, const float* dmean, const long* order, float* out, int B, int P, int E, hipStream_t stream) {
    if (!acc || !dmean || !order || !out) return PPV_ERR_NULL;
    if (B < 1 || E % 4) return PPV_ERR_BAD_SIZE;
    const long per = (long)P * E / 4;
    dec_combine_kernel<<<dim3((unsigned)((per + 255) / 256), B), 256, 0, stream>>>(acc, dmean, order, out, P, E);
    return ppv_last_error();
}

// d encoder_out in one pass (see dec_enc_grad_kernel).  part [B][P][E] f32 = score-path gradient (sorted order),
// alpha [T][B][P], dawe [T][B][E] (rows of finished captions zero).  E % 256 == 0, T <= 128.
static int dec_enc_grad_launch(const float* part, const float* dmean, const float* alpha, const float* dawe, const long* order,
                               const float* gamma, void* out, int out_bf16, int B, int P, int E, int T, hipStream_t stream) {
    if (!part || !dmean || !alpha || !dawe || !order || !out) return PPV_ERR_NULL;
    if (B < 1 || P < 1 || E % 256 || T < 1 || T > 128) return PPV_ERR_BAD_SIZE;
    const size_t lds = (size_t)T * (256 + 48) * sizeof(float);
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)dec_enc_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (256 + 48) * 4));
        PPV_ATTR(hipFuncSetAttribute((const void*)dec_enc_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (256 + 48) * 4));
        attr_once.done();
    }
    const dim3 grid(B, (P + 47) / 48, E / 256);
    if (out_bf16) dec_enc_grad_kernel<true><<<grid, 256, lds, stream>>>(part, dmean, alpha, dawe, order, gamma, out, B, P, E, T);
    else dec_enc_grad_kernel<false><<<grid, 256, lds, stream>>>(part, dmean, alpha, dawe, order, gamma, out, B, P, E, T);
    return ppv_last_error();
}

int ppv_dec_enc_grad(const float* part, const float* dmean, const float* alpha, const float* dawe, const long* order, float* out,
                     int B, int P, int E, int T, hipStream_t stream) {
    return dec_enc_grad_launch(part, dmean, alpha, dawe, order, nullptr, out, 0, B, P, E, T, stream);
}

// mean[b][e] = sum_c gamma[c] * cells[b][c][e] (cells bf16 [B][C][E], mean f32 [B][E]); E % 8 == 0.
int ppv_decc_mean(const void* cells, const float* gamma, float* mean, int B, int C, int E, hipStream_t stream) {
    if (!cells || !gamma || !mean) return PPV_ERR_NULL;
    if (B < 1 || C < 1 || E < 8 || E % 8) return PPV_ERR_BAD_SIZE;
    const long n = (long)B * (E / 8);
    decc_mean_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const bf16_t*)cells, gamma, mean, B, C, E);
    return ppv_last_error();
}

// The compact path's form: gradient with respect to the CELL map [B][C][E] (out_bf16: bf16, else f32), un-sorted:
//   out[order[b]][c][e] = part[b][c][e] + gamma[c] * dmean[b][e] + sum_t beta[t][b][c] * dawe[t][b][e]
// beta [T][B][C] = the per-cell attention weights of every step (rows of finished captions zero), gamma [C] = a cell's share of the
// per-image mean of the pooled tensor.  E % 256 == 0, T <= 128.
int ppv_decc_enc_grad(const float* part, const float* dmean, const float* beta, const float* dawe, const long* order, const float* gamma,
                      void* out, int out_bf16, int B, int C, int E, int T, hipStream_t stream) {
    if (!gamma) return PPV_ERR_NULL;
    return dec_enc_grad_launch(part, dmean, beta, dawe, order, gamma, out, out_bf16, B, C, E, T, stream);
}

// Compact-attention step (see "compact attention" above) for the first bt sorted images.  att1c [B][C][A] bf16 (encoder_att of
// the CELLS, no bias), feat [B][C][E] bf16, tables on the device; writes alpha_out [bt][P] (per pixel, as the reference
// returns it), alq_out [bt][Q], beta_out [bt][C], awe_save [bt][E] and the gated context into xh.  C*A*2 + (2A+Q+C)*4 <= 150 KB.
int ppv_decc_attend_fwd(const void* att1c, const void* feat, const float* hproj, int ldh, const float* wfull, const int* cls_cells,
                        const float* cls_w, const float* cls_mult, const int* pix_class, float* alpha_out, float* alq_out,
                        float* beta_out, float* awe_save, float* xh, int ldx, int x_off, int bt, int P, int Q, int C, int A, int E,
                        hipStream_t stream) {
    if (!att1c || !feat || !hproj || !wfull || !cls_cells || !cls_w || !cls_mult || !pix_class || !alpha_out || !alq_out ||
        !beta_out || !awe_save || !xh)
        return PPV_ERR_NULL;
    const size_t lds = (size_t)(C + 1) * A * 2 + (size_t)(2 * A + 6 * Q + C) * 4;
    if (bt < 1 || A % 8 || E % 8 || A > 2048 || lds > 150 * 1024 || ldh < A + E) return PPV_ERR_BAD_SIZE;
    static bool attr = false;
    if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)decc_score_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
    ClassTables tb{cls_cells, cls_w, cls_mult, pix_class};
    decc_score_fwd_kernel<<<bt, 512, lds, stream>>>((const bf16_t*)att1c, hproj, ldh, wfull, tb, alpha_out, alq_out, beta_out, P, Q, C, A);
    dec_ctx_fwd_kernel<false><<<dim3(bt, (E + 255) / 256), 256, (C + 8 * 256) * sizeof(float), stream>>>(
        (const bf16_t*)feat, beta_out, hproj, ldh, A, nullptr, awe_save, xh, ldx, x_off, C, E);
    return ppv_last_error();
}

// Adjoint of ppv_decc_attend_fwd.  cell_cls [C][12] (classes a cell belongs to, -1 padded); dfb [bt][C] scratch; datt1c
// [B][C][A] f32 ACCUMULATED; dhproj first A columns ACCUMULATED (dec_ctx_bwd writes the rest); dwfull [B][A] PER-IMAGE rows ACCUMULATED (sum over b afterwards);
// dawe_out [bt][E] (kept for the batched
// beta^T . d awe GEMM); galpha [bt][P] or null.  C*A*2 + (2A+2Q)*4 <= 150 KB.
int ppv_decc_attend_bwd(const void* att1c, const void* feat, const float* hproj, int ldh, const float* wfull, const int* cls_cells,
                        const float* cls_w, const float* cls_mult, const int* pix_class, const int* cell_cls, const float* alq,
                        const float* awe_save, const float* dxh, int ldx, int x_off, const float* galpha, float* dhproj,
                        float* dawe_out, float* dfb, float* datt1c, float* dwfull, int bt, int P, int Q, int C, int A, int E,
                        hipStream_t stream) {
    if (!att1c || !feat || !hproj || !wfull || !cls_cells || !cls_w || !cls_mult || !pix_class || !cell_cls || !alq || !awe_save ||
        !dxh || !dhproj || !dawe_out || !dfb || !datt1c || !dwfull)
        return PPV_ERR_NULL;
    const size_t lds = (size_t)(C + 1) * A * 2 + (size_t)(4 * A + 7 * Q + 12 * C) * 4;
    if (bt < 1 || A % 8 || E % 8 || lds > 150 * 1024 || ldh < A + E) return PPV_ERR_BAD_SIZE;
    static bool attr = false;
    if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)decc_score_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
    const int PS = dec_slab(C);
    dec_ctx_bwd_kernel<<<dim3(bt, (C + PS - 1) / PS), 256, E * sizeof(float), stream>>>(
        (const bf16_t*)feat, dxh, ldx, x_off, hproj, ldh, A, awe_save, nullptr, dhproj, dawe_out, dfb, C, E, PS);
    ClassTables tb{cls_cells, cls_w, cls_mult, pix_class};
    decc_score_bwd_kernel<<<bt, 1024, lds, stream>>>((const bf16_t*)att1c, hproj, ldh, wfull, tb, cell_cls, alq, dfb,
                                                                       galpha, datt1c, dhproj, dwfull, P, Q, C, A);
    return ppv_last_error();
}

}  // extern "C"
