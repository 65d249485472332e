// Small dense layers of the caption decoder in exact f32 on the matrix pipe (reference Image_Caption/models.py:199-214: the
// per-step decoder_att / f_beta projection, the LSTM gate GEMM, the 512 -> 9490 vocabulary layer and their transposes in BPTT).
//
//     out[m][n] (+)= sum_k x[m][k] * W[n][k]  (+ bias[n])        x [M][K] f32 (row stride ldx), W [N][K] f32 (row stride ldw)
//
// These GEMMs have 128 rows per time step (one per caption still alive) against 10-40 MB of weights: they are weight streams.
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate, a k-ordered fmaf chain: every product and partial sum is plain f32 arithmetic, 157
// TFLOP/s dense; the K-slices are combined with f32 atomics, so with MORE THAN TWO slices the order of that last addition -- and
// with it the last bit of the result -- varies from run to run; PPV_GEMM_DETERMINISTIC=1 caps the split at two slices, whose sum
// commutes: bit-reproducible at a few per cent of config 3's speed) keeps the
// reference's precision for the recurrence without splitting operands into bf16 terms -- a three-term bf16 split reads 1.5x the
// weight bytes of f32 and needs 16-20 workgroups' worth of 128 x 128 tiles, i.e. 100 us per step at the ~20 GB/s one CU can take in
// (measured: config 3 was 14 % slower with it than with the library GEMMs).
//   * a workgroup owns 16 weight rows (n) and up to 128 x rows (8 accumulators of 16 x 16 per wave); its 4 waves split the
//     workgroup's K range four ways and fold through LDS: N / 16 workgroups per K-slice, 2-4 K-slices (f32 atomics into a
//     pre-zeroed output; slice 0 carries the bias) -- 256-600 workgroups for the decoder's layers;
//   * operands go straight from global memory to the MFMA registers as float4 (16 rows x 64 contiguous bytes per instruction): the
//     four components of a lane's float4 feed four consecutive MFMAs, so the k order inside a 16-chunk is permuted identically for
//     both operands; W is read once, x (<= 1.5 MB) is re-read from L2 / L1.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ppv_common.h"

namespace ppv {

typedef __attribute__((ext_vector_type(4))) float gf32x4;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W, long ldw,
                                                       const float* __restrict__ bias, float* __restrict__ out, long ldo, int M, int N,
                                                       int K, int ksplit, int use_atomics) {
    __shared__ float sred[3][8][4][64];                        // waves 1-3 park their accumulators
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.z * 128;
    const int kslice = blockIdx.y;
    // this workgroup's K range (multiples of 16), split again over the 4 waves
    const int chunks = K / 16;
    const int cpw = (chunks + ksplit * 4 - 1) / (ksplit * 4);
    const int c_begin = (kslice * 4 + wave) * cpw, c_end = min(chunks, c_begin + cpw);
    const int r = lane & 15, q = lane >> 4;
    const int n = n0 + r;
    const bool n_ok = n < N;
    const float* wp = W + (long)(n_ok ? n : 0) * ldw + 4 * q;
    const float* xp[8];
    bool m_ok[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        const int m = m0 + mb * 16 + r;
        m_ok[mb] = m < M;
        xp[mb] = x + (long)(m_ok[mb] ? m : 0) * ldx + 4 * q;
    }
    gf32x4 acc[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[mb] = (gf32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = c_begin; c < c_end; ++c) {
        const int k = c * 16;
        float4 wv = *reinterpret_cast<const float4*>(wp + k);
        if (!n_ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 xv[8];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            xv[mb] = *reinterpret_cast<const float4*>(xp[mb] + k);
            if (!m_ok[mb]) xv[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // D[i = x row][j = weight row]: A = x fragment (lane: row r, k = 4 q + t), B = W fragment (lane: k = 4 q + t, column r)
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[mb].x, wv.x, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[mb].y, wv.y, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[mb].z, wv.z, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[mb].w, wv.w, acc[mb], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) sred[wave - 1][mb][j][lane] = acc[mb][j];
    }
    __syncthreads();
    if (wave == 0) {
        // C/D layout: acc[mb][j] = out[x row mb * 16 + q * 4 + j][weight row r]
        const float b = (bias && kslice == 0 && n_ok) ? bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[mb][j] + sred[0][mb][j][lane] + sred[1][mb][j][lane] + sred[2][mb][j][lane] + b;
                const int m = m0 + mb * 16 + q * 4 + j;
                if (m < M && n_ok) {
                    if (use_atomics) atomicAdd(&out[(long)m * ldo + n], v);
                    else out[(long)m * ldo + n] = v;
                }
            }
    }
}

}  // namespace ppv

using namespace ppv;

extern "C" {

// out[m][n] = sum_k x[m][k] W[n][k] + bias[n] in exact f32 (v_mfma_f32_16x16x4_f32).  x [M][K] (row stride ldx), W [N][K] (row
// stride ldw), bias [N] or NULL, out [M][N] (row stride ldo).  K % 16 == 0; rows 16-byte aligned.  ksplit > 1: partial sums are
// ADDED with f32 atomics -- the caller passes a PRE-ZEROED out.  ksplit <= 0: chosen here (then out needs no zeroing when 1 results).
static int ksplit_target() {                                   // workgroups a launch aims for before it stops splitting K (tuning: PPV_GEMM_WGS)
    static const int t = getenv("PPV_GEMM_WGS") ? atoi(getenv("PPV_GEMM_WGS")) : 1024;   // config 3, B = 128: 256 -> 3827, 512 -> 3913 / 3825, 1024 -> 3862, 2048 -> 3842 images/s (library GEMMs 3888-3964)
    return t;
}

int ppv_gemm_f32_ksplit(int M, int N, int K) {
    const long wgs = (long)((N + 15) / 16) * ((M + 127) / 128);
    static const int ks_max = (getenv("PPV_GEMM_DETERMINISTIC") && atoi(getenv("PPV_GEMM_DETERMINISTIC"))) ? 2 : 8;
    int ks = 1;
    while (ks < ks_max && wgs * ks < ksplit_target() && K / (16 * 4 * ks * 2) >= 2) ks *= 2;
    return ks;
}

int ppv_gemm_f32(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                 int ksplit, hipStream_t stream) {
    if (!x || !W || !out) return PPV_ERR_NULL;
    if (M < 1 || N < 1 || K < 16 || K % 16 || ldx % 4 || ldw % 4 || ksplit < 1 || ksplit > 64) return PPV_ERR_BAD_SIZE;
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)ksplit, (unsigned)((M + 127) / 128));
    gemm_f32_kernel<<<grid, 256, 0, stream>>>(x, ldx, W, ldw, bias, out, ldo, M, N, K, ksplit, ksplit > 1 ? 1 : 0);
    return ppv_last_error();
}

}  // extern "C"
