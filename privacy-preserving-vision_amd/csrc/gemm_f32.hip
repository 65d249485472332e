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

// ----------------------------------------------------------------------------- tiled form (round 4)
// The kernel above gives every WAVE its own copy of the x rows (eight float4 loads per lane and 16-deep k chunk against one of W):
// 7 flop per byte through the L2 -> register path, 230 MB per LSTM gate GEMM -- it ran at 38 TFLOP/s where the f32 matrix pipe offers
// 157 and the library reaches 83.  Here a workgroup owns 128 (m) x 64 (n): the 32-deep k chunk of x (16 KB) and of W (8 KB) is staged
// ONCE in LDS and shared by the four waves (wave w: columns 16 w .. 16 w + 15, all 128 rows, eight 16 x 16 accumulators), so a chunk
// costs 24 KB of L2 traffic for 256 MFMAs (21 flop per byte).  Register-staged double buffer: the loads of chunk c + 1 are in flight
// under the MFMAs of chunk c, one barrier per chunk.  Rows are padded to 36 floats: the 16 lanes of a ds_read_b128 group (fixed k
// quarter, rows r = 0..15) fall on 16 different 4-word bank groups.  Same arithmetic as above (v_mfma_f32_16x16x4_f32, k-ordered f32
// chains, split-K combined with f32 atomics); the float4 components of a lane feed four consecutive MFMAs, i.e. the k order inside a
// 16-chunk is permuted identically for both operands.
constexpr int GT_BM = 128, GT_BN = 64, GT_BK = 32, GT_LD = 36;

// TN = true: both operands K-major -- x[k][m] at x + k * ldx + m, W[k][n] at W + k * ldw + n (the batched weight gradients g^T h of the dense
// layers: k runs over (time step, caption), any K).  A thread loads float4s ALONG m (coalesced 16-byte loads, like the NT form) and the
// transposition happens on the way into LDS: the four m of a float4 go to four rows of the same [row][k] image as four ds_write_b32, and
// the 4-float k-groups of a row are XOR-swizzled by (row >> 4) & 3, which puts the 64 lanes of such a write -- 4 consecutive k x 16
// column groups -- on 64 different banks (bank = 16 (u & 3) + 4 ((k-group ^ (u >> 2)) & 3) + k % 4 for column group u); the fragment
// reads stay 16-byte ds_read_b128 with the swizzled group index (a constant per accumulator row block).  Tried before and dropped: per-k
// dword gathers (24 loads per thread and chunk: 53-63 TFLOP/s) and a K-major LDS image with scalar fragment reads (72 ds_read_b32 per
// chunk: slower still).
// NB = 16-column blocks per wave: the workgroup's tile is 128 x 64 NB.  NB = 2 (128 x 128, 128 MFMAs per wave and chunk against 20
// fragment reads, half the tiles) where the grid stays large enough; NB = 1 otherwise.
template <bool TN, int NB>
__global__ __launch_bounds__(256, 2) void gemm_f32_tiled_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W, long ldw,
                                                                const float* __restrict__ bias, float* __restrict__ out, long ldo, int M,
                                                                int N, int K, int ksplit, int use_atomics, float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) float sX[2][GT_BM * GT_LD];
    constexpr int BN = GT_BN * NB;
    __shared__ __attribute__((aligned(16))) float sW[2][BN * GT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.z * GT_BM, kslice = blockIdx.y;
    const int chunks = (K + GT_BK - 1) / GT_BK;
    const int cps = (chunks + ksplit - 1) / ksplit;
    const int c_begin = kslice * cps, c_end = min(chunks, c_begin + cps);
    if (c_begin >= c_end && !(kslice == 0) && !slab) return;     // (a slab slice without chunks still writes its zeros)
    // staging roles: a thread moves 4 float4 of x and 2 of W per chunk.
    //   NT: float4 index f -> row f / 8, k quarter-pair f % 8 (a float4 = four k of one row).
    //   TN: thread t = c + 4 u + 128 g: k = 16 g + 4 i + c (x; W: t = c + 4 u + 64 g, k = 8 g + 4 i + c), columns 4 u .. 4 u + 3
    //       (a float4 = four rows of one k); per instruction a wave covers 4 k-rows x 256 contiguous bytes.
    const int xr0 = tid >> 3, xk = (tid & 7) * 4;                          // NT: rows xr0 + 32 i, i = 0..3
    const int tc = tid & 3;
    const int txu = (tid >> 2) & 31, txg = tid >> 7;                       // TN x
    const int twu = (tid >> 2) & 15, twg = tid >> 6;                       // TN W
    const float* xsrc[4];
    bool xok[4];
    const float* wsrc[2 * NB];
    bool wok[2 * NB];
    if constexpr (!TN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + xr0 + 32 * i;
            xok[i] = m < M;
            xsrc[i] = x + (long)(xok[i] ? m : 0) * ldx + xk;
        }
#pragma unroll
        for (int i = 0; i < 2 * NB; ++i) {
            const int n = n0 + xr0 + 32 * i;
            wok[i] = n < N;
            wsrc[i] = W + (long)(wok[i] ? n : 0) * ldw + xk;
        }
    }
    float4 rx[4], rw[2 * NB];
    // TN: 16-byte loads when a row leaves room for the float4 that straddles the last column (columns >= M feed output rows that are
    // never stored) and everything is 16-byte aligned; element-wise with predicates otherwise
    const bool vx = TN && ldx % 4 == 0 && ((size_t)x % 16 == 0) && ((M + 3) / 4 * 4 <= ldx);
    const bool vw = TN && ldw % 4 == 0 && ((size_t)W % 16 == 0) && ((N + 3) / 4 * 4 <= ldw);
    auto ld4 = [&](const float* base, long ld_, int k, int c0, int C, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && c0 < C) {
            const float* p = base + (long)k * ld_ + c0;
            if (vec) v = *reinterpret_cast<const float4*>(p);
            else {
                v.x = p[0];
                if (c0 + 1 < C) v.y = p[1];
                if (c0 + 2 < C) v.z = p[2];
                if (c0 + 3 < C) v.w = p[3];
            }
        }
        return v;
    };
    auto fetch = [&](int c) {
        if constexpr (!TN) {
            const int k = c * GT_BK + xk;
            const bool kok = k < K;                                         // K % 16 == 0, xk % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
            for (int i = 0; i < 4; ++i) rx[i] = (xok[i] && kok) ? *reinterpret_cast<const float4*>(xsrc[i] + (long)c * GT_BK) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) rw[i] = (wok[i] && kok) ? *reinterpret_cast<const float4*>(wsrc[i] + (long)c * GT_BK) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) rx[i] = ld4(x, ldx, c * GT_BK + 16 * txg + 4 * i + tc, m0 + 4 * txu, M, vx);
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) rw[i] = ld4(W, ldw, c * GT_BK + 8 * twg + 4 * (i & 1) + tc, n0 + 64 * (i >> 1) + 4 * twu, N, vw);
        }
    };
    auto park = [&](int buf) {
        if constexpr (!TN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&sX[buf][(xr0 + 32 * i) * GT_LD + xk]) = rx[i];
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) *reinterpret_cast<float4*>(&sW[buf][(xr0 + 32 * i) * GT_LD + xk]) = rw[i];
        } else {
            // element (row, k) lives at row * 36 + 4 * ((k >> 2) ^ ((row >> 4) & 3)) + (k & 3); row >> 4 == u >> 2 for all four rows of a float4
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kg = (4 * txg + i) ^ ((txu >> 2) & 3);
                float* d = &sX[buf][(4 * txu) * GT_LD + 4 * kg + tc];
                d[0] = rx[i].x; d[GT_LD] = rx[i].y; d[2 * GT_LD] = rx[i].z; d[3 * GT_LD] = rx[i].w;
            }
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) {
                const int kg = (2 * twg + (i & 1)) ^ ((twu >> 2) & 3);        // (rows 64 apart share (row >> 4) & 3)
                float* d = &sW[buf][(64 * (i >> 1) + 4 * twu) * GT_LD + 4 * kg + tc];
                d[0] = rw[i].x; d[GT_LD] = rw[i].y; d[2 * GT_LD] = rw[i].z; d[3 * GT_LD] = rw[i].w;
            }
        }
    };
    gf32x4 acc[NB][8];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = (gf32x4){0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, q = lane >> 4;
    if (c_begin < c_end) {
        fetch(c_begin);
        park(0);
        __syncthreads();
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            if (c + 1 < c_end) fetch(c + 1);
            const float* tx = &sX[buf][r * GT_LD];
            const float* tw = &sW[buf][(wave * 16 + r) * GT_LD];
            // k-group of this lane's fragment in half kk of the chunk: 4 kk + q, swizzled by the row block in the TN image
            auto xg = [&](int kk, int mb) { return 4 * ((4 * kk + q) ^ (TN ? (mb & 3) : 0)); };
            auto wg_ = [&](int kk) { return 4 * ((4 * kk + q) ^ (TN ? (wave & 3) : 0)); };
            // all fragment reads of the chunk are issued up front (LDS returns in order: the second half's reads fly under the first
            // half's MFMAs); MFMAs component-major, so that eight independent accumulators separate two dependent ones.  Wave w owns the
            // columns nb * 64 + w * 16 + r of the tile.
            float4 wv[2][NB], xv[2][8];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wv[kk][nb] = *reinterpret_cast<const float4*>(tw + nb * 64 * GT_LD + wg_(kk));
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) xv[kk][mb] = *reinterpret_cast<const float4*>(tx + mb * 16 * GT_LD + xg(kk, mb));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[kk][mb].x, wv[kk][nb].x, acc[nb][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[kk][mb].y, wv[kk][nb].y, acc[nb][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[kk][mb].z, wv[kk][nb].z, acc[nb][mb], 0, 0, 0);
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[kk][mb].w, wv[kk][nb].w, acc[nb][mb], 0, 0, 0);
                }
            if (c + 1 < c_end) park(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // C/D layout: acc[nb][mb][j] = out[x row mb * 16 + q * 4 + j][weight row nb * 64 + wave * 16 + r]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 64 + wave * 16 + r;
        if (n >= N) continue;
        if (slab) {                                                         // split K without atomics: slice kslice of [ksplit][M][N], summed by gemm_f32_slab_sum_kernel
            float* dst = slab + (long)kslice * M * N;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + mb * 16 + q * 4 + j;
                    if (m < M) dst[(long)m * N + n] = acc[nb][mb][j];
                }
            continue;
        }
        const float b = (bias && kslice == 0) ? bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + mb * 16 + q * 4 + j;
                if (m < M) {
                    const float v = acc[nb][mb][j] + b;
                    if (use_atomics) atomicAdd(&out[(long)m * ldo + n], v);
                    else out[(long)m * ldo + n] = v;
                }
            }
    }
}

// out[m][n] = bias[n] + sum_s slab[s][m][n]  (slices in index order: the result does not depend on which workgroup finished first)
__global__ __launch_bounds__(256) void gemm_f32_slab_sum_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                                float* __restrict__ out, long ldo, int M, int N, int ksplit) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;                   // one float4 of a row (N % 4 == 0)
    const int n4 = N / 4;
    if (i >= (long)M * n4) return;
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    float4 a = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long stride = (long)M * N;
    const float* p = slab + (long)m * N + n;
    for (int s_ = 0; s_ < ksplit; ++s_) {
        const float4 v = *reinterpret_cast<const float4*>(p + s_ * stride);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (long)m * ldo + n) = a;
}


// ----------------------------------------------------------------------------- bf16 x 3 form of the batched weight gradients (round 5)
// out[m][n] = sum_k a[k][m] * b[k][n], both operands f32 and K-major as they lie in memory (ppv_gemm_f32_tn's contract), on the BF16
// matrix pipe: every f32 operand is split IN THE KERNEL, on its way into LDS, into hi = bf16(v) and lo = bf16(v - hi), and a product
// is taken as hi*hi + hi*lo + lo*hi with f32 accumulation (the dropped lo*lo term and lo's own rounding are ~2^-17 of the product:
// the error of a sum is ~1e-5 of sum |a b|, two orders above exact f32 and two below the step's bf16 trunk).  Three
// v_mfma_f32_16x16x32_bf16 do the work of eight v_mfma_f32_16x16x4_f32 at 1/16 of their cycles each: 5.3x the matrix rate of
// gemm_f32_tiled_kernel<true, .>, with the same bytes read (the split costs no pass over memory).
//   * workgroup = 128 (m) x 128 (n), four waves as 2 x 2, a wave owns 64 x 64 = 4 x 4 accumulator blocks; K chunk = 32;
//   * staging: a thread loads float4s ALONG m / n (four columns of one k-row; a wave-instruction covers 4 k-rows x 256 contiguous
//     bytes), splits them and writes 8 bytes of hi and 8 of lo into K-MAJOR images [32 k][128 columns] bf16 whose rows are padded to
//     288 bytes (= 8 banks mod 64): the 16 lanes of a ds_write_b64 group (4 k-rows x 4 column quads) and the 32 lanes of a
//     ds_read_b64_tr_b16 half (8 k-rows x 32 bytes) fall on distinct banks;
//   * fragments by ds_read_b64_tr_b16 (the transposing read turns "4 k-rows x 16 columns" into "column i: 4 consecutive k" per lane):
//     lane group g = lane >> 4 reads k-rows 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3, i.e. fragment element j sits at
//     k = 16 (j >> 2) + 4 g + (j & 3) -- the same permutation of the chunk's 32 k for both operands, which is all an MFMA needs;
//   * register-staged double buffer (the loads of chunk c + 1 fly under the MFMAs of chunk c), one barrier per chunk; split K through
//     slabs summed in index order by gemm_f32_slab_sum_kernel (bit-reproducible, no atomics).
typedef __attribute__((ext_vector_type(8))) __bf16 x3_bf16x8;
typedef __attribute__((ext_vector_type(4))) short x3_s16x4;
typedef __attribute__((ext_vector_type(8))) short x3_s16x8;
constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32, X3_ROWB = 288, X3_IMG = X3_BK * X3_ROWB;

__device__ __forceinline__ void x3_split4(const float4 v, uint2& hi, uint2& lo) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ a = {v.x, v.y}, b = {v.z, v.w};
    const bf16x2_ ha = __builtin_convertvector(a, bf16x2_), hb = __builtin_convertvector(b, bf16x2_);     // v_cvt_pk_bf16_f32: RNE
    const f32x2_ ra = a - __builtin_convertvector(ha, f32x2_), rb = b - __builtin_convertvector(hb, f32x2_);   // exact in f32
    const bf16x2_ la = __builtin_convertvector(ra, bf16x2_), lb = __builtin_convertvector(rb, bf16x2_);
    hi = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
    lo = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_tn_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                                float* __restrict__ out, long ldo, int M, int N, int K, int ksplit,
                                                                float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) char x3_smem[];      // [2 buffers][A hi | A lo | B hi | B lo] images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * X3_BN, m0 = blockIdx.z * X3_BM, kslice = blockIdx.y;
    const int chunks = (K + X3_BK - 1) / X3_BK;
    const int cps = (chunks + ksplit - 1) / ksplit;
    const int c_begin = kslice * cps, c_end = min(chunks, c_begin + cps);
    // staging role: k-row 16 g + 4 i + tc of the chunk (i = 0..3), columns 4 u .. 4 u + 3
    const int tc = tid & 3, u = (tid >> 2) & 31, sg = tid >> 7;
    const bool va = lda % 4 == 0 && ((size_t)a % 16 == 0) && ((M + 3) / 4 * 4 <= lda);
    const bool vb = ldb % 4 == 0 && ((size_t)b % 16 == 0) && ((N + 3) / 4 * 4 <= ldb);
    auto ld4 = [&](const float* base, long ld_, int k, int c0, int C, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && c0 < C) {
            const float* p = base + (long)k * ld_ + c0;
            if (vec) v = *reinterpret_cast<const float4*>(p);
            else {
                v.x = p[0];
                if (c0 + 1 < C) v.y = p[1];
                if (c0 + 2 < C) v.z = p[2];
                if (c0 + 3 < C) v.w = p[3];
            }
        }
        return v;
    };
    float4 ra[4], rb[4];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = ld4(a, lda, c * X3_BK + 16 * sg + 4 * i + tc, m0 + 4 * u, M, va);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = ld4(b, ldb, c * X3_BK + 16 * sg + 4 * i + tc, n0 + 4 * u, N, vb);
    };
    auto park = [&](int buf) {
        char* base = x3_smem + buf * 4 * X3_IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (16 * sg + 4 * i + tc) * X3_ROWB + 8 * u;
            uint2 hi, lo;
            x3_split4(ra[i], hi, lo);
            *reinterpret_cast<uint2*>(base + off) = hi;
            *reinterpret_cast<uint2*>(base + X3_IMG + off) = lo;
            x3_split4(rb[i], hi, lo);
            *reinterpret_cast<uint2*>(base + 2 * X3_IMG + off) = hi;
            *reinterpret_cast<uint2*>(base + 3 * X3_IMG + off) = lo;
        }
    };
    gf32x4 acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (gf32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;
    const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    // fragment of column block cb of an image: two transposing reads (k-rows 4 fg + fq and 16 + 4 fg + fq, 8 bytes at column quad fp)
    auto frag = [&](const char* img, int cb) {
        const char* p0 = img + (4 * fg + fq) * X3_ROWB + cb * 32 + fp * 8;
        const x3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4*)(p0));
        const x3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4*)(p0 + 16 * X3_ROWB));
        const x3_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(x3_bf16x8, v);
    };
    if (c_begin < c_end) {
        fetch(c_begin);
        park(0);
        __syncthreads();
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            if (c + 1 < c_end) fetch(c + 1);
            const char* base = x3_smem + buf * 4 * X3_IMG;
            x3_bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = frag(base, wm * 4 + i);
                bh[i] = frag(base + 2 * X3_IMG, wn * 4 + i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                al[i] = frag(base + X3_IMG, wm * 4 + i);
                bl[i] = frag(base + 3 * X3_IMG, wn * 4 + i);
            }
            // the two small terms first, the large one last (one fewer rounding of the large partial sum against small addends)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
            if (c + 1 < c_end) park(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // C/D layout: acc[mb][nb][j] = out[m0 + wm * 64 + mb * 16 + fg * 4 + j][n0 + wn * 64 + nb * 16 + (lane & 15)]
    float* dst = slab ? slab + (long)kslice * M * N : out;
    const long ldd = slab ? N : ldo;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int n = n0 + wn * 64 + nb * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wm * 64 + mb * 16 + fg * 4 + j;
                if (m < M) dst[(long)m * ldd + n] = acc[mb][nb][j];
            }
    }
}


// The same three-product form for ROW-major operands: out[m][n] = sum_k x[m][k] W[n][k] (+ bias[n]) -- ppv_gemm_f32's contract -- for the
// two large NON-recurrent dense products of the decoder: the vocabulary layer over all time steps (models.py:211 batched, [T B, 512] x
// [9490, 512]^T) and its transposed data gradient ([T B, 9504] x [512, 9504]^T); the per-step layers that feed the LSTM state back stay
// exact f32.  Staging: a thread loads float4s along k (8 lanes = one 128-byte line of a row), splits them and writes 8 bytes of hi and
// of lo into [row][32 k] bf16 images (64-byte rows, 16-byte chunk index XOR 3 * ((row >> 3) & 1): the conflict-free ds_read_b128 pattern
// of conv_gemm_pipe_kernel's BK = 32 tiles); fragments are plain ds_read_b128 (8 consecutive k of a row).  Same tile (128 x 128, four
// waves of 64 x 64), pipeline and slab split-K as gemm_bf16x3_tn_kernel.  K % 4 == 0, rows 16-byte aligned.
constexpr int X3N_IMG = 128 * 64;                                  // one [128 rows][32 k] bf16 image

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_nt_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W, long ldw,
                                                                const float* __restrict__ bias, float* __restrict__ out, long ldo, int M,
                                                                int N, int K, int ksplit, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) char x3n_smem[];     // [2 buffers][x hi | x lo | W hi | W lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * X3_BN, m0 = blockIdx.z * X3_BM, kslice = blockIdx.y;
    const int chunks = (K + X3_BK - 1) / X3_BK;
    const int cps = (chunks + ksplit - 1) / ksplit;
    const int c_begin = kslice * cps, c_end = min(chunks, c_begin + cps);
    // staging role: float4 kq (4 k) of rows r0 + 32 i
    const int kq = tid & 7, r0 = tid >> 3;
    const float* xs[4];
    const float* ws[4];
    bool xok[4], wok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i, n = n0 + r0 + 32 * i;
        xok[i] = m < M; wok[i] = n < N;
        xs[i] = x + (long)(xok[i] ? m : 0) * ldx + 4 * kq;
        ws[i] = W + (long)(wok[i] ? n : 0) * ldw + 4 * kq;
    }
    float4 rx[4], rw[4];
    auto fetch = [&](int c) {
        const bool kok = c * X3_BK + 4 * kq < K;                       // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[i] = (xok[i] && kok) ? *reinterpret_cast<const float4*>(xs[i] + (long)c * X3_BK) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = (wok[i] && kok) ? *reinterpret_cast<const float4*>(ws[i] + (long)c * X3_BK) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto park = [&](int buf) {
        char* base = x3n_smem + buf * 4 * X3N_IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 32 * i;
            const int off = row * 64 + (((kq >> 1) ^ (((row >> 3) & 1) * 3)) * 16) + (kq & 1) * 8;
            uint2 hi, lo;
            x3_split4(rx[i], hi, lo);
            *reinterpret_cast<uint2*>(base + off) = hi;
            *reinterpret_cast<uint2*>(base + X3N_IMG + off) = lo;
            x3_split4(rw[i], hi, lo);
            *reinterpret_cast<uint2*>(base + 2 * X3N_IMG + off) = hi;
            *reinterpret_cast<uint2*>(base + 3 * X3N_IMG + off) = lo;
        }
    };
    gf32x4 acc[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (gf32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    auto frag = [&](const char* img, int rb) {
        const int row = rb * 16 + fr;
        return *reinterpret_cast<const x3_bf16x8*>(img + row * 64 + ((fq ^ (((row >> 3) & 1) * 3)) * 16));
    };
    if (c_begin < c_end) {
        fetch(c_begin);
        park(0);
        __syncthreads();
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            if (c + 1 < c_end) fetch(c + 1);
            const char* base = x3n_smem + buf * 4 * X3N_IMG;
            x3_bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = frag(base, wm * 4 + i);
                bh[i] = frag(base + 2 * X3N_IMG, wn * 4 + i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                al[i] = frag(base + X3N_IMG, wm * 4 + i);
                bl[i] = frag(base + 3 * X3N_IMG, wn * 4 + i);
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
            if (c + 1 < c_end) park(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // C/D layout: acc[mb][nb][j] = out[m0 + wm * 64 + mb * 16 + fq * 4 + j][n0 + wn * 64 + nb * 16 + fr]
    float* dst = slab ? slab + (long)kslice * M * N : out;
    const long ldd = slab ? N : ldo;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int n = n0 + wn * 64 + nb * 16 + fr;
        if (n >= N) continue;
        const float b = (bias && !slab) ? bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wm * 64 + mb * 16 + fq * 4 + j;
                if (m < M) dst[(long)m * ldd + n] = acc[mb][nb][j] + b;
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

static int gemm_tiled() {                                       // PPV_GEMM_TILED=0: the wave-private form above for every shape (A/B)
    static const int t = getenv("PPV_GEMM_TILED") ? atoi(getenv("PPV_GEMM_TILED")) : 1;
    return t;
}
static bool use_tiled(int M, int N, int K) { return gemm_tiled() && M >= 32 && N >= 64 && K >= 64; }
// 128 x 128 tiles where they still give every CU a workgroup; 128 x 64 otherwise (PPV_GEMM_NB=1 / 2 forces one form)
static int tiled_nb(int M, int N) {
    static const int f = getenv("PPV_GEMM_NB") ? atoi(getenv("PPV_GEMM_NB")) : 0;
    if (f == 1 || f == 2) return f;
    const long t128 = (long)((N + 127) / 128) * ((M + GT_BM - 1) / GT_BM);
    return t128 >= 256 ? 2 : 1;         // measured (tools/bench_gemm_tn.py): 300-320 tiles of 128 x 128 beat 600-640 of 128 x 64 by 9-11 %
}
}  // extern "C" (a template needs C++ linkage)
template <bool TN>
static void launch_tiled(int nb, dim3 grid, hipStream_t stream, const float* x, long ldx, const float* W, long ldw, const float* bias, float* out,
                         long ldo, int M, int N, int K, int ksplit, int atom, float* slab) {
    if (nb == 2) gemm_f32_tiled_kernel<TN, 2><<<grid, 256, 0, stream>>>(x, ldx, W, ldw, bias, out, ldo, M, N, K, ksplit, atom, slab);
    else gemm_f32_tiled_kernel<TN, 1><<<grid, 256, 0, stream>>>(x, ldx, W, ldw, bias, out, ldo, M, N, K, ksplit, atom, slab);
}
extern "C" {
static dim3 tiled_grid(int nb, int M, int N, int ksplit) {
    const int bn = GT_BN * nb;
    return dim3((unsigned)((N + bn - 1) / bn), (unsigned)ksplit, (unsigned)((M + GT_BM - 1) / GT_BM));
}
static long tiled_tiles(int M, int N) {
    const int bn = GT_BN * tiled_nb(M, N);
    return (long)((N + bn - 1) / bn) * ((M + GT_BM - 1) / GT_BM);
}

int ppv_gemm_f32_ksplit(int M, int N, int K) {
    static const int ks_max = (getenv("PPV_GEMM_DETERMINISTIC") && atoi(getenv("PPV_GEMM_DETERMINISTIC"))) ? 2 : 8;
    if (use_tiled(M, N, K)) {
        // 128 x 64 tiles, two workgroups per CU: split K until ~512 workgroups exist, keeping >= 4 chunks of 32 per slice
        const long tiles = tiled_tiles(M, N);
        static const int target = getenv("PPV_GEMM_TILED_WGS") ? atoi(getenv("PPV_GEMM_TILED_WGS")) : 512;
        const int chunks = (K + GT_BK - 1) / GT_BK;
        int ks = 1;
        while (ks < ks_max && tiles * ks < target && chunks / (ks * 2) >= 4) ks *= 2;
        return ks;
    }
    const long wgs = (long)((N + 15) / 16) * ((M + 127) / 128);
    int ks = 1;
    while (ks < ks_max && wgs * ks < ksplit_target() && K / (16 * 4 * ks * 2) >= 2) ks *= 2;
    return ks;
}

int ppv_gemm_f32(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                 int ksplit, hipStream_t stream) {
    if (!x || !W || !out) return PPV_ERR_NULL;
    if (M < 1 || N < 1 || K < 16 || K % 16 || ldx % 4 || ldw % 4 || ksplit < 1 || ksplit > 64) return PPV_ERR_BAD_SIZE;
    if (use_tiled(M, N, K) && ((size_t)x % 16 == 0) && ((size_t)W % 16 == 0)) {
        const int nb = tiled_nb(M, N);
        launch_tiled<false>(nb, tiled_grid(nb, M, N, ksplit), stream, x, ldx, W, ldw, bias, out, ldo, M, N, K, ksplit, ksplit > 1 ? 1 : 0, nullptr);
        return ppv_last_error();
    }
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)ksplit, (unsigned)((M + 127) / 128));
    gemm_f32_kernel<<<grid, 256, 0, stream>>>(x, ldx, W, ldw, bias, out, ldo, M, N, K, ksplit, ksplit > 1 ? 1 : 0);
    return ppv_last_error();
}

// The same product with the K slices combined WITHOUT atomics and without a pre-zeroed output: every slice stores its tile into
// workspace [ksplit][M][N] f32 and one small launch sums the slices in index order (bit-reproducible for any split; no fill launch in
// front of the GEMM).  ppv_gemm_f32_ws_plan gives the split this form wants (up to 16 slices: slabs cost a plain store, so M = 128
// layers can put two workgroups on every CU) and the bytes it needs; ksplit == 1 (or shapes the tiled kernel does not serve) falls
// through to ppv_gemm_f32.  N % 4 == 0, out rows 16-byte aligned when ksplit > 1.
int ppv_gemm_f32_ws_plan(int M, int N, int K, size_t* bytes) {
    int ks = 1;
    if (use_tiled(M, N, K) && N % 4 == 0) {
        const long tiles = tiled_tiles(M, N);
        static const int target = getenv("PPV_GEMM_WS_WGS") ? atoi(getenv("PPV_GEMM_WS_WGS")) : 512;
        const int chunks = (K + GT_BK - 1) / GT_BK;
        while (ks < 16 && tiles * ks < target && chunks / (ks * 2) >= 3) ks *= 2;
    }
    if (bytes) *bytes = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
    return ks;
}

int ppv_gemm_f32_ws(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                    int ksplit, void* workspace, hipStream_t stream) {
    if (!x || !W || !out) return PPV_ERR_NULL;
    if (M < 1 || N < 1 || K < 16 || K % 16 || ldx % 4 || ldw % 4 || ksplit < 1 || ksplit > 64) return PPV_ERR_BAD_SIZE;
    if (ksplit == 1 || !use_tiled(M, N, K) || ((size_t)x % 16) || ((size_t)W % 16)) {
        if (ksplit > 1) return PPV_ERR_BAD_SIZE;                 // (the atomics form needs a zeroed output: the caller must ask ppv_gemm_f32 for that)
        return ppv_gemm_f32(x, ldx, W, ldw, bias, out, ldo, M, N, K, 1, stream);
    }
    if (!workspace) return PPV_ERR_NULL;
    if (N % 4 || ldo % 4 || ((size_t)out % 16) || (bias && ((size_t)bias % 16))) return PPV_ERR_BAD_SIZE;
    const int nb = tiled_nb(M, N);
    launch_tiled<false>(nb, tiled_grid(nb, M, N, ksplit), stream, x, ldx, W, ldw, nullptr, nullptr, 0, M, N, K, ksplit, 0, (float*)workspace);
    const long n4 = (long)M * (N / 4);
    gemm_f32_slab_sum_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>((const float*)workspace, bias, out, ldo, M, N, ksplit);
    return ppv_last_error();
}

// out[m][n] = sum_k a[k][m] * b[k][n]: the batched weight gradient g^T h of a dense layer (Image_Caption/models.py:199-214 under autograd:
// d W = sum over time steps and captions of (d out)^T in), both operands as they lie in memory -- a [K][M] (row stride lda), b [K][N]
// (row stride ldb), out [M][N] (row stride ldo); any K.  Same tile, arithmetic and slab combination as ppv_gemm_f32_ws;
// ppv_gemm_f32_tn_plan gives the split and the workspace bytes (0: none needed).  N % 4 == 0 and out 16-byte aligned when split.
int ppv_gemm_f32_tn_plan(int M, int N, int K, size_t* bytes) {
    const long tiles = tiled_tiles(M, N);
    static const int target = getenv("PPV_GEMM_WS_WGS") ? atoi(getenv("PPV_GEMM_WS_WGS")) : 512;
    const int chunks = (K + GT_BK - 1) / GT_BK;
    int ks = 1;
    if (N % 4 == 0)
        while (ks < 16 && tiles * ks < target && chunks / (ks * 2) >= 3) ks *= 2;
    if (bytes) *bytes = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
    return ks;
}

int ppv_gemm_f32_tn(const float* a, long lda, const float* b, long ldb, float* out, long ldo, int M, int N, int K, int ksplit,
                    void* workspace, hipStream_t stream) {
    if (!a || !b || !out) return PPV_ERR_NULL;
    if (M < 1 || N < 1 || K < 1 || ksplit < 1 || ksplit > 64) return PPV_ERR_BAD_SIZE;
    const int nb = tiled_nb(M, N);
    const dim3 grid = tiled_grid(nb, M, N, ksplit);
    if (ksplit == 1) {
        launch_tiled<true>(nb, grid, stream, a, lda, b, ldb, nullptr, out, ldo, M, N, K, 1, 0, nullptr);
        return ppv_last_error();
    }
    if (!workspace) return PPV_ERR_NULL;
    if (N % 4 || ldo % 4 || ((size_t)out % 16)) return PPV_ERR_BAD_SIZE;
    launch_tiled<true>(nb, grid, stream, a, lda, b, ldb, nullptr, nullptr, 0, M, N, K, ksplit, 0, (float*)workspace);
    const long n4 = (long)M * (N / 4);
    gemm_f32_slab_sum_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>((const float*)workspace, nullptr, out, ldo, M, N, ksplit);
    return ppv_last_error();
}

// The same product a^T b (ppv_gemm_f32_tn's contract: a [K][M], b [K][N] f32, out [M][N] f32, any K / M / N) on the bf16 matrix pipe as
// three products of in-kernel bf16 splits (gemm_bf16x3_tn_kernel above): ~1e-5 of sum |a b| instead of exact f32, 5.3x the matrix
// rate.  The decoder's five batched weight gradients (Image_Caption/models.py:199-214 under autograd) run on it by default
// (PPV_DEC_WGRAD=x3); ppv_gemm_f32_tn stays for exact f32.  ppv_gemm_bf16x3_tn_plan: split and workspace bytes (0: none).
int ppv_gemm_bf16x3_tn_plan(int M, int N, int K, size_t* bytes) {
    const long tiles = (long)((M + X3_BM - 1) / X3_BM) * ((N + X3_BN - 1) / X3_BN);
    static const int target = getenv("PPV_GEMM_X3_WGS") ? atoi(getenv("PPV_GEMM_X3_WGS")) : 512;
    const int chunks = (K + X3_BK - 1) / X3_BK;
    int ks = 1;
    if (N % 4 == 0)
        while (ks < 16 && tiles * ks < target && chunks / (ks * 2) >= 3) ks *= 2;
    if (bytes) *bytes = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
    return ks;
}

int ppv_gemm_bf16x3_tn(const float* a, long lda, const float* b, long ldb, float* out, long ldo, int M, int N, int K, int ksplit,
                       void* workspace, hipStream_t stream) {
    if (!a || !b || !out) return PPV_ERR_NULL;
    if (M < 1 || N < 1 || K < 1 || ksplit < 1 || ksplit > 64) return PPV_ERR_BAD_SIZE;
    constexpr int lds = 2 * 4 * X3_IMG;
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)gemm_bf16x3_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_once.done();
    }
    const dim3 grid((unsigned)((N + X3_BN - 1) / X3_BN), (unsigned)ksplit, (unsigned)((M + X3_BM - 1) / X3_BM));
    if (ksplit == 1) {
        gemm_bf16x3_tn_kernel<<<grid, 256, lds, stream>>>(a, lda, b, ldb, out, ldo, M, N, K, 1, nullptr);
        return ppv_last_error();
    }
    if (!workspace) return PPV_ERR_NULL;
    if (N % 4 || ldo % 4 || ((size_t)out % 16)) return PPV_ERR_BAD_SIZE;
    gemm_bf16x3_tn_kernel<<<grid, 256, lds, stream>>>(a, lda, b, ldb, nullptr, 0, M, N, K, ksplit, (float*)workspace);
    const long n4 = (long)M * (N / 4);
    gemm_f32_slab_sum_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>((const float*)workspace, nullptr, out, ldo, M, N, ksplit);
    return ppv_last_error();
}

// out = x W^T + bias (ppv_gemm_f32's contract: x [M][K], W [N][K] f32 row-major with row strides, bias [N] or null) as three bf16 products
// of in-kernel hi / lo splits (gemm_bf16x3_nt_kernel): for the decoder's two large non-recurrent products (vocabulary layer over all
// steps and its transposed data gradient).  K % 4 == 0, rows 16-byte aligned; ppv_gemm_bf16x3_nt_plan: split and workspace bytes.
int ppv_gemm_bf16x3_nt_plan(int M, int N, int K, size_t* bytes) {
    const long tiles = (long)((M + X3_BM - 1) / X3_BM) * ((N + X3_BN - 1) / X3_BN);
    static const int target = getenv("PPV_GEMM_X3_WGS") ? atoi(getenv("PPV_GEMM_X3_WGS")) : 512;
    const int chunks = (K + X3_BK - 1) / X3_BK;
    int ks = 1;
    if (N % 4 == 0)
        while (ks < 16 && tiles * ks < target && chunks / (ks * 2) >= 3) ks *= 2;
    if (bytes) *bytes = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
    return ks;
}

int ppv_gemm_bf16x3_nt(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                       int ksplit, void* workspace, hipStream_t stream) {
    if (!x || !W || !out) return PPV_ERR_NULL;
    if (M < 1 || N < 1 || K < 4 || K % 4 || ldx % 4 || ldw % 4 || ((size_t)x % 16) || ((size_t)W % 16) || ksplit < 1 || ksplit > 64) return PPV_ERR_BAD_SIZE;
    constexpr int lds = 2 * 4 * X3N_IMG;
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)gemm_bf16x3_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_once.done();
    }
    const dim3 grid((unsigned)((N + X3_BN - 1) / X3_BN), (unsigned)ksplit, (unsigned)((M + X3_BM - 1) / X3_BM));
    if (ksplit == 1) {
        gemm_bf16x3_nt_kernel<<<grid, 256, lds, stream>>>(x, ldx, W, ldw, bias, out, ldo, M, N, K, 1, nullptr);
        return ppv_last_error();
    }
    if (!workspace) return PPV_ERR_NULL;
    if (N % 4 || ldo % 4 || ((size_t)out % 16) || (bias && ((size_t)bias % 16))) return PPV_ERR_BAD_SIZE;
    gemm_bf16x3_nt_kernel<<<grid, 256, lds, stream>>>(x, ldx, W, ldw, nullptr, nullptr, 0, M, N, K, ksplit, (float*)workspace);
    const long n4 = (long)M * (N / 4);
    gemm_f32_slab_sum_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>((const float*)workspace, bias, out, ldo, M, N, ksplit);
    return ppv_last_error();
}

}  // extern "C"
