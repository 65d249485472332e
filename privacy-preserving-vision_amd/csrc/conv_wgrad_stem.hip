// Weight gradient of the NHWC bf16 convolutions and the ResNet stem (7x7/2, 3 -> 64 channels) on MFMA, gfx950.
//
// wgrad (layer2..4 of the Encoder are trainable, models.py:43-54):
//     dW[n][r][s][c] = sum_m G[m][n] * src[b, ho*st + r - pad, wo*st + s - pad, c]
//   Both operands have the reduction index m as their SLOW memory dimension, so both 64-row tiles are staged
//   row-major ([m][128 cols], global_load_lds, XOR-swizzled 32-byte blocks) and the MFMA fragments are read
//   TRANSPOSED with ds_read_b64_tr_b16 -- no register or LDS transpose pass.  One workgroup owns a
//   128 (n) x 128 (c) tile of one tap and a slice of m; slices combine with f32 atomics into [N][R][S][C].
//
// stem forward: K = 7*3*8 = 168 (padded to 192) is built in LDS as an im2col tile from the f32 NCHW sensor image
//   (each 16-byte chunk = 7 consecutive input columns + one zero), 64 output channels, BN partial statistics fused.
// stem data gradient (needed: the lens trains through the frozen stem): expressed as a 4x4-tap, stride-1,
//   16-column conv over 2x2 input super-pixels and run by conv_gemm (N = 16); this file only holds its weight
//   re-layout and the final scatter to NCHW f32.
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

__device__ __forceinline__ bf16_t f2bfw(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2fw(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

#define GLDS16W(gptr, lptr)                                                                                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                  \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct WgradGeom {
    int B, Hs, Ws, Cs;   // conv input (source) NHWC
    int Ho, Wo, N;       // conv output grid / channels (G is [B,Ho,Wo,N])
    int R, S, st, pad;
    long M;              // B*Ho*Wo
    int stages_per_split;  // 64-row stages each m-slice walks
};

// 32-byte block swizzle key of a staged row (conflict-free ds_read_b64_tr_b16: see file header)
__device__ __forceinline__ int trkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int k0, int colblock, int lane) {
    // operand fragment for k = k0 + 8*(lane>>4) + 0..7, 16 columns colblock*16.. of a [64][128] bf16 row-major tile
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = k0 + 8 * g + q, r1 = r0 + 4;
    const int b0 = (colblock ^ trkey(r0)), b1 = (colblock ^ trkey(r1));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + r0 * 256 + b0 * 32 + p * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + r1 * 256 + b1 * 32 + p * 8));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                            float* __restrict__ dW, const bf16_t* __restrict__ zero_page,
                                                            WgradGeom g) {
    constexpr int TILE_BYTES = 64 * 256;                     // [64 m][128 cols] bf16
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // stage s: G at 2s, X at 2s+1
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = blockIdx.x * 128;
    const int ctiles = g.Cs / 128;
    const int tap = blockIdx.y / ctiles, c0 = (blockIdx.y % ctiles) * 128;
    const int r = tap / g.S, s = tap % g.S;
    const long m_begin = (long)blockIdx.z * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    const int nst = (int)((m_end - m_begin + 63) / 64);
    if (nst <= 0) return;
    const int HoWo = g.Ho * g.Wo;

    // staging roles: per stage each thread moves 4 chunks of G and 4 of X.  One glds wave-instruction = 1 KiB =
    // 4 rows x 16 chunks; lane -> (row_in_instr = lane >> 4, lds chunk = lane & 15)
    const int rli = lane >> 4, lch = lane & 15;
    auto stage = [&](int buf, int st_idx) {
        const long mb = m_begin + (long)st_idx * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + wave * 4 + rli;             // 0..63
            const long m = mb + row;
            const int key = trkey(row);
            const int gch = (((lch >> 1) ^ key) << 1) | (lch & 1);   // global chunk that lands at lds chunk lch
            const bool ok = m < m_end;
            const bf16_t* gsrc = ok ? G + m * g.N + n0 + gch * 8 : zero_page;
            GLDS16W(gsrc, smem + (2 * buf) * TILE_BYTES + (i * 16 + wave * 4) * 256);
            const bf16_t* xsrc = zero_page;
            if (ok) {
                const int b = (int)(m / HoWo), rem = (int)(m % HoWo);
                const int ho = rem / g.Wo, wo = rem % g.Wo;
                const int hs = ho * g.st + r - g.pad, ws = wo * g.st + s - g.pad;
                if (hs >= 0 && ws >= 0 && hs < g.Hs && ws < g.Ws)
                    xsrc = X + (((long)b * g.Hs + hs) * g.Ws + ws) * g.Cs + c0 + gch * 8;
            }
            GLDS16W(xsrc, smem + (2 * buf + 1) * TILE_BYTES + (i * 16 + wave * 4) * 256);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;

    auto compute = [&](int buf) {
        const char* tg = smem + (2 * buf) * TILE_BYTES;
        const char* tx = smem + (2 * buf + 1) * TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = tr_frag(tg, kk * 32, wm * 4 + mi, lane);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bfr[ni] = tr_frag(tx, kk * 32, wn * 4 + ni, lane);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nst - 1; ++t) {
        stage(cur ^ 1, t + 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // D[row = n][col = c]: row = (lane>>4)*4 + j, col = lane & 15
    const int fr = lane & 15, fq = lane >> 4;
    const long wrow = (long)g.R * g.S * g.Cs;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wm * 64 + mi * 16 + fq * 4 + j;
                const int c = c0 + wn * 64 + ni * 16 + fr;
                atomicAdd(&dW[(long)n * wrow + (long)tap * g.Cs + c], acc[mi][ni][j]);
            }
}

// [N][R][S][C] f32 -> torch [N][C][R][S] f32
__global__ __launch_bounds__(256) void wgrad_to_torch_kernel(const float* __restrict__ dW, float* __restrict__ out, int N, int C,
                                                             int R, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)N * C * R * S;
    if (i >= tot) return;
    const int s = (int)(i % S), r = (int)((i / S) % R), c = (int)((i / ((long)S * R)) % C), n = (int)(i / ((long)S * R * C));
    out[i] = dW[(((long)n * R + r) * S + s) * C + c];
}

// ============================================================================= stem forward
// torch [64][3][7][7] f32 -> [64][24 chunks][8] bf16, chunk = r*3 + c (21 used), element s (7 used)
__global__ __launch_bounds__(256) void stem_weight_layout_kernel(const float* __restrict__ w, bf16_t* __restrict__ o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * 192) return;
    const int s = i & 7, ch = (i >> 3) % 24, n = i / 192;
    float v = 0.f;
    if (ch < 21 && s < 7) {
        const int r = ch / 3, c = ch % 3;
        v = w[((n * 3 + c) * 7 + r) * 7 + s];
    }
    o[i] = f2bfw(v);
}

__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wst,
                                                           bf16_t* __restrict__ out, float* __restrict__ stat_part, int B,
                                                           int H, int W, int tiles, int stat_rows) {
    constexpr int LDA = 384;                                   // bytes per A / W row (192 bf16)
    __shared__ __attribute__((aligned(16))) char sA[128 * LDA];
    __shared__ __attribute__((aligned(16))) char sW[64 * LDA];
    __shared__ float sStat[2][2][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Ho = H / 2, Wo = W / 2;
    const long M = (long)B * Ho * Wo;
    // weights: 64 rows x 24 chunks, swizzled chunk ^= row & 7
    for (int idx = tid; idx < 64 * 24; idx += 256) {
        const int n = idx / 24, ch = idx % 24;
        *reinterpret_cast<uint4*>(sW + n * LDA + ((ch ^ (n & 7)) * 16)) = *reinterpret_cast<const uint4*>(wst + (n * 24 + ch) * 8);
    }
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long m0 = (long)tile * 128;
        __syncthreads();
        // ---- im2col: thread -> row (tid & 127), chunks (tid >> 7) + 2*j
        {
            const int row = tid & 127;
            const long m = m0 + row;
            const bool okm = m < M;
            const long mm = okm ? m : 0;
            const int b = (int)(mm / ((long)Ho * Wo)), rem = (int)(mm % ((long)Ho * Wo));
            const int ho = rem / Wo, wo = rem % Wo;
            for (int ch = (tid >> 7); ch < 24; ch += 2) {
                unsigned wv[4] = {0, 0, 0, 0};
                if (okm && ch < 21) {
                    const int r = ch / 3, c = ch % 3;
                    const int hi = 2 * ho - 3 + r;
                    if (hi >= 0 && hi < H) {
                        const float* src = img + (((long)b * 3 + c) * H + hi) * W;
                        float v[8];
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            const int wi = 2 * wo - 3 + s;
                            v[s] = (s < 7 && wi >= 0 && wi < W) ? src[wi] : 0.f;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) wv[k] = (unsigned)f2bfw(v[2 * k]) | ((unsigned)f2bfw(v[2 * k + 1]) << 16);
                    }
                }
                *reinterpret_cast<uint4*>(sA + row * LDA + ((ch ^ (row & 7)) * 16)) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
        }
        __syncthreads();
        f32x4 acc[4][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][0] = acc[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            bf16x8 af[4], bfr[2];
            const int chunk = ((ks * 4 + fq) ^ (fr & 7)) * 16;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(sA + (wm * 64 + mi * 16 + fr) * LDA + chunk);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(sW + (wn * 32 + ni * 16 + fr) * LDA + chunk);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
        // ---- epilogue: bf16 rounding, BN partials, staged 16-byte stores (output row = 128 bytes)
        constexpr int LDO = 144;
        char* sO = sA;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float s1 = 0.f, s2 = 0.f;
            const int col = wn * 32 + ni * 16 + fr;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16_t h = f2bfw(acc[mi][ni][j]);
                    const float v = bf2fw(h);
                    s1 += v; s2 += v * v;
                    *reinterpret_cast<bf16_t*>(sO + (wm * 64 + mi * 16 + fq * 4 + j) * LDO + col * 2) = h;
                }
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (fq == 0) { sStat[wm][0][col] = s1; sStat[wm][1][col] = s2; }
        }
        __syncthreads();
        if (stat_part && tid < 128) {
            const int which = tid >> 6, col = tid & 63;
            atomicAdd(&stat_part[((long)(tile % stat_rows) * 2 + which) * 64 + col], sStat[0][which][col] + sStat[1][which][col]);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx >> 3, ch = idx & 7;
            const long m = m0 + row;
            if (m < M) *reinterpret_cast<uint4*>(out + m * 64 + ch * 8) = *reinterpret_cast<const uint4*>(sO + row * LDO + ch * 16);
        }
    }
}

// ============================================================================= stem data gradient helpers
// torch [64][3][7][7] f32 -> conv_gemm rows [16][4][4][64] bf16: row (ph*2+pw)*3 + c, tap (dy,dx), channel ch:
//   W[ch][c][ph + 5 - 2 dy][pw + 5 - 2 dx]  (0 outside 0..6; rows 12..15 zero)
__global__ __launch_bounds__(256) void stem_dgrad_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ o) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 16 * 16 * 64) return;
    const int ch = i & 63, dx = (i >> 6) & 3, dy = (i >> 8) & 3, n = i >> 10;
    float v = 0.f;
    if (n < 12) {
        const int c = n % 3, pw = (n / 3) & 1, ph = n / 6;
        const int r = ph + 5 - 2 * dy, s = pw + 5 - 2 * dx;
        if (r >= 0 && r < 7 && s >= 0 && s < 7) v = w[((ch * 3 + c) * 7 + r) * 7 + s];
    }
    o[i] = f2bfw(v);
}

// [B*Ho*Wo][16] f32 -> NCHW f32 [B,3,2Ho,2Wo]
__global__ __launch_bounds__(256) void stem_dgrad_scatter_kernel(const float* __restrict__ t, float* __restrict__ g, int B, int Ho,
                                                                 int Wo) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int H = 2 * Ho, W = 2 * Wo;
    const long tot = (long)B * 3 * H * W;
    if (i >= tot) return;
    const int wv = (int)(i % W), hv = (int)((i / W) % H), c = (int)((i / ((long)W * H)) % 3), b = (int)(i / ((long)W * H * 3));
    const int y = hv >> 1, ph = hv & 1, x = wv >> 1, pw = wv & 1;
    g[i] = t[(((long)b * Ho + y) * Wo + x) * 16 + (ph * 2 + pw) * 3 + c];
}

}  // namespace ppv

using namespace ppv;

extern "C" {

// dW [N][R][S][Cs] f32 += wgrad (caller zeroes dW).  G [B,Ho,Wo,N] bf16, X [B,Hs,Ws,Cs] bf16; N % 128 == 0, Cs % 128 == 0.
int ppv_conv_wgrad(const void* G, const void* X, float* dW, const void* zero_page, int B, int Hs, int Ws, int Cs, int Ho,
                   int Wo, int N, int R, int S, int stride, int pad, hipStream_t stream) {
    if (!G || !X || !dW || !zero_page) return PPV_ERR_NULL;
    if (N % 128 || Cs % 128) return PPV_ERR_BAD_SIZE;
    WgradGeom g;
    g.B = B; g.Hs = Hs; g.Ws = Ws; g.Cs = Cs; g.Ho = Ho; g.Wo = Wo; g.N = N; g.R = R; g.S = S; g.st = stride; g.pad = pad;
    g.M = (long)B * Ho * Wo;
    const long stages = (g.M + 63) / 64;
    const int tiles = (N / 128) * (R * S * (Cs / 128));
    long splits = (512 + tiles - 1) / tiles;                   // ~2 workgroups per CU: every extra slice costs a full
    if (splits > stages / 8) splits = stages / 8;              // tile of f32 atomics (1.3 TB/s chip-wide); >= 8 stages each
    if (splits < 1) splits = 1;
    g.stages_per_split = (int)((stages + splits - 1) / splits);
    splits = (stages + g.stages_per_split - 1) / g.stages_per_split;
    conv_wgrad_kernel<<<dim3(N / 128, R * S * (Cs / 128), (unsigned)splits), 256, 0, stream>>>(
        (const bf16_t*)G, (const bf16_t*)X, dW, (const bf16_t*)zero_page, g);
    return ppv_last_error();
}

int ppv_wgrad_to_torch(const float* dW, float* out, int N, int C, int R, int S, hipStream_t stream) {
    if (!dW || !out) return PPV_ERR_NULL;
    const long tot = (long)N * C * R * S;
    wgrad_to_torch_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(dW, out, N, C, R, S);
    return ppv_last_error();
}

// mode 0: forward layout [64][24][8] bf16; mode 1: data-gradient layout [16][4][4][64] bf16
int ppv_stem_weight_layout(const float* w, void* out, int mode, hipStream_t stream) {
    if (!w || !out) return PPV_ERR_NULL;
    if (mode == 0) stem_weight_layout_kernel<<<(64 * 192 + 255) / 256, 256, 0, stream>>>(w, (bf16_t*)out);
    else stem_dgrad_weight_kernel<<<(16 * 16 * 64 + 255) / 256, 256, 0, stream>>>(w, (bf16_t*)out);
    return ppv_last_error();
}

// img [B,3,H,W] f32 NCHW -> raw [B,H/2,W/2,64] bf16 (+ BN partial sums [stat_rows][2][64], pre-zeroed)
int ppv_stem_conv(const float* img, const void* wst, void* out, float* stat_part, int stat_rows, int B, int H, int W,
                  hipStream_t stream) {
    if (!img || !wst || !out) return PPV_ERR_NULL;
    if (H % 2 || W % 2) return PPV_ERR_BAD_SIZE;
    const long M = (long)B * (H / 2) * (W / 2);
    const int tiles = (int)((M + 127) / 128);
    const int grid = tiles < 1024 ? tiles : 1024;
    stem_conv_kernel<<<grid, 256, 0, stream>>>(img, (const bf16_t*)wst, (bf16_t*)out, stat_part, B, H, W, tiles, stat_rows < 1 ? 1 : stat_rows);
    return ppv_last_error();
}

int ppv_stem_dgrad_scatter(const float* t, float* g, int B, int Ho, int Wo, hipStream_t stream) {
    if (!t || !g) return PPV_ERR_NULL;
    const long tot = (long)B * 3 * 4 * Ho * Wo;
    stem_dgrad_scatter_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(t, g, B, Ho, Wo);
    return ppv_last_error();
}

}  // extern "C"
