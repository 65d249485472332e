// RAFT all-pairs correlation volume, pyramid and windowed bilinear lookup (fp32), gfx950.
//
// Replaces reference Face-DeId/RAFT/core/corr.py:12-60 (CorrBlock.__init__/__call__/corr) and the
// bilinear_sampler of RAFT/core/utils/utils.py:57-71 (grid_sample, align_corners=True, zeros padding):
//   corr_volume   corr[b][i][j] = (1/sqrt(C)) sum_c f1[b][c][i] f2[b][c][j]      (corr.py:53-60)   fp32 MFMA 16x16x4
//   avgpool2      3 pyramid levels                                               (corr.py:25-27)
//   corr_lookup   out[b][l*81 + a*9 + d][h1][w1] = bilinear(corr_l[b,h1,w1], x/2^l + dy[a], y/2^l + dx[d])
//                 -- the reference adds meshgrid(dy,dx) stacked as (...,2) to (x,y): x gets dy, y gets dx
//                 (corr.py:37-43); reproduced literally.
// The alt_cuda_corr CUDA extension vendored by the reference (never called there) is NOT translated.
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

typedef __attribute__((ext_vector_type(4))) float f32x4c;

// D[i][j] = scale * sum_k A[k][i] * B[k][j];  A = f1[b] as [C][HW], B = f2[b] as [C][HW]; 128 x 128 tile, 4 waves (2 x 2),
// K chunk 16 through LDS.  HW % 128 == 0, C % 16 == 0.
__global__ __launch_bounds__(256) void corr_volume_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                          float* __restrict__ out, int C, int HW, float scale) {
    __shared__ float sA[16][128 + 4], sB[16][128 + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.z, i0 = blockIdx.y * 128, j0 = blockIdx.x * 128;
    const float* A = f1 + (long)b * C * HW;
    const float* Bm = f2 + (long)b * C * HW;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fk = lane >> 4;
    f32x4c acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4c){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < C; k0 += 16) {
        __syncthreads();
        for (int idx = tid; idx < 16 * 32; idx += 256) {
            const int k = idx >> 5, c4 = idx & 31;
            const float4 va = *reinterpret_cast<const float4*>(A + (long)(k0 + k) * HW + i0 + c4 * 4);
            const float4 vb = *reinterpret_cast<const float4*>(Bm + (long)(k0 + k) * HW + j0 + c4 * 4);
            *reinterpret_cast<float4*>(&sA[k][c4 * 4]) = va;
            *reinterpret_cast<float4*>(&sB[k][c4 * 4]) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            float af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = sA[kk + fk][wm * 64 + a * 16 + fr];
#pragma unroll
            for (int c = 0; c < 4; ++c) bf[c] = sB[kk + fk][wn * 64 + c * 16 + fr];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
    }
    float* o = out + (long)b * HW * HW;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + wm * 64 + a * 16 + fk * 4 + j, jj = j0 + wn * 64 + c * 16 + fr;
                o[(long)i * HW + jj] = acc[a][c][j] * scale;
            }
}

// [n][H][W] -> [n][H/2][W/2]
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * Ho * Wo) return;
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
    const long m = i / ((long)Wo * Ho);
    const float* p = in + (m * H + 2 * y) * W + 2 * x;
    out[i] = (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
}

// one thread per (pixel n = b*H1*W1 + h1*W1 + w1, window entry e = a*win + d) of one level
__global__ __launch_bounds__(256) void corr_lookup_kernel(const float* __restrict__ corr, const float* __restrict__ coords,
                                                          float* __restrict__ out, int B, int H1, int W1, int Hl, int Wl,
                                                          int r, int level, int nlevels) {
    const int win = 2 * r + 1, ne = win * win;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long npix = (long)B * H1 * W1;
    if (i >= npix * ne) return;
    const long n = i % npix;                       // pixel fastest: coalesced output along w1
    const int e = (int)(i / npix);
    const int a = e / win, d = e % win;
    const int w1 = (int)(n % W1), h1 = (int)((n / W1) % H1), b = (int)(n / ((long)W1 * H1));
    const float inv = 1.0f / (float)(1 << level);
    const float cx = coords[(((long)b * 2 + 0) * H1 + h1) * W1 + w1] * inv;
    const float cy = coords[(((long)b * 2 + 1) * H1 + h1) * W1 + w1] * inv;
    float x = cx + (float)(a - r);                 // x + dy[a]   (reference quirk)
    float y = cy + (float)(d - r);                 // y + dx[d]
    // round trip through the normalised grid as the reference does (utils.py:61-65, align_corners=True)
    x = ((2.f * x / (float)(Wl - 1) - 1.f) + 1.f) * 0.5f * (float)(Wl - 1);
    y = ((2.f * y / (float)(Hl - 1) - 1.f) + 1.f) * 0.5f * (float)(Hl - 1);
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float wx1 = x - xf, wy1 = y - yf, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float* img = corr + n * (long)Hl * Wl;
    auto at = [&](int yy, int xx) -> float {
        return (yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) ? img[(long)yy * Wl + xx] : 0.f;
    };
    const float v = at(y0, x0) * wy0 * wx0 + at(y0, x0 + 1) * wy0 * wx1 + at(y0 + 1, x0) * wy1 * wx0 + at(y0 + 1, x0 + 1) * wy1 * wx1;
    out[(((long)b * nlevels * ne + (long)level * ne + e) * H1 + h1) * W1 + w1] = v;
}

// every pyramid level in one launch (grid.y = level): RAFT calls the lookup 20 times per sample, the four per-level
// launches were launch-rate bound
struct LookupLevels {
    const float* corr[8];
    int Hl[8], Wl[8];
    int n;
};
__global__ __launch_bounds__(256) void corr_lookup_all_kernel(LookupLevels lv, const float* __restrict__ coords,
                                                              float* __restrict__ out, int B, int H1, int W1, int r) {
    const int level = blockIdx.y, Hl = lv.Hl[level], Wl = lv.Wl[level];
    const float* __restrict__ corr = lv.corr[level];
    const int win = 2 * r + 1, ne = win * win;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long npix = (long)B * H1 * W1;
    if (i >= npix * ne) return;
    const long n = i % npix;
    const int e = (int)(i / npix);
    const int a = e / win, d = e % win;
    const int w1 = (int)(n % W1), h1 = (int)((n / W1) % H1), b = (int)(n / ((long)W1 * H1));
    const float inv = 1.0f / (float)(1 << level);
    const float cx = coords[(((long)b * 2 + 0) * H1 + h1) * W1 + w1] * inv;
    const float cy = coords[(((long)b * 2 + 1) * H1 + h1) * W1 + w1] * inv;
    float x = cx + (float)(a - r);                 // x + dy[a]   (reference quirk)
    float y = cy + (float)(d - r);                 // y + dx[d]
    x = ((2.f * x / (float)(Wl - 1) - 1.f) + 1.f) * 0.5f * (float)(Wl - 1);
    y = ((2.f * y / (float)(Hl - 1) - 1.f) + 1.f) * 0.5f * (float)(Hl - 1);
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float wx1 = x - xf, wy1 = y - yf, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float* img = corr + n * (long)Hl * Wl;
    auto at = [&](int yy, int xx) -> float {
        return (yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) ? img[(long)yy * Wl + xx] : 0.f;
    };
    const float v = at(y0, x0) * wy0 * wx0 + at(y0, x0 + 1) * wy0 * wx1 + at(y0 + 1, x0) * wy1 * wx0 + at(y0 + 1, x0 + 1) * wy1 * wx1;
    out[(((long)b * lv.n * ne + (long)level * ne + e) * H1 + h1) * W1 + w1] = v;
}

// ----------------------------------------------------------------------------- backward
// adjoint of corr_lookup for one level: gcorr_l[n][y][x] += bilinear weights * gout[b][level*ne + e][h1][w1]
__global__ __launch_bounds__(256) void corr_lookup_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ coords,
                                                              float* __restrict__ gcorr, int B, int H1, int W1, int Hl, int Wl,
                                                              int r, int level, int nlevels) {
    const int win = 2 * r + 1, ne = win * win;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long npix = (long)B * H1 * W1;
    if (i >= npix * ne) return;
    const long n = i % npix;
    const int e = (int)(i / npix);
    const int a = e / win, d = e % win;
    const int w1 = (int)(n % W1), h1 = (int)((n / W1) % H1), b = (int)(n / ((long)W1 * H1));
    const float inv = 1.0f / (float)(1 << level);
    float x = coords[(((long)b * 2 + 0) * H1 + h1) * W1 + w1] * inv + (float)(a - r);
    float y = coords[(((long)b * 2 + 1) * H1 + h1) * W1 + w1] * inv + (float)(d - r);
    x = ((2.f * x / (float)(Wl - 1) - 1.f) + 1.f) * 0.5f * (float)(Wl - 1);
    y = ((2.f * y / (float)(Hl - 1) - 1.f) + 1.f) * 0.5f * (float)(Hl - 1);
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float wx1 = x - xf, wy1 = y - yf, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float g = gout[(((long)b * nlevels * ne + (long)level * ne + e) * H1 + h1) * W1 + w1];
    float* img = gcorr + n * (long)Hl * Wl;
    auto add = [&](int yy, int xx, float wgt) {
        if (yy >= 0 && yy < Hl && xx >= 0 && xx < Wl) atomicAdd(&img[(long)yy * Wl + xx], g * wgt);
    };
    add(y0, x0, wy0 * wx0); add(y0, x0 + 1, wy0 * wx1); add(y0 + 1, x0, wy1 * wx0); add(y0 + 1, x0 + 1, wy1 * wx1);
}

// g_fine[n][y][x] += 0.25 * g_coarse[n][y/2][x/2]   (adjoint of avgpool2)
__global__ __launch_bounds__(256) void avgpool2_bwd_acc_kernel(const float* __restrict__ gc, float* __restrict__ gf, long n, int H, int W) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long m = i / ((long)W * H);
    gf[i] += 0.25f * gc[(m * (H / 2) + y / 2) * (W / 2) + x / 2];
}

// D[m][n] = scale * sum_k A(m,k) * B(k,n); A stored [M][K] (lda); B stored [K][N] (transB = 0) or [N][K] (transB = 1).
// 64 x 64 tile, K chunk 16, fp32 MFMA 16x16x4.  M, N % 64 == 0, K % 16 == 0.
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ D,
                                                         int K, long lda, long ldb, long ldd, int transB, float scale, long sa,
                                                         long sb, long sd) {
    __shared__ float sA[16][64 + 4], sB[16][64 + 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    A += (long)blockIdx.z * sa; Bm += (long)blockIdx.z * sb; D += (long)blockIdx.z * sd;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fk = lane >> 4;
    f32x4c acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = (f32x4c){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();
        for (int idx = tid; idx < 16 * 64; idx += 256) {
            const int k = idx & 15, m = idx >> 4;                 // A is K-contiguous: consecutive threads walk k
            sA[k][m] = A[(long)(m0 + m) * lda + k0 + k];
            if (transB) sB[k][m] = Bm[(long)(n0 + m) * ldb + k0 + k];
        }
        if (!transB)
            for (int idx = tid; idx < 16 * 64; idx += 256) {
                const int n = idx & 63, k = idx >> 6;             // B is N-contiguous
                sB[k][n] = Bm[(long)(k0 + k) * ldb + n0 + n];
            }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = sA[kk + fk][wm * 32 + a * 16 + fr];
#pragma unroll
            for (int c = 0; c < 2; ++c) bf[c] = sB[kk + fk][wn * 32 + c * 16 + fr];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                D[(long)(m0 + wm * 32 + a * 16 + fk * 4 + j) * ldd + n0 + wn * 32 + c * 16 + fr] = acc[a][c][j] * scale;
}

// ============================================================================= on-the-fly windowed correlation
// MI355X counterpart of the reference's only native code, RAFT/alt_cuda_corr/correlation_kernel.cu (corr_forward_kernel
// :18-119, corr_backward_kernel :122-256), re-thought for wave64 rather than its 4x8-thread blocks with [32][33] tiles:
//   forward  : ONE WAVE PER PIXEL; 16 lanes share a window position (256 contiguous bytes of its fmap2 row per load,
//              4 positions per trip), the pixel's fmap1 row is read from LDS, the 16-lane partial dots fold with DPP
//              row adds; the (2r+2)^2 <= 100 products sit in LDS and every lane then blends its four bilinear
//              corners for one or two of the (2r+1)^2 outputs.
//              fmap2 (<= 4 MB per level) lives in L2; nothing of size HW x HW is ever written.
//   backward : ONE WAVE PER PIXEL, LANE = 4 CHANNELS.  d corr is folded back to the window (adjoint of the blend) in LDS,
//              then the wave sweeps the window: d fmap1 accumulates in registers (plain store, the pixel is owned),
//              d fmap2 rows take one 16-byte-per-lane f32 atomic burst per window position (as the reference does).
// Layouts: fmap1 [B][H1][W1][C], fmap2 [B][H2][W2][C] f32 NHWC, coords [B][H1][W1][2] (x, y), out [B][(2r+1)^2][H1][W1],
// channel = iy + (2r+1) * ix.  coords get no gradient (the reference leaves coords_grad zero).
struct AltLevels {
    const float* f2[8];
    int H2[8], W2[8];
    int n;
};

// grid (ceil(H1*W1 / 4), B, levels): level l reads coords / 2^l (corr.py:87) and fills out[b][l][:][h1][w1]
__global__ __launch_bounds__(256) void alt_corr_fwd_kernel(const float* __restrict__ f1, AltLevels lv,
                                                           const float* __restrict__ coords, float* __restrict__ out, int H1,
                                                           int W1, int C, int r, float scale) {
    const int lvl = blockIdx.z;
    const float* __restrict__ f2 = lv.f2[lvl];
    const int H2 = lv.H2[lvl], W2 = lv.W2[lvl];
    const float cdiv = 1.0f / (float)(1 << lvl);
    extern __shared__ float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rd = 2 * r + 1, wd = rd + 1, WIN = wd * wd;
    float* sF1 = sm + wave * (C + 128);                       // [C] fmap1 row of this wave's pixel
    float* sS = sF1 + C;                                      // [<=128] window of dot products
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 4 + wave;
    if (pix >= H1 * W1) return;                               // (no block-wide barrier below: waves are independent)
    const float* f1p = f1 + ((long)b * H1 * W1 + pix) * C;
    for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(sF1 + c) = *reinterpret_cast<const float4*>(f1p + c);
    // (x / 2^l is exact in binary floating point, so this equals the reference's coords / 2**i bit for bit)
    const float x = coords[((long)b * H1 * W1 + pix) * 2] * cdiv, y = coords[((long)b * H1 * W1 + pix) * 2 + 1] * cdiv;
    const float xf = floorf(x), yf = floorf(y);
    const float dx = x - xf, dy = y - yf;
    const int x0 = (int)xf - r, y0 = (int)yf - r;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 16 lanes share one window position (256 contiguous bytes of its fmap2 row per load), 4 positions per trip: a
    // wave-load touches 8 cache lines instead of the 64 a lane-per-position walk would; the 16-lane sums are DPP row adds
    const int sub = lane & 15, grp = lane >> 4;
    for (int p0 = 0; p0 < WIN; p0 += 4) {
        const int p = p0 + grp;
        const int iy = p / wd, ix = p % wd;
        const int h2 = y0 + iy, w2 = x0 + ix;
        float s = 0.f;
        if (p < WIN && (unsigned)h2 < (unsigned)H2 && (unsigned)w2 < (unsigned)W2) {
            const float* q = f2 + (((long)b * H2 + h2) * W2 + w2) * C;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int c = sub * 4; c < C; c += 64) {
                const float4 v = *reinterpret_cast<const float4*>(q + c);
                const float4 u = *reinterpret_cast<const float4*>(sF1 + c);
                a0 = fmaf(v.x, u.x, a0); a1 = fmaf(v.y, u.y, a1); a2 = fmaf(v.z, u.z, a2); a3 = fmaf(v.w, u.w, a3);
            }
            s = (a0 + a1) + (a2 + a3);
        }
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (sub == 0 && p < WIN) sS[p] = s * scale;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* o = out + ((long)b * lv.n + lvl) * rd * rd * H1 * W1 + pix;
    for (int k = lane; k < rd * rd; k += 64) {
        const int j = k / rd, i = k % rd;                     // channel k = iy + rd * ix
        const float v = (1.f - dy) * (1.f - dx) * sS[i * wd + j] + dy * (1.f - dx) * sS[(i + 1) * wd + j] +
                        (1.f - dy) * dx * sS[i * wd + j + 1] + dy * dx * sS[(i + 1) * wd + j + 1];
        o[(long)k * H1 * W1] = v;
    }
}

__global__ __launch_bounds__(256) void alt_corr_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                           const float* __restrict__ coords, const float* __restrict__ gout,
                                                           float* __restrict__ df1, float* __restrict__ df2, int H1, int W1,
                                                           int H2, int W2, int C, int r, float scale, int lvl, int nlv) {
    __shared__ float sG[4][96], sDS[4][128];
    const float cdiv = 1.0f / (float)(1 << lvl);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rd = 2 * r + 1, wd = rd + 1, WIN = wd * wd;
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 4 + wave;
    if (pix >= H1 * W1) return;
    const float x = coords[((long)b * H1 * W1 + pix) * 2] * cdiv, y = coords[((long)b * H1 * W1 + pix) * 2 + 1] * cdiv;
    const float xf = floorf(x), yf = floorf(y);
    const float dx = x - xf, dy = y - yf;
    const int x0 = (int)xf - r, y0 = (int)yf - r;
    const float* go = gout + ((long)b * nlv + lvl) * rd * rd * H1 * W1 + pix;
    for (int k = lane; k < rd * rd; k += 64) sG[wave][k] = go[(long)k * H1 * W1] * scale;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int p = lane; p < WIN; p += 64) {                    // adjoint of the four-corner blend
        const int iy = p / wd, ix = p % wd;
        float d = 0.f;
        if (iy < rd && ix < rd) d += (1.f - dy) * (1.f - dx) * sG[wave][iy + rd * ix];
        if (iy > 0 && ix < rd) d += dy * (1.f - dx) * sG[wave][(iy - 1) + rd * ix];
        if (iy < rd && ix > 0) d += (1.f - dy) * dx * sG[wave][iy + rd * (ix - 1)];
        if (iy > 0 && ix > 0) d += dy * dx * sG[wave][(iy - 1) + rd * (ix - 1)];
        sDS[wave][p] = d;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float* f1p = f1 + ((long)b * H1 * W1 + pix) * C;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 u = *reinterpret_cast<const float4*>(f1p + c);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < WIN; ++p) {
            const int h2 = y0 + p / wd, w2 = x0 + p % wd;
            if ((unsigned)h2 >= (unsigned)H2 || (unsigned)w2 >= (unsigned)W2) continue;       // wave-uniform
            const float d = sDS[wave][p];
            const long off = (((long)b * H2 + h2) * W2 + w2) * C + c;
            if (df1) {
                const float4 v = *reinterpret_cast<const float4*>(f2 + off);
                acc.x = fmaf(d, v.x, acc.x); acc.y = fmaf(d, v.y, acc.y); acc.z = fmaf(d, v.z, acc.z); acc.w = fmaf(d, v.w, acc.w);
            }
            if (df2) {
                atomicAdd(df2 + off, d * u.x); atomicAdd(df2 + off + 1, d * u.y);
                atomicAdd(df2 + off + 2, d * u.z); atomicAdd(df2 + off + 3, d * u.w);
            }
        }
        if (df1) *reinterpret_cast<float4*>(df1 + ((long)b * H1 * W1 + pix) * C + c) = acc;
    }
}

}  // namespace ppv


extern "C" {

// corr [B][HW][HW] f32 = f1^T f2 / sqrt(C); f1, f2 [B,C,H,W] f32.  HW % 128 == 0, C % 16 == 0.
int ppv_corr_volume(const float* f1, const float* f2, float* corr, int B, int C, int HW, hipStream_t stream) {
    if (!f1 || !f2 || !corr) return PPV_ERR_NULL;
    if (HW % 128 || C % 16) return PPV_ERR_BAD_SIZE;
    ppv::corr_volume_kernel<<<dim3(HW / 128, HW / 128, B), 256, 0, stream>>>(f1, f2, corr, C, HW, 1.0f / sqrtf((float)C));
    return ppv_last_error();
}

int ppv_avgpool2(const float* in, float* out, long n, int H, int W, hipStream_t stream) {
    if (!in || !out) return PPV_ERR_NULL;
    if (H % 2 || W % 2) return PPV_ERR_BAD_SIZE;
    const long tot = n * (H / 2) * (W / 2);
    ppv::avgpool2_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(in, out, n, H, W);
    return ppv_last_error();
}

// one pyramid level: corr_l [B*H1*W1][Hl][Wl], coords [B,2,H1,W1] (x, y), out [B, nlevels*(2r+1)^2, H1, W1]
int ppv_corr_lookup(const float* corr_l, const float* coords, float* out, int B, int H1, int W1, int Hl, int Wl, int r,
                    int level, int nlevels, hipStream_t stream) {
    if (!corr_l || !coords || !out) return PPV_ERR_NULL;
    const long tot = (long)B * H1 * W1 * (2 * r + 1) * (2 * r + 1);
    ppv::corr_lookup_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(corr_l, coords, out, B, H1, W1, Hl, Wl, r, level, nlevels);
    return ppv_last_error();
}

// all levels of the lookup in one launch: corr_levels / Hl / Wl are HOST arrays of `levels` entries (<= 8)
int ppv_corr_lookup_all(const float* const* corr_levels, const int* Hl, const int* Wl, int levels, const float* coords, float* out,
                        int B, int H1, int W1, int r, hipStream_t stream) {
    if (!corr_levels || !Hl || !Wl || !coords || !out) return PPV_ERR_NULL;
    if (levels < 1 || levels > 8) return PPV_ERR_BAD_SIZE;
    ppv::LookupLevels lv;
    lv.n = levels;
    for (int i = 0; i < levels; ++i) {
        if (!corr_levels[i]) return PPV_ERR_NULL;
        lv.corr[i] = corr_levels[i]; lv.Hl[i] = Hl[i]; lv.Wl[i] = Wl[i];
    }
    const long tot = (long)B * H1 * W1 * (2 * r + 1) * (2 * r + 1);
    ppv::corr_lookup_all_kernel<<<dim3((unsigned)((tot + 255) / 256), levels), 256, 0, stream>>>(lv, coords, out, B, H1, W1, r);
    return ppv_last_error();
}

// adjoint of ppv_corr_lookup for one level; gcorr_l [B*H1*W1][Hl][Wl] is ACCUMULATED into (f32 atomics)
int ppv_corr_lookup_bwd(const float* gout, const float* coords, float* gcorr_l, int B, int H1, int W1, int Hl, int Wl, int r,
                        int level, int nlevels, hipStream_t stream) {
    if (!gout || !coords || !gcorr_l) return PPV_ERR_NULL;
    const long tot = (long)B * H1 * W1 * (2 * r + 1) * (2 * r + 1);
    ppv::corr_lookup_bwd_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(gout, coords, gcorr_l, B, H1, W1, Hl, Wl, r, level, nlevels);
    return ppv_last_error();
}

// g_fine [n][H][W] += adjoint of avgpool2 applied to g_coarse [n][H/2][W/2]
int ppv_avgpool2_bwd_acc(const float* g_coarse, float* g_fine, long n, int H, int W, hipStream_t stream) {
    if (!g_coarse || !g_fine) return PPV_ERR_NULL;
    const long tot = n * H * W;
    ppv::avgpool2_bwd_acc_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(g_coarse, g_fine, n, H, W);
    return ppv_last_error();
}

// adjoint of ppv_corr_volume: gcorr [B][HW][HW] -> g_f1, g_f2 [B,C,H,W] f32 (either may be null).  HW % 64 == 0, C % 64 == 0.
int ppv_corr_volume_bwd(const float* gcorr, const float* f1, const float* f2, float* g_f1, float* g_f2, int B, int C, int HW,
                        hipStream_t stream) {
    if (!gcorr || !f1 || !f2) return PPV_ERR_NULL;
    if (HW % 64 || C % 64) return PPV_ERR_BAD_SIZE;
    const float scale = 1.0f / sqrtf((float)C);
    const dim3 grid(HW / 64, C / 64, B);
    // g_f1[c][i] = scale * sum_j f2[c][j] * G[i][j]   (B operand stored [N = i][K = j])
    if (g_f1) ppv::sgemm_mfma_kernel<<<grid, 256, 0, stream>>>(f2, gcorr, g_f1, HW, HW, HW, HW, 1, scale, (long)C * HW, (long)HW * HW, (long)C * HW);
    // g_f2[c][j] = scale * sum_i f1[c][i] * G[i][j]   (B operand stored [K = i][N = j])
    if (g_f2) ppv::sgemm_mfma_kernel<<<grid, 256, 0, stream>>>(f1, gcorr, g_f2, HW, HW, HW, HW, 0, scale, (long)C * HW, (long)HW * HW, (long)C * HW);
    return ppv_last_error();
}

// On-the-fly windowed correlation, all pyramid levels in one launch (AlternateCorrBlock.__call__, corr.py:76-91 + the
// alt_cuda_corr forward): out [B][levels][(2r+1)^2][H1][W1] = blend of <fmap1[b,h1,w1,:], fmap2_l[b, floor(y_l)-r+iy,
// floor(x_l)-r+ix, :]> * scale with (x_l, y_l) = coords / 2^l.  fmap1 [B][H1][W1][C], fmap2_l [B][H2[l]][W2[l]][C] f32 NHWC
// (host array of `levels` device pointers), coords [B][H1][W1][2] (x, y) at level 0.  C % 4 == 0, 1 <= r <= 4, levels <= 8.
int ppv_alt_corr_fwd(const float* fmap1, const float* const* fmap2_levels, const int* H2, const int* W2, int levels,
                     const float* coords, float* out, int B, int H1, int W1, int C, int r, float scale, hipStream_t stream) {
    if (!fmap1 || !fmap2_levels || !H2 || !W2 || !coords || !out) return PPV_ERR_NULL;
    if (C % 4 || C < 4 || r < 1 || r > 4 || B < 1 || levels < 1 || levels > 8) return PPV_ERR_BAD_SIZE;
    const size_t lds = 4 * (size_t)(C + 128) * sizeof(float);
    if (lds > 64 * 1024) return PPV_ERR_BAD_SIZE;
    ppv::AltLevels lv;
    lv.n = levels;
    for (int i = 0; i < levels; ++i) {
        if (!fmap2_levels[i]) return PPV_ERR_NULL;
        lv.f2[i] = fmap2_levels[i]; lv.H2[i] = H2[i]; lv.W2[i] = W2[i];
    }
    ppv::alt_corr_fwd_kernel<<<dim3((H1 * W1 + 3) / 4, B, levels), 256, lds, stream>>>(fmap1, lv, coords, out, H1, W1, C, r, scale);
    return ppv_last_error();
}

// Adjoint of one level (alt_cuda_corr backward): gout [B][levels][(2r+1)^2][H1][W1]; d_fmap1 [B][H1][W1][C] WRITTEN with
// this level's contribution, d_fmap2 [B][H2][W2][C] ACCUMULATED with f32 atomics (caller zeroes it); either may be null.
// coords receive no gradient, as in the reference.
int ppv_alt_corr_bwd(const float* fmap1, const float* fmap2, const float* coords, const float* gout, float* d_fmap1,
                     float* d_fmap2, int B, int H1, int W1, int H2, int W2, int C, int r, float scale, int level, int levels,
                     hipStream_t stream) {
    if (!fmap1 || !fmap2 || !coords || !gout) return PPV_ERR_NULL;
    if (C % 4 || C < 4 || r < 1 || r > 4 || B < 1 || level < 0 || level >= levels) return PPV_ERR_BAD_SIZE;
    ppv::alt_corr_bwd_kernel<<<dim3((H1 * W1 + 3) / 4, B), 256, 0, stream>>>(fmap1, fmap2, coords, gout, d_fmap1, d_fmap2, H1, W1, H2,
                                                                          W2, C, r, scale, level, levels);
    return ppv_last_error();
}

}  // extern "C"
