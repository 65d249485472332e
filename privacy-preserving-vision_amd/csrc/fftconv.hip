// Batched 2-D FFT image (x) PSF convolution for the learned-optics camera, gfx950.
//
// Replaces reference Image_Caption/Camera/Utils.py:251-297 (img_psf_conv + psf2otf:127-158)
// and Face-DeId/Camera/Utils.py:7-12 (conv2D) with a three-kernel real-FFT pipeline:
//
//   rows_r2c   [planes][H][W] f32  -> S1 [planes][H][N/2] c64   (two rows per complex FFT;
//                                       bin 0 packs (DC.re, Nyquist.re))
//   cols_mul   S1 -> column FFT -> x OTF^T -> inverse column FFT -> S2 [planes][Hout][N/2]
//              (16-column tile per workgroup, transposed through LDS, column kx==0 carries the
//               two real DC/Nyquist columns packed as one complex column)
//   rows_c2r   S2 -> two rows per inverse complex FFT -> real output, fused epilogue
//              (IC: |.|, crop, nearest P-1 -> P index map, sign bits, partial max;
//               FD: plain store, partial amax)
//
// The convolution geometry is "PSF centre at the origin": out(s,t) = sum psf[u,v] img[s+P/2-u, t+P/2-v],
// which equals the reference's pad-129/127 + ifftshift + crop[pad+1:-pad] pipeline (the reference's
// OTF centre at (1,1) and its crop offset cancel; proven by the delta-PSF golden in tests).
// HBM traffic per plane (N=512, P=256): 0.25 + 0.5 + 0.5 + 0.5 + 0.5 + 0.25 MB = 2.5 MB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "fft_wave.h"
#include "ppv_common.h"

namespace ppv {

// ----------------------------------------------------------------------------- rows forward
// TIN = float, or unsigned char: the data set's uint8 pixels, decoded here as x / 255 (reference Image_Caption/datasets.py:46
// `imgs / 255.`): the f32 copy of the batch (4x the bytes) never exists
template <int R, typename TIN = float>
__global__ __launch_bounds__(256) void rows_r2c_kernel(const TIN* __restrict__ in, float2* __restrict__ out,
                                                       const float2* __restrict__ twg, int planes, int H, int W,
                                                       int ppw) {
    constexpr float DEC = sizeof(TIN) == 1 ? 1.0f / 255.0f : 1.0f;
    constexpr int N = 64 * R, NH = N / 2;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[4][fft_scratch_elems<R>()];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const int ppp = (H + 1) >> 1;
    const long total = (long)planes * ppp;
    const long first = ((long)blockIdx.x * 4 + wave) * ppw;
    for (int it = 0; it < ppw; ++it) {
        const long pair = first + it;
        if (pair >= total) break;
        const int plane = (int)(pair / ppp), j = (int)(pair % ppp);
        const int r0 = 2 * j;
        const bool has_b = (r0 + 1) < H;
        const TIN* pa = in + ((long)plane * H + r0) * W;
        const TIN* pb = pa + W;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = lane + 64 * r;
            const float a = (n < W) ? (float)pa[n] * DEC : 0.f;
            const float b = (has_b && n < W) ? (float)pb[n] * DEC : 0.f;
            u[r] = make_float2(a, b);
        }
        fft_wave<R>(u, s_scr[wave], s_tw, lane);
        float2* oa = out + ((long)plane * H + r0) * NH;
        float2* ob = oa + NH;
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const float2 z = u[q];
            float2 m = shfl2(u[R - 1 - q], (64 - lane) & 63);
            if (lane == 0) m = u[(R - q) % R];
            float2 A = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
            float2 B = make_float2(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));
            if (q == 0 && lane == 0) {
                A = make_float2(z.x, u[R / 2].x);
                B = make_float2(z.y, u[R / 2].y);
            }
            const int k = lane + 64 * q;
            oa[k] = A;
            if (has_b) ob[k] = B;
        }
    }
}

// ----------------------------------------------------------------------------- column pass
// otfT layout: [C][NH+1][N]  (kx-major, ky contiguous; entry kx == NH is the Nyquist column).
template <int R>
__global__ __launch_bounds__(512, R >= 16 ? 2 : 4) void cols_mul_kernel(const float2* __restrict__ S1, float2* __restrict__ S2,
                                                       const float2* __restrict__ otfT,
                                                       const float2* __restrict__ twg, int C, int H_in, int row_off,
                                                       int H_out, int conj_otf, float scale) {
    constexpr int N = 64 * R, NH = N / 2, LD = 17;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[8][fft_scratch_elems<R>()];
    extern __shared__ __attribute__((aligned(16))) float2 s_tile[];   // max(H_in, H_out) x LD: the IC geometry (256 of 512 rows) leaves
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;    // room for two workgroups per CU (load / FFT / store overlap)
    const int plane = blockIdx.y, tile = blockIdx.x, ch = plane % C;
    for (int i = tid; i < N; i += 512) s_tw[i] = twg[i];
    for (int idx = tid; idx < H_in * 8; idx += 512) {
        const int row = idx >> 3, c4 = idx & 7;
        const float4 v = *reinterpret_cast<const float4*>(&S1[((long)plane * H_in + row) * NH + tile * 16 + c4 * 2]);
        s_tile[row * LD + c4 * 2] = make_float2(v.x, v.y);
        s_tile[row * LD + c4 * 2 + 1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    for (int cc = 0; cc < 2; ++cc) {
        const int c = wave * 2 + cc;
        const int kx = tile * 16 + c;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = lane + 64 * r;
            u[r] = (row < H_in) ? s_tile[row * LD + c] : make_float2(0.f, 0.f);
        }
        // the OTF column is requested before the transform (its address does not depend on it): an L2 round trip per column
        // otherwise sits between the forward and the inverse transform of every wave
        const float2* o = otfT + ((long)ch * (NH + 1) + kx) * N;
        float2 wq[R];
#pragma unroll
        for (int q = 0; q < R; ++q) wq[q] = o[lane + 64 * q];
        fft_wave<R>(u, s_scr[wave], s_tw, lane);
        if (kx == 0) {
            const float2* on = otfT + ((long)ch * (NH + 1) + NH) * N;
            float2 v[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const float2 z = u[q];
                float2 m = shfl2(u[R - 1 - q], (64 - lane) & 63);
                if (lane == 0) m = u[(R - q) % R];
                const float2 A = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
                const float2 B = make_float2(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));
                const int k = lane + 64 * q;
                const float2 o0 = wq[q], o1 = on[k];
                const float2 pa = conj_otf ? cmul_conj(A, o0) : cmul(A, o0);
                const float2 pb = conj_otf ? cmul_conj(B, o1) : cmul(B, o1);
                v[q] = make_float2(pa.x - pb.y, pa.y + pb.x);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) u[q] = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) u[q] = conj_otf ? cmul_conj(u[q], wq[q]) : cmul(u[q], wq[q]);
        }
        ifft_wave<R>(u, s_scr[wave], s_tw, lane);
#pragma unroll
        for (int q = 0; q < R; ++q) {                         // column c of the tile belongs to this wave alone: rows may be overwritten
            const int row = lane + 64 * q - row_off;
            if ((unsigned)row < (unsigned)H_out) s_tile[row * LD + c] = make_float2(u[q].x * scale, u[q].y * scale);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < H_out * 8; idx += 512) {
        const int row = idx >> 3, c4 = idx & 7;
        const float2 a = s_tile[row * LD + c4 * 2], b = s_tile[row * LD + c4 * 2 + 1];
        *reinterpret_cast<float4*>(&S2[((long)plane * H_out + row) * NH + tile * 16 + c4 * 2]) =
            make_float4(a.x, a.y, b.x, b.y);
    }
}

// Persistent form of cols_mul_kernel for the IC geometry (H rows of data in and out, H <= N / 2; round 5, VERDICT r4 task 7).  The kernel
// above runs load -> transform -> store once per workgroup: with two workgroups per CU a CU has requests in flight for a third of the
// time (10 GB/s per CU measured, 2.6 TB/s on the chip).  Here 2 x CUs workgroups walk the tiles: the NEXT tile's rows are fetched into
// registers (R / 2 float4 per thread) before the current tile's columns are transformed, and the current tile's stores drain while the
// next tile is parked and transformed -- loads and stores of consecutive tiles overlap the FFTs between them.
template <int R>
__global__ __launch_bounds__(512, R >= 16 ? 2 : 4) void cols_mul_pf_kernel(const float2* __restrict__ S1, float2* __restrict__ S2,
                                                                           const float2* __restrict__ otfT, const float2* __restrict__ twg,
                                                                           int C, int H, int conj_otf, float scale, int planes) {
    constexpr int N = 64 * R, NH = N / 2, LD = 17, PRE = R / 2, TPP = N / 32;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[8][fft_scratch_elems<R>()];
    extern __shared__ __attribute__((aligned(16))) float2 s_tile[];   // H x LD
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 512) s_tw[i] = twg[i];
    const long total = (long)planes * TPP;
    float4 pre[PRE];
    auto fetch = [&](long t) {
        const long plane = t / TPP;
        const int tile = (int)(t % TPP);
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int idx = i * 512 + tid, row = idx >> 3, c4 = idx & 7;
            pre[i] = row < H ? *reinterpret_cast<const float4*>(&S1[(plane * H + row) * NH + tile * 16 + c4 * 2]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    long t = blockIdx.x;
    if (t < total) fetch(t);
    for (; t < total; t += gridDim.x) {
        const long plane = t / TPP;
        const int tile = (int)(t % TPP), ch = (int)(plane % C);
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int idx = i * 512 + tid, row = idx >> 3, c4 = idx & 7;
            if (row < H) {
                s_tile[row * LD + c4 * 2] = make_float2(pre[i].x, pre[i].y);
                s_tile[row * LD + c4 * 2 + 1] = make_float2(pre[i].z, pre[i].w);
            }
        }
        __syncthreads();
        if (t + gridDim.x < total) fetch(t + gridDim.x);
        for (int cc = 0; cc < 2; ++cc) {
            const int c = wave * 2 + cc;
            const int kx = tile * 16 + c;
            float2 u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = lane + 64 * r;
                u[r] = (row < H) ? s_tile[row * LD + c] : make_float2(0.f, 0.f);
            }
            const float2* o = otfT + ((long)ch * (NH + 1) + kx) * N;      // (read after the transform: the prefetch registers of the next
            fft_wave<R>(u, s_scr[wave], s_tw, lane);                       // tile take the room the early OTF loads had in cols_mul_kernel)
            if (kx == 0) {
                const float2* on = otfT + ((long)ch * (NH + 1) + NH) * N;
                float2 v[R];
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const float2 z = u[q];
                    float2 m = shfl2(u[R - 1 - q], (64 - lane) & 63);
                    if (lane == 0) m = u[(R - q) % R];
                    const float2 A = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
                    const float2 B = make_float2(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));
                    const int k = lane + 64 * q;
                    const float2 o0 = o[k], o1 = on[k];
                    const float2 pa = conj_otf ? cmul_conj(A, o0) : cmul(A, o0);
                    const float2 pb = conj_otf ? cmul_conj(B, o1) : cmul(B, o1);
                    v[q] = make_float2(pa.x - pb.y, pa.y + pb.x);
                }
#pragma unroll
                for (int q = 0; q < R; ++q) u[q] = v[q];
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const float2 w = o[lane + 64 * q];
                    u[q] = conj_otf ? cmul_conj(u[q], w) : cmul(u[q], w);
                }
            }
            ifft_wave<R>(u, s_scr[wave], s_tw, lane);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int row = lane + 64 * q;
                if (row < H) s_tile[row * LD + c] = make_float2(u[q].x * scale, u[q].y * scale);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < H * 8; idx += 512) {
            const int row = idx >> 3, c4 = idx & 7;
            const float2 a = s_tile[row * LD + c4 * 2], b = s_tile[row * LD + c4 * 2 + 1];
            *reinterpret_cast<float4*>(&S2[(plane * H + row) * NH + tile * 16 + c4 * 2]) = make_float4(a.x, a.y, b.x, b.y);
        }
        __syncthreads();
    }
}

// the column pass of the IC geometry: one tile per workgroup (default) or the persistent prefetching form (PPV_COLS_PF=1, measured slower)
template <int R>
static void launch_cols_mul_ic(const float2* S1, float2* S2, const float2* otfT, const float2* tw, int C, int H, int conj_otf, float scale,
                               int planes, hipStream_t stream);

template <int R> static void cols_mul_lds_attr() {
    static PpvDevOnce once;                  // per device: the attribute is a per-device property
    if (once.need()) {
        // (R = 16: only the IC geometry exists -- at most N / 2 = 512 rows of data; the tile shares the CU's 160 KB with 78 KB of static LDS)
        // (a failed call surfaces as the launch's own error: ppv_last_error() of the caller)
        (void)hipFuncSetAttribute((const void*)cols_mul_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (R >= 16 ? 32 : 64) * R * 17 * (int)sizeof(float2));
        once.done();
    }
}

template <int R>
static void launch_cols_mul_ic(const float2* S1, float2* S2, const float2* otfT, const float2* tw, int C, int H, int conj_otf, float scale,
                               int planes, hipStream_t stream) {
    constexpr int N = 64 * R;
    // MEASURED (round 5, tools/bench_camera.py, B = 64): the persistent form LOSES -- fftconv forward 0.257-0.267 ms against 0.158-0.165 ms,
    // IC camera 1.20 against 1.11 ms: cols_mul_kernel already sits at the 128-VGPR cap of two workgroups per CU, the R / 2 float4 of
    // prefetch spill (232-272 bytes of scratch per lane with or without the early OTF loads).  Opt-in only (PPV_COLS_PF=1).
    static const int pf = getenv("PPV_COLS_PF") ? atoi(getenv("PPV_COLS_PF")) : 0;
    const size_t lds = (size_t)H * 17 * sizeof(float2);
    if (pf && H <= N / 2) {
        static bool done = false;
        static int wgs = 512;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)cols_mul_pf_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * R * 17 * (int)sizeof(float2));
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            wgs = (R >= 16 ? 1 : 2) * cus * (getenv("PPV_COLS_PF_MULT") ? atoi(getenv("PPV_COLS_PF_MULT")) : 1);
            done = true;
        }
        const long total = (long)planes * (N / 32);
        cols_mul_pf_kernel<R><<<(unsigned)(total < wgs ? total : wgs), 512, lds, stream>>>(S1, S2, otfT, tw, C, H, conj_otf, scale, planes);
        return;
    }
    cols_mul_lds_attr<R>();
    cols_mul_kernel<R><<<dim3(N / 32, planes), 512, lds, stream>>>(S1, S2, otfT, tw, C, H, 0, H, conj_otf, scale);
}

// ----------------------------------------------------------------------------- rows inverse
// MODE 0 (IC, Utils.py:289-295): out[plane][i][j] = |r(max(i-1,0), max(j-1,0))|, P = N/2 rows/cols,
//         sign bits of r saved for the backward pass, per-workgroup max.
// MODE 1 (FD / raw): out[plane][row][n] = r(row, n), n < N, per-workgroup max (signed).
// MODE 2 (IC adjoint): out[plane][row][n] = r(row, n), row, n < P = N/2.
template <int R, int MODE>
__global__ __launch_bounds__(256) void rows_c2r_kernel(const float2* __restrict__ S2, float* __restrict__ out,
                                                       unsigned long long* __restrict__ signs,
                                                       float* __restrict__ partial_max,
                                                       const float2* __restrict__ twg, int planes, int Hs, int ppw,
                                                       float scale, int P) {
    // P (MODE 0 / 2): patch size, P <= N / 2 (N / 2 for the 128 / 256 patches; smaller on a transform padded up to the next 64 R)
    constexpr int N = 64 * R, NH = N / 2;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[4][fft_scratch_elems<R>()];
    __shared__ float s_max[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const int ppp = (Hs + 1) >> 1;
    const long total = (long)planes * ppp;
    const long first = ((long)blockIdx.x * 4 + wave) * ppw;
    float vmax = (MODE == 0) ? 0.f : -INFINITY;
    for (int it = 0; it < ppw; ++it) {
        const long pair = first + it;
        if (pair >= total) break;
        const int plane = (int)(pair / ppp), j = (int)(pair % ppp);
        const int r0 = 2 * j, r1 = r0 + 1;
        const bool has_b = r1 < Hs;
        const float2* rowA = S2 + ((long)plane * Hs + r0) * NH;
        const float2* rowB = rowA + NH;
        const float2 zero = make_float2(0.f, 0.f);
        float2 u[R];
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const int k = lane + 64 * q;
            const float2 A = rowA[k], B = has_b ? rowB[k] : zero;
            u[q] = make_float2(A.x - B.y, A.y + B.x);
            if (q == 0 && lane == 0) u[q] = make_float2(A.x, B.x);
        }
#pragma unroll
        for (int q = R / 2; q < R; ++q) {
            const int kk = N - (lane + 64 * q);           // in [1, NH]
            if (kk == NH) {                                // Nyquist: lane 0, q == R/2
                const float2 A = rowA[0], B = has_b ? rowB[0] : zero;
                u[q] = make_float2(A.y, B.y);
            } else {
                const float2 A = rowA[kk], B = has_b ? rowB[kk] : zero;
                u[q] = make_float2(A.x + B.y, B.x - A.y);
            }
        }
        ifft_wave<R>(u, s_scr[wave], s_tw, lane);
        if (MODE == 0) {
            float* op = out + (long)plane * P * P;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int s = half ? r1 : r0;
#pragma unroll
                for (int q = 0; q < R / 2; ++q) {
                    const int t = lane + 64 * q;
                    const float r = (half ? u[q].y : u[q].x) * scale;
                    const unsigned long long neg = __ballot(r < 0.f);
                    if (s <= P - 2) {
                        if (lane == 0) signs[((long)plane * P + s) * (R / 2) + q] = neg;
                        const float v = fabsf(r);
                        if (t <= P - 2) {
                            vmax = fmaxf(vmax, v);
                            float* orow = op + (long)(s + 1) * P;
                            orow[t + 1] = v;
                            if (t == 0) orow[0] = v;
                            if (s == 0) {
                                op[t + 1] = v;
                                if (t == 0) op[0] = v;
                            }
                        }
                    }
                }
            }
        } else if (MODE == 2) {
            float* op = out + ((long)plane * P + r0) * P;
#pragma unroll
            for (int q = 0; q < R / 2; ++q) {
                const int n = lane + 64 * q;
                if (n < P) {
                    op[n] = u[q].x * scale;
                    if (has_b) op[P + n] = u[q].y * scale;
                }
            }
        } else {
            float* op = out + ((long)plane * Hs + r0) * N;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int n = lane + 64 * q;
                const float a = u[q].x * scale, b = u[q].y * scale;
                op[n] = a;
                vmax = fmaxf(vmax, a);
                if (has_b) {
                    op[N + n] = b;
                    vmax = fmaxf(vmax, b);
                }
            }
        }
    }
    if (partial_max) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) s_max[wave] = vmax;
        __syncthreads();
        if (tid == 0) partial_max[blockIdx.x] = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    }
}

// ----------------------------------------------------------------------------- OTF build
// emb[ch][y][x] = psf(ch, (y+P/2) mod N, (x+P/2) mod N) if both < P else 0   (PSF centre -> origin).
// psf element (ch,y,x) lives at psf[ch*sc + y*sy + x*sx] (f32 or f64: IC psf is NHWC f64 after the mask).
template <typename T>
__global__ void psf_embed_kernel(const T* __restrict__ psf, float* __restrict__ emb, int C, int P, int N, long sc,
                                 long sy, long sx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * N * N) return;
    const int x = (int)(i % N), y = (int)((i / N) % N), ch = (int)(i / ((long)N * N));
    const int yy = (y + P / 2) % N, xx = (x + P / 2) % N;
    emb[i] = (yy < P && xx < P) ? (float)psf[ch * sc + yy * sy + xx * sx] : 0.f;
}

// adjoint of psf_embed: grad_psf(ch,yy,xx) = gemb[ch][(yy - P/2) mod N][(xx - P/2) mod N]
template <typename T>
__global__ void psf_gather_kernel(const float* __restrict__ gemb, T* __restrict__ gpsf, int C, int P, int N, long sc,
                                  long sy, long sx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)C * P * P) return;
    const int xx = (int)(i % P), yy = (int)((i / P) % P), ch = (int)(i / ((long)P * P));
    const int y = (yy - P / 2 + N) % N, x = (xx - P / 2 + N) % N;
    gpsf[ch * sc + yy * sy + xx * sx] = (T)gemb[((long)ch * N + y) * N + x];
}

// one wave per (ch, kx) column of S1[ch][N][NH]: forward column FFT, written ky-contiguous.
template <int R>
__global__ __launch_bounds__(256) void cols_fwd_T_kernel(const float2* __restrict__ S1, float2* __restrict__ otfT,
                                                         const float2* __restrict__ twg, int C) {
    constexpr int N = 64 * R, NH = N / 2;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[4][fft_scratch_elems<R>()];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const int col = blockIdx.x * 4 + wave;
    if (col >= C * NH) return;
    const int ch = col / NH, kx = col % NH;
    float2 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = S1[((long)ch * N + lane + 64 * r) * NH + kx];
    fft_wave<R>(u, s_scr[wave], s_tw, lane);
    float2* o = otfT + ((long)ch * (NH + 1) + kx) * N;
    if (kx == 0) {
        float2* on = otfT + ((long)ch * (NH + 1) + NH) * N;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const float2 z = u[q];
            float2 m = shfl2(u[R - 1 - q], (64 - lane) & 63);
            if (lane == 0) m = u[(R - q) % R];
            o[lane + 64 * q] = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
            on[lane + 64 * q] = make_float2(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));
        }
    } else {
#pragma unroll
        for (int q = 0; q < R; ++q) o[lane + 64 * q] = u[q];
    }
}

// ----------------------------------------------------------------------------- backward: d/d psf
// For a batch chunk: acc[ch][kx][ky] += sum_b conj(FFTcol(X_b))[ky] * FFTcol(G_b)[ky]   (X = rows_r2c(img),
// G = rows_r2c(grad)), written to part[chunk][C][NH+1][N].  One workgroup = (kx tile, ch, chunk).
template <int R>
__global__ __launch_bounds__(512) void cols_corr_acc_kernel(const float2* __restrict__ SX, const float2* __restrict__ SG,
                                                            float2* __restrict__ part,
                                                            const float2* __restrict__ twg, int B, int C, int HX,
                                                            int HG, int bchunk) {
    // CPW columns per wave: two for the 256- / 512-point transforms (16-column tiles), one for 1024 points (8-column tiles: the two
    // operand tiles of 512 rows and the eight waves' exchange scratch must share 160 KB of LDS)
    constexpr int N = 64 * R, NH = N / 2, CPW = R >= 16 ? 1 : 2, COLS = 8 * CPW, LD = COLS + 1, F4 = COLS / 2;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[8][fft_scratch_elems<R>()];
    __shared__ float2 s_x[N / 2 * LD];      // IC path only: image and grad have P <= N/2 rows
    __shared__ float2 s_g[N / 2 * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x, ch = blockIdx.y, chunk = blockIdx.z;
    for (int i = tid; i < N; i += 512) s_tw[i] = twg[i];
    float2 acc[CPW][R], accn[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        accn[q] = make_float2(0.f, 0.f);
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) acc[cc][q] = make_float2(0.f, 0.f);
    }
    const int b0 = chunk * bchunk, b1 = min(B, b0 + bchunk);
    // the next image's two tiles travel in registers while this one's columns are transformed (HX, HG <= N/2: four 16-byte pieces each)
    constexpr int PF = (N / 2 * F4) / 512;
    float4 px[PF], pg[PF];
    auto fetch = [&](int b) {
        const long plane = (long)b * C + ch;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int idx = i * 512 + tid, row = idx / F4, c4 = idx % F4;
            px[i] = row < HX ? *reinterpret_cast<const float4*>(&SX[(plane * HX + row) * NH + tile * COLS + c4 * 2]) : make_float4(0.f, 0.f, 0.f, 0.f);
            pg[i] = row < HG ? *reinterpret_cast<const float4*>(&SG[(plane * HG + row) * NH + tile * COLS + c4 * 2]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (b0 < b1) fetch(b0);
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int idx = i * 512 + tid, row = idx / F4, c4 = idx % F4;
            s_x[row * LD + c4 * 2] = make_float2(px[i].x, px[i].y);
            s_x[row * LD + c4 * 2 + 1] = make_float2(px[i].z, px[i].w);
            s_g[row * LD + c4 * 2] = make_float2(pg[i].x, pg[i].y);
            s_g[row * LD + c4 * 2 + 1] = make_float2(pg[i].z, pg[i].w);
        }
        if (b + 1 < b1) fetch(b + 1);
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int c = wave * CPW + cc;
            const int kx = tile * COLS + c;
            float2 x[R], g[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = lane + 64 * r;
                x[r] = (row < HX) ? s_x[row * LD + c] : make_float2(0.f, 0.f);
                g[r] = (row < HG) ? s_g[row * LD + c] : make_float2(0.f, 0.f);
            }
            fft_wave<R>(x, s_scr[wave], s_tw, lane);
            fft_wave<R>(g, s_scr[wave], s_tw, lane);
            if (kx == 0) {
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    float2 mx = shfl2(x[R - 1 - q], (64 - lane) & 63), mg = shfl2(g[R - 1 - q], (64 - lane) & 63);
                    if (lane == 0) { mx = x[(R - q) % R]; mg = g[(R - q) % R]; }
                    const float2 xa = make_float2(0.5f * (x[q].x + mx.x), 0.5f * (x[q].y - mx.y));
                    const float2 xb = make_float2(0.5f * (x[q].y + mx.y), -0.5f * (x[q].x - mx.x));
                    const float2 ga = make_float2(0.5f * (g[q].x + mg.x), 0.5f * (g[q].y - mg.y));
                    const float2 gb = make_float2(0.5f * (g[q].y + mg.y), -0.5f * (g[q].x - mg.x));
                    acc[cc][q] = cadd(acc[cc][q], cmul_conj(ga, xa));
                    accn[q] = cadd(accn[q], cmul_conj(gb, xb));
                }
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) acc[cc][q] = cadd(acc[cc][q], cmul_conj(g[q], x[q]));
            }
        }
    }
    float2* pbase = part + ((long)chunk * C + ch) * (NH + 1) * N;
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
        const int kx = tile * COLS + wave * CPW + cc;
#pragma unroll
        for (int q = 0; q < R; ++q) pbase[(long)kx * N + lane + 64 * q] = acc[cc][q];
        if (kx == 0) {
#pragma unroll
            for (int q = 0; q < R; ++q) pbase[(long)NH * N + lane + 64 * q] = accn[q];
        }
    }
}

// one wave per (ch, kx): sum the chunk partials, inverse column FFT, store S2[ch][y][kx] (DC/Nyquist packed).
template <int R>
__global__ __launch_bounds__(256) void cols_inv_from_T_kernel(const float2* __restrict__ part, float2* __restrict__ S2,
                                                              const float2* __restrict__ twg, int C, int nchunk,
                                                              float scale) {
    constexpr int N = 64 * R, NH = N / 2;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[4][fft_scratch_elems<R>()];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const int col = blockIdx.x * 4 + wave;
    if (col >= C * NH) return;
    const int ch = col / NH, kx = col % NH;
    float2 u[R];
#pragma unroll
    for (int q = 0; q < R; ++q) u[q] = make_float2(0.f, 0.f);
    for (int k = 0; k < nchunk; ++k) {
        const float2* p = part + (((long)k * C + ch) * (NH + 1) + kx) * N;
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = cadd(u[q], p[lane + 64 * q]);
        if (kx == 0) {
            const float2* pn = part + (((long)k * C + ch) * (NH + 1) + NH) * N;
#pragma unroll
            for (int q = 0; q < R; ++q) {          // pack P0 + i * Pnyq
                const float2 w = pn[lane + 64 * q];
                u[q] = make_float2(u[q].x - w.y, u[q].y + w.x);
            }
        }
    }
    ifft_wave<R>(u, s_scr[wave], s_tw, lane);
#pragma unroll
    for (int q = 0; q < R; ++q)
        S2[((long)ch * N + lane + 64 * q) * NH + kx] = make_float2(u[q].x * scale, u[q].y * scale);
}

// ----------------------------------------------------------------------------- backward of Lens.py:312 + Utils.py:289-295
// dotcnt[0] += sum g * sensor ; dotcnt[1] += #(sensor == 1)      (max() backward distributes evenly over ties)
__global__ __launch_bounds__(256) void dot_count_kernel(const float* __restrict__ g, const float* __restrict__ sensor,
                                                        double* __restrict__ dotcnt, long n4) {
    __shared__ double s_red[4][2];
    double d = 0, c = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(g)[i], b = reinterpret_cast<const float4*>(sensor)[i];
        d += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
        c += (b.x == 1.f) + (b.y == 1.f) + (b.z == 1.f) + (b.w == 1.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d += __shfl_xor(d, off, 64); c += __shfl_xor(c, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = d; s_red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    if (threadIdx.x < 2)
        atomicAdd(&dotcnt[threadIdx.x], s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// gr[plane][s][t] = sign(r) * sum_{i in dup(s), j in dup(t)} (g[i][j] - [sensor[i][j]==1] dot/cnt) / M ; row/col P-1 = 0
template <int R>
__global__ __launch_bounds__(256) void ic_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ sensor,
                                                         const unsigned long long* __restrict__ signs,
                                                         const float* __restrict__ maxv, const double* __restrict__ dotcnt,
                                                         float* __restrict__ gr, long total, int P) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int t = (int)(idx % P), s = (int)((idx / P) % P);
    const long plane = idx / ((long)P * P);
    float out = 0.f;
    if (s < P - 1 && t < P - 1) {
        const float M = *maxv;
        const float share = (float)(dotcnt[0] / dotcnt[1]);
        const int i0 = (s == 0) ? 0 : s + 1, i1 = s + 1, j0 = (t == 0) ? 0 : t + 1, j1 = t + 1;
        float acc = 0.f;
        for (int i = i0; i <= i1; ++i)
            for (int j = j0; j <= j1; ++j) {
                const long o = (plane * P + i) * P + j;
                const float gv = g[o] - ((sensor[o] == 1.f) ? share : 0.f);
                acc += gv / M;
            }
        const unsigned long long bits = signs[(plane * P + s) * (R / 2) + (t >> 6)];
        out = ((bits >> (t & 63)) & 1ull) ? -acc : acc;
    }
    gr[idx] = out;
}

// ic_out_bwd_kernel fused into the row transform of the gradient (round 5): the value ic_out_bwd_kernel would store at gr[plane][s][t]
// is computed by the lane that feeds it to the FFT -- same terms, same order of additions -- and the
// [B, C, P, P] f32 tensor gr is neither written nor re-read (one launch and 2 x 4 B per pixel less).  Row s >= 1 reads row s + 1 of g /
// sensor shifted by one column (a coalesced load, 4 bytes off alignment), row 0 reads rows 0 and 1.
template <int R>
__global__ __launch_bounds__(256) void rows_r2c_icgrad_kernel(const float* __restrict__ g, const float* __restrict__ sensor,
                                                              const unsigned long long* __restrict__ signs,
                                                              const float* __restrict__ maxv, const double* __restrict__ dotcnt,
                                                              float2* __restrict__ out, const float2* __restrict__ twg, int planes, int P,
                                                              int ppw) {
    constexpr int N = 64 * R, NH = N / 2;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[4][fft_scratch_elems<R>()];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const float M = *maxv;
    const float share = (float)(dotcnt[0] / dotcnt[1]);
    const int ppp = (P + 1) >> 1;
    const long total = (long)planes * ppp;
    const long first = ((long)blockIdx.x * 4 + wave) * ppw;
    // one term (g - [sensor == 1] share) / M per source pixel; pixel (s, t) of gr collects the 1, 2 or 4 sensor pixels the reference's
    // nearest P-1 -> P index map duplicates it into: (s + 1, t + 1), column 0 too for t == 0, row 0 too for s == 0 (the interior pixel
    // takes one load of each tensor: no data-dependent loop in the common path)
    auto term = [&](long row, int j) -> float {
        const long o = row * P + j;
        return (g[o] - ((sensor[o] == 1.f) ? share : 0.f)) / M;
    };
    auto grval = [&](long plane, int s, int t) -> float {
        if (s >= P - 1 || t >= P - 1) return 0.f;
        float acc = 0.f;
        if (s == 0) {
            if (t == 0) acc += term(plane * P, 0);
            acc += term(plane * P, t + 1);
        }
        if (t == 0) acc += term(plane * P + s + 1, 0);
        acc += term(plane * P + s + 1, t + 1);
        const unsigned long long bits = signs[(plane * P + s) * (R / 2) + (t >> 6)];
        return ((bits >> (t & 63)) & 1ull) ? -acc : acc;
    };
    for (int it = 0; it < ppw; ++it) {
        const long pair = first + it;
        if (pair >= total) break;
        const long plane = pair / ppp;
        const int r0 = 2 * (int)(pair % ppp);
        const bool has_b = (r0 + 1) < P;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = lane + 64 * r;
            u[r] = make_float2(n < P ? grval(plane, r0, n) : 0.f, (has_b && n < P) ? grval(plane, r0 + 1, n) : 0.f);
        }
        fft_wave<R>(u, s_scr[wave], s_tw, lane);
        float2* oa = out + (plane * P + r0) * NH;
        float2* ob = oa + NH;
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const float2 z = u[q];
            float2 m = shfl2(u[R - 1 - q], (64 - lane) & 63);
            if (lane == 0) m = u[(R - q) % R];
            float2 A = make_float2(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));
            float2 B = make_float2(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));
            if (q == 0 && lane == 0) {
                A = make_float2(z.x, u[R / 2].x);
                B = make_float2(z.y, u[R / 2].y);
            }
            const int k = lane + 64 * q;
            oa[k] = A;
            if (has_b) ob[k] = B;
        }
    }
}

// ----------------------------------------------------------------------------- backward of the FD sensor image
// (Optics.py:126-128: circular conv + per-image amax).  dotcnt[b][0] += sum g*sensor, dotcnt[b][1] += #(sensor == 1)
__global__ __launch_bounds__(256) void dot_count_group_kernel(const float* __restrict__ g, const float* __restrict__ sensor,
                                                              double* __restrict__ dotcnt, long per_group4) {
    __shared__ double s_red[4][2];
    const long base = (long)blockIdx.y * per_group4;
    double d = 0, c = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_group4; i += (long)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(g)[base + i], b = reinterpret_cast<const float4*>(sensor)[base + i];
        d += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
        c += (b.x == 1.f) + (b.y == 1.f) + (b.z == 1.f) + (b.w == 1.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d += __shfl_xor(d, off, 64); c += __shfl_xor(c, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6][0] = d; s_red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    if (threadIdx.x < 2)
        atomicAdd(&dotcnt[blockIdx.y * 2 + threadIdx.x], s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// gr = (g - [sensor == 1] dot_b / cnt_b) / M_b
__global__ __launch_bounds__(256) void fd_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ sensor,
                                                         const float* __restrict__ maxv, const double* __restrict__ dotcnt,
                                                         float* __restrict__ gr, long per_group, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / per_group;
    const float share = (float)(dotcnt[2 * b] / dotcnt[2 * b + 1]);
    gr[i] = (g[i] - ((sensor[i] == 1.f) ? share : 0.f)) / maxv[b];
}

// as cols_corr_acc_kernel for operands with N rows (circular FD convolution): 8-column tiles, one column per wave
template <int R>
__global__ __launch_bounds__(512) void cols_corr_acc_full_kernel(const float2* __restrict__ SX, const float2* __restrict__ SG,
                                                                 float2* __restrict__ part,
                                                                 const float2* __restrict__ twg, int B, int C, int bchunk) {
    constexpr int N = 64 * R, NH = N / 2, LD = 9;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[8][fft_scratch_elems<R>()];
    __shared__ float2 s_x[N * LD];
    __shared__ float2 s_g[N * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x, ch = blockIdx.y, chunk = blockIdx.z;
    for (int i = tid; i < N; i += 512) s_tw[i] = twg[i];
    float2 acc[R], accn[R];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] = accn[q] = make_float2(0.f, 0.f);
    const int kx = tile * 8 + wave;
    const int b0 = chunk * bchunk, b1 = min(B, b0 + bchunk);
    for (int b = b0; b < b1; ++b) {
        const long plane = (long)b * C + ch;
        __syncthreads();
        for (int idx = tid; idx < N * 4; idx += 512) {
            const int row = idx >> 2, c4 = idx & 3;
            const float4 vx = *reinterpret_cast<const float4*>(&SX[(plane * N + row) * NH + tile * 8 + c4 * 2]);
            const float4 vg = *reinterpret_cast<const float4*>(&SG[(plane * N + row) * NH + tile * 8 + c4 * 2]);
            s_x[row * LD + c4 * 2] = make_float2(vx.x, vx.y); s_x[row * LD + c4 * 2 + 1] = make_float2(vx.z, vx.w);
            s_g[row * LD + c4 * 2] = make_float2(vg.x, vg.y); s_g[row * LD + c4 * 2 + 1] = make_float2(vg.z, vg.w);
        }
        __syncthreads();
        float2 x[R], g[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { x[r] = s_x[(lane + 64 * r) * LD + wave]; g[r] = s_g[(lane + 64 * r) * LD + wave]; }
        fft_wave<R>(x, s_scr[wave], s_tw, lane);
        fft_wave<R>(g, s_scr[wave], s_tw, lane);
        if (kx == 0) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                float2 mx = shfl2(x[R - 1 - q], (64 - lane) & 63), mg = shfl2(g[R - 1 - q], (64 - lane) & 63);
                if (lane == 0) { mx = x[(R - q) % R]; mg = g[(R - q) % R]; }
                const float2 xa = make_float2(0.5f * (x[q].x + mx.x), 0.5f * (x[q].y - mx.y));
                const float2 xb = make_float2(0.5f * (x[q].y + mx.y), -0.5f * (x[q].x - mx.x));
                const float2 ga = make_float2(0.5f * (g[q].x + mg.x), 0.5f * (g[q].y - mg.y));
                const float2 gb = make_float2(0.5f * (g[q].y + mg.y), -0.5f * (g[q].x - mg.x));
                acc[q] = cadd(acc[q], cmul_conj(ga, xa));
                accn[q] = cadd(accn[q], cmul_conj(gb, xb));
            }
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = cadd(acc[q], cmul_conj(g[q], x[q]));
        }
    }
    float2* pbase = part + ((long)chunk * C + ch) * (NH + 1) * N;
#pragma unroll
    for (int q = 0; q < R; ++q) pbase[(long)kx * N + lane + 64 * q] = acc[q];
    if (kx == 0) {
#pragma unroll
        for (int q = 0; q < R; ++q) pbase[(long)NH * N + lane + 64 * q] = accn[q];
    }
}

// ----------------------------------------------------------------------------- normalisation helpers
__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                        int per_group) {
    __shared__ float s[4];
    const float* p = partial + (long)blockIdx.x * per_group;
    float v = -INFINITY;
    for (int i = threadIdx.x; i < per_group; i += 256) v = fmaxf(v, p[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

// x[g][i] /= m[g]   (true division as the reference: Lens.py:312, Optics.py:128)
__global__ __launch_bounds__(256) void div_by_group_kernel(float* __restrict__ x, const float* __restrict__ m,
                                                           long per_group4, int groups) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_group4 * groups) return;
    const float d = m[i / per_group4];
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x /= d; v.y /= d; v.z /= d; v.w /= d;
    reinterpret_cast<float4*>(x)[i] = v;
}

}  // namespace ppv

// =============================================================================== host launchers
using namespace ppv;

namespace {

template <int R>
int otf_build_t(const void* psf, int psf_is_f64, long sc, long sy, long sx, int C, int P, void* otfT, void* workspace,
                hipStream_t stream) {
    constexpr int N = 64 * R;
    const float2* tw = (const float2*)ppv_twiddles_f32(N);
    if (!tw) return PPV_ERR_INIT;
    float* emb = (float*)workspace;
    float2* S1 = (float2*)((char*)workspace + (size_t)C * N * N * sizeof(float));
    const long tot = (long)C * N * N;
    const unsigned ge = (unsigned)((tot + 255) / 256);
    if (psf_is_f64)
        psf_embed_kernel<double><<<ge, 256, 0, stream>>>((const double*)psf, emb, C, P, N, sc, sy, sx);
    else
        psf_embed_kernel<float><<<ge, 256, 0, stream>>>((const float*)psf, emb, C, P, N, sc, sy, sx);
    const int ppw = 4;
    const long pairs = (long)C * (N / 2);
    rows_r2c_kernel<R><<<(unsigned)((pairs + 4 * ppw - 1) / (4 * ppw)), 256, 0, stream>>>(emb, S1, tw, C, N, N, ppw);
    cols_fwd_T_kernel<R><<<(unsigned)((C * (N / 2) + 3) / 4), 256, 0, stream>>>(S1, (float2*)otfT, tw, C);
    return ppv_last_error();
}

template <int R, typename TIN = float>
int fftconv_fwd_t(const TIN* img, const void* otfT, float* out, void* signs, float* partial_max, void* workspace,
                  int B, int C, int mode, int conj_otf, hipStream_t stream, int P = 32 * R) {
    constexpr int N = 64 * R;
    const float2* tw = (const float2*)ppv_twiddles_f32(N);
    if (!tw) return PPV_ERR_INIT;
    const int H = (mode == 0) ? P : N;          // image rows == cols (mode 0: the P x P patch inside an N-point transform, P <= N / 2)
    const int planes = B * C;
    float2* S1 = (float2*)workspace;
    float2* S2 = S1 + (size_t)planes * H * (N / 2);
    const int ppw = 4;
    const long pairs = (long)planes * ((H + 1) / 2);
    const unsigned g1 = (unsigned)((pairs + 4 * ppw - 1) / (4 * ppw));
    const float scale = 1.0f / ((float)N * (float)N);
    rows_r2c_kernel<R, TIN><<<g1, 256, 0, stream>>>(img, S1, tw, planes, H, H, ppw);
    launch_cols_mul_ic<R>(S1, S2, (const float2*)otfT, tw, C, H, conj_otf, scale, planes, stream);
    if (mode == 0)
        rows_c2r_kernel<R, 0><<<g1, 256, 0, stream>>>(S2, out, (unsigned long long*)signs, partial_max, tw, planes, H,
                                                      ppw, 1.f, P);
    else
        rows_c2r_kernel<R, 1><<<g1, 256, 0, stream>>>(S2, out, nullptr, partial_max, tw, planes, H, ppw, 1.f, 0);
    return ppv_last_error();
}

template <int R, typename TIN = float>
int fftconv_bwd_t(const TIN* img, const float* g_sensor, const float* sensor, const void* signs, const float* maxv,
                  const double* dotcnt, const void* otfT, void* g_psf, int is_f64, long sc, long sy, long sx,
                  float* g_img, void* workspace, int B, int C, hipStream_t stream, int P = 32 * R, const float2* sx_saved = nullptr) {
    // sx_saved: the forward pass's row transform of the image (S1 of fftconv_fwd_t, [planes][P][N / 2]) when the caller kept it: the
    // backward then skips recomputing it (one rows_r2c pass over the batch)
    constexpr int N = 64 * R, NH = N / 2;
    const float2* tw = (const float2*)ppv_twiddles_f32(N);
    if (!tw) return PPV_ERR_INIT;
    const int planes = B * C;
    char* wp = (char*)workspace;
    float* gr = (float*)wp;                 wp += (size_t)planes * P * P * sizeof(float);
    float2* SX = (float2*)wp;               wp += (size_t)planes * P * NH * sizeof(float2);
    float2* SG = (float2*)wp;               wp += (size_t)planes * P * NH * sizeof(float2);
    const int nchunk = B < 16 ? B : 16;
    const int bchunk = (B + nchunk - 1) / nchunk;
    float2* part = (float2*)wp;             wp += (size_t)nchunk * C * (NH + 1) * N * sizeof(float2);
    float2* S2 = (float2*)wp;               wp += (size_t)C * N * NH * sizeof(float2);
    float* gemb = (float*)wp;
    const long total = (long)planes * P * P;
    const int ppw = 4;
    const long pairs = (long)planes * ((P + 1) / 2);
    const unsigned g1 = (unsigned)((pairs + 4 * ppw - 1) / (4 * ppw));
    static const int fused_gr = getenv("PPV_IC_BWD_FUSED") ? atoi(getenv("PPV_IC_BWD_FUSED")) : 1;   // A/B: 0 = ic_out_bwd + rows_r2c (two launches, gr materialised)
    if (fused_gr) {
        rows_r2c_icgrad_kernel<R><<<g1, 256, 0, stream>>>(g_sensor, sensor, (const unsigned long long*)signs, maxv, dotcnt, SG, tw, planes, P, ppw);
    } else {
        ic_out_bwd_kernel<R><<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(g_sensor, sensor, (const unsigned long long*)signs,
                                                                                 maxv, dotcnt, gr, total, P);
        rows_r2c_kernel<R><<<g1, 256, 0, stream>>>(gr, SG, tw, planes, P, P, ppw);
    }
    if (g_psf) {
        if (!sx_saved) rows_r2c_kernel<R, TIN><<<g1, 256, 0, stream>>>(img, SX, tw, planes, P, P, ppw);
        cols_corr_acc_kernel<R><<<dim3(R >= 16 ? N / 16 : N / 32, C, (B + bchunk - 1) / bchunk), 512, 0, stream>>>(sx_saved ? sx_saved : SX, SG, part, tw, B, C, P, P,
                                                                                           bchunk);
        cols_inv_from_T_kernel<R><<<(unsigned)((C * NH + 3) / 4), 256, 0, stream>>>(part, S2, tw, C, (B + bchunk - 1) / bchunk,
                                                                                  1.0f / ((float)N * (float)N));
        const long cp = (long)C * (N / 2);
        rows_c2r_kernel<R, 1><<<(unsigned)((cp + 4 * ppw - 1) / (4 * ppw)), 256, 0, stream>>>(S2, gemb, nullptr, nullptr, tw, C, N,
                                                                                            ppw, 1.f, 0);
        const long tp = (long)C * P * P;
        if (is_f64)
            psf_gather_kernel<double><<<(unsigned)((tp + 255) / 256), 256, 0, stream>>>(gemb, (double*)g_psf, C, P, N, sc, sy, sx);
        else
            psf_gather_kernel<float><<<(unsigned)((tp + 255) / 256), 256, 0, stream>>>(gemb, (float*)g_psf, C, P, N, sc, sy, sx);
    }
    if (g_img) {   // adjoint convolution (Utils.py:285-286): SG x conj(OTF) -> plain P x P crop
        float2* S2b = SX;   // SX is free again (or unused)
        launch_cols_mul_ic<R>(SG, S2b, (const float2*)otfT, tw, C, P, 1, 1.0f / ((float)N * (float)N), planes, stream);
        rows_c2r_kernel<R, 2><<<g1, 256, 0, stream>>>(S2b, g_img, nullptr, nullptr, tw, planes, P, ppw, 1.f, P);
    }
    return ppv_last_error();
}

template <int R>
int fftconv_fd_bwd_t(const float* img, const float* g_sensor, const float* sensor, const float* maxv, float* g_psf,
                     void* workspace, int B, int C, hipStream_t stream) {
    constexpr int N = 64 * R, NH = N / 2;
    const float2* tw = (const float2*)ppv_twiddles_f32(N);
    if (!tw) return PPV_ERR_INIT;
    const int planes = B * C;
    const long per_group = (long)C * N * N, total = per_group * B;
    char* wp = (char*)workspace;
    double* dotcnt = (double*)wp;           wp += ((size_t)B * 2 * sizeof(double) + 255) & ~(size_t)255;
    float* gr = (float*)wp;                 wp += (size_t)total * sizeof(float);
    float2* SX = (float2*)wp;               wp += (size_t)planes * N * NH * sizeof(float2);
    float2* SG = (float2*)wp;               wp += (size_t)planes * N * NH * sizeof(float2);
    const int nchunk = B < 16 ? B : 16;
    const int bchunk = (B + nchunk - 1) / nchunk;
    float2* part = (float2*)wp;             wp += (size_t)nchunk * C * (NH + 1) * N * sizeof(float2);
    float2* S2 = (float2*)wp;               wp += (size_t)C * N * NH * sizeof(float2);
    float* gemb = (float*)wp;
    (void)hipMemsetAsync(dotcnt, 0, (size_t)B * 2 * sizeof(double), stream);
    dot_count_group_kernel<<<dim3(64, B), 256, 0, stream>>>(g_sensor, sensor, dotcnt, per_group / 4);
    fd_out_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(g_sensor, sensor, maxv, dotcnt, gr, per_group, total);
    const int ppw = 4;
    const long pairs = (long)planes * (N / 2);
    const unsigned g1 = (unsigned)((pairs + 4 * ppw - 1) / (4 * ppw));
    rows_r2c_kernel<R><<<g1, 256, 0, stream>>>(gr, SG, tw, planes, N, N, ppw);
    rows_r2c_kernel<R><<<g1, 256, 0, stream>>>(img, SX, tw, planes, N, N, ppw);
    const int nch = (B + bchunk - 1) / bchunk;
    cols_corr_acc_full_kernel<R><<<dim3(NH / 8, C, nch), 512, 0, stream>>>(SX, SG, part, tw, B, C, bchunk);
    cols_inv_from_T_kernel<R><<<(unsigned)((C * NH + 3) / 4), 256, 0, stream>>>(part, S2, tw, C, nch, 1.0f / ((float)N * (float)N));
    const long cp = (long)C * (N / 2);
    rows_c2r_kernel<R, 1><<<(unsigned)((cp + 4 * ppw - 1) / (4 * ppw)), 256, 0, stream>>>(S2, gemb, nullptr, nullptr, tw, C, N, ppw, 1.f, 0);
    const long tp = (long)C * N * N;
    psf_gather_kernel<float><<<(unsigned)((tp + 255) / 256), 256, 0, stream>>>(gemb, g_psf, C, N, N, (long)N * N, N, 1);
    return ppv_last_error();
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

size_t ppv_fftconv_workspace_bytes(int B, int C, int N) {
    // S1 + S2 for the batch (largest case: H = N rows) + OTF-build scratch
    const size_t planes = (size_t)B * C;
    const size_t s = planes * N * (N / 2) * sizeof(float2);
    const size_t otf_scratch = (size_t)C * N * N * sizeof(float) + (size_t)C * N * (N / 2) * sizeof(float2);
    return 2 * s + otf_scratch + 4096;
}

size_t ppv_otf_elems(int C, int N) { return (size_t)C * (N / 2 + 1) * N; }

int ppv_otf_build(const void* psf, int psf_is_f64, long sc, long sy, long sx, int C, int P, int N, void* otfT,
                  void* workspace, hipStream_t stream) {
    if (!psf || !otfT || !workspace) return PPV_ERR_NULL;
    if (P > N) return PPV_ERR_BAD_SIZE;
    if (N == 1024) return otf_build_t<16>(psf, psf_is_f64, sc, sy, sx, C, P, otfT, workspace, stream);
    if (N == 512) return otf_build_t<8>(psf, psf_is_f64, sc, sy, sx, C, P, otfT, workspace, stream);
    if (N == 256) return otf_build_t<4>(psf, psf_is_f64, sc, sy, sx, C, P, otfT, workspace, stream);
    return PPV_ERR_BAD_SIZE;
}

// ---- the IC sensor convolution for ANY even patch size P <= 512 (round 5: the reference constructor's default 368, Lens.py:21-22, and
// every other size off the 128 / 256 grid used to go through torch.fft).  img and psf have support P x P, so their linear convolution
// has support (2 P - 1)^2 and EVERY circular transform of length N >= 2 P - 1 computes it without wrap-around: the reference's 2 P-point
// result (Utils.py:251-297) equals the N-point one, N = the next of 256 / 512 / 1024, with the PSF embedded at the same place (centre at
// the origin).  Same kernels as ppv_fftconv_fwd(mode 0) / ppv_fftconv_ic_bwd with P < N / 2 rows / columns of data.
// signs: [B * C][P][N / 128] 64-bit words; partial_max: ppv_fftconv_ic_partials(B, C, P) floats; u8: img is uint8 pixels (x / 255).
int ppv_fftconv_ic_partials(int B, int C, int P) {
    const long pairs = (long)B * C * ((P + 1) / 2);
    return (int)((pairs + 15) / 16);
}
size_t ppv_fftconv_ic_workspace_bytes(int B, int C, int P, int N) {
    return 2 * (size_t)B * C * P * (N / 2) * sizeof(float2) + 4096;
}
int ppv_fftconv_ic_fwd_p(const void* img, int u8, const void* otfT, float* out, void* signs, float* partial_max, void* workspace,
                         int B, int C, int P, int N, hipStream_t stream) {
    if (!img || !otfT || !out || !workspace || !signs) return PPV_ERR_NULL;
    if (P < 2 || P % 2 || 2 * P > N) return PPV_ERR_BAD_SIZE;
#define PPV_IC_FWD(R_)                                                                                                           \
    return u8 ? fftconv_fwd_t<R_, unsigned char>((const unsigned char*)img, otfT, out, signs, partial_max, workspace, B, C, 0, 0, stream, P) \
              : fftconv_fwd_t<R_>((const float*)img, otfT, out, signs, partial_max, workspace, B, C, 0, 0, stream, P)
    if (N == 1024) PPV_IC_FWD(16);
    if (N == 512) PPV_IC_FWD(8);
    if (N == 256) PPV_IC_FWD(4);
#undef PPV_IC_FWD
    return PPV_ERR_BAD_SIZE;
}
size_t ppv_fftconv_ic_bwd_workspace_bytes_p(int B, int C, int P, int N) {
    const size_t planes = (size_t)B * C, NH = N / 2;
    const size_t nchunk = B < 16 ? B : 16;
    return planes * P * P * sizeof(float) + 2 * planes * P * NH * sizeof(float2) + nchunk * C * (NH + 1) * N * sizeof(float2) +
           (size_t)C * N * NH * sizeof(float2) + (size_t)C * N * N * sizeof(float) + 4096;
}
int ppv_fftconv_ic_bwd_p(const void* img, int u8, const float* g_sensor, const float* sensor, const void* signs, const float* maxv,
                         const double* dotcnt, const void* otfT, void* g_psf, int g_psf_is_f64, long sc, long sy, long sx,
                         float* g_img, void* workspace, const void* sx_saved, int B, int C, int P, int N, hipStream_t stream) {
    if (!img || !g_sensor || !sensor || !signs || !maxv || !dotcnt || !workspace) return PPV_ERR_NULL;
    if (g_img && (!otfT || u8)) return g_img && u8 ? PPV_ERR_BAD_SIZE : PPV_ERR_NULL;
    if (P < 2 || P % 2 || 2 * P > N) return PPV_ERR_BAD_SIZE;
#define PPV_IC_BWD(R_)                                                                                                           \
    return u8 ? fftconv_bwd_t<R_, unsigned char>((const unsigned char*)img, g_sensor, sensor, signs, maxv, dotcnt, otfT, g_psf,   \
                                                  g_psf_is_f64, sc, sy, sx, nullptr, workspace, B, C, stream, P, (const float2*)sx_saved) \
              : fftconv_bwd_t<R_>((const float*)img, g_sensor, sensor, signs, maxv, dotcnt, otfT, g_psf, g_psf_is_f64, sc, sy, sx, \
                                  g_img, workspace, B, C, stream, P, (const float2*)sx_saved)
    if (N == 1024) PPV_IC_BWD(16);
    if (N == 512) PPV_IC_BWD(8);
    if (N == 256) PPV_IC_BWD(4);
#undef PPV_IC_BWD
    return PPV_ERR_BAD_SIZE;
}

int ppv_fftconv_fwd(const float* img, const void* otfT, float* out, void* signs, float* partial_max, void* workspace,
                    int B, int C, int N, int mode, int conj_otf, hipStream_t stream) {
    if (!img || !otfT || !out || !workspace) return PPV_ERR_NULL;
    if (mode == 0 && !signs) return PPV_ERR_NULL;
    if (N == 512) return fftconv_fwd_t<8>(img, otfT, out, signs, partial_max, workspace, B, C, mode, conj_otf, stream);
    if (N == 256) return fftconv_fwd_t<4>(img, otfT, out, signs, partial_max, workspace, B, C, mode, conj_otf, stream);
    return PPV_ERR_BAD_SIZE;
}

// The same with uint8 pixels (decoded as x / 255 inside the row transform, datasets.py:46).
int ppv_fftconv_fwd_u8(const unsigned char* img, const void* otfT, float* out, void* signs, float* partial_max, void* workspace,
                       int B, int C, int N, int mode, int conj_otf, hipStream_t stream) {
    if (!img || !otfT || !out || !workspace) return PPV_ERR_NULL;
    if (mode == 0 && !signs) return PPV_ERR_NULL;
    if (N == 512) return fftconv_fwd_t<8, unsigned char>(img, otfT, out, signs, partial_max, workspace, B, C, mode, conj_otf, stream);
    if (N == 256) return fftconv_fwd_t<4, unsigned char>(img, otfT, out, signs, partial_max, workspace, B, C, mode, conj_otf, stream);
    return PPV_ERR_BAD_SIZE;
}

int ppv_fftconv_partials_per_image(int C, int N, int mode) {
    const int H = (mode == 0) ? N / 2 : N;
    return C * (H / 2) / 16;
}

int ppv_group_max(const float* partial, float* out, int groups, int per_group, hipStream_t stream) {
    if (!partial || !out) return PPV_ERR_NULL;
    group_max_kernel<<<groups, 256, 0, stream>>>(partial, out, per_group);
    return ppv_last_error();
}

int ppv_div_by_group(float* x, const float* m, long per_group, int groups, hipStream_t stream) {
    if (!x || !m) return PPV_ERR_NULL;
    if (per_group % 4) return PPV_ERR_BAD_SIZE;
    const long n4 = per_group / 4 * groups;
    div_by_group_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>(x, m, per_group / 4, groups);
    return ppv_last_error();
}

size_t ppv_fftconv_bwd_workspace_bytes(int B, int C, int N) {
    const size_t planes = (size_t)B * C, P = N / 2, NH = N / 2;
    const size_t nchunk = B < 16 ? B : 16;
    return planes * P * P * sizeof(float) + 2 * planes * P * NH * sizeof(float2) +
           nchunk * C * (NH + 1) * N * sizeof(float2) + (size_t)C * N * NH * sizeof(float2) +
           (size_t)C * N * N * sizeof(float) + 4096;
}

// dotcnt[0] = sum g*sensor, dotcnt[1] = #(sensor == 1) over n elements (n % 4 == 0); dotcnt is zeroed first.
int ppv_sensor_dot_count(const float* g, const float* sensor, double* dotcnt, long n, hipStream_t stream) {
    if (!g || !sensor || !dotcnt) return PPV_ERR_NULL;
    if (n % 4) return PPV_ERR_BAD_SIZE;
    (void)hipMemsetAsync(dotcnt, 0, 16, stream);
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    dot_count_kernel<<<(unsigned)blocks, 256, 0, stream>>>(g, sensor, dotcnt, n / 4);
    return ppv_last_error();
}

// Backward of the IC sensor image (Lens.py:290,312 + Utils.py:251-297) given g_sensor = dL/d(sensor):
//   g_psf (strided [C][P][P], f32/f64; may be null) and g_img ([B,C,P,P] f32; may be null).
int ppv_fftconv_ic_bwd(const float* img, const float* g_sensor, const float* sensor, const void* signs, const float* maxv,
                       const double* dotcnt, const void* otfT, void* g_psf, int g_psf_is_f64, long sc, long sy, long sx,
                       float* g_img, void* workspace, int B, int C, int N, hipStream_t stream) {
    if (!img || !g_sensor || !sensor || !signs || !maxv || !dotcnt || !workspace) return PPV_ERR_NULL;
    if (g_img && !otfT) return PPV_ERR_NULL;
    if (N == 512) return fftconv_bwd_t<8>(img, g_sensor, sensor, signs, maxv, dotcnt, otfT, g_psf, g_psf_is_f64, sc, sy, sx, g_img, workspace, B, C, stream);
    if (N == 256) return fftconv_bwd_t<4>(img, g_sensor, sensor, signs, maxv, dotcnt, otfT, g_psf, g_psf_is_f64, sc, sy, sx, g_img, workspace, B, C, stream);
    return PPV_ERR_BAD_SIZE;
}

// ppv_fftconv_ic_bwd with uint8 pixels (no gradient w.r.t. the image: g_img must be null)
int ppv_fftconv_ic_bwd_u8(const unsigned char* img, const float* g_sensor, const float* sensor, const void* signs, const float* maxv,
                          const double* dotcnt, const void* otfT, void* g_psf, int g_psf_is_f64, long sc, long sy, long sx,
                          float* g_img, void* workspace, int B, int C, int N, hipStream_t stream) {
    if (!img || !g_sensor || !sensor || !signs || !maxv || !dotcnt || !workspace) return PPV_ERR_NULL;
    if (g_img) return PPV_ERR_BAD_SIZE;
    if (N == 512) return fftconv_bwd_t<8, unsigned char>(img, g_sensor, sensor, signs, maxv, dotcnt, otfT, g_psf, g_psf_is_f64, sc, sy, sx, nullptr, workspace, B, C, stream);
    if (N == 256) return fftconv_bwd_t<4, unsigned char>(img, g_sensor, sensor, signs, maxv, dotcnt, otfT, g_psf, g_psf_is_f64, sc, sy, sx, nullptr, workspace, B, C, stream);
    return PPV_ERR_BAD_SIZE;
}

size_t ppv_fftconv_fd_bwd_workspace_bytes(int B, int C, int N) {
    const size_t planes = (size_t)B * C, NH = N / 2;
    const size_t nchunk = B < 16 ? B : 16;
    return 4096 + (size_t)B * 16 + planes * N * N * sizeof(float) + 2 * planes * N * NH * sizeof(float2) +
           nchunk * C * (NH + 1) * N * sizeof(float2) + (size_t)C * N * NH * sizeof(float2) + (size_t)C * N * N * sizeof(float);
}

// Backward of the FD sensor image (Optics.py:126-128) w.r.t. the PSF: img, g_sensor, sensor [B,C,N,N] f32, maxv [B] (the
// per-image maxima of the forward pass) -> g_psf [C][N][N] f32 (centre at N/2, i.e. w.r.t. the un-rolled PSF).
int ppv_fftconv_fd_bwd(const float* img, const float* g_sensor, const float* sensor, const float* maxv, float* g_psf,
                       void* workspace, int B, int C, int N, hipStream_t stream) {
    if (!img || !g_sensor || !sensor || !maxv || !g_psf || !workspace) return PPV_ERR_NULL;
    if (N == 512) return fftconv_fd_bwd_t<8>(img, g_sensor, sensor, maxv, g_psf, workspace, B, C, stream);
    if (N == 256) return fftconv_fd_bwd_t<4>(img, g_sensor, sensor, maxv, g_psf, workspace, B, C, stream);
    return PPV_ERR_BAD_SIZE;
}

}  // extern "C"
