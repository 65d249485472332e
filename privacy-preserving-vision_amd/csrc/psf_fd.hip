// PSF generation of the Face-DeId learned-optics camera (complex64 throughout), gfx950.
//
// Replaces reference Face-DeId/Camera/Optics.py:92-120 (Camera.get_psf) + the two losses of :113,:124-125.
// The reference runs fftn / ifftn over ALL THREE dims of the [3,N,N] field (Optics.py:101,105), i.e. a 3-point DFT
// across the wavelength axis on top of the 2-D FFT, with a wavelength-dependent transfer function in between, so
// the three wavelengths mix.  That is reproduced literally: DFT-3 across channels is applied point-wise
// before the 2-D FFT and its inverse after the 2-D inverse FFT.
//   fd_field    : A = roll_{-N/2}( base * cexp(k*flmb*h) * chirp1 ), then DFT-3 over channels
//   rows_c2c    : row FFTs (one row per wave, wave-level Stockham radix 4/8)
//   cols_mul_c2c: column FFT -> x chirp2^T -> inverse column FFT (16-column tiles transposed through LDS)
//   rows_c2c    : inverse row FFTs
//   fd_intensity: IDFT-3, roll_{+N/2}, x (L_sen/L_len) chirp3, |.|^2 * amp^2, total sum
//   fd_finalize : psf = raw / sum; loss_rad^2 and the two centering-loss sums
#include <hip/hip_runtime.h>
#include "fft_wave.h"
#include "ppv_common.h"

namespace ppv {

__global__ __launch_bounds__(256) void fd_field_kernel(const float* __restrict__ h, const float2* __restrict__ base,
                                                       const float2* __restrict__ chirp1, float2* __restrict__ out, int N,
                                                       float kf0, float kf1, float kf2) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // destination pixel (after the roll)
    const long npx = (long)N * N;
    if (idx >= npx) return;
    const int y = (int)(idx / N), x = (int)(idx % N);
    const long src = (long)((y + N / 2) % N) * N + ((x + N / 2) % N);   // rolled[i] = field[(i + N/2) mod N]
    const float hh = h[src];
    const float kf[3] = {kf0, kf1, kf2};
    float2 a[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s, co;
        sincosf(kf[c] * hh, &s, &co);
        const float2 b = base[c * npx + src], c1 = chirp1[c * npx + src];
        float2 v = make_float2(__fsub_rn(__fmul_rn(b.x, co), __fmul_rn(b.y, s)), __fadd_rn(__fmul_rn(b.x, s), __fmul_rn(b.y, co)));
        a[c] = make_float2(__fsub_rn(__fmul_rn(v.x, c1.x), __fmul_rn(v.y, c1.y)), __fadd_rn(__fmul_rn(v.x, c1.y), __fmul_rn(v.y, c1.x)));
    }
    // DFT-3 across channels: W = exp(-2 pi i / 3)
    const float hs = 0.86602540378443864676f;
    const float2 t1 = cadd(a[1], a[2]);
    const float2 t2 = make_float2(a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y);
    const float2 d = csub(a[1], a[2]);
    const float2 t3 = make_float2(hs * d.y, -hs * d.x);
    out[idx] = cadd(a[0], t1);
    out[npx + idx] = cadd(t2, t3);
    out[2 * npx + idx] = csub(t2, t3);
}

template <int R>
__global__ __launch_bounds__(256) void rows_c2c_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                       const float2* __restrict__ twg, long rows, int inverse, float scale) {
    constexpr int N = 64 * R;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[4][fft_scratch_elems<R>()];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < N; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    float2 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = in[row * N + lane + 64 * r];
    if (inverse) ifft_wave<R>(u, s_scr[wave], s_tw, lane);
    else fft_wave<R>(u, s_scr[wave], s_tw, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) out[row * N + lane + 64 * r] = make_float2(u[r].x * scale, u[r].y * scale);
}

// in/out [C][N][N] c64 (in place allowed per tile); mulT [C][kx][ky] c64
template <int R>
__global__ __launch_bounds__(512) void cols_mul_c2c_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                           const float2* __restrict__ mulT,
                                                           const float2* __restrict__ twg, int conj_mul) {
    constexpr int N = 64 * R, LD = 17;
    __shared__ float2 s_tw[N];
    __shared__ float2 s_scr[8][fft_scratch_elems<R>()];
    __shared__ float2 s_tile[N * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ch = blockIdx.y, tile = blockIdx.x;
    const long pbase = (long)ch * N * N;
    for (int i = tid; i < N; i += 512) s_tw[i] = twg[i];
    for (int idx = tid; idx < N * 8; idx += 512) {
        const int row = idx >> 3, c4 = idx & 7;
        const float4 v = *reinterpret_cast<const float4*>(&in[pbase + (long)row * N + tile * 16 + c4 * 2]);
        s_tile[row * LD + c4 * 2] = make_float2(v.x, v.y);
        s_tile[row * LD + c4 * 2 + 1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    for (int cc = 0; cc < 2; ++cc) {
        const int c = wave * 2 + cc, kx = tile * 16 + c;
        float2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = s_tile[(lane + 64 * r) * LD + c];
        fft_wave<R>(u, s_scr[wave], s_tw, lane);
        const float2* m = mulT + ((long)ch * N + kx) * N;
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = conj_mul ? cmul_conj(u[q], m[lane + 64 * q]) : cmul(u[q], m[lane + 64 * q]);
        ifft_wave<R>(u, s_scr[wave], s_tw, lane);
#pragma unroll
        for (int q = 0; q < R; ++q) s_tile[(lane + 64 * q) * LD + c] = u[q];
    }
    __syncthreads();
    for (int idx = tid; idx < N * 8; idx += 512) {
        const int row = idx >> 3, c4 = idx & 7;
        const float2 a = s_tile[row * LD + c4 * 2], b = s_tile[row * LD + c4 * 2 + 1];
        *reinterpret_cast<float4*>(&out[pbase + (long)row * N + tile * 16 + c4 * 2]) = make_float4(a.x, a.y, b.x, b.y);
    }
}

// v [3][N][N] = 2-D inverse-transformed field (already scaled by 1/N^2); apply IDFT-3 (1/3), roll +N/2, chirp3 * amp,
// intensity; raw[c][y][x] f32; total += sum
__global__ __launch_bounds__(256) void fd_intensity_kernel(const float2* __restrict__ v, const float2* __restrict__ chirp3,
                                                           float* __restrict__ raw, double* __restrict__ total, int N,
                                                           float lratio, float amp) {
    __shared__ double s_red[4];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // destination pixel
    const long npx = (long)N * N;
    double acc = 0;
    if (idx < npx) {
        const int y = (int)(idx / N), x = (int)(idx % N);
        const long src = (long)((y - N / 2 + N) % N) * N + ((x - N / 2 + N) % N);   // rolled[i] = f[(i - N/2) mod N]
        const float2 a0 = v[src], a1 = v[npx + src], a2 = v[2 * npx + src];
        // inverse DFT-3: conj twiddles, 1/3
        const float hs = 0.86602540378443864676f, third = 1.0f / 3.0f;
        const float2 t1 = cadd(a1, a2);
        const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
        const float2 d = csub(a1, a2);
        const float2 t3 = make_float2(-hs * d.y, hs * d.x);
        float2 f[3] = {cadd(a0, t1), cadd(t2, t3), csub(t2, t3)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float2 fc = make_float2(f[c].x * third, f[c].y * third);
            const float2 ch = chirp3[c * npx + idx];
            const float2 w = cmul(fc, ch);
            const float2 u = make_float2(lratio * w.x * amp, lratio * w.y * amp);
            const float a = sqrtf(u.x * u.x + u.y * u.y);                 // torch.abs then torch.square (Optics.py:110)
            const float I = a * a;
            raw[c * npx + idx] = I;
            acc += (double)I;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// psf = raw / total; acc[0] += sum (rho*psf)^2 ; acc[1] += sum (psf - psf rolled N/2 rows)^2 ; acc[2] += ... cols
__global__ __launch_bounds__(256) void fd_finalize_kernel(const float* __restrict__ raw, const double* __restrict__ total,
                                                          const float* __restrict__ rho, float* __restrict__ psf,
                                                          double* __restrict__ acc, int N) {
    __shared__ double s_red[4][3];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)N * N;
    double a[3] = {0, 0, 0};
    if (idx < 3 * npx) {
        const int c = (int)(idx / npx);
        const long pix = idx % npx;
        const int y = (int)(pix / N), x = (int)(pix % N);
        const float t = (float)*total;
        const float p = raw[idx] / t;
        psf[idx] = p;
        const float pr = raw[c * npx + (long)((y + N / 2) % N) * N + x] / t;
        const float pc = raw[c * npx + (long)y * N + (x + N / 2) % N] / t;
        const double r = (double)(rho[pix] * p);
        a[0] = r * r;
        a[1] = (double)(p - pr) * (double)(p - pr);
        a[2] = (double)(p - pc) * (double)(p - pc);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double v = a[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&acc[threadIdx.x], s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

template <int R>
int fd_psf_t(const float* h, const float2* base, const float2* chirp1, const float2* chirp2T, const float2* chirp3,
             const float* rho, const float* kf, float lratio, float amp, float* psf, double* acc, void* ws, hipStream_t stream) {
    constexpr int N = 64 * R;
    const float2* tw = (const float2*)ppv_twiddles_f32(N);
    if (!tw) return PPV_ERR_INIT;
    const long npx = (long)N * N;
    float2* A = (float2*)ws;
    float2* Bf = A + 3 * npx;
    float* raw = (float*)(Bf + 3 * npx);
    (void)hipMemsetAsync(acc, 0, 4 * sizeof(double), stream);
    const unsigned gp = (unsigned)((npx + 255) / 256);
    fd_field_kernel<<<gp, 256, 0, stream>>>(h, base, chirp1, A, N, kf[0], kf[1], kf[2]);
    rows_c2c_kernel<R><<<(unsigned)((3L * N + 3) / 4), 256, 0, stream>>>(A, Bf, tw, 3L * N, 0, 1.f);
    cols_mul_c2c_kernel<R><<<dim3(N / 16, 3), 512, 0, stream>>>(Bf, A, chirp2T, tw, 0);
    rows_c2c_kernel<R><<<(unsigned)((3L * N + 3) / 4), 256, 0, stream>>>(A, Bf, tw, 3L * N, 1, 1.0f / ((float)N * (float)N));
    fd_intensity_kernel<<<gp, 256, 0, stream>>>(Bf, chirp3, raw, acc + 3, N, lratio, amp);
    fd_finalize_kernel<<<(unsigned)((3 * npx + 255) / 256), 256, 0, stream>>>(raw, acc + 3, rho, psf, acc, N);
    return ppv_last_error();
}


// ----------------------------------------------------------------------------- backward (d / d Zernike coefficients)
// g_tot = g_psf + g_lr * rho^2 psf / loss_rad + g_cl * 4/(3 N^2) ((psf - R_h psf) + (psf - R_w psf)); dot += sum g_tot * psf
__global__ __launch_bounds__(256) void fd_finalize_bwd_kernel(const float* __restrict__ g_psf, const double* __restrict__ g_lr,
                                                              const double* __restrict__ g_cl, const float* __restrict__ psf,
                                                              const float* __restrict__ rho, const double* __restrict__ acc,
                                                              float* __restrict__ g_tot, double* __restrict__ dot, int N) {
    __shared__ double s_red[4];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)N * N;
    double a = 0;
    if (idx < 3 * npx) {
        const int c = (int)(idx / npx);
        const long pix = idx % npx;
        const int y = (int)(pix / N), x = (int)(pix % N);
        const float p = psf[idx];
        double g = g_psf ? (double)g_psf[idx] : 0.0;
        if (g_lr) {
            const double lr = sqrt(acc[0]);
            if (lr > 0) g += (*g_lr) * (double)rho[pix] * (double)rho[pix] * (double)p / lr;
        }
        if (g_cl) {
            const float pr = psf[c * npx + (long)((y + N / 2) % N) * N + x], pc = psf[c * npx + (long)y * N + (x + N / 2) % N];
            g += (*g_cl) * (4.0 / (3.0 * (double)npx)) * ((double)(p - pr) + (double)(p - pc));
        }
        g_tot[idx] = (float)g;
        a = g * (double)p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dot, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// g_I = (g_tot - dot) / total; u recomputed from v; g_f = lratio*amp*conj(chirp3) * 2 g_I u; adjoint of (1/3) IDFT-3 and of
// the +N/2 roll -> gv [3][N][N] (complex) at the un-rolled position
__global__ __launch_bounds__(256) void fd_intensity_bwd_kernel(const float2* __restrict__ v, const float2* __restrict__ chirp3,
                                                               const float* __restrict__ g_tot, const double* __restrict__ dot,
                                                               const double* __restrict__ total, float2* __restrict__ gv, int N,
                                                               float lratio, float amp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)N * N;
    if (idx >= npx) return;
    const int y = (int)(idx / N), x = (int)(idx % N);
    const long src = (long)((y - N / 2 + N) % N) * N + ((x - N / 2 + N) % N);
    const float2 a0 = v[src], a1 = v[npx + src], a2 = v[2 * npx + src];
    const float hs = 0.86602540378443864676f, third = 1.0f / 3.0f;
    const float2 t1 = cadd(a1, a2);
    const float2 t2 = make_float2(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
    const float2 d = csub(a1, a2);
    const float2 t3 = make_float2(-hs * d.y, hs * d.x);
    const float2 f[3] = {cadd(a0, t1), cadd(t2, t3), csub(t2, t3)};
    const float T = (float)*total, dt = (float)*dot;
    float2 gf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float2 ch = chirp3[c * npx + idx];
        const float s = lratio * amp;
        const float2 w = cmul(make_float2(f[c].x * third, f[c].y * third), ch);
        const float2 u = make_float2(s * w.x, s * w.y);
        const float gI = (g_tot[c * npx + idx] - dt) / T;
        const float2 gu = make_float2(2.f * gI * u.x, 2.f * gI * u.y);
        const float2 t = cmul_conj(gu, ch);                      // gu * conj(chirp3)
        gf[c] = make_float2(s * t.x, s * t.y);
    }
    // adjoint of f_c = (1/3) sum_k a_k w^{ck} (w = e^{+2 pi i/3}):  g_a_k = (1/3) sum_c conj(w^{ck}) g_f_c = DFT-3 / 3
    const float2 s1 = cadd(gf[1], gf[2]);
    const float2 s2 = make_float2(gf[0].x - 0.5f * s1.x, gf[0].y - 0.5f * s1.y);
    const float2 dd = csub(gf[1], gf[2]);
    const float2 s3 = make_float2(hs * dd.y, -hs * dd.x);        // -i hs (g1 - g2)
    const float2 o0 = cadd(gf[0], s1), o1 = cadd(s2, s3), o2 = csub(s2, s3);
    gv[src] = make_float2(o0.x * third, o0.y * third);
    gv[npx + src] = make_float2(o1.x * third, o1.y * third);
    gv[2 * npx + src] = make_float2(o2.x * third, o2.y * third);
}

// gB [3][N][N] = gradient w.r.t. the DFT-3'd, rolled field.  Adjoint of DFT-3 (= conj twiddles, no 1/3), un-roll, then
// g_h[src] = sum_c kf_c * Im(g_A_c * conj(A_c)),  A_c = base*cexp(kf_c h)*chirp1 recomputed
__global__ __launch_bounds__(256) void fd_field_bwd_kernel(const float2* __restrict__ gB, const float* __restrict__ h,
                                                           const float2* __restrict__ base, const float2* __restrict__ chirp1,
                                                           float* __restrict__ gh, int N, float kf0, float kf1, float kf2) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)N * N;
    if (idx >= npx) return;
    const int y = (int)(idx / N), x = (int)(idx % N);
    const long src = (long)((y + N / 2) % N) * N + ((x + N / 2) % N);
    const float2 b0 = gB[idx], b1 = gB[npx + idx], b2 = gB[2 * npx + idx];
    const float hs = 0.86602540378443864676f;
    const float2 t1 = cadd(b1, b2);
    const float2 t2 = make_float2(b0.x - 0.5f * t1.x, b0.y - 0.5f * t1.y);
    const float2 d = csub(b1, b2);
    const float2 t3 = make_float2(-hs * d.y, hs * d.x);          // +i hs (b1 - b2)
    const float2 ga[3] = {cadd(b0, t1), cadd(t2, t3), csub(t2, t3)};
    const float hh = h[src];
    const float kf[3] = {kf0, kf1, kf2};
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s, co;
        sincosf(kf[c] * hh, &s, &co);
        const float2 bb = base[c * npx + src], c1 = chirp1[c * npx + src];
        const float2 a = cmul(cmul(bb, make_float2(co, s)), c1);
        acc += kf[c] * (ga[c].y * a.x - ga[c].x * a.y);
    }
    gh[src] = acc;
}

template <int R>
int fd_psf_bwd_t(const float* g_psf, const double* g_lr, const double* g_cl, const float* psf, const float2* base,
                 const float2* chirp1, const float2* chirp2T, const float2* chirp3, const float* rho, const float* kf,
                 float lratio, float amp, const float* h, const double* acc, float* gh, void* ws, void* ws2, hipStream_t stream) {
    constexpr int N = 64 * R;
    const float2* tw = (const float2*)ppv_twiddles_f32(N);
    if (!tw) return PPV_ERR_INIT;
    const long npx = (long)N * N;
    const float2* v = (const float2*)ws + 3 * npx;                 // Bf of the forward pass: the field before IDFT-3
    float2* gv = (float2*)ws2;
    float2* tmp = gv + 3 * npx;
    float* g_tot = (float*)(tmp + 3 * npx);
    double* dot = (double*)(g_tot + 3 * npx);
    (void)hipMemsetAsync(dot, 0, sizeof(double), stream);
    const unsigned gp = (unsigned)((npx + 255) / 256), g3 = (unsigned)((3 * npx + 255) / 256);
    fd_finalize_bwd_kernel<<<g3, 256, 0, stream>>>(g_psf, g_lr, g_cl, psf, rho, acc, g_tot, dot, N);
    fd_intensity_bwd_kernel<<<gp, 256, 0, stream>>>(v, chirp3, g_tot, dot, acc + 3, gv, N, lratio, amp);
    rows_c2c_kernel<R><<<(unsigned)((3L * N + 3) / 4), 256, 0, stream>>>(gv, tmp, tw, 3L * N, 0, 1.f);
    cols_mul_c2c_kernel<R><<<dim3(N / 16, 3), 512, 0, stream>>>(tmp, gv, chirp2T, tw, 1);
    rows_c2c_kernel<R><<<(unsigned)((3L * N + 3) / 4), 256, 0, stream>>>(gv, tmp, tw, 3L * N, 1, 1.0f / ((float)N * (float)N));
    fd_field_bwd_kernel<<<gp, 256, 0, stream>>>(tmp, h, base, chirp1, gh, N, kf[0], kf[1], kf[2]);
    return ppv_last_error();
}

}  // namespace ppv

extern "C" {

size_t ppv_fd_psf_workspace_bytes(int N) { return (size_t)N * N * (6 * sizeof(float2) + 3 * sizeof(float)) + 256; }

// h [N*N] f32 height map; base = rad * (t * focus), chirp1, chirp3 [3][N][N] c64; chirp2T [3][kx][ky] c64; rho [N*N] f32;
// kf[3] HOST floats k * flmb; outputs psf [3][N][N] f32, acc[4] f64 = {sum (rho psf)^2, centering rows, centering cols, total}.
int ppv_fd_psf_fwd(const float* h, const void* base, const void* chirp1, const void* chirp2T, const void* chirp3,
                   const float* rho, const float* kf, float lratio, float amp, float* psf, double* acc, void* workspace,
                   int N, hipStream_t stream) {
    if (!h || !base || !chirp1 || !chirp2T || !chirp3 || !rho || !kf || !psf || !acc || !workspace) return PPV_ERR_NULL;
    using namespace ppv;
    if (N == 512) return fd_psf_t<8>(h, (const float2*)base, (const float2*)chirp1, (const float2*)chirp2T, (const float2*)chirp3, rho, kf, lratio, amp, psf, acc, workspace, stream);
    if (N == 256) return fd_psf_t<4>(h, (const float2*)base, (const float2*)chirp1, (const float2*)chirp2T, (const float2*)chirp3, rho, kf, lratio, amp, psf, acc, workspace, stream);
    return PPV_ERR_BAD_SIZE;
}

size_t ppv_fd_psf_bwd_workspace_bytes(int N) { return (size_t)N * N * (6 * sizeof(float2) + 3 * sizeof(float)) + 256; }

// Backward of ppv_fd_psf_fwd: g_psf [3][N][N] f32 (may be null), g_lr / g_cl device f64 scalars (grads of loss_rad and of
// the centering loss; may be null) -> gh [N*N] f32 = d/d(height map).  `workspace` must be the forward call's buffer
// (it still holds the propagated field), h the forward height map, acc the forward accumulators.
int ppv_fd_psf_bwd(const float* g_psf, const double* g_lr, const double* g_cl, const float* psf, const void* base,
                   const void* chirp1, const void* chirp2T, const void* chirp3, const float* rho, const float* kf, float lratio,
                   float amp, const float* h, const double* acc, float* gh, void* workspace, void* workspace2, int N,
                   hipStream_t stream) {
    if (!psf || !base || !chirp1 || !chirp2T || !chirp3 || !rho || !kf || !h || !acc || !gh || !workspace || !workspace2) return PPV_ERR_NULL;
    using namespace ppv;
    if (N == 512) return fd_psf_bwd_t<8>(g_psf, g_lr, g_cl, psf, (const float2*)base, (const float2*)chirp1, (const float2*)chirp2T, (const float2*)chirp3, rho, kf, lratio, amp, h, acc, gh, workspace, workspace2, stream);
    if (N == 256) return fd_psf_bwd_t<4>(g_psf, g_lr, g_cl, psf, (const float2*)base, (const float2*)chirp1, (const float2*)chirp2T, (const float2*)chirp3, rho, kf, lratio, amp, h, acc, gh, workspace, workspace2, stream);
    return PPV_ERR_BAD_SIZE;
}

}  // extern "C"
