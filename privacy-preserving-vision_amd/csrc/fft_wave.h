// Wave-level (64 lanes) Stockham FFT of N = R^S points, N/R == 64, for gfx950.
//
// Register layout on entry AND exit: lane l holds element  l + 64*r  in u[r]
// (natural order, stride 64), so  fft -> pointwise multiply -> inverse fft  chains
// need no exchange at the seams and global loads/stores of u[r] are lane-contiguous.
// Between stages the wave exchanges through its own LDS scratch (N + N/R float2,
// index padded by idx >> log2(R) so both the strided writes and the unit-stride
// reads are bank-conflict free for ds_write_b64 / ds_read_b64).
#pragma once
#include <hip/hip_runtime.h>

namespace ppv {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }   // a * (+i)

template <int R> struct FftCfg;
template <> struct FftCfg<8> { static constexpr int S = 3, LOG2R = 3; };   // 512 = 8^3
template <> struct FftCfg<4> { static constexpr int S = 4, LOG2R = 2; };   // 256 = 4^4
template <> struct FftCfg<16> { static constexpr int S = 5, LOG2R = 4; };  // 1024 = 4^5: radix-4 stages, four butterflies per lane (below)

// forward (exp(-i...)) butterflies, in place, natural output order
__device__ __forceinline__ void bfly4(float2& a0, float2& a1, float2& a2, float2& a3) {
    float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
    a0 = cadd(s02, s13); a2 = csub(s02, s13); a1 = cadd(d02, d13); a3 = csub(d02, d13);
}
__device__ __forceinline__ void bfly(float2 (&u)[4]) { bfly4(u[0], u[1], u[2], u[3]); }
__device__ __forceinline__ void bfly(float2 (&u)[8]) {
    // radix-8 = two radix-4 on even/odd + twiddles W8^k
    float2 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6];
    float2 o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
    bfly4(e0, e1, e2, e3);
    bfly4(o0, o1, o2, o3);
    const float h = 0.70710678118654752440f;
    o1 = make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));        // * W8^1 = (1 - i)/sqrt2
    o2 = mul_mi(o2);                                               // * W8^2 = -i
    o3 = make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));       // * W8^3 = (-1 - i)/sqrt2
    u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
    u[1] = cadd(e1, o1); u[5] = csub(e1, o1);
    u[2] = cadd(e2, o2); u[6] = csub(e2, o2);
    u[3] = cadd(e3, o3); u[7] = csub(e3, o3);
}

template <int R> __device__ __forceinline__ int lds_pad(int i) { return i + (i >> FftCfg<R>::LOG2R); }
template <int R> constexpr int fft_scratch_elems() { return 64 * R + 64; }   // float2 per wave

// tw[t] = exp(-2 pi i t / N), t in [0, N), any address space readable by the lane (LDS copy preferred).
template <int R>
__device__ __forceinline__ void fft_wave(float2 (&u)[R], float2* __restrict__ scratch,
                                         const float2* __restrict__ tw, int lane) {
    constexpr int S = FftCfg<R>::S;
    constexpr int N = 64 * R;
    int p = 1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = lane & (p - 1);
        if (s > 0) {
            const int step = N / (p * R);
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[r * k * step]);
        }
        bfly(u);
        if (s < S - 1) {
            const int j = (lane - k) * R + k;
#pragma unroll
            for (int q = 0; q < R; ++q) scratch[lds_pad<R>(j + q * p)] = u[q];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = scratch[lds_pad<R>(lane + 64 * r)];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        p *= R;
    }
}

// N = 1024 (R = 16 registers per lane): 1024 is not a power of 16 with 64 butterflies per stage, so the transform runs as FIVE radix-4
// Stockham stages in which every lane owns four butterflies: butterfly t = lane + 64 b (b = 0..3) takes its inputs x[t + 256 r] from
// registers u[b + 4 r] (the natural layout: lane l holds element l + 64 i in u[i]) and its outputs land, after the last stage, in the
// same registers in natural order.  Used by the image (x) PSF convolution for patch sizes above 256 (the reference constructor's default
// 368: any transform length >= 2 P - 1 gives the same linear convolution, csrc/fftconv.hip).
template <>
__device__ __forceinline__ void fft_wave<16>(float2 (&u)[16], float2* __restrict__ scratch, const float2* __restrict__ tw, int lane) {
    constexpr int N = 1024;
    int p = 1;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = (lane + 64 * b) & (p - 1);
            if (s > 0) {
                const int step = N / (p * 4);
                u[b + 4] = cmul(u[b + 4], tw[k * step]);
                u[b + 8] = cmul(u[b + 8], tw[2 * k * step]);
                u[b + 12] = cmul(u[b + 12], tw[3 * k * step]);
            }
            bfly4(u[b], u[b + 4], u[b + 8], u[b + 12]);
        }
        if (s < 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int t = lane + 64 * b, k = t & (p - 1), j = (t - k) * 4 + k;
#pragma unroll
                for (int q = 0; q < 4; ++q) scratch[lds_pad<16>(j + q * p)] = u[b + 4 * q];
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = scratch[lds_pad<16>(lane + 64 * r)];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        p *= 4;
    }
}

// inverse (unnormalised): ifft(x) = conj(fft(conj(x)))
template <int R>
__device__ __forceinline__ void ifft_wave(float2 (&u)[R], float2* __restrict__ scratch,
                                          const float2* __restrict__ tw, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) u[r].y = -u[r].y;
    fft_wave<R>(u, scratch, tw, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) u[r].y = -u[r].y;
}

__device__ __forceinline__ float2 shfl2(float2 v, int src) {
    return make_float2(__shfl(v.x, src, 64), __shfl(v.y, src, 64));
}

// Hermitian partner: for the element k = lane + 64 q held in u[q], return Z[(N - k) mod N].
template <int R, int Q>
__device__ __forceinline__ float2 mirror(const float2 (&u)[R], int lane) {
    float2 v = shfl2(u[R - 1 - Q], (64 - lane) & 63);     // lane > 0: element (64-lane) + 64 (R-1-q)
    if (lane == 0) v = u[(R - Q) % R];                    // lane 0: element 64 (R-q) mod N
    return v;
}

}  // namespace ppv
