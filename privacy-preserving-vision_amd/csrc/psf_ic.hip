// PSF generation of the Image_Caption learned-optics camera (batch independent, once per step), gfx950.
//
// Replaces reference Image_Caption/Camera/Lens.py:158-274 and the Utils.py functions it calls:
//   zernike_contract   Lens.py:176            height map  sum_k c_k Z_k          (HBM-bound: reads the basis)
//   ic_field           Utils.py:396-413,192-205,88-97 + Lens.py:210-213  noise, fp64 phase, aperture -> c128
//   mixed-radix fp64 FFT (rows / cols x H / inverse rows)  Utils.py:328-378 Fresnel propagation
//   intensity + area down-sample + normalise + mask/regulariser  Utils.py:208-248, Lens.py:239,269-274
// and their adjoints for d/d(zernike coefficients).
//
// fp64 throughout the field chain: k*sqrt(x^2+y^2+d^2) ~ 7e6 rad cannot live in fp32.
// Fields are planar per wavelength [L][RR][RR] double2.  The FFT length M = RR + 2*(RR/4)
// (1344 = 2^6*3*7 for RR = 896) is kept (padding to 2048 would change the physics); radices 2,3,4,7 as butterflies, 5,11,13,23 by direct summation.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ppv_common.h"

namespace ppv {

constexpr int MAXM = 1344;
constexpr int MAXST = 8;

struct FftPlan {
    int M;
    int nst;
    int radix[MAXST];
};

// The Fresnel transforms exist for two element types (round 5): C2 = double2 (c128, as the reference's fields are typed after the f64
// aperture mask, Utils.py:88-97) and C2 = float2 (c64: the VALUES the reference feeds the transform are c64 products -- the plate
// and the spherical wavefront are cast to c64 before they are multiplied, Utils.py:80-85 -- and its transfer function is c64, so a
// c64 transform adds 4-5e-7 to PSF, sensor image and lens gradient (tools/micro/fresnel_c64_vs_c128.py on the CPU oracle) against
// the 1e-3 tolerance; half the bytes and half the LDS per workgroup).  PPV_PSF_F32 selects (default 1).
template <typename C2> struct CT;
template <> struct CT<double2> {
    typedef double R;
    static __device__ __forceinline__ double2 mk(double x, double y) { return make_double2(x, y); }
};
template <> struct CT<float2> {
    typedef float R;
    static __device__ __forceinline__ float2 mk(float x, float y) { return make_float2(x, y); }
};
template <typename C2> __device__ __forceinline__ C2 dmul(C2 a, C2 b) { return CT<C2>::mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename C2> __device__ __forceinline__ C2 dadd(C2 a, C2 b) { return CT<C2>::mk(a.x + b.x, a.y + b.y); }
template <typename C2> __device__ __forceinline__ C2 dsub(C2 a, C2 b) { return CT<C2>::mk(a.x - b.x, a.y - b.y); }
template <typename C2> __device__ __forceinline__ C2 dmi(C2 a) { return CT<C2>::mk(a.y, -a.x); }   // * (-i)

template <typename C2> __device__ __forceinline__ void dbfly(C2 (&u)[2]) {
    const C2 a = u[0], b = u[1];
    u[0] = dadd(a, b); u[1] = dsub(a, b);
}
template <typename C2> __device__ __forceinline__ void dbfly(C2 (&u)[4]) {
    const C2 s02 = dadd(u[0], u[2]), d02 = dsub(u[0], u[2]), s13 = dadd(u[1], u[3]), d13 = dmi(dsub(u[1], u[3]));
    u[0] = dadd(s02, s13); u[2] = dsub(s02, s13); u[1] = dadd(d02, d13); u[3] = dsub(d02, d13);
}
template <typename C2> __device__ __forceinline__ void dbfly(C2 (&u)[3]) {
    typedef typename CT<C2>::R R_;
    const R_ h = (R_)0.86602540378443864676, half = (R_)0.5;
    const C2 t1 = dadd(u[1], u[2]);
    const C2 t2 = CT<C2>::mk(u[0].x - half * t1.x, u[0].y - half * t1.y);
    const C2 d = dsub(u[1], u[2]);
    const C2 t3 = CT<C2>::mk(h * d.y, -h * d.x);       // -i h (u1 - u2)
    u[0] = dadd(u[0], t1); u[1] = dadd(t2, t3); u[2] = dsub(t2, t3);
}
template <typename C2> __device__ __forceinline__ void dbfly(C2 (&u)[7]) {
    typedef typename CT<C2>::R R_;
    const R_ c1 = (R_)0.62348980185873353053, c2 = (R_)-0.22252093395631440429, c3 = (R_)-0.90096886790241912624;
    const R_ s1 = (R_)0.78183148246802980871, s2 = (R_)0.97492791218182360702, s3 = (R_)0.43388373911755812048;
    const C2 a1 = dadd(u[1], u[6]), a2 = dadd(u[2], u[5]), a3 = dadd(u[3], u[4]);
    const C2 b1 = dsub(u[1], u[6]), b2 = dsub(u[2], u[5]), b3 = dsub(u[3], u[4]);
    const C2 u0 = u[0];
    // q = 1: cos(1,2,3)  sin(1,2,3);  q = 2: cos(2,4->3,6->1) sin(2, 4->-3, 6->-1);  q = 3: cos(3,6->1,9->2) sin(3, 6->-1, 9->2)
    const C2 p1 = CT<C2>::mk(u0.x + c1 * a1.x + c2 * a2.x + c3 * a3.x, u0.y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
    const C2 p2 = CT<C2>::mk(u0.x + c2 * a1.x + c3 * a2.x + c1 * a3.x, u0.y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
    const C2 p3 = CT<C2>::mk(u0.x + c3 * a1.x + c1 * a2.x + c2 * a3.x, u0.y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
    const C2 q1 = CT<C2>::mk(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y);
    const C2 q2 = CT<C2>::mk(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y);
    const C2 q3 = CT<C2>::mk(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y);
    u[0] = dadd(dadd(u0, a1), dadd(a2, a3));
    // X[q] = p - i q ; X[7-q] = p + i q
    u[1] = CT<C2>::mk(p1.x + q1.y, p1.y - q1.x); u[6] = CT<C2>::mk(p1.x - q1.y, p1.y + q1.x);
    u[2] = CT<C2>::mk(p2.x + q2.y, p2.y - q2.x); u[5] = CT<C2>::mk(p2.x - q2.y, p2.y + q2.x);
    u[3] = CT<C2>::mk(p3.x + q3.y, p3.y - q3.x); u[4] = CT<C2>::mk(p3.x - q3.y, p3.y + q3.x);
}

// one Stockham stage over a length-M sequence in LDS (threads tid0, tid0+nthr, ... cooperate)
template <int R, typename C2>
__device__ __forceinline__ void dstage(const C2* __restrict__ in, C2* __restrict__ out,
                                       const C2* __restrict__ tw, int M, int p, int tid, int nthr) {
    const int T = M / R;
    const int step = M / (p * R);
    for (int i = tid; i < T; i += nthr) {
        const int k = i % p;
        C2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u[r] = in[i + r * T];
            if (r > 0 && p > 1) u[r] = dmul(u[r], tw[r * k * step]);
        }
        dbfly(u);
        const int j = (i - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) out[j + q * p] = u[q];
    }
}

// Stockham stage of ANY radix by direct summation (R^2 twiddled adds per butterfly): the odd primes the hand-written butterflies
// above do not cover (5, 11, 13, 23: e.g. 1104 = 1.5 x 736 = 2^4 * 3 * 23, the transform of the reference constructor's default
// wave_resolution, Lens.py:21).  W_R^j = tw[(j mod R) * M / R].
template <typename C2>
__device__ __forceinline__ void dstage_any(const C2* __restrict__ in, C2* __restrict__ out, const C2* __restrict__ tw,
                                           int M, int R, int p, int tid, int nthr) {
    const int T = M / R;
    const int step = M / (p * R), wr = M / R;
    for (int i = tid; i < T; i += nthr) {
        const int k = i % p;
        const int j = (i - k) * R + k;
        for (int q = 0; q < R; ++q) {
            C2 acc = in[i];
            int e = 0;                                         // (q * r) mod R
            for (int r = 1; r < R; ++r) {
                e += q;
                if (e >= R) e -= R;
                C2 v = in[i + r * T];
                if (p > 1) v = dmul(v, tw[r * k * step]);
                acc = dadd(acc, dmul(v, tw[e * wr]));
            }
            out[j + q * p] = acc;
        }
    }
}

// Layout of the two intermediates of the 2-D transform (T1 after the row pass, T2 after the column pass): [L][M / CB][RR][CB] -- blocks of
// CB = 8 columns (2 when M % 8 != 0).  The row passes then touch 128-byte pieces, and the column pass (two columns per workgroup) reads
// 32-byte pieces that are ADJACENT to its neighbours' inside one contiguous RR x 128-byte block, instead of 32-byte pieces 16 * M bytes
// apart (row-major [L][RR][M]: 0.64 TB/s, 181 us per pass at M = 1344).
__device__ __forceinline__ long tix(int l, int y, int kx, int RR, int M, int CB) {
    return (((long)l * (M / CB) + kx / CB) * RR + y) * CB + kx % CB;
}

// forward FFT of the sequence in buf a (scratch b); returns the buffer that holds the result.
// Callers must __syncthreads()-separate groups: `sync` = workgroup barrier functor is implicit (all threads of the
// workgroup call this together, possibly on different sequences).
template <typename C2>
__device__ __forceinline__ C2* dfft(C2* a, C2* b, const C2* tw, const FftPlan& pl, int tid,
                                         int nthr) {
    int p = 1;
    for (int s = 0; s < pl.nst; ++s) {
        const int R = pl.radix[s];
        if (R == 4) dstage<4, C2>(a, b, tw, pl.M, p, tid, nthr);
        else if (R == 2) dstage<2, C2>(a, b, tw, pl.M, p, tid, nthr);
        else if (R == 3) dstage<3, C2>(a, b, tw, pl.M, p, tid, nthr);
        else if (R == 7) dstage<7, C2>(a, b, tw, pl.M, p, tid, nthr);
        else dstage_any(a, b, tw, pl.M, R, p, tid, nthr);
        __syncthreads();
        C2* t = a; a = b; b = t;
        p *= R;
    }
    return a;
}

// The same transform with the plan of M = 1344 = 4 * 4 * 4 * 3 * 7 (RR = 896, the benchmark geometry) known at compile time: stage
// radix, stride and twiddle step are constants (no integer division by a run-time p, no radix dispatch per stage), 256 threads, and
// a thread's twiddles -- the same for every transform it takes part in -- are loaded ONCE into registers (Tw1344: 22 C2): with
// the table in global memory every butterfly waited ~0.7 us for an L2 hit inside each of the ten barrier-separated stages of the
// column pass, with the table in LDS the workgroup is 64.5 KB and only two fit a CU.
template <int R, int P, int M> struct StageGeom { static constexpr int T = M / R, STEP = M / (P * R), ITERS = (T + 255) / 256; };
template <typename C2> struct Tw1344 {
    C2 s2[2][3], s3[2][3], s4[2][2], s5[1][6];
};
template <int R, int P, int M, int NI, typename C2>
__device__ __forceinline__ void tw_load(C2 (&w)[NI][R - 1], const C2* __restrict__ tw, int tid) {
    using G = StageGeom<R, P, M>;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int i = it * 256 + tid;
        const int k = (i < G::T ? i : 0) % P;
#pragma unroll
        for (int r = 1; r < R; ++r) w[it][r - 1] = tw[r * k * G::STEP];
    }
}
template <typename C2>
__device__ __forceinline__ void tw1344_load(Tw1344<C2>& w, const C2* __restrict__ tw, int tid) {
    tw_load<4, 4, 1344, 2>(w.s2, tw, tid);
    tw_load<4, 16, 1344, 2>(w.s3, tw, tid);
    tw_load<3, 64, 1344, 2>(w.s4, tw, tid);
    tw_load<7, 192, 1344, 1>(w.s5, tw, tid);
}
template <int R, int P, int M, int NI, typename C2>
__device__ __forceinline__ void dstage_c(const C2* __restrict__ in, C2* __restrict__ out, const C2 (*w)[R - 1], int tid) {
    using G = StageGeom<R, P, M>;
    static_assert(NI == G::ITERS, "twiddle rows per thread");
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int i = it * 256 + tid;
        if (G::T % 256 != 0 && i >= G::T) break;
        const int k = i % P;
        C2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u[r] = in[i + r * G::T];
            if (r > 0 && P > 1) u[r] = dmul(u[r], w[it][r - 1]);
        }
        dbfly(u);
        const int j = (i - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) out[j + q * P] = u[q];
    }
}
template <typename C2>
__device__ __forceinline__ C2* dfft_1344(C2* a, C2* b, const Tw1344<C2>& w, int tid) {
    dstage_c<4, 1, 1344, 2>(a, b, w.s2, tid); __syncthreads();          // P = 1: no twiddles read
    dstage_c<4, 4, 1344, 2>(b, a, w.s2, tid); __syncthreads();
    dstage_c<4, 16, 1344, 2>(a, b, w.s3, tid); __syncthreads();
    dstage_c<3, 64, 1344, 2>(b, a, w.s4, tid); __syncthreads();
    dstage_c<7, 192, 1344, 1>(a, b, w.s5, tid); __syncthreads();
    return b;
}
// plan-aware entry: the compile-time form when it applies (256 threads, the make_plan() radix order 4,4,4,3,7; w loaded by
// tw1344_load when pl.M == 1344, untouched otherwise)
template <typename C2>
__device__ __forceinline__ C2* dfft256(C2* a, C2* b, const C2* tw, const Tw1344<C2>& w, const FftPlan& pl, int tid) {
    if (pl.M == 1344) return dfft_1344(a, b, w, tid);
    FftPlan q = pl;
    if (q.M < 0) q.M = -q.M;
    return dfft(a, b, tw, q, tid, 256);
}

// ----------------------------------------------------------------------------- height map
// h[px] = sum_k c[k] * Z[k][px]   (Lens.py:176; fp64 accumulate, one rounding to f32)
// support (may be null): a 256-byte header + one byte per float4 pixel group, 0 where EVERY plane of the basis is zero there (outside
// the aperture disk: 21.5 % of the 896^2 grid; ppv_ic_psf_mark_support marks it from the data, so any basis is handled): those groups
// are not read.  The header holds a magic number AND the fingerprint of the basis it was marked for (address of Z, K), written by the
// marking kernel: a state that was never marked (ppv_ic_psf_state_init clears the header), or one that is used with another basis
// buffer than the one it was marked for (a recycled state block, a reassigned zernike_volume: r3 advisor), is read in full.
constexpr unsigned long long SUPPORT_MAGIC = 0x5050565f53555050ull;
struct SupportHdr { unsigned long long magic, zptr; unsigned K, pad; };
__device__ __forceinline__ bool support_on(const unsigned char* support, const float* Z, int K) {
    if (!support) return false;
    const SupportHdr* h = reinterpret_cast<const SupportHdr*>(support);
    return h->magic == SUPPORT_MAGIC && h->zptr == (unsigned long long)(size_t)Z && h->K == (unsigned)K;
}
// ----------------------------------------------------------------------------- mirror symmetry of the basis
// Z_j(x, y) = R(rho) * {cos | sin}(m theta) on a grid that is symmetric about its centre (poppy: x_i = (i - (n-1)/2) / ((n-1)/2)):
// every plane is even or odd under x -> -x and under y -> -y, so ONE QUADRANT of each plane holds all of it.  Whether the buffer at
// hand has that property BITWISE (the basis of csrc/zernike.hip has: IEEE products and sums commute with negation; a user-supplied
// volume may not) is decided from the data, like the support mask: per plane k the marking pass keeps the set of sign pairs
// (sx, sy) under which Z[k][y][R-1-x] == sx * Z[k][y][x], Z[k][R-1-y][x] == sy * Z[k][y][x] and Z[k][R-1-y][R-1-x] == sx * sy * Z[k][y][x]
// hold for every pixel; a plane with an empty set switches the quadrant path off for this state.  With it on, the height map
// (zernike_contract_sym_kernel) and its adjoint (zernike_grad_sym_kernel) read 1/4 of the basis; the height map is bit-identical
// to the full pass (same products up to an exact sign, same summation order per pixel).
constexpr unsigned long long SYM_MAGIC = 0x5050565f53594d4dull;
struct SymHdr { unsigned long long magic; unsigned bad; unsigned K; unsigned long long zptr; };
__device__ __forceinline__ bool sym_on(const unsigned char* sym, const float* Z, int K) {
    if (!sym) return false;
    const SymHdr* h = reinterpret_cast<const SymHdr*>(sym);
    return h->magic == SYM_MAGIC && h->zptr == (unsigned long long)(size_t)Z && h->K == (unsigned)K;
}
__device__ __forceinline__ unsigned* sym_masks(unsigned char* sym) { return reinterpret_cast<unsigned*>(sym + 256); }
__device__ __forceinline__ const unsigned char* sym_cls(const unsigned char* sym, int K) { return sym + 256 + (size_t)K * 4; }

__global__ __launch_bounds__(256) void zernike_contract_kernel(const float* __restrict__ Z, const float* __restrict__ c,
                                                               float* __restrict__ h, int K, long npx4,
                                                               const unsigned char* __restrict__ support,
                                                               const unsigned char* __restrict__ sym) {
    extern __shared__ float s_c[];
    if (sym_on(sym, Z, K)) return;                       // zernike_contract_sym_kernel (launched beside this one) does the work
    for (int k = threadIdx.x; k < K; k += 256) s_c[k] = c[k];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx4) return;
    if (support_on(support, Z, K) && !support[256 + i]) {
        reinterpret_cast<float4*>(h)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4* z = reinterpret_cast<const float4*>(Z) + i;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
        const float4 v = z[(long)k * npx4];
        const float ck = s_c[k];
        a0 += (double)(ck * v.x); a1 += (double)(ck * v.y); a2 += (double)(ck * v.z); a3 += (double)(ck * v.w);
    }
    reinterpret_cast<float4*>(h)[i] = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
}

// gc_partial[wave][k] = sum_{px in the wave's 256 pixels} Z[k][px] * gh[px]: one partial row per WAVE (no block barrier in
// the K loop, four independent 16-byte loads in flight per lane), folded by sum_partials_kernel
__global__ __launch_bounds__(256) void zernike_grad_kernel(const float* __restrict__ Z, const float* __restrict__ gh,
                                                           double* __restrict__ part, int K, long npx4,
                                                           const unsigned char* __restrict__ support,
                                                           const unsigned char* __restrict__ sym) {
    if (sym_on(sym, Z, K)) return;                       // zernike_grad_sym_kernel does the work
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < npx4 && (!support_on(support, Z, K) || support[256 + i]);   // groups outside the basis' support contribute exactly zero
    const float4 g = ok ? reinterpret_cast<const float4*>(gh)[i] : make_float4(0, 0, 0, 0);
    const float4* z = reinterpret_cast<const float4*>(Z) + (ok ? i : 0);
    double* row = part + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * K;
    const int lane = threadIdx.x & 63;
    int k = 0;
    for (; k + 4 <= K; k += 4) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ok ? z[(long)(k + j) * npx4] : make_float4(0.f, 0.f, 0.f, 0.f);
        double a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = (double)v[j].x * g.x + (double)v[j].y * g.y + (double)v[j].z * g.z + (double)v[j].w * g.w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += __shfl_xor(a[j], off, 64);
        if (lane < 4) row[k + lane] = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : a[3];
    }
    for (; k < K; ++k) {
        const float4 v = ok ? z[(long)k * npx4] : make_float4(0.f, 0.f, 0.f, 0.f);
        double a = (double)v.x * g.x + (double)v.y * g.y + (double)v.z * g.z + (double)v.w * g.w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) row[k] = a;
    }
}

// support[i] = 1 iff some plane of the basis is non-zero in float4 group i
__global__ __launch_bounds__(256) void zernike_support_kernel(const float* __restrict__ Z, unsigned char* __restrict__ support, int K,
                                                              long npx4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npx4) return;
    const float4* z = reinterpret_cast<const float4*>(Z) + i;
    bool any = false;
    for (int k = 0; k < K; ++k) {
        const float4 v = z[(long)k * npx4];
        any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    support[256 + i] = any ? 1 : 0;
}
__global__ void zernike_support_seal_kernel(unsigned char* support, const float* Z, int K) {
    SupportHdr* h = reinterpret_cast<SupportHdr*>(support);
    h->zptr = (unsigned long long)(size_t)Z;
    h->K = (unsigned)K;
    h->magic = SUPPORT_MAGIC;
}

// thread <-> (row y < R/2, column pair 2j < R/2): its float2 and the three mirror images
struct QuadIdx { long q, qx, qy, qxy; bool ok; };
__device__ __forceinline__ QuadIdx quad_index(long t, int R) {
    const int half2 = R / 4;                       // float2 columns of a quadrant row
    QuadIdx r;
    r.ok = t < (long)(R / 2) * half2;
    const int y = r.ok ? (int)(t / half2) : 0, j = r.ok ? (int)(t % half2) : 0;
    const int R2 = R / 2;                          // float2 per row
    r.q = (long)y * R2 + j;
    r.qx = (long)y * R2 + (R2 - 1 - j);
    r.qy = (long)(R - 1 - y) * R2 + j;
    r.qxy = (long)(R - 1 - y) * R2 + (R2 - 1 - j);
    return r;
}

__global__ __launch_bounds__(256) void zernike_sym_check_kernel(const float* __restrict__ Z, unsigned char* __restrict__ sym, int K, int R) {
    const QuadIdx ix = quad_index((long)blockIdx.x * 256 + threadIdx.x, R);
    const long npx2 = (long)R * R / 2;
    unsigned* masks = sym_masks(sym);
    for (int k = 0; k < K; ++k) {
        const float2* z = reinterpret_cast<const float2*>(Z) + (long)k * npx2;
        unsigned m = 0xF;
        if (ix.ok) {
            const float2 v = z[ix.q], vx = z[ix.qx], vy = z[ix.qy], vxy = z[ix.qxy];
            m = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sx = (c & 1) ? -1.f : 1.f, sy = (c & 2) ? -1.f : 1.f;
                const bool okc = vx.y == sx * v.x && vx.x == sx * v.y && vy.x == sy * v.x && vy.y == sy * v.y &&
                                 vxy.y == sx * sy * v.x && vxy.x == sx * sy * v.y;
                m |= okc ? (1u << c) : 0u;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m &= (unsigned)__shfl_xor((int)m, off, 64);
        if ((threadIdx.x & 63) == 0 && m != 0xF) atomicAnd(&masks[k], m);
    }
}
__global__ __launch_bounds__(256) void zernike_sym_seal_kernel(unsigned char* sym, int K, const float* Z) {
    __shared__ unsigned s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const unsigned* masks = sym_masks(sym);
    unsigned char* cls = sym + 256 + (size_t)K * 4;
    for (int k = threadIdx.x; k < K; k += 256) {
        const unsigned m = masks[k] & 0xF;
        if (!m) atomicOr(&s_bad, 1u);
        cls[k] = (unsigned char)(m ? __ffs(m) - 1 : 0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        SymHdr* h = reinterpret_cast<SymHdr*>(sym);
        h->bad = s_bad;
        h->zptr = (unsigned long long)(size_t)Z;
        h->K = (unsigned)K;
        h->magic = s_bad ? 0ull : SYM_MAGIC;
    }
}

// h = sum_k c[k] Z[k] from one quadrant of Z (launched only when sym_on): same products and the same k order per pixel as
// zernike_contract_kernel
__global__ __launch_bounds__(256) void zernike_contract_sym_kernel(const float* __restrict__ Z, const float* __restrict__ c,
                                                                   float* __restrict__ h, int K, int R,
                                                                   const unsigned char* __restrict__ support,
                                                                   const unsigned char* __restrict__ sym) {
    extern __shared__ float s_c4[];                 // [4][K]: c, c*sx, c*sy, c*sx*sy
    if (!sym_on(sym, Z, K)) return;
    const unsigned char* cls = sym_cls(sym, K);
    for (int k = threadIdx.x; k < K; k += 256) {
        const float ck = c[k];
        const int cl = cls[k];
        const float sx = (cl & 1) ? -1.f : 1.f, sy = (cl & 2) ? -1.f : 1.f;
        s_c4[k] = ck; s_c4[K + k] = ck * sx; s_c4[2 * K + k] = ck * sy; s_c4[3 * K + k] = ck * sx * sy;
    }
    __syncthreads();
    const QuadIdx ix = quad_index((long)blockIdx.x * 256 + threadIdx.x, R);
    if (!ix.ok) return;
    float2* h2 = reinterpret_cast<float2*>(h);
    if (support_on(support, Z, K) && !support[256 + (ix.q >> 1)]) {
        // the whole float4 group is outside the support, and so are its mirror images
        const float2 z0 = make_float2(0.f, 0.f);
        h2[ix.q] = z0; h2[ix.qx] = z0; h2[ix.qy] = z0; h2[ix.qxy] = z0;
        return;
    }
    const long npx2 = (long)R * R / 2;
    const float2* z = reinterpret_cast<const float2*>(Z) + ix.q;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // 32 planes in flight per thread: the quadrant has only 100 k threads (six waves per CU), eight 8-byte loads each did not cover
    // the memory latency (94 us for 220 MB); the sums keep their k order (bit-identical to the full pass)
    int k = 0;
    for (; k + 32 <= K; k += 32) {
        float2 v32[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v32[j] = z[(long)(k + j) * npx2];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float2 v = v32[j];
            const float c0 = s_c4[k + j], c1 = s_c4[K + k + j], c2 = s_c4[2 * K + k + j], c3 = s_c4[3 * K + k + j];
            a[0] += (double)(c0 * v.x); a[1] += (double)(c0 * v.y);
            a[2] += (double)(c1 * v.x); a[3] += (double)(c1 * v.y);
            a[4] += (double)(c2 * v.x); a[5] += (double)(c2 * v.y);
            a[6] += (double)(c3 * v.x); a[7] += (double)(c3 * v.y);
        }
    }
#pragma unroll 8
    for (; k < K; ++k) {
        const float2 v = z[(long)k * npx2];
        const float c0 = s_c4[k], c1 = s_c4[K + k], c2 = s_c4[2 * K + k], c3 = s_c4[3 * K + k];
        a[0] += (double)(c0 * v.x); a[1] += (double)(c0 * v.y);
        a[2] += (double)(c1 * v.x); a[3] += (double)(c1 * v.y);
        a[4] += (double)(c2 * v.x); a[5] += (double)(c2 * v.y);
        a[6] += (double)(c3 * v.x); a[7] += (double)(c3 * v.y);
    }
    h2[ix.q] = make_float2((float)a[0], (float)a[1]);
    h2[ix.qx] = make_float2((float)a[3], (float)a[2]);          // mirrored in x: the pair swaps
    h2[ix.qy] = make_float2((float)a[4], (float)a[5]);
    h2[ix.qxy] = make_float2((float)a[7], (float)a[6]);
}

// adjoint from one quadrant: g[k] = sum_q Z[k][q] * (gh[q] + sx gh[qx] + sy gh[qy] + sx sy gh[qxy]); one partial row per wave.
// Eight planes per round: their loads are issued one round ahead (with ~6 waves per CU nothing else covers the miss latency), and the
// eight per-lane sums are folded over the wave together -- each of the first three exchange steps keeps the half of the values the
// lane's bit selects, so a round costs 10 f64 exchanges instead of 48.
__device__ __forceinline__ double shx(double v, int off) { return __shfl_xor(v, off, 64); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {      // the other lane's f64 through two 32-bit DPP moves (full EXEC)
    const long long i = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)(i & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)i >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__global__ __launch_bounds__(256) void zernike_grad_sym_kernel(const float* __restrict__ Z, const float* __restrict__ gh,
                                                               double* __restrict__ part, int K, int R,
                                                               const unsigned char* __restrict__ support,
                                                               const unsigned char* __restrict__ sym) {
    extern __shared__ unsigned char s_cls[];
    if (!sym_on(sym, Z, K)) return;
    const unsigned char* cls = sym_cls(sym, K);
    for (int k = threadIdx.x; k < K; k += 256) s_cls[k] = cls[k];
    __syncthreads();
    const QuadIdx ix = quad_index((long)blockIdx.x * 256 + threadIdx.x, R);
    const bool ok = ix.ok && (!support_on(support, Z, K) || support[256 + (ix.q >> 1)]);
    const float2* g2 = reinterpret_cast<const float2*>(gh);
    double dx[4] = {0, 0, 0, 0}, dy[4] = {0, 0, 0, 0};
    if (ok) {
        const float2 g = g2[ix.q], gx = g2[ix.qx], gy = g2[ix.qy], gxy = g2[ix.qxy];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double sx = (c & 1) ? -1.0 : 1.0, sy = (c & 2) ? -1.0 : 1.0;
            dx[c] = (double)g.x + sx * (double)gx.y + sy * (double)gy.x + sx * sy * (double)gxy.y;
            dy[c] = (double)g.y + sx * (double)gx.x + sy * (double)gy.y + sx * sy * (double)gxy.x;
        }
    }
    const long npx2 = (long)R * R / 2;
    const float2* z = reinterpret_cast<const float2*>(Z) + (ok ? ix.q : 0);      // lanes outside read a valid address with weight 0
    double* row = part + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * K;
    const int lane = threadIdx.x & 63;
    float2 v[8], vn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = z[(long)min(j, K - 1) * npx2];
    for (int k = 0; k < K; k += 8) {
        if (k + 8 < K) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vn[j] = z[(long)min(k + 8 + j, K - 1) * npx2];
        }
        double a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cl = s_cls[min(k + j, K - 1)];
            const double ex = cl == 0 ? dx[0] : cl == 1 ? dx[1] : cl == 2 ? dx[2] : dx[3];
            const double ey = cl == 0 ? dy[0] : cl == 1 ? dy[1] : cl == 2 ? dy[2] : dy[3];
            a[j] = (double)v[j].x * ex + (double)v[j].y * ey;
        }
        // fold.  The three HALVING steps (4, 2, 1 values exchanged) use the lane bits the DPP path can pair exactly -- xor 1 and xor 2
        // (quad_perm), xor 8 (row_ror:8) --, the single value left is then summed over bit 2 (row_shl / row_shr by 4) and over bits 4 and
        // 5 through the LDS crossbar: 4 ds_bpermute per round instead of 20 (each a dependent LDS round trip: they were most of the
        // kernel's 73 us).  Lane l ends with value (l & 1) * 4 + ((l >> 1) & 1) * 2 + ((l >> 3) & 1) summed over the wave.
        const bool h0 = lane & 1, h1 = lane & 2, h3 = lane & 8;
        double b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double keep = h0 ? a[4 + j] : a[j], give = h0 ? a[j] : a[4 + j];
            b[j] = keep + dpp_d<0xB1>(give);                      // quad_perm [1,0,3,2]: lane ^ 1
        }
        double c2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double keep = h1 ? b[2 + j] : b[j], give = h1 ? b[j] : b[2 + j];
            c2[j] = keep + dpp_d<0x4E>(give);                     // quad_perm [2,3,0,1]: lane ^ 2
        }
        double d = (h3 ? c2[1] : c2[0]) + dpp_d<0x128>(h3 ? c2[0] : c2[1]);   // row_ror:8 = lane ^ 8 inside a 16-lane row
        {
            const double up = dpp_d<0x104>(d), dn = dpp_d<0x114>(d);          // row_shl:4 (lane + 4), row_shr:4 (lane - 4)
            d += (lane & 4) ? dn : up;                                        // lane ^ 4
        }
        d += shx(d, 16);
        d += shx(d, 32);
        const int which = (lane & 1) * 4 + ((lane >> 1) & 1) * 2 + ((lane >> 3) & 1);
        if ((lane & 0x34) == 0 && k + which < K) row[k + which] = d;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = vn[j];
    }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, float* __restrict__ out,
                                                           int nwg, int K, const unsigned char* __restrict__ sym = nullptr,
                                                           int nwg_sym = 0, const float* __restrict__ Z = nullptr) {
    if (sym_on(sym, Z, K)) nwg = nwg_sym;                // the quadrant form left fewer partial rows
    // one workgroup per coefficient: 256 threads stride over the per-workgroup partials
    __shared__ double s_red[4];
    const int k = blockIdx.x;
    double a = 0;
    for (int w = threadIdx.x; w < nwg; w += 256) a += part[(long)w * K + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = (float)(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// ----------------------------------------------------------------------------- field at the phase plate
// F0[l][y][x] = aperture * sph[y][x][l] * c64(exp(i * kdn[l] * (h + noise)))      (c64 product, promoted to c128)
template <typename C2>
__global__ __launch_bounds__(256) void ic_field_kernel(const float* __restrict__ h, const float* __restrict__ noise,
                                                       const float2* __restrict__ sph, C2* __restrict__ F0,
                                                       int RR, double kdn0, double kdn1, double kdn2, float tol,
                                                       int use_tol) {
    typedef typename CT<C2>::R R_;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)RR * RR;
    if (idx >= npx) return;
    const int y = (int)(idx / RR), x = (int)(idx % RR);
    float hh = h[idx];
    if (use_tol) hh = __fadd_rn(hh, __fadd_rn(__fmul_rn(-tol - tol, noise[idx]), tol));   // f32, no FMA (Utils.py:403-406)
    const long yy = y - RR / 2, xx = x - RR / 2;
    const long rmax = RR / 2 - 1;
    const bool open = (yy * yy + xx * xx) < rmax * rmax;
    const double kd[3] = {kdn0, kdn1, kdn2};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        C2 f = CT<C2>::mk((R_)0, (R_)0);
        if (open) {
            double s, c;
            sincos(kd[l] * (double)hh, &s, &c);                  // the phase (up to ~4e3 rad) and its sine / cosine stay f64 in both forms
            const float2 pl = make_float2((float)c, (float)s);
            const float2 sp = sph[idx * 3 + l];
            f = CT<C2>::mk((R_)__fsub_rn(__fmul_rn(sp.x, pl.x), __fmul_rn(sp.y, pl.y)),
                           (R_)__fadd_rn(__fmul_rn(sp.x, pl.y), __fmul_rn(sp.y, pl.x)));
        }
        F0[(long)l * npx + idx] = f;
    }
}

// gh[px] = sum_l kdn[l] * Im(GF * conj(F0))
template <typename C2>
__global__ __launch_bounds__(256) void ic_field_bwd_kernel(const C2* __restrict__ GF, const C2* __restrict__ F0,
                                                           float* __restrict__ gh, long npx, double kdn0, double kdn1,
                                                           double kdn2) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npx) return;
    const double kd[3] = {kdn0, kdn1, kdn2};
    double a = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const C2 g = GF[(long)l * npx + idx], f = F0[(long)l * npx + idx];
        a += kd[l] * ((double)g.y * (double)f.x - (double)g.x * (double)f.y);
    }
    gh[idx] = (float)a;
}

// ----------------------------------------------------------------------------- Fresnel FFT passes
// rows: in [L][RR][RR] placed at column offset pad inside a zero row of length M -> out [L][RR][M]
template <bool C1344, typename C2>       // C1344: M = 1344, twiddles in registers (no LDS copy of the table: 43 KB per workgroup, three per CU)
__global__ __launch_bounds__(256) void dfft_rows_kernel(const C2* __restrict__ in, C2* __restrict__ out,
                                                        const C2* __restrict__ twg, FftPlan pl, int RR, int pad) {
    __shared__ C2 s_tw[C1344 ? 1 : MAXM];
    __shared__ C2 s_a[MAXM];
    __shared__ C2 s_b[MAXM];
    const int tid = threadIdx.x, M = pl.M;
    const long row = blockIdx.x;          // l * RR + y
    Tw1344<C2> w;
    if (C1344) tw1344_load(w, twg, tid);
    {   // the row's <= 6 elements per thread in flight together (see dfft_cols1_kernel)
        C2 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = in[row * RR + min(max(tid + k * 256 - pad, 0), RR - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + k * 256, x = i - pad;
            if (i < M) s_a[i] = (x >= 0 && x < RR) ? v[k] : CT<C2>::mk(0, 0);
        }
    }
    if (!C1344)
        for (int i = tid; i < M; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const C2* r = C1344 ? dfft_1344(s_a, s_b, w, tid) : dfft(s_a, s_b, (const C2*)s_tw, pl, tid, 256);
    const int l = (int)(row / RR), y = (int)(row % RR), CB = (M % 8 == 0) ? 8 : 2;
    for (int i = tid; i < M; i += 256) out[tix(l, y, i, RR, M, CB)] = r[i];
}

// cols: two columns per workgroup.  T1 [L][RR][M] rows placed at row offset pad -> column FFT -> x Ht[l][kx][ky]
// (c64 table, conj for the adjoint) -> inverse column FFT -> rows pad..pad+RR-1 -> T2 [L][RR][M], scaled.
__global__ __launch_bounds__(512) void dfft_cols_kernel(const double2* __restrict__ T1, double2* __restrict__ T2,
                                                        const float2* __restrict__ Ht,
                                                        const double2* __restrict__ twg, FftPlan pl, int RR, int pad,
                                                        int conj_h, double scale) {
    __shared__ double2 s_tw[MAXM];
    __shared__ double2 s_a[2][MAXM];
    __shared__ double2 s_b[2][MAXM];
    __shared__ const double2* s_ptr[2];
    const int tid = threadIdx.x, M = pl.M;
    // the four workgroups that share an 8-column block run on ONE XCD (blocks b, b + 8, ... share an XCD): its L2 fetches each 128-byte
    // piece once for all four
    int pb = blockIdx.x;
    if (gridDim.x % 32 == 0) {
        const int xcd = pb & 7, idx = pb >> 3;
        pb = (idx >> 2) * 32 + xcd * 4 + (idx & 3);
    }
    const int l = blockIdx.y, kx0 = pb * 2;
    const int col = tid >> 8, t = tid & 255;
    const int kx = kx0 + col;
    for (int i = tid; i < M; i += 512) s_tw[i] = twg[i];
    // both columns of a row in one 32-byte access (kx0 is even: 32-byte aligned), all 512 threads over the rows
    for (int i = tid; i < M; i += 512) {
        const int y = i - pad;
        double4 v = make_double4(0.0, 0.0, 0.0, 0.0);
        if (y >= 0 && y < RR) v = *reinterpret_cast<const double4*>(T1 + tix(l, y, kx0, RR, M, (M % 8 == 0) ? 8 : 2));
        s_a[0][i] = make_double2(v.x, v.y);
        s_a[1][i] = make_double2(v.z, v.w);
    }
    __syncthreads();
    double2* r = dfft(s_a[col], s_b[col], s_tw, pl, t, 256);
    double2* o = (r == s_a[col]) ? s_b[col] : s_a[col];
    const float2* hcol = Ht + ((long)l * M + kx) * M;
    for (int i = t; i < M; i += 256) {
        const float2 hf = hcol[i];
        const double2 hv = make_double2((double)hf.x, conj_h ? -(double)hf.y : (double)hf.y);
        const double2 v = dmul(r[i], hv);
        r[i] = make_double2(v.x, -v.y);                  // conj for the inverse transform
    }
    __syncthreads();
    const double2* z = dfft(r, o, s_tw, pl, t, 256);
    // the two columns may have ended in different ping-pong buffers only if their plans differed: they do not (same pl)
    if (t == 0) s_ptr[col] = z;
    __syncthreads();
    const double2 *z0 = s_ptr[0], *z1 = s_ptr[1];
    for (int i = tid; i < RR; i += 512) {
        const double2 a = z0[i + pad], b = z1[i + pad];
        *reinterpret_cast<double4*>(T2 + tix(l, i, kx0, RR, M, (M % 8 == 0) ? 8 : 2)) = make_double4(a.x * scale, -a.y * scale, b.x * scale, -b.y * scale);
    }
}

// One column per workgroup (round 3).  dfft_cols_kernel holds two columns + the twiddle table in 107.5 KB of LDS: one 512-thread
// workgroup per CU walking ten barrier-separated FFT stages of ~1.3 butterflies per thread -- latency-bound (2 x 150 us per step at
// 1 TB/s).  Here a workgroup owns ONE column (256 threads, 43 KB: ping-pong buffers only) and reads the twiddles from the 21.5-KB
// global table (L1 / L2 resident), so three workgroups share a CU and cover each other's barriers and load latencies; the eight
// workgroups of an 8-column block (one 128-byte line per row) run on one XCD.
template <bool TW_LDS, typename C2>
__global__ __launch_bounds__(256) void dfft_cols1_kernel(const C2* __restrict__ T1, C2* __restrict__ T2,
                                                         const float2* __restrict__ Ht, const C2* __restrict__ twg, FftPlan pl,
                                                         int RR, int pad, int conj_h, double scale_, int static_plan) {
    typedef typename CT<C2>::R R_;
    const R_ scale = (R_)scale_;
    __shared__ C2 s_tw[TW_LDS ? MAXM : 1];
    __shared__ C2 s_a[MAXM];
    __shared__ C2 s_b[MAXM];
    const int tid = threadIdx.x, M = pl.M;
    int pb = blockIdx.x;
    if (gridDim.x % 64 == 0) {                       // blocks b, b + 8, ... share an XCD: give it the 8 columns of one 128-byte block
        const int xcd = pb & 7, idx = pb >> 3;
        pb = (idx >> 3) * 64 + xcd * 8 + (idx & 7);
    }
    const int l = blockIdx.y, kx = pb;
    const int CB = (M % 8 == 0) ? 8 : 2;
    if (TW_LDS)
        for (int i = tid; i < M; i += 256) s_tw[i] = twg[i];
    {   // the <= 6 elements of a thread are requested together (MAXM = 1344 = 5.25 x 256): one round trip instead of one per element
        static_assert(MAXM <= 6 * 256, "six elements per thread");
        C2 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int y = min(max(tid + k * 256 - pad, 0), RR - 1);
            v[k] = T1[tix(l, y, kx, RR, M, CB)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + k * 256, y = i - pad;
            if (i < M) s_a[i] = (y >= 0 && y < RR) ? v[k] : CT<C2>::mk(0, 0);
        }
    }
    __syncthreads();
    const C2* tw = TW_LDS ? (const C2*)s_tw : twg;
    Tw1344<C2> w;
    const bool c1344 = !TW_LDS && static_plan && M == 1344;          // compile-time plan + twiddles in registers (uniform branch)
    if (c1344) tw1344_load(w, twg, tid);
    FftPlan pl2 = pl;
    if (!c1344 && M == 1344) pl2.M = -1344;            // dfft256 keys on M == 1344: hide it when the static form is off
    C2* r = dfft256(s_a, s_b, tw, w, pl2, tid);
    C2* o = (r == s_a) ? s_b : s_a;
    const float2* hcol = Ht + ((long)l * M + kx) * M;
    {
        float2 hf6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) hf6[k] = hcol[min(tid + k * 256, M - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + k * 256;
            if (i < M) {
                const float2 hf = hf6[k];
                const C2 hv = CT<C2>::mk((R_)hf.x, conj_h ? -(R_)hf.y : (R_)hf.y);
                const C2 v = dmul(r[i], hv);
                r[i] = CT<C2>::mk(v.x, -v.y);          // conj for the inverse transform
            }
        }
    }
    __syncthreads();
    const C2* z = dfft256(r, o, tw, w, pl2, tid);
    for (int i = tid; i < RR; i += 256) {
        const C2 a = z[i + pad];
        T2[tix(l, i, kx, RR, M, CB)] = CT<C2>::mk(a.x * scale, -a.y * scale);
    }
}

// inverse rows: T2 [L][RR][M] -> crop columns pad..pad+RR-1 -> U [L][RR][RR] (c128), optional intensity f32
template <bool C1344, typename C2>
__global__ __launch_bounds__(256) void difft_rows_kernel(const C2* __restrict__ T2, C2* __restrict__ U,
                                                         float* __restrict__ I32, const C2* __restrict__ twg,
                                                         FftPlan pl, int RR, int pad) {
    __shared__ C2 s_tw[C1344 ? 1 : MAXM];
    __shared__ C2 s_a[MAXM];
    __shared__ C2 s_b[MAXM];
    const int tid = threadIdx.x, M = pl.M;
    const long row = blockIdx.x;
    const int l_ = (int)(row / RR), y_ = (int)(row % RR), CB = (M % 8 == 0) ? 8 : 2;
    Tw1344<C2> w;
    if (C1344) tw1344_load(w, twg, tid);
    {
        C2 v6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v6[k] = T2[tix(l_, y_, min(tid + k * 256, M - 1), RR, M, CB)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int i = tid + k * 256;
            if (i < M) s_a[i] = CT<C2>::mk(v6[k].x, -v6[k].y);
        }
    }
    if (!C1344)
        for (int i = tid; i < M; i += 256) s_tw[i] = twg[i];
    __syncthreads();
    const C2* r = C1344 ? dfft_1344(s_a, s_b, w, tid) : dfft(s_a, s_b, (const C2*)s_tw, pl, tid, 256);
    for (int i = tid; i < RR; i += 256) {
        const C2 v = CT<C2>::mk(r[i + pad].x, -r[i + pad].y);
        U[row * RR + i] = v;
        if (I32) I32[row * RR + i] = (float)(v.x * v.x + v.y * v.y);      // Utils.py:208-209, cast Utils.py:218
    }
}

// ----------------------------------------------------------------------------- area down-sample (Utils.py:216-248)
// raw[Y][X][l] = (1/up^2) * sum_{a,b<up} I32[l][src(up*Y+a)][src(up*X+b)],  src(U) = min(floor(U*scale), RR-1)
__global__ __launch_bounds__(256) void area_down_kernel(const float* __restrict__ I32, float* __restrict__ raw,
                                                        double* __restrict__ sums, int RR, int P, int up, float scale) {
    __shared__ double s_red[4][3];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = idx < (long)P * P;
    const int Y = ok ? (int)(idx / P) : 0, X = ok ? (int)(idx % P) : 0;
    double acc[3] = {0, 0, 0};
    if (ok) {
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const float* src = I32 + (long)l * RR * RR;
            float s = 0.f;
            for (int a = 0; a < up; ++a) {
                const int r = min((int)floorf((float)(up * Y + a) * scale), RR - 1);
                for (int b = 0; b < up; ++b) {
                    const int c = min((int)floorf((float)(up * X + b) * scale), RR - 1);
                    s += src[(long)r * RR + c];
                }
            }
            s = s / (float)(up * up);
            raw[idx * 3 + l] = s;
            acc[l] = (double)s;
        }
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        double a = acc[l];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][l] = a;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int l = threadIdx.x;
        atomicAdd(&sums[l], s_red[0][l] + s_red[1][l] + s_red[2][l] + s_red[3][l]);
    }
}

// psf_n = raw / sum (f32, Lens.py:239);  loss_acc += sum((psf_n*m1 - psf_n)^2) (f64, Lens.py:271);
// psf_m = psf_n * m2 (f64, Lens.py:274).  m1 / m2 / psf_m / loss_acc may be null.
__global__ __launch_bounds__(256) void psf_finalize_kernel(const float* __restrict__ raw, const double* __restrict__ sums,
                                                           const double* __restrict__ m1, const double* __restrict__ m2,
                                                           float* __restrict__ psf_n, double* __restrict__ psf_m,
                                                           double* __restrict__ loss_acc, long n) {
    __shared__ double s_red[4];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    double a = 0;
    if (idx < n) {
        const int l = (int)(idx % 3);
        const float v = raw[idx] / (float)sums[l];
        psf_n[idx] = v;
        if (m1) {
            const double d = (double)v * m1[idx] - (double)v;
            a = d * d;
        }
        if (psf_m) psf_m[idx] = (double)v * m2[idx];
    }
    if (loss_acc) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(loss_acc, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    }
}

// backward of finalize: g_n = g_psf_m*m2 + g_psf_n + g_loss * psf_n*(m1-1)^2 / loss;  dots[l] += sum g_n * psf_n
__global__ __launch_bounds__(256) void psf_finalize_bwd1_kernel(const float* __restrict__ psf_n,
                                                                const double* __restrict__ m1,
                                                                const double* __restrict__ m2,
                                                                const double* __restrict__ g_psf_m,
                                                                const float* __restrict__ g_psf_n,
                                                                const double* __restrict__ g_loss,
                                                                const double* __restrict__ loss, double* __restrict__ g_n,
                                                                double* __restrict__ dots, long n) {
    __shared__ double s_red[4][3];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    double acc[3] = {0, 0, 0};
    if (idx < n) {
        const int l = (int)(idx % 3);
        const double v = (double)psf_n[idx];
        double g = 0;
        if (g_psf_m) g += g_psf_m[idx] * m2[idx];
        if (g_psf_n) g += (double)g_psf_n[idx];
        if (g_loss && m1) {
            const double d = m1[idx] - 1.0;
            const double L = *loss;
            if (L > 0) g += (*g_loss) * v * d * d / L;
        }
        g_n[idx] = g;
        acc[l] = g * v;
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        double a = acc[l];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][l] = a;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int l = threadIdx.x;
        atomicAdd(&dots[l], s_red[0][l] + s_red[1][l] + s_red[2][l] + s_red[3][l]);
    }
}

// g_raw = (g_n - dot_l) / sum_l ; then adjoint of area_down fused with the intensity adjoint:
// GU[l][r][c] = 2 * U[l][r][c] * (1/up^2) * sum_{U0: src(U0)=r} sum_{V0: src(V0)=c} g_raw[U0/up][V0/up][l]
template <typename C2>
__global__ __launch_bounds__(256) void area_down_bwd_kernel(const double* __restrict__ g_n, const double* __restrict__ dots,
                                                            const double* __restrict__ sums,
                                                            const C2* __restrict__ U, C2* __restrict__ GU,
                                                            int RR, int P, int up, float scale) {
    typedef typename CT<C2>::R R_;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)RR * RR) return;
    const int r = (int)(idx / RR), c = (int)(idx % RR);
    const int ulo = max(0, (int)floorf((float)r / scale) - 2), uhi = min(up * P - 1, (int)floorf((float)(r + 1) / scale) + 2);
    const int vlo = max(0, (int)floorf((float)c / scale) - 2), vhi = min(up * P - 1, (int)floorf((float)(c + 1) / scale) + 2);
    double g[3] = {0, 0, 0};
    for (int u0 = ulo; u0 <= uhi; ++u0) {
        if (min((int)floorf((float)u0 * scale), RR - 1) != r) continue;
        for (int v0 = vlo; v0 <= vhi; ++v0) {
            if (min((int)floorf((float)v0 * scale), RR - 1) != c) continue;
            const long o = ((long)(u0 / up) * P + (v0 / up)) * 3;
#pragma unroll
            for (int l = 0; l < 3; ++l) g[l] += (g_n[o + l] - dots[l]) / (double)(float)sums[l];
        }
    }
    const double inv = 1.0 / (double)(up * up);
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const C2 u = U[(long)l * RR * RR + idx];
        const double w = 2.0 * g[l] * inv;
        GU[(long)l * RR * RR + idx] = CT<C2>::mk((R_)(w * (double)u.x), (R_)(w * (double)u.y));
    }
}

}  // namespace ppv

// =============================================================================== host side
using namespace ppv;

namespace {

int make_plan(int M, FftPlan* pl) {
    if (M > MAXM || M < 8) return PPV_ERR_BAD_SIZE;
    pl->M = M;
    pl->nst = 0;
    int m = M;
    while (m % 4 == 0 && pl->nst < MAXST) { pl->radix[pl->nst++] = 4; m /= 4; }
    while (m % 2 == 0 && pl->nst < MAXST) { pl->radix[pl->nst++] = 2; m /= 2; }
    while (m % 3 == 0 && pl->nst < MAXST) { pl->radix[pl->nst++] = 3; m /= 3; }
    while (m % 7 == 0 && pl->nst < MAXST) { pl->radix[pl->nst++] = 7; m /= 7; }
    for (int q : {5, 11, 13, 23})                               // direct-summation stages (dstage_any)
        while (m % q == 0 && pl->nst < MAXST) { pl->radix[pl->nst++] = q; m /= q; }
    return m == 1 ? PPV_OK : PPV_ERR_BAD_SIZE;
}

struct IcWs {            // carve-up of the caller's persistent state buffer (saved for backward)
    float* h;            // [RR*RR]
    double2* F0;         // [3][RR][RR]
    double2* U;          // [3][RR][RR]
    double2* T1;         // [3][RR][M]
    double2* T2;         // [3][RR][M]
    float* I32;          // [3][RR][RR]
    float* raw;          // [P][P][3]
    double* sums;        // [3] + dots[3] + loss_acc[1] + pad
    double* g_n;         // [P*P*3]
    double* part;        // [nwg][K]
    float* gh;           // [RR*RR]
    unsigned char* support;   // [RR*RR/4] basis support per float4 group (ppv_ic_psf_mark_support), behind a one-byte "marked" flag
    unsigned char* sym;       // mirror symmetry of the basis: 256-byte header (SymHdr), [K] u32 sign-pair sets, [K] u8 classes
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(IcWs* w, char* base, int RR, int P, int K) {
    const int M = RR + 2 * (RR / 4);
    const size_t npx = (size_t)RR * RR;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    w->h = (float*)take(npx * 4);
    w->F0 = (double2*)take(3 * npx * 16);
    w->U = (double2*)take(3 * npx * 16);
    w->T1 = (double2*)take((size_t)3 * RR * M * 16);
    w->T2 = (double2*)take((size_t)3 * RR * M * 16);
    w->I32 = (float*)take(3 * npx * 4);
    w->raw = (float*)take((size_t)P * P * 3 * 4);
    w->sums = (double*)take(8 * 8);
    w->g_n = (double*)take((size_t)P * P * 3 * 8);
    const size_t nwg = (npx / 4 + 255) / 256;
    w->part = (double*)take(nwg * 4 * K * 8);
    w->gh = (float*)take(npx * 4);
    w->support = (unsigned char*)take(npx / 4 + 256);          // 256-byte header (magic word) + one byte per float4 group
    w->sym = (unsigned char*)take(256 + (size_t)K * 5);
    return off;
}

}  // namespace

namespace {

unsigned quad_blocks(int RR) { return (unsigned)(((long)(RR / 2) * (RR / 4) + 255) / 256); }

// PPV_ZERNIKE_SYM=0 keeps the full-basis passes even when the basis was found mirror-symmetric.  Whether a state's basis IS symmetric
// is known on the device only (header written by the marking pass; the host never reads it back in the step): both forms are
// launched, and the one the header does not select returns at once (one empty launch per direction, ~4 us, against ~130 us saved).
// PPV_DFFT_STATIC=0: the run-time-plan transforms also for M = 1344 (A/B of the compile-time plan with register twiddles)
bool dfft_static(int M) {
    static const bool on = !(getenv("PPV_DFFT_STATIC") && atoi(getenv("PPV_DFFT_STATIC")) == 0);
    return on && M == 1344;
}

bool sym_allowed() {
    static const bool on = !(getenv("PPV_ZERNIKE_SYM") && atoi(getenv("PPV_ZERNIKE_SYM")) == 0);
    return on;
}

void launch_contract(const float* Z, const float* coeffs, const IcWs& w, int K, int RR, hipStream_t stream) {
    const long npx4 = (long)RR * RR / 4;
    const unsigned char* sym = sym_allowed() ? w.sym : nullptr;
    if (sym) zernike_contract_sym_kernel<<<quad_blocks(RR), 256, 4 * K * sizeof(float), stream>>>(Z, coeffs, w.h, K, RR, w.support, sym);
    zernike_contract_kernel<<<(unsigned)((npx4 + 255) / 256), 256, K * sizeof(float), stream>>>(Z, coeffs, w.h, K, npx4, w.support, sym);
}
void launch_grad(const float* Z, float* g_coeffs, const IcWs& w, int K, int RR, hipStream_t stream) {
    const long npx4 = (long)RR * RR / 4;
    const unsigned nwg = (unsigned)((npx4 + 255) / 256);
    const unsigned char* sym = sym_allowed() ? w.sym : nullptr;
    if (sym) zernike_grad_sym_kernel<<<quad_blocks(RR), 256, K, stream>>>(Z, w.gh, w.part, K, RR, w.support, sym);
    zernike_grad_kernel<<<nwg, 256, 0, stream>>>(Z, w.gh, w.part, K, npx4, w.support, sym);
    sum_partials_kernel<<<K, 256, 0, stream>>>(w.part, g_coeffs, (int)nwg * 4, K, sym, (int)quad_blocks(RR) * 4, Z);
}

}  // namespace

extern "C" {

size_t ppv_ic_psf_state_bytes(int RR, int P, int K) {
    IcWs w;
    return carve(&w, nullptr, RR, P, K);
}

// column pass of the Fresnel transform: PPV_DFFT_COLS = 0 two columns per workgroup (rounds 1-2), 1 one column + twiddles in LDS (two
// workgroups per CU), 2 one column, twiddles from L2 (three per CU)
static int dfft_cols_mode() {
    static const int mode = getenv("PPV_DFFT_COLS") ? atoi(getenv("PPV_DFFT_COLS")) : 2;
    return mode;
}
static void launch_dfft_cols(const double2* T1, double2* T2, const float2* Ht, const double2* tw, const FftPlan& pl, int RR, int pad, int M,
                             int conj_h, double scale, hipStream_t stream) {
    const int mode = dfft_cols_mode();
    if (mode == 1) dfft_cols1_kernel<true, double2><<<dim3(M, 3), 256, 0, stream>>>(T1, T2, Ht, tw, pl, RR, pad, conj_h, scale, 0);
    else if (mode == 2) dfft_cols1_kernel<false, double2><<<dim3(M, 3), 256, 0, stream>>>(T1, T2, Ht, tw, pl, RR, pad, conj_h, scale, dfft_static(M) ? 1 : 0);
    else dfft_cols_kernel<<<dim3(M / 2, 3), 512, 0, stream>>>(T1, T2, Ht, tw, pl, RR, pad, conj_h, scale);
}
static void launch_dfft_cols(const float2* T1, float2* T2, const float2* Ht, const float2* tw, const FftPlan& pl, int RR, int pad, int M,
                             int conj_h, double scale, hipStream_t stream) {
    // c64: one column per workgroup, 21.5 KB of LDS (seven workgroups per CU); twiddles in LDS only on request (PPV_DFFT_COLS=1)
    if (dfft_cols_mode() == 1) dfft_cols1_kernel<true, float2><<<dim3(M, 3), 256, 0, stream>>>(T1, T2, Ht, tw, pl, RR, pad, conj_h, scale, 0);
    else dfft_cols1_kernel<false, float2><<<dim3(M, 3), 256, 0, stream>>>(T1, T2, Ht, tw, pl, RR, pad, conj_h, scale, dfft_static(M) ? 1 : 0);
}

// element type of the Fresnel transforms and of the fields kept for backward (F0, U): c64 unless PPV_PSF_F32=0 (c128, rounds 1-4)
static int g_psf_f32 = -1;                                    // -1: the environment decides; 0 / 1: ppv_ic_psf_set_fields_f32
static bool psf_f32() {
    static const bool env_on = !(getenv("PPV_PSF_F32") && atoi(getenv("PPV_PSF_F32")) == 0);
    const int v = __atomic_load_n(&g_psf_f32, __ATOMIC_RELAXED);
    return v < 0 ? env_on : v != 0;
}
}  // extern "C" (templates need C++ linkage)

namespace {
template <typename C2> const C2* fresnel_twiddles(int M);
template <> const double2* fresnel_twiddles<double2>(int M) { return (const double2*)ppv_twiddles_f64(M); }
template <> const float2* fresnel_twiddles<float2>(int M) { return (const float2*)ppv_twiddles_f32(M); }

template <typename C2>
int psf_fwd_t(const float* Z, const float* coeffs, const float* noise, const void* sph, const void* Ht, const double* kdn, float tol,
              const double* m1, const double* m2, float* psf_n, double* psf_m, double* loss_acc, void* state, int RR, int P, int K, int up,
              float up_scale, hipStream_t stream) {
    const int pad = RR / 4, M = RR + 2 * pad;
    FftPlan pl;
    if (int e = make_plan(M, &pl)) return e;
    const C2* tw = fresnel_twiddles<C2>(M);
    if (!tw) return PPV_ERR_INIT;
    IcWs w;
    carve(&w, (char*)state, RR, P, K);
    C2 *F0 = (C2*)w.F0, *U = (C2*)w.U, *T1 = (C2*)w.T1, *T2 = (C2*)w.T2;      // (the c64 form uses the first half of each c128-sized region)
    const long npx = (long)RR * RR;
    (void)hipMemsetAsync(w.sums, 0, 64, stream);
    if (loss_acc) (void)hipMemsetAsync(loss_acc, 0, 8, stream);
    launch_contract(Z, coeffs, w, K, RR, stream);
    ic_field_kernel<C2><<<(unsigned)((npx + 255) / 256), 256, 0, stream>>>(w.h, noise, (const float2*)sph, F0, RR, kdn[0],
                                                                         kdn[1], kdn[2], tol, (tol >= 0.f && noise) ? 1 : 0);
    if (dfft_static(M)) dfft_rows_kernel<true, C2><<<3 * RR, 256, 0, stream>>>(F0, T1, tw, pl, RR, pad);
    else dfft_rows_kernel<false, C2><<<3 * RR, 256, 0, stream>>>(F0, T1, tw, pl, RR, pad);
    launch_dfft_cols(T1, T2, (const float2*)Ht, tw, pl, RR, pad, M, 0, 1.0 / ((double)M * (double)M), stream);
    if (dfft_static(M)) difft_rows_kernel<true, C2><<<3 * RR, 256, 0, stream>>>(T2, U, w.I32, tw, pl, RR, pad);
    else difft_rows_kernel<false, C2><<<3 * RR, 256, 0, stream>>>(T2, U, w.I32, tw, pl, RR, pad);
    area_down_kernel<<<(unsigned)(((long)P * P + 255) / 256), 256, 0, stream>>>(w.I32, w.raw, w.sums, RR, P, up, up_scale);
    const long n = (long)P * P * 3;
    psf_finalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(w.raw, w.sums, m1, m2, psf_n, psf_m,
                                                                       m1 ? loss_acc : nullptr, n);
    return ppv_last_error();
}

template <typename C2>
int psf_bwd_t(const float* Z, const void* Ht, const double* kdn, const double* m1, const double* m2, const float* psf_n,
              const double* g_psf_m, const float* g_psf_n, const double* g_loss, const double* loss, float* g_coeffs, void* state, int RR,
              int P, int K, int up, float up_scale, hipStream_t stream) {
    const int pad = RR / 4, M = RR + 2 * pad;
    FftPlan pl;
    if (int e = make_plan(M, &pl)) return e;
    const C2* tw = fresnel_twiddles<C2>(M);
    if (!tw) return PPV_ERR_INIT;
    IcWs w;
    carve(&w, (char*)state, RR, P, K);
    C2 *F0 = (C2*)w.F0, *U = (C2*)w.U, *T1 = (C2*)w.T1, *T2 = (C2*)w.T2;
    const long npx = (long)RR * RR, n = (long)P * P * 3;
    double* dots = w.sums + 3;
    (void)hipMemsetAsync(dots, 0, 24, stream);
    psf_finalize_bwd1_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(psf_n, m1, m2, g_psf_m, g_psf_n, g_loss, loss,
                                                                            w.g_n, dots, n);
    C2* GU = T2;      // T2 is dead after forward and holds 3*RR*M >= 3*RR*RR elements; U and F0 stay intact
    area_down_bwd_kernel<C2><<<(unsigned)((npx + 255) / 256), 256, 0, stream>>>(w.g_n, dots, w.sums, U, GU, RR, P, up, up_scale);
    if (dfft_static(M)) dfft_rows_kernel<true, C2><<<3 * RR, 256, 0, stream>>>(GU, T1, tw, pl, RR, pad);
    else dfft_rows_kernel<false, C2><<<3 * RR, 256, 0, stream>>>(GU, T1, tw, pl, RR, pad);
    // cols: T1 -> T2 would overwrite GU while reading T1 only: fine (GU no longer needed)
    launch_dfft_cols(T1, T2, (const float2*)Ht, tw, pl, RR, pad, M, 1, 1.0 / ((double)M * (double)M), stream);
    C2* GF = T1;                                            // T1 dead again
    if (dfft_static(M)) difft_rows_kernel<true, C2><<<3 * RR, 256, 0, stream>>>(T2, GF, nullptr, tw, pl, RR, pad);
    else difft_rows_kernel<false, C2><<<3 * RR, 256, 0, stream>>>(T2, GF, nullptr, tw, pl, RR, pad);
    ic_field_bwd_kernel<C2><<<(unsigned)((npx + 255) / 256), 256, 0, stream>>>(GF, F0, w.gh, npx, kdn[0], kdn[1], kdn[2]);
    launch_grad(Z, g_coeffs, w, K, RR, stream);
    return ppv_last_error();
}
}  // namespace

extern "C" {

// 1: the Fresnel transforms and the saved fields F0 / U (ppv_ic_psf_state_offsets) are c64, 0: c128 (PPV_PSF_F32=0)
int ppv_ic_psf_fields_f32(void) { return psf_f32() ? 1 : 0; }
// on = 1 / 0: c64 / c128 fields from now on; -1: back to the environment's choice.  The state buffer's layout depends on it: set it BEFORE
// sizing (ppv_ic_psf_state_bytes) and initialising a state, and keep it while that state is in use.  Returns the previous setting.
int ppv_ic_psf_set_fields_f32(int on) { return __atomic_exchange_n(&g_psf_f32, on < 0 ? -1 : (on ? 1 : 0), __ATOMIC_RELAXED); }


// Optional, once per (state, Z): mark where the basis Z [K][RR][RR] is non-zero, so that ppv_ic_psf_fwd / _bwd skip the pixel groups
// outside its support (the aperture disk of poppy's zernike_basis(outside = 0): 21.5 % of the 1.12 GB read per direction).  Exact:
// the skipped products are zeros.  Call again when Z changes; a state that was never marked is simply read in full.
// Required once for a fresh state buffer (before the first ppv_ic_psf_fwd / _mark_support): clears the support and symmetry headers, so
// that bytes left behind by an earlier owner of the memory can never be taken for a marking.
int ppv_ic_psf_state_init(void* state, int RR, int P, int K, hipStream_t stream) {
    if (!state) return PPV_ERR_NULL;
    IcWs w;
    carve(&w, (char*)state, RR, P, K);
    if (hipError_t e = hipMemsetAsync(w.support, 0, 256, stream)) return -(int)e;
    if (hipError_t e = hipMemsetAsync(w.sym, 0, 256, stream)) return -(int)e;
    return PPV_OK;
}

int ppv_ic_psf_mark_support(const float* Z, void* state, int RR, int P, int K, hipStream_t stream) {
    if (!Z || !state) return PPV_ERR_NULL;
    if (RR % 4 || (RR * (long)RR) % 4) return PPV_ERR_BAD_SIZE;
    IcWs w;
    carve(&w, (char*)state, RR, P, K);
    const long npx4 = (long)RR * RR / 4;
    (void)hipMemsetAsync(w.support, 0, 256, stream);
    zernike_support_kernel<<<(unsigned)((npx4 + 255) / 256), 256, 0, stream>>>(Z, w.support, K, npx4);
    zernike_support_seal_kernel<<<1, 1, 0, stream>>>(w.support, Z, K);
    // mirror symmetry (see zernike_sym_check_kernel): header off, every sign pair still possible, then one pass over the basis
    (void)hipMemsetAsync(w.sym, 0, 256, stream);
    (void)hipMemsetAsync(w.sym + 256, 0xFF, (size_t)K * 4, stream);
    zernike_sym_check_kernel<<<quad_blocks(RR), 256, 0, stream>>>(Z, w.sym, K, RR);
    zernike_sym_seal_kernel<<<1, 256, 0, stream>>>(w.sym, K, Z);
    return ppv_last_error();
}

// 1 when the state's basis was found mirror-symmetric by ppv_ic_psf_mark_support (the PSF passes then read one quadrant of it), 0 when
// not (or never marked), < 0 on error.  Synchronises the stream (diagnostic / test entry point).
int ppv_ic_psf_symmetric(const void* state, int RR, int P, int K, hipStream_t stream) {
    if (!state) return PPV_ERR_NULL;
    IcWs w;
    carve(&w, (char*)state, RR, P, K);
    SymHdr h;
    if (hipError_t e = hipMemcpyAsync(&h, w.sym, sizeof(h), hipMemcpyDeviceToHost, stream)) return -(int)e;
    if (hipError_t e = hipStreamSynchronize(stream)) return -(int)e;
    return h.magic == SYM_MAGIC ? 1 : 0;
}

// Forward PSF generation (Lens.py:158-274).
//   Z [K][RR][RR] f32, coeffs [K] f32, noise [RR*RR] f32 U[0,1) (may be null when tol < 0),
//   sph [RR][RR][3] c64 (Lens.py:191-210, cached constant), Ht [3][M][M] c64 = Fresnel transfer function transposed
//   (kx-major; Utils.py:339-373, cached constant), kdn[3] = 2 pi / lambda * (n - 1) on the HOST,
//   m1/m2 [P][P][3] f64 or null; outputs psf_n [P][P][3] f32, psf_m [P][P][3] f64 (if m2), loss_acc [1] f64 (if m1;
//   = sum of squares, caller takes sqrt).  `state` (ppv_ic_psf_state_bytes) keeps what backward needs.
int ppv_ic_psf_fwd(const float* Z, const float* coeffs, const float* noise, const void* sph, const void* Ht,
                   const double* kdn, float tol, const double* m1, const double* m2, float* psf_n, double* psf_m,
                   double* loss_acc, void* state, int RR, int P, int K, int up, float up_scale, hipStream_t stream) {
    if (!Z || !coeffs || !sph || !Ht || !kdn || !psf_n || !state) return PPV_ERR_NULL;
    if (RR % 4 || (RR * (long)RR) % 4) return PPV_ERR_BAD_SIZE;
    if (psf_f32())
        return psf_fwd_t<float2>(Z, coeffs, noise, sph, Ht, kdn, tol, m1, m2, psf_n, psf_m, loss_acc, state, RR, P, K, up, up_scale, stream);
    return psf_fwd_t<double2>(Z, coeffs, noise, sph, Ht, kdn, tol, m1, m2, psf_n, psf_m, loss_acc, state, RR, P, K, up, up_scale, stream);
}

// Backward: given g_psf_m (f64, grad of the masked psf; null if unused), g_psf_n (f32, grad of the un-masked
// normalised psf; null if unused), g_loss (device scalar f64, grad of the loss = sqrt(loss_acc); null if unused) and
// loss (device scalar f64 = sqrt(loss_acc)), writes g_coeffs [K] f32.  Uses the state left by ppv_ic_psf_fwd.
int ppv_ic_psf_bwd(const float* Z, const void* Ht, const double* kdn, const double* m1, const double* m2,
                   const float* psf_n, const double* g_psf_m, const float* g_psf_n, const double* g_loss,
                   const double* loss, float* g_coeffs, void* state, int RR, int P, int K, int up, float up_scale,
                   hipStream_t stream) {
    if (!Z || !Ht || !kdn || !psf_n || !g_coeffs || !state) return PPV_ERR_NULL;
    if (psf_f32())
        return psf_bwd_t<float2>(Z, Ht, kdn, m1, m2, psf_n, g_psf_m, g_psf_n, g_loss, loss, g_coeffs, state, RR, P, K, up, up_scale, stream);
    return psf_bwd_t<double2>(Z, Ht, kdn, m1, m2, psf_n, g_psf_m, g_psf_n, g_loss, loss, g_coeffs, state, RR, P, K, up, up_scale, stream);
}

// h[px] = sum_k c[k] Z[k][px]  (IC Lens.py:176 / FD Optics.py:79-83); npx % 4 == 0
int ppv_zernike_contract(const float* Z, const float* coeffs, float* h, int K, long npx, hipStream_t stream) {
    if (!Z || !coeffs || !h) return PPV_ERR_NULL;
    if (npx % 4) return PPV_ERR_BAD_SIZE;
    const long npx4 = npx / 4;
    zernike_contract_kernel<<<(unsigned)((npx4 + 255) / 256), 256, K * sizeof(float), stream>>>(Z, coeffs, h, K, npx4, nullptr, nullptr);
    return ppv_last_error();
}

// g_coeffs[k] = sum_px Z[k][px] * gh[px]  (adjoint of ppv_zernike_contract); part: scratch of
// ppv_zernike_grad_scratch_bytes(K, npx)
size_t ppv_zernike_grad_scratch_bytes(int K, long npx) { return ((size_t)((npx / 4 + 255) / 256)) * 4 * K * sizeof(double); }
int ppv_zernike_grad(const float* Z, const float* gh, float* g_coeffs, void* part, int K, long npx, hipStream_t stream) {
    if (!Z || !gh || !g_coeffs || !part) return PPV_ERR_NULL;
    if (npx % 4) return PPV_ERR_BAD_SIZE;
    const long npx4 = npx / 4;
    const unsigned nwg = (unsigned)((npx4 + 255) / 256);
    zernike_grad_kernel<<<nwg, 256, 0, stream>>>(Z, gh, (double*)part, K, npx4, nullptr, nullptr);
    sum_partials_kernel<<<K, 256, 0, stream>>>((const double*)part, g_coeffs, (int)nwg * 4, K);
    return ppv_last_error();
}

// debug / parity taps into the saved state
int ppv_ic_psf_state_offsets(int RR, int P, int K, size_t* off_h, size_t* off_F0, size_t* off_U, size_t* off_I32,
                             size_t* off_raw) {
    IcWs w;
    carve(&w, (char*)256, RR, P, K);
    *off_h = (size_t)((char*)w.h - (char*)256);
    *off_F0 = (size_t)((char*)w.F0 - (char*)256);
    *off_U = (size_t)((char*)w.U - (char*)256);
    *off_I32 = (size_t)((char*)w.I32 - (char*)256);
    *off_raw = (size_t)((char*)w.raw - (char*)256);
    return PPV_OK;
}

}  // extern "C"
