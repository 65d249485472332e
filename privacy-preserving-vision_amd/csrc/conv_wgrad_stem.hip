// Weight gradient of the NHWC bf16 convolutions and the ResNet stem (7x7/2, 3 -> 64 channels) on MFMA, gfx950.
//
// wgrad (layer2..4 of the Encoder are trainable, models.py:43-54):
//     dW[n][r][s][c] = sum_m G[m][n] * src[b, ho*st + r - pad, wo*st + s - pad, c]
//   Both operands have the reduction index m as their SLOW memory dimension, so both 64-row tiles are staged
//   row-major ([m][128 cols], global_load_lds, XOR-swizzled 32-byte blocks) and the MFMA fragments are read
//   TRANSPOSED with ds_read_b64_tr_b16 -- no register or LDS transpose pass.  Kernels: conv_wgrad3x3_kernel (stride-1 3x3:
//   one G stage feeds three taps, X staged once as a halo tile), conv_wgrad_pipe_kernel<TN, NSTAGE> (everything else: one
//   tap per workgroup; TN = 128 x 2 stages is the default because it shares a CU with a conv workgroup when the weight
//   gradients run on their side stream), conv_wgrad_kernel (the first, two-stage + f32-atomics form, kept as variant 1).
//   Each workgroup owns a tile and a slice of m; slices write private f32 slabs that wgrad_to_torch sums into [N][C][R][S].
//
// stem forward: K = 7*3*8 = 168 (padded to 192) is built in LDS as an im2col tile from the f32 NCHW sensor image
//   (each 16-byte chunk = 7 consecutive input columns + one zero), 64 output channels, BN partial statistics fused.
// stem data gradient (needed: the lens trains through the frozen stem): expressed as a 4x4-tap, stride-1,
//   16-column conv over 2x2 input super-pixels and run by conv_gemm (N = 16); this file only holds its weight
//   re-layout and the final scatter to NCHW f32.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <utility>
#include "ppv_common.h"
#include "ppv_hip.h"

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
    int splits;            // number of m-slices
    int xcd_group;         // 1: tiles of one m-slice share an XCD (1-D grid decode)
    long slab_elems;       // > 0: slice z stores its tile plainly into dW + z * slab_elems (no atomics)
    int xcc_slabs;         // 1: every slice ADDS (f32 atomics) into the slab of the XCD it runs on (dW + XCC_ID * slab_elems, pre-zeroed)
    int pf_dist;           // cooperative L2 prefetch distance in 64-row stages (PF instantiations of conv_wgrad_pipe_kernel)
    int chunked;           // layout experiment (1x1 / unit stride): G is [N / 128][M][128], X is [Cs / 128][M][128]
    int native_slabs;      // 1: slabs are written in the accumulators' own order (one 16-byte store per lane and MFMA tile, whole 1-KB
                           //    wave stores) and wgrad_reduce_native_kernel sums + scatters them; 0: [N][R][S][C] slabs (wgrad_to_torch_kernel)
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


// ---- inline-asm transposed reads for the pipelined kernel.  hipcc (ROCm 7.2) places s_waitcnt vmcnt(0) in front of
// every __builtin_amdgcn_ds_read_tr16_b64 while a global_load_lds is in flight (it cannot prove the LDS-DMA does not
// alias the read), which drains the prefetch ring every stage.  The asm form is invisible to that pass; completion is
// tracked by hand: tr_wait_all() = s_waitcnt lgkmcnt(0) + sched_barrier (cdna_hip_programming.md 5.7 item 1 (iii), rule 18).
__device__ __forceinline__ void tr_issue(s16x4& lo, s16x4& hi, const char* tile, int k0, int colblock, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = k0 + 8 * g + q, r1 = r0 + 4;
    const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tile + r0 * 256 + (colblock ^ trkey(r0)) * 32 + p * 8);
    const unsigned a1 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tile + r1 * 256 + (colblock ^ trkey(r1)) * 32 + p * 8);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
}
__device__ __forceinline__ void tr_wait_all() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 tr_pack(const s16x4& lo, const s16x4& hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <class F, int... Is>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int OFF> __device__ __forceinline__ void tr_read_off(s16x4& v, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// exact floor(n / d), n - q*d for 0 <= n < 2^24 via a float reciprocal and one correction step
__device__ __forceinline__ void fast_divmod(int n, int d, float rcp, int& q, int& r) {
    q = (int)((float)n * rcp);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

// ----------------------------------------------------------------------------- multi-stage wgrad
// Same tile maths as conv_wgrad_kernel with the latency-hiding structure of conv_gemm_pipe_kernel: NSTAGE LDS stages,
// NSTAGE-1 stages of global_load_lds in flight, one raw s_barrier + one counted s_waitcnt vmcnt per 64-row stage,
// one workgroup per CU (TN = 256: 8 waves, 48 KB stages x 3; TN = 128: 4 waves, 32 KB stages x 4).  Halving the
// workgroup count against the two-stage kernel also halves the f32-atomic bytes (the other bound of this kernel).
template <int N> __device__ __forceinline__ void wg_wait_vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One 128 (c) x TN (n) tile of dW over the rows [m_begin, m_end): the body of conv_wgrad_pipe_kernel and conv_wgrad_group_kernel.
// PF (1x1 / unit stride, xcd_group): COOPERATIVE L2 PREFETCH.  The tiles of one m-slice run on one XCD in lockstep and share their
// operand rows (a G piece is staged by every c-tile, an X piece by every n-tile: 3.2x the compulsory bytes at 128 x 128 tiles).  In
// lockstep the sharers' LDS-DMA requests all arrive while the line is still in flight from HBM, so every one of them waits the
// full miss latency: the staged bytes move at the per-CU MISS rate (~20 GB/s: the CU's outstanding-miss budget x 128 B / latency)
// although only 1/3.2 of them leave the L2.  Here every workgroup pulls ITS SHARE of the group's unique lines (rows r with
// r % sharers == own index: 80 of the 1280 lines of a stage) into the XCD's L2 pf_dist stages ahead with one global_load_dword per
// wave (one lane per 128-byte line, result discarded), so that the LDS-DMA requests of the stage itself are L2 hits (~100 GB/s per
// CU).  Speed only: a late or missing prefetch costs a miss, never correctness.
// MEASURED (round 3, tools/bench_wgrad.py 0 0x800, interleaved in one process, B = 128, times incl. the slab reduce): NO GAIN --
// layer 3 (256 -> 1024 / 1024 -> 256 at 16 x 16) 59.6 / 57.1 us with the prefetch against 60.2 / 57.2 without, layer 2 (128 -> 512 at
// 32 x 32) 80.3 against 71.5 (the extra 80 line requests per stage compete with the stage's own 256), layer 4 unchanged.  The
// lockstep hit-under-miss picture is therefore NOT what bounds this kernel; kept opt-in (PPV_WGRAD_PF=<stages ahead>), default off.
// MEASURED (round 5, PPV_WGRAD_DEBUG=1 / 2 = loads only / K loop only, rocprofv3 kernel times, layer-3 1x1 shapes, 256-wide tile): whole
// kernel 35.4 us, loads only 24.8, K loop only 31.0 -- of which ~15 us are fixed per workgroup (first DMA round trip, the 128-KB slab
// store, launch), not the loop: the same tile rebuilt on 32-row stages in a ring of six buffers with the fragment reads software-
// pipelined across the stage barrier (the form that took the nine-tap 3x3 kernel from 97 to 76 us) measured 35.4 / 24.6 / 31.6 us: no
// change, removed again.  What is left to take is the per-workgroup fixed cost (16 stages per workgroup), i.e. the split count.
template <int TN, int NSTAGE, bool PF = false>
__device__ __forceinline__ void wgrad_pipe_body(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X, float* __restrict__ dst,
                                                const bf16_t* __restrict__ zero_page, const WgradGeom& g, long m_begin, long m_end,
                                                int n0, int tap, int c0, bool atomic, int tile_id = 0) {
    constexpr int NT = TN * 2, NW = NT / 64, NH = TN / 128;               // threads, waves, 128-column halves of G
    constexpr int HALF = 64 * 256;                                        // one [64][128] bf16 tile
    constexpr int STAGE_BYTES = (NH + 1) * HALF;
    constexpr int GI = (NH * 16) / NW, XI = 16 / NW, L = GI + XI + (PF ? 1 : 0);   // wave-instructions per thread per stage (+ the prefetch)
    constexpr int PFO = PF ? 1 : 0;                                       // the youngest prefetch may stay outstanding at a stage wait
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = tap / g.S, s = tap % g.S;
    const int nst = (int)((m_end - m_begin + 63) / 64);
    if (nst <= 0) return;
    const int HoWo = g.Ho * g.Wo;
    const int rli = lane >> 4, lch = lane & 15;
    const long zdG = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(G);
    const long zdX = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);
    const float rcp_howo = 1.0f / (float)HoWo, rcp_wo = 1.0f / (float)g.Wo;

    // prefetch roles (PF): threads 0 .. NT/2-1 cover the 64 rows x (TN*2/128) lines of the G piece, the rest the 64 x 2 lines of the X piece
    const int pf_nt = g.N / TN, pf_ct = g.Cs / 128;                       // sharers of an X piece / of a G piece (powers of two: launcher)
    const int pf_ni = n0 / TN, pf_ci = c0 / 128;
    unsigned pf_sink = 0;
    int st_next = 0;
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE_BYTES;
        const long mb = m_begin + (long)st_next * 64;
        ++st_next;
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int q = i * NW + wave, half = q >> 4, row = (q & 15) * 4 + rli;
            const long m = mb + row;
            const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
            const long off = (m >= m_end) ? zdG : g.chunked ? (((long)(n0 / 128 + half) * g.M + m) * 128 + gch * 8) * 2
                                                           : (m * g.N + n0 + half * 128 + gch * 8) * 2;
            GLDS16W(reinterpret_cast<const char*>(G) + off, sb + half * HALF + (q & 15) * 1024);
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int q = i * NW + wave, row = q * 4 + rli;
            const long m = mb + row;
            const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
            const int mm = (int)((m < m_end) ? m : m_begin);
            int b, rem, ho, wo;
            fast_divmod(mm, HoWo, rcp_howo, b, rem);
            fast_divmod(rem, g.Wo, rcp_wo, ho, wo);
            const int hs = ho * g.st + r - g.pad, ws = wo * g.st + s - g.pad;
            const bool ok = (m < m_end) & ((unsigned)hs < (unsigned)g.Hs) & ((unsigned)ws < (unsigned)g.Ws);
            const long off = !ok ? zdX : g.chunked ? (((long)(c0 / 128) * g.M + m) * 128 + gch * 8) * 2
                                                   : ((((long)b * g.Hs + hs) * g.Ws + ws) * g.Cs + c0 + gch * 8) * 2;
            GLDS16W(reinterpret_cast<const char*>(X) + off, sb + NH * HALF + q * 1024);
        }
        if constexpr (PF) {
            // one load per wave, every lane active (the hand-counted vmcnt waits below assume exactly one per stage): lanes without a
            // line of their own re-touch the zero page
            const long pm0 = mb + (long)g.pf_dist * 64;
            const char* src = reinterpret_cast<const char*>(zero_page);
            if (tid < NT / 2) {                                            // G piece: TN * 2 bytes per row = TN / 64 lines
                constexpr int LPR = TN / 64;
                const int line = tid * (64 * LPR) / (NT / 2) ;             // NT / 2 threads over 64 * LPR lines (1 : 1)
                const int row = line / LPR, part = line % LPR;
                const long m = pm0 + row;
                if (m < m_end && (row & (pf_ct - 1)) == pf_ci) src = reinterpret_cast<const char*>(G) + (m * g.N + n0 + part * 64) * 2;
            } else {                                                       // X piece: 256 bytes per row = 2 lines
                const int l2 = tid - NT / 2;
                if (l2 < 128) {
                    const int row = l2 >> 1, part = l2 & 1;
                    const long m = pm0 + row;
                    if (m < m_end && (row & (pf_nt - 1)) == pf_ni) src = reinterpret_cast<const char*>(X) + (m * g.Cs + c0 + part * 64) * 2;
                }
            }
            // the load lands LATER: its destination must stay reserved for the whole loop ("+v" on a loop-carried variable that is
            // consumed after the final vmcnt(0)), or the register allocator hands the register to a live value that the returning
            // load then overwrites
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(src) : "memory");
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;                              // wm: 64-column group of n, wn: of c

    auto compute = [&](int buf) {
        const char* tg = smem + buf * STAGE_BYTES + (wm >> 1) * HALF;
        const char* tx = smem + buf * STAGE_BYTES + NH * HALF;
        s16x4 alo[2][4], ahi[2][4], blo[2][4], bhi[2][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tr_issue(alo[0][mi], ahi[0][mi], tg, 0, (wm & 1) * 4 + mi, lane);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) tr_issue(blo[0][ni], bhi[0][ni], tx, 0, wn * 4 + ni, lane);
        tr_wait_all();
        // second half's reads fly under the first half's MFMAs
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tr_issue(alo[1][mi], ahi[1][mi], tg, 32, (wm & 1) * 4 + mi, lane);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) tr_issue(blo[1][ni], bhi[1][ni], tx, 32, wn * 4 + ni, lane);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1) tr_wait_all();
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = tr_pack(alo[kk][mi], ahi[kk][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bfr[ni] = tr_pack(blo[kk][ni], bhi[kk][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
        // the ring slot is re-filled after the next barrier: every read of it has retired (lgkmcnt(0) above)
    };

#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nst) stage(s0);
    int rd = 0, wr = (NSTAGE - 1) % NSTAGE;
    for (int t = 0; t < nst; ++t) {
        const int younger = min(nst, t + NSTAGE - 1) - (t + 1);
        if (NSTAGE >= 3 && younger >= NSTAGE - 2) wg_wait_vmcnt_le<(NSTAGE - 2) * L + PFO>();
        else if (NSTAGE >= 4 && younger == NSTAGE - 3) wg_wait_vmcnt_le<(NSTAGE >= 4 ? (NSTAGE - 3) * L : 0) + PFO>();
        else wg_wait_vmcnt_le<PFO>();
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nst && !(!PF && g.pf_dist == 102)) stage(wr);
        if (!(!PF && g.pf_dist == 101)) compute(rd);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
    }

    if constexpr (PF) {                                                   // retire the last prefetch, then release its register
        wg_wait_vmcnt_le<0>();
        asm volatile("" ::"v"(pf_sink));
    }
    if (g.native_slabs && !atomic) {
        // slab in accumulator order: record ((tile * NW + wave) * 16 + mi * 4 + ni) * 64 + lane = the lane's four rows (j) of MFMA tile
        // (mi, ni): 16 stores of 16 bytes per lane, each wave store one contiguous KB (the [N][R][S][C] form: 64 stores of 4 bytes
        // in 64-byte runs); wgrad_reduce_native_kernel knows the same map
        f32x4* d4 = reinterpret_cast<f32x4*>(dst) + ((long)(tile_id * NW + wave) * 16) * 64 + lane;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) d4[(mi * 4 + ni) * 64] = acc[mi][ni];
        return;
    }
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
                float* q = &dst[(long)n * wrow + (long)tap * g.Cs + c];
                if (!atomic) *q = acc[mi][ni][j];             // per-slice slab / final tensor: plain stores (4-5x the f32-atomic rate)
                else atomicAdd(q, acc[mi][ni][j]);
            }
}

template <int TN, int NSTAGE, bool PF = false>
__global__ __launch_bounds__(TN * 2, 1) void conv_wgrad_pipe_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                                    float* __restrict__ dW,
                                                                    const bf16_t* __restrict__ zero_page, WgradGeom g) {
    // XCD-aware decode of a 1-D grid: every tile of one m-slice runs on the SAME XCD at the same time, so the G rows
    // (shared by the c-tiles / taps) and the X rows (shared by the n-tiles) are fetched from HBM once and re-read from
    // that XCD's L2 (measured before: L2 hit rate 1 %, 2.3x over-fetch).  blocks b and b+8 share an XCD (speed only).
    const int ctiles = g.Cs / 128;
    const int tiles_y = g.R * g.S * ctiles, tiles = (g.N / TN) * tiles_y;
    int zslice, tl;
    if (g.xcd_group) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        zslice = (idx / tiles) * 8 + xcd;
        tl = idx % tiles;
    } else {
        zslice = blockIdx.x / tiles;
        tl = blockIdx.x % tiles;
    }
    if (zslice >= g.splits) return;
    const int n0 = (tl % (g.N / TN)) * TN, by = tl / (g.N / TN);
    const int tap = by / ctiles, c0 = (by % ctiles) * 128;
    const long m_begin = (long)zslice * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    if (g.xcc_slabs) {
        // One slab per XCD instead of one per m-slice: the slices that run on an XCD add into ITS slab with f32 atomics.  A slab's lines
        // then live in one L2 only (no cross-XCD line migration, which is what made atomics into a single accumulator slow), 8 slabs are
        // reduced instead of 24-48, and the index comes from the hardware (HW_REG_XCC_ID), so it is right for any block -> XCD mapping.
        const int xcc = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
        wgrad_pipe_body<TN, NSTAGE, PF>(G, X, dW + (long)xcc * g.slab_elems, zero_page, g, m_begin, m_end, n0, tap, c0, true);
        return;
    }
    wgrad_pipe_body<TN, NSTAGE, PF>(G, X, dW + (long)zslice * g.slab_elems, zero_page, g, m_begin, m_end, n0, tap, c0, g.slab_elems == 0, tl);
}

// ----------------------------------------------------------------------------- 1x1 / unit stride: linear addresses (round 6)
// The ISA of conv_wgrad_pipe_kernel<256, 3> showed ~500 instructions per 64-row stage and wave for 32 MFMAs: the general stage() rebuilds
// every source address from (m, tap, geometry) with a float-reciprocal divmod, a zero-page select and a `chunked` branch per load (six
// branch ladders per stage), and every transposed read recomputes row * 256 + (block ^ key) * 32: ~300 vector-ALU instructions x 4
// cycles per wave and stage beside 32 x 16 cycles of MFMA, two waves per SIMD -- the loop was bound by vector-instruction ISSUE, as the
// first nine-tap 3x3 kernel was.  For a 1x1 / unit-stride / unpadded convolution with M % 64 == 0 both operands are plain
// row-major matrices: a lane's six source pointers advance by a constant per stage, and its eight LDS read offsets are fixed (the
// second row group and the second k-half are immediate offsets of the DS instruction).  Same tile (256 n x 128 c, eight waves of 64 x 64),
// same ring, same slabs as wgrad_pipe_body<256, NSTAGE>: bit-identical results.
template <int NSTAGE>
__global__ __launch_bounds__(512, 1) void conv_wgrad_lin_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                                float* __restrict__ dW, WgradGeom g) {
    constexpr int TN = 256, NW = 8, NH = 2, HALF = 64 * 256, STAGE_BYTES = (NH + 1) * HALF, GI = 4, XI = 2, L = GI + XI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ntile = g.N / TN, tiles = ntile * (g.Cs / 128);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int zslice = (idx / tiles) * 8 + xcd, tl = idx % tiles;
    if (zslice >= g.splits) return;
    const int n0 = (tl % ntile) * TN, c0 = (tl / ntile) * 128;
    const long m_begin = (long)zslice * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    const int nst = (int)((m_end - m_begin) >> 6);
    if (nst <= 0) return;

    // ---- staging: per lane six source pointers, advanced by one 64-row stage per call
    const int rli = lane >> 4, lch = lane & 15;
    const char* pg[GI];
    const char* px[XI];
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        const int q = i * NW + wave, half = q >> 4, row = (q & 15) * 4 + rli;
        const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
        pg[i] = reinterpret_cast<const char*>(G) + ((m_begin + row) * g.N + n0 + half * 128 + gch * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (i * NW + wave) * 4 + rli;
        const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
        px[i] = reinterpret_cast<const char*>(X) + ((m_begin + row) * g.Cs + c0 + gch * 8) * 2;
    }
    const long step_g = 128L * g.N, step_x = 128L * g.Cs;                 // 64 rows x 2 bytes
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int q = i * NW + wave;
            GLDS16W(pg[i], sb + (q >> 4) * HALF + (q & 15) * 1024);
            pg[i] += step_g;
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            GLDS16W(px[i], sb + NH * HALF + (i * NW + wave) * 1024);
            px[i] += step_x;
        }
    };

    // ---- fragment reads: fixed per-lane offsets inside a stage
    const int wm = wave >> 1, wn = wave & 1;                              // wm: 64-column group of n, wn: of c
    unsigned ao[4], bo[4];
    {
        const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
        const int r0 = 8 * fg + fq, key = trkey(r0);
        const unsigned s0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ao[mi] = s0 + (wm >> 1) * HALF + r0 * 256 + ((((wm & 1) * 4 + mi) ^ key) << 5) + fp * 8;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bo[ni] = s0 + NH * HALF + r0 * 256 + (((wn * 4 + ni) ^ key) << 5) + fp * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const unsigned bofs = (unsigned)buf * STAGE_BYTES;
        s16x4 alo[2][4], ahi[2][4], blo[2][4], bhi[2][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { tr_read_off<0>(alo[0][mi], ao[mi] + bofs); tr_read_off<1024>(ahi[0][mi], ao[mi] + bofs); }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { tr_read_off<0>(blo[0][ni], bo[ni] + bofs); tr_read_off<1024>(bhi[0][ni], bo[ni] + bofs); }
        tr_wait_all();
        // second k-half's reads fly under the first half's MFMAs
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { tr_read_off<8192>(alo[1][mi], ao[mi] + bofs); tr_read_off<8192 + 1024>(ahi[1][mi], ao[mi] + bofs); }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { tr_read_off<8192>(blo[1][ni], bo[ni] + bofs); tr_read_off<8192 + 1024>(bhi[1][ni], bo[ni] + bofs); }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1) tr_wait_all();
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = tr_pack(alo[kk][mi], ahi[kk][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bfr[ni] = tr_pack(blo[kk][ni], bhi[kk][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    };

#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nst) stage(s0);
    int rd = 0, wr = NSTAGE - 1;
    for (int t = 0; t < nst; ++t) {
        // stages younger than t still in flight: NSTAGE - 2 in the steady state, fewer at the tail
        if (t + NSTAGE - 1 <= nst) wg_wait_vmcnt_le<(NSTAGE - 2) * L>();
        else if (NSTAGE >= 4 && t + NSTAGE - 2 <= nst) wg_wait_vmcnt_le<(NSTAGE >= 4 ? (NSTAGE - 3) * L : 0)>();
        else wg_wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nst) stage(wr);
        compute(rd);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
    }
    // slab in accumulator order (wgrad_reduce_native_kernel<0>'s map): see wgrad_pipe_body
    f32x4* d4 = reinterpret_cast<f32x4*>(dW + (long)zslice * g.slab_elems) + ((long)(tl * NW + wave) * 16) * 64 + lane;
    if (g.pf_dist == 777) {                                    // DIAGNOSTIC (PPV_WGRAD_NOSLAB=1, results wrong): what the slab traffic costs the step
        if (acc[0][0][0] == 123456.789f) d4[0] = acc[0][0];
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) d4[(mi * 4 + ni) * 64] = acc[mi][ni];
}

// The same tile with the two waves of every SIMD in OPPOSITE phases (round 6; "ping-pong"): a stage is a READ phase (the lane's share of
// the DMA for stage t + 2, all 32 transposed fragment reads of stage t, wait) and an MFMA phase (32 MFMAs), a barrier after each; waves
// 4-7 run one barrier behind waves 0-3, so while one wave of a SIMD issues its MFMAs the other one's LDS reads and DMA requests fill the
// issue slots between them -- in conv_wgrad_lin_kernel all eight waves read together behind the stage barrier (the first k-half's reads
// exposed) and then all issue MFMAs together.  Barrier instance k: group A (waves 0-3) reads stage t behind instance 2t, group B behind
// 2t + 1.  Visibility of stage t + 1: every wave waits for ITS DMA of stage t + 1 inside read phase t, i.e. before instance 2t + 1 (A) /
// 2t + 2 (B), and the first read of stage t + 1 comes behind instance 2t + 2.  Re-use of ring slot (t + 2) % 3 = (t - 1) % 3 in read
// phase t: the other group's reads of stage t - 1 retired (lgkmcnt(0)) before the instance this phase starts behind.  Results bit-identical.
template <int NSTAGE>
__global__ __launch_bounds__(512, 1) void conv_wgrad_lin_pp_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                                   float* __restrict__ dW, WgradGeom g) {
    static_assert(NSTAGE == 3, "the waits below are counted for a ring of three");
    constexpr int TN = 256, NW = 8, NH = 2, HALF = 64 * 256, STAGE_BYTES = (NH + 1) * HALF, GI = 4, XI = 2, L = GI + XI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ntile = g.N / TN, tiles = ntile * (g.Cs / 128);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int zslice = (idx / tiles) * 8 + xcd, tl = idx % tiles;
    if (zslice >= g.splits) return;
    const int n0 = (tl % ntile) * TN, c0 = (tl / ntile) * 128;
    const long m_begin = (long)zslice * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    const int nst = (int)((m_end - m_begin) >> 6);
    if (nst <= 0) return;

    const int rli = lane >> 4, lch = lane & 15;
    const char* pg[GI];
    const char* px[XI];
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        const int q = i * NW + wave, half = q >> 4, row = (q & 15) * 4 + rli;
        const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
        pg[i] = reinterpret_cast<const char*>(G) + ((m_begin + row) * g.N + n0 + half * 128 + gch * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (i * NW + wave) * 4 + rli;
        const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
        px[i] = reinterpret_cast<const char*>(X) + ((m_begin + row) * g.Cs + c0 + gch * 8) * 2;
    }
    const long step_g = 128L * g.N, step_x = 128L * g.Cs;
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int q = i * NW + wave;
            GLDS16W(pg[i], sb + (q >> 4) * HALF + (q & 15) * 1024);
            pg[i] += step_g;
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            GLDS16W(px[i], sb + NH * HALF + (i * NW + wave) * 1024);
            px[i] += step_x;
        }
    };
    const int wm = wave >> 1, wn = wave & 1;
    unsigned ao[4], bo[4];
    {
        const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
        const int r0 = 8 * fg + fq, key = trkey(r0);
        const unsigned s0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ao[mi] = s0 + (wm >> 1) * HALF + r0 * 256 + ((((wm & 1) * 4 + mi) ^ key) << 5) + fp * 8;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bo[ni] = s0 + NH * HALF + r0 * 256 + (((wn * 4 + ni) ^ key) << 5) + fp * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0);
    if (nst > 1) { stage(1); wg_wait_vmcnt_le<L>(); } else wg_wait_vmcnt_le<0>();     // own share of stage 0 has landed
    const bool late = wave >= 4;                                                          // group B: one barrier behind
    if (late) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    int rd = 0, wr = NSTAGE - 1;
    for (int t = 0; t < nst; ++t) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- read phase
        if (t + 2 < nst) stage(wr);
        const unsigned bofs = (unsigned)rd * STAGE_BYTES;
        s16x4 alo[2][4], ahi[2][4], blo[2][4], bhi[2][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { tr_read_off<0>(alo[0][mi], ao[mi] + bofs); tr_read_off<1024>(ahi[0][mi], ao[mi] + bofs); }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { tr_read_off<0>(blo[0][ni], bo[ni] + bofs); tr_read_off<1024>(bhi[0][ni], bo[ni] + bofs); }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { tr_read_off<8192>(alo[1][mi], ao[mi] + bofs); tr_read_off<8192 + 1024>(ahi[1][mi], ao[mi] + bofs); }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) { tr_read_off<8192>(blo[1][ni], bo[ni] + bofs); tr_read_off<8192 + 1024>(bhi[1][ni], bo[ni] + bofs); }
        if (t + 2 < nst) wg_wait_vmcnt_le<L>();          // own share of stage t + 1 has landed (stage t + 2 may still fly)
        else wg_wait_vmcnt_le<0>();
        tr_wait_all();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA phase
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = tr_pack(alo[kk][mi], ahi[kk][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bfr[ni] = tr_pack(blo[kk][ni], bhi[kk][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();             // every wave has executed 2 nst + 1 barriers
    f32x4* d4 = reinterpret_cast<f32x4*>(dW + (long)zslice * g.slab_elems) + ((long)(tl * NW + wave) * 16) * 64 + lane;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) d4[(mi * 4 + ni) * 64] = acc[mi][ni];
}

// TWO 1x1 / unit-stride weight gradients of DIFFERENT shapes in one launch (round 6): conv1 of the bottleneck that just finished its
// backward and conv3 of the next one to run become ready within ~75 us of each other (trunk_plan.hip).  Launched one after the other each
// fills the chip with 8 tiles x 32 m-slices: 16 stages per workgroup, of whose ~35 us ~15 are fixed (first DMA round trip, address
// set-up, the 128-KB slab store: tools/look_timeline.py measures the same 4.8 us of 23 on the forward tile) and 32 slabs per problem are
// written and read back.  Together each problem takes half the workgroups: 16 m-slices of 32 stages -- half the slabs, half the
// prologues / slab stores per unit of work, one launch and one reduce launch instead of two each.  Body: wgrad_pipe_body, unchanged.
struct WgradPair {
    const bf16_t* G[2];
    const bf16_t* X[2];
    float* slabs[2];
    WgradGeom g[2];
    int wgs0;                // workgroups of problem 0 (a multiple of 8: the XCD decode of problem 1 starts on XCD 0)
};

template <int TN, int NSTAGE>
__global__ __launch_bounds__(TN * 2, 1) void conv_wgrad_pair_kernel(WgradPair p, const bf16_t* __restrict__ zero_page) {
    const int which = (int)blockIdx.x >= p.wgs0 ? 1 : 0;
    const int bx = which ? (int)blockIdx.x - p.wgs0 : (int)blockIdx.x;
    const WgradGeom g = which ? p.g[1] : p.g[0];
    const bf16_t* G = which ? p.G[1] : p.G[0];
    const bf16_t* X = which ? p.X[1] : p.X[0];
    float* dW = which ? p.slabs[1] : p.slabs[0];
    const int ctiles = g.Cs / 128;
    const int tiles = (g.N / TN) * ctiles;
    const int xcd = bx & 7, idx = bx >> 3;
    const int zslice = (idx / tiles) * 8 + xcd, tl = idx % tiles;
    if (zslice >= g.splits) return;
    const int n0 = (tl % (g.N / TN)) * TN, c0 = (tl / (g.N / TN)) * 128;
    const long m_begin = (long)zslice * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    wgrad_pipe_body<TN, NSTAGE>(G, X, dW + (long)zslice * g.slab_elems, zero_page, g, m_begin, m_end, n0, 0, c0, false, tl);
}

// Several weight gradients of ONE shape (1x1, unit stride: the conv1 / conv3 of a layer's bottlenecks) in one launch, each reduced
// over ALL its rows by one workgroup per tile: no split-M slabs, no reduce launch, the result goes straight to the torch tensor.
// tools/micro/wgrad_longk.py: the slab form costs 59-64 us per layer-3 problem (kernel + reduce), the un-split steady state 39-43 us.
// Problem p runs on XCD p % 8 with all its tiles at once, so the tiles walk the same rows of G and X through that XCD's L2.
struct WgradGroupPtrs {
    const bf16_t* G[24];
    const bf16_t* X[24];
    float* out[24];
};

template <int TN, int NSTAGE>
__global__ __launch_bounds__(TN * 2, 1) void conv_wgrad_group_kernel(WgradGroupPtrs ptrs, int P, const bf16_t* __restrict__ zero_page, WgradGeom g) {
    const int tiles = (g.N / TN) * (g.Cs / 128);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int p = (idx / tiles) * 8 + xcd, tl = idx % tiles;
    if (p >= P) return;
    const int n0 = (tl % (g.N / TN)) * TN, c0 = (tl / (g.N / TN)) * 128;
    wgrad_pipe_body<TN, NSTAGE>(ptrs.G[p], ptrs.X[p], ptrs.out[p], zero_page, g, 0, g.M, n0, 0, c0, false);
}

// ----------------------------------------------------------------------------- streamed weight gradient (one tap per workgroup)
// conv_wgrad_pipe_kernel<128, 2> keeps ONE 32-KB stage in flight per workgroup: with ~2 us from request to landing under load it
// is latency-bound (tools/bench_wgrad.py: 53 us for the layer-3 1x1 shapes against 14 us of HBM time and 16 us of L2 -> LDS fill),
// and deeper rings of the same form did not help because every stage costs all waves a vmcnt wait and a barrier.  Here the roles
// are split as in conv_stream.hip: 4 loader waves keep a ring of NSLOT 32-row stages full (global_load_lds, run-ahead bounded only
// by free slots), NCW consumer waves (64 n x 64 c each) read transposed fragments and issue MFMAs; progress travels in per-wave
// LDS words, there is no barrier in the m loop, so the consumers never wait for a request they did not issue.
// MEASURED (tools/bench_wgrad.py, tools/wgrad_timeline.py, B = 128): NOT faster, kept as variant 7 only.  The stamps show stages
// landing every 1.6 us per CU whatever the flight depth (2 stages or 5): 24 KB per 1.6 us = 15 GB/s per CU, i.e. the CU's LDS-DMA
// path takes ~65 ns per 1-KB global_load_lds instruction from HBM-resident rows, and four loader waves do not add up (the guide's
// one-loader figure is 25 GB/s).  At 256 workgroups the layer-3 1x1 shapes take 69 us against 65-70 us for the pipelined kernel,
// and in the whole step (weight gradients beside the data-gradient chain) the 147-KB ring that keeps a CU to itself costs 4-8 %
// (4769 / 4552 images/s at 256 / 384 workgroups against 4963 with the 64-KB two-stage ring).
constexpr int WS_NLW = 4;
typedef unsigned ws_u32x4 __attribute__((ext_vector_type(4)));
// progress words are read / written through address-space-3 pointers: through a generic `volatile unsigned*` hipcc emits
// flat_load / flat_store ... sc0 sc1 + s_waitcnt vmcnt(0), i.e. every poll of the LOADER waves drained their whole LDS-DMA ring (round
// 3 finding: that, not the hardware, is why rounds 1-2 saw "one stage per 1.6 us whatever the ring depth" in this kernel)
__device__ __forceinline__ ws_u32x4 ws_poll4(const volatile unsigned* p) {
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) ws_u32x4*>((const volatile __attribute__((address_space(3))) unsigned*)p);
}
__device__ __forceinline__ void ws_post(volatile unsigned* p, unsigned v) { *((volatile __attribute__((address_space(3))) unsigned*)p) = v; }
#ifdef PPV_STAMPS   // diagnostic build (csrc/build_stamps.sh, tools/wgrad_timeline.py): phase stamps of consumer wave 0 and loader wave 0
extern __device__ unsigned long long* g_stamps;
#define WS_STAMP_DECL unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define WS_STAMP(i) do { stamp_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WS_STAMP_FLUSH(base, who) do { if (g_stamps && threadIdx.x == (who)) { for (int i_ = 0; i_ < 8; ++i_) g_stamps[(long)blockIdx.x * 16 + (base) + i_] = stamp_[i_]; } } while (0)
#else
#define WS_STAMP_DECL do { } while (0)
#define WS_STAMP(i) do { } while (0)
#define WS_STAMP_FLUSH(base, who) do { } while (0)
#endif

template <int TN>
__global__ __launch_bounds__((TN / 32 + WS_NLW) * 64, (TN / 32 + WS_NLW) / 4) void conv_wgrad_stream_kernel(
    const bf16_t* __restrict__ G, const bf16_t* __restrict__ X, float* __restrict__ dW, const bf16_t* __restrict__ zero_page, WgradGeom g) {
    constexpr int NCW = TN / 32, NH = TN / 128;                           // consumer waves (wn 0..TN/64-1, wc 0..1); 128-column halves of G
    constexpr int HALF = 32 * 256;                                        // one [32 m][128] bf16 tile
    constexpr int STAGE = (NH + 1) * HALF;                                // G halves, then the X tile
    constexpr int NSLOT = TN == 256 ? 6 : 8;                              // 144 KB / 128 KB of ring
    constexpr int NI = (NH + 1) * 8, LI = NI / WS_NLW;                    // 1-KB LDS-DMA instructions per stage / per loader wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile unsigned* sLanded = reinterpret_cast<volatile unsigned*>(smem + NSLOT * STAGE);   // [4] stages each loader wave has landed
    volatile unsigned* sDone = sLanded + 4;                                                      // [NCW] stages each consumer has read
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctiles = g.Cs / 128;
    const int tiles_y = g.R * g.S * ctiles, tiles = (g.N / TN) * tiles_y;
    int zslice, tl;
    if (g.xcd_group) {                                                    // every tile of one m-slice on the same XCD (G / X rows from its L2)
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        zslice = (idx / tiles) * 8 + xcd;
        tl = idx % tiles;
    } else {
        zslice = blockIdx.x / tiles;
        tl = blockIdx.x % tiles;
    }
    if (zslice >= g.splits) return;
    const int n0 = (tl % (g.N / TN)) * TN, by = tl / (g.N / TN);
    const int tap = by / ctiles, c0 = (by % ctiles) * 128;
    const int r = tap / g.S, s = tap % g.S;
    const long m_begin = (long)zslice * g.stages_per_split * 32;          // stages of 32 rows here
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 32);
    const int nst = (int)((m_end - m_begin + 31) / 32);
    if (nst <= 0) return;
    WS_STAMP_DECL;
    WS_STAMP(0);
    if (tid < 4 + NCW) sLanded[tid] = 0;
    __syncthreads();

    if (wave >= NCW) {
        // ------------------------------------------------------------ loader waves
        const int lw = wave - NCW;
        const int rli = lane >> 4, lch = lane & 15;
        const int HoWo = g.Ho * g.Wo;
        const float rcp_howo = 1.0f / (float)HoWo, rcp_wo = 1.0f / (float)g.Wo;
        const long zdG = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(G);
        const long zdX = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);
        const bool flat = g.R == 1 && g.S == 1 && g.st == 1 && g.pad == 0;
        // Event loop: issue stages as far ahead as the ring has free slots (up to NSLOT - 1 stages = 120 KB in flight per CU: what
        // it takes to cover 2-3 us of loaded HBM latency at the 40-70 GB/s a CU can take in), publish a stage when its requests
        // have landed (counted vmcnt: stages land in issue order).  The first version published stage t - 1 right after issuing
        // stage t, which capped the flight depth at two stages: 1.5 us per stage instead of 0.35.
        auto wait_oldest = [&](int younger) {                                // all but the `younger` youngest stages' requests done
            switch (younger) {
                case 0: wg_wait_vmcnt_le<0>(); break;
                case 1: wg_wait_vmcnt_le<LI>(); break;
                case 2: wg_wait_vmcnt_le<2 * LI>(); break;
                case 3: wg_wait_vmcnt_le<3 * LI>(); break;
                case 4: wg_wait_vmcnt_le<4 * LI>(); break;
                case 5: wg_wait_vmcnt_le<5 * LI>(); break;
                case 6: wg_wait_vmcnt_le<6 * LI>(); break;
                default: wg_wait_vmcnt_le<7 * LI>(); break;
            }
        };
        static_assert(7 * LI < 64, "vmcnt is a 6-bit counter");
        auto consumed = [&]() {
            const ws_u32x4 a = ws_poll4(sDone);
            unsigned lo = min(min(a.x, a.y), min(a.z, a.w));
            if (NCW == 8) {
                const ws_u32x4 b = ws_poll4(sDone + 4);
                lo = min(lo, min(min(b.x, b.y), min(b.z, b.w)));
            }
            return (int)__builtin_amdgcn_readfirstlane(lo);
        };
        int issued = 0, published = 0, done = 0, slot = 0;
        while (published < nst) {
            if (issued < nst && issued >= done + NSLOT) done = consumed();   // ring looks full: refresh the consumers' progress
            if (issued < nst && issued < done + NSLOT && issued - published < 8) {
                asm volatile("" ::: "memory");
                char* sb = smem + slot * STAGE;
                const long mb = m_begin + (long)issued * 32;
#pragma unroll
                for (int i = 0; i < LI; ++i) {
                    const int q = i * WS_NLW + lw;                          // instruction 0 .. NI-1: tile q / 8, rows (q % 8) * 4 ..
                    const int tile = q >> 3, row = (q & 7) * 4 + rli;
                    const long m = mb + row;
                    const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
                    if (tile < NH) {
                        const long off = (m < m_end) ? (m * g.N + n0 + tile * 128 + gch * 8) * 2 : zdG;
                        GLDS16W(reinterpret_cast<const char*>(G) + off, sb + tile * HALF + (q & 7) * 1024);
                    } else {
                        long off = zdX;
                        if (m < m_end) {
                            if (flat) {
                                off = (m * g.Cs + c0 + gch * 8) * 2;
                            } else {
                                int b, rem, ho, wo;
                                fast_divmod((int)m, HoWo, rcp_howo, b, rem);
                                fast_divmod(rem, g.Wo, rcp_wo, ho, wo);
                                const int hs = ho * g.st + r - g.pad, ws = wo * g.st + s - g.pad;
                                if (((unsigned)hs < (unsigned)g.Hs) & ((unsigned)ws < (unsigned)g.Ws))
                                    off = ((((long)b * g.Hs + hs) * g.Ws + ws) * g.Cs + c0 + gch * 8) * 2;
                            }
                        }
                        GLDS16W(reinterpret_cast<const char*>(X) + off, sb + NH * HALF + (q & 7) * 1024);
                    }
                }
                slot = slot + 1 == NSLOT ? 0 : slot + 1;
                ++issued;
                if (issued == 1) WS_STAMP(1);
                if (issued == 5) WS_STAMP(2);
                continue;
            }
            if (published < issued) {                                        // nothing to issue right now: retire the oldest stage in flight
                wait_oldest(issued - published - 1);
                ++published;
                if (lane == 0) ws_post(sLanded + lw, (unsigned)published);
                if (published == 1) WS_STAMP(3);
                if (published == 5) WS_STAMP(4);
                if (published == 13) WS_STAMP(5);
            } else {
                __builtin_amdgcn_s_sleep(2);                                 // ring full, everything landed: the consumers are behind
            }
        }
        WS_STAMP(7);
        WS_STAMP_FLUSH(8, NCW * 64);
        return;
    }
    // ---------------------------------------------------------------- consumer waves
    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave >> 1, wc = wave & 1;                              // 64-column group of n, of c
    unsigned landed = 0;
    auto wait_landed = [&](int j) {                                       // stage j has landed (all four loader waves)
        while (landed <= (unsigned)j) {
            const ws_u32x4 a = ws_poll4(sLanded);
            landed = __builtin_amdgcn_readfirstlane(min(min(a.x, a.y), min(a.z, a.w)));
            if (landed <= (unsigned)j) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    // one stage = one K = 32 MFMA step per wave: reads, wait, release the slot, 16 MFMAs.  No software pipelining inside a wave
    // (the second fragment set does not fit the 168-VGPR budget of three waves per SIMD: it spilled, and scratch accesses sit in
    // the vmcnt queue): the two consumer waves of a SIMD are not in lockstep, one's reads fly under the other's MFMAs.
    int slot = 0;
    for (int j = 0; j < nst; ++j) {
        wait_landed(j);
        if (j == 0) WS_STAMP(1);
        if (j == 1) WS_STAMP(2);
        if (j == 5) WS_STAMP(3);
        if (j == 13) WS_STAMP(4);
        const char* tg = smem + slot * STAGE + (wn >> 1) * HALF;
        const char* tx = smem + slot * STAGE + NH * HALF;
        s16x4 alo[4], ahi[4], blo[4], bhi[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tr_issue(alo[mi], ahi[mi], tg, 0, (wn & 1) * 4 + mi, lane);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) tr_issue(blo[ni], bhi[ni], tx, 0, wc * 4 + ni, lane);
        tr_wait_all();                                                      // fragments of stage j are in registers
        if (lane == 0) ws_post(sDone + wave, (unsigned)(j + 1));
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = tr_pack(alo[mi], ahi[mi]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bfr[ni] = tr_pack(blo[ni], bhi[ni]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);   // D[c][n]
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    WS_STAMP(5);
    // operands swapped (x fragment in the A position): a lane's four accumulator registers are four consecutive c of one n, i.e.
    // 16 contiguous bytes of the slab -- 16 dwordx4 stores per lane instead of 64 dword stores (the slab write of a workgroup was
    // 128 KB of 4-byte stores: ~15 us of store issue)
    const int fr = lane & 15, fq = lane >> 4;
    const long wrow = (long)g.R * g.S * g.Cs;
    float* dst = dW + (long)zslice * g.slab_elems;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + mi * 16 + fr;
            const int c = c0 + wc * 64 + ni * 16 + fq * 4;
            *reinterpret_cast<f32x4*>(&dst[(long)n * wrow + (long)tap * g.Cs + c]) = acc[mi][ni];
        }
    WS_STAMP(7);
    WS_STAMP_FLUSH(0, 0);
}

// ----------------------------------------------------------------------------- 3x3 / stride 1 / pad 1 wgrad, taps fused
// conv_wgrad_pipe_kernel re-reads the G rows of an m-slice once per (tap, c-tile): 18 times for the layer-3 3x3 convs, and
// its 9 taps fetch 9 shifted copies of the same X rows -- the kernel ran at the L2->LDS rate (453 MB per launch).  Here one
// workgroup owns 128 (n) x 128 (c) x THREE taps (one kernel row r, s = 0..2):
//   * the 64-row G stage is staged once and feeds all three taps;
//   * X is staged ONCE as a halo tile: the 64/W image rows of the stage shifted by r-1, each with its W+2 columns
//     (zero page outside the image), pixel-major [pixel][128 c]; tap s reads row (k / W) * (W + 2) + k % W + s of it
//     through the same transposed ds_read_b64_tr_b16 fragments (the swizzle key follows the halo row);
//   => 92 MAC per staged byte instead of 44, 12 tiles of 3 taps instead of 18 of one.
// 8 waves: wave (wn, wc) owns 64 n x 32 c of every tap: acc[3][4][2] (96 VGPRs).  W in {8, 16, 32, 64}, H*W % 64 == 0.
constexpr int W3_XSLOTS = 24;                                  // 4-pixel wave-instructions reserved per stage (<= 96 pixels)
constexpr int W3_STAGE = 64 * 256 + W3_XSLOTS * 1024;          // G tile + halo tile

template <int NSTAGE>
__global__ __launch_bounds__(512, 1) void conv_wgrad3x3_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                               float* __restrict__ dW, const bf16_t* __restrict__ zero_page,
                                                               WgradGeom g, int log2W) {
    constexpr int NW = 8, GI = 16 / NW, XI = W3_XSLOTS / NW, L = GI + XI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int W = 1 << log2W, WP = W + 2, rows_ps = 64 >> log2W;          // image rows per 64-pixel stage
    const int npix = rows_ps * WP, nslots = (npix + 3) >> 2;
    const int ctiles = g.Cs / 128, ntiles = g.N / 128;
    const int tiles = ntiles * ctiles * 3;
    int zslice, tl;
    if (g.xcd_group) {                                         // all tiles of an m-slice on one XCD: G / X rows come from its L2
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        zslice = (idx / tiles) * 8 + xcd;
        tl = idx % tiles;
    } else {
        zslice = blockIdx.x / tiles;
        tl = blockIdx.x % tiles;
    }
    if (zslice >= g.splits) return;
    const int n0 = (tl % ntiles) * 128, by = tl / ntiles;
    const int r = by / ctiles, c0 = (by % ctiles) * 128;
    const long m_begin = (long)zslice * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    int nst = (int)((m_end - m_begin + 63) / 64);
    if (nst <= 0) return;
    const int HoWo = g.Ho * g.Wo;
    const int rli = lane >> 4, lch = lane & 15;
    const long zdX = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);

    int st_next = 0;
    auto stage = [&](int buf) {
        char* sb = smem + buf * W3_STAGE;
        const long mb = m_begin + (long)st_next * 64;
        ++st_next;
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int q = i * NW + wave, row = q * 4 + rli;
            const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
            GLDS16W(reinterpret_cast<const char*>(G) + ((mb + row) * g.N + n0 + gch * 8) * 2, sb + q * 1024);
        }
        const int b = (int)(mb / HoWo), row0 = (int)(mb % HoWo) >> log2W;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int q = i * NW + wave;
            if (q >= nslots) q -= nslots;                                 // spare slots repeat a valid one (same bytes, same place)
            const int pp = q * 4 + rli;
            const int pr = pp / WP, pc = pp - pr * WP;
            const int ho = row0 + pr + r - 1, wo = pc - 1;
            const int gch = (((lch >> 1) ^ trkey(pp)) << 1) | (lch & 1);
            const bool ok = (pp < npix) & ((unsigned)ho < (unsigned)g.Hs) & ((unsigned)wo < (unsigned)W);
            const long off = ok ? ((((long)b * g.Hs + ho) * W + wo) * g.Cs + c0 + gch * 8) * 2 : zdX;
            GLDS16W(reinterpret_cast<const char*>(X) + off, sb + 64 * 256 + q * 1024);
        }
    };

    f32x4 acc[3][4][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[s][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave >> 2, wc = wave & 3;

    // transposed fragment of the halo tile for tap s: k-row r0 of the stage lives in halo row (r0 / W) * (W + 2) + r0 % W + s
    auto x_issue = [&](s16x4& lo, s16x4& hi, const char* tile, int k0, int s, int colblock) {
        const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int k = k0 + 8 * gq + q;
        const int h0 = (k >> log2W) * WP + (k & (W - 1)) + s, h1 = h0 + 4;           // k % 8 < 4: k + 4 stays in the image row
        const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tile + h0 * 256 + (colblock ^ trkey(h0)) * 32 + p * 8);
        const unsigned a1 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tile + h1 * 256 + (colblock ^ trkey(h1)) * 32 + p * 8);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    };

    // six units per stage (2 k-halves x 3 taps, 8 MFMAs each); the fragments of unit u+1 (and the G fragments of the second
    // k-half) are in flight under the MFMAs of unit u
    auto compute = [&](int buf) {
        const char* tg = smem + buf * W3_STAGE;
        const char* tx = tg + 64 * 256;
        s16x4 alo[2][4], ahi[2][4], blo[2][2], bhi[2][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) tr_issue(alo[0][mi], ahi[0][mi], tg, 0, wn * 4 + mi, lane);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) x_issue(blo[0][ni], bhi[0][ni], tx, 0, 0, wc * 2 + ni);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int kk = u / 3, s = u % 3;
            tr_wait_all();
            if (u + 1 < 6) {
                const int kk1 = (u + 1) / 3, s1 = (u + 1) % 3;
                if (s1 == 0) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) tr_issue(alo[kk1][mi], ahi[kk1][mi], tg, kk1 * 32, wn * 4 + mi, lane);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) x_issue(blo[(u + 1) & 1][ni], bhi[(u + 1) & 1][ni], tx, kk1 * 32, s1, wc * 2 + ni);
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 af[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = tr_pack(alo[kk][mi], ahi[kk][mi]);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bf16x8 bf = tr_pack(blo[u & 1][ni], bhi[u & 1][ni]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[s][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bf, acc[s][mi][ni], 0, 0, 0);
            }
        }
    };

#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nst) stage(s0);
    int rd = 0, wr = (NSTAGE - 1) % NSTAGE;
    for (int t = 0; t < nst; ++t) {
        const int younger = min(nst, t + NSTAGE - 1) - (t + 1);
        if (younger >= NSTAGE - 2) wg_wait_vmcnt_le<(NSTAGE - 2) * L>();
        else if (NSTAGE >= 4 && younger == NSTAGE - 3) wg_wait_vmcnt_le<(NSTAGE >= 4 ? (NSTAGE - 3) * L : 0)>();
        else wg_wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nst) stage(wr);
        compute(rd);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
    }

    const int fr = lane & 15, fq = lane >> 4;
    const long wrow = 9L * g.Cs;
    float* dst = dW + (long)zslice * g.slab_elems;
    if (g.native_slabs) {                                      // accumulator-order slab (see wgrad_pipe_body): 24 records per lane
        f32x4* d4 = reinterpret_cast<f32x4*>(dst) + ((long)(tl * 8 + wave) * 24) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) d4[(s * 8 + mi * 2 + ni) * 64] = acc[s][mi][ni];
        return;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + mi * 16 + fq * 4 + j;
                    const int c = c0 + wc * 32 + ni * 16 + fr;
                    dst[(long)n * wrow + (long)(r * 3 + s) * g.Cs + c] = acc[s][mi][ni][j];
                }
}

// ----------------------------------------------------------------------------- 3x3 / stride 1 / pad 1 wgrad, ALL NINE taps per workgroup (round 5)
// conv_wgrad3x3_kernel stages 40 KB per 64-row stage for 3 x 128 x 128 outputs: a G row is fetched by 3 kernel rows x Cs / 128 tiles, an X
// row by 3 x N / 128 (each time with its halo).  The launch is bound by the bytes a CU can pull through its L2 -> LDS path (24 GB/s per
// CU measured: 1.7 MB per workgroup in 72 us at the layer-3 shape), not by the matrix pipe (21 % busy).  Here a workgroup owns
// 128 (n) x 64 (c) x NINE taps: ONE halo tile of the stage's image rows -1 .. rows_ps (each with W + 2 columns, [pixel][64 c] = 128-byte
// rows) serves every tap, the G stage is fetched once per c-tile instead of once per (kernel row, c-tile): 16 + ~14 KB per stage for
// 1.5 x the MFMA work (184 MAC per staged byte against 92), 8 tiles instead of 12 for layer 3.  Wave (wn, wc) owns 64 n x 16 c of every
// tap: acc[9][4] (144 VGPRs); the four G fragments of a k-half are reused by nine taps (26 transposed reads per 36 MFMAs).
// Halo rows are 128 bytes = 32 banks wide: the 32-byte block of a pixel row is XOR-swizzled with key2(h) = bit 1 | bit 3 << 1 of the halo
// pixel index, which together with the row's parity (the half of the 64 banks it starts in) separates the eight pixel rows a
// ds_read_b64_tr_b16 pass touches.  W in {8, 16, 32} (W = 64 keeps the three-tap kernel: its 198-pixel halo), H * W % 64 == 0.
__device__ __forceinline__ int trkey2(int h) { return ((h >> 1) & 1) | (((h >> 3) & 1) << 1); }

template <int NSTAGE, int LOG2W>
__global__ __launch_bounds__(512, 1) void conv_wgrad3x3_t9_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                                  float* __restrict__ dW, const bf16_t* __restrict__ zero_page,
                                                                  WgradGeom g) {
    constexpr int log2W = LOG2W, XI = LOG2W == 5 ? 3 : 2;
    constexpr int NW = 8, GI = 16 / NW, L = GI + XI;
    constexpr int XBYTES = XI * NW * 1024, STAGE = 64 * 256 + XBYTES;        // G tile [64 m][128 n] + halo tile [<= XI * 64 pixels][64 c]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int W = 1 << log2W, WP = W + 2, rows_ps = 64 >> log2W;         // image rows per 64-pixel stage
    constexpr int npix = (rows_ps + 2) * WP, nslots = (npix + 7) >> 3;
    const int ctiles = g.Cs / 64, ntiles = g.N / 128;
    const int tiles = ntiles * ctiles;
    int zslice, tl;
    if (g.xcd_group) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        zslice = (idx / tiles) * 8 + xcd;
        tl = idx % tiles;
    } else {
        zslice = blockIdx.x / tiles;
        tl = blockIdx.x % tiles;
    }
    if (zslice >= g.splits) return;
    const int n0 = (tl % ntiles) * 128, c0 = (tl / ntiles) * 64;
    const long m_begin = (long)zslice * g.stages_per_split * 64;
    const long m_end = min(g.M, m_begin + (long)g.stages_per_split * 64);
    const int nst = (int)((m_end - m_begin + 63) / 64);
    if (nst <= 0) return;
    const int HoWo = g.Ho * g.Wo;
    const long zdX = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);
    const float rcp_wp = 1.0f / (float)WP;

    int st_next = 0;
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE;
        const long mb = m_begin + (long)st_next * 64;
        ++st_next;
        {
            const int rli = lane >> 4, lch = lane & 15;
#pragma unroll
            for (int i = 0; i < GI; ++i) {
                const int q = i * NW + wave, row = q * 4 + rli;
                const int gch = (((lch >> 1) ^ trkey(row)) << 1) | (lch & 1);
                GLDS16W(reinterpret_cast<const char*>(G) + ((mb + row) * g.N + n0 + gch * 8) * 2, sb + q * 1024);
            }
        }
        const int b = (int)(mb / HoWo), row0 = (int)(mb % HoWo) >> log2W;
        const int pli = lane >> 3, lch = lane & 7;                            // 8 pixels x 8 chunks per wave-instruction
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int q = i * NW + wave;
            if (q >= nslots) q -= nslots;                                     // spare slots repeat a valid one (same bytes, same place)
            if (q >= nslots) q -= nslots;
            const int pp = q * 8 + pli;
            int pr, pc;
            fast_divmod(pp, WP, rcp_wp, pr, pc);
            const int ho = row0 + pr - 1, wo = pc - 1;
            const int gch = (((lch >> 1) ^ trkey2(pp)) << 1) | (lch & 1);
            const bool ok = (pp < npix) & ((unsigned)ho < (unsigned)g.Hs) & ((unsigned)wo < (unsigned)W);
            const long off = ok ? ((((long)b * g.Hs + ho) * W + wo) * g.Cs + c0 + gch * 8) * 2 : zdX;
            GLDS16W(reinterpret_cast<const char*>(X) + off, sb + 64 * 256 + q * 1024);
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave >> 2, wc = wave & 3;

    // Fragment addresses.  The loop issues 52 transposed reads per wave and stage beside 72 MFMAs: with the addresses computed per read
    // (~9 integer instructions each) the loop was bound by vector-instruction ISSUE (340 VALU x 4 cycles per wave against 72 x 16 of MFMA:
    // tools/micro/tr_mfma_loop.hip, 2.17 us per stage against 1.08 us of MFMAs alone).  Everything that does not depend on the lane is an
    // immediate offset of the DS instruction now (W is a template parameter): a G fragment = one of four per-lane bases + {0, 1 KB} +
    // k-half x 8 KB (trkey is the same for rows r, r + 4, r + 32); an X fragment = per-lane base of the k-half + (r WP + s) x 128 (+ 512),
    // plus the 32-byte block of the swizzle, which depends on bits 1 and 3 of the lane's halo pixel: nine 2-bit entries per k-half and
    // row group packed in one register, one v_bfe + one v_lshl_add per read.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
    const int p8 = (lane & 3) * 8;
    unsigned abase[4];
    {
        const int r0 = 8 * (lane >> 4) + ((lane >> 2) & 3);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) abase[mi] = lds0 + r0 * 256 + (((wn * 4 + mi) ^ trkey(r0)) * 32) + p8;
    }
    unsigned bbase[2], swz[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = kk * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
        const int hb = (k >> log2W) * WP + (k & (W - 1));
        bbase[kk] = lds0 + 64 * 256 + hb * 128 + p8;
        swz[kk][0] = swz[kk][1] = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int h0 = hb + (t / 3) * WP + (t % 3);
            swz[kk][0] |= (unsigned)(wc ^ trkey2(h0)) << (2 * t);
            swz[kk][1] |= (unsigned)(wc ^ trkey2(h0 + 4)) << (2 * t);                // k % 8 < 4: k + 4 stays in the image row
        }
    }

    // 18 units per stage (2 k-halves x 9 taps, 4 MFMAs each), software-pipelined ACROSS stages.  With the stage's barrier in front of its
    // first fragment reads, all eight waves sat out the same LDS round trip in lockstep with the matrix pipe idle (2.2 us per stage in
    // the kernel against 1.4 us for the same loop without a barrier, tools/micro/tr_mfma_loop.hip).  Here the barrier that publishes
    // stage t + 1 (and frees the buffer of stage t - 1 for the DMA of stage t + NSTAGE - 1) sits between the two k-halves of stage t, and
    // the units after it fetch across the boundary: the X fragments run LOOK units ahead in a ring of LOOK + 1 register pairs whatever the
    // stage (18 % (LOOK + 1) == 0: static registers), the G fragments of the second k-half arrive during units 1-4, those of the NEXT
    // stage's first k-half during units 10-13 (their registers are dead after unit 8).  Waits are counted (LDS operations return in
    // order): unit u waits until only the reads issued after its own fragment remain.  The last stage fetches a stale buffer instead of
    // a next stage (in range, never used): the counts stay static.
    constexpr int LOOK = 2;
    static_assert(18 % (LOOK + 1) == 0, "ring slots must be static across stages");
    s16x4 alo[2][4], ahi[2][4], blo[LOOK + 1], bhi[LOOK + 1];
    unsigned ab_cur[4], bb_cur[2], ab_nxt[4], bb_nxt, sw[2][2];
    auto set_bases = [&](int cur, int nxt) {
        const unsigned oc = (unsigned)(cur * STAGE), on = (unsigned)(nxt * STAGE);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { ab_cur[mi] = abase[mi] + oc; ab_nxt[mi] = abase[mi] + on; }
        bb_cur[0] = bbase[0] + oc;
        bb_cur[1] = bbase[1] + oc;
        bb_nxt = bbase[0] + on;
        sw[0][0] = swz[0][0]; sw[0][1] = swz[0][1]; sw[1][0] = swz[1][0]; sw[1][1] = swz[1][1];
        asm volatile("" : "+v"(sw[0][0]), "+v"(sw[0][1]), "+v"(sw[1][0]), "+v"(sw[1][1]));   // the 36 block offsets are extracted per read, not kept in registers
    };
    // X fragment of unit v (0 .. 17: this stage; 18 ..: unit v - 18 of the next stage)
    auto x_issue = [&](auto V_) {
        constexpr int v = decltype(V_)::value, u = v % 18, kk = u / 9, t = u % 9, slot = v % (LOOK + 1);
        constexpr int off = ((t / 3) * WP + (t % 3)) * 128;
        const unsigned base = v < 18 ? bb_cur[kk] : bb_nxt;
        tr_read_off<off>(blo[slot], base + (__builtin_amdgcn_ubfe(sw[kk][0], 2 * t, 2) << 5));
        tr_read_off<off + 512>(bhi[slot], base + (__builtin_amdgcn_ubfe(sw[kk][1], 2 * t, 2) << 5));
    };
    auto a_issue_cur1 = [&](auto MI_) {                                     // second k-half of this stage
        constexpr int mi = decltype(MI_)::value;
        tr_read_off<8192>(alo[1][mi], ab_cur[mi]);
        tr_read_off<8192 + 1024>(ahi[1][mi], ab_cur[mi]);
    };
    auto a_issue_nxt0 = [&](auto MI_, const unsigned (&ab)[4]) {            // first k-half of the next stage (or of stage 0 in the prologue)
        constexpr int mi = decltype(MI_)::value;
        tr_read_off<0>(alo[0][mi], ab[mi]);
        tr_read_off<1024>(ahi[0][mi], ab[mi]);
    };
    auto units = [&](auto FIRST_) {                                          // units FIRST .. FIRST + 8
        static_for([&](auto I_) {
            constexpr int u = decltype(FIRST_)::value + decltype(I_)::value, kk = u / 9, t = u % 9;
            // reads issued after the fragment of unit u (issued in unit u - LOOK in front of that unit's G piece): that piece, then the
            // fragment and the piece of every unit up to u - 1
            constexpr int after = [] {
                auto pc = [](int w) { w = ((w % 18) + 18) % 18; return ((w >= 1 && w <= 4) || (w >= 10 && w <= 13)) ? 2 : 0; };
                int a = pc(u - LOOK);
                for (int w = u - LOOK + 1; w < u; ++w) a += 2 + pc(w);
                return a;
            }();
            static_assert(after + 4 <= 15, "lgkmcnt is a 4-bit counter");
            __builtin_amdgcn_sched_barrier(0);                                // the previous unit's MFMAs stay in front of this wait
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(after) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            x_issue(std::integral_constant<int, u + LOOK>{});
            if constexpr (u >= 1 && u <= 4) a_issue_cur1(std::integral_constant<int, u - 1>{});
            if constexpr (u >= 10 && u <= 13) a_issue_nxt0(std::integral_constant<int, u - 10>{}, ab_nxt);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 bf = tr_pack(blo[u % (LOOK + 1)], bhi[u % (LOOK + 1)]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pack(alo[kk][mi], ahi[kk][mi]), bf, acc[t][mi], 0, 0, 0);
        }, std::make_integer_sequence<int, 9>{});
    };

#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nst) stage(s0);
    {   // stage 0 has landed: its first fragments go out in front of the loop (the one exposed LDS round trip of the workgroup)
        const int younger = min(nst, NSTAGE - 1) - 1;
        if (NSTAGE >= 4 && younger >= 2) wg_wait_vmcnt_le<(NSTAGE >= 4 ? 2 * L : 0)>();
        else if (younger >= 1) wg_wait_vmcnt_le<L>();
        else wg_wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        set_bases(0, 0);
        static_for([&](auto MI_) { a_issue_nxt0(MI_, ab_cur); }, std::make_integer_sequence<int, 4>{});
        static_for([&](auto V_) { x_issue(V_); }, std::make_integer_sequence<int, LOOK>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    int cur = 0;
    for (int t = 0; t < nst; ++t) {
        const int nxt = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        set_bases(cur, nxt);
        if (g.pf_dist != 1) units(std::integral_constant<int, 0>{});
        // stage t + 1 visible to every wave; every wave is past stage t - 1: its buffer takes the DMA of stage t + NSTAGE - 1
        const int younger = min(nst - 1, t + NSTAGE - 2) - (t + 1);
        if (younger >= 1) wg_wait_vmcnt_le<(NSTAGE >= 4 ? (NSTAGE - 3) * L : 0)>();
        else wg_wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nst && g.pf_dist != 2) stage((cur + NSTAGE - 1) % NSTAGE);
        if (g.pf_dist != 1) units(std::integral_constant<int, 9>{});
        cur = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the stale look-ahead reads of the last stage

    float* dst = dW + (long)zslice * g.slab_elems;
    if (g.native_slabs) {                                      // accumulator-order slab: 36 records per lane (wgrad_reduce_native_body MODE 2)
        f32x4* d4 = reinterpret_cast<f32x4*>(dst) + ((long)(tl * 8 + wave) * 36) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) d4[(t * 4 + mi) * 64] = acc[t][mi];
        return;
    }
    const int fr = lane & 15, fq = lane >> 4;
    const long wrow = 9L * g.Cs;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + mi * 16 + fq * 4 + j;
                const int c = c0 + wc * 16 + fr;
                dst[(long)n * wrow + (long)t * g.Cs + c] = acc[t][mi][j];
            }
}

// sum of nslab slabs [N][R][S][C] f32 -> torch [N][C][R][S] f32
__global__ __launch_bounds__(256) void wgrad_to_torch_kernel(const float* __restrict__ dW, float* __restrict__ out, int N, int C,
                                                             int R, int S, int nslab) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;         // index in the [N][R][S][C] (source) order: coalesced reads
    const long tot = (long)N * C * R * S;
    if (i >= tot) return;
    const int c = (int)(i % C), s = (int)((i / C) % S), r = (int)((i / ((long)C * S)) % R), n = (int)(i / ((long)C * S * R));
    float a = 0.f;
    for (int k = 0; k < nslab; ++k) a += dW[(long)k * tot + i];
    out[(((long)n * C + c) * R + r) * S + s] = a;
}

// sum of nslab accumulator-order slabs (native_slabs) -> torch [N][C][R][S] f32.  One thread per 16-byte record (a lane's four rows
// of one 16 x 16 MFMA tile): nslab coalesced 16-byte loads, eight in flight, then four 4-byte stores (16 lanes = 64 contiguous bytes
// of one output row for 1x1 weights).  MODE 0: conv_wgrad_pipe_kernel<TN> tiles (NW = TN / 32 waves, 16 records per wave and lane, one
// tap per tile); MODE 1: conv_wgrad3x3_kernel tiles (8 waves, 24 records: three taps of kernel row r); MODE 2: the nine-tap tiles.
template <int MODE>
__device__ __forceinline__ void wgrad_reduce_native_body(const float* __restrict__ slabs, float* __restrict__ out, int N, int C,
                                                            int R, int S, int nslab, int TN, long rec) {
    const long tot4 = (long)N * C * R * S / 4;
    if (rec >= tot4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + rec;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long)(k + u) * tot4];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; k < nslab; ++k) a += p[(long)k * tot4];
    const int lane = (int)(rec & 63), fr = lane & 15, fq = lane >> 4;
    int n, c, tap;
    if (MODE == 0) {
        const int NW = TN / 32;
        const int t = (int)((rec >> 6) & 15), wave = (int)((rec >> 10) % NW), tl = (int)((rec >> 10) / NW);
        const int mi = t >> 2, ni = t & 3, wm = wave >> 1, wn = wave & 1;
        const int ntn = N / TN, ctiles = C / 128;
        const int n0 = (tl % ntn) * TN, by = tl / ntn;
        tap = by / ctiles;
        n = n0 + wm * 64 + mi * 16 + fq * 4;
        c = (by % ctiles) * 128 + wn * 64 + ni * 16 + fr;
    } else if (MODE == 2) {                                    // conv_wgrad3x3_t9_kernel: 8 waves x 36 records (nine taps x four n-tiles)
        const long q = rec >> 6;
        const int t = (int)(q % 36), wave = (int)((q / 36) & 7), tl = (int)(q / 288);
        const int mi = t & 3, wn = wave >> 2, wc = wave & 3;
        const int ntiles = N / 128;
        tap = t >> 2;
        n = (tl % ntiles) * 128 + wn * 64 + mi * 16 + fq * 4;
        c = (tl / ntiles) * 64 + wc * 16 + fr;
    } else {
        const long q = rec >> 6;
        const int t = (int)(q % 24), wave = (int)((q / 24) & 7), tl = (int)(q / 192);
        const int s3 = t >> 3, mi = (t >> 1) & 3, ni = t & 1, wn = wave >> 2, wc = wave & 3;
        const int ntiles = N / 128, ctiles = C / 128;
        const int n0 = (tl % ntiles) * 128, by = tl / ntiles;
        tap = (by / ctiles) * 3 + s3;
        n = n0 + wn * 64 + mi * 16 + fq * 4;
        c = (by % ctiles) * 128 + wc * 32 + ni * 16 + fr;
    }
    const long RS = (long)R * S;
    float* o = out + ((long)n * C + c) * RS + tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(long)j * C * RS] = a[j];
}
template <int MODE>
__global__ __launch_bounds__(256) void wgrad_reduce_native_kernel(const float* __restrict__ slabs, float* __restrict__ out, int N, int C,
                                                                  int R, int S, int nslab, int TN) {
    wgrad_reduce_native_body<MODE>(slabs, out, N, C, R, S, nslab, TN, (long)blockIdx.x * 256 + threadIdx.x);
}
// up to four pending reduces (ppv_conv_wgrad_ex with `deferred`) in one launch: workgroup ranges [blk0, blk0 + blocks) per problem
struct ReduceMulti { PpvWgradReduce p[4]; int n; };
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(ReduceMulti m) {
    int i = 0;
    unsigned b = blockIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (i + 1 < m.n && b >= (unsigned)m.p[i].blocks) { b -= (unsigned)m.p[i].blocks; ++i; }
    const PpvWgradReduce& q = m.p[i];
    const long rec = (long)b * 256 + threadIdx.x;
    if (q.mode == 0) wgrad_reduce_native_body<0>((const float*)q.slabs, q.out, q.N, q.C, q.R, q.S, q.nslab, q.TN, rec);
    else if (q.mode == 2) wgrad_reduce_native_body<2>((const float*)q.slabs, q.out, q.N, q.C, q.R, q.S, q.nslab, q.TN, rec);
    else wgrad_reduce_native_body<1>((const float*)q.slabs, q.out, q.N, q.C, q.R, q.S, q.nslab, q.TN, rec);
}

// ============================================================================= stem forward
// torch [64][CIN][7][7] f32 -> [64][NCHP chunks][8] bf16, chunk = r*CIN + c (7*CIN used), element s (7 used)
template <int CIN>
__global__ __launch_bounds__(256) void stem_weight_layout_kernel(const float* __restrict__ w, bf16_t* __restrict__ o) {
    constexpr int NCH = 7 * CIN, NCHP = (NCH + 7) / 8 * 8;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * NCHP * 8) return;
    const int s = i & 7, ch = (i >> 3) % NCHP, n = i / (NCHP * 8);
    float v = 0.f;
    if (ch < NCH && s < 7) {
        const int r = ch / CIN, c = ch % CIN;
        v = w[((n * CIN + c) * 7 + r) * 7 + s];
    }
    o[i] = f2bfw(v);
}

template <int CIN>
__global__ __launch_bounds__(256, CIN == 3 ? 2 : 1) void stem_conv_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wst,
                                                                         bf16_t* __restrict__ out, float* __restrict__ stat_part,
                                                                         int B, int H, int W, int tiles, int stat_rows) {
    constexpr int NCH = 7 * CIN, NCHP = (NCH + 7) / 8 * 8, KS = NCHP / 4;
    constexpr int LDA = NCHP * 16;                              // bytes per A / W row
    extern __shared__ __attribute__((aligned(16))) char stem_smem[];
    char* sA = stem_smem;
    char* sW = stem_smem + 128 * LDA;
    float (*sStat)[2][64] = reinterpret_cast<float (*)[2][64]>(stem_smem + 192 * LDA);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Ho = H / 2, Wo = W / 2;
    const long M = (long)B * Ho * Wo;
    for (int idx = tid; idx < 64 * NCHP; idx += 256) {
        const int n = idx / NCHP, ch = idx % NCHP;
        *reinterpret_cast<uint4*>(sW + n * LDA + ((ch ^ (n & 7)) * 16)) = *reinterpret_cast<const uint4*>(wst + (n * NCHP + ch) * 8);
    }
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long m0 = (long)tile * 128;
        __syncthreads();
        {
            const int row = tid & 127;
            const long m = m0 + row;
            const bool okm = m < M;
            const long mm = okm ? m : 0;
            const int b = (int)(mm / ((long)Ho * Wo)), rem = (int)(mm % ((long)Ho * Wo));
            const int ho = rem / Wo, wo = rem % Wo;
            for (int ch = (tid >> 7); ch < NCHP; ch += 2) {
                unsigned wv[4] = {0, 0, 0, 0};
                if (okm && ch < NCH) {
                    const int r = ch / CIN, c = ch % CIN;
                    const int hi = 2 * ho - 3 + r;
                    if (hi >= 0 && hi < H) {
                        const float* src = img + (((long)b * CIN + c) * H + hi) * W;
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
        for (int ks = 0; ks < KS; ++ks) {
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

// Row-staged form of the 3-channel stem (W/2 % 128 == 0, the 256^2 workload): a tile is 128 consecutive output pixels of ONE
// output row, so its 7 x 3 input rows are staged once with coalesced float4 loads (f32 -> bf16 LDS rows, 66 x 21 loads per tile
// instead of 128 x 147 scalar gathers) and the A fragments are read straight from the rows: k-chunk (r, c) of output pixel wo is
// the 8 consecutive columns 2 wo - 4 .. 2 wo + 3 (a 4-byte aligned 16-byte window; the weights' taps move up one slot, slot 0 = 0).
// Row pitch 576 B = 144 dwords = 16 mod 64 banks: the four k-groups of a fragment read hit disjoint banks.
__global__ __launch_bounds__(256, 2) void stem_conv_rows_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wst,
                                                               bf16_t* __restrict__ out, float* __restrict__ stat_part,
                                                               int B, int H, int W, int tiles, int stat_rows) {
    constexpr int NCH = 21, NCHP = 24, KS = 6, LDA = NCHP * 16, ROWB = 576, LDO = 144;
    extern __shared__ __attribute__((aligned(16))) char stem_smem[];
    char* sIn = stem_smem;                                      // 21 rows x 576 B
    char* sW = stem_smem + 12288;                               // 64 x 384 B
    char* sO = stem_smem + 12288 + 64 * LDA;                    // 128 x 144 B
    float (*sStat)[2][64] = reinterpret_cast<float (*)[2][64]>(stem_smem + 12288 + 64 * LDA + 128 * LDO);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Ho = H / 2, Wo = W / 2, tpr = Wo / 128;
    for (int idx = tid; idx < 64 * NCHP; idx += 256) {
        const int n = idx / NCHP, ch = idx % NCHP;
        const uint4 u = *reinterpret_cast<const uint4*>(wst + (n * NCHP + ch) * 8);
        const uint4 v = make_uint4(u.x << 16, (u.x >> 16) | (u.y << 16), (u.y >> 16) | (u.z << 16), (u.z >> 16) | (u.w << 16));
        *reinterpret_cast<uint4*>(sW + n * LDA + ((ch ^ (n & 7)) * 16)) = v;
    }
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    // the next tile's input rows are requested before this tile's MFMA phase and land in registers behind it
    constexpr int NLD = (NCH * 66 + 255) / 256;
    float4 pf[NLD];
    auto fetch = [&](int tile) {
        const int wseg = tile % tpr, bh = tile / tpr;
        const int b = bh / Ho, ho = bh % Ho;
        const int wi0 = wseg * 256 - 4;                         // input column of staged element 0
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = k * 256 + tid;
            const int rc = idx / 66, j = idx % 66;
            const int r = rc / 3, c = rc % 3;
            const int hi = 2 * ho - 3 + r, wi = wi0 + 4 * j;
            pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NCH * 66 && hi >= 0 && hi < H && wi >= 0 && wi < W)
                pf[k] = *reinterpret_cast<const float4*>(img + (((long)b * 3 + c) * H + hi) * W + wi);
        }
    };
    // XCD-aware tile order: workgroup i runs on XCD i & 7; each XCD walks its own contiguous eighth of the tiles, so the
    // output rows that share input rows are neighbours in ONE L2 instead of being fetched by eight
    const int nx = gridDim.x >> 3, per = (tiles + 7) >> 3;
    const int tile0 = (blockIdx.x & 7) * per + (blockIdx.x >> 3), tile_end = min(tiles, ((int)(blockIdx.x & 7) + 1) * per);
    if (tile0 < tile_end) fetch(tile0);
    for (int tile = tile0; tile < tile_end; tile += nx) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = k * 256 + tid;
            const int rc = idx / 66, j = idx % 66;
            const unsigned lo = (unsigned)f2bfw(pf[k].x) | ((unsigned)f2bfw(pf[k].y) << 16);
            const unsigned hi2 = (unsigned)f2bfw(pf[k].z) | ((unsigned)f2bfw(pf[k].w) << 16);
            if (idx < NCH * 66) *reinterpret_cast<uint2*>(sIn + rc * ROWB + j * 8) = make_uint2(lo, hi2);
        }
        __syncthreads();
        if (tile + nx < tile_end) fetch(tile + nx);
        f32x4 acc[4][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi][0] = acc[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[4], bfr[2];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                uint4 a = make_uint4(0, 0, 0, 0);
                if (chunk < NCH) {
                    const unsigned* pa = reinterpret_cast<const unsigned*>(sIn + chunk * ROWB + (wm * 64 + mi * 16 + fr) * 4);
                    a = make_uint4(pa[0], pa[1], pa[2], pa[3]);
                }
                af[mi] = __builtin_bit_cast(bf16x8, a);
            }
            const int wchunk = (chunk ^ (fr & 7)) * 16;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(sW + (wn * 32 + ni * 16 + fr) * LDA + wchunk);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
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
        __syncthreads();                                        // sO / sStat complete; every wave is done reading sIn
        if (stat_part && tid < 128) {
            const int which = tid >> 6, col = tid & 63;
            atomicAdd(&stat_part[((long)(tile % stat_rows) * 2 + which) * 64 + col], sStat[0][which][col] + sStat[1][which][col]);
        }
        const long m0 = (long)tile * 128;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx >> 3, ch = idx & 7;
            *reinterpret_cast<uint4*>(out + (m0 + row) * 64 + ch * 8) = *reinterpret_cast<const uint4*>(sO + row * LDO + ch * 16);
        }
        // no barrier here: the next tile's sIn writes only race with reads that finished before the barrier above, and its
        // sO / sStat writes come after its own staging barrier, which every thread reaches after this store loop
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

// Row-staged stem data gradient (Wo % 128 == 0): d/d(img) of the 7x7/2 stem as the same 4x4-tap GEMM on 2x2 super-pixels
// (M = B*Ho*Wo, N = 16 of which 12 used, K = 16 taps x 64 ch), but a tile is 128 super-pixels of ONE row: the four gradient rows
// y-1..y+2 it touches are staged once by LDS-DMA (4 x 136 pixels x 128 B, double-buffered: the next tile lands while this one
// multiplies; 16-byte chunks swizzled c ^ (p & 7) on the SOURCE address so that the four 16-lane groups of a ds_read_b128 each
// hit 16 disjoint bank slots) instead of 16 gathered taps per pixel (3.8x less L2 -> LDS traffic), the weights' 32 B fragments
// stay in registers for the whole launch, and the result is written straight in NCHW f32 (six contiguous 1-KB row segments per
// tile) -- no [M][16] temporary, no scatter launch.  One 8-wave workgroup per CU (2 x 68 KB of staging).
__global__ __launch_bounds__(512, 1) void stem_dgrad_rows_kernel(const bf16_t* __restrict__ gx, const bf16_t* __restrict__ wsd,
                                                                float* __restrict__ gimg, const bf16_t* __restrict__ zero_page,
                                                                int B, int Ho, int Wo, int tiles) {
    constexpr int NP = 136, BUF = 4 * NP * 128, NGRP = 4 * NP / 8, NDMA = (NGRP + 7) / 8;   // 68 one-KiB wave-instructions per tile
    extern __shared__ __attribute__((aligned(16))) char stem_smem[];
    float* sOut = reinterpret_cast<float*>(stem_smem + 2 * BUF);           // [6][256] f32
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int tpr = Wo / 128, H = 2 * Ho, W = 2 * Wo;
    bf16x8 wreg[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(wsd + fr * 1024 + ks * 32 + fq * 8);
    // lane -> (pixel within the 8-pixel group, destination chunk slot); the source chunk is slot ^ (pixel & 7)
    const int lp = lane >> 3, lc = lane & 7;
    auto stage = [&](int tile, int buf) {
        const int xseg = tile % tpr, by = tile / tpr;
        const int b = by / Ho, y = by % Ho;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            const int grp = k * 8 + wave;                       // 8-pixel group 0..67 = (dy, 17 groups per row)
            if (grp >= NGRP) break;                             // wave-uniform
            const int dy = grp / 17, pix = (grp % 17) * 8 + lp;
            const int yy = y - 1 + dy, xx = xseg * 128 - 1 + pix;
            const bf16_t* src = zero_page;
            if (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo) src = gx + (((long)b * Ho + yy) * Wo + xx) * 64 + (lc ^ (pix & 7)) * 8;
            GLDS16W(src, stem_smem + buf * BUF + grp * 1024);
        }
    };
    const int nx = gridDim.x >> 3, per = (tiles + 7) >> 3;      // XCD-aware tile order, as stem_conv_rows_kernel
    const int tile0 = (blockIdx.x & 7) * per + (blockIdx.x >> 3), tile_end = min(tiles, ((int)(blockIdx.x & 7) + 1) * per);
    if (tile0 < tile_end) stage(tile0, 0);
    int cur = 0;
    for (int tile = tile0; tile < tile_end; tile += nx, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // this tile has landed; the other buffer and sOut are free
        if (tile + nx < tile_end) stage(tile + nx, cur ^ 1);
        const char* sA = stem_smem + cur * BUF;
        // eight waves, 16 super-pixels each (two waves per SIMD cover each other's LDS latency); fragment reads in batches of 8
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 af[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ks = kb * 8 + q;
                const int tap = ks >> 1, dy = tap >> 2, dx = tap & 3, c8 = (ks & 1) * 4 + fq;
                const int pix = wave * 16 + fr + dx;
                af[q] = *reinterpret_cast<const bf16x8*>(sA + ((dy * NP + pix) * 8 + (c8 ^ (pix & 7))) * 16);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q], wreg[kb * 8 + q], acc, 0, 0, 0);
            // pin the order (the machine scheduler otherwise sinks every read to just before its MFMA: read, wait, multiply)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        // column n = (ph*2 + pw)*3 + c of super-pixel m -> image row segment (c, ph), column 2 m + pw
        if (fr < 12) {
            const int c = fr % 3, pw = (fr / 3) & 1, ph = fr / 6;
#pragma unroll
            for (int j = 0; j < 4; ++j) sOut[(c * 2 + ph) * 256 + 2 * (wave * 16 + fq * 4 + j) + pw] = acc[j];
        }
        __syncthreads();
        const int xseg = tile % tpr, by = tile / tpr;
        const int b = by / Ho, y = by % Ho;
        for (int idx = tid; idx < 6 * 64; idx += 512) {
            const int seg = idx >> 6, q = idx & 63;
            const int c = seg >> 1, ph = seg & 1;
            *reinterpret_cast<float4*>(gimg + (((long)b * 3 + c) * H + 2 * y + ph) * W + xseg * 256 + q * 4) =
                *reinterpret_cast<const float4*>(sOut + seg * 256 + q * 4);
        }
    }
}

// Strip form of the above (round 5).  stem_dgrad_rows_kernel stages the FOUR gradient rows of a tile (70 KB) for 6 KB of output and
// its tiles are interleaved across XCDs, so every gradient row travels L2 -> LDS four times: the kernel's loads alone take 108 us, its
// MFMAs + stores alone 106 us, the whole 185 us (cold, PPV phase experiment).  Here a workgroup owns a STRIP of consecutive output rows
// of one image segment and keeps a ring of SIX row slots (17 KB each): a tile stages only the one row that is new (y + 3, under this
// tile's multiply), reads rows y - 1 .. y + 2 from their slots, and does not wait for its own stores (the wait at the head of the next
// tile is counted: the DMA is older than the stores).
__global__ __launch_bounds__(512, 1) void stem_dgrad_strip_kernel(const bf16_t* __restrict__ gx, const bf16_t* __restrict__ wsd,
                                                                 float* __restrict__ gimg, const bf16_t* __restrict__ zero_page,
                                                                 int B, int Ho, int Wo, int strip, int strips_per_col) {
    constexpr int NP = 136, ROWB = NP * 128, NSLOT = 6, NGRP = NP / 8;    // 17 one-KiB wave-instructions per gradient row
    extern __shared__ __attribute__((aligned(16))) char stem_smem[];
    float* sOut = reinterpret_cast<float*>(stem_smem + NSLOT * ROWB);     // [6][256] f32
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int tpr = Wo / 128, H = 2 * Ho, W = 2 * Wo;
    bf16x8 wreg[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) wreg[ks] = *reinterpret_cast<const bf16x8*>(wsd + fr * 1024 + ks * 32 + fq * 8);
    // workgroup -> (image, column segment, strip of rows)
    const int sidx = blockIdx.x % strips_per_col, col = blockIdx.x / strips_per_col;
    const int xseg = col % tpr, b = col / tpr;
    const int y0 = sidx * strip, y1 = min(Ho, y0 + strip);
    if (b >= B || y0 >= y1) return;
    const int lp = lane >> 3, lc = lane & 7;
    auto stage_row = [&](int yy) {                              // gradient row yy -> slot yy mod 6 (rows outside the image: zeros)
        const int slot = ((yy % NSLOT) + NSLOT) % NSLOT;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int grp = k * 8 + wave;
            if (grp >= NGRP) break;                             // wave-uniform
            const int pix = grp * 8 + lp;
            const int xx = xseg * 128 - 1 + pix;
            const bf16_t* src = zero_page;
            if (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo) src = gx + (((long)b * Ho + yy) * Wo + xx) * 64 + (lc ^ (pix & 7)) * 8;
            GLDS16W(src, stem_smem + slot * ROWB + grp * 1024);
        }
    };
    for (int yy = y0 - 1; yy <= y0 + 2; ++yy) stage_row(yy);
    for (int y = y0; y < y1; ++y) {
        // rows y - 1 .. y + 2 have landed (the last of them was requested before the previous tile's stores: those may still be in flight)
        if (y == y0 || tid >= 384) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __syncthreads();                                        // ... for every wave; the slot of row y - 2 and sOut are free
        if (y + 1 < y1) stage_row(y + 3);
        const char* rowp[4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) rowp[dy] = stem_smem + ((((y - 1 + dy) % NSLOT) + NSLOT) % NSLOT) * ROWB;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 af[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ks = kb * 8 + q;
                const int tap = ks >> 1, dy = tap >> 2, dx = tap & 3, c8 = (ks & 1) * 4 + fq;
                const int pix = wave * 16 + fr + dx;
                af[q] = *reinterpret_cast<const bf16x8*>(rowp[dy] + (pix * 8 + (c8 ^ (pix & 7))) * 16);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[q], wreg[kb * 8 + q], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        if (fr < 12) {
            const int c = fr % 3, pw = (fr / 3) & 1, ph = fr / 6;
#pragma unroll
            for (int j = 0; j < 4; ++j) sOut[(c * 2 + ph) * 256 + 2 * (wave * 16 + fq * 4 + j) + pw] = acc[j];
        }
        __syncthreads();
        if (tid < 6 * 64) {
            const int seg = tid >> 6, q = tid & 63;
            const int c = seg >> 1, ph = seg & 1;
            *reinterpret_cast<float4*>(gimg + (((long)b * 3 + c) * H + 2 * y + ph) * W + xseg * 256 + q * 4) =
                *reinterpret_cast<const float4*>(sOut + seg * 256 + q * 4);
        }
    }
}

}  // namespace ppv

using namespace ppv;

static int g_wgrad_variant = 0;

extern "C" {

// workgroups a weight-gradient launch aims for (split-M slices x tiles); PPV_WGRAD_WGS overrides (A/B: fewer slices = fewer
// slab bytes and more CUs left to the data-gradient chain the launches overlap with, but longer launches)
static int wgrad_target_wgs(long M = 0) {
    // round 3 (tools/sweep_env.sh, whole step, same box, two passes): 256 -> 5255, 320 -> 5282, 384 -> 5329, 448 -> 5360, 512 -> 5400,
    // 576 -> 5253, 640 -> 5280, 768 -> 5252 images/s (512 = two 64-KB workgroups per CU); round 2's optimum was 384.
    // PPV_WGRAD_WGS_BIGM: the target for launches with >= 64 Ki rows (layer 2), default the same.
    static const int t = getenv("PPV_WGRAD_WGS") ? atoi(getenv("PPV_WGRAD_WGS")) : 512;
    static const int tb = getenv("PPV_WGRAD_WGS_BIGM") ? atoi(getenv("PPV_WGRAD_WGS_BIGM")) : t;
    const int v = M >= 65536 ? tb : t;
    return v < 16 ? 16 : v;
}

// stages the cooperative L2 prefetch of the 1x1 weight gradients runs ahead (PPV_WGRAD_PF; 0 = off)
static int wgrad_pf_dist() {
    static const int d = getenv("PPV_WGRAD_PF") ? atoi(getenv("PPV_WGRAD_PF")) : 0;   // measured: no gain (see wgrad_pipe_body)
    return d;
}

// the fused-tap 3x3 kernel's own target (PPV_WGRAD3_WGS): its 120-KB workgroups cannot share a CU with a convolution workgroup, and
// every m-slice writes a 9-tap slab -- fewer, longer slices than the 1x1 kernels want
static int wgrad3_target_wgs() {
    // whole step, same box: 384 -> 5177, 240 -> 5315, 192 -> 5330, 144 -> 5397, 120 -> 5395, 96 -> 5346, 72 -> 5329 images/s
    // (alone: 192 -> 77 us, 384 -> 89 us on the layer-3 shape: 16 instead of 32 nine-tap slabs)
    static const int t = getenv("PPV_WGRAD3_WGS") ? atoi(getenv("PPV_WGRAD3_WGS")) : 144;
    return t < 12 ? 12 : t;
}
// slabs in accumulator order + wgrad_reduce_native_kernel (PPV_WGRAD_NATIVE=0: the [N][R][S][C] slabs of rounds 1-2)
static int wgrad_native_slabs() {
    static const int t = getenv("PPV_WGRAD_NATIVE") ? atoi(getenv("PPV_WGRAD_NATIVE")) : 1;
    return t ? 1 : 0;
}
static int wgrad3_stages() {
    static const int t = getenv("PPV_WGRAD3_NS") ? atoi(getenv("PPV_WGRAD3_NS")) : 3;
    return t == 2 ? 2 : 3;
}

static int wgrad_target_wgs256() {
    static const int t = getenv("PPV_WGRAD_WGS256") ? atoi(getenv("PPV_WGRAD_WGS256")) : 256;
    return t < 16 ? 16 : t;
}
static void wgrad_plan(long M, int N, int R, int S, int Cs, int variant, int* TN, long* splits, int* sps, int target = 0) {
    const long stages = (M + 63) / 64;
    int tn = (N % 256 == 0 && variant != 2) ? 256 : 128;
    if (variant == 1) tn = 128;
    const int tiles = (N / tn) * (R * S * (Cs / 128));
    long sp = ((target ? target : variant == 1 ? 512 : wgrad_target_wgs(M)) + tiles - 1) / tiles;
    if (sp > stages / 8) sp = stages / 8;
    if (sp < 1) sp = 1;
    *sps = (int)((stages + sp - 1) / sp);
    *splits = (stages + *sps - 1) / *sps;
    *TN = tn;
}

// m-slices of the fused-tap 3x3 kernel: ~one workgroup per CU over (N/128) x (Cs/128) x 3 tiles
static void wgrad3_plan(long M, int N, int Cs, long* splits, int* sps) {
    const long stages = M / 64;
    const int tiles = (N / 128) * (Cs / 128) * 3;
    long sp = (wgrad3_target_wgs() + tiles - 1) / tiles;
    if (sp > stages / 8) sp = stages / 8;
    if (sp < 1) sp = 1;
    *sps = (int)((stages + sp - 1) / sp);
    *splits = (stages + *sps - 1) / *sps;
}

// the nine-tap kernel (conv_wgrad3x3_t9_kernel): (N/128) x (Cs/64) tiles; PPV_WGRAD3_T9=0 keeps the three-tap kernel, PPV_WGRAD3_T9_WGS its
// workgroup target
static int wgrad3_t9() {
    static const int t = getenv("PPV_WGRAD3_T9") ? atoi(getenv("PPV_WGRAD3_T9")) : 1;
    return t;
}
static void wgrad3_t9_plan(long M, int N, int Cs, long* splits, int* sps) {
    static const int target = getenv("PPV_WGRAD3_T9_WGS") ? atoi(getenv("PPV_WGRAD3_T9_WGS")) : 144;
    const long stages = M / 64;
    const int tiles = (N / 128) * (Cs / 64);
    long sp = ((target < 8 ? 8 : target) + tiles - 1) / tiles;
    if (sp > stages / 8) sp = stages / 8;
    if (sp < 1) sp = 1;
    *sps = (int)((stages + sp - 1) / sp);
    *splits = (stages + *sps - 1) / *sps;
}

// m-slices of the streamed kernel (32-row stages)
static void wgrad_stream_plan(long M, int N, int R, int S, int Cs, int* TN, long* splits, int* sps) {
    const long stages = (M + 31) / 32;
    const int tn = (N % 256 == 0) ? 256 : 128;
    const int tiles = (N / tn) * (R * S * (Cs / 128));
    long sp = (wgrad_target_wgs() + tiles - 1) / tiles;
    if (sp > stages / 16) sp = stages / 16;
    if (sp < 1) sp = 1;
    *sps = (int)((stages + sp - 1) / sp);
    *splits = (stages + *sps - 1) / *sps;
    *TN = tn;
}

// bytes of f32 scratch ppv_conv_wgrad needs (per-slice slabs of the [N][R][S][Cs] gradient)
size_t ppv_conv_wgrad_scratch_bytes(long M, int N, int R, int S, int Cs) {
    int TN, sps;
    long splits;
    // whichever kernel ppv_conv_wgrad picks (tuning hook included): the largest slice count of the candidate plans
    wgrad_plan(M, N, R, S, Cs, 2, &TN, &splits, &sps);
    for (int v = 1; v <= 3; v += 2) {
        if (v == 3 && N % 256) continue;
        long sp;
        wgrad_plan(M, N, R, S, Cs, v, &TN, &sp, &sps);
        if (sp > splits) splits = sp;
    }
    if (N % 256 == 0) {                                          // variants 8 / 9: the 256-wide tile at its own target
        long sp;
        wgrad_plan(M, N, R, S, Cs, 3, &TN, &sp, &sps, wgrad_target_wgs256());
        if (sp > splits) splits = sp;
    }
    if (R == 3 && S == 3 && M % 64 == 0) {                       // the fused-tap kernel may be chosen: cover its plan too
        long sp3;
        int sps3;
        wgrad3_plan(M, N, Cs, &sp3, &sps3);
        if (sp3 > splits) splits = sp3;
        wgrad3_t9_plan(M, N, Cs, &sp3, &sps3);
        if (sp3 > splits) splits = sp3;
    }
    {
        long sps_;
        int tn_, st_;
        wgrad_stream_plan(M, N, R, S, Cs, &tn_, &sps_, &st_);
        if (sps_ > splits) splits = sps_;
    }
    return (size_t)splits * N * R * S * Cs * sizeof(float);
}

// Weight gradient of a conv: G [B,Ho,Wo,N] bf16 (gradient of the conv output), X [B,Hs,Ws,Cs] bf16 (conv input) ->
// dW_out in torch layout [N][Cs][R][S] f32.  scratch: ppv_conv_wgrad_scratch_bytes (no zeroing needed).
// N % 128 == 0, Cs % 128 == 0.
int ppv_conv_wgrad(const void* G, const void* X, float* dW_out, void* scratch, const void* zero_page, int B, int Hs, int Ws,
                   int Cs, int Ho, int Wo, int N, int R, int S, int stride, int pad, hipStream_t stream) {
    return ppv_conv_wgrad_ex(G, X, dW_out, scratch, zero_page, B, Hs, Ws, Cs, Ho, Wo, N, R, S, stride, pad, stream, nullptr);
}

// ppv_conv_wgrad whose slab reduce can be left to the caller: with `deferred` non-null and a kernel form that writes accumulator-order
// slabs (the default forms), the reduce is NOT launched; *deferred describes it (blocks > 0) for ppv_wgrad_reduce_multi, and `scratch`
// must stay untouched until that has run.  Forms that finish by themselves (atomics, streamed, [N][R][S][C] slabs) set blocks = 0.
int ppv_conv_wgrad_ex(const void* G, const void* X, float* dW_out, void* scratch, const void* zero_page, int B, int Hs, int Ws,
                      int Cs, int Ho, int Wo, int N, int R, int S, int stride, int pad, hipStream_t stream, PpvWgradReduce* deferred) {
    if (deferred) deferred->blocks = 0;
    if (!G || !X || !dW_out || !scratch || !zero_page) return PPV_ERR_NULL;
    if (N % 128 || Cs % 128) return PPV_ERR_BAD_SIZE;
    // DIAGNOSTIC (results wrong): PPV_WGRAD_SKIP=1 launches nothing -- what the step costs without the weight-gradient class
    static const int skip_all = getenv("PPV_WGRAD_SKIP") ? atoi(getenv("PPV_WGRAD_SKIP")) : 0;
    if (skip_all == 1 || (skip_all == 2 && R * S == 9) || (skip_all == 3 && R * S == 1)) return PPV_OK;    // 2: the 3x3 ones only, 3: the 1x1 ones only
    WgradGeom g;
    g.B = B; g.Hs = Hs; g.Ws = Ws; g.Cs = Cs; g.Ho = Ho; g.Wo = Wo; g.N = N; g.R = R; g.S = S; g.st = stride; g.pad = pad;
    g.M = (long)B * Ho * Wo;
    g.xcc_slabs = 0;
    g.pf_dist = 0;
    g.chunked = 0;
    g.native_slabs = 0;
    if (g_wgrad_variant & 0x1000) {                            // layout A/B (tools/bench_layout_ab.py): chunked operands, 1x1 / unit stride only
        if (R * S != 1 || stride != 1) return PPV_ERR_BAD_SIZE;
        g.chunked = 1;
    }
    if (g.M >= (1L << 24)) return PPV_ERR_BAD_SIZE;            // fast_divmod range
    int variant = g_wgrad_variant & 0xff;
    g.xcd_group = (g_wgrad_variant & 0x100) ? 0 : 1;
    if (variant == 0) {
        // Alone on the device the 256-wide three-stage tile (144 KB of LDS) is the fastest 1x1 form (tools/bench_wgrad.py), but
        // the trunk runs its weight gradients on a side stream beside the data-gradient / BN chain: there a 128-wide TWO-stage
        // ring (64 KB) that can share a CU with a conv workgroup wins the whole step by 1.9 % (4396 -> 4480 images/s).
        // Round 3, with the split targets re-tuned per kernel: the 256-wide tile is back for the 1x1 shapes whose N allows it, aimed at
        // ONE workgroup per CU (PPV_WGRAD_WGS256 = 256; 25 % fewer staged lines per flop than 128 x 128): whole step +0.8 %
        // (5557 -> 5603 images/s; its two-stage 96-KB form +0.5 %; PPV_WGRAD_VARIANT=6 restores the small ring everywhere).
        variant = 8;
        g.xcd_group = (R * S == 1) ? 1 : 0;                    //   XCD grouping pays for 1x1 only
    }
    const long elems = (long)N * R * S * Cs;
    float* slabs = (float*)scratch;
    const bool w_ok = Wo == 8 || Wo == 16 || Wo == 32 || Wo == 64;
    if (R == 3 && S == 3 && stride == 1 && pad == 1 && Hs == Ho && Ws == Wo && w_ok && (Ho * Wo) % 64 == 0 &&
        ((g_wgrad_variant & 0xff) == 0 || (g_wgrad_variant & 0xff) == 4 || (g_wgrad_variant & 0xff) == 6 || (g_wgrad_variant & 0xff) == 8 ||
         (g_wgrad_variant & 0xff) == 9)) {
        long sp3;
        int sps3;
        if (wgrad3_t9() && Wo <= 32 && !(g_wgrad_variant & 0x2000)) {            // all nine taps per workgroup (0x2000: the three-tap kernel, A/B)
            wgrad3_t9_plan(g.M, N, Cs, &sp3, &sps3);
            g.stages_per_split = sps3;
            g.splits = (int)sp3;
            g.slab_elems = elems;
            g.native_slabs = wgrad_native_slabs();
            g.xcd_group = (g_wgrad_variant & 0x200) ? 1 : 0;
            static const int t9_debug = getenv("PPV_WGRAD3_T9_DEBUG") ? atoi(getenv("PPV_WGRAD3_T9_DEBUG")) : 0;   // timing experiments: 1 = loads only, 2 = compute only (wrong results)
            g.pf_dist = t9_debug;
            const long t9 = (long)(N / 128) * (Cs / 64);
            const unsigned grid9 = (unsigned)(g.xcd_group ? 8 * ((sp3 + 7) / 8) * t9 : sp3 * t9);
            constexpr int lds2 = 4 * (64 * 256 + 2 * 8192), lds3 = 3 * (64 * 256 + 3 * 8192);   // four 32-KB stages (W <= 16), three 40-KB stages (W = 32)
            static bool attr9 = false;
            if (!attr9) {
                PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad3x3_t9_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
                PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad3x3_t9_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
                PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad3x3_t9_kernel<3, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
                attr9 = true;
            }
            if (Wo == 32)
                conv_wgrad3x3_t9_kernel<3, 5><<<grid9, 512, lds3, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
            else if (Wo == 16)
                conv_wgrad3x3_t9_kernel<4, 4><<<grid9, 512, lds2, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
            else
                conv_wgrad3x3_t9_kernel<4, 3><<<grid9, 512, lds2, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
            if (g.native_slabs && deferred) {
                *deferred = PpvWgradReduce{slabs, dW_out, N, Cs, R, S, (int)sp3, 128, 2, (int)((elems / 4 + 255) / 256)};
                return ppv_last_error();
            }
            if (g.native_slabs) wgrad_reduce_native_kernel<2><<<(unsigned)((elems / 4 + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, (int)sp3, 128);
            else wgrad_to_torch_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, (int)sp3);
            return ppv_last_error();
        }
        wgrad3_plan(g.M, N, Cs, &sp3, &sps3);
        g.stages_per_split = sps3;
        g.splits = (int)sp3;
        g.slab_elems = elems;
        // 2 or 4 stages measured no different alone (the loop is compute-side bound); two stages = 80 KB, which can share a CU with a
        // 72-KB convolution workgroup of the main stream (PPV_WGRAD3_NS)
        static bool attr = false;
        if (!attr) {
            PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad3x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * W3_STAGE));
            PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad3x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W3_STAGE));
            attr = true;
        }
        const int log2W = Wo == 8 ? 3 : Wo == 16 ? 4 : Wo == 32 ? 5 : 6;
        g.native_slabs = wgrad_native_slabs();
        g.xcd_group = (g_wgrad_variant & 0x200) ? 1 : 0;         // XCD grouping measured slower here (111 vs 88 us, layer 3)
        const long t3 = (long)(N / 128) * (Cs / 128) * 3;
        const unsigned grid3 = (unsigned)(g.xcd_group ? 8 * ((sp3 + 7) / 8) * t3 : sp3 * t3);
        if (wgrad3_stages() == 2)
            conv_wgrad3x3_kernel<2><<<grid3, 512, 2 * W3_STAGE, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g, log2W);
        else
            conv_wgrad3x3_kernel<3><<<grid3, 512, 3 * W3_STAGE, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g, log2W);
        if (g.native_slabs && deferred) {
            *deferred = PpvWgradReduce{slabs, dW_out, N, Cs, R, S, (int)sp3, 128, 1, (int)((elems / 4 + 255) / 256)};
            return ppv_last_error();
        }
        if (g.native_slabs) wgrad_reduce_native_kernel<1><<<(unsigned)((elems / 4 + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, (int)sp3, 128);
        else wgrad_to_torch_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, (int)sp3);
        return ppv_last_error();
    }
    if ((g_wgrad_variant & 0xff) == 7) {   // streamed kernel: loader / consumer waves, deep ring (opt-in: see its header comment)
        int TN, sps;
        long splits;
        wgrad_stream_plan(g.M, N, R, S, Cs, &TN, &splits, &sps);
        g.stages_per_split = sps;
        g.splits = (int)splits;
        g.slab_elems = elems;
        g.xcd_group = (R * S == 1 && !(g_wgrad_variant & 0x100)) ? 1 : 0;
        const int tiles = (N / TN) * (R * S * (Cs / 128));
        const unsigned grid = g.xcd_group ? (unsigned)(8 * ((splits + 7) / 8) * tiles) : (unsigned)(splits * tiles);
        if (TN == 256) {
            constexpr int lds = 6 * 3 * 32 * 256 + 64;
            static bool attr = false;
            if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_stream_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
            conv_wgrad_stream_kernel<256><<<grid, 768, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
        } else {
            constexpr int lds = 8 * 2 * 32 * 256 + 64;
            static bool attr = false;
            if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_stream_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
            conv_wgrad_stream_kernel<128><<<grid, 512, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
        }
        wgrad_to_torch_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, (int)splits);
        return ppv_last_error();
    }
    if (variant == 4) variant = (N % 256 == 0) ? 3 : 1;
    // 8 / 9: 256-wide tile (three / two 48-KB stages) where N allows it, aimed at its own workgroup count (PPV_WGRAD_WGS256, default 256:
    // one per CU), the small ring elsewhere
    int wide_stages = 0;
    if (variant == 8 || variant == 9) {
        wide_stages = variant == 8 ? 3 : 2;
        variant = (N % 256 == 0 && R * S == 1) ? 3 : 6;
        if (variant == 6) wide_stages = 0;
    }
    const bool small_ring = variant == 6;                      // TN = 128, two stages (64 KB)
    if (small_ring) variant = 2;
    int TN, sps;
    long splits;
    wgrad_plan(g.M, N, R, S, Cs, variant, &TN, &splits, &sps, wide_stages ? wgrad_target_wgs256() : 0);
    g.stages_per_split = sps;
    g.splits = (int)splits;
    const int tiles = (N / TN) * (R * S * (Cs / 128));
    if (variant == 1) {                                        // two-stage kernel, atomics into one zeroed accumulator
        g.slab_elems = 0;
        (void)hipMemsetAsync(slabs, 0, elems * sizeof(float), stream);
        conv_wgrad_kernel<<<dim3(N / 128, R * S * (Cs / 128), (unsigned)splits), 256, 0, stream>>>(
            (const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
        wgrad_to_torch_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, 1);
        return ppv_last_error();
    }
    g.slab_elems = elems;
    int nslab = (int)splits;
    g.native_slabs = (wgrad_native_slabs() && !((g_wgrad_variant & 0x400) && splits > 8)) ? 1 : 0;
    if ((g_wgrad_variant & 0x400) && splits > 8) {             // XCC-local atomic slabs (see conv_wgrad_pipe_kernel): 8 pre-zeroed slabs
        g.xcc_slabs = 1;
        nslab = 8;
        (void)hipMemsetAsync(slabs, 0, 8 * elems * sizeof(float), stream);
    }
    const unsigned grid = g.xcd_group ? (unsigned)(8 * ((splits + 7) / 8) * tiles) : (unsigned)(splits * tiles);
    static const int wgrad_debug = getenv("PPV_WGRAD_DEBUG") ? atoi(getenv("PPV_WGRAD_DEBUG")) : 0;   // timing experiments: 1 = loads only, 2 = compute only (wrong results)
    if (wgrad_debug > 0) g.pf_dist = 100 + wgrad_debug;
    if (TN == 256 && wide_stages == 2) {
        constexpr int lds = 2 * 3 * 64 * 256;
        static bool attr = false;
        if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_pipe_kernel<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
        conv_wgrad_pipe_kernel<256, 2><<<grid, 512, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
    } else if (TN == 256) {
        constexpr int lds = 3 * 3 * 64 * 256;
        // 1x1 / unit stride / unpadded, whole 64-row stages: the linear-address form of the same tile (PPV_WGRAD_LIN=0 or variant bit 0x4000: the
        // general kernel)
        static const int lin_on = getenv("PPV_WGRAD_LIN") ? atoi(getenv("PPV_WGRAD_LIN")) : 1;
        const bool lin = lin_on && !(g_wgrad_variant & 0x4000) && R * S == 1 && stride == 1 && pad == 0 && Hs == Ho && Ws == Wo && !g.chunked && g.M % 64 == 0 && g.native_slabs &&
                         !g.xcc_slabs && g.xcd_group && wgrad_debug == 0;
        static const int noslab = getenv("PPV_WGRAD_NOSLAB") ? atoi(getenv("PPV_WGRAD_NOSLAB")) : 0;
        if (lin && noslab) {
            g.pf_dist = 777;
            conv_wgrad_lin_kernel<3><<<grid, 512, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, g);
            return ppv_last_error();                       // (no reduce launch either)
        }
        if (lin) {
            static PpvDevOnce lin_once;
            if (lin_once.need()) {
                PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_lin_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                lin_once.done();
            }
            // PPV_WGRAD_PP / variant bit 0x8000 (toggles the default): the two waves of a SIMD in opposite phases
            static const int pp_on = getenv("PPV_WGRAD_PP") ? atoi(getenv("PPV_WGRAD_PP")) : 0;
            if ((pp_on != 0) != ((g_wgrad_variant & 0x8000) != 0)) {
                static PpvDevOnce pp_once;
                if (pp_once.need()) {
                    PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_lin_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                    pp_once.done();
                }
                conv_wgrad_lin_pp_kernel<3><<<grid, 512, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, g);
            } else
                conv_wgrad_lin_kernel<3><<<grid, 512, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, g);
        } else {
            static bool attr = false;
            if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_pipe_kernel<256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
            conv_wgrad_pipe_kernel<256, 3><<<grid, 512, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
        }
    } else if (small_ring) {
        constexpr int lds = 2 * 2 * 64 * 256;
        static bool attr = false;
        if (!attr) {
            PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_pipe_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_pipe_kernel<128, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr = true;
        }
        // cooperative L2 prefetch (see wgrad_pipe_body): 1x1 / unit stride with the m-slice's tiles grouped on one XCD
        const int pfd = wgrad_pf_dist();
        const int nt_ = N / 128, ct_ = Cs / 128;
        if (pfd > 0 && R * S == 1 && stride == 1 && g.xcd_group && !g.xcc_slabs && !(g_wgrad_variant & 0x800) && !(nt_ & (nt_ - 1)) && !(ct_ & (ct_ - 1)) &&
            nt_ <= 64 && ct_ <= 64) {
            g.pf_dist = pfd;
            conv_wgrad_pipe_kernel<128, 2, true><<<grid, 256, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
        } else
            conv_wgrad_pipe_kernel<128, 2><<<grid, 256, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
    } else {
        constexpr int lds = 4 * 2 * 64 * 256;
        static bool attr = false;
        if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_pipe_kernel<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
        conv_wgrad_pipe_kernel<128, 4><<<grid, 256, lds, stream>>>((const bf16_t*)G, (const bf16_t*)X, slabs, (const bf16_t*)zero_page, g);
    }
    if (g.native_slabs && deferred) {
        *deferred = PpvWgradReduce{slabs, dW_out, N, Cs, R, S, nslab, TN, 0, (int)((elems / 4 + 255) / 256)};
        return ppv_last_error();
    }
    if (g.native_slabs) wgrad_reduce_native_kernel<0><<<(unsigned)((elems / 4 + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, nslab, TN);
    else wgrad_to_torch_kernel<<<(unsigned)((elems + 255) / 256), 256, 0, stream>>>(slabs, dW_out, N, Cs, R, S, nslab);
    return ppv_last_error();
}

// Two 1x1 / unit-stride weight gradients in ONE launch (conv_wgrad_pair_kernel) + ONE reduce launch: G[k] [B,H[k],W[k],N[k]] bf16,
// X[k] [B,H[k],W[k],Cs[k]] bf16 -> dW[k] [N[k]][Cs[k]] f32 (torch layout of a 1x1 weight).  N[k] % 256 == 0, Cs[k] % 128 == 0.
// scratch: scratch_bytes bytes (PPV_ERR_WORKSPACE if the two slab sets do not fit).  ppv_conv_wgrad_pair_supported: 1 where this form
// runs (callers fall back to two ppv_conv_wgrad calls elsewhere).
int ppv_conv_wgrad_pair_supported(int B, int H0, int W0, int Cs0, int N0, int H1, int W1, int Cs1, int N1) {
    if (B < 1 || N0 % 256 || N1 % 256 || Cs0 % 128 || Cs1 % 128) return 0;
    const long M0 = (long)B * H0 * W0, M1 = (long)B * H1 * W1;
    if (M0 < 64 * 64 || M1 < 64 * 64 || M0 >= (1L << 24) || M1 >= (1L << 24) || M0 % 64 || M1 % 64) return 0;
    return (g_wgrad_variant & 0xff) == 0 && !(g_wgrad_variant & 0x1f00) && wgrad_native_slabs() ? 1 : 0;
}

int ppv_conv_wgrad_pair(const void* G0, const void* X0, float* dW0, int H0, int W0, int Cs0, int N0, const void* G1, const void* X1, float* dW1,
                        int H1, int W1, int Cs1, int N1, void* scratch, size_t scratch_bytes, const void* zero_page, int B, hipStream_t stream) {
    if (!G0 || !X0 || !dW0 || !G1 || !X1 || !dW1 || !scratch || !zero_page) return PPV_ERR_NULL;
    if (!ppv_conv_wgrad_pair_supported(B, H0, W0, Cs0, N0, H1, W1, Cs1, N1)) return PPV_ERR_BAD_SIZE;
    static const int target = getenv("PPV_WGRAD_PAIR_WGS") ? atoi(getenv("PPV_WGRAD_PAIR_WGS")) : 128;    // workgroups per problem
    WgradPair p;
    PpvWgradReduce red[2];
    const void* Gs[2] = {G0, G1};
    const void* Xs[2] = {X0, X1};
    float* outs[2] = {dW0, dW1};
    const int Hs_[2] = {H0, H1}, Ws_[2] = {W0, W1}, Cs_[2] = {Cs0, Cs1}, Ns_[2] = {N0, N1};
    size_t off = 0;
    unsigned grid = 0;
    for (int k = 0; k < 2; ++k) {
        WgradGeom& g = p.g[k];
        g.B = B; g.Hs = Hs_[k]; g.Ws = Ws_[k]; g.Cs = Cs_[k]; g.Ho = Hs_[k]; g.Wo = Ws_[k]; g.N = Ns_[k]; g.R = 1; g.S = 1; g.st = 1; g.pad = 0;
        g.M = (long)B * Hs_[k] * Ws_[k];
        g.xcc_slabs = 0; g.pf_dist = 0; g.chunked = 0; g.native_slabs = 1; g.xcd_group = 1;
        int TN, sps;
        long splits;
        wgrad_plan(g.M, g.N, 1, 1, g.Cs, 3, &TN, &splits, &sps, target < 16 ? 16 : target);
        if (TN != 256) return PPV_ERR_BAD_SIZE;
        g.stages_per_split = sps;
        g.splits = (int)splits;
        const long elems = (long)g.N * g.Cs;
        g.slab_elems = elems;
        const int tiles = (g.N / 256) * (g.Cs / 128);
        const unsigned wgs = (unsigned)(8 * ((splits + 7) / 8) * tiles);
        if (k == 0) p.wgs0 = (int)wgs;
        grid += wgs;
        p.G[k] = (const bf16_t*)Gs[k]; p.X[k] = (const bf16_t*)Xs[k];
        p.slabs[k] = (float*)((char*)scratch + off);
        red[k] = PpvWgradReduce{p.slabs[k], outs[k], g.N, g.Cs, 1, 1, (int)splits, 256, 0, (int)((elems / 4 + 255) / 256)};
        off += (size_t)splits * elems * sizeof(float);
        off = (off + 255) & ~(size_t)255;
    }
    if (off > scratch_bytes) return PPV_ERR_WORKSPACE;
    constexpr int lds = 3 * 3 * 64 * 256;
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_pair_kernel<256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_once.done();
    }
    conv_wgrad_pair_kernel<256, 3><<<grid, 512, lds, stream>>>(p, (const bf16_t*)zero_page);
    if (int e = ppv_last_error()) return e;
    return ppv_wgrad_reduce_multi(red, 2, stream);
}

// P <= 24 weight gradients of one 1x1 / unit-stride shape in one launch, each reduced over all its rows (no scratch, no reduce
// launch): G[p] [B,H,W,N] bf16, X[p] [B,H,W,Cs] bf16 -> out[p] [N][Cs] f32 (torch layout of a 1x1 weight).  N % 128 == 0, Cs % 128 == 0.
int ppv_conv_wgrad_group(const void* const* G, const void* const* X, float* const* out, int P, const void* zero_page, int B, int H, int W,
                         int Cs, int N, hipStream_t stream) {
    if (!G || !X || !out || !zero_page) return PPV_ERR_NULL;
    if (P < 1 || P > 24 || N % 128 || Cs % 128) return PPV_ERR_BAD_SIZE;
    WgradGeom g;
    g.B = B; g.Hs = H; g.Ws = W; g.Cs = Cs; g.Ho = H; g.Wo = W; g.N = N; g.R = 1; g.S = 1; g.st = 1; g.pad = 0;
    g.M = (long)B * H * W;
    if (g.M >= (1L << 24)) return PPV_ERR_BAD_SIZE;
    g.stages_per_split = (int)((g.M + 63) / 64); g.splits = 1; g.xcd_group = 1; g.slab_elems = (long)N * Cs; g.xcc_slabs = 0; g.pf_dist = 0; g.chunked = 0; g.native_slabs = 0;
    WgradGroupPtrs ptrs;
    for (int p = 0; p < 24; ++p) {
        const int q = p < P ? p : 0;
        if (!G[q] || !X[q] || !out[q]) return PPV_ERR_NULL;
        ptrs.G[p] = (const bf16_t*)G[q]; ptrs.X[p] = (const bf16_t*)X[q]; ptrs.out[p] = out[q];
    }
    const int tiles = (N / 128) * (Cs / 128);
    constexpr int lds = 2 * 2 * 64 * 256;
    static bool attr = false;
    if (!attr) { PPV_ATTR(hipFuncSetAttribute((const void*)conv_wgrad_group_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
    conv_wgrad_group_kernel<128, 2><<<(unsigned)(8 * ((P + 7) / 8) * tiles), 256, lds, stream>>>(ptrs, P, (const bf16_t*)zero_page, g);
    return ppv_last_error();
}

// the reduces ppv_conv_wgrad_ex left pending (n <= 4; entries with blocks == 0 are skipped), one launch
int ppv_wgrad_reduce_multi(const PpvWgradReduce* probs, int n, hipStream_t stream) {
    if (!probs) return PPV_ERR_NULL;
    if (n < 0 || n > 4) return PPV_ERR_BAD_SIZE;
    ReduceMulti m;
    m.n = 0;
    unsigned total = 0;
    for (int i = 0; i < n; ++i)
        if (probs[i].blocks > 0) { m.p[m.n++] = probs[i]; total += (unsigned)probs[i].blocks; }
    if (!m.n) return PPV_OK;
    wgrad_reduce_multi_kernel<<<total, 256, 0, stream>>>(m);
    return ppv_last_error();
}

// tuning / A-B hook: low byte 0 auto (= 6), 7 streamed kernel (conv_wgrad_stream_kernel) for everything but the fused-tap 3x3, 1 two-stage atomics kernel, 2 pipe TN=128 x 4 stages, 3 pipe TN=256 x 3 stages,
// 4 = fused-tap 3x3 + (3 | 1), 6 = fused-tap 3x3 + pipe TN=128 x 2 stages, 8 / 9 = 6 with the 256-wide tile (3 / 2 stages) where N % 256 == 0; 0x100 disables XCD grouping
int ppv_wgrad_set_variant(int v) { g_wgrad_variant = v; return PPV_OK; }

// mode 0: forward layout [64][24][8] bf16; mode 1: data-gradient layout [16][4][4][64] bf16
int ppv_stem_weight_layout(const float* w, void* out, int mode, hipStream_t stream) {
    if (!w || !out) return PPV_ERR_NULL;
    if (mode == 0) stem_weight_layout_kernel<3><<<(64 * 192 + 255) / 256, 256, 0, stream>>>(w, (bf16_t*)out);
    else if (mode == 2) stem_weight_layout_kernel<6><<<(64 * 384 + 255) / 256, 256, 0, stream>>>(w, (bf16_t*)out);   // [64][6][7][7] -> [64][48][8]
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
    constexpr int lds3 = 192 * 24 * 16 + 1024;
    static bool attr3 = false;
    if (!attr3) { PPV_ATTR(hipFuncSetAttribute((const void*)stem_conv_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3)); attr3 = true; }
    static const int rows_form = getenv("PPV_STEM_ROWS") ? atoi(getenv("PPV_STEM_ROWS")) : 1;
    if (rows_form && (W / 2) % 128 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0) {                      // one output row segment per tile: staged input rows, no gathers
        constexpr int ldsr = 12288 + 64 * 384 + 128 * 144 + 1024;
        static bool attrr = false;
        if (!attrr) { PPV_ATTR(hipFuncSetAttribute((const void*)stem_conv_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsr)); attrr = true; }
        stem_conv_rows_kernel<<<tiles < 512 ? (tiles + 7) / 8 * 8 : 512, 256, ldsr, stream>>>(img, (const bf16_t*)wst, (bf16_t*)out, stat_part, B, H, W, tiles,
                                                                            stat_rows < 1 ? 1 : stat_rows);
        return ppv_last_error();
    }
    stem_conv_kernel<3><<<grid, 256, lds3, stream>>>(img, (const bf16_t*)wst, (bf16_t*)out, stat_part, B, H, W, tiles, stat_rows < 1 ? 1 : stat_rows);
    return ppv_last_error();
}

// FAN CoordConv stem (Face-DeId/core/wing.py:184-186): img [B,6,H,W] f32 NCHW (3 image + 3 coordinate channels),
// wst from ppv_stem_weight_layout(mode 2) -> raw [B,H/2,W/2,64] bf16 (bias is folded into the following BatchNorm)
int ppv_stem_conv6(const float* img, const void* wst, void* out, int B, int H, int W, hipStream_t stream) {
    if (!img || !wst || !out) return PPV_ERR_NULL;
    if (H % 2 || W % 2) return PPV_ERR_BAD_SIZE;
    const long M = (long)B * (H / 2) * (W / 2);
    const int tiles = (int)((M + 127) / 128);
    const int grid = tiles < 512 ? tiles : 512;
    constexpr int lds6 = 192 * 48 * 16 + 1024;
    static bool attr6 = false;
    if (!attr6) { PPV_ATTR(hipFuncSetAttribute((const void*)stem_conv_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds6)); attr6 = true; }
    stem_conv_kernel<6><<<grid, 256, lds6, stream>>>(img, (const bf16_t*)wst, (bf16_t*)out, nullptr, B, H, W, tiles, 1);
    return ppv_last_error();
}

// Whole stem data gradient in one launch (row-staged form): g_raw [B,Ho,Wo,64] bf16, wsd from ppv_stem_weight_layout(mode 1),
// g_img [B,3,2Ho,2Wo] f32 NCHW, zero_page >= 128 zero bytes.  Wo % 128 == 0; other widths use ppv_conv_gemm (N = 16) + ppv_stem_dgrad_scatter.
int ppv_stem_dgrad(const void* g_raw, const void* wsd, float* g_img, const void* zero_page, int B, int Ho, int Wo,
                   hipStream_t stream) {
    if (!g_raw || !wsd || !g_img || !zero_page) return PPV_ERR_NULL;
    if (Wo % 128) return PPV_ERR_BAD_SIZE;
    constexpr int lds = 2 * 4 * 136 * 128 + 6 * 256 * 4;
    static bool attr = false;
    if (!attr) {
        PPV_ATTR(hipFuncSetAttribute((const void*)stem_dgrad_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        PPV_ATTR(hipFuncSetAttribute((const void*)stem_dgrad_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 136 * 128 + 6 * 256 * 4));
        attr = true;
    }
    static const int strip_form = getenv("PPV_STEM_DGRAD_STRIP") ? atoi(getenv("PPV_STEM_DGRAD_STRIP")) : 1;
    if (strip_form && Ho >= 8) {
        // strips of consecutive rows: ~256 workgroups, at least 8 rows each (the four-row prologue is staged once per strip)
        const int cols = B * (Wo / 128);
        int per_col = (256 + cols - 1) / cols;
        if (per_col > Ho / 8) per_col = Ho / 8;
        if (per_col < 1) per_col = 1;
        const int strip = (Ho + per_col - 1) / per_col;
        per_col = (Ho + strip - 1) / strip;
        stem_dgrad_strip_kernel<<<cols * per_col, 512, 6 * 136 * 128 + 6 * 256 * 4, stream>>>((const bf16_t*)g_raw, (const bf16_t*)wsd, g_img,
                                                                                           (const bf16_t*)zero_page, B, Ho, Wo, strip, per_col);
        return ppv_last_error();
    }
    const int tiles = B * Ho * (Wo / 128);
    stem_dgrad_rows_kernel<<<tiles < 256 ? (tiles + 7) / 8 * 8 : 256, 512, lds, stream>>>((const bf16_t*)g_raw, (const bf16_t*)wsd, g_img,
                                                                                  (const bf16_t*)zero_page, B, Ho, Wo, tiles);
    return ppv_last_error();
}

int ppv_stem_dgrad_scatter(const float* t, float* g, int B, int Ho, int Wo, hipStream_t stream) {
    if (!t || !g) return PPV_ERR_NULL;
    const long tot = (long)B * 3 * 4 * Ho * Wo;
    stem_dgrad_scatter_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(t, g, B, Ho, Wo);
    return ppv_last_error();
}

}  // extern "C"
