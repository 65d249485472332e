// NHWC bf16 implicit-GEMM convolution on MFMA (v_mfma_f32_16x16x32_bf16), gfx950.
//
// One kernel serves the forward convolution and the data gradient of every ResNet-101 bottleneck conv
// (reference Image_Caption/models.py:17-21 -> torchvision ResNet-101: 1x1 / 3x3, stride 1 / 2):
//
//   out[m][n] = sum_{r,s,c} src[b, (ho*a + r + off)/div, (wo*a + s + off)/div, c] * wt[n][r][s][c]
//
//   forward : a = stride, off = -pad,        div = 1,      wt = W[cout][r][s][cin]
//   dgrad   : a = 1,      off = -(k-1-pad),  div = stride, wt = W^T flipped [cin][k-1-r][k-1-s][cout]
//             (taps whose numerator is not divisible by `div`, or that fall outside, read a zero page)
//
// Two kernels:
//   conv_gemm_pipe_kernel<BM, BN, NSTAGE, BK, WG/CU> -- the product path.  NSTAGE LDS stages filled by global_load_lds_dwordx4
//     (the per-lane SOURCE address carries both the im2col gather and the XOR swizzle that makes every ds_read_b128 fragment
//     read bank-conflict free), counted s_waitcnt vmcnt + one raw s_barrier per K-step, XCD-aware tile order.  Tile shapes are
//     picked per problem by ppv_conv_gemm (cold-cache sweep, tools/bench_conv_cold.py): 256x128 BK32 two workgroups per CU
//     (grids of several rounds), 256x128 BK64 (one round), 128x128 BK64 four stages (layer 4), 128x64 four per CU (layer 1).
//   conv_gemm_kernel<BN, WM> -- the original 128-row two-stage kernel, kept for N = 16 (stem data gradient) and small grids.
// Epilogue of both: bf16 rounding; per-channel sum / sum-of-squares of the ROUNDED values (train-mode BatchNorm statistics,
// SURVEY 8a-18) folded per tile and added with f32 atomics into <= 32 partial rows; optional residual-gradient addend (one
// rounding of acc + addend) and ReLU bit mask of the destination tensor; the tile goes through LDS and leaves as 16-byte chunks.
// The RED instantiations of the pipe kernel (ppv_conv_gemm_red: data-gradient launches of the trunk's backward) also take, in that
// store loop, the sums the following BatchNorm backward needs (sum g, sum g * x per channel; optional recomputed ReLU mask), so
// that BatchNorm's own reduce pass over the stored tensor disappears.
#include <climits>
#include <utility>
#include "conv_common.h"
#include "conv_tile_epilogue.h"

namespace ppv {

// stat_part: [stat_rows][2][N] f32 partial (sum, sum of squares) of the bf16-rounded outputs (pre-zeroed; row tiles
// fold into row tile_m % stat_rows with f32 atomics), or null.
// addend: optional bf16 [M][N] tensor added to the result before rounding (residual-gradient accumulation).
template <int BN, int WM, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                           void* __restrict__ Out, float* __restrict__ stat_part,
                                                           const bf16_t* __restrict__ addend, const unsigned char* __restrict__ mask_bits,
                                                           const bf16_t* __restrict__ zero_page, ConvGeom g,
                                                           int tiles_n, int stat_rows) {
    conv_signal_start(g);
    constexpr int BM = 128, BK = 64, WN = 4 / WM;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int MI = BM / WM / 16, NI = BN / WN / 16;   // 16 x 16 fragments per wave
    __shared__ __attribute__((aligned(16))) char smem[2 * (A_BYTES + B_BYTES)];
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;           // stage s: A at s*STAGE_BYTES, B right behind it

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware tile order: blocks that share an XCD (bid % 8) walk a contiguous run of tiles, n fastest
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread staging roles: 4 A rows (+ NI*? B rows), one 16-byte chunk each
    const int rl = lane >> 3, p = lane & 7, cch = p ^ rl;       // global chunk that lands at LDS chunk p of row rl
    int a_h0[4], a_w0[4];
    long a_pix[4];
    bool a_ok[4];
    const int HoWo = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + i * 32 + wave * 8 + rl;
        a_ok[i] = m < g.M;
        const long mm = a_ok[i] ? m : 0;
        const int b = (int)(mm / HoWo), rem = (int)(mm % HoWo);
        const int ho = rem / g.Wo, wo = rem % g.Wo;
        a_h0[i] = ho * g.a + g.off;
        a_w0[i] = wo * g.a + g.offw;
        a_pix[i] = (long)b * g.Hs * g.Ws;
    }
    const int ktaps = g.R * g.S, kc = g.Cs / BK, nk = ktaps * kc;
    const long wrow = (long)ktaps * g.Cs;                       // elements per weight row

    auto stage = [&](int buf, int kstep) {
        const int tap = kstep / kc, c0 = (kstep % kc) * BK;
        const int r = tap / g.S, s = tap % g.S;
        const int dm = (1 << g.sh) - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hn = a_h0[i] + r, wn = a_w0[i] + s;
            const int hq = hn >> g.sh, wq = wn >> g.sh;
            const bool ok = a_ok[i] && hn >= 0 && wn >= 0 && ((hn | wn) & dm) == 0 && hq < g.Hs && wq < g.Ws;
            const bf16_t* src = ok ? X + ((a_pix[i] + (long)hq * g.Ws + wq) * g.Cs + c0 + cch * 8) : zero_page + cch * 8;
            GLDS16(src, smem + buf * STAGE_BYTES + (i * 32 + wave * 8) * 128);
        }
#pragma unroll
        for (int i = 0; i < (BN + 31) / 32; ++i) {
            if (BN >= 32 || wave < BN / 8) {
                const int n = n0 + i * 32 + wave * 8 + rl;
                const bf16_t* src = Wt + ((long)n * wrow + (long)tap * g.Cs + c0 + cch * 8);
                GLDS16(src, smem + buf * STAGE_BYTES + A_BYTES + (i * 32 + wave * 8) * 128);
            }
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WN, wn = wave % WN;
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    const int fr = lane & 15, fq = lane >> 4;

    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[MI], bfr[NI];
            const int chunk = ((kk * 4 + fq) ^ (fr & 7)) * 16;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(smem + buf * STAGE_BYTES + (wm * WROWS + mi * 16 + fr) * 128 + chunk);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bfr[ni] = *reinterpret_cast<const bf16x8*>(smem + buf * STAGE_BYTES + A_BYTES + (wn * WCOLS + ni * 16 + fr) * 128 + chunk);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nk - 1; ++t) {
        stage(cur ^ 1, t + 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);
    __syncthreads();

    // ---------------------------------------------------------------- epilogue
    if (OUT_F32) {
        float* out = reinterpret_cast<float*>(Out);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long m = m0 + wm * WROWS + mi * 16 + fq * 4 + j;
                    const int n = n0 + wn * WCOLS + ni * 16 + fr;
                    if (m < g.M) out[m * g.N + n] = acc[mi][ni][j];
                }
        return;
    }
    constexpr int LDO = BN * 2 + 16;                           // bytes per staged output row
    char* sO = smem;                                           // 128 * LDO bytes
    float* sStat = reinterpret_cast<float*>(smem + BM * LDO);  // [WM][2][BN]
    constexpr int CPR = BN / 8;                                // 16-byte chunks per row
    if (addend) {                                              // acc += addend, fragment-wise (bf16 reads, L2 resident or streamed once)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long m = m0 + wm * WROWS + mi * 16 + fq * 4 + j;
                    const int n = n0 + wn * WCOLS + ni * 16 + fr;
                    if (m < g.M) acc[mi][ni][j] += bf2f(addend[m * g.N + n]);
                }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        float s1 = 0.f, s2 = 0.f;
        const int col = wn * WCOLS + ni * 16 + fr;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t h = f2bf(acc[mi][ni][j]);
                const float v = bf2f(h);
                s1 += v;
                s2 += v * v;
                *reinterpret_cast<bf16_t*>(sO + (wm * WROWS + mi * 16 + fq * 4 + j) * LDO + col * 2) = h;
            }
        if (stat_part) {
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (fq == 0) {
                sStat[(wm * 2 + 0) * BN + col] = s1;
                sStat[(wm * 2 + 1) * BN + col] = s2;
            }
        }
    }
    __syncthreads();
    if (stat_part && tid < 2 * BN) {
        const int which = tid / BN, col = tid % BN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += sStat[(w * 2 + which) * BN + col];
        // rows are shared by tile_m % stat_rows: <= tiles_m / stat_rows adders per address, 256-byte runs
        atomicAdd(&stat_part[((long)(tile_m % stat_rows) * 2 + which) * g.N + n0 + col], v);
    }
    bf16_t* out = reinterpret_cast<bf16_t*>(Out);
#pragma unroll
    for (int it = 0; it < (BM * CPR + 255) / 256; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx / CPR, ch = idx % CPR;
        const long m = m0 + row;
        if (idx < BM * CPR && m < g.M)
        {
            uint4 v = *reinterpret_cast<const uint4*>(sO + row * LDO + ch * 16);
            if (mask_bits) v = relu_mask8(v, mask_bits[(m * g.N + n0) / 8 + ch]);
            *reinterpret_cast<uint4*>(out + m * g.N + n0 + ch * 8) = v;
        }
    }
}

// ============================================================================= multi-stage pipeline
// Same math as conv_gemm_kernel, restructured for latency hiding (the two-stage kernel above is latency-bound: one
// 64-deep K-step of prefetch covers ~0.5 us of compute against >1 us of L2/HBM latency):
//   * NSTAGE LDS stages, NSTAGE-1 K-steps of global_load_lds in flight, ONE raw s_barrier per K-step and a COUNTED
//     s_waitcnt vmcnt(L) (L = loads per thread per stage) so younger stages stay in flight across the barrier
//     (__syncthreads() would drain them: cdna_hip_programming.md "Pipelining across barriers");
//   * BM = 256 variant (8 waves, 48 KB per stage) halves the W-panel re-reads and cuts L2->LDS bytes per flop by 25 %;
//   * incremental (tap, channel) counters instead of per-stage integer divisions.
// RAW: a stage is read one barrier after every wave's counted wait retired its own loads of that stage.
// WAR: stage t+NSTAGE-1 overwrites the buffer last read in compute(t-1), which every wave finished before barrier t.

// RED (data-gradient launches that feed a BatchNorm backward): stat_part receives, instead of the forward statistics, the
// BN-backward sums of the tile as it is STORED (after addend, rounding and ReLU mask): sum g and sum g * red_x per column,
// red_x [M][N] bf16 = the raw conv output that BatchNorm normalised.  The separate reduce pass over g and x (2 T) becomes one
// extra read of x (1 T) in this store loop.  red_coef (BN scale / shift [2][N], may be null): the BatchNorm is followed by a ReLU
// without residual, so its mask (x * scale + shift > 0, the forward kernel's expression) is recomputed here, applied to the
// stored gradient and to the sums (single-pass store path only: launches without addend).
// LOOK (round 6, BK = 32 stages in a ring of six): the fragments of K-step t + 1 are read from LDS while the MFMAs of step t run.  The
// plain loop reads a step's fragments behind the step's barrier: all eight waves sit out the same LDS round trip with the matrix pipe idle,
// then issue MFMAs with the LDS pipe idle -- tools/ladder_1x1.py, 1024 -> 256 @16x16: K loop alone 21.8 us for 16 x (512 MFMA cycles + 512
// LDS cycles per SIMD) = 7 us of either pipe, loads alone 22.5 us, whole launch 29.4 us.  Here step t's counted wait covers stage t + 1 (one
// stage deeper), the barrier publishes it, and the look-ahead reads go out in front of the MFMA batch (two static register sets: the loop is
// unrolled by two).  Bytes in flight per workgroup stay 96 KB (four 24-KB stages instead of two 48-KB ones).
template <class F, int... Is>
__device__ __forceinline__ void static_for_pipe_(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for_pipe(F&& f) { static_for_pipe_(f, std::make_integer_sequence<int, N>{}); }
template <int OFF> __device__ __forceinline__ void lds_read128_off(bf16x8& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int STRIDE, int N, int... Is>
__device__ __forceinline__ void lds_read_frags(bf16x8 (&f)[N], unsigned addr, std::integer_sequence<int, Is...>) {
    (lds_read128_off<Is * STRIDE>(f[Is], addr), ...);
}
template <int BM, int BN, int NSTAGE, int BK, int WGPCU, bool OUT_F32, bool RED = false, bool COOP = false, bool LOOK = false, bool PP = false>
__global__ __launch_bounds__(BM * 2, (BM * 2 / 256) * WGPCU) void conv_gemm_pipe_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                                   void* __restrict__ Out, float* __restrict__ stat_part,
                                                                   const bf16_t* __restrict__ addend, const unsigned char* __restrict__ mask_bits,
                                                                   const bf16_t* __restrict__ zero_page, ConvGeom g,
                                                                   int tiles_n, int stat_rows, const bf16_t* __restrict__ red_x,
                                                                   const float* __restrict__ red_coef, CoopBn cb) {
    conv_signal_start(g);
    constexpr int NT = BM * 2, NWAVE = NT / 64;
    constexpr int ROWB = BK * 2, CH = BK / 8;                             // bytes / 16-byte chunks per staged row
    constexpr int RPI = 1024 / ROWB;                                       // rows per 1-KiB LDS-DMA wave-instruction
    constexpr int WN = BN / 64 > 0 ? BN / 64 : 1, WM = NWAVE / WN;        // wave grid; each wave owns 64 x (BN / WN)
    constexpr int WROWS = BM / WM, WCOLS = BN / WN, MI = WROWS / 16, NI = WCOLS / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int RPR = NWAVE * RPI;                                       // rows staged per round
    constexpr int ASLOTS = BM / RPR, BSLOTS = BN / RPR, L = ASLOTS + BSLOTS;
    static_assert(BN % RPR == 0 && WM * WN == NWAVE && MI * 16 == WROWS, "tile / wave grid mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    PPV_STAMP_DECL;
    PPV_STAMP(0);
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;

    // swizzle key of a row: 128-byte rows chunk ^= row & 7; 64-byte rows chunk ^= 3 * ((row >> 3) & 1) -- both make every
    // ds_read_b128 fragment read conflict-free (worked out per 16-lane b128 group)
    auto key = [](int row) { return BK == 64 ? (row & 7) : (((row >> 3) & 1) * 3); };
    const int rl = lane / CH, p = lane % CH, cch = p ^ key(rl);
    int a_h0[ASLOTS], a_w0[ASLOTS], a_pix[ASLOTS];
    bool a_ok[ASLOTS];
    const long zdelta = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);
    const int HoWo = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
        const long m = m0 + i * RPR + wave * RPI + rl;
        a_ok[i] = m < g.M;
        const long mm = a_ok[i] ? m : 0;
        if (g.flat) {                                           // 132 of the trunk's 206 launches: no divisions in the prologue
            a_h0[i] = 0; a_w0[i] = 0; a_pix[i] = (int)mm;
        } else {
            const int b = (int)(mm / HoWo), rem = (int)(mm % HoWo);
            const int ho = rem / g.Wo, wo = rem % g.Wo;
            a_h0[i] = ho * g.a + g.off;
            a_w0[i] = wo * g.a + g.offw;
            a_pix[i] = b * g.Hs * g.Ws;
        }
    }
    const int ktaps = g.R * g.S, kc = g.Cs / BK, nk = ktaps * kc;
    const long wrow = (long)ktaps * g.Cs;
    const int dm = (1 << g.sh) - 1;
    const bf16_t* wbase[BSLOTS];
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) wbase[i] = Wt + (long)(n0 + i * RPR + wave * RPI + rl) * wrow + cch * 8;

    // staging cursor (advances one K-step per call).  The gather addresses are rebuilt only when the tap changes
    // (once per Cs/64 K-steps); inside a tap every valid row just walks 128 bytes along its channel run, so the
    // steady-state cost per K-step is one 64-bit add per slot instead of ~25 VALU instructions.
    int sr = 0, ss = 0, sc0 = 0;
    long skoff = 0;
    long a_off[ASLOTS];
    int a_inc[ASLOTS];
    auto retap = [&]() {
        if (g.flat) {
            if (BK == 64 && g.chunked == 1) {      // channel-chunked source [Cs / 64][M][64]: a K-step's 256 rows are ONE contiguous 32-KB block
#pragma unroll
                for (int i = 0; i < ASLOTS; ++i) {
                    a_off[i] = (a_ok[i] ? (long)a_pix[i] * 128 : zdelta) + cch * 16;
                    a_inc[i] = a_ok[i] ? (int)(g.M * 128) : 0;
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < ASLOTS; ++i) {
                a_off[i] = (a_ok[i] ? (long)a_pix[i] * g.Cs * 2 : zdelta) + cch * 16;
                a_inc[i] = a_ok[i] ? BK * 2 : 0;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int hn = a_h0[i] + sr, wn = a_w0[i] + ss;
            const int hq = hn >> g.sh, wq = wn >> g.sh;
            const bool ok = a_ok[i] & ((unsigned)hq < (unsigned)g.Hs) & ((unsigned)wq < (unsigned)g.Ws) & (((hn | wn) & dm) == 0);
            const long eoff = (long)(a_pix[i] + hq * g.Ws + wq) * g.Cs;
            a_off[i] = (ok ? eoff * 2 : zdelta) + cch * 16;
            a_inc[i] = ok ? BK * 2 : 0;
        }
    };
    retap();
    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            GLDS16(reinterpret_cast<const char*>(X) + a_off[i], sa + (i * RPR + wave * RPI) * ROWB);
            a_off[i] += a_inc[i];
        }
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i) GLDS16(wbase[i] + skoff, sa + A_BYTES + (i * RPR + wave * RPI) * ROWB);
        skoff += BK;
        sc0 += BK;
        if (sc0 == g.Cs) {
            sc0 = 0;
            if (++ss == g.S) { ss = 0; ++sr; }
            retap();
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;

    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[MI], bfr[NI];
            const int chunk = ((kk * 4 + fq) ^ key(fr)) * 16;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(sa + (wm * WROWS + mi * 16 + fr) * ROWB + chunk);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(sa + A_BYTES + (wn * WCOLS + ni * 16 + fr) * ROWB + chunk);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // prologue: NSTAGE-1 stages in flight
    PPV_STAMP(5);
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nk) stage(s0);
    PPV_STAMP(1);
    int rd = 0, wr = (NSTAGE - 1) % NSTAGE;
    if constexpr (LOOK) {
        static_assert(BK == 32 && NSTAGE >= 4, "LOOK: one k-half per stage, a ring deep enough to keep stage t + 1 landed");
        bf16x8 afA[MI], bfA[NI], afB[MI], bfB[NI];
        // inline asm + hand-counted lgkmcnt (LDS returns in order): left to hipcc the MFMA batch sits behind s_waitcnt lgkmcnt(0), i.e. behind
        // the look-ahead reads issued just before it (its waitcnt pass loses the per-register picture across the loop back-edge)
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
        const unsigned a_base = lds0 + (wm * WROWS + fr) * ROWB + (fq ^ key(fr)) * 16;
        const unsigned b_base = lds0 + A_BYTES + (wn * WCOLS + fr) * ROWB + (fq ^ key(fr)) * 16;
        auto read_frags = [&](int buf, bf16x8 (&af)[MI], bf16x8 (&bfr)[NI]) __attribute__((always_inline)) {
            const unsigned a = a_base + buf * STAGE_BYTES, b = b_base + buf * STAGE_BYTES;
            lds_read_frags<16 * ROWB>(af, a, std::make_integer_sequence<int, MI>{});
            lds_read_frags<16 * ROWB>(bfr, b, std::make_integer_sequence<int, NI>{});
        };
        constexpr int LOOKN = MI + NI;
        auto frags_of_previous_step_are_in = [&]() __attribute__((always_inline)) {         // all but the MI + NI reads issued last
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LOOKN) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfmas = [&](const bf16x8 (&af)[MI], const bf16x8 (&bfr)[NI]) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        };
        // own loads of every stage <= s have landed when at most (issued - (s + 1)) stages' worth of instructions are outstanding
        auto wait_stage = [&](int s, int issued) __attribute__((always_inline)) {
            const int younger = issued - (s + 1);
            if (younger >= NSTAGE - 2) wait_vmcnt_le<(NSTAGE - 2) * L>();
            else if (younger == NSTAGE - 3) wait_vmcnt_le<(NSTAGE - 3) * L>();
            else if (NSTAGE >= 5 && younger == NSTAGE - 4) wait_vmcnt_le<(NSTAGE >= 5 ? (NSTAGE - 4) * L : 0)>();
            else if (NSTAGE >= 6 && younger == NSTAGE - 5) wait_vmcnt_le<(NSTAGE >= 6 ? (NSTAGE - 5) * L : 0)>();
            else wait_vmcnt_le<0>();
        };
        static_assert(NSTAGE == 6, "the tail below is written for a ring of six");
        wait_stage(0, min(nk, NSTAGE - 1));                                // (the one ladder of the workgroup)
        __builtin_amdgcn_s_barrier();
        PPV_STAMP(2);
        read_frags(0, afA, bfA);                                           // the one exposed LDS round trip of the workgroup
        const bool do_stage = g.chunked != 3, do_comp = g.chunked != 2;    // PPV_CONV_DEBUG timing modes
        // One step.  W = stages that may stay in flight across its wait (stage t + 1 has landed: this wave's part ... and, behind the
        // barrier, everybody's); every wave is past the MFMAs of step t - 1, i.e. past its wait for the fragments of stage t - 1, whose
        // buffer takes the DMA of stage t + NSTAGE - 1 (ISSUE).  W and ISSUE are compile-time: the steady-state loop carries NO scalar
        // branch ladder (tools/micro/lds_mfma_rate.hip: a four-way ladder costs ~90 ns per step, a third of the step).
        auto step = [&](auto W_, auto ISSUE_, const bf16x8 (&afc)[MI], const bf16x8 (&bfc)[NI], bf16x8 (&afn)[MI], bf16x8 (&bfn)[NI]) __attribute__((always_inline)) {
            wait_vmcnt_le<decltype(W_)::value * L>();
            __builtin_amdgcn_s_barrier();
            if constexpr (decltype(ISSUE_)::value) {
                if (do_stage) stage(wr);
            }
            const int nxt = (rd + 1 == NSTAGE) ? 0 : rd + 1;
            if (do_comp) {
                read_frags(nxt, afn, bfn);                                 // (the last step re-reads a landed buffer for nothing: no branch in the loop)
                frags_of_previous_step_are_in();
                mfmas(afc, bfc);
                __builtin_amdgcn_sched_barrier(0);
            }
            rd = nxt;
            wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
        };
        using std::integral_constant;
        int t = 0;
        for (; t + NSTAGE < nk; t += 2) {                                  // both steps of the pair issue a stage (nk is even: Cs % 64 == 0)
            step(integral_constant<int, NSTAGE - 3>{}, integral_constant<bool, true>{}, afA, bfA, afB, bfB);
            step(integral_constant<int, NSTAGE - 3>{}, integral_constant<bool, true>{}, afB, bfB, afA, bfA);
        }
        // tail: 2, 4 or 6 steps remain; their waits shrink with the stages left in flight (ladder: <= 6 times per workgroup), the first
        // of six still issues the last stage
        int issued = min(nk, t + NSTAGE - 1);
        auto step_tail = [&](int tt, const bf16x8 (&afc)[MI], const bf16x8 (&bfc)[NI], bf16x8 (&afn)[MI], bf16x8 (&bfn)[NI]) __attribute__((always_inline)) {
            wait_stage(min(tt + 1, nk - 1), issued);
            __builtin_amdgcn_s_barrier();
            if (tt + NSTAGE - 1 < nk) {
                if (do_stage) stage(wr);
                ++issued;
            }
            const int nxt = (rd + 1 == NSTAGE) ? 0 : rd + 1;
            if (do_comp) {
                read_frags(nxt, afn, bfn);
                frags_of_previous_step_are_in();
                mfmas(afc, bfc);
                __builtin_amdgcn_sched_barrier(0);
            }
            rd = nxt;
            wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
        };
        for (; t < nk; t += 2) {
            step_tail(t, afA, bfA, afB, bfB);
            step_tail(t + 1, afB, bfB, afA, bfA);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the stale look-ahead of the last step: its registers are free only now
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(afA[i]), "v"(afB[i]));
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(bfA[i]), "v"(bfB[i]));
        __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (PP) {
        // PING-PONG (round 6, variant 11; flat 1x1 launches on the one-round 256 x 128 BK64 tile): a K-step is a READ phase (this wave's
        // share of the DMA of stage t + 2, all sixteen fragment reads of stage t, wait) and an MFMA phase (32 MFMAs), a barrier after each;
        // waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave issues MFMAs while the other's LDS reads and DMA requests
        // fill the slots between them (conv_wgrad_lin_pp_kernel has the hazard argument).  The plain loop has all eight waves read
        // together behind the step barrier and then issue MFMAs together: 18.0 us for 16 steps where the loads alone take 13.5 and the
        // fragment + MFMA side alone 12.6 (profiles/r06_1x1_ladder.json).
        static_assert(NSTAGE == 3 && BK == 64 && NWAVE == 8, "written for the three-stage BK64 tile of eight waves");
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
        unsigned a_base[2], b_base[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned chunk = (unsigned)(((kk * 4 + fq) ^ key(fr)) * 16);
            a_base[kk] = lds0 + (wm * WROWS + fr) * ROWB + chunk;
            b_base[kk] = lds0 + A_BYTES + (wn * WCOLS + fr) * ROWB + chunk;
        }
        if (nk > 1) wait_vmcnt_le<L>();
        else wait_vmcnt_le<0>();                                          // this wave's share of stage 0 has landed
        const bool late = wave >= NWAVE / 2;
        if (late) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        for (int t = 0; t < nk; ++t) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef PPV_STAMPS
            if (t == 0) PPV_STAMP(2);
#endif
            // ---- read phase
            if (t + NSTAGE - 1 < nk) stage(wr);
            const unsigned bofs = (unsigned)rd * STAGE_BYTES;
            bf16x8 af[2][MI], bfr[2][NI];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                lds_read_frags<16 * ROWB>(af[kk], a_base[kk] + bofs, std::make_integer_sequence<int, MI>{});
                lds_read_frags<16 * ROWB>(bfr[kk], b_base[kk] + bofs, std::make_integer_sequence<int, NI>{});
            }
            if (t + NSTAGE - 1 < nk) wait_vmcnt_le<L>();                  // own share of stage t + 1 has landed (stage t + 2 may still fly)
            else wait_vmcnt_le<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA phase
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][mi], bfr[kk][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
            wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
        }
        if (!late) __builtin_amdgcn_s_barrier();                          // every wave has executed 2 nk + 1 barriers
    } else
    {
    const bool do_stage = g.chunked != 3, do_comp = g.chunked != 2;       // chunked 2 / 3: timing experiments (PPV_CONV_DEBUG: loads / compute only)
    int t = 0;
    // steady state (round 6): every step issues a stage and NSTAGE - 2 younger stages stay in flight -- one constant wait, no per-step
    // ladder of scalar branches (tools/micro/lds_mfma_rate.hip: ~90 ns per step)
    for (; t + NSTAGE - 1 < nk; ++t) {
        wait_vmcnt_le<(NSTAGE - 2) * L>();
        __builtin_amdgcn_s_barrier();
#ifdef PPV_STAMPS
        if (t == 0) PPV_STAMP(2);
#endif
        if (do_stage) stage(wr);
        if (do_comp) compute(rd);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
    }
    for (; t < nk; ++t) {                                                  // tail: nothing left to issue
        // stages issued: nk; those younger than stage t may stay in flight
        const int younger = nk - (t + 1);
        if (NSTAGE >= 3 && younger >= NSTAGE - 2) wait_vmcnt_le<(NSTAGE - 2) * L>();
        else if (NSTAGE >= 4 && younger == NSTAGE - 3) wait_vmcnt_le<(NSTAGE >= 4 ? (NSTAGE - 3) * L : 0)>();
        else wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
#ifdef PPV_STAMPS
        if (t == 0) PPV_STAMP(2);
#endif
        if (do_comp) compute(rd);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
    }
    }
    __syncthreads();
    PPV_STAMP(3);

    constexpr int LDO_ = BN * 2 + 32;
    constexpr int LDS_TOTAL = NSTAGE * STAGE_BYTES > BM * LDO_ + 4096 ? NSTAGE * STAGE_BYTES : BM * LDO_ + 4096;   // = the host's `lds`
#ifdef PPV_STAMPS
    tile_epilogue<BM, BN, LDS_TOTAL, WGPCU, OUT_F32, RED, MI, NI, COOP>(acc, smem, Out, stat_part, addend, mask_bits, g, tile_m, stat_rows, red_x, red_coef, m0, n0, cb, stamp_);
#else
    tile_epilogue<BM, BN, LDS_TOTAL, WGPCU, OUT_F32, RED, MI, NI, COOP>(acc, smem, Out, stat_part, addend, mask_bits, g, tile_m, stat_rows, red_x, red_coef, m0, n0, cb);
#endif
}

// ----------------------------------------------------------------------------- weight re-layouts
// torch [Cout][Cin][R][S] f32 -> forward GEMM rows [Cout][R][S][Cin] bf16
__global__ __launch_bounds__(256) void weight_fwd_layout_kernel(const float* __restrict__ w, bf16_t* __restrict__ o, int Cout,
                                                                int Cin, int R, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)Cout * Cin * R * S;
    if (i >= tot) return;
    const int c = (int)(i % Cin);
    const int s = (int)((i / Cin) % S), r = (int)((i / ((long)Cin * S)) % R);
    const int n = (int)(i / ((long)Cin * S * R));
    o[i] = f2bf(w[(((long)n * Cin + c) * R + r) * S + s]);
}

// torch [Cout][Cin][R][S] f32 -> dgrad GEMM rows [Cin][R][S][Cout] bf16, taps flipped
__global__ __launch_bounds__(256) void weight_dgrad_layout_kernel(const float* __restrict__ w, bf16_t* __restrict__ o,
                                                                  int Cout, int Cin, int R, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)Cout * Cin * R * S;
    if (i >= tot) return;
    const int n = (int)(i % Cout);
    const int s = (int)((i / Cout) % S), r = (int)((i / ((long)Cout * S)) % R);
    const int c = (int)(i / ((long)Cout * S * R));
    o[i] = f2bf(w[(((long)n * Cin + c) * R + (R - 1 - r)) * S + (S - 1 - s)]);
}

// one launch for a whole list of convolutions: desc[i] = {w f32*, fwd bf16*, dgrad bf16*, Cout, Cin, R, S, first block}
struct WLayoutDesc { const float* w; bf16_t* fwd; bf16_t* dg; int Cout, Cin, R, S, blk0; };
__global__ __launch_bounds__(256) void weight_layout_multi_kernel(const WLayoutDesc* __restrict__ desc, int ndesc) {
    // one block = a 32 (n) x 32 (c) x R*S tile through LDS: the torch rows are read as whole 32*R*S-float runs, both bf16
    // layouts leave as 64-byte runs (the element-per-thread form scattered 2-byte stores with a Cout*2-byte stride)
    __shared__ bf16_t sT[32][32 * 9 + 2];
    int lo = 0, hi = ndesc - 1;                                            // binary search of the descriptor that owns this block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WLayoutDesc d = desc[lo];
    const int RS = d.R * d.S, tiles_c = d.Cin / 32, lb = blockIdx.x - d.blk0;
    const int n0 = (lb / tiles_c) * 32, c0 = (lb % tiles_c) * 32, run = 32 * RS;
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int n = i / run, rem = i % run;
        sT[n][rem] = f2bf(d.w[((long)(n0 + n) * d.Cin + c0) * RS + rem]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int c = i & 31, rs = (i >> 5) % RS, n = i / run;
        d.fwd[((long)(n0 + n) * RS + rs) * d.Cin + c0 + c] = sT[n][c * RS + rs];
    }
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
        const int n = i & 31, rs = (i >> 5) % RS, c = i / run;
        d.dg[((long)(c0 + c) * RS + (RS - 1 - rs)) * d.Cout + n0 + n] = sT[n][c * RS + rs];
    }
}

}  // namespace ppv

using namespace ppv;

static int g_conv_variant = 0;
static thread_local bool g_addend_compact = false;         // conv_set_addend_compact: one-shot, consumed by the next conv_gemm_impl
static thread_local int g_conv_offw_override = INT_MIN;   // set by ppv_conv_gemm_rect around its call into conv_gemm_impl
#ifdef PPV_STAMPS
namespace ppv { __device__ unsigned long long* g_stamps = nullptr; }
#endif

namespace ppv {
void conv_set_addend_compact(bool on) { g_addend_compact = on; }
static thread_local int g_nt_once = -1;                       // conv_set_output_nt_once: one-shot, consumed by the next conv_gemm_impl
void conv_set_output_nt_once(int nt) { g_nt_once = nt; }
// one-shot: the next conv_gemm_impl launch stores `val` at `flag` when it starts (ConvGeom::start_flag); consumed whatever happens.  A
// call that returns PPV_OK has launched exactly one kernel that carries the flag; on any other status the CALLER settles the flag
// (ppv::fork_flag_settle), or the stream that waits for it never runs again.
static thread_local unsigned long long* g_start_flag = nullptr;
static thread_local unsigned long long g_start_val = 0;
void conv_set_start_flag_once(unsigned long long* flag, unsigned long long val) { g_start_flag = flag; g_start_val = val; }
bool conv_addend_compact_supported(int B, int H, int W, int Cs, int N) {
    static const int on = getenv("PPV_ADDEND_COMPACT") ? atoi(getenv("PPV_ADDEND_COMPACT")) : 1;   // A/B: 0 = dense shortcut gradient
    if (!on || g_conv_variant != 0 || H < 2 || W < 2 || (H & (H - 1)) || (W & (W - 1))) return false;
    ConvGeom g;
    g.B = B; g.Hs = H; g.Ws = W; g.Cs = Cs; g.Ho = H; g.Wo = W; g.N = N; g.R = 1; g.S = 1; g.a = 1; g.off = 0; g.offw = 0; g.sh = 0;
    g.M = (long)B * H * W; g.flat = 1; g.chunked = 0;
    return Cs <= 128 && conv1x1_stream_supported(g, Cs, 1);    // layer 2 (K = 128: 288 -> 211 us for the pair of launches); layer 3's K = 256 stream
}                                                              // kernel has no registers left for the gather (measured: no gain), layer 4 is tiled
}  // namespace ppv

extern "C" {

// tuning / A-B hook: 0 auto, 1 two-stage, 2 = 128x128x4-stage, 3 = 256x128x3-stage, 4 = 256x128 BK32 two per CU, 7 = tiled kernels only
// (no conv_stream.hip / conv_halo.hip), 8 = conv_stream.hip wherever it can run, 9 = conv_halo.hip wherever it can run,
// 10 = conv_dgrad_s2.hip wherever it can run
int ppv_conv_set_variant(int v) { g_conv_variant = v; return PPV_OK; }

#ifdef PPV_STAMPS
// diagnostic build: buf = device array of 16 * (largest grid) u64, or null to stop stamping
int ppv_debug_set_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(ppv::g_stamps), &buf, sizeof(buf)) == hipSuccess ? PPV_OK : PPV_ERR_INIT;
}
#endif

// Generic NHWC bf16 gather-GEMM convolution (see file header).  X [B,Hs,Ws,Cs] bf16, Wt [N][R*S*Cs] bf16,
// out [B*Ho*Wo][N] bf16 (out_f32 = 0) or f32 (out_f32 = 1: parity tests and the stem data gradient),
// stat_part [stat_rows][2][N] f32 BN partial sums, PRE-ZEROED by the caller (may be null), addend [M][N] bf16 (may be null),
// mask_bits [M * N / 8] bytes (may be null; bit k of byte i <-> element 8 i + k, as ppv_bn_act's pos_bits writes them): output
// lanes whose bit is clear are zeroed (ReLU backward folded into the data-gradient store); zero_page: >= 128 zero bytes.  Cs % 64 == 0; N % 64 == 0 or N == 16.
static int conv_gemm_impl(const void* X, const void* Wt, void* out, float* stat_part, const void* addend, const void* mask_bits,
                          const void* zero_page, const void* red_x_, const float* red_coef,
                          int B, int Hs, int Ws, int Cs, int Ho, int Wo, int N, int R, int S, int a, int off, int div,
                          int out_f32, int stat_rows, hipStream_t stream) {
    const bool compact = g_addend_compact;                   // one-shot request of conv_set_addend_compact, consumed whatever happens below
    g_addend_compact = false;
    const int nt_once = g_nt_once;
    g_nt_once = -1;
    unsigned long long* const start_flag = ppv::g_start_flag;
    const unsigned long long start_val = ppv::g_start_val;
    ppv::g_start_flag = nullptr;
    if (!X || !Wt || !out || !zero_page) return PPV_ERR_NULL;
    if (Cs % 64 || (N % 64 && N != 16) || (div != 1 && div != 2)) return PPV_ERR_BAD_SIZE;
    if (stat_part && stat_rows < 1) return PPV_ERR_BAD_SIZE;
    if (stat_part && addend && !red_x_) return PPV_ERR_BAD_SIZE; // the epilogue parks the addend tile where the statistics are folded
    if (red_x_ && (!stat_part || out_f32 || N % 64 || (red_coef && addend))) return PPV_ERR_BAD_SIZE;
    if (mask_bits && out_f32) return PPV_ERR_BAD_SIZE;          // the mask applies to the bf16 store path only (checked before ANY dispatch)
    ConvGeom g;
    g.B = B; g.Hs = Hs; g.Ws = Ws; g.Cs = Cs; g.Ho = Ho; g.Wo = Wo; g.N = N; g.R = R; g.S = S;
    g.a = a; g.off = off; g.offw = (g_conv_offw_override != INT_MIN) ? g_conv_offw_override : off; g.sh = (div == 2) ? 1 : 0;
    g.M = (long)B * Ho * Wo;
    g.flat = (R == 1 && S == 1 && a == 1 && off == 0 && g.offw == 0 && div == 1 && Hs == Ho && Ws == Wo) ? 1 : 0;
    g.chunked = 0;
    static const int nt_store = getenv("PPV_NT_STORE") ? atoi(getenv("PPV_NT_STORE")) : 1;         // A/B: 0 = ordinary output stores
    g.nt = nt_once >= 0 ? nt_once : nt_store;
    g.start_flag = start_flag;
    g.start_val = start_val;
    static const int conv_debug = getenv("PPV_CONV_DEBUG") ? atoi(getenv("PPV_CONV_DEBUG")) : 0;   // 1 = loads only, 2 = compute only (wrong results: timing experiments)
    if (conv_debug > 0) g.chunked = 1 + conv_debug;
    if (g_conv_variant & 0x1000) {            // layout A/B (tools/bench_layout_ab.py): flat launches on the BK = 64 tiles read a chunked source
        if (!g.flat || g.M * 128 >= (1L << 31)) return PPV_ERR_BAD_SIZE;
        g.chunked = 1;
    }
    // stride-2 data gradients as four parity-class problems (variant 10: wherever it can run; automatic: launches that fill the chip)
    static const int dgrad_s2 = getenv("PPV_DGRAD_S2") ? atoi(getenv("PPV_DGRAD_S2")) : 1;         // A/B: 0 = the general kernels (zero-page taps)
    if ((g_conv_variant == 10 || (g_conv_variant == 0 && dgrad_s2 && ((long)B * Hs * Ws / 256) * (N / 128) >= 64)) && !out_f32 && !addend &&
        !mask_bits && !(red_x_ && N % 128) && !(stat_part && !red_x_) && conv_dgrad_s2_supported(g, Cs, div))
        return conv_dgrad_s2_launch((const bf16_t*)X, (const bf16_t*)Wt, out, stat_part, (const bf16_t*)zero_page, (const bf16_t*)red_x_, red_coef, g,
                                    stat_rows, stream);
    if (compact) {
        int lw = 0, lh = 0;
        while ((1 << lw) < Wo) ++lw;
        while ((1 << lh) < Ho) ++lh;
        if (!addend || out_f32 || g_conv_variant != 0 || (1 << lw) != Wo || (1 << lh) != Ho || lw < 1 || lh < 1 || Cs > 128 || !conv1x1_stream_supported(g, Cs, div))
            return PPV_ERR_BAD_SIZE;
        g.add_lw = lw; g.add_lh = lh;
    }
    // few input channels, wide output: conv_stream.hip (variant 8: wherever it can run; automatic: launches with a residual addend)
    // (its f32 instantiation has no epilogue options: variant 8 leaves f32 launches with an addend / sums to the tiled kernels)
    static const int stream_auto = getenv("PPV_STREAM_DGRAD") ? atoi(getenv("PPV_STREAM_DGRAD")) : 1;   // A/B: 0 = tiled kernels for the addend launches too
    if (((g_conv_variant == 0 && addend && !out_f32 && (stream_auto || compact)) || (g_conv_variant == 8 && (!out_f32 || (!addend && !red_x_ && !stat_part)))) &&
        conv1x1_stream_supported(g, Cs, div))
        return conv1x1_stream_launch((const bf16_t*)X, (const bf16_t*)Wt, out, stat_part, (const bf16_t*)addend,
                                     (const unsigned char*)mask_bits, (const bf16_t*)zero_page, (const bf16_t*)red_x_, red_coef, g,
                                     out_f32, stat_rows, stream);
    // 3x3 / stride 1 with the input tile resident in LDS (variant 9: wherever it can run; automatic: launches that fill the chip)
    static const int halo_dgrad = getenv("PPV_HALO_DGRAD") ? atoi(getenv("PPV_HALO_DGRAD")) : 1;   // A/B: 0 = tiled kernels for the 3x3 data gradients
    static const int no_v3 = getenv("PPV_CONV_NO_V3") ? atoi(getenv("PPV_CONV_NO_V3")) : 0;        // A/B: 1 = 72-KB BK32 tile instead of the 144-KB BK64 one (2: data gradients only)
    if ((g_conv_variant == 9 || (g_conv_variant == 0 && (g.M / 256) * (N / 128) >= 200 && (halo_dgrad || !red_x_))) && !out_f32 && conv3x3_halo_supported(g, Cs, div))
        return conv3x3_halo_launch((const bf16_t*)X, (const bf16_t*)Wt, out, stat_part, (const bf16_t*)addend,
                                   (const unsigned char*)mask_bits, (const bf16_t*)zero_page, (const bf16_t*)red_x_, red_coef, g,
                                   stat_rows, stream);
    static const int halo_n64 = getenv("PPV_HALO_N64") ? atoi(getenv("PPV_HALO_N64")) : 1;         // A/B: 0 = 128 x 128 tiled kernel for layer 4's stride-1 3x3
    if ((g_conv_variant == 9 || (g_conv_variant == 0 && halo_n64 && (g.M / 256) * (N / 64) >= 200)) && !out_f32 && conv3x3_halo_n64_supported(g, Cs, div))
        return conv3x3_halo_n64_launch((const bf16_t*)X, (const bf16_t*)Wt, out, stat_part, (const bf16_t*)addend,
                                       (const unsigned char*)mask_bits, (const bf16_t*)zero_page, (const bf16_t*)red_x_, red_coef, g,
                                       stat_rows, stream);
    static const int halo64 = getenv("PPV_HALO64") ? atoi(getenv("PPV_HALO64")) : 1;               // A/B: 0 = 128 x 64 tiled kernel for layer 1's 3x3
    if ((g_conv_variant == 9 || (g_conv_variant == 0 && halo64 && g.M / 256 >= 512)) && !out_f32 && conv3x3_halo64_supported(g, Cs, div))
        return conv3x3_halo64_launch((const bf16_t*)X, (const bf16_t*)Wt, out, stat_part, (const bf16_t*)addend,
                                     (const unsigned char*)mask_bits, (const bf16_t*)zero_page, (const bf16_t*)red_x_, red_coef, g,
                                     stat_rows, stream);
    const int tiles_m = (int)((g.M + 127) / 128);
    const bf16_t* x = (const bf16_t*)X;
    const bf16_t* w = (const bf16_t*)Wt;
    const bf16_t* ad = (const bf16_t*)addend;
    const bf16_t* rx = (const bf16_t*)red_x_;
    const unsigned char* mk = (const unsigned char*)mask_bits;
    const bf16_t* z = (const bf16_t*)zero_page;
#define PPV_LAUNCH(BN_, WM_, TN_)                                                                                       \
    do {                                                                                                                \
        if (out_f32) conv_gemm_kernel<BN_, WM_, true><<<tiles_m * (TN_), 256, 0, stream>>>(x, w, out, stat_part, ad, mk, z, g, TN_, stat_rows); \
        else conv_gemm_kernel<BN_, WM_, false><<<tiles_m * (TN_), 256, 0, stream>>>(x, w, out, stat_part, ad, mk, z, g, TN_, stat_rows);        \
    } while (0)
#define PPV_LAUNCH_PIPE_(BM_, BN_, NS_, BK_, WG_, REDOK_)                                                               \
    do {                                                                                                                \
        constexpr int ring = NS_ * (BM_ + BN_) * BK_ * 2, epi = BM_ * (BN_ * 2 + 32) + 4096, lds = ring > epi ? ring : epi;             \
        const int tm = (int)((g.M + BM_ - 1) / BM_), tn = N / BN_;                                                      \
        auto kf = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, false, false>;                                         \
        auto kt = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, true, false>;                                          \
        auto kr = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, false, REDOK_>;                                        \
        static PpvDevOnce attr_once;                                                                                   \
        if (attr_once.need()) {                                                                                                \
            PPV_ATTR(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
            PPV_ATTR(hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
            PPV_ATTR(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
            attr_once.done();                                                                                            \
        }                                                                                                               \
        if (rx && !(REDOK_)) return PPV_ERR_BAD_SIZE;                                                                   \
        if (rx) kr<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, rx, red_coef, CoopBn{});           \
        else if (out_f32) kt<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, nullptr, nullptr, CoopBn{}); \
        else kf<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, nullptr, nullptr, CoopBn{}); \
    } while (0)
#define PPV_LAUNCH_PIPE_LOOK(BM_, BN_, NS_, BK_, WG_)                                                                    \
    do {                                                                                                                \
        constexpr int ring = NS_ * (BM_ + BN_) * BK_ * 2, epi = BM_ * (BN_ * 2 + 32) + 4096 + 8192, lds = ring > epi ? ring : epi; \
        const int tm = (int)((g.M + BM_ - 1) / BM_), tn = N / BN_;                                                      \
        auto kf = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, false, false, false, true>;                            \
        auto kr = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, false, true, false, true>;                             \
        static PpvDevOnce attr_once;                                                                                    \
        if (attr_once.need()) {                                                                                         \
            PPV_ATTR(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds));            \
            PPV_ATTR(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, lds));            \
            attr_once.done();                                                                                           \
        }                                                                                                               \
        if (out_f32) return PPV_ERR_BAD_SIZE;                                                                           \
        if (rx) kr<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, rx, red_coef, CoopBn{}); \
        else kf<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, nullptr, nullptr, CoopBn{}); \
    } while (0)
#define PPV_LAUNCH_PIPE_PP(BM_, BN_, NS_, BK_, WG_)                                                                      \
    do {                                                                                                                \
        constexpr int ring = NS_ * (BM_ + BN_) * BK_ * 2, epi = BM_ * (BN_ * 2 + 32) + 4096, lds = ring > epi ? ring : epi; \
        const int tm = (int)((g.M + BM_ - 1) / BM_), tn = N / BN_;                                                      \
        auto kf = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, false, false, false, false, true>;                     \
        auto kr = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, WG_, false, true, false, false, true>;                      \
        static PpvDevOnce attr_once;                                                                                    \
        if (attr_once.need()) {                                                                                         \
            PPV_ATTR(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds));            \
            PPV_ATTR(hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, lds));            \
            attr_once.done();                                                                                           \
        }                                                                                                               \
        if (rx) kr<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, rx, red_coef, CoopBn{}); \
        else kf<<<tm * tn, BM_ * 2, lds, stream>>>(x, w, out, stat_part, ad, mk, z, g, tn, stat_rows, nullptr, nullptr, CoopBn{}); \
    } while (0)
#define PPV_LAUNCH_PIPE(BM_, BN_, NS_, BK_, WG_) PPV_LAUNCH_PIPE_(BM_, BN_, NS_, BK_, WG_, false)
#define PPV_LAUNCH_PIPE_R(BM_, BN_, NS_, BK_, WG_) PPV_LAUNCH_PIPE_(BM_, BN_, NS_, BK_, WG_, true)
    // variant: 0 = auto, 1 = two-stage 128-row kernel, 2 = 128 x 128 x 4 stages, 3 = 256 x 128 x 3 stages (BK 64),
    // 4 = 256 x 128 x 3 stages of BK 32, two workgroups per CU
    const int CUS = 256;
    int v = g_conv_variant & 0xfff;
    if (v >= 7) v = 0;
    if (g.chunked == 1 && v != 2 && v != 3) return PPV_ERR_BAD_SIZE;   // the experiment exists on the two BK = 64 tiles
    if (N != 16 && N % 128 && (v == 0 || v >= 3) && g.M >= 128 * 1024) {
        PPV_LAUNCH_PIPE_R(128, 64, 3, 32, 4);   // 64-column layers (layer1): HBM-bound, four small-ring workgroups per CU
        return ppv_last_error();
    }
    if (N == 16 || N % 128) v = 1;
    if (v == 0) {                               // measured on the ResNet-101 shapes (tools/bench_conv.py, B = 128)
        const long t256 = ((g.M + 255) / 256) * (N / 128);
        if (t256 >= 2 * CUS) v = 4;             // several rounds of tiles: two workgroups per CU hide tile pro/epilogues
        else if (t256 >= CUS) v = (no_v3 == 1 || (no_v3 == 2 && rx)) ? 4 : 3;            // one round: deepest prefetch per workgroup
        else v = 2;                             // the 256-row tile only when it still fills the chip
    }
    static const int look = getenv("PPV_CONV_LOOK") ? atoi(getenv("PPV_CONV_LOOK")) : 0;          // A/B: 1 = the look-ahead form of the one-round 256 x 128 tile
    if (v == 3 && look && !out_f32 && (g_conv_variant & 0xfff) == 0 && g.chunked != 1) v = 6;
    if (v == 6 && (out_f32 || g.chunked == 1)) v = 3;     // (forced variant 6: the f32-output / layout-experiment launches stay on the plain tile)
    // ping-pong form of the same tile (variant 11 forces it; PPV_CONV_PP=1: wherever the automatic rule picks the one-round BK64 tile):
    // flat 1x1 launches with whole 64-channel K-steps, bf16 output, no timing mode
    static const int pp_auto = getenv("PPV_CONV_PP") ? atoi(getenv("PPV_CONV_PP")) : 0;
    const bool pp_ok = g.flat && !out_f32 && g.chunked == 0 && Cs % 64 == 0 && N % 128 == 0;
    if (((g_conv_variant & 0xfff) == 11 || (v == 3 && pp_auto && (g_conv_variant & 0xfff) == 0)) && pp_ok) v = 11;     // (elsewhere: the automatic rule's tile)
    // 256 x 256 tile (variant 12 forces it where N % 256 == 0; PPV_CONV_T256=1: the one-round launches with N == 256): one workgroup covers
    // every column of its 256 rows -- A is staged once instead of once per 128-column tile (134 instead of 201 MB at the layer-3
    // 1024 -> 256 shape) on HALF the workgroups: longer alone, less CU time where the launch shares the chip (backward: DESIGN 4c)
    static const int t256_auto = getenv("PPV_CONV_T256") ? atoi(getenv("PPV_CONV_T256")) : 0;
    const bool t256_ok = N % 256 == 0 && !out_f32 && g.chunked == 0;
    if (t256_ok && ((g_conv_variant & 0xfff) == 12 ||
                    (v == 3 && (g_conv_variant & 0xfff) == 0 && N == 256 && (t256_auto == 1 || (t256_auto == 2 && rx))))) v = 12;
    if (rx && (v < 2 || (v > 4 && v != 6 && v != 11 && v != 12))) return PPV_ERR_BAD_SIZE;  // the fused BN-backward sums exist in the production tiles only
    if (v == 5) PPV_LAUNCH_PIPE(128, 128, 3, 32, 3);      // 16 KB stages, three 4-wave workgroups per CU
    else if (v == 4) PPV_LAUNCH_PIPE_R(256, 128, 3, 32, 2); // 24 KB stages, two workgroups per CU (tile pro/epilogues overlap)
    else if (v == 3) PPV_LAUNCH_PIPE_R(256, 128, 3, 64, 1);
    else if (v == 12) PPV_LAUNCH_PIPE_R(256, 256, 4, 32, 1);
    else if (v == 11) PPV_LAUNCH_PIPE_PP(256, 128, 3, 64, 1);
    else if (v == 6) PPV_LAUNCH_PIPE_LOOK(256, 128, 6, 32, 1);  // fragment reads of step t + 1 under the MFMAs of step t (round 6)
    else if (v == 2) PPV_LAUNCH_PIPE_R(128, 128, 4, 64, 1);
    else if (N == 16) PPV_LAUNCH(16, 4, 1);
    else if (N % 128 == 0) PPV_LAUNCH(128, 2, N / 128);
    else PPV_LAUNCH(64, 2, N / 64);
#undef PPV_LAUNCH
#undef PPV_LAUNCH_PIPE
#undef PPV_LAUNCH_PIPE_R
#undef PPV_LAUNCH_PIPE_
#undef PPV_LAUNCH_PIPE_LOOK
#undef PPV_LAUNCH_PIPE_PP
    return ppv_last_error();
}

int ppv_conv_gemm(const void* X, const void* Wt, void* out, float* stat_part, const void* addend, const void* mask_bits,
                  const void* zero_page,
                  int B, int Hs, int Ws, int Cs, int Ho, int Wo, int N, int R, int S, int a, int off, int div,
                  int out_f32, int stat_rows, hipStream_t stream) {
    return conv_gemm_impl(X, Wt, out, stat_part, addend, mask_bits, zero_page, nullptr, nullptr, B, Hs, Ws, Cs, Ho, Wo, N, R, S, a, off, div,
                          out_f32, stat_rows, stream);
}

// Data-gradient launch that also takes the BN-backward sums of the tensor it stores: red_part [red_rows][2][N] f32 (PRE-ZEROED)
// receives sum g and sum g * red_x per column, g = the stored (addend-added, rounded, masked) output, red_x [M][N] bf16 = the raw
// convolution output the following BatchNorm normalised.  ppv_bn_bwd(..., part_prezeroed = 2) then skips its reduce pass.
// red_coef (may be null; not together with addend): that BatchNorm's [scale | shift] rows (ppv_bn_finalize's coef) when it is
// followed by a ReLU without residual: lanes with x * scale + shift <= 0 are stored as 0 and left out of the sums, so the
// BatchNorm backward runs with relu = 0.  bf16 output; N % 128 == 0, or N % 64 == 0 with M >= 128 Ki rows (the 128 x 64 tile).
// conv2 of an identity bottleneck with bn1 + ReLU in its operand path (conv_halo.hip BNIN): x_raw = conv1's raw output [B,H,W,C] bf16,
// sums [T][2][C] its partial sums, count = B*H*W; leaves coef [4][C] + running statistics (as ppv_bn_act_fold_rows), y_act = relu(bn(x_raw))
// (may be null), out [B,H,W,N] = conv3x3(y_act) raw and its statistics in stat_part [stat_rows][2][N] (PRE-ZEROED).
int ppv_conv3x3_bnin_supported(int B, int H, int W, int C, int N) {
    ConvGeom g;
    g.B = B; g.Hs = H; g.Ws = W; g.Cs = C; g.Ho = H; g.Wo = W; g.N = N; g.R = 3; g.S = 3; g.a = 1; g.off = -1; g.offw = -1; g.sh = 0;
    g.M = (long)B * H * W; g.flat = 0; g.chunked = 0;
    return (g_conv_variant == 0 && C % 64 == 0 && N % 128 == 0 && conv3x3_halo_bnin_supported(g, C)) ? 1 : 0;
}

int ppv_conv3x3_bnin(const void* x_raw, const float* sums, int T, double count, const float* gamma, const float* beta, float* run_mean,
                     float* run_var, float momentum, float eps, float* coef, void* y_act, const void* wt, void* out, float* stat_part,
                     int stat_rows, const void* zero_page, int B, int H, int W, int C, int N, hipStream_t stream) {
    if (!x_raw || !sums || !gamma || !beta || !coef || !wt || !out || !zero_page) return PPV_ERR_NULL;
    if (T < 1 || count < 1 || (stat_part && stat_rows < 1) || C % 64 || N % 128) return PPV_ERR_BAD_SIZE;
    ConvGeom g;
    g.B = B; g.Hs = H; g.Ws = W; g.Cs = C; g.Ho = H; g.Wo = W; g.N = N; g.R = 3; g.S = 3; g.a = 1; g.off = -1; g.offw = -1; g.sh = 0;
    g.M = (long)B * H * W; g.flat = 0; g.chunked = 0;
    static const int nt_store = getenv("PPV_NT_STORE") ? atoi(getenv("PPV_NT_STORE")) : 1;
    g.nt = nt_store;
    if (!conv3x3_halo_bnin_supported(g, C)) return PPV_ERR_BAD_SIZE;
    HaloBn bn;
    bn.sums = sums; bn.T = T; bn.inv_count = 1.0 / count; bn.unbias = count > 1 ? count / (count - 1.0) : 1.0;
    bn.gamma = gamma; bn.beta = beta; bn.run_mean = run_mean; bn.run_var = run_mean ? run_var : nullptr; bn.momentum = momentum; bn.eps = eps;
    bn.coef = coef; bn.y_act = (bf16_t*)y_act;
    if (run_mean && !run_var) return PPV_ERR_NULL;
    return conv3x3_halo_bnin_launch((const bf16_t*)x_raw, (const bf16_t*)wt, out, stat_part, (const bf16_t*)zero_page, g, stat_rows, bn, stream);
}

int ppv_conv_gemm_red(const void* X, const void* Wt, void* out, float* red_part, const void* red_x, const float* red_coef,
                      const void* addend, const void* mask_bits, const void* zero_page,
                      int B, int Hs, int Ws, int Cs, int Ho, int Wo, int N, int R, int S, int a, int off, int div,
                      int red_rows, hipStream_t stream) {
    if (!red_part || !red_x) return PPV_ERR_NULL;
    return conv_gemm_impl(X, Wt, out, red_part, addend, mask_bits, zero_page, red_x, red_coef, B, Hs, Ws, Cs, Ho, Wo, N, R, S, a, off, div, 0,
                          red_rows, stream);
}

// EXPERIMENT (round 4; not on the product path): 1x1 / unit-stride convolution + train-mode BatchNorm + ReLU in ONE launch
// of the 256 x 128 BK-64 tile, one workgroup per CU: the tiles leave their statistics, cross a grid barrier and apply the BatchNorm to
// the tile they still hold in LDS.  x_raw [M][N] bf16 (backward reads it) and y [M][N] bf16 are both written.  stats [stat_rows][2][N]
// and counter[2] PRE-ZEROED.  Refused (PPV_ERR_BAD_SIZE) unless every workgroup of the grid can be resident at once.
int ppv_conv_bn_relu_coop(const void* X, const void* Wt, void* x_raw, void* y, float* stats, unsigned* counter, const float* gamma,
                          const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* coef,
                          const void* zero_page, int B, int H, int W, int Cs, int N, int stat_rows, hipStream_t stream) {
    if (!X || !Wt || !x_raw || !y || !stats || !counter || !gamma || !beta || !coef || !zero_page) return PPV_ERR_NULL;
    if (Cs % 64 || N % 128 || stat_rows < 1) return PPV_ERR_BAD_SIZE;
    ConvGeom g;
    g.B = B; g.Hs = H; g.Ws = W; g.Cs = Cs; g.Ho = H; g.Wo = W; g.N = N; g.R = 1; g.S = 1;
    g.a = 1; g.off = 0; g.offw = 0; g.sh = 0; g.M = (long)B * H * W; g.flat = 1; g.chunked = 0;
    constexpr int BM_ = 256, BN_ = 128, NS_ = 3, BK_ = 64;
    constexpr int ring = NS_ * (BM_ + BN_) * BK_ * 2, epi = BM_ * (BN_ * 2 + 32) + 4096 + 8192, lds = ring > epi ? ring : epi;
    const int tm = (int)((g.M + BM_ - 1) / BM_), tn = N / BN_;
    auto k = conv_gemm_pipe_kernel<BM_, BN_, NS_, BK_, 1, false, false, true>;
    static PpvDevOnce attr_once;
    static int per_cu = 0, cus = 0;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, BM_ * 2, lds)) return -(int)e;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) return -(int)e;
        attr_once.done();
    }
    if ((long)tm * tn > (long)per_cu * cus) return PPV_ERR_BAD_SIZE;       // a grid barrier needs every workgroup resident
    CoopBn cb;
    cb.counter = counter; cb.gamma = gamma; cb.beta = beta; cb.run_mean = run_mean; cb.run_var = run_var; cb.coef = coef; cb.y = y;
    cb.count = (float)g.M; cb.momentum = momentum; cb.eps = eps;
    k<<<tm * tn, BM_ * 2, lds, stream>>>((const bf16_t*)X, (const bf16_t*)Wt, x_raw, stats, nullptr, nullptr, (const bf16_t*)zero_page, g, tn,
                                         stat_rows, nullptr, nullptr, cb);
    return ppv_last_error();
}

// Stride-1 convolution with a rectangular kernel and separate row / column paddings (RAFT SepConvGRU: 1 x 5 with padding (0, 2),
// 5 x 1 with padding (2, 0); Face-DeId/RAFT/core/update.py:36-42).  X [B,H,W,Cs] bf16, Wt [N][R][S][Cs] bf16 -> out [B,H,W,N],
// bf16 or f32 (out_f32 = 1).  Same kernels as ppv_conv_gemm.
int ppv_conv_gemm_rect(const void* X, const void* Wt, void* out, const void* zero_page, int B, int H, int W, int Cs, int N, int R,
                       int S, int pad_h, int pad_w, int out_f32, hipStream_t stream) {
    g_conv_offw_override = -pad_w;
    const int rc = conv_gemm_impl(X, Wt, out, nullptr, nullptr, nullptr, zero_page, nullptr, nullptr, B, H, W, Cs, H, W, N, R, S, 1,
                                  -pad_h, 1, out_f32, 0, stream);
    g_conv_offw_override = INT_MIN;
    return rc;
}

// rows of the BN partial buffer a conv with M output pixels should use
int ppv_conv_stat_tiles(long M) {
    const long t = (M + 127) / 128;
    return (int)(t < 32 ? t : 32);
}

// desc: device array of ndesc records {const float* w; void* fwd; void* dgrad; int Cout, Cin, R, S, blk0} (48 bytes each,
// blk0 = prefix sum of (Cout/32)*(Cin/32)); total_blocks = sum.  Converts every listed weight to both bf16 layouts.
// Cout % 32 == 0, Cin % 32 == 0, R*S <= 9.
int ppv_weight_layout_multi(const void* desc, int ndesc, int total_blocks, hipStream_t stream) {
    if (!desc || ndesc < 1) return PPV_ERR_NULL;
    weight_layout_multi_kernel<<<total_blocks, 256, 0, stream>>>((const WLayoutDesc*)desc, ndesc);
    return ppv_last_error();
}

// mode 0: [Cout][Cin][R][S] f32 -> [Cout][R][S][Cin] bf16 (forward);  mode 1: -> [Cin][R][S][Cout] bf16 flipped (dgrad)
int ppv_weight_layout(const float* w, void* out, int Cout, int Cin, int R, int S, int mode, hipStream_t stream) {
    if (!w || !out) return PPV_ERR_NULL;
    const long tot = (long)Cout * Cin * R * S;
    const unsigned gb = (unsigned)((tot + 255) / 256);
    if (mode == 0) weight_fwd_layout_kernel<<<gb, 256, 0, stream>>>(w, (bf16_t*)out, Cout, Cin, R, S);
    else weight_dgrad_layout_kernel<<<gb, 256, 0, stream>>>(w, (bf16_t*)out, Cout, Cin, R, S);
    return ppv_last_error();
}

}  // extern "C"
