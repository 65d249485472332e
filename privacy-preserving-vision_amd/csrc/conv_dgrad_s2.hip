// Data gradient of the stride-2 convolutions (3x3 / pad 1 and the 1x1 projection shortcut of the first block of layers 2-4), gfx950.
//
// Reference semantics: autograd of the stride-2 torchvision Bottleneck convolutions behind Image_Caption/models.py:17-21
// (conv2 of layer{2,3,4}[0] and downsample[0]).  The gradient map dX [B, H, W, Cin] is twice the size of g [B, H/2, W/2, Cout]; a pixel
// (y, x) of dX receives tap (r, s) only if y + off + r and x + off + s are even.  ppv_conv_gemm's general form runs all R * S taps for
// every pixel and points the invalid ones at the zero page: three quarters of the staged bytes and of the MFMAs multiply zeros
// (tools/bench_layer4.py: 120-207 us per launch, 166-310 TFLOP/s of the algorithmic flops, six launches per step).
//
// Here the four parity classes (y & 1, x & 1) are four independent stride-1 problems on the g grid: class (py, px) owns the taps with
// (py + off + r) and (px + off + s) even (3x3: 1 + 2 + 2 + 4 taps; 1x1: one class has the tap, three are zero), its GEMM rows are the
// B * H/2 * W/2 pixels of that class, tap (r, s) reads g at (iy + (py + off + r) / 2, jx + (px + off + s) / 2), and the store loop of
// the shared epilogue scatters row (b, iy, jx) to pixel (2 iy + py, 2 jx + px) (RowMap of conv_tile_epilogue.h).  One launch, class-major
// grid, the four-tap class first (the four classes of a tile back to back on one XCD measured 1.3-2.1x slower).  K loop: the
// 256 x 128 / three-stage LDS-DMA pipeline of conv_gemm_pipe_kernel.
// Served: bf16 output, no addend, no ReLU bit mask (what trunk_plan.hip issues); with or without the BatchNorm-backward sums (RED).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"
#include "conv_tile_epilogue.h"

namespace ppv {

struct S2Class { int py, px, nt, tap[4], dr[4], ds[4]; };
struct S2Tab {
    S2Class c[4];
    int H, W, lh2, lw2;                  // gradient map; log2 of the g grid (H / 2, W / 2: powers of two)
};

struct S2RowMap {
    static constexpr bool identity = false;
    int lh2, lw2, H, W, py, px;
    __device__ __forceinline__ long operator()(long m) const {
        const int jx = (int)m & ((1 << lw2) - 1), iy = (int)(m >> lw2) & ((1 << lh2) - 1);
        const long b = m >> (lw2 + lh2);
        return (b * H + 2 * iy + py) * W + 2 * jx + px;
    }
};

constexpr int S2_BM = 256, S2_BN = 128, S2_NS = 3;
// BK = 32 (default): 76 KB (the epilogue's staging area), two workgroups per CU -- a class runs 2-64 K-steps, a workgroup is mostly
// prologue and store loop, and a second one beside it hides them (545 against 598 us over the six launches); BK = 64: 144 KB, one per CU
template <int BK> constexpr int s2_lds() {
    constexpr int ring = S2_NS * (S2_BM + S2_BN) * BK * 2, epi = S2_BM * (S2_BN * 2 + 32) + 4096;
    return ring > epi ? ring : epi;
}

template <bool RED, int BK>
__global__ __launch_bounds__(512, BK == 64 ? 1 : 2) void conv_dgrad_s2_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                               bf16_t* __restrict__ Out, float* __restrict__ stat_part,
                                                               const bf16_t* __restrict__ zero_page, ConvGeom g, S2Tab tab,
                                                               int tiles_per_class, int tiles_n, int stat_rows,
                                                               const bf16_t* __restrict__ red_x, const float* __restrict__ red_coef) {
    conv_signal_start(g);
    constexpr int BM = S2_BM, BN = S2_BN, NSTAGE = S2_NS, S2_LDS = s2_lds<BK>();
    constexpr int NT = 512, NWAVE = 8, ROWB = BK * 2, CH = BK / 8, RPI = 1024 / ROWB, RPR = NWAVE * RPI;
    constexpr int WN = 2, WROWS = 64, WCOLS = 64, MI = 4, NI = 4;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ASLOTS = BM / RPR, BSLOTS = BN / RPR, L = ASLOTS + BSLOTS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ci = blockIdx.x / tiles_per_class;                          // class-major, the class with most taps first
    int t = blockIdx.x % tiles_per_class;
    {
        const int nwg = tiles_per_class, q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const S2Class& cl = tab.c[ci];
    const int tile_m = t / tiles_n, tile_n = t % tiles_n;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;
    const S2RowMap rm{tab.lh2, tab.lw2, tab.H, tab.W, cl.py, cl.px};
    const int nt = cl.nt;

    if (nt == 0 && !RED) {                                                 // a class without taps (1x1): its pixels are zero
        constexpr int CPR = BN / 8, RSTEP = NT / CPR;
        const int row0 = tid / CPR, ch = tid % CPR;
        for (int row = row0; row < BM; row += RSTEP)
            if (m0 + row < g.M) store16_nt(Out + rm(m0 + row) * g.N + n0 + ch * 8, make_uint4(0, 0, 0, 0), g.nt & 4);
        return;
    }

    auto key = [](int row) { return BK == 64 ? (row & 7) : (((row >> 3) & 1) * 3); };      // as conv_gemm_pipe_kernel
    const int rl = lane / CH, p = lane % CH, cch = p ^ key(rl);
    int a_h0[ASLOTS], a_w0[ASLOTS], a_pix[ASLOTS];
    bool a_ok[ASLOTS];
    const long zdelta = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
        const long m = m0 + i * RPR + wave * RPI + rl;
        a_ok[i] = m < g.M;
        const long mm = a_ok[i] ? m : 0;
        a_w0[i] = (int)mm & ((1 << tab.lw2) - 1);
        a_h0[i] = (int)(mm >> tab.lw2) & ((1 << tab.lh2) - 1);
        a_pix[i] = (int)(mm >> (tab.lw2 + tab.lh2)) * g.Hs * g.Ws;
    }
    const int kc = g.Cs / BK, nk = nt * kc;
    const long wrow = (long)g.R * g.S * g.Cs;
    const bf16_t* wbase[BSLOTS];
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) wbase[i] = Wt + (long)(n0 + i * RPR + wave * RPI + rl) * wrow + cch * 8;

    int sti = 0, sc0 = 0;
    long skoff = 0;
    long a_off[ASLOTS];
    int a_inc[ASLOTS];
    auto retap = [&]() {
        const int ti = sti < nt ? sti : 0;
        const int dr = cl.dr[ti], ds = cl.ds[ti];
        skoff = (long)cl.tap[ti] * g.Cs;
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int hq = a_h0[i] + dr, wq = a_w0[i] + ds;
            const bool ok = a_ok[i] & ((unsigned)hq < (unsigned)g.Hs) & ((unsigned)wq < (unsigned)g.Ws);
            a_off[i] = (ok ? (long)(a_pix[i] + hq * g.Ws + wq) * g.Cs * 2 : zdelta) + cch * 16;
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
            ++sti;
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
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 < nk) stage(s0);
    int rd = 0, wr = (NSTAGE - 1) % NSTAGE;
    for (int t2 = 0; t2 < nk; ++t2) {
        const int younger = min(nk, t2 + NSTAGE - 1) - (t2 + 1);          // stages younger than t2 may stay in flight
        if (younger >= NSTAGE - 2) wait_vmcnt_le<(NSTAGE - 2) * L>();
        else wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        if (t2 + NSTAGE - 1 < nk) stage(wr);
        compute(rd);
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
    }
    __syncthreads();
#ifdef PPV_STAMPS
    unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tile_epilogue<BM, BN, S2_LDS, (BK == 64 ? 1 : 2), false, RED, MI, NI, false, S2RowMap>(acc, smem, Out, stat_part, nullptr, nullptr, g, ci * 7 + tile_m, stat_rows, red_x, red_coef, m0, n0, CoopBn{}, stamp_, rm);
#else
    tile_epilogue<BM, BN, S2_LDS, (BK == 64 ? 1 : 2), false, RED, MI, NI, false, S2RowMap>(acc, smem, Out, stat_part, nullptr, nullptr, g, ci * 7 + tile_m, stat_rows, red_x, red_coef, m0, n0, CoopBn{}, rm);
#endif
}

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// g: the geometry ppv_conv_gemm built for the general kernel (rows = pixels of the gradient map, source = g grid, div = 2)
bool conv_dgrad_s2_supported(const ConvGeom& g, int Cs, int div) {
    if (div != 2 || g.a != 1 || g.off != g.offw || g.R != g.S) return false;
    if (!((g.R == 3 && g.off == -1) || (g.R == 1 && g.off == 0))) return false;
    if (g.Ho != 2 * g.Hs || g.Wo != 2 * g.Ws || ilog2_exact(g.Hs) < 0 || ilog2_exact(g.Ws) < 0) return false;
    return Cs % 64 == 0 && g.N % S2_BN == 0;
}

int conv_dgrad_s2_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* zero_page, const bf16_t* red_x,
                         const float* red_coef, const ConvGeom& g, int stat_rows, hipStream_t stream) {
    if (!conv_dgrad_s2_supported(g, g.Cs, 2)) return PPV_ERR_BAD_SIZE;
    S2Tab tab;
    tab.H = g.Ho; tab.W = g.Wo; tab.lh2 = ilog2_exact(g.Hs); tab.lw2 = ilog2_exact(g.Ws);
    S2Class cls[4];
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            S2Class& c = cls[py * 2 + px];
            c.py = py; c.px = px; c.nt = 0;
            for (int i = 0; i < 4; ++i) c.tap[i] = c.dr[i] = c.ds[i] = 0;
            for (int r = 0; r < g.R; ++r)
                for (int s = 0; s < g.S; ++s)
                    if (((py + g.off + r) & 1) == 0 && ((px + g.offw + s) & 1) == 0) {
                        c.tap[c.nt] = r * g.S + s;
                        c.dr[c.nt] = (py + g.off + r) >> 1;                // arithmetic shift: -1 .. 1
                        c.ds[c.nt] = (px + g.offw + s) >> 1;
                        ++c.nt;
                    }
        }
    for (int i = 0; i < 4; ++i) {                                          // most taps first
        int best = 0;
        for (int j = 1; j < 4; ++j)
            if (cls[j].nt > cls[best].nt) best = j;
        tab.c[i] = cls[best];
        cls[best].nt = -1 - cls[best].nt;
    }
    for (int i = 0; i < 4; ++i)
        if (tab.c[i].nt < 0) tab.c[i].nt = -1 - tab.c[i].nt;
    ConvGeom gc = g;                                                       // one class: rows = B * Hs * Ws pixels of the g grid
    gc.Ho = g.Hs; gc.Wo = g.Ws; gc.M = (long)g.B * g.Hs * g.Ws; gc.sh = 0; gc.flat = 0; gc.chunked = 0;
    const int tiles_n = g.N / S2_BN, tiles_m = (int)((gc.M + S2_BM - 1) / S2_BM), tpc = tiles_m * tiles_n;
    static const int bk = getenv("PPV_S2_BK") ? atoi(getenv("PPV_S2_BK")) : 32;                    // A/B: 64 = one 144-KB workgroup per CU
    auto go = [&](auto BK_) -> int {
        constexpr int BK = decltype(BK_)::value, LDS = s2_lds<BK>();
        static PpvDevOnce attr_once;
        if (attr_once.need()) {
            PPV_ATTR(hipFuncSetAttribute((const void*)conv_dgrad_s2_kernel<false, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            PPV_ATTR(hipFuncSetAttribute((const void*)conv_dgrad_s2_kernel<true, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr_once.done();
        }
        if (red_x)
            conv_dgrad_s2_kernel<true, BK><<<4 * tpc, 512, LDS, stream>>>(X, Wt, (bf16_t*)out, stat_part, zero_page, gc, tab, tpc, tiles_n, stat_rows, red_x, red_coef);
        else
            conv_dgrad_s2_kernel<false, BK><<<4 * tpc, 512, LDS, stream>>>(X, Wt, (bf16_t*)out, nullptr, zero_page, gc, tab, tpc, tiles_n, stat_rows, nullptr, nullptr);
        return ppv_last_error();
    };
    return bk == 64 ? go(std::integral_constant<int, 64>{}) : go(std::integral_constant<int, 32>{});
}

}  // namespace ppv
