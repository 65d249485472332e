// 3x3 / stride 1 / pad 1 NHWC bf16 convolution with the input tile RESIDENT in LDS ("halo" form), gfx950.
//
// Same GEMM as conv_gemm_pipe_kernel (rows = output pixels, columns = output channels, K = taps x input channels; reference
// semantics: the 3x3 convolutions of torchvision's Bottleneck behind Image_Caption/models.py:17-21 and their data gradients), other
// K order.  The tiled kernel walks K tap-major and stages a [256 pixels x 64 channels] slice of the input for EVERY tap: the same
// pixels nine times, shifted.  tools/tiled_timeline.py (layer 3, 256 -> 256 @ 16 x 16, B = 128): K loop 39.5 of 44.6 us per
// workgroup at 45 GB/s of LDS-DMA per CU (the L2 -> LDS rate), MFMA pipe 30 % busy -- the launch is bound by staged bytes, and
// 2/3 of them are those re-staged input slices.  Here a workgroup's 256 output pixels are whole image rows (16 x 16: one image;
// 32 x 32: 8 rows), so the pixels all nine taps touch are that rectangle plus a one-pixel border: per 64-channel chunk the
// (TH + 2) x (W + 2) halo tile is staged ONCE (<= 344 pixels x 128 B), the nine taps read it at shifted rows, and only the weights
// stream (one [128 x 64] tap slice per step through a four-slot ring).  Staged bytes per workgroup and chunk: 41-44 KB + 9 x 16 KB
// against 9 x 48 KB.
//
// Pipeline: one barrier and one counted vmcnt wait per step (tap); step t computes weight slot t % 4 from fragments whose first
// half was read during step t-1, while slice t+1 has landed and t+2, t+3 are in flight; the NEXT chunk's halo tile is requested in
// 1-KiB slices during the first steps of the current chunk into the other halo buffer (free since the previous chunk's last step).
// LDS-DMA completes in order, so a step's wait covers everything older than the weight slice it needs -- the halo slices of a
// chunk are all older than the chunk's first weight slice.
// Bank conflicts: a halo row is 128 B = 8 chunks of 16 B, stored at chunk ^ (halo column & 7): the 16 lanes of an A-fragment read
// hold 16 consecutive columns of one image row at one logical chunk -> 8 distinct physical chunks twice = conflict-free b128 reads
// for every tap shift (8-pixel-wide images: two rows per fragment, 2-way).  Weight rows as in conv_gemm_pipe_kernel (chunk ^ row & 7).
#include <utility>
#include "conv_common.h"
#include "conv_tile_epilogue.h"
#include "bn_coef.h"

namespace ppv {

// BNIN (round 6): the convolution's INPUT is the raw output of the previous convolution; train-mode BatchNorm + ReLU (coefficients folded
// from that convolution's partial sums by every workgroup's prologue) is applied to the LDS-resident halo tile once per 64-channel
// chunk, so the element-wise bn1 + ReLU launch between conv1 and conv2 of a bottleneck (read 2 B + write 2 B per element, one dispatch)
// disappears.  Workgroup 0 publishes coef [4][C] and the running statistics (what ppv_bn_act_fold_rows did); the column-tile-0
// workgroups write the activated tile back (y_act: the operand of conv2's weight gradient) from the registers that transform it.

struct HaloGeom {
    int H, W, TH, IMGS;      // image rows / columns, tile rows per image, images per tile (IMGS * TH * W == 256)
    int pitch;               // W + 2
    int hpx;                 // halo pixels of a tile: IMGS * (TH + 2) * pitch  (<= 344)
    int hpi;                 // 1-KiB halo instructions per chunk: ceil(hpx / 8)
    int hpw;                 // steps that carry a halo slice: ceil(hpi / 8)  (<= 6)
    int tiles_n;
};

// BN = 128: the form above (layers 2-3).  BN = 64 (round 5, layer 4: 512 -> 512 on 8 x 8 maps, M = 8192 rows): the 128 x 128 tiled kernel
// staged 2.3 MB per workgroup (79 us at the ~45 GB/s a CU pulls through L2 -> LDS, MFMA 15 us); with 256 workgroups the staged bytes per
// workgroup are least for 256 pixels x 64 columns (four images, 400 halo pixels per chunk: 0.41 MB of pixels + 0.59 MB of filter), eight
// waves x (32 pixels x 64 columns).
constexpr int HL_BM = 256, HL_BK = 64, HL_NT = 512, HL_NWS = 4;
template <int BN> struct HaloCfg {
    static constexpr int HALO_PX = BN == 128 ? 344 : 400, HALO_BYTES = HALO_PX * 128, WSTAGE = BN * HL_BK * 2;
    static constexpr int LDS = 2 * HALO_BYTES + HL_NWS * WSTAGE;             // 153 600 B / 135 168 B: one workgroup per CU
    static constexpr int WN = BN / 64, MI = BN == 128 ? 4 : 2, NI = 4, WI = WSTAGE / 8192;   // WI: 1-KiB filter instructions per wave and slice
    static constexpr int MAXS = (HALO_PX + 63) / 64;
};

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {       // n is wave-uniform
    if (n >= 3) wait_vmcnt_le<3>();
    else if (n == 2) wait_vmcnt_le<2>();
    else if (n == 1) wait_vmcnt_le<1>();
    else wait_vmcnt_le<0>();
}
// the BNIN form also has up to MAXS write-back stores in flight (stores and loads retire through the same in-order counter)
__device__ __forceinline__ void wait_vmcnt_dyn_wide(int n) {  // n is wave-uniform, any value >= 0 (larger values wait for less)
    switch (n < 11 ? n : 11) {
        case 0: wait_vmcnt_le<0>(); break;
        case 1: wait_vmcnt_le<1>(); break;
        case 2: wait_vmcnt_le<2>(); break;
        case 3: wait_vmcnt_le<3>(); break;
        case 4: wait_vmcnt_le<4>(); break;
        case 5: wait_vmcnt_le<5>(); break;
        case 6: wait_vmcnt_le<6>(); break;
        case 7: wait_vmcnt_le<7>(); break;
        case 8: wait_vmcnt_le<8>(); break;
        case 9: wait_vmcnt_le<9>(); break;
        case 10: wait_vmcnt_le<10>(); break;
        default: wait_vmcnt_le<11>(); break;
    }
}

template <bool RED, int HL_BN, bool BNIN = false>
__global__ __launch_bounds__(HL_NT, 1) void conv3x3_halo_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                                void* __restrict__ Out, float* __restrict__ stat_part,
                                                                const bf16_t* __restrict__ addend, const unsigned char* __restrict__ mask_bits,
                                                                const bf16_t* __restrict__ zero_page, ConvGeom g, HaloGeom hg,
                                                                int stat_rows, const bf16_t* __restrict__ red_x,
                                                                const float* __restrict__ red_coef, HaloBn bn = HaloBn{}) {
    conv_signal_start(g);
    static_assert(!(BNIN && RED), "BNIN is a forward form");
    using Cfg = HaloCfg<HL_BN>;
    constexpr int MI = Cfg::MI, NI = Cfg::NI, WN = Cfg::WN, WROWS = MI * 16, WCOLS = 64, WI = Cfg::WI;
    constexpr int HL_HALO_BYTES = Cfg::HALO_BYTES, HL_WSTAGE = Cfg::WSTAGE, HL_LDS = Cfg::LDS;
    static_assert(NI == 4, "four B fragments per k-half");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const s_halo = smem;
    char* const s_w = smem + 2 * HL_HALO_BYTES;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave: scalar, so the step's
    int bid = blockIdx.x;                                                                              // wait / issue conditions are too
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / hg.tiles_n, tile_n = bid % hg.tiles_n;
    const long m0 = (long)tile_m * HL_BM;
    const int n0 = tile_n * HL_BN;
    const int HW = hg.H * hg.W;
    const int b0 = (int)(m0 / HW), y0 = (int)(m0 % HW) / hg.W;
    const int rl = lane >> 3, p8 = lane & 7;
    const long zdelta = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);

    // ---- halo slices of this wave: slice t = 1-KiB instruction t * 8 + wave = halo pixels (t * 8 + wave) * 8 .. + 7
    constexpr int MAXS = Cfg::MAXS;                         // ceil(halo pixels / 64): 6 / 7
    long h_off[MAXS];
    int h_inc[MAXS];
    const int img_px = (hg.TH + 2) * hg.pitch;
    unsigned bn_ok = 0, bn_wb = 0, bn_hx7 = 0;              // BNIN, per slice t: pixel inside the image / inside THIS tile's rows / halo column & 7
#pragma unroll
    for (int t = 0; t < MAXS; ++t) {
        const int hp = (t * 8 + wave) * 8 + rl;
        const int img = hp / img_px, rem = hp % img_px;
        const int hy = rem / hg.pitch, hx = rem % hg.pitch;
        const int y = y0 + hy - 1, x = hx - 1;
        const bool ok = (hp < hg.hpx) & ((unsigned)y < (unsigned)hg.H) & ((unsigned)x < (unsigned)hg.W);
        const int gch = p8 ^ (hx & 7);
        h_off[t] = (ok ? ((long)(b0 + img) * HW + (long)y * hg.W + x) * g.Cs * 2 : zdelta) + gch * 16;
        h_inc[t] = ok ? HL_BK * 2 : 0;
        if constexpr (BNIN) {
            bn_ok |= (ok ? 1u : 0u) << t;
            bn_wb |= ((ok && hy >= 1 && hy <= hg.TH) ? 1u : 0u) << t;
            bn_hx7 |= (unsigned)(hx & 7) << (3 * t);
        }
    }
    auto issue_halo_slice = [&](int t, int buf) __attribute__((always_inline)) {       // t < hg.hpw, t * 8 + wave < hg.hpi checked by caller
        GLDS16(reinterpret_cast<const char*>(X) + h_off[t], s_halo + buf * HL_HALO_BYTES + (t * 8 + wave) * 1024);
        h_off[t] += h_inc[t];
    };
    // ---- weight slices: rows n0 .. n0 + 127 of Wt [N][9][Cs], 64 channels of one tap per step
    const long wrow = 9L * g.Cs;
    const bf16_t* wbase[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = (i * 8 + wave) * 8 + rl;
        wbase[i] = Wt + (long)(n0 + row) * wrow + (p8 ^ (row & 7)) * 8;
    }
    auto issue_w = [&](int slot, int tap, int c0) __attribute__((always_inline)) {
        char* sb = s_w + slot * HL_WSTAGE;
        const int koff = tap * g.Cs + c0;
#pragma unroll
        for (int i = 0; i < WI; ++i) GLDS16(wbase[i] + koff, sb + (i * 8 + wave) * 1024);
    };

    // ---- A-fragment addressing
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    int hb[MI];
    int xk = 0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * WROWS + mi * 16 + fr;
        const int x = p % hg.W, yy = (p / hg.W) % hg.TH, img = p / (hg.W * hg.TH);
        hb[mi] = (img * img_px + yy * hg.pitch + x) * 128;
        xk = x;                                             // x & 7 is the same for every mi (tiles start at multiples of 16 or 8)
    }
    int aoff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) aoff[s][kk] = ((kk * 4 + fq) ^ ((xk + s) & 7)) * 16;
    int boff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) boff[kk] = ((kk * 4 + fq) ^ (fr & 7)) * 16;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = g.Cs / HL_BK, T = nchunks * 9;
    // ---- BNIN: scale / shift of every input channel, folded from the producer's partial sums (thread c <-> channel c; the arithmetic of
    // bn_act_fold_wg_kernel, bn_coef.h), in a table behind the ring.  The sums are requested BEFORE the DMA prologue and consumed after it.
    static_assert(!BNIN || HL_BN == 128, "BNIN: the 128-column form (halo slices all issued by tap 5)");
    float* const s_sc = reinterpret_cast<float*>(smem + HL_LDS);
    float* const s_sh = s_sc + g.Cs;
    float bn_ps[4], bn_pq[4], bn_g = 0.f, bn_b = 0.f;
    if constexpr (BNIN) {
        const int cc = min(tid, g.Cs - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long o = ((long)min(t, bn.T - 1) * 2) * g.Cs + cc;
            bn_ps[t] = bn.sums[o];
            bn_pq[t] = bn.sums[o + g.Cs];
        }
        bn_g = bn.gamma[cc];
        bn_b = bn.beta[cc];
    }
    // prologue: halo tile of chunk 0, weight slices of steps 0, 1, 2
#pragma unroll
    for (int t = 0; t < MAXS; ++t)
        if (t < hg.hpw && t * 8 + wave < hg.hpi) issue_halo_slice(t, 0);
    issue_w(0, 0, 0);
    issue_w(1, 1, 0);
    issue_w(2, 2, 0);
    if constexpr (BNIN) {
        if (tid < g.Cs) {
            double s_ = 0, q_ = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < bn.T) { s_ += (double)bn_ps[t]; q_ += (double)bn_pq[t]; }
            for (int t = 4; t < bn.T; ++t) { s_ += (double)bn.sums[((long)t * 2) * g.Cs + tid]; q_ += (double)bn.sums[((long)t * 2 + 1) * g.Cs + tid]; }
            const BnCoef k = bn_coef_pinned(s_, q_, bn.inv_count, bn.unbias, bn_g, bn_b, bn.eps);
            s_sc[tid] = k.sc;
            s_sh[tid] = k.sh;
            if (blockIdx.x == 0) {                            // what ppv_bn_act_fold_rows left behind: coefficients for backward, running statistics
                const int C = g.Cs;
                bn.coef[tid] = k.sc; bn.coef[C + tid] = k.sh; bn.coef[2 * C + tid] = k.mean; bn.coef[3 * C + tid] = k.invstd;
                if (bn.run_mean) {
                    bn.run_mean[tid] = (1.f - bn.momentum) * bn.run_mean[tid] + bn.momentum * k.mean;
                    bn.run_var[tid] = (1.f - bn.momentum) * bn.run_var[tid] + bn.momentum * k.var_unbiased;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is written before the barrier below publishes it
        if (blockIdx.x == 0) wait_vmcnt_le<0>();              // (workgroup 0's coefficient stores are younger than the DMA: plain full wait there)
    }
    wait_vmcnt_le<2 * WI>();                                  // halo 0 and slice 0 are older than slices 1, 2
    __builtin_amdgcn_s_barrier();

    // BNIN: BatchNorm + ReLU on this wave's OWN slices of a landed halo tile, in place (lane = pixel rl of the slice, LOGICAL chunk p8 --
    // the same eight channels for every slice, so one set of coefficients per chunk; physical position p8 ^ (halo column & 7)); pixels
    // outside the image stay zero (the convolution's padding is applied AFTER the activation).  Column tile 0 also stores the activated
    // interior pixels to y_act from the same registers.  Returns nothing; bn_nst = store instructions issued (wave-uniform, exact: the
    // counted vmcnt waits of the next two steps allow exactly that many more operations in flight).
    int bn_nst = 0, bn_pending = 0;
    const bool bn_wb_tile = BNIN && bn.y_act != nullptr && tile_n == 0;
    auto bn_transform = [&](int c, char* hbuf) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        const int cb = c * HL_BK + p8 * 8;
        const float4 sa = *reinterpret_cast<const float4*>(s_sc + cb), sb = *reinterpret_cast<const float4*>(s_sc + cb + 4);
        const float4 ta = *reinterpret_cast<const float4*>(s_sh + cb), tb = *reinterpret_cast<const float4*>(s_sh + cb + 4);
        const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w}, sh[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
        int nst = 0;
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
            if (t < hg.hpw && t * 8 + wave < hg.hpi) {
                const int phys = (p8 ^ ((bn_hx7 >> (3 * t)) & 7)) * 16;
                char* a = hbuf + (t * 8 + wave) * 1024 + rl * 128 + phys;
                const uint4 v = *reinterpret_cast<const uint4*>(a);
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                unsigned o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = bn_relu_pinned(__builtin_bit_cast(float, w[i] << 16), sc[2 * i], sh[2 * i]);
                    const float hi = bn_relu_pinned(__builtin_bit_cast(float, w[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]);
                    o[i] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
                }
                const bool ok = (bn_ok >> t) & 1;
                const uint4 r = ok ? uint4{o[0], o[1], o[2], o[3]} : uint4{0u, 0u, 0u, 0u};
                *reinterpret_cast<uint4*>(a) = r;
                const bool mine = (bn_wb >> t) & 1;
                if (bn_wb_tile && __builtin_amdgcn_ballot_w64(mine) != 0) {     // wave-uniform: the store instruction is issued or not
                    ++nst;
                    if (mine) store16_nt(reinterpret_cast<char*>(bn.y_act) + (h_off[t] - HL_BK * 2 - phys + p8 * 16), r, g.nt);   // read next in backward
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the in-place writes are done before the next barrier publishes the tile
        __builtin_amdgcn_sched_barrier(0);
        bn_nst = nst;
        bn_pending = 2;                                       // the stores may stay in flight across the next two counted waits
    };
    if constexpr (BNIN) {
        bn_transform(0, s_halo);
        __builtin_amdgcn_s_barrier();
    }

    // Register pipeline: a step's first-half fragments (kk = 0) are read during the PREVIOUS step's second-half MFMAs, its
    // second-half fragments during its own first-half MFMAs -- the matrix pipe never waits for the LDS round trip that used to
    // open every step (measured before: 2530 cycles per step against 1024 of MFMA issue per SIMD).
    // The fragment reads are inline asm with hand-counted lgkmcnt waits: left to hipcc, every first-half MFMA batch sat behind
    // s_waitcnt lgkmcnt(0), i.e. behind the second-half reads issued just before it (LDS returns in order: lgkmcnt(8) is enough).
    bf16x8 af0[MI], bf0[NI];
    auto lds_addr = [](const char* p) __attribute__((always_inline)) {
        return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
    };
    auto load_frags = [&](bf16x8 (&af)[MI], bf16x8 (&bfr)[NI], const char* hbuf, const char* sb, int shift, int s, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const unsigned a = lds_addr(hbuf + hb[mi] + shift + aoff[s][kk]);
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[mi]) : "v"(a));
        }
        const unsigned b = lds_addr(sb + (wn * WCOLS + fr) * 128 + boff[kk]);
        asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[0]) : "v"(b));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(bfr[1]) : "v"(b));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bfr[2]) : "v"(b));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(bfr[3]) : "v"(b));
    };
    auto lgkm_wait8 = []() __attribute__((always_inline)) {                       // all but the MI + NI reads issued last
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(Cfg::MI + Cfg::NI) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    load_frags(af0, bf0, s_halo, s_w, 0, 0, 0);

    int step = 0, slot = 0;
    bool had_prev = false;                                   // this wave issued a halo slice in the previous step
    for (int c = 0; c < nchunks; ++c) {
        const char* hcur = s_halo + (c & 1) * HL_HALO_BYTES;
        const char* hoth = s_halo + ((c + 1) & 1) * HL_HALO_BYTES;
        const bool more = c + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap % 3;
            // slice step + 1 (and, at a chunk's last tap, the next halo tile: older than that slice) has landed for this wave ...
            if constexpr (BNIN) {
                wait_vmcnt_dyn_wide((step + 2 < T ? WI : 0) + (had_prev ? 1 : 0) + (bn_pending > 0 ? bn_nst : 0));
                bn_pending = bn_pending > 0 ? bn_pending - 1 : 0;
            } else {
                wait_vmcnt_dyn((step + 2 < T ? WI : 0) + (had_prev ? 1 : 0));
            }
            __builtin_amdgcn_s_barrier();                    // ... and for every wave; every wave is past step - 1
            had_prev = false;
            if (tap < MAXS) {
                if (more && tap < hg.hpw && tap * 8 + wave < hg.hpi) {
                    issue_halo_slice(tap, (c + 1) & 1);
                    had_prev = true;
                }
            }
            if (step + 3 < T) {
                int t3 = tap + 3, c3 = c;
                if (t3 >= 9) { t3 -= 9; ++c3; }
                issue_w((slot + 3) & 3, t3, c3 * HL_BK);
            }
            if constexpr (BNIN) {
                // tap 7: every slice of the next halo tile that THIS wave requested (taps 0-5) has landed (the wait above left only tap 6's
                // filter slice in flight); the barrier of tap 8 publishes the activated tile to the look-ahead reads of that step
                if (tap == 7 && more) bn_transform(c + 1, const_cast<char*>(hoth));
            }
            const char* sb = s_w + slot * HL_WSTAGE;
            const int shift = (r * hg.pitch + s) * 128;
            bf16x8 af1[MI], bf1[NI];
            load_frags(af1, bf1, hcur, sb, shift, s, 1);
            lgkm_wait8();                                    // the first-half fragments (read one step ago) are in
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0[mi], bf0[ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            {   // unconditional (the last step reads a landed slot for nothing): a branch here makes hipcc drain lgkmcnt to 0 below
                const int tn = (tap + 1) % 9;
                const int rn = tn / 3, sn = tn % 3;
                load_frags(af0, bf0, tap == 8 ? hoth : hcur, s_w + ((slot + 1) & 3) * HL_WSTAGE, (rn * hg.pitch + sn) * 128, sn, 0);
            }
            lgkm_wait8();                                    // this step's second-half fragments are in
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1[mi], bf1[ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ++step;
            slot = (slot + 1) & 3;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the last step's look-ahead reads: their registers are free only now
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af0[i]));
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(bf0[i]));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#ifdef PPV_STAMPS
    unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tile_epilogue<HL_BM, HL_BN, HL_LDS, 1, false, RED, MI, NI>(acc, smem, Out, stat_part, addend, mask_bits, g, tile_m, stat_rows, red_x, red_coef, m0, n0, CoopBn{}, stamp_);
#else
    tile_epilogue<HL_BM, HL_BN, HL_LDS, 1, false, RED, MI, NI>(acc, smem, Out, stat_part, addend, mask_bits, g, tile_m, stat_rows, red_x, red_coef, m0, n0, CoopBn{});
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Layer 1: 64 -> 64 channels on 64-column maps (three convolutions forward, three data gradients per step).  On the 128 x 64 tiled
// kernel these ran at 84 / 110 us against 30 / 45 us of compulsory bytes (tools/bench_layer1.py): K = 9 x 64 means every 128-pixel tile
// staged its pixels nine times AND the whole 72-KB filter -- 600 MB through the L2 -> LDS path for a 67-MB input.  Here a workgroup's
// 256 output pixels are four rows of one image; the 6 x 66-pixel halo tile (396 x 128 B) is staged once, one 8-KB tap slice of the
// filter per step runs through a three-slot ring (slices t + 1, t + 2 in flight while step t computes), and two workgroups share a CU
// (74 KB each) so one's prologue / store loop runs under the other's nine steps.  Eight waves x (32 pixels x 64 channels).
constexpr int H6_PITCH = 66, H6_HPX = 6 * H6_PITCH, H6_HPI = (H6_HPX + 7) / 8, H6_HALO = H6_HPI * 1024, H6_WSLOT = 64 * 128, H6_NWS = 3;
constexpr int H6_LDS = H6_HALO + H6_NWS * H6_WSLOT;              // 75 776 B

template <class F, int... Is>
__device__ __forceinline__ void h6_static_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int OFF> __device__ __forceinline__ void h6_read128(bf16x8& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <bool RED>
__global__ __launch_bounds__(512, 2) void conv3x3_halo64_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                                void* __restrict__ Out, float* __restrict__ stat_part,
                                                                const bf16_t* __restrict__ addend, const unsigned char* __restrict__ mask_bits,
                                                                const bf16_t* __restrict__ zero_page, ConvGeom g, int stat_rows,
                                                                const bf16_t* __restrict__ red_x, const float* __restrict__ red_coef) {
    conv_signal_start(g);
    constexpr int MI = 2, NI = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid;
    const long m0 = (long)tile_m * 256;
    const int HW = g.Ho * 64;
    const int b0 = (int)(m0 / HW), y0 = (int)(m0 % HW) >> 6;
    const int rl = lane >> 3, p8 = lane & 7;
    const long zdelta = reinterpret_cast<const char*>(zero_page) - reinterpret_cast<const char*>(X);
    // ---- halo tile: 1-KiB instruction i = halo pixels 8 i .. 8 i + 7, wave w issues i = 8 t + w
#pragma unroll
    for (int t = 0; t < (H6_HPI + 7) / 8; ++t) {
        const int inst = t * 8 + wave;
        const int hp = inst * 8 + rl;
        const int hy = hp / H6_PITCH, hx = hp % H6_PITCH;
        const int y = y0 + hy - 1, x = hx - 1;
        const bool ok = (hp < H6_HPX) & ((unsigned)y < (unsigned)g.Ho) & ((unsigned)x < 64u);
        const long off = (ok ? ((long)b0 * HW + (long)y * 64 + x) * 128 : zdelta) + ((p8 ^ (hx & 7)) * 16);
        if (inst < H6_HPI) GLDS16(reinterpret_cast<const char*>(X) + off, smem + inst * 1024);
    }
    // ---- filter slices: rows 0 .. 63 of Wt [64][9][64], one tap per step; wave w carries rows 8 w .. 8 w + 7
    const bf16_t* wbase;
    {
        const int row = wave * 8 + rl;
        wbase = Wt + (long)row * (9 * 64) + (p8 ^ (row & 7)) * 8;
    }
    const int dbg = g.chunked > 1 ? g.chunked - 1 : 0;        // PPV_CONV_DEBUG bits (timing experiments, wrong results): 1 = no fragment reads /
                                                              // MFMAs, 2 = no epilogue, 4 = only the first two filter slices are staged
    auto issue_w = [&](int slot, int tap) __attribute__((always_inline)) {
        GLDS16(wbase + tap * 64, smem + H6_HALO + slot * H6_WSLOT + wave * 1024);
    };
    issue_w(0, 0);
    issue_w(1, 1);

    const int fr = lane & 15, fq = lane >> 4;
    auto lds_addr = [](const char* p) __attribute__((always_inline)) {
        return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
    };
    unsigned abase[MI][3][2], bbase[2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wave * 32 + mi * 16 + fr;
        const unsigned hb = lds_addr(smem) + ((p >> 6) * H6_PITCH + (p & 63)) * 128;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) abase[mi][s][kk] = hb + (((kk * 4 + fq) ^ ((fr + s) & 7)) * 16);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) bbase[kk] = lds_addr(smem + H6_HALO) + fr * 128 + (((kk * 4 + fq) ^ (fr & 7)) * 16);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    h6_static_for([&](auto T_) {
        constexpr int T = decltype(T_)::value, r = T / 3, s = T % 3, slot = T % 3;
        // slice T (and, at T = 0, the halo tile: older) has landed for this wave; slice T + 1 may still be in flight
        if constexpr (T < 8) wait_vmcnt_le<1>(); else wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();                          // ... for every wave; every wave is past step T - 1 (slot (T + 2) % 3)
        if constexpr (T + 2 < 9) {
            if (!(dbg & 4)) issue_w((T + 2) % 3, T + 2);
        }
        if (dbg & 1) return;
        bf16x8 af[2][MI], bfr[2][NI];
        constexpr int ashift = (r * H6_PITCH + s) * 128, bshift = slot * H6_WSLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h6_read128<ashift>(af[kk][0], abase[0][s][kk]);
            h6_read128<ashift>(af[kk][1], abase[1][s][kk]);
            h6_read128<bshift>(bfr[kk][0], bbase[kk]);
            h6_read128<bshift + 2048>(bfr[kk][1], bbase[kk]);
            h6_read128<bshift + 4096>(bfr[kk][2], bbase[kk]);
            h6_read128<bshift + 6144>(bfr[kk][3], bbase[kk]);
        }
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][mi], bfr[0][ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][mi], bfr[1][ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, 9>{});
    if (dbg & 2) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(Out)[tid] = acc[1][3][2];
        return;
    }
    __syncthreads();
#ifdef PPV_STAMPS
    unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tile_epilogue<256, 64, H6_LDS, 2, false, RED, MI, NI>(acc, smem, Out, stat_part, addend, mask_bits, g, tile_m, stat_rows, red_x, red_coef, m0, 0, CoopBn{}, stamp_);
#else
    tile_epilogue<256, 64, H6_LDS, 2, false, RED, MI, NI>(acc, smem, Out, stat_part, addend, mask_bits, g, tile_m, stat_rows, red_x, red_coef, m0, 0, CoopBn{});
#endif
}

bool conv3x3_halo64_supported(const ConvGeom& g, int Cs, int div) {
    return g.R == 3 && g.S == 3 && g.a == 1 && g.off == -1 && g.offw == -1 && div == 1 && g.Hs == g.Ho && g.Ws == g.Wo && g.Wo == 64 &&
           g.Ho % 4 == 0 && Cs == 64 && g.N == 64 && g.M % 256 == 0;
}

int conv3x3_halo64_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                          const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                          const ConvGeom& g, int stat_rows, hipStream_t stream) {
    if (!conv3x3_halo64_supported(g, g.Cs, 1)) return PPV_ERR_BAD_SIZE;
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)conv3x3_halo64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, H6_LDS));
        PPV_ATTR(hipFuncSetAttribute((const void*)conv3x3_halo64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, H6_LDS));
        attr_once.done();
    }
    const int grid = (int)(g.M / 256);
    if (red_x)
        conv3x3_halo64_kernel<true><<<grid, 512, H6_LDS, stream>>>(X, Wt, out, stat_part, addend, mask_bits, zero_page, g, stat_rows, red_x, red_coef);
    else
        conv3x3_halo64_kernel<false><<<grid, 512, H6_LDS, stream>>>(X, Wt, out, stat_part, addend, mask_bits, zero_page, g, stat_rows, nullptr, nullptr);
    return ppv_last_error();
}

static bool halo_geom(const ConvGeom& g, int Cs, int div, HaloGeom* hg, int HL_BN = 128) {
    if (g.R != 3 || g.S != 3 || g.a != 1 || g.off != -1 || g.offw != -1 || div != 1 || g.Hs != g.Ho || g.Ws != g.Wo) return false;
    const int W = g.Wo, H = g.Ho;
    if (HL_BN == 128 ? (W != 16 && W != 32) : W != 8) return false;   // 128 columns: 16- / 32-wide maps; 64 columns: 8-wide maps (400 halo pixels); layer 1: above
    if (Cs % HL_BK || g.N % HL_BN || g.M % HL_BM) return false;
    const int rows = HL_BM / W;                      // image rows per tile
    const int TH = rows < H ? rows : H;
    if (H % TH || (HL_BM % (TH * W))) return false;
    const int IMGS = HL_BM / (TH * W);
    if (IMGS > 1 && TH != H) return false;
    hg->H = H; hg->W = W; hg->TH = TH; hg->IMGS = IMGS; hg->pitch = W + 2;
    hg->hpx = IMGS * (TH + 2) * (W + 2);
    hg->hpi = (hg->hpx + 7) / 8;
    hg->hpw = (hg->hpi + 7) / 8;
    hg->tiles_n = g.N / HL_BN;
    return hg->hpx <= (HL_BN == 128 ? HaloCfg<128>::HALO_PX : HaloCfg<64>::HALO_PX) && hg->hpw <= (HL_BN == 128 ? HaloCfg<128>::MAXS : HaloCfg<64>::MAXS);
}

bool conv3x3_halo_supported(const ConvGeom& g, int Cs, int div) {
    HaloGeom hg;
    return halo_geom(g, Cs, div, &hg);
}

bool conv3x3_halo_n64_supported(const ConvGeom& g, int Cs, int div) {
    HaloGeom hg;
    return halo_geom(g, Cs, div, &hg, 64);
}

template <int BN>
static int halo_launch_t(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                         const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                         const ConvGeom& g, int stat_rows, hipStream_t stream) {
    HaloGeom hg;
    if (!halo_geom(g, g.Cs, 1, &hg, BN)) return PPV_ERR_BAD_SIZE;
    constexpr int LDS = HaloCfg<BN>::LDS;
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        PPV_ATTR(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_once.done();
    }
    const int grid = (int)(g.M / HL_BM) * hg.tiles_n;
    if (red_x)
        conv3x3_halo_kernel<true, BN><<<grid, HL_NT, LDS, stream>>>(X, Wt, out, stat_part, addend, mask_bits, zero_page, g, hg, stat_rows, red_x, red_coef);
    else
        conv3x3_halo_kernel<false, BN><<<grid, HL_NT, LDS, stream>>>(X, Wt, out, stat_part, addend, mask_bits, zero_page, g, hg, stat_rows, nullptr, nullptr);
    return ppv_last_error();
}

// BNIN launch (forward of an identity bottleneck's conv2 with bn1 + ReLU in its operand path): 128-column form, 16- / 32-wide maps
bool conv3x3_halo_bnin_supported(const ConvGeom& g, int Cs) {
    HaloGeom hg;
    return halo_geom(g, Cs, 1, &hg, 128) && Cs <= 512 && hg.hpw <= 6 && (g.M / 256) * (g.N / 128) >= 200;
}

int conv3x3_halo_bnin_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* zero_page, const ConvGeom& g,
                             int stat_rows, const HaloBn& bn, hipStream_t stream) {
    HaloGeom hg;
    if (!conv3x3_halo_bnin_supported(g, g.Cs) || !halo_geom(g, g.Cs, 1, &hg, 128)) return PPV_ERR_BAD_SIZE;
    const int LDS = HaloCfg<128>::LDS + 2 * 512 * (int)sizeof(float);        // + the scale / shift table
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        PPV_ATTR(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<false, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_once.done();
    }
    const int grid = (int)(g.M / HL_BM) * hg.tiles_n;
    conv3x3_halo_kernel<false, 128, true><<<grid, HL_NT, LDS, stream>>>(X, Wt, out, stat_part, nullptr, nullptr, zero_page, g, hg, stat_rows, nullptr, nullptr, bn);
    return ppv_last_error();
}

int conv3x3_halo_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                        const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                        const ConvGeom& g, int stat_rows, hipStream_t stream) {
    return halo_launch_t<128>(X, Wt, out, stat_part, addend, mask_bits, zero_page, red_x, red_coef, g, stat_rows, stream);
}

int conv3x3_halo_n64_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                            const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                            const ConvGeom& g, int stat_rows, hipStream_t stream) {
    return halo_launch_t<64>(X, Wt, out, stat_part, addend, mask_bits, zero_page, red_x, red_coef, g, stat_rows, stream);
}

}  // namespace ppv
