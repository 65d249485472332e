// 1x1 convolutions with few input channels and many outputs (K = Cs <= 256, N = 4 K in the trunk: conv3 of every bottleneck
// forward, conv1's data gradient backward, the layer-1 / layer-2 projections) on MFMA, gfx950.
// Reference: torchvision bottleneck convolutions behind Image_Caption/models.py:17-21.
//
// These launches are STREAMS, not GEMMs: at B = 128 the layer-3 conv3 reads 17 MB and writes 67 MB for 17 GFLOP (7 us of
// MFMA at peak, 14 us of HBM).  The tiled kernel (conv_gemm.hip) restages the same pixel rows once per 128-column tile --
// eight times for N = 1024 -- through the L2 -> LDS path, which is what its K loop waits for, and its workgroups alternate
// between a load phase and a store phase in lockstep.  Here the roles are turned around:
//   * a workgroup owns 256 pixel rows and walks over the output channels in steps of 64;
//   * its pixel rows live in REGISTERS for the whole walk: each of the 8 consumer waves holds the MFMA fragments of its 32 rows
//     x K channels (K / 32 x 2 fragments = 64 VGPRs at K = 256), loaded once, straight from global memory;
//   * only the weights stream: 4 loader waves fill a 3-slot LDS ring with [64 channels][K] tiles by global_load_lds (source-side
//     XOR swizzle); there is NO barrier in the walk: every loader wave publishes "tiles landed" and every consumer wave "tiles
//     consumed" in its own LDS word, readers poll those words.  The waves of a SIMD therefore drift apart and one wave's epilogue
//     (an LDS round trip + stores: a latency chain, not a throughput load) runs beside another's MFMAs; with a barrier per step
//     every wave paid compute + epilogue back to back (3 us per step, 1.4 us is the HBM bound);
//   * a wave turns the 16 x 64 block of a step into row-major 16-byte chunks through a PRIVATE LDS patch (ds_write_b16 from the
//     packed accumulators, ds_read_b128 back: no barrier, other waves are not involved) and stores 128 contiguous bytes per row;
//     the residual addend / ReLU bit mask / BN-backward sums work on those chunks;
//   * the forward BatchNorm statistics cost no vector work: the packed (rounded) block is itself an MFMA operand, so
//     ones x block gives the column sums and block^T x block the Gram matrix whose diagonal is the sum of squares -- 8 extra
//     MFMAs per step beside 64.  (Element-wise statistics took 8-12 VALU instructions per output element and made the first
//     version of this kernel VALU-bound at 3.2 us per step; the HBM-bound step is 1.4 us.)
// Loaders never store and consumers never issue LDS-DMA, so neither's s_waitcnt vmcnt queue holds the other's operations.
#include <cstdlib>
#include "conv_common.h"

namespace ppv {

typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
constexpr int ST_NCW = 8, ST_NLW = 4, ST_NT = (ST_NCW + ST_NLW) * 64, ST_NSLOT = 3;

template <int KC, bool OUT_F32, bool RED, bool ADD, bool MASK>
__global__ __launch_bounds__(ST_NT, 3) void conv1x1_stream_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                                   void* __restrict__ Out, float* __restrict__ stat_part,
                                                                   const bf16_t* __restrict__ addend,
                                                                   const unsigned char* __restrict__ mask_bits, ConvGeom g,
                                                                   int n_splits, int nspan, int stat_rows,
                                                                   const bf16_t* __restrict__ red_x, const float* __restrict__ red_coef) {
    conv_signal_start(g);
    constexpr int ROWB = KC * 2, CH = KC / 8, RPI = 1024 / ROWB;          // bytes / 16-byte chunks per weight row; rows per LDS-DMA
    constexpr int TILE_BYTES = 64 * ROWB, LI = (64 / RPI) / ST_NLW;       // weight tile of one step; DMA instructions per loader wave
    constexpr int KK = KC / 32, MI = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PATCH = 16 * 272;                                       // a wave's private transposition patch: 16 rows, bf16 (144-byte
    char* sPatch = smem + ST_NSLOT * TILE_BYTES;                          // rows) or f32 (272-byte rows)
    float* sStat = reinterpret_cast<float*>(sPatch + ST_NCW * PATCH);     // [2][nspan] f32: cross-wave fold of the statistics
    // progress words, one per wave, each written by its owner only: sLanded[lw] = weight tiles loader lw has landed,
    // sDone[cw] = tiles consumer cw has finished reading
    volatile unsigned* sLanded = reinterpret_cast<volatile unsigned*>(sStat + 2 * nspan);
    volatile unsigned* sDone = sLanded + 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    PPV_STAMP_DECL;
    PPV_STAMP(0);
    int bid = blockIdx.x;
    {   // XCD-aware order: the n-slices of one row tile (same pixel rows) are neighbours inside one XCD's share of the grid
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / n_splits, ns = bid % n_splits;
    const long m0 = (long)tile_m * 256;
    const int n_begin = ns * nspan, NS = nspan / 64;                      // this workgroup's channels, steps of 64

    for (int t = tid; t < 2 * nspan + 12; t += ST_NT) sStat[t] = 0.f;     // statistics + the 12 progress words
    __syncthreads();

    if (wave >= ST_NCW) {
        // ------------------------------------------------------------ loader waves: weights -> LDS ring
        const int lw = wave - ST_NCW;
        const int rl = lane / CH, p = lane % CH;
        const bf16_t* src[LI];
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int row = (i * ST_NLW + lw) * RPI + rl;                 // LDS row of the tile (0..63)
            const int key = KC == 64 ? (row & 7) : (row & 15);            // chunk swizzle (see the fragment read)
            src[i] = Wt + (long)(n_begin + row) * KC + (p ^ key) * 8;
        }
        int slot = 0;
        long adv = 0;
        for (int t = 0; t < NS; ++t) {
            if (t >= ST_NSLOT) {                                            // the slot's previous tile: read by every consumer?
                const unsigned need = (unsigned)(t - ST_NSLOT + 1);
                for (;;) {
                    const st_u32x4 a = lds_poll4(sDone);             // address-space-3 reads: a generic volatile pointer compiles to
                    const st_u32x4 b = lds_poll4(sDone + 4);         // flat_load sc0 sc1 + s_waitcnt vmcnt(0) (conv_common.h)
                    const unsigned lo = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
                    if (__builtin_amdgcn_readfirstlane(lo) >= need) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
            }
            char* dst = smem + slot * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < LI; ++i) GLDS16(src[i] + adv, dst + ((i * ST_NLW + lw) * RPI) * ROWB);
            adv += 64L * KC;
            slot = slot + 1 == ST_NSLOT ? 0 : slot + 1;
            wait_vmcnt_le<0>();                                             // landed; the slots that are free bound the run-ahead
            if (lane == 0) lds_post(sLanded + lw, (unsigned)(t + 1));
            if (t == 0) PPV_STAMP(1);
            if (t == 1) PPV_STAMP(2);
            if (t == 2) PPV_STAMP(3);
            if (t == 3) PPV_STAMP(4);
        }
        PPV_STAMP(6);
        __builtin_amdgcn_s_barrier();                                        // the statistics fold's barrier
        PPV_STAMP(7);
        PPV_STAMP_FLUSH(8, ST_NCW * 64);
    } else {
        // ------------------------------------------------------------ consumer waves
        const int fr = lane & 15, fq = lane >> 4;
        // pixel-row fragments (MFMA A position): rows m0 + wave * 32 + mi * 16 + fr, channels kk * 32 + fq * 8 .. + 7
        bf16x8 af[MI][KK];
        const int HoWo = g.Ho * g.Wo;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const long m = m0 + wave * 32 + mi * 16 + fr;
            long pix = m;
            if (!g.flat) {
                const long mm = m < g.M ? m : 0;
                const int b = (int)(mm / HoWo), rem = (int)(mm % HoWo);
                const int ho = rem / g.Wo, wo = rem % g.Wo;
                pix = ((long)b * g.Hs + ho * g.a) * g.Ws + wo * g.a;
            }
            const bf16_t* src = X + pix * KC + fq * 8;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (m < g.M) v = *reinterpret_cast<const uint4*>(src + kk * 32);
                af[mi][kk] = __builtin_bit_cast(bf16x8, v);
            }
        }
        f32x4 acc[MI][4];
        const int key = KC == 64 ? (fr & 7) : fr;

        auto compute = [&](int slot) __attribute__((always_inline)) {
            const char* sw = smem + slot * TILE_BYTES;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                bf16x8 wf[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    wf[ni] = *reinterpret_cast<const bf16x8*>(sw + (ni * 16 + fr) * ROWB + (((kk * 4 + fq) ^ key) * 16));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][kk], wf[ni], acc[mi][ni], 0, 0, 0);
            }
        };

        // accumulator layout (v_mfma_f32_16x16x32_bf16 C/D): acc[mi][ni][r] = out[row mi * 16 + fq * 4 + r][channel ni * 16 + fr]
        char* patch = sPatch + wave * PATCH;
        const int ch = lane & 7, prow = lane >> 3;                           // chunk layout: 16-byte chunk ch of rows prow, prow + 8
        // FULL: every row of the tile exists (all but a ragged last tile): no branch around any load or store, so the compiler's
        // s_waitcnt vmcnt counts stay exact -- a conditional load makes it wait vmcnt(0) at every loop head, i.e. for the previous
        // step's STORES (measured: 3.3 us per step instead of 1.4)
        auto epilogue = [&](int j, auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int nc0 = n_begin + j * 64;                                // first channel of the step
            if constexpr (OUT_F32) {
                float* out = reinterpret_cast<float*>(Out);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const long m = m0 + wave * 32 + mi * 16 + fq * 4 + r;
                            if (FULL || m < g.M) out[m * g.N + nc0 + ni * 16 + fr] = acc[mi][ni][r];
                        }
            } else {
                bf16_t* out = reinterpret_cast<bf16_t*>(Out);
                const unsigned c = nc0 + ch * 8;                             // this lane's 8 channels in the chunk layout
                float ra[8], rb[8];                                          // RED: sum g, sum g * x of this lane's chunks
#pragma unroll
                for (int k = 0; k < 8; ++k) ra[k] = rb[k] = 0.f;
                float rsc[8], rsh[8];
                if constexpr (RED && !ADD) {                                 // (the host never pairs red_coef with an addend)
                    if (red_coef) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            rsc[k] = red_coef[c + k];
                            rsh[k] = red_coef[g.N + c + k];
                        }
                    }
                }
                unsigned pk[MI][4][2];                                       // packed bf16 pairs (rows r, r + 1) of the rounded block
                if constexpr (!ADD) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            pk[mi][ni][0] = pack2(acc[mi][ni][0], acc[mi][ni][1]);
                            pk[mi][ni][1] = pack2(acc[mi][ni][2], acc[mi][ni][3]);
                        }
                    if (!RED && stat_part) {
                        // BN statistics on the matrix pipe: the rounded 32 x 16 block of column group ni IS a K = 32 operand
                        // (k <-> its 32 rows, in any order): ones x block = column sums in every row of D; block^T x block =
                        // Gram matrix, diagonal = sums of squares (bf16 x bf16 products are exact in f32)
                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                        const bf16x8 ones = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            const bf16x8 blk = __builtin_bit_cast(bf16x8, (u32x4){pk[0][ni][0], pk[0][ni][1], pk[1][ni][0], pk[1][ni][1]});
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, blk, z, 0, 0, 0);
                            const f32x4 gm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blk, blk, z, 0, 0, 0);
                            // D[i = fq * 4 + r][j = fr]: column sum of channel fr in every row -> lanes fq == 0, r = 0;
                            // diagonal i == j -> lane fq == fr >> 2, r == fr & 3
                            const int rs = fr & 3;
                            const float dg = rs == 0 ? gm[0] : rs == 1 ? gm[1] : rs == 2 ? gm[2] : gm[3];
                            if (fq == 0) atomicAdd(&sStat[nc0 - n_begin + ni * 16 + fr], s1[0]);
                            if (fq == (fr >> 2)) atomicAdd(&sStat[nspan + nc0 - n_begin + ni * 16 + fr], dg);
                        }
                    }
                }
                // operands of the four chunks this lane stores in this step (rows prow, prow + 8 of both row blocks), ALL requested before
                // the first store of the step: a wait for loads issued behind a store also waits for that store (one in-order counter)
                uint4 av_[MI][2], xv_[MI][2];
                unsigned mb_[MI][2];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned m = (unsigned)m0 + wave * 32 + mi * 16 + h * 8 + prow;
                        const bool ok = FULL || m < (unsigned)g.M;
                        const unsigned off = (ok ? m : 0u) * (unsigned)g.N + c;  // ragged tile: row 0 stands in, the result is dropped
                        // (32-bit element offsets: SGPR base + VGPR offset addressing, the host checks M * N < 2^31)
                        av_[mi][h] = xv_[mi][h] = make_uint4(0, 0, 0, 0);
                        mb_[mi][h] = 0xff;
                        if constexpr (ADD) {
                            if (KC > 128 || g.add_lw == 0) {                 // (KC = 256 sits at its register cap: no compact form there, the host knows)
                                av_[mi][h] = *reinterpret_cast<const uint4*>(addend + off);
                            } else {                                         // compact addend: only the even-even pixels of the map have one
                                const unsigned x = m & ((1u << g.add_lw) - 1u), y = (m >> g.add_lw) & ((1u << g.add_lh) - 1u);
                                const unsigned b = m >> (g.add_lw + g.add_lh);
                                const unsigned cm = (((b << (g.add_lh - 1)) + (y >> 1)) << (g.add_lw - 1)) + (x >> 1);
                                if (ok && ((x | y) & 1u) == 0u) av_[mi][h] = *reinterpret_cast<const uint4*>(addend + (cm * (unsigned)g.N + c));
                            }
                        }
                        if constexpr (RED) {
                            xv_[mi][h] = *reinterpret_cast<const uint4*>(red_x + off);
                            if (!ok) xv_[mi][h] = make_uint4(0, 0, 0, 0);
                        }
                        if constexpr (MASK) mb_[mi][h] = mask_bits[off >> 3];
                    }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    uint4 (&av)[2] = av_[mi];
                    uint4 (&xv)[2] = xv_[mi];
                    unsigned (&mb)[2] = mb_[mi];
                    // ---- transposition of the 16 x 64 block through the wave's private patch
                    if constexpr (!ADD) {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                *reinterpret_cast<bf16_t*>(patch + (fq * 4 + r) * 144 + (ni * 16 + fr) * 2) =
                                    (bf16_t)(r & 1 ? pk[mi][ni][r >> 1] >> 16 : pk[mi][ni][r >> 1] & 0xffffu);
                    } else {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                *reinterpret_cast<float*>(patch + (fq * 4 + r) * 272 + (ni * 16 + fr) * 4) = acc[mi][ni][r];
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned m = (unsigned)m0 + wave * 32 + mi * 16 + h * 8 + prow;
                        uint4 pv;
                        if constexpr (!ADD) {
                            pv = *reinterpret_cast<const uint4*>(patch + (h * 8 + prow) * 144 + ch * 16);
                        } else {                                             // one rounding of (acc + addend)
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + (h * 8 + prow) * 272 + ch * 32);
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + (h * 8 + prow) * 272 + ch * 32 + 16);
                            float a[8];
                            unpack8(FULL || m < (unsigned)g.M ? av[h] : make_uint4(0, 0, 0, 0), a);
                            const float v[8] = {lo[0] + a[0], lo[1] + a[1], lo[2] + a[2], lo[3] + a[3],
                                                hi[0] + a[4], hi[1] + a[5], hi[2] + a[6], hi[3] + a[7]};
                            pv = pack8(v);
                        }
                        if constexpr (MASK) pv = relu_mask8(pv, mb[h]);
                        if constexpr (RED) {
                            if constexpr (!ADD) {
                                if (red_coef) pv = red_mask8(pv, xv[h], rsc, rsh);
                            }
                            red_acc8(pv, xv[h], ra, rb);
                        }
                        if (FULL || m < (unsigned)g.M) store16_nt(out + (m * (unsigned)g.N + c), pv, g.nt & 2);
                    }
                }
                if constexpr (RED) {                                         // fold the 8 lanes that share a chunk column, then the waves (LDS)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        ra[k] += __shfl_xor(ra[k], 8, 64); rb[k] += __shfl_xor(rb[k], 8, 64);
                        ra[k] += __shfl_xor(ra[k], 16, 64); rb[k] += __shfl_xor(rb[k], 16, 64);
                        ra[k] += __shfl_xor(ra[k], 32, 64); rb[k] += __shfl_xor(rb[k], 32, 64);
                    }
                    if (lane < 8) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            atomicAdd(&sStat[c - n_begin + k], ra[k]);
                            atomicAdd(&sStat[nspan + c - n_begin + k], rb[k]);
                        }
                    }
                }
            }
        };

        // the pixel-row loads retire HERE: left to the compiler, their wait lands at the first MFMA of the loop body and, being
        // a vmcnt(0), also waits for the previous step's stores in every later iteration (stores and loads share the counter)
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0)
        PPV_STAMP(1);
        auto walk = [&](auto full_tag) __attribute__((always_inline)) {
            int slot = 0;
            for (int j = 0; j < NS; ++j) {
                for (;;) {                                                   // weight tile j landed (all four loader waves)?
                    const st_u32x4 a = lds_poll4(sLanded);
                    const unsigned lo = min(min(a.x, a.y), min(a.z, a.w));
                    if (__builtin_amdgcn_readfirstlane(lo) > (unsigned)j) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                if (j == 0) PPV_STAMP(2);
                compute(slot);
                asm volatile("" ::: "memory");
                if (lane == 0) lds_post(sDone + wave, (unsigned)(j + 1));   // LDS executes a wave's instructions in order: behind the reads
                if (j == 0) { asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][3][3])); PPV_STAMP(3); }
                epilogue(j, full_tag);
                if (j == 0) PPV_STAMP(4);
                if (j == 1) PPV_STAMP(5);
                slot = slot + 1 == ST_NSLOT ? 0 : slot + 1;
            }
            PPV_STAMP(6);
        };
        if (m0 + 256 <= g.M) walk(std::true_type{}); else walk(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // this wave's LDS adds have been performed
        __builtin_amdgcn_s_barrier();
        PPV_STAMP(7);
        PPV_STAMP_FLUSH(0, 0);
    }
    if (!OUT_F32 && stat_part) {
        for (int t = tid; t < 2 * nspan; t += ST_NT) {
            const int which = t / nspan, col = t % nspan;
            atomicAdd(&stat_part[((long)(tile_m % stat_rows) * 2 + which) * g.N + n_begin + col], sStat[t]);
        }
    }
}

// Shapes the kernel can run.  The automatic rule of ppv_conv_gemm takes it for the data-gradient launches that add a residual
// gradient (addend: conv1's data gradient of every bottleneck, the heaviest store loop of the step: 61 vs 67 us at 256 -> 1024,
// 104 vs 112 us at 128 -> 512, 203 vs 213 us at 64 -> 256, B = 128); plain forward launches stay on the tiled kernel, which is
// 20-45 % faster there (tools/conv_timeline.py: per wave and step the epilogue here is an LDS round trip + stores behind the
// MFMAs, 2.3-2.9 us against the 1.4 us a step's 32 KB of output takes at HBM rate).
bool conv1x1_stream_supported(const ConvGeom& g, int Cs, int div) {
    if (g.R != 1 || g.S != 1 || g.off != 0 || g.offw != 0 || div != 1) return false;
    if ((g.M + 256) * g.N >= (1L << 31)) return false;          // 32-bit element offsets in the epilogue
    if (Cs != 64 && Cs != 128 && Cs != 256) return false;
    if (g.N % 64 || g.N < 2 * Cs) return false;                 // wide outputs only: the narrow ones are read-bound (tiled kernel)
    return true;
}

template <int KC>
static int stream_launch_k(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                           const unsigned char* mask_bits, const bf16_t* red_x, const float* red_coef, const ConvGeom& g,
                           int out_f32, int stat_rows, hipStream_t stream) {
    const int tiles_m = (int)((g.M + 255) / 256), steps = g.N / 64;
    // enough workgroups for every CU; a row tile's channel range is cut at most 8 ways (each slice re-reads the pixel rows)
    static const int wg_target = getenv("PPV_STREAM_WGS") ? atoi(getenv("PPV_STREAM_WGS")) : 256;   // A/B: finer slices flow around blocked CUs
    int n_splits = 1;
    while (n_splits < 8 && tiles_m * n_splits < wg_target && steps % (n_splits * 2) == 0 && steps / (n_splits * 2) >= 2) n_splits *= 2;
    const int nspan = g.N / n_splits;
    const int lds = ST_NSLOT * 64 * KC * 2 + ST_NCW * 16 * 272 + 2 * nspan * 4 + 64;
    typedef void (*kern_t)(const bf16_t*, const bf16_t*, void*, float*, const bf16_t*, const unsigned char*, ConvGeom, int, int, int,
                           const bf16_t*, const float*);
    // [RED][ADD][MASK] for the bf16 output; the f32 output (parity tests) has no epilogue options
    static const kern_t tab[2][2][2] = {
        {{conv1x1_stream_kernel<KC, false, false, false, false>, conv1x1_stream_kernel<KC, false, false, false, true>},
         {conv1x1_stream_kernel<KC, false, false, true, false>, conv1x1_stream_kernel<KC, false, false, true, true>}},
        {{conv1x1_stream_kernel<KC, false, true, false, false>, conv1x1_stream_kernel<KC, false, true, false, true>},
         {conv1x1_stream_kernel<KC, false, true, true, false>, conv1x1_stream_kernel<KC, false, true, true, true>}}};
    const kern_t kt = conv1x1_stream_kernel<KC, true, false, false, false>;
    static PpvDevOnce attr_once;
    if (attr_once.need()) {
        for (int i = 0; i < 8; ++i)
            PPV_ATTR(hipFuncSetAttribute((const void*)tab[i >> 2][(i >> 1) & 1][i & 1], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PPV_ATTR(hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done();
    }
    const unsigned grid = (unsigned)(tiles_m * n_splits);
    const kern_t k = out_f32 ? kt : tab[red_x ? 1 : 0][addend ? 1 : 0][mask_bits ? 1 : 0];
    k<<<grid, ST_NT, lds, stream>>>(X, Wt, out, stat_part, addend, mask_bits, g, n_splits, nspan, stat_rows, red_x, red_coef);
    return ppv_last_error();
}

int conv1x1_stream_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                          const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                          const ConvGeom& g, int out_f32, int stat_rows, hipStream_t stream) {
    (void)zero_page;
    if (g.Cs == 64) return stream_launch_k<64>(X, Wt, out, stat_part, addend, mask_bits, red_x, red_coef, g, out_f32, stat_rows, stream);
    if (g.Cs == 128) return stream_launch_k<128>(X, Wt, out, stat_part, addend, mask_bits, red_x, red_coef, g, out_f32, stat_rows, stream);
    return stream_launch_k<256>(X, Wt, out, stat_part, addend, mask_bits, red_x, red_coef, g, out_f32, stat_rows, stream);
}

}  // namespace ppv
