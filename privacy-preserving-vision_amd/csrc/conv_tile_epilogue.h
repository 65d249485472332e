// The epilogue shared by the tiled NHWC bf16 convolution kernels (conv_gemm.hip: conv_gemm_pipe_kernel; conv_halo.hip): accumulators
// -> (+ addend) -> bf16 -> LDS -> coalesced 16-byte stores, with the BatchNorm forward statistics (matrix pipe) or the BatchNorm
// backward sums (RED) and the ReLU bit mask folded into the store loop.
#pragma once
#include "conv_common.h"

namespace ppv {

// ----------------------------------------------------------------------------- shared epilogue of the 256/128-row tiled kernels
// acc: the wave's MI x NI accumulator tiles of a BM x BN output tile whose rows are GEMM rows m0 .. m0 + BM - 1 and columns n0 ..;
// smem: the workgroup's whole dynamic LDS (LDS_TOTAL bytes, free: every wave is past its last read of the K loop's stages).
// RM: tile row m -> row of the tensors in memory (output, mask, BatchNorm input).  Identity for every convolution but the stride-2 data
// gradients of conv_dgrad_s2.hip, whose tiles hold the pixels of ONE parity class of the gradient map (single-pass bf16 paths only).
struct RowIdentity {
    static constexpr bool identity = true;
    __device__ __forceinline__ long operator()(long m) const { return m; }
};
template <int BM, int BN, int LDS_TOTAL, int WGPCU, bool OUT_F32, bool RED, int MI, int NI, bool COOP = false, class RM = RowIdentity>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[MI][NI], char* smem, void* __restrict__ Out, float* __restrict__ stat_part,
                                              const bf16_t* __restrict__ addend, const unsigned char* __restrict__ mask_bits,
                                              const ConvGeom& g, int tile_m, int stat_rows, const bf16_t* __restrict__ red_x,
                                              const float* __restrict__ red_coef, long m0, int n0, const CoopBn& cb
#ifdef PPV_STAMPS
                                              , unsigned long long* stamp_
#endif
                                              , const RM rm = RM{}) {
    static_assert(RM::identity || (!OUT_F32 && !COOP), "row maps: bf16 store paths");
    constexpr int NT = BM * 2, NWAVE = NT / 64;
    constexpr int WN = BN / 64 > 0 ? BN / 64 : 1, WM = NWAVE / WN;
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    static_assert(MI * 16 == WROWS && NI * 16 == WCOLS, "accumulator shape");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
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
    constexpr int LDO = BN * 2 + 32;                           // row stride = 8 words mod 64 (packed tile write below)
    char* sO = smem;
    float* sStat = reinterpret_cast<float*>(smem + BM * LDO);
    constexpr int CPR = BN / 8;
    static_assert(!RED || ((CPR == 32 || CPR == 16 || CPR == 8) && NT % CPR == 0 && !OUT_F32), "RED: 64-, 128- or 256-column bf16 tiles");
    float ra[8], rb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ra[k] = rb[k] = 0.f;
    // every thread's chunk column ch = tid % CPR is the same in all its store iterations: fold the rows of a wave by
    // shuffles, the waves through LDS (the staging area is free once the caller has passed a barrier), then one atomic per column
    auto red_finish = [&]() {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (CPR == 8) { ra[k] += __shfl_xor(ra[k], 8, 64); rb[k] += __shfl_xor(rb[k], 8, 64); }
            if (CPR <= 16) { ra[k] += __shfl_xor(ra[k], 16, 64); rb[k] += __shfl_xor(rb[k], 16, 64); }
            ra[k] += __shfl_xor(ra[k], 32, 64); rb[k] += __shfl_xor(rb[k], 32, 64);
        }
        float* sRed = reinterpret_cast<float*>(smem);
        if (lane < CPR) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sRed[(wave * 2 + 0) * BN + lane * 8 + k] = ra[k];
                sRed[(wave * 2 + 1) * BN + lane * 8 + k] = rb[k];
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += NT) {
            const int which = t / BN, col = t % BN;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) v += sRed[(w * 2 + which) * BN + col];
            atomicAdd(&stat_part[((long)(tile_m % stat_rows) * 2 + which) * g.N + n0 + col], v);
        }
    };
    if (addend) {
        // the addend tile comes in as whole 16-byte chunks (coalesced), is parked in LDS behind the output staging
        // area and added fragment-wise in f32: one rounding of (acc + addend), no 2-byte global gathers
        char* sAdd = smem + BM * LDO + 2048;
        if constexpr (2 * BM * LDO + 2048 > LDS_TOTAL && WM % 2 != 0) {
            return;                                             // host never pairs this tile shape with an addend
        } else if constexpr (2 * BM * LDO + 2048 > LDS_TOTAL) {
            // small ring (BK = 32): the output tile and the addend tile are processed in two 128-row halves
            static_assert(BM * LDO + 2048 <= LDS_TOTAL, "two-pass epilogue geometry");
            constexpr int HR = BM / 2;                          // rows per pass
            char* sOh = smem;
            char* sAh = smem + HR * LDO;
            float* sSt = reinterpret_cast<float*>(smem + 2 * HR * LDO);
            bf16_t* outp = reinterpret_cast<bf16_t*>(Out);
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                // the mask bytes of this pass's store loop are requested first: their latency hides behind the addend round trip
                constexpr int SIT = (HR * CPR + NT - 1) / NT;
                unsigned char mb[SIT];
#pragma unroll
                for (int it = 0; it < SIT; ++it) {
                    const int idx = it * NT + tid;
                    const int row = idx / CPR, ch = idx % CPR;
                    const long m = m0 + pass * HR + row;
                    mb[it] = 0xff;
                    if (idx < HR * CPR && m < g.M) {
                        if (mask_bits) mb[it] = mask_bits[(m * g.N + n0) / 8 + ch];
                    }
                }
#pragma unroll
                for (int it = 0; it < (HR * CPR + NT - 1) / NT; ++it) {
                    const int idx = it * NT + tid;
                    const int row = idx / CPR, ch = idx % CPR;
                    const long m = m0 + pass * HR + row;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (idx < HR * CPR && m < g.M) v = *reinterpret_cast<const uint4*>(addend + m * g.N + n0 + ch * 8);
                    if (idx < HR * CPR) *reinterpret_cast<uint4*>(sAh + row * LDO + ch * 16) = v;
                }
                __syncthreads();
                if (wm / (WM / 2) == pass) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        float s1 = 0.f, s2 = 0.f;
                        const int col = wn * WCOLS + ni * 16 + fr;
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int rowl = (wm % (WM / 2)) * WROWS + mi * 16 + fq * 4 + j;
                                const float a = acc[mi][ni][j] + bf2f(*reinterpret_cast<const bf16_t*>(sAh + rowl * LDO + col * 2));
                                const bf16_t h = f2bf(a);
                                const float v = bf2f(h);
                                s1 += v;
                                s2 += v * v;
                                *reinterpret_cast<bf16_t*>(sOh + rowl * LDO + col * 2) = h;
                            }
                        if (!RED && stat_part) {
                            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                            if (fq == 0) {
                                sSt[(wm * 2 + 0) * BN + col] = s1;
                                sSt[(wm * 2 + 1) * BN + col] = s2;
                            }
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < (HR * CPR + NT - 1) / NT; ++it) {
                    const int idx = it * NT + tid;
                    const int row = idx / CPR, ch = idx % CPR;
                    const long m = m0 + pass * HR + row;
                    if (idx < HR * CPR && m < g.M)
                    {
                        uint4 v = *reinterpret_cast<const uint4*>(sOh + row * LDO + ch * 16);
                        if (mask_bits) v = relu_mask8(v, mb[it]);
                        if constexpr (RED) red_acc8(v, *reinterpret_cast<const uint4*>(red_x + m * g.N + n0 + ch * 8), ra, rb);
                        store16_nt(outp + m * g.N + n0 + ch * 8, v, g.nt & 1);
                    }
                }
                __syncthreads();
            }
            if constexpr (RED) {
                red_finish();
            } else if (stat_part && tid < 2 * BN) {
                const int which = tid / BN, col = tid % BN;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) v += sSt[(w * 2 + which) * BN + col];
                atomicAdd(&stat_part[((long)(tile_m % stat_rows) * 2 + which) * g.N + n0 + col], v);
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < (BM * CPR + NT - 1) / NT; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / CPR, ch = idx % CPR;
            const long m = m0 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < BM * CPR && m < g.M) v = *reinterpret_cast<const uint4*>(addend + m * g.N + n0 + ch * 8);
            if (idx < BM * CPR) *reinterpret_cast<uint4*>(sAdd + row * LDO + ch * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * WROWS + mi * 16 + fq * 4 + j, col = wn * WCOLS + ni * 16 + fr;
                    acc[mi][ni][j] += bf2f(*reinterpret_cast<const bf16_t*>(sAdd + row * LDO + col * 2));
                }
    }
    // accumulators -> bf16 tile in LDS.  The MFMA result layout gives a lane ONE column (fr) and four consecutive rows per tile;
    // written as it stands that is 64 two-byte LDS writes per lane, two lanes to a bank word (measured: 2.0 us of a 256 x 128
    // tile's 3.4-us epilogue).  Here a lane packs its rows pairwise (v_cvt_pk_bf16_f32: [row j | row j+1]), swaps with its
    // column neighbour (DPP quad_perm [1,0,3,2]) and one v_perm_b32 makes a full word of two ADJACENT columns: even lanes take
    // row j, odd lanes row j+1 -- 32 conflict-free ds_write_b32 per lane (row stride = 8 words mod 64).  The BatchNorm sums come
    // off the matrix pipe from the same packed registers: ones x P = column sums, P^T x P = Gram matrix whose diagonal is the
    // sum of squares (the k index of an MFMA operand may be any permutation of the 32 rows of two tiles).
    static_assert(MI % 2 == 0, "row tiles are taken in pairs");
    {
        const int par = fr & 1;
        const unsigned sel = par ? 0x03020706u : 0x05040100u;
        char* lbase = sO + (wm * WROWS + fq * 4 + par) * LDO + (wn * WCOLS + (fr & ~1)) * 2;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        const u32x4_t ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mp = 0; mp < MI / 2; ++mp) {
                unsigned pk[4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        const int mi = mp * 2 + h;
                        const unsigned own = pack2(acc[mi][ni][jp * 2], acc[mi][ni][jp * 2 + 1]);
                        pk[h * 2 + jp] = own;
                        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xf, 0xf, true);
                        *reinterpret_cast<unsigned*>(lbase + (mi * 16 + jp * 2) * LDO + ni * 32) = __builtin_amdgcn_perm(nb, own, sel);
                    }
                if (!RED && stat_part) {
                    const u32x4_t pu = {pk[0], pk[1], pk[2], pk[3]};
                    const bf16x8 P = __builtin_bit_cast(bf16x8, pu);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, P, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(P, P, c2, 0, 0, 0);
                }
            }
            if (!RED && stat_part) {
                const int col = wn * WCOLS + ni * 16 + fr;
                const int d = fr & 3;
                const float q = d == 0 ? c2[0] : d == 1 ? c2[1] : d == 2 ? c2[2] : c2[3];
                if (fq == 0) sStat[(wm * 2 + 0) * BN + col] = c1[0];
                if (fq == (fr >> 2)) sStat[(wm * 2 + 1) * BN + col] = q;
            }
        }
    }
    PPV_STAMP(7);
    __syncthreads();
    if (!RED && stat_part && tid < 2 * BN) {
        const int which = tid / BN, col = tid % BN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) v += sStat[(w * 2 + which) * BN + col];
        atomicAdd(&stat_part[((long)(tile_m % stat_rows) * 2 + which) * g.N + n0 + col], v);
    }
    bf16_t* out = reinterpret_cast<bf16_t*>(Out);
    float rsc[8], rsh[8];
    if constexpr (RED) {
        if (red_coef) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                rsc[k] = red_coef[n0 + (tid % CPR) * 8 + k];
                rsh[k] = red_coef[g.N + n0 + (tid % CPR) * 8 + k];
            }
        }
    }
    // store loop.  Whole tiles (all but the last row tile of a launch) take the three-phase form: every global read this loop
    // needs (mask bytes, BatchNorm input rows) is requested first, then every LDS chunk, then the stores -- one latency of each
    // kind per tile instead of one per 16-byte chunk (the per-chunk form measured 1.0-2.8 us per tile: eight dependent
    // LDS round trips behind 64-bit address products and row predicates).
    static_assert(NT % CPR == 0 && (BM * CPR) % NT == 0, "a thread keeps its chunk column");
    constexpr int RSTEP = NT / CPR, SITERS = BM / RSTEP;
    if (m0 + BM <= g.M) {
        const int row0 = tid / CPR, ch = tid % CPR;
        const long e0 = (m0 + row0) * g.N + n0 + ch * 8;
        const long estep = (long)RSTEP * g.N;
        auto eoff = [&](int k) __attribute__((always_inline)) -> long {
            if constexpr (RM::identity) return e0 + k * estep;
            else return rm(m0 + row0 + (long)k * RSTEP) * g.N + n0 + ch * 8;
        };
        constexpr int GRP = RED ? (WGPCU > 1 ? 2 : 4) : SITERS;            // RED: rounds of a few chunks (register budget)
#pragma unroll
        for (int i0 = 0; i0 < SITERS; i0 += GRP) {
            unsigned char mbv[GRP];
            uint4 xv[RED ? GRP : 1];
            if (mask_bits) {
#pragma unroll
                for (int it = 0; it < GRP; ++it) mbv[it] = mask_bits[eoff(i0 + it) >> 3];
            }
            if constexpr (RED) {
#pragma unroll
                for (int it = 0; it < GRP; ++it) xv[it] = *reinterpret_cast<const uint4*>(red_x + eoff(i0 + it));
            }
            uint4 v[GRP];
#pragma unroll
            for (int it = 0; it < GRP; ++it) v[it] = *reinterpret_cast<const uint4*>(sO + (row0 + (i0 + it) * RSTEP) * LDO + ch * 16);
            if (mask_bits) {
#pragma unroll
                for (int it = 0; it < GRP; ++it) v[it] = relu_mask8(v[it], mbv[it]);
            }
#pragma unroll
            for (int it = 0; it < GRP; ++it) {
                if constexpr (RED) {
                    if (red_coef) v[it] = red_mask8(v[it], xv[it], rsc, rsh);
                    red_acc8(v[it], xv[it], ra, rb);
                }
                store16_nt(out + eoff(i0 + it), v[it], g.nt & 1);
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < SITERS; ++it) {
            const int idx = it * NT + tid;
            const int row = idx / CPR, ch = idx % CPR;
            if (m0 + row < g.M) {
                const long m = rm(m0 + row);
                uint4 v = *reinterpret_cast<const uint4*>(sO + row * LDO + ch * 16);
                if (mask_bits) v = relu_mask8(v, mask_bits[(m * g.N + n0) / 8 + ch]);
                if constexpr (RED) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(red_x + m * g.N + n0 + ch * 8);
                    if (red_coef) v = red_mask8(v, xv, rsc, rsh);
                    red_acc8(v, xv, ra, rb);
                }
                store16_nt(out + m * g.N + n0 + ch * 8, v, g.nt & 1);
            }
        }
    }
#ifdef PPV_STAMPS
    PPV_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PPV_STAMP(6);
    PPV_STAMP_FLUSH(0, 0);
#endif
    if constexpr (RED) {
        __syncthreads();                                        // every chunk of the staged tile has been read
        red_finish();
    }
    if constexpr (COOP && !RED && !OUT_F32) {
        // ---- grid barrier, then train-mode BatchNorm + ReLU on the tile that is still staged in LDS (sO): the raw tensor has been
        // stored above (backward reads it), the activation is stored here -- the element-wise launch that would re-read the raw
        // tensor does not exist.  Residency: the host launches this form only with <= one workgroup per CU and checks the grid
        // against the occupancy query; the spin is bounded all the same (counter[1] reports a barrier that never completed).
        static_assert(LDS_TOTAL >= BM * LDO + 8192 + 2 * BN * 4, "room for the coefficient rows behind the statistics");
        float* sCo = reinterpret_cast<float*>(smem + BM * LDO + 8192);             // [2][BN]: scale, shift
        // this thread's statistics atomic must be at memory before the arrival; the SITERS tile stores issued after it may still be in
        // flight (vmcnt retires in order): waiting for them too put a full store round trip in front of every arrival
        if (m0 + BM <= g.M) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SITERS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cb.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(cb.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 19)) {
                    __hip_atomic_store(cb.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (tid < BN) {
            // totals of this column: the adds that built them executed at the memory side (nothing of them stays in an L2), these are
            // the first loads of those lines in this launch, and agent-scope loads bypass the CU's own L1
            const int col = n0 + tid;
            float s1 = 0.f, s2 = 0.f;
            for (int r = 0; r < stat_rows; ++r) {
                s1 += __hip_atomic_load(&stat_part[((long)r * 2 + 0) * g.N + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(&stat_part[((long)r * 2 + 1) * g.N + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const double mean = (double)s1 / (double)cb.count;
            double var = (double)s2 / (double)cb.count - mean * mean;
            if (var < 0) var = 0;
            const float invstd = (float)(1.0 / sqrt(var + (double)cb.eps));
            const float scale = cb.gamma[col] * invstd, shift = cb.beta[col] - (float)mean * scale;
            sCo[tid] = scale;
            sCo[BN + tid] = shift;
            if (tile_m == 0) {                                                      // one row tile publishes what backward / the module reads
                cb.coef[col] = scale; cb.coef[g.N + col] = shift; cb.coef[2 * g.N + col] = (float)mean; cb.coef[3 * g.N + col] = invstd;
                if (cb.run_mean) cb.run_mean[col] = (1.f - cb.momentum) * cb.run_mean[col] + cb.momentum * (float)mean;
                if (cb.run_var) {
                    const double unb = cb.count > 1.f ? var * (double)cb.count / ((double)cb.count - 1.0) : var;
                    cb.run_var[col] = (1.f - cb.momentum) * cb.run_var[col] + cb.momentum * (float)unb;
                }
            }
        }
        __syncthreads();
        {
            bf16_t* yout = reinterpret_cast<bf16_t*>(cb.y);
            const int row0 = tid / CPR, ch = tid % CPR;
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { sc[k] = sCo[ch * 8 + k]; sh[k] = sCo[BN + ch * 8 + k]; }
#pragma unroll
            for (int it = 0; it < SITERS; ++it) {
                const int row = row0 + it * RSTEP;
                const long m = m0 + row;
                if (m < g.M) {
                    const uint4 v = *reinterpret_cast<const uint4*>(sO + row * LDO + ch * 16);
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = fmaxf(__builtin_fmaf(f[k], sc[k], sh[k]), 0.f);
                    *reinterpret_cast<uint4*>(yout + m * g.N + n0 + ch * 8) =
                        make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
                }
            }
        }
    }
}

}  // namespace ppv
