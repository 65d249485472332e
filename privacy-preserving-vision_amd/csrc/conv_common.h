// Device helpers shared by the NHWC bf16 convolution kernels (conv_gemm.hip, conv_stream.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <type_traits>
#include "ppv_common.h"

namespace ppv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

struct ConvGeom {
    int B, Hs, Ws, Cs;   // source tensor NHWC
    int Ho, Wo;          // output pixels per image (GEMM rows m = (b, ho, wo))
    int N;               // GEMM columns (output channels)
    int R, S;            // taps
    int a, off, sh;      // source step, tap offset (rows; columns too unless offw differs), log2(div)
    int offw;            // tap offset of the columns (rectangular kernels: 1 x 5 / 5 x 1 of RAFT's SepConvGRU)
    int flat;            // 1x1, unit step, no offset, same grid: GEMM row m IS source pixel m (no (b, ho, wo) decomposition)
    long M;              // B * Ho * Wo
    int chunked;         // layout experiment (flat launches, BK = 64): the source is [Cs / 64][M][64] instead of [M][Cs]
    int nt = 0;          // 1: the bf16 output tile is stored NON-TEMPORALLY (round 5: -0.43 ms per step, tools/ab_env01.sh PPV_NT_STORE; the
                         // outputs are 17-270 MB each and the consumer's first read comes a whole launch later)
    int add_lw = 0, add_lh = 0;   // != 0: the addend is COMPACT -- [B][Ho / 2][Wo / 2][N], the even-even pixels of the output map (every other
                                  // pixel adds zero), Wo = 1 << add_lw, Ho = 1 << add_lh: the projection shortcut's stride-2 data gradient
                                  // as trunk_plan.hip hands it to conv1's data gradient (conv_stream.hip only)
    unsigned long long* start_flag = nullptr;   // != null: the launch stores start_val there when its first workgroup starts (conv_signal_start): a fork
    unsigned long long start_val = 0;           // of the weight-gradient stream waits for it with hipStreamWaitValue64 -- no packet on the main chain
};

// The launch has STARTED, i.e. everything enqueued before it on its stream is complete and visible: tell a stream that waits for that
// (trunk_plan.hip fork_flag_*).  One relaxed system-scope store by one lane; the memory is HSA signal memory (uncached).
__device__ __forceinline__ void conv_signal_start(const ConvGeom& g) {
    if (g.start_flag != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(g.start_flag, g.start_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cooperative (grid-barrier) BatchNorm of the tile epilogue (conv_tile_epilogue.h, COOP; experiment of round 4): the
// launch's workgroups are all resident (one per CU), leave their statistics, cross ONE grid barrier and apply train-mode BatchNorm +
// ReLU to the tile they still hold in LDS.
struct CoopBn {
    unsigned* counter;             // [2] PRE-ZEROED: arrivals; [1] is set when a spin ran out (the launch then finished WITHOUT a barrier)
    const float *gamma, *beta;     // BatchNorm weight / bias [N]
    float *run_mean, *run_var;     // running statistics (may be null)
    float* coef;                   // [4][N] out: scale, shift, mean, invstd (what backward reads)
    void* y;                       // [M][N] bf16 out: relu(bn(raw))
    float count, momentum, eps;
};

__device__ __forceinline__ bf16_t f2bf(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
    return __builtin_bit_cast(float, (unsigned)h << 16);
}

// keep the bf16 lanes of v whose bit is set in b (bit k <-> lane k): the ReLU mask of the tensor the gradient flows into,
// as written by ppv_bn_act's pos_bits
__device__ __forceinline__ uint4 relu_mask8(uint4 v, unsigned b) {
    auto keep = [](unsigned two) { return ((two & 1u) ? 0xffffu : 0u) | ((two & 2u) ? 0xffff0000u : 0u); };
    v.x &= keep(b); v.y &= keep(b >> 2); v.z &= keep(b >> 4); v.w &= keep(b >> 6);
    return v;
}

// 16-byte global store, non-temporal when nt (wave-uniform) is set
__device__ __forceinline__ void store16_nt(void* p, uint4 v, int nt) {
    typedef unsigned nt_u32x4_t __attribute__((ext_vector_type(4)));
    const nt_u32x4_t nv = {v.x, v.y, v.z, v.w};
    if (nt) __builtin_nontemporal_store(nv, reinterpret_cast<nt_u32x4_t*>(p));
    else *reinterpret_cast<nt_u32x4_t*>(p) = nv;
}

#define GLDS16(gptr, lptr)                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                  \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// Progress words in LDS (loader / consumer waves without barriers: conv_stream.hip).  Read and written through
// address-space-3 pointers: through a generic `volatile unsigned*` hipcc emits flat_load / flat_store ... sc0 sc1 followed by
// s_waitcnt vmcnt(0) -- every poll then also waits for the wave's outstanding GLOBAL stores and loads.
typedef unsigned lds_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lds_u32x4_t lds_poll4(const volatile unsigned* p) {
    return *reinterpret_cast<const volatile __attribute__((address_space(3))) lds_u32x4_t*>(
        (const volatile __attribute__((address_space(3))) unsigned*)p);
}
__device__ __forceinline__ void lds_post(volatile unsigned* p, unsigned v) {
    *((volatile __attribute__((address_space(3))) unsigned*)p) = v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BatchNorm-backward sums taken in a data-gradient store loop (see conv_gemm.hip, RED): mask from x * scale + shift > 0 ...
__device__ __forceinline__ uint4 red_mask8(const uint4 gv, const uint4 xv, const float (&sc)[8], const float (&sh)[8]) {
    unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w};
    const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = __builtin_bit_cast(float, xw[i] << 16), x1 = __builtin_bit_cast(float, xw[i] & 0xffff0000u);
        if (!(__builtin_fmaf(x0, sc[2 * i], sh[2 * i]) > 0.f)) gw[i] &= 0xffff0000u;
        if (!(__builtin_fmaf(x1, sc[2 * i + 1], sh[2 * i + 1]) > 0.f)) gw[i] &= 0x0000ffffu;
    }
    return make_uint4(gw[0], gw[1], gw[2], gw[3]);
}
__device__ __forceinline__ void red_acc8(const uint4 gv, const uint4 xv, float (&ra)[8], float (&rb)[8]) {
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float g0 = __builtin_bit_cast(float, gw[i] << 16), g1 = __builtin_bit_cast(float, gw[i] & 0xffff0000u);
        const float x0 = __builtin_bit_cast(float, xw[i] << 16), x1 = __builtin_bit_cast(float, xw[i] & 0xffff0000u);
        ra[2 * i] += g0; ra[2 * i + 1] += g1;
        rb[2 * i] += g0 * x0; rb[2 * i + 1] += g1 * x1;
    }
}


// Output layout of the kernels that store straight from the accumulators.  The MFMA operands are SWAPPED (weight fragment in
// the A position, pixel fragment in the B position), so a lane's four accumulator registers are four consecutive MFMA rows =
// four weight rows, and LDS row l of a staged 64-row weight tile holds weight row nperm64(l) of its 64-column group:
//     l = ni * 16 + i  ->  (ni >> 1) * 32 + (i >> 2) * 8 + (ni & 1) * 4 + (i & 3)
// With that, lane (fq = lane >> 4, fr = lane & 15) holds for pixel row mi * 16 + fr the channels s * 32 + fq * 8 + e
// (acc[mi][2 s][0..3] -> e = 0..3, acc[mi][2 s + 1][0..3] -> e = 4..7): one 16-byte chunk per (mi, s); the four fq lanes of a
// row cover 64 contiguous bytes.  The tile leaves the registers as it is -- no LDS staging, no barrier in front of the stores --
// and the residual addend / raw-x / mask operands arrive in the same chunks.
__device__ __forceinline__ int nperm64(int l) {
    const int ni = (l >> 4) & 3, i = l & 15;
    return (l & ~63) + (ni >> 1) * 32 + (i >> 2) * 8 + (ni & 1) * 4 + (i & 3);
}
// sum over the 16 lanes of a DPP row (lanes fq * 16 .. fq * 16 + 15), result in every lane of the row
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_mov<0xB1>(x);      // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);      // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);     // row_half_mirror
    x += dpp_mov<0x140>(x);     // row_mirror
    return x;
}
__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
}
// two f32 -> one register of two bf16 (round to nearest even, NaN kept): ONE v_cvt_pk_bf16_f32 -- written as separate scalar
// casts + shift + or, hipcc emits three instructions per pair
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}

// PPV_STAMPS (diagnostic build only, tools/conv_timeline.py): one thread per role records s_memrealtime at its phase boundaries;
// the stamps stay in registers until the workgroup is done (a store inside the loops would sit in a counted vmcnt queue) and go
// to a debug buffer that nothing else reads: 16 slots per workgroup.
#ifdef PPV_STAMPS
extern __device__ unsigned long long* g_stamps;
#define PPV_STAMP_DECL unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PPV_STAMP(i) do { stamp_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PPV_STAMP_FLUSH(base, who) do { if (g_stamps && threadIdx.x == (who)) { for (int i_ = 0; i_ < 8; ++i_) g_stamps[(long)blockIdx.x * 16 + (base) + i_] = stamp_[i_]; } } while (0)
#else
#define PPV_STAMP_DECL do { } while (0)
#define PPV_STAMP(i) do { } while (0)
#define PPV_STAMP_FLUSH(base, who) do { } while (0)
#endif

// conv_halo.hip: 3x3 / stride 1 / pad 1 convolutions whose 256-row tiles are whole image rows (input tile + border resident in LDS).
bool conv3x3_halo_supported(const ConvGeom& g, int Cs, int div);
int conv3x3_halo_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                        const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                        const ConvGeom& g, int stat_rows, hipStream_t stream);
// The next ppv_conv_gemm / ppv_conv_gemm_red call of this thread takes its addend in the compact form of ConvGeom::add_lw (internal: the
// whole-trunk executor; the launch fails if it does not land on conv_stream.hip -- ask conv_addend_compact_supported first).
void conv_set_addend_compact(bool on);
// The next ppv_conv_gemm call of this thread stores its bf16 output temporally (0) / non-temporally (1) whatever PPV_NT_STORE says: an
// output whose consumer is the very next launch AND reads it through a latency-bound path (the BNIN halo kernel) wants it in the L2.
void conv_set_output_nt_once(int nt);
bool conv_addend_compact_supported(int B, int H, int W, int Cs, int N);
// conv_dgrad_s2.hip: data gradient of the stride-2 3x3 / 1x1 convolutions as four parity-class problems (bf16, no addend, no bit mask).
bool conv_dgrad_s2_supported(const ConvGeom& g, int Cs, int div);
int conv_dgrad_s2_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* zero_page, const bf16_t* red_x,
                         const float* red_coef, const ConvGeom& g, int stat_rows, hipStream_t stream);
// ... its 64-column tile on 8-wide maps (layer 4: four images per tile) ...
bool conv3x3_halo_n64_supported(const ConvGeom& g, int Cs, int div);
// BNIN (conv_halo.hip, round 6): train-mode BatchNorm + ReLU of the convolution's INPUT applied to the LDS-resident halo tile
struct HaloBn {
    const float* sums;     // [T][2][Cs] partial sums of the raw input
    int T;
    double inv_count, unbias;
    const float *gamma, *beta;
    float *run_mean, *run_var;
    float momentum, eps;
    float* coef;           // [4][Cs] out
    bf16_t* y_act;         // [M][Cs] out (null: not wanted)
};
bool conv3x3_halo_bnin_supported(const ConvGeom& g, int Cs);
int conv3x3_halo_bnin_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* zero_page, const ConvGeom& g,
                             int stat_rows, const HaloBn& bn, hipStream_t stream);
int conv3x3_halo_n64_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                            const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                            const ConvGeom& g, int stat_rows, hipStream_t stream);
// ... and the 64 -> 64-channel, 64-column form (layer 1): 256 pixels = four rows of one image, two workgroups per CU.
bool conv3x3_halo64_supported(const ConvGeom& g, int Cs, int div);
int conv3x3_halo64_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                          const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                          const ConvGeom& g, int stat_rows, hipStream_t stream);

// conv_stream.hip: 1x1 convolutions with at most 256 input channels (pixel rows resident in registers, weights streamed).
// Returns false when the problem is outside that kernel (the caller then takes a tiled kernel).
bool conv1x1_stream_supported(const ConvGeom& g, int Cs, int div);
int conv1x1_stream_launch(const bf16_t* X, const bf16_t* Wt, void* out, float* stat_part, const bf16_t* addend,
                          const unsigned char* mask_bits, const bf16_t* zero_page, const bf16_t* red_x, const float* red_coef,
                          const ConvGeom& g, int out_f32, int stat_rows, hipStream_t stream);

}  // namespace ppv
