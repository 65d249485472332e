// What does the K loop of the nine-tap 3x3 weight-gradient kernel (csrc/conv_wgrad_stem.hip conv_wgrad3x3_t9_kernel) cost without its
// global loads?  One 512-thread workgroup per CU runs STAGES "stages" on a resident LDS image: per stage and wave 8 + 36 + 8 transposed reads
// (ds_read_b64_tr_b16) and 72 v_mfma_f32_16x16x32_bf16, as in the kernel.  MODE 0: both; 1: reads only; 2: MFMAs only; 3: both, but the reads
// of a stage are issued as ONE burst in front of the MFMAs (fragments of all 18 units in registers is impossible in the kernel: this
// variant only shows what the LDS array delivers when nothing waits on single reads); 4: MODE 0 with plain ds_read_b64 instead of the
// transposing read (wrong operands, same traffic).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/tr_mfma_loop.hip -o /tmp/trloop && /tmp/trloop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int STAGES = 256;

__device__ __forceinline__ int trkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int trkey2(int h) { return ((h >> 1) & 1) | (((h >> 3) & 1) << 1); }
template <bool TR> __device__ __forceinline__ void rd(s16x4& v, unsigned a) {
    if (TR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a));
    else asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a));
}
__device__ __forceinline__ bf16x8 pack(const s16x4& lo, const s16x4& hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <class F, int... Is>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int OFF> __device__ __forceinline__ void tr_read_off(s16x4& v, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
// the kernel's loop after the address rework: immediates for everything lane-independent, one v_bfe + v_lshl_add per X read.
// SUB 0: reads + MFMAs; 1: reads only, NO waits inside the stage (LDS array throughput); 2: reads + MFMAs, no waits inside the stage
template <int SUB, int LOOK>
__global__ __launch_bounds__(512, 1) void k2(float* sink) {
    constexpr int log2W = 4, W = 16, WP = 18;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 36864 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int wn = wave >> 2, wc = wave & 3;
    f32x4 acc[9][4];
    for (int t = 0; t < 9; ++t) for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
    const int p8 = (lane & 3) * 8;
    unsigned abase[4];
    {
        const int r0 = 8 * (lane >> 4) + ((lane >> 2) & 3);
        for (int mi = 0; mi < 4; ++mi) abase[mi] = lds0 + r0 * 256 + (((wn * 4 + mi) ^ trkey(r0)) * 32) + p8;
    }
    unsigned bbase[2], swz[2][2];
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = kk * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
        const int hb = (kq >> log2W) * WP + (kq & (W - 1));
        bbase[kk] = lds0 + 64 * 256 + hb * 128 + p8;
        swz[kk][0] = swz[kk][1] = 0;
        for (int t = 0; t < 9; ++t) {
            const int h0 = hb + (t / 3) * WP + (t % 3);
            swz[kk][0] |= (unsigned)(wc ^ trkey2(h0)) << (2 * t);
            swz[kk][1] |= (unsigned)(wc ^ trkey2(h0 + 4)) << (2 * t);
        }
    }
    for (int st = 0; st < STAGES; ++st) {
        unsigned sw[2][2] = {{swz[0][0], swz[0][1]}, {swz[1][0], swz[1][1]}};
        asm volatile("" : "+v"(sw[0][0]), "+v"(sw[0][1]), "+v"(sw[1][0]), "+v"(sw[1][1]));
        s16x4 alo[2][4], ahi[2][4], blo[LOOK + 1], bhi[LOOK + 1];
        auto x_issue = [&](auto U_) {
            constexpr int u = decltype(U_)::value, kk = u / 9, t = u % 9, slot = u % (LOOK + 1);
            constexpr int off = ((t / 3) * WP + (t % 3)) * 128;
            tr_read_off<off>(blo[slot], bbase[kk] + (__builtin_amdgcn_ubfe(sw[kk][0], 2 * t, 2) << 5));
            tr_read_off<off + 512>(bhi[slot], bbase[kk] + (__builtin_amdgcn_ubfe(sw[kk][1], 2 * t, 2) << 5));
        };
        auto a_issue = [&](auto KK_, auto MI_) {
            constexpr int kk = decltype(KK_)::value, mi = decltype(MI_)::value;
            tr_read_off<kk * 8192>(alo[kk][mi], abase[mi]);
            tr_read_off<kk * 8192 + 1024>(ahi[kk][mi], abase[mi]);
        };
        static_for([&](auto MI_) { a_issue(std::integral_constant<int, 0>{}, MI_); }, std::make_integer_sequence<int, 4>{});
        static_for([&](auto V_) { x_issue(V_); }, std::make_integer_sequence<int, LOOK>{});
        static_for([&](auto U_) {
            constexpr int u = decltype(U_)::value, kk = u / 9, t = u % 9;
            constexpr int after = [] {
                int a = 0;
                if (u < LOOK) {
                    a += 2 * (LOOK - 1 - u);
                    for (int w = 0; w < u; ++w) a += (w + LOOK < 18 ? 2 : 0) + ((w >= 1 && w <= 4) ? 2 : 0);
                } else {
                    a += ((u - LOOK >= 1 && u - LOOK <= 4) ? 2 : 0);
                    for (int w = u - LOOK + 1; w < u; ++w) a += (w + LOOK < 18 ? 2 : 0) + ((w >= 1 && w <= 4) ? 2 : 0);
                }
                return a;
            }();
            __builtin_amdgcn_sched_barrier(0);
            if (SUB == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(after) : "memory");
            else if (u == 0 || u == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u + LOOK < 18) x_issue(std::integral_constant<int, u + LOOK>{});
            if constexpr (u >= 1 && u <= 4) a_issue(std::integral_constant<int, 1>{}, std::integral_constant<int, u - 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if (SUB == 1) {
                if (u == 17) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc[0][0][0] += (float)(blo[0][0] + bhi[1][1] + alo[0][0][0] + ahi[1][3][1] + blo[2][2]); }
            } else {
                const bf16x8 bf = pack(blo[u % (LOOK + 1)], bhi[u % (LOOK + 1)]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack(alo[kk][mi], ahi[kk][mi]), bf, acc[t][mi], 0, 0, 0);
            }
        }, std::make_integer_sequence<int, 18>{});
    }
    float v = 0.f;
    for (int t = 0; t < 9; ++t) for (int mi = 0; mi < 4; ++mi) v += acc[t][mi][0] + acc[t][mi][3];
    if (v == 12345.678f) sink[0] = v;
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* sink, int log2W) {
    constexpr bool TR = MODE != 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 36864 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int W = 1 << log2W, WP = W + 2;
    const int wn = wave >> 2, wc = wave & 3;
    const char* tg = smem;
    const char* tx = smem + 16384;
    f32x4 acc[9][4];
    for (int t = 0; t < 9; ++t) for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int hbase[2];
    for (int kk = 0; kk < 2; ++kk) {
        const int kq = kk * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
        hbase[kk] = (kq >> log2W) * WP + (kq & (W - 1));
    }
    const int p8 = (lane & 3) * 8;
    auto a_issue = [&](s16x4& lo, s16x4& hi, int k0, int colblock) {
        const int g = lane >> 4, q = (lane >> 2) & 3;
        const int r0 = k0 + 8 * g + q, r1 = r0 + 4;
        rd<TR>(lo, (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tg + r0 * 256 + (colblock ^ trkey(r0)) * 32 + p8));
        rd<TR>(hi, (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tg + r1 * 256 + (colblock ^ trkey(r1)) * 32 + p8));
    };
    auto x_issue = [&](s16x4& lo, s16x4& hi, int kk, int t) {
        const int r = t / 3, s = t - 3 * r;
        int hb = hbase[kk];
        asm volatile("" : "+v"(hb));
        const int h0 = hb + r * WP + s, h1 = h0 + 4;
        rd<TR>(lo, (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tx + h0 * 128 + ((wc ^ trkey2(h0)) * 32) + p8));
        rd<TR>(hi, (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tx + h1 * 128 + ((wc ^ trkey2(h1)) * 32) + p8));
    };
    auto lgkm_le = [](int n) {
        __builtin_amdgcn_sched_barrier(0);
        if (n <= 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (n == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int LOOK = 2;
    for (int st = 0; st < STAGES; ++st) {
        s16x4 alo[2][4], ahi[2][4], blo[LOOK + 1], bhi[LOOK + 1];
        if (MODE == 2) {
            for (int kk = 0; kk < 2; ++kk) for (int mi = 0; mi < 4; ++mi) { alo[kk][mi] = (s16x4){1, 2, 3, (short)st}; ahi[kk][mi] = (s16x4){4, 5, 6, (short)lane}; }
            for (int v = 0; v <= LOOK; ++v) { blo[v] = (s16x4){1, 1, 1, (short)st}; bhi[v] = (s16x4){2, 2, 2, 2}; }
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                const int kk = u / 9, t = u % 9;
                const bf16x8 bf = pack(blo[u % (LOOK + 1)], bhi[u % (LOOK + 1)]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack(alo[kk][mi], ahi[kk][mi]), bf, acc[t][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a_issue(alo[0][mi], ahi[0][mi], 0, wn * 4 + mi);
#pragma unroll
        for (int v = 0; v < LOOK; ++v) x_issue(blo[v], bhi[v], 0, v);
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int kk = u / 9, t = u % 9;
            int after = 0;
            if (u < LOOK) {
                after += 2 * (LOOK - 1 - u);
                for (int w = 0; w < u; ++w) after += (w + LOOK < 18 ? 2 : 0) + ((w >= 1 && w <= 4) ? 2 : 0);
            } else {
                after += ((u - LOOK >= 1 && u - LOOK <= 4) ? 2 : 0);
                for (int w = u - LOOK + 1; w < u; ++w) after += (w + LOOK < 18 ? 2 : 0) + ((w >= 1 && w <= 4) ? 2 : 0);
            }
            if (MODE != 3) lgkm_le(after);
            else if (u == 0) lgkm_le(0);
            if (u + LOOK < 18) x_issue(blo[(u + LOOK) % (LOOK + 1)], bhi[(u + LOOK) % (LOOK + 1)], (u + LOOK) / 9, (u + LOOK) % 9);
            if (u >= 1 && u <= 4) a_issue(alo[1][u - 1], ahi[1][u - 1], 32, wn * 4 + (u - 1));
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1) {                                  // keep the fragments alive without the matrix pipe
                if (u == 17) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc[0][0][0] += (float)(blo[0][0] + bhi[1][1] + alo[0][0][0] + ahi[1][3][1] + blo[2][2]); }
                continue;
            }
            const bf16x8 bf = pack(blo[u % (LOOK + 1)], bhi[u % (LOOK + 1)]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack(alo[kk][mi], ahi[kk][mi]), bf, acc[t][mi], 0, 0, 0);
        }
    }
    float v = 0.f;
    for (int t = 0; t < 9; ++t) for (int mi = 0; mi < 4; ++mi) v += acc[t][mi][0] + acc[t][mi][3];
    if (v == 12345.678f) sink[0] = v;
}

int main() {
    float* sink; (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[5] = {"reads + MFMAs (kernel's loop)", "reads only", "MFMAs only", "reads + MFMAs, one wait per stage", "ds_read_b64 (no transpose) + MFMAs"};
    const int lds = 36864;
    for (int mode = 0; mode < 5; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<256, 512, lds>>>(sink, 4);
            if (mode == 1) k<1><<<256, 512, lds>>>(sink, 4);
            if (mode == 2) k<2><<<256, 512, lds>>>(sink, 4);
            if (mode == 3) k<3><<<256, 512, lds>>>(sink, 4);
            if (mode == 4) k<4><<<256, 512, lds>>>(sink, 4);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-45s %.3f us per stage (72 MFMAs + 52 reads per wave, 8 waves)\n", names[mode], ms * 1e3 / STAGES);
        }
    const char* n2[5] = {"immediate offsets: reads + MFMAs, LOOK 3", "immediate offsets: reads only, no waits", "immediate offsets: reads + MFMAs, no waits", "immediate offsets: reads + MFMAs, LOOK 2", "immediate offsets: reads + MFMAs, LOOK 4"};
    for (int mode = 0; mode < 5; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k2<0, 3><<<256, 512, lds>>>(sink);
            if (mode == 1) k2<1, 3><<<256, 512, lds>>>(sink);
            if (mode == 2) k2<2, 3><<<256, 512, lds>>>(sink);
            if (mode == 3) k2<0, 2><<<256, 512, lds>>>(sink);
            if (mode == 4) k2<0, 4><<<256, 512, lds>>>(sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-45s %.3f us per stage\n", n2[mode], ms * 1e3 / STAGES);
        }
    return 0;
}
