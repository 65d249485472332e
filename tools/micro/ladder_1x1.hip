// VERDICT r5 task 3: where does the 1x1 forward class lose the LDS-DMA rate of tools/micro/glds_pattern.hip (5.3-5.6 TB/s)?  A LADDER of
// kernels, each adding ONE property of conv_gemm_pipe_kernel<256,128,3,BK,WG/CU> (csrc/conv_gemm.hip) on the two layer-3 shapes
//   S1: 1024 -> 256 @16x16, B = 128 (M = 32768, K = 1024, N = 256): BK = 64, one 147-KB workgroup per CU, grid 128 x 2, 16 K-steps
//   S2: 256 -> 1024 @16x16          (M = 32768, K =  256, N = 1024): BK = 32, two 74-KB workgroups per CU, grid 128 x 8, 8 K-steps
// with no MFMA and (where it matters) no fragment reads -- only the memory system's part of the launch:
//   rung 0  the published pattern: 1024 workgroups x 16 steps x 32 KB, private tiles, 96 KB of LDS (glds_pattern.hip mode 0)
//   rung 1  the kernel's A operand alone: its grid, LDS footprint, stage size, ring depth, counted waits; PRIVATE rows per workgroup
//   rung 2  ... rows SHARED by the tiles_n column tiles of an m-tile, block -> tile map of the kernel (XCD-contiguous ranges)
//   rung 3  ... + the weight slices (B operand: BN rows x BK per step out of the L2-resident filter)
//   rung 4  ... + the epilogue: the 256 x 128 bf16 output tile stored from LDS (64 KB per workgroup) after the loop
//   rung 5  ... + one ds_read_b128 per lane and step of the landed stage (the fragment reads' place in the barrier structure)
// The real kernel's own modes (loads only / K loop only / whole) come from the library (tools/ladder_1x1.py, PPV_CONV_DEBUG).
// Operands rotate through NB buffer sets (> 256 MB Infinity Cache): every launch reads cold HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/ladder_1x1.hip -o /tmp/ladder && /tmp/ladder
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GLDS16(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct P {
    const char* A; const char* W; char* O; int* sink;
    int K;            // channels of the source rows (row pitch = 2 K bytes)
    int N;            // output channels (row pitch of the output = 2 N bytes)
    int tiles_n;      // column tiles per m-tile
    int nk;           // K-steps
    int flags;        // 1: stage A, 2: stage B, 4: A rows shared by the column tiles (kernel's map), 8: epilogue store, 16: ds_read per step
};

template <int BK, int WGPCU>
__global__ __launch_bounds__(512, 2 * WGPCU) void ladder(P p) {
    constexpr int BM = 256, BN = 128, NS = 3, ROWB = BK * 2, CH = BK / 8, RPI = 1024 / ROWB, RPR = 8 * RPI;
    constexpr int ASLOTS = BM / RPR, BSLOTS = BN / RPR, L = ASLOTS + BSLOTS, A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const bool shared = p.flags & 4;
    const int tile_m = shared ? bid / p.tiles_n : bid, tile_n = shared ? bid % p.tiles_n : bid % p.tiles_n;
    const int rl = lane / CH, pc = lane % CH;
    const char* a_ptr[ASLOTS];
    const char* w_ptr[BSLOTS];
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) a_ptr[i] = p.A + ((long)tile_m * BM + i * RPR + wave * RPI + rl) * p.K * 2 + pc * 16;
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) w_ptr[i] = p.W + ((long)tile_n * BN + i * RPR + wave * RPI + rl) * p.K * 2 + pc * 16;
    const bool doA = p.flags & 1, doB = p.flags & 2;
    auto stage = [&](int buf) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            if (doA) GLDS16(a_ptr[i], sb + (i * 8 + wave) * 1024);
            else GLDS16(w_ptr[0], sb + (i * 8 + wave) * 1024);      // keeps the instruction count (counted waits) with L2-resident bytes
            a_ptr[i] += ROWB;
        }
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i) {
            if (doB) GLDS16(w_ptr[i], sb + A_BYTES + (i * 8 + wave) * 1024);
            else GLDS16(w_ptr[i] - (p.flags & 2 ? 0 : 0), sb + A_BYTES + (i * 8 + wave) * 1024);
            if (doB) w_ptr[i] += ROWB;
        }
    };
    const int nk = p.nk;
    stage(0);
    if (nk > 1) stage(1);
    int acc = 0;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) wait_vm<L>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nk) stage((t + 2) % NS);
        if (p.flags & 16) acc += *reinterpret_cast<const int*>(smem + (t % NS) * STAGE + tid * 16);
    }
    if (p.flags & 8) {
        __syncthreads();
        // the kernel's store loop: the bf16 tile [256][128] leaves LDS in 16-byte pieces, a row = 256 contiguous bytes
        constexpr int CPR = BN * 2 / 16;
#pragma unroll
        for (int it = 0; it < BM * CPR / 512; ++it) {
            const int idx = it * 512 + tid, row = idx / CPR, ch = idx % CPR;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 272 + ch * 16);
            typedef unsigned nt4 __attribute__((ext_vector_type(4)));
            const nt4 nv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<nt4*>(p.O + ((long)(shared ? tile_m : tile_m / p.tiles_n) * BM + row) * p.N * 2 + tile_n * BN * 2 + ch * 16));
        }
    }
    if (acc == 0x7fffffff) p.sink[0] = acc;
}

// rung 0: glds_pattern.hip mode 0 verbatim (1024 x 16 x 32 KB, private 2-KiB-strided rows)
__global__ __launch_bounds__(512) void pattern0(const char* __restrict__ src, long tile_stride, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* base = src + (long)blockIdx.x * tile_stride;
    auto stage = [&](int buf, int kstep) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 8 + wave;
            GLDS16(base + (long)(q * 8 + (lane >> 3)) * 2048 + kstep * 128 + (lane & 7) * 16, smem + buf * 32768 + q * 1024);
        }
    };
    stage(0, 0); stage(1, 1);
    int acc = 0;
    for (int t = 0; t < 16; ++t) {
        if (t + 2 < 16) wait_vm<4>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + 2 < 16) stage((t + 2) % 3, t + 2);
        acc += *reinterpret_cast<const int*>(smem + (t % 3) * 32768 + threadIdx.x * 16);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

static float time_us(const std::function<void(int)>& launch, int nb, int reps = 4) {
    for (int i = 0; i < nb; ++i) launch(i);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> v;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < reps * nb; ++i) launch(i % nb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        v.push_back(ms * 1e3f / (reps * nb));
    }
    std::sort(v.begin(), v.end());
    return v[1];
}

#include <functional>
int main() {
    const int M = 32768, NB = 6;
    const long big = (long)M * 1024 * 2;                      // 67 MB: the 1024-channel tensor (S1's source, S2's result)
    const long small = (long)M * 256 * 2;                     // 16.8 MB
    std::vector<char*> bigs(NB), smalls(NB), priv(2);
    for (int i = 0; i < NB; ++i) { hipMalloc(&bigs[i], big); hipMalloc(&smalls[i], small); hipMemset(bigs[i], 1, big); hipMemset(smalls[i], 1, small); }
    char* pat; hipMalloc(&pat, 512L << 20); hipMemset(pat, 1, 512L << 20);
    char* wt; hipMalloc(&wt, 1024 * 256 * 2); hipMemset(wt, 1, 1024 * 256 * 2);
    int* sink; hipMalloc(&sink, 4);
    const int lds64 = 3 * (256 + 128) * 128, lds32 = 3 * (256 + 128) * 64;
    hipFuncSetAttribute((const void*)ladder<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds64);
    hipFuncSetAttribute((const void*)ladder<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds32);
    hipFuncSetAttribute((const void*)pattern0, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    printf("{\n");
    {
        const float us = time_us([&](int) { pattern0<<<1024, 512, 98304>>>(pat, 256L * 2048, sink); }, 1, 6);
        printf(" \"rung0_pattern_1024wg_x_512KB_private\": {\"us\": %.1f, \"staged_MB\": 512, \"TBps_staged\": %.2f},\n", us, 512.0 * 1.048576 / us);
    }
    struct Shape { const char* name; int K, N, tiles_n, nk, bk; double hbm_in_MB, staged_MB, out_MB; } shapes[2] = {
        {"S1_1024_to_256", 1024, 256, 2, 16, 64, 67.1 + 0.5, 256 * (256 + 128) * 2048.0 / 1e6, 16.8},
        {"S2_256_to_1024", 256, 1024, 8, 8, 32, 16.8 + 0.5, 1024 * (256 + 128) * 512.0 / 1e6, 67.1}};
    const struct { const char* name; int flags; } rungs[] = {
        {"rung1_A_private", 1}, {"rung2_A_shared", 1 | 4}, {"rung3_A_shared_plus_W", 1 | 2 | 4}, {"rung4_plus_epilogue_store", 1 | 2 | 4 | 8},
        {"rung5_plus_lds_read_per_step", 1 | 2 | 4 | 8 | 16}, {"rungX_W_only", 2 | 4}, {"rungY_epilogue_store_only", 4 | 8}};
    for (int s = 0; s < 2; ++s) {
        const Shape& sh = shapes[s];
        printf(" \"%s\": {\n", sh.name);
        for (const auto& r : rungs) {
            auto launch = [&](int i) {
                P p;
                p.A = s == 0 ? bigs[i] : smalls[i];
                p.W = wt; p.O = s == 0 ? smalls[i] : bigs[i]; p.sink = sink; p.K = sh.K; p.N = sh.N; p.tiles_n = sh.tiles_n; p.nk = sh.nk; p.flags = r.flags;
                const int grid = 128 * sh.tiles_n;
                // private rows: the source must hold grid x 256 rows -- S1: 256 tiles x 512 KB = 134 MB > one 67-MB tensor: walk two of them
                if (!(r.flags & 4) && s == 0) p.A = bigs[i];
                if (sh.bk == 64) ladder<64, 1><<<grid, 512, lds64>>>(p); else ladder<32, 2><<<grid, 512, lds32>>>(p);
            };
            if (!(r.flags & 4)) {
                // private variant: every workgroup its own 256 rows -> rows beyond one tensor; allocate a source of grid x 256 rows once
                static char* privS[2] = {nullptr, nullptr};
                if (!privS[s]) { hipMalloc(&privS[s], (long)128 * sh.tiles_n * 256 * sh.K * 2 * 2); hipMemset(privS[s], 1, (long)128 * sh.tiles_n * 256 * sh.K * 2 * 2); }
                auto launch_p = [&](int i) {
                    P p;
                    p.A = privS[s] + (long)(i & 1) * 128 * sh.tiles_n * 256 * sh.K * 2;
                    p.W = wt; p.O = bigs[0]; p.sink = sink; p.K = sh.K; p.N = sh.N; p.tiles_n = sh.tiles_n; p.nk = sh.nk; p.flags = r.flags;
                    const int grid = 128 * sh.tiles_n;
                    if (sh.bk == 64) ladder<64, 1><<<grid, 512, lds64>>>(p); else ladder<32, 2><<<grid, 512, lds32>>>(p);
                };
                const float us = time_us(launch_p, 2, 8);
                printf("  \"%s\": {\"us\": %.1f, \"staged_MB\": %.0f, \"TBps_staged\": %.2f},\n", r.name, us, 128 * sh.tiles_n * 256 * sh.K * 2 / 1e6,
                       128.0 * sh.tiles_n * 256 * sh.K * 2 / 1e6 / us);
                continue;
            }
            const float us = time_us(launch, NB);
            const double hbm = ((r.flags & 1) ? sh.hbm_in_MB : 0.5) + ((r.flags & 8) ? sh.out_MB : 0.0);
            printf("  \"%s\": {\"us\": %.1f, \"compulsory_MB\": %.1f, \"TBps_compulsory\": %.2f},\n", r.name, us, hbm, hbm / us);
        }
        printf("  \"staged_MB_per_launch\": %.0f\n }%s\n", sh.staged_MB, s == 0 ? "," : "");
    }
    printf("}\n");
    return 0;
}
