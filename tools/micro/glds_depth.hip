// HBM-sourced LDS-DMA per CU against bytes in flight: 16 KB per step (128-B pieces of 8-KiB rows), ring of NST stages (NST - 1 in flight),
// one 512-thread workgroup per CU, 64 steps.   hipcc -O3 --offload-arch=gfx950 tools/micro/glds_depth.hip -o /tmp/gd && /tmp/gd
#include <hip/hip_runtime.h>
#include <cstdio>
#define GLDS16(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
constexpr int STEPS = 64;
template <int NST, int KB>     // KB per step: 16 or 32
__global__ __launch_bounds__(512) void k(const char* __restrict__ hbm, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int PW = KB / 8;                                           // 1-KiB pieces per wave per stage
    const char* mine = hbm + (long)blockIdx.x * (KB * 8L * 8192);        // KB * 8 rows x 8 KiB
    auto stage = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = i * 8 + wave;
            GLDS16(mine + (long)(q * 8 + (lane >> 3)) * 8192 + t * 128 + (lane & 7) * 16, smem + buf * (KB * 1024) + q * 1024);
        }
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) stage(s, s);
    int acc = 0;
    for (int t = 0; t < STEPS; ++t) {
        if (t + NST - 1 < STEPS) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PW) : "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (t + NST - 1 < STEPS) stage((t + NST - 1) % NST, t + NST - 1);
        acc += *reinterpret_cast<const int*>(smem + (t % NST) * (KB * 1024) + threadIdx.x * 16);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
template <int NST, int KB> void run(const char* hbm, int* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<NST, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * KB * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        k<NST, KB><<<256, 512, NST * KB * 1024>>>(hbm + (long)rep * (KB * 8L * 8192 * 256), sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%2d KB per step, ring %d (%3d KB in flight per CU): %.2f us per step = %.1f GB/s per CU = %.2f TB/s\n", KB, NST, (NST - 1) * KB,
           best * 1e3 / STEPS, KB * 1.024 / (best * 1e3 / STEPS), KB * 1.024 / (best * 1e3 / STEPS) * 256 / 1e3);
}
int main() {
    char* hbm; int* sink;
    (void)hipMalloc(&hbm, 32 * 8L * 8192 * 256 * 4); (void)hipMalloc(&sink, 4);
    (void)hipMemset(hbm, 1, 32 * 8L * 8192 * 256 * 4);
    run<3, 16>(hbm, sink); run<4, 16>(hbm, sink); run<5, 16>(hbm, sink); run<7, 16>(hbm, sink); run<9, 16>(hbm, sink);
    run<3, 32>(hbm, sink); run<4, 32>(hbm, sink); run<5, 32>(hbm, sink);
    return 0;
}
