// Does an LDS-DMA stage made of HBM-sourced and L2-sourced pieces take the SUM or the MAX of the two fills?  Every workgroup (one per
// CU, 512 threads) runs 64 steps of a 3-stage ring; a 48-KB stage = 16 KB of its own HBM-resident rows (128-B pieces of 2-KiB rows) +
// 32 KB of a tile every workgroup shares (L2 hits) -- the layer-3 1x1 K-step.  MODE 0: HBM part only; 1: L2 part only; 2: both, every wave
// issues its HBM pieces then its L2 pieces; 3: both, waves 0-2 (+ wave 3 half) issue the HBM pieces, the others the L2 pieces.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/glds_mix.hip -o /tmp/gmix && /tmp/gmix
#include <hip/hip_runtime.h>
#include <cstdio>
#define GLDS16(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
constexpr int STEPS = 64;

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ hbm, const char* __restrict__ shared, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* mine = hbm + (long)blockIdx.x * (128L * 8192);            // 128 rows x 8 KiB: 64 steps x 128 B per row
    // pieces of a stage: 16 HBM pieces (1 KiB = 8 rows x 128 B) + 32 shared pieces
    auto hbm_piece = [&](int buf, int t, int q) { GLDS16(mine + (long)(q * 8 + (lane >> 3)) * 8192 + t * 128 + (lane & 7) * 16, smem + buf * 49152 + q * 1024); };
    auto l2_piece = [&](int buf, int t, int q) { GLDS16(shared + ((t & 15) * 32 + q) * 1024 + lane * 16, smem + buf * 49152 + 16384 + q * 1024); };
    auto stage = [&](int buf, int t) {
        if (MODE == 0) { for (int i = 0; i < 2; ++i) hbm_piece(buf, t, i * 8 + wave); }
        if (MODE == 1) { for (int i = 0; i < 4; ++i) l2_piece(buf, t, i * 8 + wave); }
        if (MODE == 2) { for (int i = 0; i < 2; ++i) hbm_piece(buf, t, i * 8 + wave); for (int i = 0; i < 4; ++i) l2_piece(buf, t, i * 8 + wave); }
        if (MODE == 3) {                                                   // 6 instructions per wave either way
            if (wave < 2) { for (int i = 0; i < 6; ++i) hbm_piece(buf, t, wave * 6 + i); }
            else if (wave == 2) { for (int i = 0; i < 4; ++i) hbm_piece(buf, t, 12 + i); for (int i = 0; i < 2; ++i) l2_piece(buf, t, i); }
            else { for (int i = 0; i < 6; ++i) l2_piece(buf, t, 2 + (wave - 3) * 6 + i); }
        }
    };
    constexpr int L = MODE == 0 ? 2 : MODE == 1 ? 4 : 6;
    stage(0, 0); stage(1, 1);
    int acc = 0;
    for (int t = 0; t < STEPS; ++t) {
        if (t + 2 < STEPS) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (t + 2 < STEPS) stage((t + 2) % 3, t + 2);
        acc += *reinterpret_cast<const int*>(smem + (t % 3) * 49152 + threadIdx.x * 16);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main() {
    const int WGS = 256;
    char *hbm, *sh; int* sink;
    (void)hipMalloc(&hbm, 128L * 8192 * WGS * 4); (void)hipMalloc(&sh, 512 * 1024); (void)hipMalloc(&sink, 4);
    (void)hipMemset(hbm, 1, 128L * 8192 * WGS * 4); (void)hipMemset(sh, 1, 512 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[4] = {"HBM part only (16 KB/step)", "L2 part only (32 KB/step)", "both, every wave issues HBM then L2 pieces", "both, separate waves per source"};
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 4; ++rep) {
            const char* base = hbm + (long)(rep % 4) * 128L * 8192 * WGS;   // a fresh 256-MB region per repetition: HBM, not cache
            (void)hipEventRecord(e0);
            if (mode == 0) { (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456); k<0><<<WGS, 512, 147456>>>(base, sh, sink); }
            if (mode == 1) { (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456); k<1><<<WGS, 512, 147456>>>(base, sh, sink); }
            if (mode == 2) { (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456); k<2><<<WGS, 512, 147456>>>(base, sh, sink); }
            if (mode == 3) { (void)hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456); k<3><<<WGS, 512, 147456>>>(base, sh, sink); }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-50s %.2f us per step\n", names[mode], ms * 1e3 / STEPS);
        }
    return 0;
}
