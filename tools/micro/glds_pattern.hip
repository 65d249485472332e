// LDS-DMA read rate by access pattern: every workgroup stages 16 steps x 32 KB (256 "rows" x 128 B), either as 128-B pieces of
// 2-KiB-strided rows (what a 1x1 convolution's K-step reads from an NHWC [M][1024] bf16 tensor) or as one contiguous 32-KiB block
// (what it would read from a channel-chunked [K/64][M][64] layout).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/glds_pattern.hip -o /tmp/glds && /tmp/glds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GLDS16(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long tile_stride, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* base = src + (long)blockIdx.x * tile_stride;
    // 3-stage ring, 32 KB per stage: 32 wave-instructions of 1 KB -> 4 per wave
    auto stage = [&](int buf, int kstep) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 8 + wave;                         // 1-KiB piece index 0..31 = rows q*8 .. q*8+7
            const char* g;
            if (MODE == 0) g = base + (long)(q * 8 + (lane >> 3)) * 2048 + kstep * 128 + (lane & 7) * 16;   // rows x 128-B pieces
            else g = base + (long)kstep * 32768 + q * 1024 + lane * 16;                                     // contiguous block
            GLDS16(g, smem + buf * 32768 + q * 1024);
        }
    };
    stage(0, 0); stage(1, 1);
    int acc = 0;
    for (int t = 0; t < 16; ++t) {
        if (t + 2 < 16) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (t + 2 < 16) stage((t + 2) % 3, t + 2);
        acc += *reinterpret_cast<const int*>(smem + (t % 3) * 32768 + threadIdx.x * 16);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

int main() {
    const int WGS = 256 * 4;                                    // 4 rounds of 256 tiles, 512 KB each, far apart: 512 MB walked
    const long tile = 256L * 2048;                              // one tile = 256 rows x 2 KB = 16 steps x 32 KB
    char* buf; int* sink;
    hipMalloc(&buf, tile * WGS); hipMalloc(&sink, 4);
    hipMemset(buf, 1, tile * WGS);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"128-B pieces of 2-KiB rows (HBM)", "contiguous 32-KiB blocks (HBM)", "128-B pieces, 8 tiles shared by all workgroups (L2)",
                            "contiguous, 8 tiles shared (L2)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            const long stride = mode < 2 ? tile : 0;            // stride 0: every workgroup of an XCD walks the same tile -> L2 hits
            hipEventRecord(e0);
            if (mode % 2 == 0) k<0><<<WGS, 512, 98304>>>(buf, stride, sink); else k<1><<<WGS, 512, 98304>>>(buf, stride, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us, %.2f TB/s = %.1f GB/s per CU\n", names[mode], ms * 1e3, tile * WGS / ms / 1e9, tile * WGS / ms / 1e6 / 256);
        }
    return 0;
}
