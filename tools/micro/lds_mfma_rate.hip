// What bounds the K loop of the 256 x 128 tile (8 waves, 64 x 64 per wave)?  Per 32-deep k-half a wave reads 4 + 4 fragments
// (ds_read_b128, the kernel's swizzled addresses) and issues 16 v_mfma_f32_16x16x32_bf16.  Three loops on a RESIDENT LDS image, one
// workgroup per CU, no barriers, no global traffic: reads only, MFMAs only, both (reads of iteration i + 1 in flight under the MFMAs of i).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_mfma_rate.hip -o /tmp/ldsrate && /tmp/ldsrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE, int ROWB>   // MODE bit 0: reads, 1: MFMAs, 2: s_barrier per k-half, 3: + a counted-wait ladder (four scalar branches) per k-half,
                                // 4: the buffer of the reads rotates through a six-slot ring (address arithmetic per k-half); ROWB: bytes per staged row
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < 384 * ROWB / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1;
    auto key = [](int row) { return ROWB == 128 ? (row & 7) : (((row >> 3) & 1) * 3); };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
    const unsigned a = lds0 + (wm * 64 + fr) * ROWB + ((fq ^ key(fr)) * 16), b = lds0 + 256 * ROWB + (wn * 64 + fr) * ROWB + ((fq ^ key(fr)) * 16);
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4], bf[2][4];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 4; ++i) { af[s][i] = (bf16x8){}; bf[s][i] = (bf16x8){}; }
    auto rd = [&](bf16x8 (&x)[4], bf16x8 (&y)[4]) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(x[0]) : "v"(a));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[1]) : "v"(a), "n"(16 * ROWB));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[2]) : "v"(a), "n"(32 * ROWB));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[3]) : "v"(a), "n"(48 * ROWB));
        asm volatile("ds_read_b128 %0, %1" : "=v"(y[0]) : "v"(b));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y[1]) : "v"(b), "n"(16 * ROWB));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y[2]) : "v"(b), "n"(32 * ROWB));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y[3]) : "v"(b), "n"(48 * ROWB));
    };
    auto mm = [&](const bf16x8 (&x)[4], const bf16x8 (&y)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[i], y[j], acc[i][j], 0, 0, 0);
    };
    const long long t0 = clock64();
    if (MODE & 1) rd(af[0], bf[0]);
    int issued = iters < 5 ? iters : 5;
    auto sync = [&](int it) __attribute__((always_inline)) {
        if (MODE & 8) {
            const int younger = issued - (it + 2);
            if (younger >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (younger == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (it + 5 < iters) ++issued;
        }
        if (MODE & 4) __builtin_amdgcn_s_barrier();
    };
    for (int it = 0; it < iters; it += 2) {
        sync(it);
        if (MODE & 1) { rd(af[1], bf[1]); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 2) mm(af[0], bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        sync(it + 1);
        if (MODE & 1) { rd(af[0], bf[0]); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 2) mm(af[1], bf[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[s][i]), "v"(bf[s][i]));
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* clk; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    const int iters = 4096;
    auto run = [&](auto kern, const char* name, int lds) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<<<256, 512, lds>>>(out, iters, clk); hipDeviceSynchronize();
        hipEventRecord(e0); kern<<<256, 512, lds>>>(out, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[256]; hipMemcpy(c, clk, 256 * 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.1f ns per k-half per workgroup (%.0f clock64 ticks); 8 waves: 64 x 1 KB reads, 128 MFMAs\n", name, ms * 1e6 / iters, (double)c[0] / iters);
    };
    run(k<1, 128>, "reads only, 128-byte rows (BK 64)", 49152 + 1024);
    run(k<2, 128>, "MFMAs only", 49152 + 1024);
    run(k<3, 128>, "reads(i+1) under MFMAs(i), 128-byte rows", 49152 + 1024);
    run(k<1, 64>, "reads only, 64-byte rows (BK 32)", 24576 + 1024);
    run(k<3, 64>, "reads(i+1) under MFMAs(i), 64-byte rows", 24576 + 1024);
    run(k<7, 64>, "... + s_barrier per k-half", 24576 + 1024);
    run(k<15, 64>, "... + barrier + counted-wait ladder", 24576 + 1024);
    run(k<6, 64>, "MFMAs + s_barrier per k-half (no reads)", 24576 + 1024);
    run(k<5, 64>, "reads + s_barrier per k-half (no MFMAs)", 24576 + 1024);
    return 0;
}
