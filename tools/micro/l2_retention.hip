// Does a consumer launch find its producer's output in the producer XCD's L2?  (round 6)  In the step a tensor passes from one launch to
// the next; the L2 is per XCD (4 MB x 8) and the workgroup -> XCD mapping is round-robin, so a consumer workgroup can be put on the XCD
// whose workgroup WROTE the rows it reads -- if the L2 keeps them across the kernel boundary.  Producer: workgroup i writes chunk i of a
// buffer.  Consumer: workgroup i reads chunk (i + shift) % n: shift 0 = same XCD, shift 1 = the neighbour XCD.  Sizes from 4 MB (fits
// every L2) to 256 MB.  A cache-flushing launch between the two gives the cold reference.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l2_retention.hip -o /tmp/l2_retention && /tmp/l2_retention
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void produce(uint4* __restrict__ buf, int chunk16) {
    uint4* p = buf + (size_t)blockIdx.x * chunk16;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (int i = threadIdx.x; i < chunk16; i += 256) p[i] = v;
}
__global__ __launch_bounds__(256) void consume(const uint4* __restrict__ buf, int chunk16, int shift, unsigned* __restrict__ sink) {
    const int c = (blockIdx.x + shift) % gridDim.x;
    const uint4* p = buf + (size_t)c * chunk16;
    unsigned acc = 0;
    for (int i = threadIdx.x; i < chunk16; i += 256 * 4) {
        uint4 a = p[i], b = i + 256 < chunk16 ? p[i + 256] : a, d = i + 512 < chunk16 ? p[i + 512] : a, e = i + 768 < chunk16 ? p[i + 768] : a;
        acc += a.x + b.y + d.z + e.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void flush(uint4* __restrict__ junk, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) junk[i] = make_uint4(1, 2, 3, 4);
}

int main() {
    uint4 *buf, *junk;
    unsigned* sink;
    const size_t maxb = 256u << 20, junkb = 512u << 20;
    hipMalloc(&buf, maxb); hipMalloc(&junk, junkb); hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nwg = 2048;                                       // 8 per CU, i % 8 = XCD
    for (size_t mb : {4, 8, 16, 32, 64, 128, 256}) {
        const size_t bytes = mb << 20;
        const int chunk16 = (int)(bytes / 16 / nwg);
        float t[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {                  // 0: same XCD, 1: neighbour XCD, 2: caches flushed in between
            float sum = 0;
            const int reps = 10;
            for (int r = 0; r < reps + 2; ++r) {
                produce<<<nwg, 256>>>(buf, chunk16);
                if (mode == 2) flush<<<2048, 256>>>(junk, junkb / 16);
                hipEventRecord(e0);
                consume<<<nwg, 256>>>(buf, chunk16, mode == 1 ? 1 : 0, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) sum += ms;
            }
            t[mode] = sum / reps * 1e3f;
        }
        printf("%4zu MB: consumer on the producer's XCD %7.1f us (%5.2f TB/s) | on the neighbour XCD %7.1f us (%5.2f TB/s) | after a 512-MB flush %7.1f us (%5.2f TB/s)\n",
               mb, t[0], bytes / t[0] / 1e6, t[1], bytes / t[1] / 1e6, t[2], bytes / t[2] / 1e6);
    }
    return 0;
}
