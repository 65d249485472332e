// How fast can the chip WRITE a [M][N] bf16 tensor when each workgroup stores a (rows x piece-bytes) block at a time?
// The conv kernels store 128- or 256-byte pieces of 2-KB rows; the BN kernels store contiguous streams.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/write_pattern.hip -o /tmp/write_pattern && /tmp/write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// one workgroup = 256 rows; it walks the row in pieces of PIECE bytes; 256 threads, 16 B per lane per store
template <int PIECE>
__global__ __launch_bounds__(256) void write_rows(uint4* __restrict__ out, int row_bytes, int rows_per_wg, int pieces_per_wg, int splits) {
    const int tile_m = blockIdx.x / splits, sp = blockIdx.x % splits;
    constexpr int LPR = PIECE / 16;                 // lanes per row piece
    constexpr int RPI = 256 / LPR;                  // rows per pass of the workgroup
    const int l = threadIdx.x % LPR, r = threadIdx.x / LPR;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    char* base = reinterpret_cast<char*>(out) + (size_t)tile_m * rows_per_wg * row_bytes;
    for (int p = 0; p < pieces_per_wg; ++p) {
        const size_t col = (size_t)(sp * pieces_per_wg + p) * PIECE + l * 16;
        for (int r0 = 0; r0 < rows_per_wg; r0 += RPI)
            if (r0 + r < rows_per_wg) *reinterpret_cast<uint4*>(base + (size_t)(r0 + r) * row_bytes + col) = v;
    }
}

__global__ __launch_bounds__(256) void write_stream(uint4* __restrict__ out, size_t n16) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}

int main() {
    const int rows = 32768 * 4;                     // 4 x the layer-3 tensor: 268 MB at 2-KB rows (beyond the Infinity Cache)
    const int row_bytes = 2048;
    const size_t bytes = (size_t)rows * row_bytes;
    uint4* buf;
    hipMalloc(&buf, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char* name) {
        for (int i = 0; i < 2; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 5;
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %7.1f us  %5.2f TB/s\n", name, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
    };
    time([&] { write_stream<<<2048, 256>>>(buf, bytes / 16); }, "contiguous stream, 2048 WGs");
    const int tiles = rows / 256;
    time([&] { write_rows<128><<<tiles * 2, 256>>>(buf, row_bytes, 256, 8, 2); }, "256 rows x 128 B pieces, 8 per WG (stream kernel, N split 2)");
    time([&] { write_rows<128><<<tiles, 256>>>(buf, row_bytes, 256, 16, 1); }, "256 rows x 128 B pieces, 16 per WG (whole row)");
    time([&] { write_rows<256><<<tiles * 8, 256>>>(buf, row_bytes, 256, 1, 8); }, "256 rows x 256 B, one piece per WG (tiled 256x128)");
    time([&] { write_rows<256><<<tiles, 256>>>(buf, row_bytes, 256, 8, 1); }, "256 rows x 256 B pieces, 8 per WG");
    time([&] { write_rows<512><<<tiles, 256>>>(buf, row_bytes, 256, 4, 1); }, "256 rows x 512 B pieces, 4 per WG");
    time([&] { write_rows<1024><<<tiles, 256>>>(buf, row_bytes, 256, 2, 1); }, "256 rows x 1 KB pieces, 2 per WG");
    time([&] { write_rows<2048><<<tiles, 256>>>(buf, row_bytes, 256, 1, 1); }, "256 rows x 2 KB (whole rows)");
    time([&] { write_rows<2048><<<tiles * 4, 256>>>(buf, row_bytes, 64, 1, 1); }, "64 rows x 2 KB (whole rows), 4x the WGs");
    time([&] { write_rows<128><<<tiles * 4, 256>>>(buf, row_bytes, 64, 16, 1); }, "64 rows x 128 B pieces, 16 per WG");
    time([&] { write_rows<128><<<tiles * 8, 256>>>(buf, row_bytes, 32, 16, 1); }, "32 rows x 128 B pieces, 16 per WG");
    return 0;
}
