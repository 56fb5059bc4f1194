// How fast can 4-byte-per-lane stores fill an [N*L, S] fp32 matrix for different workgroup tile shapes?
// (guides the K1 pass-B tile geometry)   usage: store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ROWS, int COLS>   // workgroup tile ROWS x COLS floats, 256 threads, wave w: (ROWS/4 rows) x COLS or so
__global__ void fill(float* out, int L, int S) {
    const int tiles_n = S / COLS;
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave instruction stores two 128-B segments (two rows x 32 floats), like an MFMA 32x32 accumulator register
    constexpr int SEGS_PER_ROW = COLS / 32;
    constexpr int TOTAL_SEG = ROWS * SEGS_PER_ROW;          // 128-B segments in the tile
    for (int s = wave * 2 + (lane >> 5); s < TOTAL_SEG; s += 8) {
        const int row = bm * ROWS + s / SEGS_PER_ROW, col = bn * COLS + (s % SEGS_PER_ROW) * 32 + (lane & 31);
        out[(size_t)row * S + col] = (float)s;
    }
}
template <int ROWS, int COLS>
void run(float* d, int L, int S, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (L / ROWS) * (S / COLS);
    for (int i = 0; i < 3; ++i) fill<ROWS, COLS><<<grid, 256>>>(d, L, S);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) fill<ROWS, COLS><<<grid, 256>>>(d, L, S);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.1f us  %7.1f GB/s\n", name, ms * 100, (double)L * S * 4 / (ms / 10 * 1e-3) / 1e9);
}
int main() {
    const int L = 8 * 6400, S = 6400;
    float* d; hipMalloc(&d, (size_t)L * S * 4);
    run<128, 64>(d, L, S, "tile 128 rows x 256 B");
    run<128, 128>(d, L, S, "tile 128 rows x 512 B");
    run<32, 256>(d, L, S, "tile  32 rows x 1 KiB");
    run<16, 512>(d, L, S, "tile  16 rows x 2 KiB");
    run<8, 1280>(d, L, S, "tile   8 rows x 5 KiB");
    run<4, 6400>(d, L, S, "tile   4 full rows");
    return 0;
}
