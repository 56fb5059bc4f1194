// How fast can a [P][224] 16-bit map (448-byte pixel rows: NOT a multiple of the 128-byte line) be written by 256 persistent workgroups x 8
// waves, a wave owning 16 consecutive pixels (7168 contiguous bytes) per 128-pixel tile?
//   pairs  : K10 / K12's register epilogue - per instruction 8 pixels x 128 B (even pixels, then odd pixels: their rows start mid-line),
//            + a lone 16 pixels x 64 B piece
//   linear : 7 instructions of 1024 contiguous bytes (what a transposition through LDS would allow)
//   rows512: the pairs pattern on a [P][256] map (512-byte rows: every 128-byte segment is a whole line)
// usage: row448
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void k(char* out, int P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntiles = P / 128;
    const int ROW = PAT == 2 ? 512 : 448;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t p0 = (size_t)t * 128 + wave * 16;
        char* base = out + p0 * ROW;
        const v4u v{(unsigned)t, (unsigned)lane, 1u, 2u};
        if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 7; ++i) *reinterpret_cast<v4u*>(base + i * 1024 + lane * 16) = v;
        } else {
            const int lp = lane & 15, g4 = lane >> 4, odd = lane & 1;
            const int cb = 64 * odd + 16 * g4;
            const int NM = PAT == 2 ? 4 : 3;
#pragma unroll
            for (int m = 0; m < 4; ++m) if (m < NM) {
                *reinterpret_cast<v4u*>(base + (lp & 14) * ROW + cb + 128 * m) = v;
                *reinterpret_cast<v4u*>(base + ((lp & 14) + 1) * ROW + cb + 128 * m) = v;
            }
            if (PAT == 0) *reinterpret_cast<v4u*>(base + lp * ROW + 384 + 16 * g4) = v;
        }
    }
}
template <int PAT>
void run(char* out, int P, const char* name) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<PAT><<<256, 512>>>(out, P);
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<PAT><<<256, 512>>>(out, P);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)P * (PAT == 2 ? 512 : 448);
    printf("%-10s %7.1f us  %6.2f TB/s\n", name, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
}
int main() {
    const int P = 16 * 320 * 320;
    char* out; (void)hipMalloc(&out, (size_t)P * 512);
    run<0>(out, P, "pairs");
    run<1>(out, P, "linear");
    run<2>(out, P, "rows512");
    run<0>(out, P, "pairs");
    run<1>(out, P, "linear");
    return 0;
}
