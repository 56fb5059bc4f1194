// K10 epilogue probe: 256 workgroups x 8 waves, every wave writes (and optionally first reads) 16 KiB of a [pixels][128 channels] 16-bit map as 16
// dwordx4 instructions, in three lane -> byte patterns:
//   seg64 : 16 pixels x 64 B per instruction (lane = pixel l%16, 16-B piece l/16; pixel stride 256 B)   - MFMA 16x16 accumulators stored directly
//   seg128:  8 pixels x 128 B per instruction                                                             - the LDS-slab epilogue of rounds 2-3
//   seg256:  4 pixels x 256 B per instruction (1 KiB contiguous)
// prints the median cycles a wave spends issuing them (s_memtime) and the kernel time.   the last lines: the same with the workgroups of an XCD started in 2 / 4 / 16 phases of the period.   usage: epi_store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int PAT, bool LOAD>
__global__ __launch_bounds__(512) void k(char* out, const char* in, long long* t, int rounds, int phases) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long acc_t = 0;
    // de-phased workgroups: workgroup b starts (b / 8 % phases) / phases of a period late (b % 8 = its XCD)
    for (int s = 0; s < 40 * (int)((blockIdx.x >> 3) % phases) / phases; ++s) __builtin_amdgcn_s_sleep(127);
    for (int r = 0; r < rounds; ++r) {
        const size_t tile = ((size_t)r * gridDim.x + blockIdx.x) * 8 + wave;            // 16 KiB per (round, workgroup, wave)
        char* o = out + tile * 16384;
        const char* ii = in + tile * 16384;
        v4u v[16];
        const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int off;
            if (PAT == 0) off = ((i >> 2) * 16 + (lane & 15)) * 256 + (i & 3) * 64 + (lane >> 4) * 16;      // 64 pixels x 256 B
            else if (PAT == 1) off = ((i >> 1) * 8 + (lane >> 3)) * 256 + (i & 1) * 128 + (lane & 7) * 16;
            else off = (i * 4 + (lane >> 4)) * 256 + (lane & 15) * 16;
            if (LOAD) v[i] = *reinterpret_cast<const v4u*>(ii + off);
            else v[i] = v4u{(unsigned)i, (unsigned)lane, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int off;
            if (PAT == 0) off = ((i >> 2) * 16 + (lane & 15)) * 256 + (i & 3) * 64 + (lane >> 4) * 16;
            else if (PAT == 1) off = ((i >> 1) * 8 + (lane >> 3)) * 256 + (i & 1) * 128 + (lane & 7) * 16;
            else off = (i * 4 + (lane >> 4)) * 256 + (lane & 15) * 16;
            *reinterpret_cast<v4u*>(o + off) = v[i];
        }
        acc_t += __builtin_amdgcn_s_memtime() - t0;
        // ~ a main loop's worth of idle time between epilogues, all workgroups in phase
        for (int s = 0; s < 40; ++s) __builtin_amdgcn_s_sleep(127);
        __syncthreads();
    }
    if (lane == 0) t[blockIdx.x * 8 + wave] = acc_t / rounds;
}
template <int PAT, bool LOAD>
void run(char* out, char* in, long long* t, const char* name, int phases = 1) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 12;
    k<PAT, LOAD><<<256, 512>>>(out, in, t, rounds, phases);
    hipEventRecord(a);
    k<PAT, LOAD><<<256, 512>>>(out, in, t, rounds, phases);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(2048);
    hipMemcpy(h.data(), t, 2048 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-34s issue cycles per wave and epilogue: median %6lld  p90 %6lld | kernel %.1f us\n", name, h[1024], h[1843], ms * 1e3);
}
int main() {
    const size_t bytes = (size_t)12 * 256 * 8 * 16384;
    char *out, *in; long long* t;
    hipMalloc(&out, bytes); hipMalloc(&in, bytes); hipMalloc(&t, 2048 * 8);
    hipMemset(in, 1, bytes);
    run<0, false>(out, in, t, "store  16 px x  64 B");
    run<1, false>(out, in, t, "store   8 px x 128 B");
    run<2, false>(out, in, t, "store   4 px x 256 B");
    run<0, true>(out, in, t, "load+store 16 px x  64 B");
    run<1, true>(out, in, t, "load+store  8 px x 128 B");
    run<2, true>(out, in, t, "load+store  4 px x 256 B");
    run<1, false>(out, in, t, "store 8 px x 128 B, 4 phases", 4);
    run<1, false>(out, in, t, "store 8 px x 128 B, 16 phases", 16);
    run<1, true>(out, in, t, "load+store 8 px x 128 B, 2 phases", 2);
    run<1, true>(out, in, t, "load+store 8 px x 128 B, 4 phases", 4);
    run<1, true>(out, in, t, "load+store 8 px x 128 B, 16 phases", 16);
    return 0;
}
