// How many vector instructions does a v_mfma_f32_32x32x16_f16 cover, from the SAME wave and from a partner wave?  (round 5: the K4 / K9 experiments
// contradicted the expectation that interleaved vector work rides under MFMAs.)
// One workgroup per CU, W waves per SIMD (W = 1, 2, 4); every wave runs a loop of [1 MFMA + F filler instructions] x 64 per iteration on registers only
// (no memory), F = 0..12, filler = v_fma_f32 (FMA) or v_exp_f32 (EXP) on independent registers, hand-placed with inline asm so that the compiler
// cannot move them.  Prints cycles per MFMA per SIMD (s_memtime over the loop / MFMAs issued by ALL waves of the SIMD) - 32 = the pipe's rate.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_filler mfma_filler.hip && ./mfma_filler
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int F, bool EXP, bool SPLIT>
__global__ void k(float* out, long long* t, int iters) {
    // SPLIT: the even waves of a SIMD issue ONLY MFMAs, the odd ones ONLY fillers (a 'matrix wave' beside a 'vector wave'); waves go to SIMDs
    // round-robin, so waves w and w + 4 share a SIMD
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = !SPLIT || ((wave >> 2) & 1) == 0, do_fill = !SPLIT || ((wave >> 2) & 1) == 1;
    v16f acc = {0};
    v8h a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = threadIdx.x * 0.01f + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (do_mfma) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            if (do_fill) {
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f[j]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j]) : "v"(f[(j + 1) % 12]));
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 12; ++i) s += f[i];
    if (s == 1.2345f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int F, bool EXP, bool SPLIT>
void run(float* out, long long* t, int W) {
    const int iters = 200;
    k<F, EXP, SPLIT><<<256, 256 * W>>>(out, t, iters);
    hipDeviceSynchronize();
    k<F, EXP, SPLIT><<<256, 256 * W>>>(out, t, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 16);
    hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> v, vf;                                                    // loop times of the waves that issue MFMAs / of the filler-only waves
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 4 * W; ++w) (SPLIT && ((w >> 2) & 1) ? vf : v).push_back(h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    std::sort(vf.begin(), vf.end());
    const double cyc = (double)v[v.size() / 2];
    const double mfma_per_simd = (double)iters * 64 * (SPLIT ? W / 2 : W);           // MFMAs issued on one SIMD during the loop
    printf("  F = %2d %s: %6.1f cycles per MFMA and SIMD (loop %7.0f cycles per MFMA wave", F, EXP ? "v_exp_f32" : "v_fma_f32", cyc / mfma_per_simd, cyc);
    if (SPLIT) printf("; filler wave %7.0f = %.1f cycles per filler", (double)vf[vf.size() / 2], F ? (double)vf[vf.size() / 2] / ((double)iters * 64 * F) : 0.0);
    printf(")\n");
}

template <bool EXP, bool SPLIT>
void sweep(float* out, long long* t, int W) {
    run<0, EXP, SPLIT>(out, t, W); run<2, EXP, SPLIT>(out, t, W); run<4, EXP, SPLIT>(out, t, W); run<5, EXP, SPLIT>(out, t, W);
    run<6, EXP, SPLIT>(out, t, W); run<8, EXP, SPLIT>(out, t, W); run<12, EXP, SPLIT>(out, t, W);
}

int main() {
    float* out; long long* t;
    hipMalloc(&out, 4096 * 4); hipMalloc(&t, 256 * 16 * 8);
    for (int W : {1, 2}) {
        printf("%d wave(s) per SIMD, every wave: [MFMA + F fillers] (the MFMAs of all waves share the SIMD's pipe)\n", W);
        sweep<false, false>(out, t, W);
        if (W == 1) { printf("  -- fillers = v_exp_f32\n"); sweep<true, false>(out, t, W); }
    }
    printf("2 waves per SIMD, SPLIT: one wave issues only MFMAs, its partner only fillers (F per MFMA of the partner)\n");
    sweep<false, true>(out, t, 2);
    printf("  -- fillers = v_exp_f32\n");
    sweep<true, true>(out, t, 2);
    return 0;
}
