// K3: the encoder-layer GEMM chain with fused epilogues (CDNA4 / gfx950).
//
// Replaces, inside LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:45-60,
// ReLU) and its Geo twin (model/geo_transformer/transformer.py:49-66, Tanh):
//     q/k/v projections                         -> EPI_NONE
//     merge + norm1                              -> EPI_LN
//     mlp.0 on cat([x, message]) + ReLU/Tanh     -> EPI_RELU / EPI_TANH with a two-part A operand (no cat)
//     mlp.2 + norm2 + residual (+ per-sample     -> EPI_LN_RES
//         "layer skipped" predicate of GeoTransformer)
// and the two biased linears of FinePreprocess (fine_preprocess.py:61-72), whose per-match context term
// enters as a row-group bias.
//
// out[m, n] = epi( sum_k A[m,k] * W[n,k] )          A = [A1 | A2] along k, W row-major [N, K] (nn.Linear)
//
// The product is computed TRANSPOSED (MFMA A-operand = weight rows, B-operand = token rows) so that the
// accumulator layout has the TOKEN on the lane and its channels in registers: LayerNorm statistics,
// the activation, the residual and the predicate are then lane-local (one lane-pair exchange for the
// mean/variance), with no cross-lane reduction tree.  One wave = 32 tokens x (32*NB) channels (the fp16 plain /
// ReLU / Tanh epilogues on 256-wide tiles use the 2 x 2 wave tiling of linear_kernel_w2: 64 tokens x 128 channels);
// workgroup = 4 waves = 128 tokens; LN epilogues need NB*32 == N (whole rows in one wave pair).
// fp16 kernels are held to 256 registers (two workgroups per CU; the unconstrained build took 330-380
// and ran one) and transpose their result through LDS for row-contiguous 16-B stores.
#include <math.h>

#include <type_traits>

#include "gf_common.h"

// -DK3_TRACE=1 records clock64() at the phase boundaries of every workgroup (tools/k3_trace.py reads them).
#ifndef K3_TRACE
#define K3_TRACE 0
#endif
#if K3_TRACE
__device__ long long k3_trace[1024 * 4 * 16];
#define K3_T(slot) do { if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 1024) k3_trace[(blockIdx.x * 4 + wave) * 16 + (slot)] = clock64(); } while (0)
extern "C" int gf_debug_k3_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k3_trace), sizeof(long long) * 1024 * 4 * 16);
}
#else
#define K3_T(slot)
#endif

namespace {

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_TANH = 2, EPI_LN = 3, EPI_LN_RES = 4, EPI_UPADD = 5 };

struct LinArgs {
    const void* a1;
    const void* a2;
    long lda1, lda2;
    int k1, k2;              // K = k1 + k2 (k2 == 0: single operand)
    const void* w;           // [N][K]
    const float* bias;       // [N] or null
    const void* rgbias;      // [M / rg_rows][N] of T or null  (added per row group)
    int rg_rows;
    const float* gamma;      // LN affine (fp32)
    const float* beta;
    float eps;
    const void* res;         // residual [M][ldres] of T (EPI_LN_RES)
    long ldres;
    const int32_t* flag;     // [M / flag_rows] or null: 0 -> out = res (update skipped)
    int flag_rows;
    void* out;
    long ldo;
    int M, N;
    // EPI_UPADD (the FPN merge fused into the 1x1 lateral convolution): rows are the pixels of [n][H][W] maps and
    // out += bilinear(lo -> H x W, align_corners=True), lo = [n][h][w][N] of T
    const void* up_lo;
    int up_h, up_w, up_H, up_W, up_n;
    float up_ry, up_rx;
    // row map of a1 (a strided 1x1 convolution's input pixels; k2 == 0): row r of the product reads a1 at element offset
    // (r / rm_w) * rm_line + (r % rm_w) * rm_pix instead of r * lda1; rm_w == 0: off
    int rm_w;
    long rm_line, rm_pix;
};

// Tile of a workgroup.  Linear workgroup ids go round-robin over the 8 XCDs (= 8 L2s); with the grid's native order the column
// tiles of one row tile are gridDim.x ids apart: same XCD, but its 4 MB L2 has long dropped the token rows, and every column
// tile re-reads them from HBM.  Re-deal the ids in groups of 8 x gridDim.y: the gridDim.y column tiles of row tile 8 g + x all run
// on XCD x, 8 ids apart - the second and third read of the rows hit that L2.
__device__ __forceinline__ void lin_tile(int& row_tile, int& col_tile) {
    const int gx = gridDim.x, gy = gridDim.y;
    row_tile = blockIdx.x;
    col_tile = blockIdx.y;
    if (gy > 1 && (gx & 7) == 0) {
        const int id = blockIdx.y * gx + blockIdx.x, grp = id / (8 * gy), j = id - grp * (8 * gy);
        row_tile = grp * 8 + (j & 7);
        col_tile = j >> 3;
    }
}

// CONVX: the forms the backbone's 1x1 convolutions need (compiled only into their instances: the matching path's kernels keep their
// register budget) - K not a multiple of the K step, and the strided-pixel row map of LinArgs
template <typename T, int NB, int EPI, bool CONVX = false>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void linear_kernel(LinArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 128 / sizeof(T);
    constexpr int WROWS = 32 * NB;
    constexpr int ROWS = 128 + WROWS;
    constexpr int NLD = ROWS * 8 / 256;          // 16-B chunks staged per thread per K step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* st = smem;                              // token rows
    char* sw = smem + 128 * 128;                  // weight rows
    float* vec = reinterpret_cast<float*>(smem + ROWS * 128);   // bias | gamma | beta  [3][WROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    int row_tile, col_tile;
    lin_tile(row_tile, col_tile);
    const int m0 = row_tile * 128, n0 = col_tile * WROWS;
    const int K = a.k1 + a.k2;

    if (tid < WROWS) {
        vec[tid] = (a.bias && n0 + tid < a.N) ? a.bias[n0 + tid] : 0.f;
        if constexpr (EPI == EPI_LN || EPI == EPI_LN_RES) {
            vec[WROWS + tid] = a.gamma[n0 + tid];
            vec[2 * WROWS + tid] = a.beta[n0 + tid];
        }
    }
    // per-thread staging slots: chunk e -> (row, chunk-in-row); rows < 128 are tokens, the rest weights
    const T* src[NLD];
    int dst[NLD];
    bool isa[NLD];
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
        const int e = p * 256 + tid, row = e >> 3, c = e & 7;
        isa[p] = row < 128;
        if (isa[p]) {
            src[p] = nullptr;                      // resolved per K step (two-part operand)
            dst[p] = gf_lds_off(row, c);
        } else {
            const int wr = row - 128;
            src[p] = (const T*)a.w + (size_t)min(n0 + wr, a.N - 1) * K + c * EPC;
            dst[p] = 128 * 128 + gf_lds_off(wr, c);
        }
    }
    v4u regs[NLD];
    // K need not be a multiple of the 128-byte K step (a 224-channel 1x1 convolution): chunks at or behind a row's end are staged
    // as zeros on both sides (k2 == 0 then: the entry point checks)
    const bool ragged = CONVX && (K % BK) != 0;
    auto gload = [&](int k0) {
        const bool first = k0 < a.k1;
        const T* ab = (const T*)(first ? a.a1 : a.a2);
        const long ld = first ? a.lda1 : a.lda2;
        const int kk = first ? k0 : k0 - a.k1;
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int e = p * 256 + tid, row = e >> 3, c = e & 7;
            const T* g;
            if (isa[p]) {
                const int r = min(m0 + row, a.M - 1);
                const size_t ro = (CONVX && a.rm_w > 0) ? (size_t)(r / a.rm_w) * a.rm_line + (size_t)(r % a.rm_w) * a.rm_pix : (size_t)r * ld;
                g = ab + ro + kk + c * EPC;
            } else {
                g = src[p] + k0;
            }
            if (!ragged || k0 + c * EPC < K) regs[p] = *reinterpret_cast<const v4u*>(g);
            else regs[p] = v4u{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < NLD; ++p) *reinterpret_cast<v4u*>(smem + dst[p]) = regs[p];
    };
    v16f acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    K3_T(0);
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                           // previous step's fragments are consumed
        lstore();
        __syncthreads();
        K3_T(1 + kt);
        if (kt + 1 < nk) gload((kt + 1) * BK);     // in flight while the MFMAs run
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + h;
            const Frag tf = *reinterpret_cast<const Frag*>(st + gf_lds_off(wave * 32 + lr, chunk));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const Frag wf = *reinterpret_cast<const Frag*>(sw + gf_lds_off(nb * 32 + lr, chunk));
                Mm::mma(wf, tf, acc[nb]);
            }
        }
    }
    K3_T(10);
    // ---------------- epilogue: lane = token, registers = channels n0 + nb*32 + acc_row(r, h)
    const int token = m0 + wave * 32 + lr;
    const bool live = token < a.M;
    const int tk = live ? token : a.M - 1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] += vec[nb * 32 + gf_acc_row(r, h)];
    if (a.rgbias) {
        const T* rb = (const T*)a.rgbias + (size_t)(tk / a.rg_rows) * a.N + n0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] += gf_to_float(rb[nb * 32 + gf_acc_row(r, h)]);
    }
    if constexpr (EPI == EPI_RELU) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = fmaxf(acc[nb][r], 0.f);
    } else if constexpr (EPI == EPI_TANH) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (std::is_same<T, float>::value) acc[nb][r] = tanhf(acc[nb][r]);
                else acc[nb][r] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * acc[nb][r]) + 1.0f);   // tanh, ~1e-6 abs
            }
    } else if constexpr (EPI == EPI_LN || EPI == EPI_LN_RES) {
        // nn.LayerNorm over the WROWS channels of the token: two-pass mean / variance in fp32
        float s = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[nb][r];
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)WROWS;
        float q = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[nb][r] - mean;
                q += d * d;
            }
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q / (float)WROWS + a.eps);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = nb * 32 + gf_acc_row(r, h);
                acc[nb][r] = (acc[nb][r] - mean) * rstd * vec[WROWS + c] + vec[2 * WROWS + c];
            }
    }
    if constexpr (!std::is_same<T, float>::value) {
        using V4 = gf_vec<T, 4>;
        using V8 = gf_vec<T, 8>;
        // fp16: transpose through LDS so that global stores (and the residual loads) are 16 B per lane
        // along the row - 256-B contiguous segments instead of 64 scattered 8-B pieces per instruction.
        // One 32-token x 128-channel slab per wave at a time (the staging buffers are free now).
        constexpr int RS = 272;                                          // slab row stride (256 B + pad)
        K3_T(11);
        __syncthreads();                                                 // every wave is done with the fragments
        K3_T(12);
        char* ot = smem + wave * 32 * RS;
        const int prow = lane >> 4, pch = lane & 15;
        int upx = 0, upy = 0, upn = 0;
        if constexpr (EPI == EPI_UPADD) {
            const unsigned t0 = (unsigned)__builtin_amdgcn_readfirstlane(m0 + wave * 32);
            const unsigned q = t0 / (unsigned)a.up_W;
            upx = (int)(t0 - q * (unsigned)a.up_W);
            upy = (int)(q % (unsigned)a.up_H);
            upn = (int)(q / (unsigned)a.up_H);
        }
#pragma unroll
        for (int hb = 0; hb < (NB + 3) / 4; ++hb) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int nb = hb * 4 + q4, c = q4 * 32 + 8 * r4 + 4 * h;
                    if (nb < NB)                                         // NB = 7: the last slab holds three blocks
                        *reinterpret_cast<V4*>(ot + lr * RS + c * 2) =
                            V4{(T)acc[nb][4 * r4], (T)acc[nb][4 * r4 + 1], (T)acc[nb][4 * r4 + 2], (T)acc[nb][4 * r4 + 3]};
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (EPI == EPI_UPADD) {
                // Two phases, no branch between them: ALL 32 tap rows of the slab are requested, then consumed.  (Round 3's form - taps
                // loaded and used row group by row group, with `continue` / the wrap loop between the groups - exposed one L2 round trip
                // per group: 16 thousand cycles per slab, 30 of a tile's 40 thousand; tools/k3_upadd_trace.py.)
                const int cg = n0 + hb * 128 + pch * 8;
                const bool cok = cg < a.N;
                const T* lo = (const T*)a.up_lo + (cok ? cg : 0);
#ifdef K3_EXP_NOGATHER     // ablation: the four tap rows are one row (L1 hits)
#define K3_TAP(p) (*reinterpret_cast<const V8*>(lp))
#else
#define K3_TAP(p) (*reinterpret_cast<const V8*>(p))
#endif
#ifndef K3_UPADD_BATCH
#define K3_UPADD_BATCH 4
#endif
                constexpr int UB = K3_UPADD_BATCH;                       // row groups per batch (8: 194 registers = two waves per SIMD instead of three)
#pragma unroll
                for (int b0 = 0; b0 < 8; b0 += UB) {
                V8 tap[UB][4];
                float wgt[UB][2];
#pragma unroll
                for (int ib = 0; ib < UB; ++ib) {
                    const int it = b0 + ib;
                    // pixel of this row: the wave's first row is decoded once (uniform), rows step along x and wrap (rows of 32+ pixels:
                    // at most once; narrower maps take the division)
                    const int row = it * 4 + prow;
                    int px = upx + row, py = upy, pn = upn;
                    if (a.up_W >= 32) {
                        const bool wr = px >= a.up_W;
                        px -= wr ? a.up_W : 0;
                        py += wr ? 1 : 0;
                        const bool wy = py >= a.up_H;
                        py = wy ? 0 : py;
                        pn += wy ? 1 : 0;
                    } else {
                        const unsigned t = (unsigned)(m0 + wave * 32 + row), q = t / (unsigned)a.up_W;
                        px = (int)(t - q * (unsigned)a.up_W);
                        py = (int)(q % (unsigned)a.up_H);
                        pn = (int)(q / (unsigned)a.up_H);
                    }
                    pn = pn < a.up_n ? pn : a.up_n - 1;                  // (rows behind the last pixel: any valid address; not stored)
                    const float fy = a.up_ry * py, fx = a.up_rx * px;
                    const int y0 = (int)fy, x0 = (int)fx;
                    const int y1 = y0 + (y0 < a.up_h - 1), x1 = x0 + (x0 < a.up_w - 1);
                    wgt[ib][0] = fy - y0;
                    wgt[ib][1] = fx - x0;
                    const T* lp = lo + (size_t)pn * a.up_h * a.up_w * a.N;
                    tap[ib][0] = K3_TAP(lp + ((size_t)y0 * a.up_w + x0) * a.N);
                    tap[ib][1] = K3_TAP(lp + ((size_t)y0 * a.up_w + x1) * a.N);
                    tap[ib][2] = K3_TAP(lp + ((size_t)y1 * a.up_w + x0) * a.N);
                    tap[ib][3] = K3_TAP(lp + ((size_t)y1 * a.up_w + x1) * a.N);
                }
#pragma unroll
                for (int ib = 0; ib < UB; ++ib) {
                    const int row = (b0 + ib) * 4 + prow, tg = m0 + wave * 32 + row;
                    V8 v = *reinterpret_cast<const V8*>(ot + row * RS + pch * 16);
                    const float wy1 = wgt[ib][0], wx1 = wgt[ib][1], wy0 = 1.f - wy1, wx0 = 1.f - wx1;
                    if constexpr (std::is_same<T, _Float16>::value) {
                        // v + w00 v00 + w01 v01 + w10 v10 + w11 v11 as four v_fma_mix_f32 per value (half operands converted inside the
                        // instruction, fp32 product and sum): 4.5 vector instructions per output value instead of 11 (five conversions
                        // and the packed fp32 forms the compiler makes of the nested products)
                        const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
                        const v4u uv = __builtin_bit_cast(v4u, v), u0 = __builtin_bit_cast(v4u, tap[ib][0]), u1 = __builtin_bit_cast(v4u, tap[ib][1]),
                                  u2 = __builtin_bit_cast(v4u, tap[ib][2]), u3 = __builtin_bit_cast(v4u, tap[ib][3]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float lo_, hi_;
                            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(lo_) : "v"(u0[i]), "v"(w00), "v"(uv[i]));
                            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hi_) : "v"(u0[i]), "v"(w00), "v"(uv[i]));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u1[i]), "v"(w01));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u1[i]), "v"(w01));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u2[i]), "v"(w10));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u2[i]), "v"(w10));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u3[i]), "v"(w11));
                            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u3[i]), "v"(w11));
                            v[2 * i] = (T)lo_;
                            v[2 * i + 1] = (T)hi_;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            v[i] = (T)((float)v[i] + wy0 * (wx0 * (float)tap[ib][0][i] + wx1 * (float)tap[ib][1][i]) +
                                              wy1 * (wx0 * (float)tap[ib][2][i] + wx1 * (float)tap[ib][3][i]));
                    }
                    if (tg < a.M && cok) *reinterpret_cast<V8*>((T*)a.out + (size_t)tg * a.ldo + cg) = v;
                }
                }
            } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + prow, tg = m0 + wave * 32 + row, cg = n0 + hb * 128 + pch * 8;
                if (tg >= a.M || cg >= a.N) continue;
                V8 v = *reinterpret_cast<const V8*>(ot + row * RS + pch * 16);
                if constexpr (EPI == EPI_LN_RES) {
                    const V8 x = *reinterpret_cast<const V8*>((const T*)a.res + (size_t)tg * a.ldres + cg);
                    const bool keep = a.flag == nullptr || a.flag[tg / a.flag_rows] != 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = keep ? (T)((float)x[i] + (float)v[i]) : x[i];
                }
                *reinterpret_cast<V8*>((T*)a.out + (size_t)tg * a.ldo + cg) = v;
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            K3_T(13 + hb);
        }
    } else {
        bool keep = true;
        if constexpr (EPI == EPI_LN_RES) keep = a.flag == nullptr || a.flag[tk / a.flag_rows] != 0;
        if (!live) return;
        T* op = (T*)a.out + (size_t)token * a.ldo + n0;
        const T* rp = EPI == EPI_LN_RES ? (const T*)a.res + (size_t)token * a.ldres + n0 : nullptr;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c = nb * 32 + 8 * r4 + 4 * h;                     // 4 consecutive channels
                if (n0 + c >= a.N) continue;
                v4f v{acc[nb][4 * r4], acc[nb][4 * r4 + 1], acc[nb][4 * r4 + 2], acc[nb][4 * r4 + 3]};
                if constexpr (EPI == EPI_LN_RES) {
                    const v4f x = *reinterpret_cast<const v4f*>(rp + c);
                    v = keep ? v4f{x.x + v.x, x.y + v.y, x.z + v.z, x.w + v.w} : x;
                }
                *reinterpret_cast<v4f*>(op + c) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// 2 x 2 wave tiling (fp16, 256-channel column tiles, epilogues without LayerNorm): a wave owns 64 tokens x 128
// channels instead of 32 x 256.  Same 128 accumulator registers, but a k-group now needs 2 token + 4 weight
// fragments for its 8 MFMAs instead of 1 + 8: a third less LDS read traffic in the K loop, which is LDS-bound.
// ---------------------------------------------------------------------------------------------
template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void linear_kernel_w2(LinArgs a) {
    using Frag = typename Mma32<T>::Frag;
    using V4 = gf_vec<T, 4>;
    using V8 = gf_vec<T, 8>;
    constexpr int EPC = 8, BK = 64, WROWS = 256, ROWS = 128 + WROWS, NLD = ROWS * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* st = smem;
    char* sw = smem + 128 * 128;
    float* vec = reinterpret_cast<float*>(smem + ROWS * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    const int tm = wave >> 1, tn = wave & 1;
    int row_tile, col_tile;
    lin_tile(row_tile, col_tile);
    const int m0 = row_tile * 128, n0 = col_tile * WROWS;
    const int K = a.k1 + a.k2;
    if (tid < WROWS) vec[tid] = (a.bias && n0 + tid < a.N) ? a.bias[n0 + tid] : 0.f;
    const T* src[NLD];
    int dst[NLD];
    bool isa[NLD];
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
        const int e = p * 256 + tid, row = e >> 3, c = e & 7;
        isa[p] = row < 128;
        if (isa[p]) {
            src[p] = nullptr;
            dst[p] = gf_lds_off(row, c);
        } else {
            const int wr = row - 128;
            src[p] = (const T*)a.w + (size_t)min(n0 + wr, a.N - 1) * K + c * EPC;
            dst[p] = 128 * 128 + gf_lds_off(wr, c);
        }
    }
    v4u regs[NLD];
    auto gload = [&](int k0) {
        const bool first = k0 < a.k1;
        const T* ab = (const T*)(first ? a.a1 : a.a2);
        const long ld = first ? a.lda1 : a.lda2;
        const int kk = first ? k0 : k0 - a.k1;
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int e = p * 256 + tid, row = e >> 3, c = e & 7;
            const T* g = isa[p] ? ab + (size_t)min(m0 + row, a.M - 1) * ld + kk + c * EPC : src[p] + k0;
            regs[p] = *reinterpret_cast<const v4u*>(g);
        }
    };
    v16f acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nb][r] = 0.f;
    const int nk = K / BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NLD; ++p) *reinterpret_cast<v4u*>(smem + dst[p]) = regs[p];
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + h;
            const Frag t0 = *reinterpret_cast<const Frag*>(st + gf_lds_off(tm * 64 + lr, chunk));
            const Frag t1 = *reinterpret_cast<const Frag*>(st + gf_lds_off(tm * 64 + 32 + lr, chunk));
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const Frag wf = *reinterpret_cast<const Frag*>(sw + gf_lds_off(tn * 128 + nb * 32 + lr, chunk));
                Mma32<T>::mma(wf, t0, acc[0][nb]);
                Mma32<T>::mma(wf, t1, acc[1][nb]);
            }
        }
    }
    // epilogue: lane = token (m0 + tm*64 + t*32 + lr), registers = channels n0 + tn*128 + nb*32 + acc_row(r, h)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][nb][r] + vec[tn * 128 + nb * 32 + gf_acc_row(r, h)];
                if constexpr (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                else if constexpr (EPI == EPI_TANH) v = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f);
                acc[t][nb][r] = v;
            }
    constexpr int RS = 272;
    __syncthreads();                                     // every wave is done with the fragments
    char* ot = smem + wave * 32 * RS;
    const int prow = lane >> 4, pch = lane & 15;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c = nb * 32 + 8 * r4 + 4 * h;
                *reinterpret_cast<V4*>(ot + lr * RS + c * 2) =
                    V4{(T)acc[t][nb][4 * r4], (T)acc[t][nb][4 * r4 + 1], (T)acc[t][nb][4 * r4 + 2], (T)acc[t][nb][4 * r4 + 3]};
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + prow, tg = m0 + tm * 64 + t * 32 + row, cg = n0 + tn * 128 + pch * 8;
            if (tg >= a.M || cg >= a.N) continue;
            *reinterpret_cast<V8*>((T*)a.out + (size_t)tg * a.ldo + cg) = *reinterpret_cast<const V8*>(ot + row * RS + pch * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int NB, int EPI, bool CONVX = false>
void lin_launch1(const LinArgs& a, hipStream_t st) {
    constexpr int WROWS = 32 * NB;
    size_t lds = (size_t)(128 + WROWS) * 128 + 3 * WROWS * sizeof(float);
    if (lds < 4 * 32 * 272) lds = 4 * 32 * 272;                          // fp16 epilogue slabs
    linear_kernel<T, NB, EPI, CONVX><<<dim3((a.M + 127) / 128, (a.N + WROWS - 1) / WROWS), 256, lds, st>>>(a);
}

template <typename T, int NB>
void lin_launch(const LinArgs& a, int epi, hipStream_t st) {
    switch (epi) {
        case EPI_NONE: lin_launch1<T, NB, EPI_NONE>(a, st); break;
        case EPI_RELU: lin_launch1<T, NB, EPI_RELU>(a, st); break;
        case EPI_TANH: lin_launch1<T, NB, EPI_TANH>(a, st); break;
        case EPI_LN: lin_launch1<T, NB, EPI_LN>(a, st); break;
        default: lin_launch1<T, NB, EPI_LN_RES>(a, st); break;
    }
}

}   // namespace

extern "C" int gf_linear(const void* a1, long lda1, int k1, const void* a2, long lda2, int k2, const void* w,
                         const float* bias, const void* rowgroup_bias, int rowgroup_rows, int epilogue,
                         const float* ln_gamma, const float* ln_beta, float ln_eps, const void* residual,
                         long ldres, const int32_t* row_flag, int flag_rows, void* out, long ldo, int dtype, int M,
                         int N, void* stream) {
    GF_CHECK_ARG(a1 && w && out, "null pointer");
    GF_CHECK_ARG(M > 0 && N > 0 && k1 > 0 && k2 >= 0, "bad sizes");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    const int bk = dtype == GF_F32 ? 32 : 64;
    GF_CHECK_ARG(k1 % bk == 0 && k2 % bk == 0, "k1 and k2 must be multiples of 32 (f32) / 64 (f16)");
    GF_CHECK_ARG(k2 == 0 || a2 != nullptr, "a2 missing");
    GF_CHECK_ARG(N % 32 == 0, "N must be a multiple of 32");
    GF_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_LN_RES, "unknown epilogue");
    GF_CHECK_ARG(rowgroup_bias == nullptr || rowgroup_rows > 0, "rowgroup_rows must be > 0");
    const int es = dtype == GF_F32 ? 4 : 2;
    GF_CHECK_ARG((lda1 * es) % 16 == 0 && (k2 == 0 || (lda2 * es) % 16 == 0), "operand rows must be 16-byte aligned");
    GF_CHECK_ARG((ldo * es) % 16 == 0 && (uintptr_t)out % 16 == 0, "output rows must be 16-byte aligned");
    GF_CHECK_ARG(residual == nullptr || ((ldres * es) % 16 == 0 && (uintptr_t)residual % 16 == 0),
                 "residual rows must be 16-byte aligned");
    if (epilogue >= EPI_LN) {
        GF_CHECK_ARG(ln_gamma && ln_beta, "LayerNorm epilogue needs gamma and beta");
        GF_CHECK_ARG(N == 128 || N == 256, "LayerNorm epilogue is built for N = 128 or 256 (whole rows per wave pair)");
        GF_CHECK_ARG(epilogue != EPI_LN_RES || residual != nullptr, "residual missing");
        GF_CHECK_ARG(row_flag == nullptr || flag_rows > 0, "flag_rows must be > 0");
    }
    LinArgs a{a1, a2, lda1, lda2, k1, k2, w, bias, rowgroup_bias, rowgroup_rows, ln_gamma, ln_beta, ln_eps, residual,
              ldres, row_flag, flag_rows, out, ldo, M, N};
    hipStream_t st = (hipStream_t)stream;
    // declared work = the algorithmic BYTES: with K = 128 ... 512 these GEMMs sit below the machine balance (N K / (N + K) = 64 ... 170
    // flop per byte against 2.5 PFLOP/s : 8 TB/s = 312), so HBM is the roof that bounds them: operand rows + weights + output
    // (+ residual) once
    void* pt = gf_prof_begin("k3_linear", st, (double)es * ((double)M * (k1 + k2) + (double)N * (k1 + k2) + (double)M * N * (residual ? 2.0 : 1.0)));
    const bool wide = (epilogue >= EPI_LN) ? N == 256 : N % 256 == 0;
    if (dtype != GF_F32 && wide && epilogue <= EPI_TANH && rowgroup_bias == nullptr) {
        const size_t lds = (size_t)(128 + 256) * 128 + 256 * sizeof(float);
        const dim3 grid((M + 127) / 128, (N + 255) / 256);
#define GF_W2(T)                                                                   \
    do {                                                                           \
        if (epilogue == EPI_NONE) linear_kernel_w2<T, EPI_NONE><<<grid, 256, lds, st>>>(a);       \
        else if (epilogue == EPI_RELU) linear_kernel_w2<T, EPI_RELU><<<grid, 256, lds, st>>>(a);  \
        else linear_kernel_w2<T, EPI_TANH><<<grid, 256, lds, st>>>(a);                            \
    } while (0)
        if (dtype == GF_F16) GF_W2(_Float16);
        else GF_W2(gf_bf16);
#undef GF_W2
    } else if (dtype == GF_F32) {
        if (wide) lin_launch<float, 8>(a, epilogue, st);
        else lin_launch<float, 4>(a, epilogue, st);
    } else if (dtype == GF_F16) {
        if (wide) lin_launch<_Float16, 8>(a, epilogue, st);
        else lin_launch<_Float16, 4>(a, epilogue, st);
    } else {
        if (wide) lin_launch<gf_bf16, 8>(a, epilogue, st);
        else lin_launch<gf_bf16, 4>(a, epilogue, st);
    }
    gf_prof_end("k3_linear", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// backbone glue: 1x1 lateral convolution of the FPN with the top-down merge fused into its epilogue
//   out[n,y,x,:] = w . x[n,y,x,:] + bilinear(lo -> HxW, align_corners=True)[n,y,x,:]
extern "C" int gf_conv1x1_upsample_add_nhwc(const void* x, const void* w, const void* lo, void* out, int N, int h, int wl,
                                            int H, int W, int Cin, int Cout, int dtype, void* stream) {
    GF_CHECK_ARG(x && w && lo && out, "null pointer");
    GF_CHECK_ARG(N > 0 && h > 0 && wl > 0 && H > 0 && W > 0, "empty problem");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit maps (the inference backbone)");
    GF_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0, "Cin and Cout must be multiples of 32");
    GF_CHECK_ARG((long)N * H * W < (1l << 31), "too many pixels");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)lo % 16 == 0, "tensors must be 16-byte aligned");
    LinArgs a{};
    a.a1 = x; a.lda1 = Cin; a.k1 = Cin; a.w = w; a.out = out; a.ldo = Cout; a.M = N * H * W; a.N = Cout;
    a.up_lo = lo; a.up_h = h; a.up_w = wl; a.up_H = H; a.up_W = W; a.up_n = N;
    a.up_ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    a.up_rx = W > 1 ? (float)(wl - 1) / (float)(W - 1) : 0.f;
    hipStream_t st = (hipStream_t)stream;
    // backbone (f4), not the matching path: own tag.  Declared work = algorithmic bytes (x + out + the coarser map + weights once):
    // 67 flop per byte, HBM is the roof
    void* pt = gf_prof_begin("k3_upadd", st, 2.0 * ((double)a.M * (Cin + Cout) + (double)N * h * wl * Cout + (double)Cin * Cout));
    // 128-wide column tiles (152 registers, three waves per SIMD): with K = 128 the kernel is all epilogue, and
    // a single 224-wide tile (NB = 7, two waves per SIMD) measured slower (607 vs 529 us) despite reading x once
    if (dtype == GF_F16) lin_launch1<_Float16, 4, EPI_UPADD, true>(a, st);
    else lin_launch1<gf_bf16, 4, EPI_UPADD, true>(a, st);
    gf_prof_end("k3_upadd", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// 1x1 convolution of a channels-last 16-bit map, stride 1 or 2, no bias (the backbone's lateral and downsample-shortcut
// convolutions with the BatchNorm scale folded into w; a shift is added by whoever consumes the map): out[n, y, x, :] =
// W x[n, s y, s x, :] - the K3 tile engine on the strided pixel rows
extern "C" int gf_conv1x1_nhwc(const void* x, const void* w, void* out, int N, int H, int W, int Cin, int Cout, int stride, int dtype,
                               void* stream) {
    GF_CHECK_ARG(x && w && out, "null pointer");
    GF_CHECK_ARG(N > 0 && H > 0 && W > 0, "empty problem");
    GF_CHECK_ARG(stride == 1 || (stride == 2 && H % 2 == 0 && W % 2 == 0), "stride 1, or 2 on even-sized maps");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit maps (the inference backbone)");
    GF_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0, "Cin and Cout must be multiples of 32");
    GF_CHECK_ARG((long)N * H * W < (1l << 31), "too many pixels");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)w % 16 == 0, "tensors must be 16-byte aligned");
    const int oh = H / stride, ow = W / stride;
    LinArgs a{};
    a.a1 = x; a.lda1 = Cin; a.k1 = Cin; a.w = w; a.out = out; a.ldo = Cout; a.M = N * oh * ow; a.N = Cout;
    if (stride == 2) { a.rm_w = ow; a.rm_line = 2l * W * Cin; a.rm_pix = 2l * Cin; }
    hipStream_t st = (hipStream_t)stream;
    void* pt = gf_prof_begin("conv1x1", st, 2.0 * ((double)a.M * (Cin + Cout) + (double)Cin * Cout));
    if (dtype == GF_F16) lin_launch1<_Float16, 4, EPI_NONE, true>(a, st);
    else lin_launch1<gf_bf16, 4, EPI_NONE, true>(a, st);
    gf_prof_end("conv1x1", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
