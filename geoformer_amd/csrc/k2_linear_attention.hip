// K2: linear attention (elu+1 feature map, D x D key-value state per head) - replaces
// LinearAttention.forward (model/loftr_src/loftr/loftr_module/linear_attention.py:21-51).
//
//   Q = elu(q)+1, K = elu(k)+1 (masked rows -> 0), V = v / S
//   KV[n,h] = sum_s K[s,h,:]^T V[s,h,:]          (D x D, fp32)     | kernel la_kv_partial + la_kv_final
//   Ksum[n,h] = sum_s K[s,h,:]
//   out[l,h,:] = (Q[l,h,:] . KV[n,h]) * 1/(Q[l,h,:].Ksum[n,h] + eps) * S        | kernel la_apply
//
// HBM-bound (reads q,k,v once, writes out once); the D x D state never leaves registers/LDS/L2.
// One thread per channel c = h*D + d; C = H*D <= 256.
#include <math.h>

#include <type_traits>

#include "gf_common.h"

namespace {

constexpr int TOK = 16;        // tokens staged per step
constexpr int CHUNK = 128;     // tokens per workgroup in the KV reduction

struct LaArgs {
    const void *q, *k, *v;
    void* out;
    int N, L, S, H, D, C;
    long ldq, ldk, ldv;      // elements between consecutive tokens
    const uint8_t* q_mask;
    const uint8_t* kv_mask;
    float eps;
    float* kvpart;           // [N][nchunks][C*D + C]
    float* kvfinal;          // [N][C*D + C]
    int nchunks;
};

// F.elu(x) + 1 evaluated like torch: expm1 first, then the add
__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expm1f(x)) + 1.0f; }
// fp16 path: elu(x) + 1 = exp(x) for x <= 0; the hardware exponential is exact to fp16 rounding and ~20 VALU
// instructions cheaper than expm1f (which made la16_kv VALU-bound: 32 of them per thread and sub-tile)
// (both sides are evaluated and selected: a conditional exponential compiles to a divergent branch per element)
__device__ __forceinline__ float elu1_fast(float x) {
    const float e = __expf(fminf(x, 0.f));
    return x > 0.f ? x + 1.0f : e;
}
template <typename T>
__device__ __forceinline__ float elu1_t(float x) {          // parity (fp32) mode keeps torch's evaluation
    if constexpr (std::is_same<T, float>::value) return elu1(x);
    else return elu1_fast(x);
}

template <typename T, int D>
__global__ __launch_bounds__(256) void la_kv_partial(LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vrow = reinterpret_cast<float*>(smem);   // [TOK][C]
    const int chunk = blockIdx.x, n = blockIdx.y, t = threadIdx.x, C = a.C;
    const int h0 = (t / D) * D;
    const int s_begin = chunk * CHUNK, s_end = min(a.S, s_begin + CHUNK);
    const T* kp = (const T*)a.k + (size_t)n * a.S * a.ldk;
    const T* vp = (const T*)a.v + (size_t)n * a.S * a.ldv;
    const float slen = (float)a.S;
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
    float ksum = 0.f;
    for (int s0 = s_begin; s0 < s_end; s0 += TOK) {
        float kk[TOK];
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const int s = s0 + j;
            float kv = 0.f, vv = 0.f;
            if (s < s_end) {
                const bool ok = a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + s] != 0;
                if (ok) {
                    kv = elu1_t<T>(gf_to_float(kp[(size_t)s * a.ldk + t]));
                    vv = gf_to_float(vp[(size_t)s * a.ldv + t]) / slen;   // values / v_length (:45)
                }
            }
            kk[j] = kv;
            vrow[j * C + t] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const float kv = kk[j];
            ksum += kv;
            const v4f* vr = reinterpret_cast<const v4f*>(vrow + j * C + h0);
#pragma unroll
            for (int i = 0; i < D / 4; ++i) {
                const v4f x = vr[i];
                acc[4 * i + 0] += kv * x.x;
                acc[4 * i + 1] += kv * x.y;
                acc[4 * i + 2] += kv * x.z;
                acc[4 * i + 3] += kv * x.w;
            }
        }
        __syncthreads();
    }
    float* dst = (a.nchunks == 1 ? a.kvfinal + (size_t)n * (C * D + C)
                                 : a.kvpart + ((size_t)n * a.nchunks + chunk) * (C * D + C));
    // layout: [v][c] so that the apply kernel's thread (h,v) reads KV[h][d][v] for d = 0..D-1 coalesced
#pragma unroll
    for (int i = 0; i < D; ++i) dst[i * C + t] = acc[i];     // element (c = h*D+d, v = i)
    dst[C * D + t] = ksum;
}

__global__ void la_kv_final(LaArgs a) {
    const int n = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, len = a.C * a.D + a.C;
    if (i >= len) return;
    const float* p = a.kvpart + (size_t)n * a.nchunks * len + i;
    float s = 0.f;
    for (int c = 0; c < a.nchunks; ++c) s += p[(size_t)c * len];
    a.kvfinal[(size_t)n * len + i] = s;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void la_apply(LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qrow = reinterpret_cast<float*>(smem);   // [TOK][C]
    const int n = blockIdx.y, t = threadIdx.x, C = a.C;
    const int h0 = (t / D) * D, vi = t % D;
    const float* kvf = a.kvfinal + (size_t)n * (C * D + C);
    float kv[D], ks[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        kv[d] = kvf[vi * C + h0 + d];      // KV[h][d][v]
        ks[d] = kvf[C * D + h0 + d];
    }
    const T* qp = (const T*)a.q + (size_t)n * a.L * a.ldq;
    T* op = (T*)a.out + (size_t)n * a.L * C;
    const float slen = (float)a.S;
    const int l_begin = blockIdx.x * CHUNK, l_end = min(a.L, l_begin + CHUNK);
    for (int l0 = l_begin; l0 < l_end; l0 += TOK) {
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const int l = l0 + j;
            float qv = 0.f;
            if (l < l_end) {
                const bool ok = a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + l] != 0;
                if (ok) qv = elu1_t<T>(gf_to_float(qp[(size_t)l * a.ldq + t]));
            }
            qrow[j * C + t] = qv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const int l = l0 + j;
            const v4f* qr = reinterpret_cast<const v4f*>(qrow + j * C + h0);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int i = 0; i < D / 4; ++i) {
                const v4f x = qr[i];
                num += x.x * kv[4 * i] + x.y * kv[4 * i + 1] + x.z * kv[4 * i + 2] + x.w * kv[4 * i + 3];
                den += x.x * ks[4 * i] + x.y * ks[4 * i + 1] + x.z * ks[4 * i + 2] + x.w * ks[4 * i + 3];
            }
            const float z = 1.0f / (den + a.eps);
            if (l < l_end) op[(size_t)l * C + t] = gf_from_float<T>(num * z * slen);
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// fp16 path: both contractions on the matrix cores.
//
//   la16_kv  : per 128-token chunk, per head:  KV[d][v] += sum_tok K[tok][d] V[tok][v]  is an MFMA whose
//              K dimension is the TOKEN: the 32-token sub-tiles go to LDS row-major and the operands are
//              fetched with the transpose read ds_read_b64_tr_b16 (la16_tr_frag); Ksum comes
//              from the same A operand against a ones B operand.  V is left unscaled in the fp32 state
//              (v/S would be an fp16 subnormal); the 1/S scaling is applied when the state is cast to fp16.
//   la16_apply: computed transposed, out^T[v][tok] = KV^T . Q^T, so the TOKEN sits on the lane: the
//              numerator rows, the denominator (same MFMA against Ksum replicated over the rows) and the
//              1/(den+eps) scaling are lane-local; Q fragments are 16-B loads straight from the token row.
// ---------------------------------------------------------------------------------------------
constexpr int LT = 32;     // tokens per sub-tile


// [32 tok][576 B] row-major sub-tile image (512 B of channels + 64 B pad): 16-B vector writes, and the
// token-major MFMA operand is read with the gfx950 transpose read (ds_read_b64_tr_b16: a 16-lane group
// fetches a 4-token x 16-channel block and each lane receives one channel's 4 tokens).  With the 576-B
// stride the 16 (token, channel-quad) addresses of both groups of a wave half fall on distinct banks.
constexpr int TRS = 576;

// operand fragment of one head for the 16-token k-step s2: lane (lr = channel, h2) gets tokens 16*s2 + 8*h2 + 0..7
template <typename H>
__device__ __forceinline__ gf_vec<H, 8> la16_tr_frag(const char* img, int head_ch0, int s2, int lane) {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int tok0 = 16 * s2 + 8 * (G >> 1);
    const char* base = img + (tok0 + q) * TRS + (head_ch0 + 16 * (G & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base));
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base + 4 * TRS));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(gf_vec<H, 8>, both);
}

template <typename H, int D>
__global__ __launch_bounds__(256) void la16_kv(LaArgs a) {
    using V8 = gf_vec<H, 8>;
    static_assert(D == 32, "coarse configuration");
    __shared__ __attribute__((aligned(16))) char kt[LT * TRS];
    __shared__ __attribute__((aligned(16))) char vt[LT * TRS];
    const int chunk = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, lr = lane & 31, C = a.C;
    const int s_begin = chunk * CHUNK, s_end = min(a.S, s_begin + CHUNK);
    const H* kp = (const H*)a.k + (size_t)n * a.S * a.ldk;
    const H* vp = (const H*)a.v + (size_t)n * a.S * a.ldv;
    v16f acc[2], ksum[2];                        // this wave's two heads: 2*wave, 2*wave+1
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; ksum[i][r] = 0.f; }
    const V8 ones{(H)1, (H)1, (H)1, (H)1, (H)1, (H)1, (H)1, (H)1};
    const V8 zero{(H)0, (H)0, (H)0, (H)0, (H)0, (H)0, (H)0, (H)0};
    V8 rk[4], rv[4];                            // the next sub-tile, in flight while this one is multiplied
    auto fetch = [&](int s0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = p * 256 + tid, tok = e >> 5, c8 = (e & 31) * 8;      // C = 256: 32 chunks per token
            const int s = s0 + tok;
            rk[p] = zero;
            rv[p] = zero;
            if (s < s_end && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + s] != 0)) {
                rk[p] = *reinterpret_cast<const V8*>(kp + (size_t)s * a.ldk + c8);
                rv[p] = *reinterpret_cast<const V8*>(vp + (size_t)s * a.ldv + c8);
            }
        }
    };
    fetch(s_begin);
    for (int s0 = s_begin; s0 < s_end; s0 += LT) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = p * 256 + tid, tok = e >> 5, c8 = (e & 31) * 8;
            const bool ok = s0 + tok < s_end && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + s0 + tok] != 0);
            V8 kk = rk[p];
#pragma unroll
            for (int i = 0; i < 8; ++i) kk[i] = ok ? (H)elu1_fast((float)kk[i]) : (H)0;
            *reinterpret_cast<V8*>(kt + tok * TRS + c8 * 2) = kk;
            *reinterpret_cast<V8*>(vt + tok * TRS + c8 * 2) = rv[p];
        }
        __syncthreads();
        if (s0 + LT < s_end) fetch(s0 + LT);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch0 = (2 * wave + i) * D;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const V8 kf = la16_tr_frag<H>(kt, ch0, s2, lane);
                const V8 vf = la16_tr_frag<H>(vt, ch0, s2, lane);
                Mma32<H>::mma(kf, vf, acc[i]);
                Mma32<H>::mma(kf, ones, ksum[i]);
            }
        }
        __syncthreads();
    }
    float* dst = (a.nchunks == 1 ? a.kvfinal + (size_t)n * (C * D + C)
                                 : a.kvpart + ((size_t)n * a.nchunks + chunk) * (C * D + C));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int h = 2 * wave + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = gf_acc_row(r, h2);      // rows = d, column = v (lane)
            dst[lr * C + h * D + d] = acc[i][r];
            if (lr == 0) dst[C * D + h * D + d] = ksum[i][r];
        }
    }
}

template <typename H, int D>
__global__ __launch_bounds__(256) void la16_apply(LaArgs a) {
    using V8 = gf_vec<H, 8>;
    using V4 = gf_vec<H, 4>;
    static_assert(D == 32, "coarse configuration");
    constexpr int RS = 272;                                              // slab row: 4 heads x 64 B + pad
    __shared__ __attribute__((aligned(16))) H kvt[8 * 32 * 32];   // [h][v][d]
    __shared__ __attribute__((aligned(16))) H ksh[8 * 32];        // [h][d]
    __shared__ __attribute__((aligned(16))) char qs[4 * 32 * RS];        // per wave: 32 tokens x 4 heads of Q, then of out
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, lr = lane & 31;
    const int C = a.C;
    const float* kvf = a.kvfinal + (size_t)n * (C * D + C);
    // the state is brought to fp16 scaled by 1/S (the reference's "prevent fp16 overflow" scaling, :45):
    // out = (Q.KV/S) / (Q.Ksum/S + eps/S), identical to (Q.KV/S) * 1/(Q.Ksum + eps) * S
    const float inv_s = 1.0f / (float)a.S;
    for (int e = tid; e < C * D; e += 256) {      // kvf element (v, c = h*D + d) at [v*C + c]
        const int v = e / C, c = e % C;
        kvt[((c / D) * 32 + v) * 32 + (c % D)] = (H)(kvf[e] * inv_s);
    }
    ksh[tid] = (H)(kvf[C * D + tid] * inv_s);
    __syncthreads();
    // Q rows come in with 16-B-per-lane row-contiguous loads (4 heads = 256 B per token at a time) through a
    // wave-private LDS slab; the MFMA B fragments are 16-B row reads of it; the result overwrites the head's
    // own 64 bytes and leaves with the same row-contiguous pattern.
    char* qt = qs + wave * 32 * RS;
    const int tok0 = blockIdx.x * 128 + wave * 32;
    const int prow = lane >> 4, pch = lane & 15;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + prow, tok = min(tok0 + row, a.L - 1);
            const bool qok = a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + tok] != 0;
            V8 qv = *reinterpret_cast<const V8*>((const H*)a.q + ((size_t)n * a.L + tok) * a.ldq + half * 128 + pch * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) qv[i] = qok ? (H)elu1_fast((float)qv[i]) : (H)0;
            *reinterpret_cast<V8*>(qt + row * RS + pch * 16) = qv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            const int h = half * 4 + hh;
            v16f num, den;
#pragma unroll
            for (int r = 0; r < 16; ++r) { num[r] = 0.f; den[r] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const V8 qf = *reinterpret_cast<const V8*>(qt + lr * RS + (hh * 4 + 2 * s2 + h2) * 16);   // B: col = token, k = d
                const V8 kf = *reinterpret_cast<const V8*>(kvt + (h * 32 + lr) * 32 + 16 * s2 + 8 * h2);   // A: row = v
                const V8 sf = *reinterpret_cast<const V8*>(ksh + h * 32 + 16 * s2 + 8 * h2);              // A: every row = Ksum
                Mma32<H>::mma(kf, qf, num);
                Mma32<H>::mma(sf, qf, den);
            }
            const float z = __builtin_amdgcn_rcpf(den[0] + a.eps * inv_s);     // every row of den holds the token's Q.Ksum / S
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<V4*>(qt + lr * RS + (hh * 32 + 8 * r4 + 4 * h2) * 2) =
                    V4{(H)(num[4 * r4] * z), (H)(num[4 * r4 + 1] * z), (H)(num[4 * r4 + 2] * z), (H)(num[4 * r4 + 3] * z)};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + prow, tok = tok0 + row;
            if (tok < a.L)
                *reinterpret_cast<V8*>((H*)a.out + ((size_t)n * a.L + tok) * C + half * 128 + pch * 8) =
                    *reinterpret_cast<const V8*>(qt + row * RS + pch * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename H>
int la16_launch(const LaArgs& a, hipStream_t st) {
    la16_kv<H, 32><<<dim3(a.nchunks, a.N), 256, 0, st>>>(a);
    if (a.nchunks > 1) {
        const int len = a.C * a.D + a.C;
        la_kv_final<<<dim3((len + 255) / 256, a.N), 256, 0, st>>>(a);
    }
    la16_apply<H, 32><<<dim3((a.L + 127) / 128, a.N), 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// ---------------------------------------------------------------------------------------------
// short sequences (S, L <= CHUNK: the fine level has 25-token windows and ~2M of them): the whole
// attention of one sample in ONE workgroup - the D x D state goes from the KV phase to the apply phase
// through LDS instead of a global round trip and two more launches.  Same summation order as
// la_kv_partial / la_apply with one chunk, so the results are bit-identical to that path.
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void la_small(LaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = reinterpret_cast<float*>(smem);                 // [TOK][C] staging (V rows, then Q rows)
    float* kvs = row + TOK * a.C;                                // [C][D]: element (c = h*D + d, v)
    float* kss = kvs + a.C * D;                                  // [C]
    const int n = blockIdx.x, t = threadIdx.x, C = a.C;
    const int h0 = (t / D) * D, vi = t % D;
    const T* kp = (const T*)a.k + (size_t)n * a.S * a.ldk;
    const T* vp = (const T*)a.v + (size_t)n * a.S * a.ldv;
    const float slen = (float)a.S;
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
    float ksum = 0.f;
    for (int s0 = 0; s0 < a.S; s0 += TOK) {
        float kk[TOK];
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const int s = s0 + j;
            float kv = 0.f, vv = 0.f;
            if (s < a.S && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + s] != 0)) {
                kv = elu1_t<T>(gf_to_float(kp[(size_t)s * a.ldk + t]));
                vv = gf_to_float(vp[(size_t)s * a.ldv + t]) / slen;
            }
            kk[j] = kv;
            row[j * C + t] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const float kv = kk[j];
            ksum += kv;
            const v4f* vr = reinterpret_cast<const v4f*>(row + j * C + h0);
#pragma unroll
            for (int i = 0; i < D / 4; ++i) {
                const v4f x = vr[i];
                acc[4 * i + 0] += kv * x.x;
                acc[4 * i + 1] += kv * x.y;
                acc[4 * i + 2] += kv * x.z;
                acc[4 * i + 3] += kv * x.w;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < D; ++i) kvs[t * D + i] = acc[i];
    kss[t] = ksum;
    __syncthreads();
    float kv[D], ks[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        kv[d] = kvs[(h0 + d) * D + vi];      // KV[h][d][v]
        ks[d] = kss[h0 + d];
    }
    const T* qp = (const T*)a.q + (size_t)n * a.L * a.ldq;
    T* op = (T*)a.out + (size_t)n * a.L * C;
    for (int l0 = 0; l0 < a.L; l0 += TOK) {
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const int l = l0 + j;
            float qv = 0.f;
            if (l < a.L && (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + l] != 0))
                qv = elu1_t<T>(gf_to_float(qp[(size_t)l * a.ldq + t]));
            row[j * C + t] = qv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TOK; ++j) {
            const int l = l0 + j;
            const v4f* qr = reinterpret_cast<const v4f*>(row + j * C + h0);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int i = 0; i < D / 4; ++i) {
                const v4f x = qr[i];
                num += x.x * kv[4 * i] + x.y * kv[4 * i + 1] + x.z * kv[4 * i + 2] + x.w * kv[4 * i + 3];
                den += x.x * ks[4 * i] + x.y * ks[4 * i + 1] + x.z * ks[4 * i + 2] + x.w * ks[4 * i + 3];
            }
            const float z = 1.0f / (den + a.eps);
            if (l < a.L) op[(size_t)l * C + t] = gf_from_float<T>(num * z * slen);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit fine level (8 heads of 16, windows of <= 32 tokens, ~2M of them) on the matrix cores: ONE WAVE per window.
// la_small is VALU-bound there (0.48 ms per 37 k windows = 2 TB/s; a one-wave-per-window VALU form measured the same);
// here the window's q / k / v move 16 bytes per lane, phi(K), V and phi(Q) sit row-major in wave-private LDS images, and
//   state   KV_h[d][v] = sum_tok phi(K)[tok][h,d] V[tok][h,v] : one v_mfma_f32_16x16x32 per head, its token-major operands
//           fetched with the transpose read ds_read_b64_tr_b16; Ksum_h = the same A operand against a ones B operand;
//   apply   out^T[v][tok] = (KV_h / S)^T phi(Q)^T and the normaliser (Ksum_h / S replicated over the rows) : v_mfma_f32_16x16x16,
//           whose A operand is exactly the register layout the state's accumulator has (column v on the lane, rows 4 g + i);
//           the token is on the lane, so num / (den + eps / S) is lane-local; the result overwrites the 8 bytes of phi(Q)
//           the lane just consumed and leaves in 16-byte row segments.
// Rounding points (oracle: linear_attention_window): phi(q), phi(k) -> storage type; KV / S, Ksum / S -> storage type;
// accumulation and the division in fp32; the message is rounded once.
// ---------------------------------------------------------------------------------------------
constexpr int LW_RS = 272, LW_IMG = 32 * LW_RS, LW_WAVE = 3 * LW_IMG;      // K | V | Q images of one wave: 26,112 B

template <typename T>
struct LwMma;
template <>
struct LwMma<_Float16> {
    static __device__ __forceinline__ v4f k32(const v8h& a, const v8h& b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4f k16(const v4h& a, const v4h& b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};
template <>
struct LwMma<gf_bf16> {
    static __device__ __forceinline__ v4f k32(const v8b& a, const v8b& b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4f k16(const v4b& a, const v4b& b, v4f c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(gf_v4s, a), __builtin_bit_cast(gf_v4s, b), c, 0, 0, 0);
    }
};

// 16x16x32 operand of one head from a [32 tok][LW_RS] image: lane (i = channel, G) gets tokens 8 G .. 8 G + 7
template <typename T>
__device__ __forceinline__ gf_vec<T, 8> lw_tr_frag(const char* img, int ch0, int lane) {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const char* base = img + (8 * G + q) * LW_RS + (ch0 + 4 * p) * 2;
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base));
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base + 4 * LW_RS));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    return __builtin_bit_cast(gf_vec<T, 8>, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <typename T>
__global__ __launch_bounds__(128) void la_window_mfma(LaArgs a) {
    using V8 = gf_vec<T, 8>;
    using V4 = gf_vec<T, 4>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 2 + wave;
    if (n >= a.N) return;                                               // wave-uniform; no workgroup barrier below
    char* kimg = smem + wave * LW_WAVE;
    char* vimg = kimg + LW_IMG;
    char* qimg = vimg + LW_IMG;
    const T* kp = (const T*)a.k + (size_t)n * a.S * a.ldk;
    const T* vp = (const T*)a.v + (size_t)n * a.S * a.ldv;
    const T* qp = (const T*)a.q + (size_t)n * a.L * a.ldq;
    V8 kr[8], vr[8], qr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i, row = e >> 4, ch = e & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) { kr[i][j] = (T)0.f; vr[i][j] = (T)0.f; qr[i][j] = (T)0.f; }
        if (row < a.S) {
            kr[i] = *reinterpret_cast<const V8*>(kp + (size_t)row * a.ldk + ch * 8);
            vr[i] = *reinterpret_cast<const V8*>(vp + (size_t)row * a.ldv + ch * 8);
        }
        if (row < a.L) qr[i] = *reinterpret_cast<const V8*>(qp + (size_t)row * a.ldq + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i, row = e >> 4, ch = e & 15;
        const bool kk = row < a.S && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + row] != 0);
        const bool qk = row < a.L && (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + row] != 0);
        V8 pk, pq, pv = vr[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pk[j] = kk ? gf_from_float<T>(elu1_fast(gf_to_float(kr[i][j]))) : (T)0.f;
            pq[j] = qk ? gf_from_float<T>(elu1_fast(gf_to_float(qr[i][j]))) : (T)0.f;
            if (!kk) pv[j] = (T)0.f;
        }
        *reinterpret_cast<V8*>(kimg + row * LW_RS + ch * 16) = pk;
        *reinterpret_cast<V8*>(vimg + row * LW_RS + ch * 16) = pv;
        *reinterpret_cast<V8*>(qimg + row * LW_RS + ch * 16) = pq;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float inv_s = 1.0f / (float)a.S, eps_s = a.eps * inv_s;
    const int G = lane >> 4, i16 = lane & 15;
    V8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const V8 af = lw_tr_frag<T>(kimg, h * 16, lane), bf = lw_tr_frag<T>(vimg, h * 16, lane);
        const v4f kv = LwMma<T>::k32(af, bf, zero4);                   // KV_h[d = 4 G + i][v = i16]
        const v4f ks = LwMma<T>::k32(af, ones, zero4);                 // Ksum_h[d = 4 G + i] in every column
        const V4 kvp = {gf_from_float<T>(kv[0] * inv_s), gf_from_float<T>(kv[1] * inv_s), gf_from_float<T>(kv[2] * inv_s), gf_from_float<T>(kv[3] * inv_s)};
        const V4 ksp = {gf_from_float<T>(ks[0] * inv_s), gf_from_float<T>(ks[1] * inv_s), gf_from_float<T>(ks[2] * inv_s), gf_from_float<T>(ks[3] * inv_s)};
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            if (tb * 16 < a.L) {
                char* qa = qimg + (tb * 16 + i16) * LW_RS + (h * 16 + 4 * G) * 2;   // phi(Q)[tok = 16 tb + i16][h, 4 G .. + 3]
                const V4 bq = *reinterpret_cast<const V4*>(qa);
                const v4f num = LwMma<T>::k16(kvp, bq, zero4), den = LwMma<T>::k16(ksp, bq, zero4);
                const float z = __builtin_amdgcn_rcpf(den[0] + eps_s);  // (every row of den holds the token's normaliser)
                *reinterpret_cast<V4*>(qa) = V4{gf_from_float<T>(num[0] * z), gf_from_float<T>(num[1] * z), gf_from_float<T>(num[2] * z),
                                                gf_from_float<T>(num[3] * z)};
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T* op = (T*)a.out + (size_t)n * a.L * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i, row = e >> 4, ch = e & 15;
        if (row < a.L) *reinterpret_cast<V8*>(op + (size_t)row * 128 + ch * 8) = *reinterpret_cast<const V8*>(qimg + row * LW_RS + ch * 16);
    }
}

template <typename T, int D>
int la_launch(const LaArgs& a, hipStream_t st) {
    const size_t lds = (size_t)TOK * a.C * sizeof(float);
    if constexpr (!std::is_same<T, float>::value && D == 16) {
        if (a.C == 128 && a.S <= 32 && a.L <= 32 && a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 &&
            (uintptr_t)a.q % 16 == 0 && (uintptr_t)a.k % 16 == 0 && (uintptr_t)a.v % 16 == 0 && (uintptr_t)a.out % 16 == 0) {
            static std::atomic<uint64_t> attr{0};
            if (gf_first_use_on_device(attr))
                (void)hipFuncSetAttribute((const void*)la_window_mfma<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LW_WAVE);
            la_window_mfma<T><<<(a.N + 1) / 2, 128, 2 * LW_WAVE, st>>>(a);
            GF_CHECK_LAUNCH();
            return GF_OK;
        }
    }
    if (a.S <= CHUNK && a.L <= CHUNK) {
        la_small<T, D><<<a.N, a.C, lds + (size_t)a.C * (D + 1) * sizeof(float), st>>>(a);
        GF_CHECK_LAUNCH();
        return GF_OK;
    }
    la_kv_partial<T, D><<<dim3(a.nchunks, a.N), a.C, lds, st>>>(a);
    if (a.nchunks > 1) {
        const int len = a.C * a.D + a.C;
        la_kv_final<<<dim3((len + 255) / 256, a.N), 256, 0, st>>>(a);
    }
    la_apply<T, D><<<dim3((a.L + CHUNK - 1) / CHUNK, a.N), a.C, lds, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

}   // namespace

extern "C" size_t gf_linear_attention_workspace_bytes(int N, int S, int H, int D) {
    if (N <= 0 || S <= 0 || H <= 0 || D <= 0) return 0;
    const size_t len = (size_t)H * D * D + (size_t)H * D;
    const size_t nchunks = (S + CHUNK - 1) / CHUNK;
    return gf_align_up(sizeof(float) * N * len * (nchunks > 1 ? nchunks + 1 : 1), 256) + 256;
}

extern "C" int gf_linear_attention(const void* q, const void* k, const void* v, int dtype, int N, int L, int S, int H,
                                   int D, long ldq, long ldk, long ldv, const uint8_t* q_mask,
                                   const uint8_t* kv_mask, float eps, void* out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(q && k && v && out, "null pointer");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG((D == 16 || D == 32 || D == 64) && H * D <= 256 && (H * D) % 64 == 0, "need D in {16,32,64}, H*D in {64,128,192,256}");
    GF_CHECK_ARG(ldq >= H * D && ldk >= H * D && ldv >= H * D, "row strides smaller than H*D");
    if (workspace == nullptr || workspace_bytes < gf_linear_attention_workspace_bytes(N, S, H, D)) {
        gf_set_error("gf_linear_attention: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    LaArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.N = N; a.L = L; a.S = S; a.H = H; a.D = D; a.C = H * D;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.q_mask = q_mask; a.kv_mask = kv_mask; a.eps = eps;
    a.nchunks = (S + CHUNK - 1) / CHUNK;
    const size_t len = (size_t)a.C * D + a.C;
    a.kvfinal = (float*)workspace;
    a.kvpart = a.kvfinal + (size_t)N * len;
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes: q, k, v read once, the message written once
    void* pt = gf_prof_begin("k2_linear_attention", st, (double)N * (2.0 * L + 2.0 * S) * a.C * (dtype == GF_F32 ? 4 : 2));
    int rc;
    if (dtype == GF_F16 && D == 32 && H == 8) rc = la16_launch<_Float16>(a, st);      // coarse level: matrix-core path
    else if (dtype == GF_BF16 && D == 32 && H == 8) rc = la16_launch<gf_bf16>(a, st);
    else {
#define GF_LA(T)                                       \
    (D == 16 ? la_launch<T, 16>(a, st) : D == 32 ? la_launch<T, 32>(a, st) : la_launch<T, 64>(a, st))
        rc = dtype == GF_F32 ? GF_LA(float) : dtype == GF_F16 ? GF_LA(_Float16) : GF_LA(gf_bf16);
#undef GF_LA
    }
    gf_prof_end("k2_linear_attention", pt, st);
    return rc;
}
