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
                    kv = elu1(gf_to_float(kp[(size_t)s * a.ldk + t]));
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
                if (ok) qv = elu1(gf_to_float(qp[(size_t)l * a.ldq + t]));
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

template <typename T, int D>
int la_launch(const LaArgs& a, hipStream_t st) {
    const size_t lds = (size_t)TOK * a.C * sizeof(float);
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
    GF_CHECK_ARG(dtype == GF_F32 || dtype == GF_F16, "dtype must be GF_F32 or GF_F16");
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
#define GF_LA(T)                                       \
    (D == 16 ? la_launch<T, 16>(a, st) : D == 32 ? la_launch<T, 32>(a, st) : la_launch<T, 64>(a, st))
    return dtype == GF_F32 ? GF_LA(float) : GF_LA(_Float16);
#undef GF_LA
}
