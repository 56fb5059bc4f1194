// Training: the backward of the K3 chain - linear + LayerNorm / ReLU / Tanh - in HIP, for the mixed-16-bit step
// (lightning/train_depth_geoformer.py:117-119 runs the reference under 16-bit autocast; fp32 master weights and gradients).
//
//   y = x W^T            (nn.Linear without bias: every linear of LoFTREncoderLayer / the Geo layers / FinePreprocess' down_proj)
//     dX = dY W          -> gf_linear itself (K3) with the transposed weight: no new kernel
//     dW = dY^T X        -> gf_linear_wgrad below: the contraction runs over the TOKENS, i.e. over the row index of both
//                           row-major operands - the same "TN" shape as the linear-attention state of K2, and the same
//                           machinery: token sub-tiles staged row-major in LDS, MFMA operands fetched with the gfx950
//                           transpose read ds_read_b64_tr_b16, fp32 accumulation, split over token chunks with fp32
//                           partials that a second kernel adds in chunk order (deterministic: no atomics)
//   out = LayerNorm(y)   -> gf_layernorm_forward keeps (mean, rstd) per row; gf_layernorm_backward:
//                           dy = rstd (g - mean_c(g) - xhat mean_c(g xhat)),  g = dout * gamma,  xhat = (y - mean) rstd,
//                           dgamma = sum_rows dout xhat, dbeta = sum_rows dout  (per-workgroup partials + ordered sum)
//   h = act(z)           -> gf_activation_backward: dz = dh * (h > 0)  (ReLU)  |  dh * (1 - h^2)  (Tanh), from the OUTPUT h
//
// The autograd side (which tensors are saved, in which precision) lives in geoformer_amd/train/hip_autograd.py.
#include <math.h>

#include "gf_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// dW[co][ci] = sum_t dY[t][co] X[t][ci]
// Workgroup = 128 x 128 of dW for one chunk of tokens; 4 waves as 2 x 2, each 64 x 64 = four 32 x 32 accumulators.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WG_T = 32;            // tokens per sub-tile (two 16-deep MFMA k-steps)
constexpr int WG_RS = 320;          // LDS row stride of a [32 tok][128 ch] image: 256 B + 64 B (80 dwords: the four token rows of a
                                    // transpose-read group land 16 banks apart, the two groups of a wave half 8 apart - conflict-free)

struct WgArgs {
    const void* dy;
    const void* x;
    long lddy, ldx, T;
    int cout, cin, chunks, chunk_tokens;
    float* part;        // [chunks][cout][cin]
    float* dw;
    long lddw;
    int accumulate;
};

// MFMA operand of 32 channels x the 16 tokens of k-step s2: lane (channel = lane & 31, half h2) gets tokens 16 s2 + 8 h2 + 0..7
template <typename H>
__device__ __forceinline__ gf_vec<H, 8> wg_tr_frag(const char* img, int ch0, int s2, int lane) {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int tok0 = 16 * s2 + 8 * (G >> 1);
    const char* base = img + (tok0 + q) * WG_RS + (ch0 + 16 * (G & 1) + 4 * p) * 2;
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base));
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base + 4 * WG_RS));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(gf_vec<H, 8>, both);
}

template <typename H>
__global__ __launch_bounds__(256) void wgrad_kernel(WgArgs a) {
    __shared__ __attribute__((aligned(16))) char yt[WG_T * WG_RS];
    __shared__ __attribute__((aligned(16))) char xt[WG_T * WG_RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128, chunk = blockIdx.z;
    const long t_begin = (long)chunk * a.chunk_tokens, t_end = t_begin + a.chunk_tokens < a.T ? t_begin + a.chunk_tokens : a.T;
    const H* yp = (const H*)a.dy + m0;
    const H* xp = (const H*)a.x + n0;
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // a sub-tile = 32 tokens x 128 channels = 512 pieces of 16 B per operand: two per thread; the next sub-tile is in flight
    // in registers while this one is multiplied
    const v4u zero{0u, 0u, 0u, 0u};
    v4u ry[2], rx[2];
    auto fetch = [&](long t0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = p * 256 + tid, tok = e >> 4, c8 = (e & 15) * 8;
            const long t = t0 + tok;
            ry[p] = zero;
            rx[p] = zero;
            if (t < t_end) {                                  // (channels behind the matrix: zeros - widths need not be multiples of 128)
                if (m0 + c8 < a.cout) ry[p] = *reinterpret_cast<const v4u*>(yp + t * a.lddy + c8);
                if (n0 + c8 < a.cin) rx[p] = *reinterpret_cast<const v4u*>(xp + t * a.ldx + c8);
            }
        }
    };
    fetch(t_begin);
    for (long t0 = t_begin; t0 < t_end; t0 += WG_T) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = p * 256 + tid, tok = e >> 4, c8 = (e & 15) * 8;
            *reinterpret_cast<v4u*>(yt + tok * WG_RS + c8 * 2) = ry[p];
            *reinterpret_cast<v4u*>(xt + tok * WG_RS + c8 * 2) = rx[p];
        }
        __syncthreads();
        if (t0 + WG_T < t_end) fetch(t0 + WG_T);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const auto a0 = wg_tr_frag<H>(yt, wm * 64, s2, lane), a1 = wg_tr_frag<H>(yt, wm * 64 + 32, s2, lane);
            const auto b0 = wg_tr_frag<H>(xt, wn * 64, s2, lane), b1 = wg_tr_frag<H>(xt, wn * 64 + 32, s2, lane);
            Mma32<H>::mma(a0, b0, acc[0][0]);
            Mma32<H>::mma(a0, b1, acc[0][1]);
            Mma32<H>::mma(a1, b0, acc[1][0]);
            Mma32<H>::mma(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
    // accumulator: row = dY channel (r, h2), lane = X channel: 128-B runs along ci
    const int h2 = lane >> 5, lr = lane & 31;
    float* dst = a.part + ((size_t)chunk * a.cout + m0 + wm * 64) * a.cin + n0 + wn * 64 + lr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (m0 + wm * 64 + i * 32 + gf_acc_row(r, h2) < a.cout && n0 + wn * 64 + j * 32 + lr < a.cin)
                    dst[(size_t)(i * 32 + gf_acc_row(r, h2)) * a.cin + j * 32] = acc[i][j][r];
}

__global__ __launch_bounds__(256) void wgrad_reduce(WgArgs a) {
    const size_t n = (size_t)a.cout * a.cin, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= a.chunks; c += 8) {                  // eight partials in flight, added in chunk order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a.part[(size_t)(c + k) * n + e];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; c < a.chunks; ++c) s += a.part[(size_t)c * n + e];
    float* d = a.dw + (e / a.cin) * a.lddw + e % a.cin;
    *d = a.accumulate ? *d + s : s;
}

#ifndef WG_CHUNK_TARGET
#define WG_CHUNK_TARGET 384
#endif
int wgrad_chunks(long T, int cout, int cin) {
    // about 1.5 workgroups per CU (round 6: 768 -> 384, -13 % at the training shapes: half the partial sums to write and add; tools/wgrad_chunks_time.py),
    // at least 128 tokens per chunk
    const long tiles = (long)((cout + 127) / 128) * ((cin + 127) / 128), want = (WG_CHUNK_TARGET + tiles - 1) / tiles, most = (T + 127) / 128;
    long c = want < most ? want : most;
    return (int)(c < 1 ? 1 : c);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension, one wave per row, C / 64 channels per lane (C in {128, 256, 512}); statistics in fp32
// ---------------------------------------------------------------------------------------------------------------------
struct LnArgs {
    const void* y;        // [T][C] pre-normalisation
    const void* dout;     // [T][C]   (backward)
    const float* gamma;
    const float* beta;
    float eps;
    void* out;            // forward: [T][C];  backward: dy [T][C]
    float* stats;         // [T][2]  (mean, rstd)
    float* part;          // backward: [workgroups][2][C]  dgamma | dbeta partials
    float* dgamma;
    float* dbeta;
    long T;
    int C, parts, accumulate;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <typename H, int PER>
__global__ __launch_bounds__(256) void ln_forward(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.T) return;
    const H* yr = (const H*)a.y + row * a.C + lane * PER;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = gf_to_float(yr[k]); s += v[k]; }
    const float mean = wave_sum(s) * (1.0f / (float)a.C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) q += (v[k] - mean) * (v[k] - mean);
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / (float)a.C) + a.eps);      // biased variance, as nn.LayerNorm
    H* o = (H*)a.out + row * a.C + lane * PER;
#pragma unroll
    for (int k = 0; k < PER; ++k) o[k] = gf_from_float<H>((v[k] - mean) * rstd * a.gamma[lane * PER + k] + a.beta[lane * PER + k]);
    if (lane == 0) { a.stats[2 * row] = mean; a.stats[2 * row + 1] = rstd; }
}

template <typename H, int PER>
__global__ __launch_bounds__(256) void ln_backward(LnArgs a) {
    __shared__ float red[4][2][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ga[PER], dg[PER], db[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { ga[k] = a.gamma[lane * PER + k]; dg[k] = 0.f; db[k] = 0.f; }
    for (long row = (long)blockIdx.x * 4 + wave; row < a.T; row += (long)gridDim.x * 4) {
        const float mean = a.stats[2 * row], rstd = a.stats[2 * row + 1];
        const H* yr = (const H*)a.y + row * a.C + lane * PER;
        const H* dr = (const H*)a.dout + row * a.C + lane * PER;
        float xh[PER], g[PER];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float d = gf_to_float(dr[k]);
            xh[k] = (gf_to_float(yr[k]) - mean) * rstd;
            g[k] = d * ga[k];
            s1 += g[k];
            s2 += g[k] * xh[k];
            dg[k] += d * xh[k];
            db[k] += d;
        }
        const float m1 = wave_sum(s1) * (1.0f / (float)a.C), m2 = wave_sum(s2) * (1.0f / (float)a.C);
        H* o = (H*)a.out + row * a.C + lane * PER;
#pragma unroll
        for (int k = 0; k < PER; ++k) o[k] = gf_from_float<H>(rstd * (g[k] - m1 - xh[k] * m2));
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) { red[wave][0][lane * PER + k] = dg[k]; red[wave][1][lane * PER + k] = db[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * a.C; c += 256) {
        const int which = c / a.C, ch = c - which * a.C;
        a.part[((size_t)blockIdx.x * 2 + which) * a.C + ch] = (red[0][which][ch] + red[1][which][ch]) + (red[2][which][ch] + red[3][which][ch]);
    }
}

// one wave per (dgamma | dbeta, channel): lanes stride over the workgroup partials, then a shuffle tree (fixed order: deterministic)
__global__ __launch_bounds__(256) void ln_param_reduce(LnArgs a) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= 2 * a.C) return;
    const int which = c / a.C, ch = c - which * a.C;
    float s = 0.f;
    for (int p = lane; p < a.parts; p += 64) s += a.part[((size_t)p * 2 + which) * a.C + ch];
    s = wave_sum(s);
    if (lane == 0) {
        float* d = (which ? a.dbeta : a.dgamma) + ch;
        *d = a.accumulate ? *d + s : s;
    }
}

template <typename H>
__global__ void act_backward(const H* dh, const H* h, H* dz, size_t n, int kind) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float hv = gf_to_float(h[e]), d = gf_to_float(dh[e]);
        dz[e] = gf_from_float<H>(kind == 0 ? (hv > 0.f ? d : 0.f) : d * (1.0f - hv * hv));
    }
}

constexpr int LN_BWD_WGS = 512;

}   // namespace

extern "C" size_t gf_linear_wgrad_workspace_bytes(long T, int cout, int cin) {
    if (T <= 0 || cout <= 0 || cin <= 0 || cout % 8 || cin % 8) return 0;
    return gf_align_up(sizeof(float) * (size_t)wgrad_chunks(T, cout, cin) * cout * cin, 256);
}

// dw[co * lddw + ci] (+)= sum_t dy[t * lddy + co] * x[t * ldx + ci]   (fp32), dy [T, cout], x [T, cin] of a 16-bit dtype
extern "C" int gf_linear_wgrad(const void* dy, long lddy, const void* x, long ldx, int dtype, long T, int cout, int cin, float* dw,
                               long lddw, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(dy && x && dw, "null pointer");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit activations (GF_F16 / GF_BF16)");
    GF_CHECK_ARG(T > 0 && cout > 0 && cin > 0 && cout % 8 == 0 && cin % 8 == 0, "cout and cin must be multiples of 8");
    GF_CHECK_ARG((lddy * 2) % 16 == 0 && (ldx * 2) % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0, "rows must be 16-byte aligned");
    GF_CHECK_ARG(lddw >= cin, "lddw < cin");
    if (workspace == nullptr || workspace_bytes < gf_linear_wgrad_workspace_bytes(T, cout, cin)) {
        gf_set_error("gf_linear_wgrad: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    WgArgs a{};
    a.dy = dy; a.x = x; a.lddy = lddy; a.ldx = ldx; a.T = T; a.cout = cout; a.cin = cin;
    a.chunks = wgrad_chunks(T, cout, cin);
    a.chunk_tokens = (int)(((T + a.chunks - 1) / a.chunks + WG_T - 1) / WG_T * WG_T);
    a.chunks = (int)((T + a.chunk_tokens - 1) / a.chunk_tokens);
    a.part = (float*)workspace; a.dw = dw; a.lddw = lddw; a.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    void* pt = gf_prof_begin("wgrad", st, 2.0 * (double)T * cout * cin);
    const dim3 grid((cin + 127) / 128, (cout + 127) / 128, a.chunks);
    if (dtype == GF_F16) wgrad_kernel<_Float16><<<grid, 256, 0, st>>>(a);
    else wgrad_kernel<gf_bf16><<<grid, 256, 0, st>>>(a);
    wgrad_reduce<<<(unsigned)(((size_t)cout * cin + 255) / 256), 256, 0, st>>>(a);
    gf_prof_end("wgrad", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

template <typename H>
static int ln_dispatch(const LnArgs& a, bool backward, hipStream_t st) {
    const unsigned fwd_grid = (unsigned)((a.T + 3) / 4);
    switch (a.C) {
    case 128: if (backward) ln_backward<H, 2><<<a.parts, 256, 0, st>>>(a); else ln_forward<H, 2><<<fwd_grid, 256, 0, st>>>(a); break;
    case 256: if (backward) ln_backward<H, 4><<<a.parts, 256, 0, st>>>(a); else ln_forward<H, 4><<<fwd_grid, 256, 0, st>>>(a); break;
    case 512: if (backward) ln_backward<H, 8><<<a.parts, 256, 0, st>>>(a); else ln_forward<H, 8><<<fwd_grid, 256, 0, st>>>(a); break;
    default: return -1;
    }
    return 0;
}

// out = LayerNorm(y; gamma, beta, eps) over the last dimension, stats[t] = (mean, rstd)   (nn.LayerNorm: biased variance)
extern "C" int gf_layernorm_forward(const void* y, int dtype, long T, int C, const float* gamma, const float* beta, float eps, void* out,
                                    float* stats, void* stream) {
    GF_CHECK_ARG(y && gamma && beta && out && stats, "null pointer");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit activations");
    GF_CHECK_ARG(T > 0 && (C == 128 || C == 256 || C == 512), "C must be 128, 256 or 512");
    LnArgs a{};
    a.y = y; a.gamma = gamma; a.beta = beta; a.eps = eps; a.out = out; a.stats = stats; a.T = T; a.C = C;
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == GF_F16 ? ln_dispatch<_Float16>(a, false, st) : ln_dispatch<gf_bf16>(a, false, st);
    GF_CHECK_ARG(rc == 0, "dispatch failed");
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" size_t gf_layernorm_backward_workspace_bytes(int C) { return C > 0 ? gf_align_up(sizeof(float) * LN_BWD_WGS * 2 * (size_t)C, 256) : 0; }

// dy, dgamma (+)=, dbeta (+)= of out = LayerNorm(y) from dout, the saved y and stats
extern "C" int gf_layernorm_backward(const void* dout, const void* y, const float* stats, int dtype, long T, int C, const float* gamma, void* dy,
                                     float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(dout && y && stats && gamma && dy && dgamma && dbeta, "null pointer");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit activations");
    GF_CHECK_ARG(T > 0 && (C == 128 || C == 256 || C == 512), "C must be 128, 256 or 512");
    if (workspace == nullptr || workspace_bytes < gf_layernorm_backward_workspace_bytes(C)) {
        gf_set_error("gf_layernorm_backward: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    LnArgs a{};
    a.y = y; a.dout = dout; a.gamma = gamma; a.out = dy; a.stats = const_cast<float*>(stats); a.T = T; a.C = C;
    a.part = (float*)workspace; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
    const long rows4 = (T + 3) / 4;
    a.parts = (int)(rows4 < LN_BWD_WGS ? rows4 : LN_BWD_WGS);
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == GF_F16 ? ln_dispatch<_Float16>(a, true, st) : ln_dispatch<gf_bf16>(a, true, st);
    GF_CHECK_ARG(rc == 0, "dispatch failed");
    ln_param_reduce<<<(2 * C + 3) / 4, 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// dz = dh * act'(z) from the activation's OUTPUT h: kind 0 ReLU (h > 0), 1 Tanh (1 - h^2); n elements of a 16-bit dtype
extern "C" int gf_activation_backward(const void* dh, const void* h, void* dz, size_t n, int kind, int dtype, void* stream) {
    GF_CHECK_ARG(dh && h && dz, "null pointer");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit activations");
    GF_CHECK_ARG(kind == 0 || kind == 1, "kind: 0 = ReLU, 1 = Tanh");
    if (n == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == GF_F16) act_backward<_Float16><<<grid, 256, 0, st>>>((const _Float16*)dh, (const _Float16*)h, (_Float16*)dz, n, kind);
    else act_backward<gf_bf16><<<grid, 256, 0, st>>>((const gf_bf16*)dh, (const gf_bf16*)h, (gf_bf16*)dz, n, kind);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// =====================================================================================================================
// K2 (training): backward of LinearAttention.forward (model/loftr_src/loftr/loftr_module/linear_attention.py:21-51)
//   Q = phi(q) [q_mask], K = phi(k) [kv_mask], vs = v [kv_mask] / S, phi = elu + 1
//   KV = sum_s K_s^T vs_s, Ksum = sum_s K_s, den_l = Q_l . Ksum + eps, out_l = (Q_l KV) S / den_l
// given dout:  num = Q KV,  dnum = dout S / den,  dden = -(dout . num) S / den^2
//   dQ = dnum KV^T + dden Ksum,  dKV = sum_l Q_l^T dnum_l,  dKsum = sum_l dden_l Q_l
//   dK_s = vs_s dKV^T + dKsum,   dvs_s = K_s dKV,   dq = dQ phi'(q) [q_mask], dk = dK phi'(k) [kv_mask], dv = dvs / S [kv_mask]
// with phi'(x) = 1 (x > 0) | exp(x) = phi(x) (x <= 0).  Heads of D = 32.  Round 6: every product on v_mfma_f32_32x32x16 (rounds 3-5 ran
// them as fp32 loops over LDS rows: 812 us per 8-image call, the per-source pass alone 477).  All per-token products are computed
// TRANSPOSED - A = the 32 x 32 state (a register fragment per workgroup), B = the token rows as they lie in memory (lane = token) - so
// that the accumulator's column is the lane's own token and everything per token (den, dout . num, phi') is lane-local arithmetic plus
// one exchange between the wave halves; an accumulator is packed as it stands into the next product's operand.  The two state sums
// contract over TOKENS: their operands are transposes of token tiles, read with ds_read_b64_tr_b16 from [token][64 B] LDS images
// (wave-private, no workgroup barrier in the loop); Ksum / dKsum ride along as a product with a column of ones / of dden.  16-bit
// operands (phi(q), phi(k), v / S, the states, dnum, dden rounded to the storage type), fp32 accumulation; per-(image, head) states by
// chunk partials added in chunk order (deterministic).  Five launches: state partials, sum, per-query pass (+ gradient-state partials),
// sum, per-source pass.
// =====================================================================================================================
namespace {

constexpr int LB_D = 32, LB_TOK = 128;                    // head width; the chunk size the workspace is sized for
constexpr int LB_STATE = LB_D * LB_D + LB_D;              // KV [d][v] | Ksum [d]
constexpr int L2_TOK = 256, L2_IMG = 32 * 64;             // tokens per workgroup (4 waves x 2 sub-tiles of 32); a [32 tokens][64 B] image

struct LbArgs {
    const void* q; const void* k; const void* v; const void* dout;
    long ldq, ldk, ldv, ldo;
    const uint8_t* q_mask; const uint8_t* kv_mask;
    void* dq; void* dk; void* dv;                           // [N][L|S][H * 32] contiguous
    float* part; float* state; float* gstate;               // [N * H][chunks][LB_STATE] | [N * H][LB_STATE] x 2
    int N, L, S, H, chunksL, chunksS;
    float eps;
};

__device__ __forceinline__ float lb_phi(float x) { return x > 0.f ? x + 1.f : __expf(x); }

template <typename T>
using LFrag = typename Mma32<T>::Frag;

// the lane's token row of a head: channels 16 g + 8 half .. + 7 (the MFMA operand fragment of k-step g), as floats
template <typename T>
__device__ __forceinline__ void lb_row(const T* p, int h, float (&o)[2][8]) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const LFrag<T> f = *reinterpret_cast<const LFrag<T>*>(p + 16 * g + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = gf_to_float(f[j]);
    }
}
// the same row in ACCUMULATOR order: element r = channel gf_acc_row(r, half) (four 8-byte pieces)
template <typename T>
__device__ __forceinline__ void lb_row_acc(const T* p, int h, float (&o)[16]) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
        const gf_vec<T, 4> f = *reinterpret_cast<const gf_vec<T, 4>*>(p + 8 * j4 + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[4 * j4 + i] = gf_to_float(f[i]);
    }
}
template <typename T>
__device__ __forceinline__ void lb_store_acc(T* p, int h, const float (&o)[16]) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
        *reinterpret_cast<gf_vec<T, 4>*>(p + 8 * j4 + 4 * h) = gf_vec<T, 4>{(T)o[4 * j4], (T)o[4 * j4 + 1], (T)o[4 * j4 + 2], (T)o[4 * j4 + 3]};
}
template <typename T, int S2>
__device__ __forceinline__ LFrag<T> lb_pack_acc(const v16f& x) {       // registers 8 S2 .. 8 S2 + 7 of an accumulator: k-step S2 of the next product
    return LFrag<T>{(T)x[8 * S2], (T)x[8 * S2 + 1], (T)x[8 * S2 + 2], (T)x[8 * S2 + 3], (T)x[8 * S2 + 4], (T)x[8 * S2 + 5], (T)x[8 * S2 + 6],
                    (T)x[8 * S2 + 7]};
}
template <typename T>
__device__ __forceinline__ LFrag<T> lb_pack(const float* x) {
    return LFrag<T>{(T)x[0], (T)x[1], (T)x[2], (T)x[3], (T)x[4], (T)x[5], (T)x[6], (T)x[7]};
}
// transposing fragment of a [32 tokens][64 B] image: lane (channel lane & 31, half) gets tokens 16 s + 8 half + 0..7
template <typename T>
__device__ __forceinline__ LFrag<T> lb_tr(const char* img, int s, int lane) {
    const int G = lane >> 4, i = lane & 15;
    const char* p = img + (16 * s + 8 * (G >> 1) + (i >> 2)) * 64 + (16 * (G & 1) + 4 * (i & 3)) * 2;
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)p);
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(p + 4 * 64));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(LFrag<T>, both);
}
// state operands from the fp32 state [d][v]:  rows(g): lane (row lr, half) elements [lr][16 g + 8 half + j]  (contiguous);
//                                             cols(g): lane (column lr, half) elements [16 g + 8 half + j][lr];
//                                             rows_acc(s): lane (row lr, half) elements [lr][16 s + 8 (i >> 2) + 4 half + (i & 3)]
template <typename T>
__device__ __forceinline__ LFrag<T> lb_state_rows(const float* st, int g, int lr, int h) {
    return lb_pack<T>(st + lr * LB_D + 16 * g + 8 * h);
}
template <typename T>
__device__ __forceinline__ LFrag<T> lb_state_cols(const float* st, int g, int lr, int h) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = st[(16 * g + 8 * h + j) * LB_D + lr];
    return lb_pack<T>(x);
}
template <typename T>
__device__ __forceinline__ LFrag<T> lb_state_rows_acc(const float* st, int s, int lr, int h) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = st[lr * LB_D + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
    return lb_pack<T>(x);
}
// the four waves' accumulators (acc[0]: the 32 x 32 sum, acc[1]: its column 0 = the weighted row sum) added in wave order -> one partial
__device__ __forceinline__ void lb_reduce_store(const v16f (&acc)[2], float* red, float* dst) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 2 + a2) * 16 + r) * 64 + lane] = acc[a2][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, r = e >> 6, ln = e & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * 2) * 16 + r) * 64 + ln];
        dst[gf_acc_row(r, ln >> 5) * LB_D + (ln & 31)] = s;
    }
    if (tid < 32) {
        const int r = tid >> 1, hh = tid & 1;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * 2 + 1) * 16 + r) * 64 + 32 * hh];
        dst[LB_D * LB_D + gf_acc_row(r, hh)] = s;
    }
}

// forward state partials: KV = K^T vs, Ksum = K^T 1 over the workgroup's tokens
template <typename T>
__global__ __launch_bounds__(256) void lb_state_partial(LbArgs a) {
    using M = Mma32<T>;
    __shared__ __attribute__((aligned(16))) char img[4 * 2 * L2_IMG];
    __shared__ float red[4 * 2 * 16 * 64];
    const int chunk = blockIdx.x, nh = blockIdx.y, n = nh / a.H, hh = nh - n * a.H, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    char* kimg = img + wave * 2 * L2_IMG;
    char* vimg = kimg + L2_IMG;
    v16f acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    LFrag<T> ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.f;
    const float inv_s = 1.0f / (float)a.S;
    for (int j2 = 0; j2 < 2; ++j2) {
        const int s = chunk * L2_TOK + (wave + 4 * j2) * 32 + lr;
        float kk[2][8], vv[2][8];
        const bool live = s < a.S && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + s] != 0);
        const size_t row = (size_t)n * a.S + min(s, a.S - 1);
        lb_row<T>((const T*)a.k + row * a.ldk + hh * LB_D, h, kk);
        lb_row<T>((const T*)a.v + row * a.ldv + hh * LB_D, h, vv);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { kk[g][j] = live ? lb_phi(kk[g][j]) : 0.f; vv[g][j] = live ? vv[g][j] * inv_s : 0.f; }
            *reinterpret_cast<LFrag<T>*>(kimg + lr * 64 + (16 * g + 8 * h) * 2) = lb_pack<T>(kk[g]);
            *reinterpret_cast<LFrag<T>*>(vimg + lr * 64 + (16 * g + 8 * h) * 2) = lb_pack<T>(vv[g]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const LFrag<T> kt = lb_tr<T>(kimg, s2, lane);
            M::mma(kt, lb_tr<T>(vimg, s2, lane), acc[0]);          // rows d, columns v
            M::mma(kt, ones, acc[1]);                              // every column: sum over the tokens of K[.][d]
        }
        __builtin_amdgcn_wave_barrier();
    }
    lb_reduce_store(acc, red, a.part + ((size_t)nh * a.chunksS + chunk) * LB_STATE);
}

__global__ __launch_bounds__(256) void lb_state_sum(const float* part, float* state, int chunks) {
    const int nh = blockIdx.x;
    for (int e = threadIdx.x; e < LB_STATE; e += 256) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += part[((size_t)nh * chunks + c) * LB_STATE + e];
        state[(size_t)nh * LB_STATE + e] = s;
    }
}

// per-query pass: dq, and the partials of the gradient state dKV = Q^T dnum, dKsum = Q^T dden
template <typename T>
__global__ __launch_bounds__(256) void lb_query_pass(LbArgs a) {
    using M = Mma32<T>;
    __shared__ __attribute__((aligned(16))) char img[4 * 3 * L2_IMG];
    __shared__ float red[4 * 2 * 16 * 64];
    const int chunk = blockIdx.x, nh = blockIdx.y, n = nh / a.H, hh = nh - n * a.H, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    char* qimg = img + wave * 3 * L2_IMG;
    char* dimg = qimg + L2_IMG;
    char* wimg = dimg + L2_IMG;
    for (int e = lane; e < L2_IMG / 16; e += 64) reinterpret_cast<v4u*>(wimg)[e] = v4u{0u, 0u, 0u, 0u};      // only channel 0 is ever written again
    const float* st = a.state + (size_t)nh * LB_STATE;
    LFrag<T> kvt[2], kvr[2];
    float ksf[2][8], ksr[16];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        kvt[g] = lb_state_cols<T>(st, g, lr, h);                   // A of num^T: rows v, k = d
        kvr[g] = lb_state_rows_acc<T>(st, g, lr, h);               // A of dQ^T: rows d, k = v in the order of a packed accumulator
#pragma unroll
        for (int j = 0; j < 8; ++j) ksf[g][j] = st[LB_D * LB_D + 16 * g + 8 * h + j];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ksr[r] = st[LB_D * LB_D + gf_acc_row(r, h)];
    v16f acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    const float sl = (float)a.S;
    for (int j2 = 0; j2 < 2; ++j2) {
        const int l = chunk * L2_TOK + (wave + 4 * j2) * 32 + lr;
        const bool live = l < a.L && (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + l] != 0);
        const size_t row = (size_t)n * a.L + min(l, a.L - 1);
        const T* qp = (const T*)a.q + row * a.ldq + hh * LB_D;
        float qq[2][8], qa[16], ga[16];
        lb_row<T>(qp, h, qq);
        lb_row_acc<T>(qp, h, qa);
        lb_row_acc<T>((const T*)a.dout + row * a.ldo + hh * LB_D, h, ga);
        float den = 0.f;
        LFrag<T> qf[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                qq[g][j] = live ? lb_phi(qq[g][j]) : 0.f;
                den += qq[g][j] * ksf[g][j];
            }
            qf[g] = lb_pack<T>(qq[g]);
        }
        den += __shfl_xor(den, 32, 64);
        den += a.eps;
        v16f num;
#pragma unroll
        for (int r = 0; r < 16; ++r) num[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g) M::mma(kvt[g], qf[g], num);                      // rows v, column = the lane's token
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dot += ga[r] * num[r];
        dot += __shfl_xor(dot, 32, 64);
        const float z = sl / den, dden = live ? -dot * z / den : 0.f;
        float dnum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dnum[r] = live ? ga[r] * z : 0.f;
        v16f dqt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) M::mma(kvr[s2], lb_pack<T>(dnum + 8 * s2), dqt);   // rows d, column = the lane's token
        float dq[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = live ? (dqt[r] + dden * ksr[r]) * (qa[r] > 0.f ? 1.f : __expf(qa[r])) : 0.f;
        if (l < a.L) lb_store_acc<T>((T*)a.dq + ((size_t)n * a.L + l) * (a.H * LB_D) + hh * LB_D, h, dq);
        // the tile's Q, dnum and dden rows -> images, then their transposes as operands of the gradient state
#pragma unroll
        for (int g = 0; g < 2; ++g) *reinterpret_cast<LFrag<T>*>(qimg + lr * 64 + (16 * g + 8 * h) * 2) = qf[g];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
            *reinterpret_cast<gf_vec<T, 4>*>(dimg + lr * 64 + (8 * j4 + 4 * h) * 2) =
                gf_vec<T, 4>{(T)dnum[4 * j4], (T)dnum[4 * j4 + 1], (T)dnum[4 * j4 + 2], (T)dnum[4 * j4 + 3]};
        if (h == 0) *reinterpret_cast<T*>(wimg + lr * 64) = (T)dden;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const LFrag<T> qt = lb_tr<T>(qimg, s2, lane);
            M::mma(qt, lb_tr<T>(dimg, s2, lane), acc[0]);          // dKV: rows d, columns v
            M::mma(qt, lb_tr<T>(wimg, s2, lane), acc[1]);          // column 0: dKsum
        }
        __builtin_amdgcn_wave_barrier();
    }
    lb_reduce_store(acc, red, a.part + ((size_t)nh * a.chunksL + chunk) * LB_STATE);
}

// per-source pass: dk, dv from the gradient state (no LDS: both products are state x token rows)
template <typename T>
__global__ __launch_bounds__(256) void lb_source_pass(LbArgs a) {
    using M = Mma32<T>;
    const int chunk = blockIdx.x, nh = blockIdx.y, n = nh / a.H, hh = nh - n * a.H, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    const float* g = a.gstate + (size_t)nh * LB_STATE;
    LFrag<T> gr[2], gc[2];
    float dks[16];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        gr[k2] = lb_state_rows<T>(g, k2, lr, h);                   // A of dK^T: rows d, k = v
        gc[k2] = lb_state_cols<T>(g, k2, lr, h);                   // A of dvs^T: rows v, k = d
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dks[r] = g[LB_D * LB_D + gf_acc_row(r, h)];
    const float inv_s = 1.0f / (float)a.S;
    for (int j2 = 0; j2 < 2; ++j2) {
        const int s = chunk * L2_TOK + (wave + 4 * j2) * 32 + lr;
        const bool live = s < a.S && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + s] != 0);
        const size_t row = (size_t)n * a.S + min(s, a.S - 1);
        const T* kp = (const T*)a.k + row * a.ldk + hh * LB_D;
        float kk[2][8], vv[2][8], ka[16];
        lb_row<T>(kp, h, kk);
        lb_row<T>((const T*)a.v + row * a.ldv + hh * LB_D, h, vv);
        lb_row_acc<T>(kp, h, ka);
        v16f dkt, dvt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dkt[r] = dvt[r] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { kk[k2][j] = lb_phi(kk[k2][j]); vv[k2][j] *= inv_s; }
            M::mma(gr[k2], lb_pack<T>(vv[k2]), dkt);               // dK^T = dKV vs^T: rows d, column = the lane's token
            M::mma(gc[k2], lb_pack<T>(kk[k2]), dvt);               // dvs^T = dKV^T K^T: rows v
        }
        float ok[16], ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ok[r] = live ? (dkt[r] + dks[r]) * (ka[r] > 0.f ? 1.f : __expf(ka[r])) : 0.f;
            ov[r] = live ? dvt[r] * inv_s : 0.f;
        }
        if (s < a.S) {
            lb_store_acc<T>((T*)a.dk + ((size_t)n * a.S + s) * (a.H * LB_D) + hh * LB_D, h, ok);
            lb_store_acc<T>((T*)a.dv + ((size_t)n * a.S + s) * (a.H * LB_D) + hh * LB_D, h, ov);
        }
    }
}

template <typename T>
void lb_launch(const LbArgs& a, hipStream_t st) {
    const int NH = a.N * a.H;
    lb_state_partial<T><<<dim3(a.chunksS, NH), 256, 0, st>>>(a);
    lb_state_sum<<<NH, 256, 0, st>>>(a.part, a.state, a.chunksS);
    lb_query_pass<T><<<dim3(a.chunksL, NH), 256, 0, st>>>(a);
    lb_state_sum<<<NH, 256, 0, st>>>(a.part, a.gstate, a.chunksL);
    lb_source_pass<T><<<dim3(a.chunksS, NH), 256, 0, st>>>(a);
}

}   // namespace

extern "C" size_t gf_linear_attention_backward_workspace_bytes(int N, int L, int S, int H) {
    if (N <= 0 || L <= 0 || S <= 0 || H <= 0) return 0;
    const size_t cl = (L + LB_TOK - 1) / LB_TOK, cs = (S + LB_TOK - 1) / LB_TOK, c = cl > cs ? cl : cs;
    return gf_align_up(sizeof(float) * (size_t)N * H * (c + 2) * LB_STATE, 256);
}

// dq [N, L, H*32], dk, dv [N, S, H*32] (contiguous, `dtype`) of out = LinearAttention(q, k, v) given dout; q, k, v, dout are
// [N, L|S, H, 32] views with row strides ldq / ldk / ldv / ldo (elements); masks uint8 [N, L] / [N, S] or NULL
extern "C" int gf_linear_attention_backward(const void* q, const void* k, const void* v, const void* dout, int dtype, int N, int L, int S, int H,
                                            int D, long ldq, long ldk, long ldv, long ldo, const uint8_t* q_mask, const uint8_t* kv_mask, float eps,
                                            void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(q && k && v && dout && dq && dk && dv, "null pointer");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit activations");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0 && H > 0 && D == LB_D, "built for heads of 32 channels (the coarse level)");
    GF_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "rows must be 16-byte aligned");
    if (workspace == nullptr || workspace_bytes < gf_linear_attention_backward_workspace_bytes(N, L, S, H)) {
        gf_set_error("gf_linear_attention_backward: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    LbArgs a{};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.q_mask = q_mask; a.kv_mask = kv_mask;
    a.dq = dq; a.dk = dk; a.dv = dv; a.N = N; a.L = L; a.S = S; a.H = H; a.eps = eps;
    a.chunksL = (L + L2_TOK - 1) / L2_TOK; a.chunksS = (S + L2_TOK - 1) / L2_TOK;     // (the workspace is sized for chunks of LB_TOK: larger)
    const size_t c = a.chunksL > a.chunksS ? a.chunksL : a.chunksS;
    a.part = (float*)workspace;
    a.state = a.part + (size_t)N * H * c * LB_STATE;
    a.gstate = a.state + (size_t)N * H * LB_STATE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GF_F16) lb_launch<_Float16>(a, st);
    else lb_launch<gf_bf16>(a, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// =====================================================================================================================
// K2 (training, fine level): backward of LinearAttention.forward (linear_attention.py:21-51) on the fine-level windows -
// [Nw, Lw <= 32, 8 heads x 16 channels] tensors, no masks (full_model.py:97-98) - whose forward is la_window_mfma
// (k2_linear_attention.hip).  With A = phi(q), B = phi(k), vs = v / Lw, KV = B^T vs, ks = sum_s B_s, den = A ks + eps,
// num = A KV, out = Lw num / den:
//   dnum = dout Lw / den,  dden = -(dout . num) Lw / den^2,  dA = dnum KV^T + dden ks,  dKV = A^T dnum,  dks = sum_l dden_l A_l,
//   dB = vs dKV^T + dks,  dvs = B dKV,  dq = dA phi'(q),  dk = dB phi'(k),  dv = dvs / Lw,   phi'(x) = x > 0 ? 1 : exp(x).
// Round 6: ONE WAVE per window, every product on v_mfma_f32_32x32x16 (rounds 4-5: one workgroup of 128 threads per window, fp32 loops over
// LDS rows, 2.6 ms per 42 k-window call).  The 128 channels go as four blocks of two heads; a block's two 16 x 16 states are the diagonal
// quadrants of one 32 x 32 product (the others are zeroed).  lane = token: rows are loaded once, in ACCUMULATOR channel order, so a row is
// at once the B operand of the per-token products and lane-local for phi' / den / dout . num; the token-contracting products (KV, dKV and
// their transposes - both orientations are multiplied, which saves transposing an accumulator - and the row sums against a column of
// ones) read transposes of [token][64 B] LDS images with ds_read_b64_tr_b16; every state operand of the per-token products is an
// accumulator packed as it stands.  16-bit operands (phi(q), phi(k), vs, the states, dnum, dden phi(q)), fp32 accumulation.
// =====================================================================================================================
namespace {

constexpr int WB_C = 128, WB_L = 32;

template <typename T>
__global__ __launch_bounds__(256) void window_la_backward(const T* q, const T* k, const T* v, const T* dout, T* dq, T* dk, T* dv, int Nw, int Lw,
                                                          float eps) {
    using M = Mma32<T>;
    __shared__ __attribute__((aligned(16))) char img[4 * 5 * L2_IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    const int win = blockIdx.x * 4 + wave;
    if (win >= Nw) return;                                          // (whole waves: no workgroup barrier below)
    char* kimg = img + wave * 5 * L2_IMG;
    char* vimg = kimg + L2_IMG;
    char* qimg = vimg + L2_IMG;
    char* dimg = qimg + L2_IMG;
    char* wimg = dimg + L2_IMG;
    const bool live = lr < Lw;
    const size_t row = ((size_t)win * Lw + min(lr, Lw - 1)) * WB_C;
    const float lw = (float)Lw, inv_l = 1.0f / lw;
    LFrag<T> ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.f;
    auto put = [&](char* im, const float (&x)[16]) {               // a row in accumulator channel order -> its image row
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
            *reinterpret_cast<gf_vec<T, 4>*>(im + lr * 64 + (8 * j4 + 4 * h) * 2) =
                gf_vec<T, 4>{(T)x[4 * j4], (T)x[4 * j4 + 1], (T)x[4 * j4 + 2], (T)x[4 * j4 + 3]};
    };
    auto zero = [](v16f& x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = 0.f;
    };
    auto diag = [&](v16f& x) {                                      // keep the two heads' own quadrants: row head r >> 3, column head lr >> 4
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = (r >> 3) == (lr >> 4) ? x[r] : 0.f;
    };
    for (int pb = 0; pb < 4; ++pb) {
        const int c0 = 32 * pb;
        float qa[16], ka[16], va[16], ga[16], qp[16], kp[16], vs[16];
        lb_row_acc<T>(q + row + c0, h, qa);
        lb_row_acc<T>(k + row + c0, h, ka);
        lb_row_acc<T>(v + row + c0, h, va);
        lb_row_acc<T>(dout + row + c0, h, ga);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            qp[r] = live ? lb_phi(qa[r]) : 0.f;
            kp[r] = live ? lb_phi(ka[r]) : 0.f;
            vs[r] = live ? va[r] * inv_l : 0.f;
            ga[r] = live ? ga[r] : 0.f;
        }
        put(kimg, kp);
        put(vimg, vs);
        put(qimg, qp);
        __builtin_amdgcn_wave_barrier();
        v16f kv, vk, ks;
        zero(kv); zero(vk); zero(ks);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const LFrag<T> kt = lb_tr<T>(kimg, s2, lane), vt = lb_tr<T>(vimg, s2, lane);
            M::mma(kt, vt, kv);                                     // rows d, columns v
            M::mma(vt, kt, vk);                                     // rows v, columns d
            M::mma(kt, ones, ks);                                   // every column: ks[d]
        }
        diag(kv); diag(vk);
        // per token and head: den, num, dout . num
        float den[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) den[r >> 3] += qp[r] * ks[r];
        v16f num;
        zero(num);
        M::mma(lb_pack_acc<T, 0>(kv), lb_pack<T>(qp), num);          // rows v, column = the lane's token
        M::mma(lb_pack_acc<T, 1>(kv), lb_pack<T>(qp + 8), num);
        float dot[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) dot[r >> 3] += ga[r] * num[r];
        float z[2], dden[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            den[e] += __shfl_xor(den[e], 32, 64);
            dot[e] += __shfl_xor(dot[e], 32, 64);
            den[e] += eps;
            z[e] = lw / den[e];
            dden[e] = -dot[e] * z[e] / den[e];
        }
        float dnum[16], qw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dnum[r] = ga[r] * z[r >> 3];
            qw[r] = qp[r] * dden[r >> 3];
        }
        v16f dqt;
        zero(dqt);
        M::mma(lb_pack_acc<T, 0>(vk), lb_pack<T>(dnum), dqt);        // rows d, column = the lane's token
        M::mma(lb_pack_acc<T, 1>(vk), lb_pack<T>(dnum + 8), dqt);
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (dqt[r] + dden[r >> 3] * ks[r]) * (qa[r] > 0.f ? 1.f : __expf(qa[r]));
        if (live) lb_store_acc<T>(dq + row + c0, h, o);
        // the gradient state: dKV = Q^T dnum (both orientations), dks = (dden Q)^T 1
        put(dimg, dnum);
        put(wimg, qw);
        __builtin_amdgcn_wave_barrier();
        v16f gkv, gvk, gks;
        zero(gkv); zero(gvk); zero(gks);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const LFrag<T> qt = lb_tr<T>(qimg, s2, lane), dt = lb_tr<T>(dimg, s2, lane);
            M::mma(qt, dt, gkv);                                    // rows d, columns v
            M::mma(dt, qt, gvk);                                    // rows v, columns d
            M::mma(lb_tr<T>(wimg, s2, lane), ones, gks);            // every column: dks[d]
        }
        diag(gkv); diag(gvk);
        v16f dkt, dvt;
        zero(dkt); zero(dvt);
        M::mma(lb_pack_acc<T, 0>(gvk), lb_pack<T>(vs), dkt);         // dB^T = dKV vs^T: rows d
        M::mma(lb_pack_acc<T, 1>(gvk), lb_pack<T>(vs + 8), dkt);
        M::mma(lb_pack_acc<T, 0>(gkv), lb_pack<T>(kp), dvt);         // dvs^T = dKV^T B^T: rows v
        M::mma(lb_pack_acc<T, 1>(gkv), lb_pack<T>(kp + 8), dvt);
        float ok[16], ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ok[r] = (dkt[r] + gks[r]) * (ka[r] > 0.f ? 1.f : __expf(ka[r]));
            ov[r] = dvt[r] * inv_l;
        }
        if (live) {
            lb_store_acc<T>(dk + row + c0, h, ok);
            lb_store_acc<T>(dv + row + c0, h, ov);
        }
        __builtin_amdgcn_wave_barrier();                           // the images are rewritten by the next block
    }
}

}   // namespace

// dq, dk, dv [Nw, Lw, 128] (contiguous, `dtype`) of the fine-level window attention given dout; q, k, v, dout contiguous [Nw, Lw, 128]
extern "C" int gf_window_linear_attention_backward(const void* q, const void* k, const void* v, const void* dout, int dtype, int Nw, int Lw,
                                                   float eps, void* dq, void* dk, void* dv, void* stream) {
    GF_CHECK_ARG(q && k && v && dout && dq && dk && dv, "null pointer");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit activations");
    GF_CHECK_ARG(Nw > 0 && Lw > 0 && Lw <= WB_L, "windows of 1 .. 32 tokens");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((Nw + 3) / 4);
    if (dtype == GF_F16)
        window_la_backward<_Float16><<<grid, 256, 0, st>>>((const _Float16*)q, (const _Float16*)k, (const _Float16*)v, (const _Float16*)dout,
                                                        (_Float16*)dq, (_Float16*)dk, (_Float16*)dv, Nw, Lw, eps);
    else
        window_la_backward<gf_bf16><<<grid, 256, 0, st>>>((const gf_bf16*)q, (const gf_bf16*)k, (const gf_bf16*)v, (const gf_bf16*)dout, (gf_bf16*)dq,
                                                       (gf_bf16*)dk, (gf_bf16*)dv, Nw, Lw, eps);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// =====================================================================================================================
// K8 (training): backward of FineMatching2.forward's confidence (model/fine_matching2.py:52-63):
//   sim = f0 f1^T / (C temperature), A = softmax(sim, dim 1), B = softmax(sim, dim 2), conf = A o B        (per match, 25 x 25)
// given G = dL/dconf:  T = G o conf,  dsim = 2 T - A o (1 colsum(T)^T) - B o (rowsum(T) 1^T),
//   df0 = dsim f1 / (C temperature),  df1 = dsim^T f0 / (C temperature).
// One workgroup per match, everything in LDS, fp32 arithmetic (the forward keeps the confidence in fp32 in every precision mode).
// =====================================================================================================================
namespace {

constexpr int FB_W = 25;

template <typename T>
__global__ __launch_bounds__(256) void fine_match_backward(const T* f0, const T* f1, const float* dconf, T* df0, T* df1, int C, float temperature) {
    __shared__ float s0[FB_W][129], s1[FB_W][129];
    __shared__ float sim[FB_W][FB_W + 1], ds[FB_W][FB_W + 1];
    __shared__ float rmax[FB_W], rsum[FB_W], cmax[FB_W], csum[FB_W], rt[FB_W], ct[FB_W];
    const int m = blockIdx.x, t = threadIdx.x;
    const float rs = sqrtf((float)C);
    const T* p0 = f0 + (size_t)m * FB_W * C;
    const T* p1 = f1 + (size_t)m * FB_W * C;
    for (int i = t; i < FB_W * C; i += 256) {
        s0[i / C][i % C] = gf_to_float(p0[i]) / rs;          // the forward's arithmetic (fine_match: feat / C**.5 on both sides)
        s1[i / C][i % C] = gf_to_float(p1[i]) / rs;
    }
    __syncthreads();
    for (int o = t; o < FB_W * FB_W; o += 256) {
        const int i = o / FB_W, j = o % FB_W;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += s0[i][c] * s1[j][c];
        sim[i][j] = acc / temperature;
    }
    __syncthreads();
    if (t < FB_W) {                                           // softmax over dim 2 (row statistics)
        float mx = -INFINITY;
        for (int j = 0; j < FB_W; ++j) mx = fmaxf(mx, sim[t][j]);
        float s = 0.f;
        for (int j = 0; j < FB_W; ++j) s += expf(sim[t][j] - mx);
        rmax[t] = mx; rsum[t] = s;
    } else if (t >= 64 && t < 64 + FB_W) {                    // softmax over dim 1 (column statistics)
        const int j = t - 64;
        float mx = -INFINITY;
        for (int i = 0; i < FB_W; ++i) mx = fmaxf(mx, sim[i][j]);
        float s = 0.f;
        for (int i = 0; i < FB_W; ++i) s += expf(sim[i][j] - mx);
        cmax[j] = mx; csum[j] = s;
    }
    __syncthreads();
    // T = G o conf into ds (for now), then its row / column sums
    for (int o = t; o < FB_W * FB_W; o += 256) {
        const int i = o / FB_W, j = o % FB_W;
        const float a = expf(sim[i][j] - cmax[j]) / csum[j], b = expf(sim[i][j] - rmax[i]) / rsum[i];
        ds[i][j] = dconf[(size_t)m * FB_W * FB_W + o] * a * b;
    }
    __syncthreads();
    if (t < FB_W) {
        float s = 0.f;
        for (int j = 0; j < FB_W; ++j) s += ds[t][j];
        rt[t] = s;
    } else if (t >= 64 && t < 64 + FB_W) {
        const int j = t - 64;
        float s = 0.f;
        for (int i = 0; i < FB_W; ++i) s += ds[i][j];
        ct[j] = s;
    }
    __syncthreads();
    for (int o = t; o < FB_W * FB_W; o += 256) {
        const int i = o / FB_W, j = o % FB_W;
        const float a = expf(sim[i][j] - cmax[j]) / csum[j], b = expf(sim[i][j] - rmax[i]) / rsum[i];
        ds[i][j] = (2.0f * ds[i][j] - a * ct[j] - b * rt[i]) / temperature;     // d / d(the pre-temperature product)
    }
    __syncthreads();
    // df0[i][c] = sum_j ds[i][j] s1[j][c] / sqrt(C);  df1[j][c] = sum_i ds[i][j] s0[i][c] / sqrt(C)
    for (int o = t; o < FB_W * C; o += 256) {
        const int r = o / C, c = o % C;
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < FB_W; ++k) {
            a0 += ds[r][k] * s1[k][c];
            a1 += ds[k][r] * s0[k][c];
        }
        df0[(size_t)m * FB_W * C + o] = gf_from_float<T>(a0 / rs);
        df1[(size_t)m * FB_W * C + o] = gf_from_float<T>(a1 / rs);
    }
}

}   // namespace

// df0, df1 [M, 25, C] (`dtype`) of fine_matrix = FineMatching2's 25 x 25 dual-softmax confidence given dconf fp32 [M, 25, 25]
extern "C" int gf_fine_match_backward(const void* f0, const void* f1, int dtype, int M, int WWin, int C, float temperature, const float* dconf,
                                      void* df0, void* df1, void* stream) {
    GF_CHECK_ARG(f0 && f1 && dconf && df0 && df1, "null pointer");
    GF_CHECK_ARG(M > 0 && WWin == FB_W && C > 0 && C <= 128, "built for 5x5 windows and C <= 128");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16 && temperature > 0.f, "bad dtype / temperature");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GF_F32) fine_match_backward<float><<<M, 256, 0, st>>>((const float*)f0, (const float*)f1, dconf, (float*)df0, (float*)df1, C, temperature);
    else if (dtype == GF_F16) fine_match_backward<_Float16><<<M, 256, 0, st>>>((const _Float16*)f0, (const _Float16*)f1, dconf, (_Float16*)df0, (_Float16*)df1, C, temperature);
    else fine_match_backward<gf_bf16><<<M, 256, 0, st>>>((const gf_bf16*)f0, (const gf_bf16*)f1, dconf, (gf_bf16*)df0, (gf_bf16*)df1, C, temperature);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
