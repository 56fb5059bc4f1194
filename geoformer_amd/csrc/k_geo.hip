// GeoModule geometry + windowed cross-attention (CDNA4 / gfx950).
//
//   gf_window_geometry        a8 : get_map_keypoints (utils/common_utils.py:137-144) -> warp_points_batch
//                                  (utils/homography.py:86-105) -> generate_window (utils/common_utils.py:65-91)
//                                  -> the coarse cell each window position samples (sample_descriptors,
//                                  utils/common_utils.py:171-181), fused: one int32 cell index (or -1) per
//                                  (query cell, window position); nothing else is materialised.
//   gf_inlier_index           a7 : inlier occupancy maps (model/geo_module.py:82-94) + the ascending
//                                  token index list that `feat[mask]` (geo_transformer/transformer.py:118,121)
//                                  would gather.
//   gf_window_cross_attention a11/a12 (K5): FullAttention (model/geo_transformer/geo_attention.py:72-101)
//                                  over the 25 gathered keys of each query; keys/values are gathered from
//                                  ALREADY PROJECTED maps (project-then-gather: same math as the reference's
//                                  gather-then-project because k_proj/v_proj are bias-free linear maps).
#include <math.h>

#include <type_traits>

#include "gf_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
struct WgArgs {
    const float* Hm;        // [N][9] homography applied to the grid of the QUERY image
    const int32_t* valid;   // [N] or null
    int N, hq, wq;          // query grid (coarse cells)
    int Himg, Wimg;         // image that receives the windows (pixels)
    int wk;                 // its coarse grid width
    int scale, wsz;
    const float* wscale;    // [N][2] per-sample window scale (x,y) or null -> scale
    int32_t* win;           // [N][hq*wq][wsz*wsz]
    int32_t* kps;           // optional [N][L][ww][2] integer window coordinates (tests)
    float* warped;          // optional [N][L][2]
};

__global__ void window_geometry(WgArgs a) {
    const int n = blockIdx.y, L = a.hq * a.wq, ww = a.wsz * a.wsz;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    int32_t* win = a.win + ((size_t)n * L + l) * ww;
    if (a.valid && !a.valid[n]) {
        for (int k = 0; k < ww; ++k) win[k] = -1;
        return;
    }
    const float* h = a.Hm + 9 * n;
    const float x = (float)((l % a.wq) * a.scale), y = (float)((l / a.wq) * a.scale);
    // bmm of the fp32 homography with (x, y, 1) (homography.py:97), then the w==0 guard (:101-103)
    float X = h[0] * x + h[1] * y + h[2];
    float Y = h[3] * x + h[4] * y + h[5];
    float Wc = h[6] * x + h[7] * y + h[8];
    if (Wc == 0.f) Wc = 1e-6f;
    const float px = X / Wc, py = Y / Wc;
    if (a.warped) {
        a.warped[((size_t)n * L + l) * 2] = px;
        a.warped[((size_t)n * L + l) * 2 + 1] = py;
    }
    const float sx = a.wscale ? (float)a.scale * a.wscale[2 * n] : (float)a.scale;
    const float sy = a.wscale ? (float)a.scale * a.wscale[2 * n + 1] : (float)a.scale;
    const int half = a.wsz / 2;
    for (int k = 0; k < ww; ++k) {
        const float qx = px + (float)(k % a.wsz - half) * sx;      // x fastest (common_utils.py:71-78)
        const float qy = py + (float)(k / a.wsz - half) * sy;
        const bool oob = (qx < 0.f) | (qy < 0.f) | (qx >= (float)a.Wimg) | (qy >= (float)a.Himg);   // on floats (:84)
        const long ix = oob ? 0 : (long)qx, iy = oob ? 0 : (long)qy;                                  // .long() (:89)
        if (a.kps) {
            a.kps[(((size_t)n * L + l) * ww + k) * 2] = (int32_t)ix;
            a.kps[(((size_t)n * L + l) * ww + k) * 2 + 1] = (int32_t)iy;
        }
        // cell = float(kps) // s  (common_utils.py:171-172)
        const int cx = (int)floorf((float)ix / (float)a.scale), cy = (int)floorf((float)iy / (float)a.scale);
        win[k] = oob ? -1 : cy * a.wk + cx;
    }
}

// ------------------------------------------------------------------------------------------------
struct ImArgs {
    const float* kp0;        // [cap][2] integer-valued keypoints (from gf_ransac_homography)
    const float* kp1;
    const uint8_t* keep;     // [cap]
    const int32_t* counts;   // [1+N]
    int N, L, S, w0, w1, scale;
    uint8_t* map0;           // [N][L]
    uint8_t* map1;           // [N][S]
    int32_t* idx0;           // [N][L]  ascending indices of set cells
    int32_t* idx1;           // [N][S]
    int32_t* nidx;           // [N][2]  K0, K1
};

__global__ void inlier_scatter(ImArgs a) {
    const int n = blockIdx.y;
    int off = 0;
    for (int b = 0; b < n; ++b) off += a.counts[1 + b];
    const int cnt = a.counts[1 + n];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        if (!a.keep[off + i]) continue;
        const int x0 = (int)a.kp0[2 * (off + i)], y0 = (int)a.kp0[2 * (off + i) + 1];
        const int x1 = (int)a.kp1[2 * (off + i)], y1 = (int)a.kp1[2 * (off + i) + 1];
        const int c0 = (y0 / a.scale) * a.w0 + x0 / a.scale, c1 = (y1 / a.scale) * a.w1 + x1 / a.scale;
        if (c0 >= 0 && c0 < a.L) a.map0[(size_t)n * a.L + c0] = 1;
        if (c1 >= 0 && c1 < a.S) a.map1[(size_t)n * a.S + c1] = 1;
    }
}

// one workgroup per (sample, side): ordered compaction of the set cells
__global__ __launch_bounds__(1024) void inlier_compact(ImArgs a) {
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int n = blockIdx.x, side = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = side ? a.S : a.L;
    const uint8_t* map = side ? a.map1 + (size_t)n * a.S : a.map0 + (size_t)n * a.L;
    int32_t* idx = side ? a.idx1 + (size_t)n * a.S : a.idx0 + (size_t)n * a.L;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int start = 0; start < len; start += 1024) {
        const int i = start + tid;
        const bool f = i < len && map[i] != 0;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wave_tot[w];
            woff += (w < wave) ? c : 0;
            tot += c;
        }
        const int base = running;
        if (f) idx[base + woff + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) running = base + tot;
        __syncthreads();
    }
    if (tid == 0) a.nidx[2 * n + side] = running;
}

// ------------------------------------------------------------------------------------------------
// K5 windowed cross-attention.  One wave per query; lane owns 4 consecutive channels (C = 256 ->
// head = lane / 16, D = 64).  Per key: 4 FMAs + a 16-lane DPP butterfly for the head's logit.
// ------------------------------------------------------------------------------------------------
struct CaArgs {
    const void* q;          // [N][L][C]
    const void* kmap;       // [N][S][ldk] projected keys of the OTHER image
    const void* vmap;
    long ldq, ldk, ldv;
    const int32_t* win;     // [N][L][WW]
    const int32_t* valid;   // [N] or null
    void* out;              // [N][L][C]
    int N, L, S, WW;
    float softmax_temp;     // 1/sqrt(D)
};

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row16_sum(float x) {
    x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror
    x = dpp_add<0x140>(x);   // row_mirror
    return x;
}

template <typename T>
__device__ __forceinline__ v4f load4(const T* p);
template <>
__device__ __forceinline__ v4f load4<float>(const float* p) { return *reinterpret_cast<const v4f*>(p); }
template <>
__device__ __forceinline__ v4f load4<_Float16>(const _Float16* p) {
    const v4h h = *reinterpret_cast<const v4h*>(p);
    return v4f{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
}
template <typename T>
__device__ __forceinline__ void store4(T* p, v4f v);
template <>
__device__ __forceinline__ void store4<float>(float* p, v4f v) { *reinterpret_cast<v4f*>(p) = v; }
template <>
__device__ __forceinline__ void store4<_Float16>(_Float16* p, v4f v) {
    *reinterpret_cast<v4h*>(p) = v4h{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}
template <>
__device__ __forceinline__ v4f load4<gf_bf16>(const gf_bf16* p) {
    const v4b h = *reinterpret_cast<const v4b*>(p);
    return v4f{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
}
template <>
__device__ __forceinline__ void store4<gf_bf16>(gf_bf16* p, v4f v) {
    *reinterpret_cast<v4b*>(p) = v4b{(gf_bf16)v.x, (gf_bf16)v.y, (gf_bf16)v.z, (gf_bf16)v.w};
}

template <typename T> struct Raw4;
template <> struct Raw4<float> { using type = v4f; };
template <> struct Raw4<_Float16> { using type = v4h; };
template <> struct Raw4<gf_bf16> { using type = v4b; };
template <typename R>
__device__ __forceinline__ v4f widen4(R r) { return v4f{(float)r.x, (float)r.y, (float)r.z, (float)r.w}; }

template <typename T, int WW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void window_cross_attention(CaArgs a) {
    using Raw = typename Raw4<T>::type;
    constexpr bool FAST = !std::is_same<T, float>::value;     // 16-bit storage: hardware exp / reciprocal, value rows requested up front
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    // the wave index as a scalar: the query cell, its window table row and the 25 row offsets are wave-uniform (scalar loads,
    // scalar-base row requests)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup ids go round-robin over the 8 XCDs: each XCD (= each L2) gets a contiguous band of query cells, so that it
    // fetches that band's windows of the key / value maps and not the whole maps
    int bx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int l = bx * 4 + wave;
    if (l >= a.L) return;
    T* out = (T*)a.out + ((size_t)n * a.L + l) * 256 + lane * 4;
    if (a.valid && !a.valid[n]) {          // layer skipped for this sample; caller keeps x
        store4<T>(out, v4f{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const int32_t* win = a.win + ((size_t)n * a.L + l) * WW;
    int cell[WW];
#pragma unroll
    for (int k = 0; k < WW; ++k) cell[k] = win[k];
    const v4f q = load4<T>((const T*)a.q + ((size_t)n * a.L + l) * a.ldq + lane * 4);
    const T* kb = (const T*)a.kmap + (size_t)n * a.S * a.ldk + lane * 4;
    const T* vb = (const T*)a.vmap + (size_t)n * a.S * a.ldv + lane * 4;
    // branch-free gathers: a masked key reads row 0 and is overridden afterwards.  All 25 key rows (and, in the 16-bit modes,
    // all 25 value rows: they do not depend on the softmax) are requested back to back before the first reduction waits.
    Raw kraw[WW], vraw[FAST ? WW : 1];
#pragma unroll
    for (int k = 0; k < WW; ++k)
        kraw[k] = *reinterpret_cast<const Raw*>(kb + (unsigned)max(cell[k], 0) * (unsigned)a.ldk);    // 32-bit offsets: S * ld < 2^32
    if constexpr (FAST) {
#pragma unroll
        for (int k = 0; k < WW; ++k)
            vraw[k] = *reinterpret_cast<const Raw*>(vb + (unsigned)max(cell[k], 0) * (unsigned)a.ldv);
    }
    __builtin_amdgcn_sched_barrier(0);
    float dot[WW];
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        const v4f kv = widen4(kraw[k]);
        dot[k] = q.x * kv.x + q.y * kv.y + q.z * kv.z + q.w * kv.w;
    }
    // the 16-lane DPP reductions of the 25 keys interleave (two wait states in front of every DPP read otherwise)
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0xB1>(dot[k]);     // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0x4E>(dot[k]);     // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0x141>(dot[k]);    // row_half_mirror
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0x140>(dot[k]);    // row_mirror
    float logit[WW];
    float mx = -INFINITY;
    bool any = false;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        any = any || cell[k] >= 0;
        logit[k] = (cell[k] >= 0 ? dot[k] : -1e8f) * a.softmax_temp;   // masked_fill BEFORE the temperature (geo_attention.py:83,92)
        mx = fmaxf(mx, logit[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        logit[k] = FAST ? __expf(logit[k] - mx) : expf(logit[k] - mx);
        den += logit[k];
    }
    const float rden = __builtin_amdgcn_rcpf(den);
    v4f acc{0.f, 0.f, 0.f, 0.f};
    if (any) {
#pragma unroll
        for (int k = 0; k < WW; ++k) {
            // masked keys have weight exp(-1.25e7 - mx) == 0 whenever any key is valid: their (row 0) values add 0
            const float p = cell[k] >= 0 ? (FAST ? logit[k] * rden : logit[k] / den) : 0.f;
            v4f vv;
            if constexpr (FAST) vv = widen4(vraw[k]);
            else vv = load4<T>(vb + (unsigned)max(cell[k], 0) * (unsigned)a.ldv);
            acc.x += p * vv.x; acc.y += p * vv.y; acc.z += p * vv.z; acc.w += p * vv.w;
        }
    }                                       // no valid key: the row is zeroed (geo_attention.py:98-100)
    store4<T>(out, acc);
}

// ------------------------------------------------------------------------------------------------
// K5 backward (training): gradients of window_cross_attention's output with respect to q and to the two projected
// maps.  One wave per query, lane = 4 channels (its head = lane / 16), as the forward: the 25 logits and the 25 products
// dP_k = dout . v_k are recomputed from the rows (no attention matrix is kept by the forward), then with p = softmax(logit):
//     dlogit_k = p_k (dP_k - sum_j p_j dP_j) / sqrt(D),   dq = sum_k dlogit_k k_k,   dk[cell_k] += dlogit_k q,   dv[cell_k] += p_k dout.
// The windows of neighbouring queries overlap, so dk / dv are scatter-ADDS: fp32 atomic adds (256 contiguous bytes per wave
// instruction) into zero-initialised fp32 maps [N][S][256] - the sums of ~25 contributions per cell stay in fp32 and are rounded
// to the storage type once by the caller.  Masked window positions (cell < 0) have p = 0: no gradient; a query without any valid
// key has a zeroed output (geo_attention.py:98-100): no gradient at all.
// ------------------------------------------------------------------------------------------------
struct CbArgs {
    const void* q;
    const void* kmap;
    const void* vmap;
    const void* dout;       // [N][L][256]
    long ldq, ldk, ldv;
    const int32_t* win;     // [N][L][WW]
    void* dq;               // [N][L][256] of T
    float* dk;              // [N][S][256] fp32, zeroed by the caller      (scatter form)
    float* dv;
    int N, L, S;
    float softmax_temp;
    // gather form (round 6): the per-query pass leaves (dlogit, p) per (query, window position, head) here instead of scattering, and
    // window_cross_gather adds every cell's contributions in the order of the caller's inverse index: no atomics, bit-reproducible
    float2* dlp;            // [N][L][WW][4]
    const int32_t* entries; // [N * L * WW]: l * WW + k of the contributions, sorted by cell (stable); cells < 0 at the end
    const int32_t* offsets; // [N * S + 1]: the contributions of global cell n * S + s are entries[offsets[.] .. offsets[. + 1])
    void* dk16;             // [N][S][256] of T
    void* dv16;
};

template <typename T, int WW, bool GATHER>
__global__ __launch_bounds__(256) void window_cross_attention_backward(CbArgs a) {
    using Raw = typename Raw4<T>::type;
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = blockIdx.x * 4 + wave;
    if (l >= a.L) return;
    const int32_t* win = a.win + ((size_t)n * a.L + l) * WW;
    int cell[WW];
    bool any = false;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        cell[k] = win[k];
        any = any || cell[k] >= 0;
    }
    T* dq = (T*)a.dq + ((size_t)n * a.L + l) * 256 + lane * 4;
    if (!any) {
        store4<T>(dq, v4f{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const v4f q = load4<T>((const T*)a.q + ((size_t)n * a.L + l) * a.ldq + lane * 4);
    const v4f go = load4<T>((const T*)a.dout + ((size_t)n * a.L + l) * 256 + lane * 4);
    const T* kb = (const T*)a.kmap + (size_t)n * a.S * a.ldk + lane * 4;
    const T* vb = (const T*)a.vmap + (size_t)n * a.S * a.ldv + lane * 4;
    Raw kraw[WW];
    float dot[WW], dp[WW];
#pragma unroll
    for (int k = 0; k < WW; ++k) kraw[k] = *reinterpret_cast<const Raw*>(kb + (unsigned)max(cell[k], 0) * (unsigned)a.ldk);
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        const v4f vv = widen4(*reinterpret_cast<const Raw*>(vb + (unsigned)max(cell[k], 0) * (unsigned)a.ldv));
        dp[k] = go.x * vv.x + go.y * vv.y + go.z * vv.z + go.w * vv.w;
    }
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        const v4f kv = widen4(kraw[k]);
        dot[k] = q.x * kv.x + q.y * kv.y + q.z * kv.z + q.w * kv.w;
    }
#pragma unroll
    for (int k = 0; k < WW; ++k) { dot[k] = dpp_add<0xB1>(dot[k]); dp[k] = dpp_add<0xB1>(dp[k]); }
#pragma unroll
    for (int k = 0; k < WW; ++k) { dot[k] = dpp_add<0x4E>(dot[k]); dp[k] = dpp_add<0x4E>(dp[k]); }
#pragma unroll
    for (int k = 0; k < WW; ++k) { dot[k] = dpp_add<0x141>(dot[k]); dp[k] = dpp_add<0x141>(dp[k]); }
#pragma unroll
    for (int k = 0; k < WW; ++k) { dot[k] = dpp_add<0x140>(dot[k]); dp[k] = dpp_add<0x140>(dp[k]); }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        dot[k] = (cell[k] >= 0 ? dot[k] : -1e8f) * a.softmax_temp;
        mx = fmaxf(mx, dot[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        dot[k] = cell[k] >= 0 ? expf(dot[k] - mx) : 0.f;
        den += dot[k];
    }
    float pd = 0.f;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        dot[k] = dot[k] / den;                                   // p_k (this head's)
        pd += dot[k] * dp[k];
    }
    v4f acc{0.f, 0.f, 0.f, 0.f};
    // a lane holds 4 adjacent channels; an atomic instruction is at full rate when its 64 lanes cover 256 contiguous bytes, so the
    // 256 values of a row go through a wave-private LDS row and leave as 4 instructions of 64 adjacent floats each (4 adds per
    // lane at a 16-byte stride measured 770 us per 6400-query call: a quarter of every 1 KiB span useful)
    __shared__ __attribute__((aligned(16))) float tr[4][2][256];
    float* dkb = a.dk + (size_t)n * a.S * 256 + lane;
    float* dvb = a.dv + (size_t)n * a.S * 256 + lane;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        if (cell[k] < 0) continue;                               // (wave-uniform)
        const float p = dot[k], dl = p * (dp[k] - pd) * a.softmax_temp;
        const v4f kv = widen4(kraw[k]);
        acc.x += dl * kv.x; acc.y += dl * kv.y; acc.z += dl * kv.z; acc.w += dl * kv.w;
        if constexpr (GATHER) {                                  // the head's (dlogit, p) from its first lane; the cell's sum is window_cross_gather's
            if ((lane & 15) == 0) a.dlp[(((size_t)n * a.L + l) * WW + k) * 4 + (lane >> 4)] = make_float2(dl, p);
            continue;
        }
        *reinterpret_cast<v4f*>(&tr[wave][0][lane * 4]) = v4f{dl * q.x, dl * q.y, dl * q.z, dl * q.w};
        *reinterpret_cast<v4f*>(&tr[wave][1][lane * 4]) = v4f{p * go.x, p * go.y, p * go.z, p * go.w};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* dkp = dkb + (size_t)cell[k] * 256;
        float* dvp = dvb + (size_t)cell[k] * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            atomicAdd(dkp + 64 * r, tr[wave][0][64 * r + lane]);
            atomicAdd(dvp + 64 * r, tr[wave][1][64 * r + lane]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    store4<T>(dq, acc);
}

// dk[cell] = sum over the (query, window position) pairs that look at the cell of dlogit q_query, dv[cell] = sum of p dout_query: one wave
// per cell (lane = 4 channels, head = lane / 16) walks the cell's slice of the inverse index - ~25 entries for a smooth warp - in its order
template <typename T, int WW>
__global__ __launch_bounds__(256) void window_cross_gather(CbArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long gc = (long)blockIdx.x * 4 + wave;
    if (gc >= (long)a.N * a.S) return;
    const int n = (int)(gc / a.S), head = lane >> 4;
    const int beg = a.offsets[gc], end = a.offsets[gc + 1];
    const T* qb = (const T*)a.q + (size_t)n * a.L * a.ldq + lane * 4;
    const T* gb = (const T*)a.dout + (size_t)n * a.L * 256 + lane * 4;
    const float2* wb = a.dlp + (size_t)n * a.L * WW * 4 + head;
    v4f dk{0.f, 0.f, 0.f, 0.f}, dv{0.f, 0.f, 0.f, 0.f};
    int e = beg;
    for (; e + 2 <= end; e += 2) {                               // two contributions in flight
        const int i0 = a.entries[e], i1 = a.entries[e + 1], l0 = i0 / WW, l1 = i1 / WW;
        const float2 w0 = wb[(size_t)i0 * 4], w1 = wb[(size_t)i1 * 4];
        const v4f q0 = load4<T>(qb + (size_t)l0 * a.ldq), g0 = load4<T>(gb + (size_t)l0 * 256);
        const v4f q1 = load4<T>(qb + (size_t)l1 * a.ldq), g1 = load4<T>(gb + (size_t)l1 * 256);
        dk.x += w0.x * q0.x; dk.y += w0.x * q0.y; dk.z += w0.x * q0.z; dk.w += w0.x * q0.w;
        dv.x += w0.y * g0.x; dv.y += w0.y * g0.y; dv.z += w0.y * g0.z; dv.w += w0.y * g0.w;
        dk.x += w1.x * q1.x; dk.y += w1.x * q1.y; dk.z += w1.x * q1.z; dk.w += w1.x * q1.w;
        dv.x += w1.y * g1.x; dv.y += w1.y * g1.y; dv.z += w1.y * g1.z; dv.w += w1.y * g1.w;
    }
    if (e < end) {
        const int i0 = a.entries[e], l0 = i0 / WW;
        const float2 w0 = wb[(size_t)i0 * 4];
        const v4f q0 = load4<T>(qb + (size_t)l0 * a.ldq), g0 = load4<T>(gb + (size_t)l0 * 256);
        dk.x += w0.x * q0.x; dk.y += w0.x * q0.y; dk.z += w0.x * q0.z; dk.w += w0.x * q0.w;
        dv.x += w0.y * g0.x; dv.y += w0.y * g0.y; dv.z += w0.y * g0.z; dv.w += w0.y * g0.w;
    }
    store4<T>((T*)a.dk16 + (size_t)gc * 256 + lane * 4, dk);
    store4<T>((T*)a.dv16 + (size_t)gc * 256 + lane * 4, dv);
}

// ------------------------------------------------------------------------------------------------
// K5, tiled form (16-bit storage, map widths known).  One workgroup = a tile of 8 x 4 query cells and ONE head.  The windows
// of neighbouring queries overlap: the union of the tile's 32 x 25 window cells is a small rectangle of the key map (12 x 8
// cells for a translation), so the head's 128-byte slices of the key and value rows of that rectangle are brought into LDS
// once (LDS-DMA, 16 bytes per lane, no registers) and the 800 row reads of the tile are LDS reads.  A tile whose rectangle
// exceeds CT_CAP cells (zoom between the images) reads its rows from global memory with the same arithmetic.
// Lane = 8 channels of the head (8 lanes per query, 8 queries per wave): v_dot2 for q.k, a 3-step DPP sum over the 8
// lanes, exp2 softmax, mixed-precision FMAs for p.v.
// ------------------------------------------------------------------------------------------------
constexpr int CT_QX = 8, CT_QY = 4, CT_Q = CT_QX * CT_QY, CT_WW = 25, CT_TS = 28;   // table stride: 28 ints = 7 x 16 bytes
// 144 cells x 128 B x 2 maps + the table = 40.5 KB of LDS and <= 128 registers: four workgroups per CU (measured: 192 cells /
// three workgroups 60 us, 144 / four 51 us per 8-image call).  A translation needs 12 x 8 = 96 cells, a zoom of 1.35 fills 144.
constexpr int CT_CAP = 144, CT_WAVES = 4;
struct CtArgs {
    const void* q;          // [N][L][ldq]
    const void* kmap;       // [N][S][ldk] projected keys of the OTHER image
    const void* vmap;
    long ldq, ldk, ldv;
    const int32_t* win;     // [N][L][25]
    const int32_t* valid;   // [N] or null
    void* out;              // [N][L][256]
    int N, L, S, hq, wq, wk, tiles_x;
    float scale2;           // log2(e) / sqrt(D)
};
struct CtRsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ CtRsrc ct_rsrc(const void* p, unsigned bytes) {
    return CtRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ void ct_lds_dma(const CtRsrc& rs, char* dst, int voffset) {        // 64 lanes x 16 B -> 1 KiB at dst
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.r, (__attribute__((address_space(3))) void*)dst, 16, voffset, 0, 0, 0);
}
typedef _Float16 ct_v2h __attribute__((ext_vector_type(2)));
typedef gf_bf16 ct_v2b __attribute__((ext_vector_type(2)));
typedef unsigned ct_u4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float ct_dot2(unsigned a, unsigned b, float c);       // c + a.lo * b.lo + a.hi * b.hi in fp32
template <>
__device__ __forceinline__ float ct_dot2<_Float16>(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(ct_v2h, a), __builtin_bit_cast(ct_v2h, b), c, false);
}
template <>
__device__ __forceinline__ float ct_dot2<gf_bf16>(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ct_v2b, a), __builtin_bit_cast(ct_v2b, b), c, false);
}
template <typename T>
__device__ __forceinline__ float ct_dot8(ct_u4 a, ct_u4 b) {
    const unsigned a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;   // (a bit_cast of a swizzle reads element 0)
    return ct_dot2<T>(a3, b3, ct_dot2<T>(a2, b2, ct_dot2<T>(a1, b1, ct_dot2<T>(a0, b0, 0.f))));
}
template <typename T>
__device__ __forceinline__ void ct_axpy8(float (&acc)[8], float p, ct_u4 v);
template <>
__device__ __forceinline__ void ct_axpy8<_Float16>(float (&acc)[8], float p, ct_u4 v) {
    // acc += p * (float)half: one v_fma_mix_f32 per channel (the compiler's choice, two conversions + a packed FMA per pair, is 1.5)
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * j]) : "v"(w[j]), "v"(p));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * j + 1]) : "v"(w[j]), "v"(p));
    }
}
template <>
__device__ __forceinline__ void ct_axpy8<gf_bf16>(float (&acc)[8], float p, ct_u4 v) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[2 * j] += p * __uint_as_float(w[j] << 16);
        acc[2 * j + 1] += p * __uint_as_float(w[j] & 0xffff0000u);
    }
}
template <int CTRL>
__device__ __forceinline__ int dpp_min(int x) {
    return min(x, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
// minimum over the 64 lanes, wave-uniform result: four DPP steps inside each row of 16 lanes, then the four rows through SGPRs
__device__ __forceinline__ int ct_wave_min(int v) {
    v = dpp_min<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_min<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_min<0x141>(v);   // row_half_mirror
    v = dpp_min<0x140>(v);   // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

template <typename T, bool TILED>
__device__ __forceinline__ void ct_attend(const CtArgs& a, const char* sK, const char* sV, const int* tab, ct_u4 qraw,
                                          const char* kbase, const char* vbase, int sub, T* out, bool qok) {
    // tab: this query's 25 entries (LDS): -1 = masked; TILED: cell index inside the staged rectangle, else the cell of the map
    int t[CT_TS];
#pragma unroll
    for (int i = 0; i < CT_TS / 4; ++i) {
        const int4 v = *reinterpret_cast<const int4*>(tab + 4 * i);
        t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
    }
    float dot[CT_WW];
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) {
        const unsigned c = (unsigned)max(t[k], 0);            // a masked key reads row 0 and is overridden afterwards
        ct_u4 kr;
        if (TILED) kr = *reinterpret_cast<const ct_u4*>(sK + (c * 128u + (unsigned)sub * 16u));
        else kr = *reinterpret_cast<const ct_u4*>(kbase + (size_t)c * (size_t)(a.ldk * sizeof(T)));
        dot[k] = ct_dot8<T>(qraw, kr);
    }
    // sum over the 8 lanes of the query; the reductions of the 25 keys interleave
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) dot[k] = dpp_add<0xB1>(dot[k]);     // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) dot[k] = dpp_add<0x4E>(dot[k]);     // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) dot[k] = dpp_add<0x141>(dot[k]);    // row_half_mirror
    // softmax of temperature * logit with masked_fill(-1e8) BEFORE the temperature (geo_attention.py:83,92), in the log2
    // domain: exp2(scale2 * (x - max x)); the maximum is taken on the unscaled logits (scale2 > 0)
    float mx = -INFINITY;
    bool any = false;
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) {
        any = any || t[k] >= 0;
        dot[k] = t[k] >= 0 ? dot[k] : -1e8f;
        mx = fmaxf(mx, dot[k]);
    }
    const float nmx = -mx * a.scale2;
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) {
        dot[k] = __builtin_amdgcn_exp2f(fmaf(dot[k], a.scale2, nmx));    // a masked key: exp2(-1.8e7 - ...) == 0 whenever any key is valid
        den += dot[k];
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CT_WW; ++k) {
        const unsigned c = (unsigned)max(t[k], 0);
        ct_u4 vr;
        if (TILED) vr = *reinterpret_cast<const ct_u4*>(sV + (c * 128u + (unsigned)sub * 16u));
        else vr = *reinterpret_cast<const ct_u4*>(vbase + (size_t)c * (size_t)(a.ldv * sizeof(T)));
        ct_axpy8<T>(acc, dot[k], vr);
    }
    const float rden = any ? __builtin_amdgcn_rcpf(den) : 0.f;          // no valid key: the row is zeroed (geo_attention.py:98-100)
    if (qok) {
        typedef T V8 __attribute__((ext_vector_type(8)));
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)(acc[j] * rden);
        *reinterpret_cast<V8*>(out) = o;
    }
    // the two forms end differently on purpose: the compiler otherwise merges them into one body with a branch around every row read
    if (!TILED) asm volatile("; rows read from global memory" ::: "memory");
}

template <typename T>
__global__ __launch_bounds__(256, CT_WAVES) void window_cross_tiled(CtArgs a) {
    __shared__ __attribute__((aligned(16))) char sK[CT_CAP * 128];
    __shared__ __attribute__((aligned(16))) char sV[CT_CAP * 128];
    __shared__ __attribute__((aligned(16))) int s_tab[CT_Q * CT_TS];
    __shared__ __attribute__((aligned(16))) int s_box[16];    // per wave: ymin, xmin, -ymax, -xmax over its valid window cells
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y;
    // workgroup ids go round-robin over the 8 XCDs: each XCD (= each L2) gets a contiguous band of tiles
    int bx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int head = bx & 3, tile = bx >> 2;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int qi = tid >> 3, sub = tid & 7;                   // 32 queries x 8 lanes
    const int qx = tx * CT_QX + (qi & 7), qy = ty * CT_QY + (qi >> 3);
    const bool qok = qx < a.wq && qy < a.hq;
    const size_t row = (size_t)n * a.L + (size_t)(qy * a.wq + qx);
    T* out = (T*)a.out + row * 256 + head * 64 + sub * 8;
    if (a.valid && !a.valid[n]) {                             // layer skipped for this sample; caller keeps x
        if (qok) *reinterpret_cast<ct_u4*>(out) = ct_u4{0u, 0u, 0u, 0u};
        return;
    }
    ct_u4 qraw{0u, 0u, 0u, 0u};
    if (qok) qraw = *reinterpret_cast<const ct_u4*>((const T*)a.q + row * a.ldq + head * 64 + sub * 8);
    // the tile's window table: entry e = query * 25 + key, four per thread
    int cell[4], cy[4], cx[4];
    int ymin = 0x7fffffff, xmin = 0x7fffffff, nymax = 0x7fffffff, nxmax = 0x7fffffff;
    const float rwk = 1.0f / (float)a.wk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j, eq = e / CT_WW, ek = e - eq * CT_WW;
        const int ex = tx * CT_QX + (eq & 7), ey = ty * CT_QY + (eq >> 3);
        cell[j] = -1;
        if (e < CT_Q * CT_WW && ex < a.wq && ey < a.hq)
            cell[j] = a.win[((size_t)n * a.L + (size_t)(ey * a.wq + ex)) * CT_WW + ek];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cy[j] = cx[j] = 0;
        if (cell[j] >= 0) {
            // cell / wk: the float estimate is within one of the quotient for every S < 2^31 / ld; one correction step
            int y = (int)((float)cell[j] * rwk), x = cell[j] - y * a.wk;
            if (x < 0) { --y; x += a.wk; }
            if (x >= a.wk) { ++y; x -= a.wk; }
            cy[j] = y; cx[j] = x;
            ymin = min(ymin, y); nymax = min(nymax, -y);
            xmin = min(xmin, x); nxmax = min(nxmax, -x);
        }
    }
    ymin = ct_wave_min(ymin); xmin = ct_wave_min(xmin); nymax = ct_wave_min(nymax); nxmax = ct_wave_min(nxmax);
    if (lane == 0) *reinterpret_cast<int4*>(&s_box[4 * wave]) = int4{ymin, xmin, nymax, nxmax};
    __syncthreads();
    int4 box = *reinterpret_cast<const int4*>(&s_box[0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const int4 o = *reinterpret_cast<const int4*>(&s_box[4 * w]);
        box.x = min(box.x, o.x); box.y = min(box.y, o.y); box.z = min(box.z, o.z); box.w = min(box.w, o.w);
    }
    const int y0 = box.x, x0 = box.y;
    if (y0 == 0x7fffffff) {                                   // no valid key anywhere in the tile (geo_attention.py:98-100)
        if (qok) *reinterpret_cast<ct_u4*>(out) = ct_u4{0u, 0u, 0u, 0u};
        return;
    }
    const int bh = -box.z - y0 + 1, bw = -box.w - x0 + 1;
    const bool tiled = bh <= CT_CAP && bw <= CT_CAP && bh * bw <= CT_CAP;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j, eq = e / CT_WW, ek = e - eq * CT_WW;
        if (e < CT_Q * CT_WW) s_tab[eq * CT_TS + ek] = cell[j] < 0 ? -1 : (tiled ? (cy[j] - y0) * bw + (cx[j] - x0) : cell[j]);
    }
    const char* kimg = (const char*)a.kmap + (size_t)n * a.S * a.ldk * sizeof(T);
    const char* vimg = (const char*)a.vmap + (size_t)n * a.S * a.ldv * sizeof(T);
    if (tiled) {
        // 8 cells (x 128 bytes of this head) per request: lane -> (cell = 8 g + lane / 8, 16-byte piece = lane % 8)
        const CtRsrc rk = ct_rsrc(kimg, (unsigned)((size_t)a.S * a.ldk * sizeof(T)));
        const CtRsrc rv = ct_rsrc(vimg, (unsigned)((size_t)a.S * a.ldv * sizeof(T)));
        const int ncell = bh * bw, groups = (ncell + 7) >> 3;
        const float rbw = 1.0f / (float)bw;
        for (int g = wave; g < groups; g += 4) {
            const int c = min(g * 8 + (lane >> 3), ncell - 1);
            const int by = (int)(((float)c + 0.5f) * rbw);    // exact for c, bw <= CT_CAP
            const int cellg = (y0 + by) * a.wk + x0 + (c - by * bw);
            const unsigned piece = (unsigned)head * 128u + (unsigned)(lane & 7) * 16u;
            ct_lds_dma(rk, sK + g * 1024, (int)((unsigned)cellg * (unsigned)(a.ldk * sizeof(T)) + piece));
            ct_lds_dma(rv, sV + g * 1024, (int)((unsigned)cellg * (unsigned)(a.ldv * sizeof(T)) + piece));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const char* kb = kimg + head * 128 + sub * 16;
    const char* vb = vimg + head * 128 + sub * 16;
    if (tiled) ct_attend<T, true>(a, sK, sV, s_tab + qi * CT_TS, qraw, kb, vb, sub, out, qok);
    else ct_attend<T, false>(a, sK, sV, s_tab + qi * CT_TS, qraw, kb, vb, sub, out, qok);
}

}   // namespace

extern "C" int gf_window_geometry(const float* H, const int32_t* valid, int N, int hq, int wq, int Himg, int Wimg,
                                  int wk, int scale, int window_size, const float* window_scale, int32_t* win,
                                  int32_t* kps_or_null, float* warped_or_null, void* stream) {
    GF_CHECK_ARG(H && win, "null pointer");
    GF_CHECK_ARG(N > 0 && hq > 0 && wq > 0 && Himg > 0 && Wimg > 0 && wk > 0 && scale > 0 && window_size > 0, "bad sizes");
    WgArgs a{H, valid, N, hq, wq, Himg, Wimg, wk, scale, window_size, window_scale, win, kps_or_null, warped_or_null};
    window_geometry<<<dim3((hq * wq + 127) / 128, N), 128, 0, (hipStream_t)stream>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_inlier_index(const float* kp0, const float* kp1, const uint8_t* keep, const int32_t* counts,
                               int N, int L, int S, int w0, int w1, int scale, uint8_t* map0, uint8_t* map1,
                               int32_t* idx0, int32_t* idx1, int32_t* nidx, void* stream) {
    GF_CHECK_ARG(kp0 && kp1 && keep && counts && map0 && map1 && idx0 && idx1 && nidx, "null pointer");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0 && w0 > 0 && w1 > 0 && scale > 0, "bad sizes");
    ImArgs a{kp0, kp1, keep, counts, N, L, S, w0, w1, scale, map0, map1, idx0, idx1, nidx};
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(map0, 0, (size_t)N * L, st);
    (void)hipMemsetAsync(map1, 0, (size_t)N * S, st);
    inlier_scatter<<<dim3(8, N), 256, 0, st>>>(a);
    inlier_compact<<<dim3(N, 2), 1024, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_window_cross_attention(const void* q, const void* kmap, const void* vmap, int dtype, int N, int L,
                                         int S, int H, int D, long ldq, long ldk, long ldv, const int32_t* win,
                                         int WW, const int32_t* valid, void* out, void* stream) {
    GF_CHECK_ARG(q && kmap && vmap && win && out, "null pointer");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(H == 4 && D == 64 && WW == 25, "built for nhead=4, head dim 64, 5x5 windows (geo_config.py:12,16)");
    GF_CHECK_ARG((double)S * (double)(ldk > ldv ? ldk : ldv) < 4294967296.0, "key map too large for 32-bit row offsets");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    CaArgs a{q, kmap, vmap, ldq, ldk, ldv, win, valid, out, N, L, S, WW, 1.0f / sqrtf((float)D)};
    const dim3 grid((L + 3) / 4, N);
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes: q read, out written, each projected key / value map read once (served from L2 after that)
    void* pt = gf_prof_begin("k5_window_attention", st, (double)N * (2.0 * L + 2.0 * S) * H * D * (dtype == GF_F32 ? 4 : 2));
    if (dtype == GF_F32) window_cross_attention<float, 25><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_F16) window_cross_attention<_Float16, 25><<<grid, 256, 0, st>>>(a);
    else window_cross_attention<gf_bf16, 25><<<grid, 256, 0, st>>>(a);
    gf_prof_end("k5_window_attention", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// dq [N, L, 256] of `dtype`, dk / dv fp32 [N, S, 256] (ZEROED by the caller: the kernel adds into them) of gf_window_cross_attention
extern "C" int gf_window_cross_attention_backward(const void* q, const void* kmap, const void* vmap, const void* dout, int dtype, int N, int L,
                                                  int S, int H, int D, long ldq, long ldk, long ldv, const int32_t* win, int WW, void* dq,
                                                  float* dk, float* dv, void* stream) {
    GF_CHECK_ARG(q && kmap && vmap && dout && win && dq && dk && dv, "null pointer");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(H == 4 && D == 64 && WW == 25, "built for nhead=4, head dim 64, 5x5 windows (geo_config.py:12,16)");
    GF_CHECK_ARG((double)S * (double)(ldk > ldv ? ldk : ldv) < 4294967296.0, "key map too large for 32-bit row offsets");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    CbArgs a{};
    a.q = q; a.kmap = kmap; a.vmap = vmap; a.dout = dout; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.win = win; a.dq = dq; a.dk = dk; a.dv = dv;
    a.N = N; a.L = L; a.S = S; a.softmax_temp = 1.0f / sqrtf((float)D);
    const dim3 grid((L + 3) / 4, N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GF_F32) window_cross_attention_backward<float, 25, false><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_F16) window_cross_attention_backward<_Float16, 25, false><<<grid, 256, 0, st>>>(a);
    else window_cross_attention_backward<gf_bf16, 25, false><<<grid, 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" size_t gf_window_cross_attention_backward_workspace_bytes(int N, int L, int WW) {
    return (N <= 0 || L <= 0 || WW <= 0) ? 0 : gf_align_up((size_t)N * L * WW * 4 * sizeof(float2), 256);
}

// The same gradients WITHOUT atomics: dk, dv [N][S][256] of `dtype` are gathered cell by cell along the caller's inverse index of `win`
// (entries sorted by global cell n * S + cell, stable; offsets [N * S + 1]) - bit-reproducible, and the fp32 maps are gone.
extern "C" int gf_window_cross_attention_backward_gather(const void* q, const void* kmap, const void* vmap, const void* dout, int dtype, int N,
                                                         int L, int S, int H, int D, long ldq, long ldk, long ldv, const int32_t* win, int WW,
                                                         const int32_t* entries, const int32_t* offsets, void* dq, void* dk, void* dv,
                                                         void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(q && kmap && vmap && dout && win && entries && offsets && dq && dk && dv && workspace, "null pointer");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(H == 4 && D == 64 && WW == 25, "built for nhead=4, head dim 64, 5x5 windows (geo_config.py:12,16)");
    GF_CHECK_ARG((double)S * (double)(ldk > ldv ? ldk : ldv) < 4294967296.0, "key map too large for 32-bit row offsets");
    GF_CHECK_ARG((double)N * L * WW < 2147483648.0 && (double)N * S < 2147483647.0, "index ranges");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(workspace_bytes >= gf_window_cross_attention_backward_workspace_bytes(N, L, WW), "workspace too small");
    CbArgs a{};
    a.q = q; a.kmap = kmap; a.vmap = vmap; a.dout = dout; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.win = win; a.dq = dq;
    a.N = N; a.L = L; a.S = S; a.softmax_temp = 1.0f / sqrtf((float)D);
    a.dlp = (float2*)workspace; a.entries = entries; a.offsets = offsets; a.dk16 = dk; a.dv16 = dv;
    const dim3 grid((L + 3) / 4, N);
    const unsigned cells = (unsigned)(((long)N * S + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GF_F32) {
        window_cross_attention_backward<float, 25, true><<<grid, 256, 0, st>>>(a);
        window_cross_gather<float, 25><<<cells, 256, 0, st>>>(a);
    } else if (dtype == GF_F16) {
        window_cross_attention_backward<_Float16, 25, true><<<grid, 256, 0, st>>>(a);
        window_cross_gather<_Float16, 25><<<cells, 256, 0, st>>>(a);
    } else {
        window_cross_attention_backward<gf_bf16, 25, true><<<grid, 256, 0, st>>>(a);
        window_cross_gather<gf_bf16, 25><<<cells, 256, 0, st>>>(a);
    }
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_window_cross_attention_tiled(const void* q, const void* kmap, const void* vmap, int dtype, int N, int hq,
                                               int wq, int hk, int wk, int H, int D, long ldq, long ldk, long ldv,
                                               const int32_t* win, int WW, const int32_t* valid, void* out, void* stream) {
    GF_CHECK_ARG(q && kmap && vmap && win && out, "null pointer");
    GF_CHECK_ARG(N > 0 && hq > 0 && wq > 0 && hk > 0 && wk > 0, "empty problem");
    GF_CHECK_ARG(H == 4 && D == 64 && WW == 25, "built for nhead=4, head dim 64, 5x5 windows (geo_config.py:12,16)");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    const int L = hq * wq, S = hk * wk;
    if (dtype == GF_F32)                    // exact-fp32 mode: the one-wave-per-query form
        return gf_window_cross_attention(q, kmap, vmap, dtype, N, L, S, H, D, ldq, ldk, ldv, win, WW, valid, out, stream);
    GF_CHECK_ARG((double)S * (double)(ldk > ldv ? ldk : ldv) * 2.0 < 4294967296.0, "key map too large for 32-bit byte offsets");
    GF_CHECK_ARG(hk < 32768 && wk < 32768 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "rows must be 16-byte aligned");
    const int tiles_x = (wq + CT_QX - 1) / CT_QX, tiles_y = (hq + CT_QY - 1) / CT_QY;
    CtArgs a{q, kmap, vmap, ldq, ldk, ldv, win, valid, out, N, L, S, hq, wq, wk, tiles_x, 1.4426950408889634f / sqrtf((float)D)};
    const dim3 grid(tiles_x * tiles_y * 4, N);
    hipStream_t st = (hipStream_t)stream;
    void* pt = gf_prof_begin("k5_window_attention", st, (double)N * (2.0 * L + 2.0 * S) * H * D * 2);
    if (dtype == GF_F16) window_cross_tiled<_Float16><<<grid, 256, 0, st>>>(a);
    else window_cross_tiled<gf_bf16><<<grid, 256, 0, st>>>(a);
    gf_prof_end("k5_window_attention", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
