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

template <typename T, int WW>
__global__ __launch_bounds__(256) void window_cross_attention(CaArgs a) {
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= a.L) return;
    T* out = (T*)a.out + ((size_t)n * a.L + l) * 256 + lane * 4;
    if (a.valid && !a.valid[n]) {          // layer skipped for this sample; caller keeps x
        store4<T>(out, v4f{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const v4f q = load4<T>((const T*)a.q + ((size_t)n * a.L + l) * a.ldq + lane * 4);
    const int32_t* win = a.win + ((size_t)n * a.L + l) * WW;
    const T* kb = (const T*)a.kmap + (size_t)n * a.S * a.ldk + lane * 4;
    const T* vb = (const T*)a.vmap + (size_t)n * a.S * a.ldv + lane * 4;
    float logit[WW];
    int cell[WW];
#pragma unroll
    for (int k = 0; k < WW; ++k) cell[k] = win[k];
    // branch-free gathers: a masked key reads row 0 and is overridden afterwards, so the 25 row loads are issued
    // back to back and the 16-lane DPP reductions of different keys interleave (the per-key branches had serialised
    // load -> 4 dependent DPP adds -> next key, with two wait states in front of every DPP read)
    float dot[WW];
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        const v4f kv = load4<T>(kb + (unsigned)max(cell[k], 0) * (unsigned)a.ldk);     // 32-bit offsets: S * ld < 2^32
        dot[k] = q.x * kv.x + q.y * kv.y + q.z * kv.z + q.w * kv.w;
    }
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0xB1>(dot[k]);     // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0x4E>(dot[k]);     // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0x141>(dot[k]);    // row_half_mirror
#pragma unroll
    for (int k = 0; k < WW; ++k) dot[k] = dpp_add<0x140>(dot[k]);    // row_mirror
    float mx = -INFINITY;
    bool any = false;
#pragma unroll
    for (int k = 0; k < WW; ++k) {
        any = any || cell[k] >= 0;
        logit[k] = (cell[k] >= 0 ? dot[k] : -1e8f) * a.softmax_temp;   // masked_fill BEFORE the temperature (geo_attention.py:83,92)
        mx = fmaxf(mx, logit[k]);
    }
    constexpr bool FAST = !std::is_same<T, float>::value;     // fp16 mode: hardware exp / reciprocal
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
            const v4f vv = load4<T>(vb + (unsigned)max(cell[k], 0) * (unsigned)a.ldv);
            acc.x += p * vv.x; acc.y += p * vv.y; acc.z += p * vv.z; acc.w += p * vv.w;
        }
    }                                       // no valid key: the row is zeroed (geo_attention.py:98-100)
    store4<T>(out, acc);
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
