// On-device homography RANSAC for GeoModule (replaces the cv2.findHomography host round trip of
// model/geo_module.py:45-52).  OpenCV parity is UNPINNED (no OpenCV in the reference tree, no test
// fixing its output); this kernel implements, operation for operation and in fp64 with contraction
// off, the algorithm stated in oracle/ransac_oracle.c, so the inlier mask can be checked bit-exactly.
//
//   ransac_score : grid (T/4, N), one wave per hypothesis: lane 0 draws 4 correspondences with the
//                  counter-based hash and solves the 8x8 system; all lanes count inliers.
//   ransac_final : grid N: best hypothesis (most inliers, then smallest t), inlier mask, Hartley-
//                  normalised least-squares refit (block reductions), M, M^-1 (fp64 adjugate), fp32 casts.
// No host synchronisation: match counts are read from device memory.
#include <math.h>

#include "gf_common.h"

#pragma clang fp contract(off)

namespace {

struct RsArgs {
    const float* mk0;        // [cap][2] matched keypoints (px) of image0, sorted by sample
    const float* mk1;
    const int32_t* counts;   // [1+N]: total, per sample
    int N, iters;
    float scale;             // hw0_i[0] // hw0_c[0]
    const float* scale0;     // [N][2] or null
    const float* scale1;
    double thr2;
    uint32_t seed;
    float* kp0;              // [cap][2] keypoints fed to RANSAC: integer-valued (geo_module.py:110-111, 38-43) or raw
    float* kp1;
    int lm_iters;            // Levenberg-Marquardt steps behind the least-squares refit (OpenCV's findHomography: 10)
    int min_points;          // a sample with fewer matches gets no model (GeoModule: 9, i.e. len > 8, geo_module.py:46)
    int integer_kp;          // 1: the reference's .long() keypoints (GeoModule); 0: sub-pixel keypoints as given (eval)
    double* hyp;             // [N][iters][9]
    int32_t* hyp_cnt;        // [N][iters]  (-1 = invalid hypothesis)
    double* M;               // [N][9]
    float* Mf;               // [N][9]   M cast to fp32           (geo_module.py:58)
    float* Minv;             // [N][9]   inverse(M) in fp64, cast (geo_module.py:67)
    int32_t* valid;          // [N]
    uint8_t* keep;           // [cap] 1 = match feeds the inlier maps (inlier, or any match when no model)
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t draw(uint32_t seed, uint32_t sample, uint32_t t, uint32_t k, uint32_t attempt) {
    uint32_t x = seed * 0x9E3779B1u;
    x = mix32(x ^ (sample + 0x7F4A7C15u));
    x = mix32(x ^ (t * 0x85EBCA6Bu + 0x165667B1u));
    x = mix32(x ^ (k * 0xC2B2AE35u + 0x27D4EB2Fu));
    x = mix32(x ^ (attempt * 0x9E3779B1u + 0x61C88647u));
    return x;
}

__device__ int rs_solve(double* a, double* b, int n) {
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = fabs(a[r * n + c]);
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 1e-12)) return 0;
        if (p != c) {
            for (int k = 0; k < n; ++k) { const double tmp = a[c * n + k]; a[c * n + k] = a[p * n + k]; a[p * n + k] = tmp; }
            const double tb = b[c]; b[c] = b[p]; b[p] = tb;
        }
        const double inv = 1.0 / a[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const double f = a[r * n + c] * inv;
            if (f != 0.0) {
                for (int k = c; k < n; ++k) a[r * n + k] = a[r * n + k] - f * a[c * n + k];
                b[r] = b[r] - f * b[c];
            }
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double s = b[c];
        for (int k = c + 1; k < n; ++k) s = s - a[c * n + k] * b[k];
        b[c] = s / a[c * n + c];
    }
    return 1;
}

__device__ __forceinline__ int rs_inlier(const double* h, double x, double y, double u, double v, double thr2) {
    const double w = h[6] * x + h[7] * y + h[8];
    if (w == 0.0) return 0;
    const double px = (h[0] * x + h[1] * y + h[2]) / w;
    const double py = (h[3] * x + h[4] * y + h[5]) / w;
    const double dx = px - u, dy = py - v;
    return dx * dx + dy * dy <= thr2;
}

__device__ __forceinline__ void rs_range(const RsArgs& a, int n, int& off, int& cnt) {
    off = 0;
    for (int b = 0; b < n; ++b) off += a.counts[1 + b];
    cnt = a.counts[1 + n];
}

// integer keypoints exactly as the reference derives them: .long() of the float px coordinates,
// then (only with per-image scales) kp / (scale*scale0[b]) * scale and .long() again
__global__ void ransac_keypoints(RsArgs a) {
    const int n = blockIdx.y;
    int off, cnt;
    rs_range(a, n, off, cnt);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!a.integer_kp) {
                a.kp0[2 * (off + i) + c] = a.mk0[2 * (off + i) + c];
                a.kp1[2 * (off + i) + c] = a.mk1[2 * (off + i) + c];
                continue;
            }
            long k0 = (long)a.mk0[2 * (off + i) + c], k1 = (long)a.mk1[2 * (off + i) + c];
            if (a.scale0) {
                k0 = (long)((float)k0 / (a.scale * a.scale0[2 * n + c]) * a.scale);
                k1 = (long)((float)k1 / (a.scale * a.scale1[2 * n + c]) * a.scale);
            }
            a.kp0[2 * (off + i) + c] = (float)k0;
            a.kp1[2 * (off + i) + c] = (float)k1;
        }
    }
}

__global__ __launch_bounds__(256) void ransac_score(RsArgs a) {
    __shared__ double sh_h[4][9];
    __shared__ int sh_ok[4];
    const int n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    int off, cnt;
    rs_range(a, n, off, cnt);
    if (cnt < a.min_points || t >= a.iters) return;  // geo_module.py:46 (uniform per block / wave)
    const float* k0 = a.kp0 + 2 * (size_t)off;
    const float* k1 = a.kp1 + 2 * (size_t)off;
    if (lane == 0) {
        int idx[4], ok = 1;
        for (int k = 0; k < 4 && ok; ++k) {
            int found = 0;
            for (uint32_t attempt = 0; attempt < 16 && !found; ++attempt) {
                const int c = (int)(draw(a.seed, (uint32_t)n, (uint32_t)t, (uint32_t)k, attempt) % (uint32_t)cnt);
                int dup = 0;
                for (int j = 0; j < k; ++j) dup |= (idx[j] == c);
                if (!dup) { idx[k] = c; found = 1; }
            }
            ok = found;
        }
        double m[64], b[8];
        if (ok) {
            for (int k = 0; k < 4; ++k) {
                const double x = (double)k0[2 * idx[k]], y = (double)k0[2 * idx[k] + 1];
                const double u = (double)k1[2 * idx[k]], v = (double)k1[2 * idx[k] + 1];
                double* r0 = m + (2 * k) * 8;
                double* r1 = m + (2 * k + 1) * 8;
                r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -(u * x); r0[7] = -(u * y);
                r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -(v * x); r1[7] = -(v * y);
                b[2 * k] = u; b[2 * k + 1] = v;
            }
            ok = rs_solve(m, b, 8);
        }
        sh_ok[wave] = ok;
        if (ok) {
            for (int k = 0; k < 8; ++k) sh_h[wave][k] = b[k];
            sh_h[wave][8] = 1.0;
        }
    }
    __syncthreads();
    int32_t* out_cnt = a.hyp_cnt + (size_t)n * a.iters + t;
    if (!sh_ok[wave]) {
        if (lane == 0) *out_cnt = -1;
        return;
    }
    double h[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = sh_h[wave][k];
    int c = 0;
    for (int i = lane; i < cnt; i += 64)
        c += rs_inlier(h, (double)k0[2 * i], (double)k0[2 * i + 1], (double)k1[2 * i], (double)k1[2 * i + 1], a.thr2);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane == 0) {
        *out_cnt = c;
        double* hp = a.hyp + ((size_t)n * a.iters + t) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) hp[k] = h[k];
    }
}

// sum over the 256 threads with TWO barriers for all NV values: lanes by shuffles, the four waves through LDS (added in wave
// order).  `part` = [4][NV] doubles of LDS; the result is returned in every thread.
template <int NV>
__device__ void block_sum_fast(double (&v)[NV], double* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double x = v[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(x);
            const unsigned lo = __shfl_xor((unsigned)b, d, 64), hi = __shfl_xor((unsigned)(b >> 32), d, 64);
            x = x + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
        v[k] = x;
    }
    __syncthreads();                                 // the previous call's readers are done with `part`
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) part[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ((part[k] + part[NV + k]) + part[2 * NV + k]) + part[3 * NV + k];
}

__global__ __launch_bounds__(256) void ransac_final(RsArgs a) {
    __shared__ double sh[256];
    __shared__ double sh_lm[9];
    __shared__ int sh_lm_ok;
    __shared__ long long sh_key[256];
    __shared__ double sh_h[9];
    __shared__ int sh_state;
    const int n = blockIdx.x, t = threadIdx.x;
    int off, cnt;
    rs_range(a, n, off, cnt);
    uint8_t* keep = a.keep + off;
    const float* k0 = a.kp0 + 2 * (size_t)off;
    const float* k1 = a.kp1 + 2 * (size_t)off;
    // ---- best hypothesis: max count, ties -> smallest t  (key = count * 2^32 + (2^31 - t))
    long long key = -1;
    if (cnt >= a.min_points)
        for (int i = t; i < a.iters; i += 256) {
            const int c = a.hyp_cnt[(size_t)n * a.iters + i];
            if (c >= 0) {
                const long long kk = ((long long)c << 32) | (long long)(0x7FFFFFFF - i);
                key = kk > key ? kk : key;
            }
        }
    sh_key[t] = key;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) sh_key[t] = sh_key[t] > sh_key[t + s] ? sh_key[t] : sh_key[t + s];
        __syncthreads();
    }
    key = sh_key[0];
    const int best_cnt = key < 0 ? -1 : (int)(key >> 32);
    const int best_t = key < 0 ? -1 : 0x7FFFFFFF - (int)(key & 0xFFFFFFFFll);
    const bool have = best_cnt >= 4;
    if (!have) {                                   // no model: every match feeds the maps (geo_module.py:77-94)
        for (int i = t; i < cnt; i += 256) keep[i] = 1;
        if (t == 0) {
            a.valid[n] = 0;
            for (int k = 0; k < 9; ++k) { a.M[9 * n + k] = 0.0; a.Mf[9 * n + k] = 0.f; a.Minv[9 * n + k] = 0.f; }
        }
        return;
    }
    if (t < 9) sh_h[t] = a.hyp[((size_t)n * a.iters + best_t) * 9 + t];
    __syncthreads();
    double h[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) h[k] = sh_h[k];
    // ---- mask + centroids
    double cen[5] = {0, 0, 0, 0, 0};
    for (int i = t; i < cnt; i += 256) {
        const double x = (double)k0[2 * i], y = (double)k0[2 * i + 1], u = (double)k1[2 * i], v = (double)k1[2 * i + 1];
        const int in = rs_inlier(h, x, y, u, v, a.thr2);
        keep[i] = (uint8_t)in;
        if (in) { cen[0] += x; cen[1] += y; cen[2] += u; cen[3] += v; cen[4] += 1.0; }
    }
    block_sum_fast(cen, sh);
    const double m = cen[4], c0x = cen[0] / m, c0y = cen[1] / m, c1x = cen[2] / m, c1y = cen[3] / m;
    double dd[2] = {0, 0};
    for (int i = t; i < cnt; i += 256)
        if (keep[i]) {
            const double x = (double)k0[2 * i], y = (double)k0[2 * i + 1], u = (double)k1[2 * i], v = (double)k1[2 * i + 1];
            dd[0] += sqrt((x - c0x) * (x - c0x) + (y - c0y) * (y - c0y));
            dd[1] += sqrt((u - c1x) * (u - c1x) + (v - c1y) * (v - c1y));
        }
    block_sum_fast(dd, sh);
    const double s0 = dd[0] > 0 ? sqrt(2.0) * m / dd[0] : 1.0, s1 = dd[1] > 0 ? sqrt(2.0) * m / dd[1] : 1.0;
    // ---- normal equations (upper triangle of the symmetric 8x8 + rhs)
    double ne[44];
#pragma unroll
    for (int k = 0; k < 44; ++k) ne[k] = 0.0;
    for (int i = t; i < cnt; i += 256)
        if (keep[i]) {
            const double x = ((double)k0[2 * i] - c0x) * s0, y = ((double)k0[2 * i + 1] - c0y) * s0;
            const double u = ((double)k1[2 * i] - c1x) * s1, v = ((double)k1[2 * i + 1] - c1y) * s1;
            const double r0[8] = {x, y, 1, 0, 0, 0, -(u * x), -(u * y)};
            const double r1[8] = {0, 0, 0, x, y, 1, -(v * x), -(v * y)};
            int q = 0;
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) {
#pragma unroll
                for (int j2 = i2; j2 < 8; ++j2) ne[q++] += r0[i2] * r0[j2] + r1[i2] * r1[j2];
            }
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) ne[36 + i2] += r0[i2] * u + r1[i2] * v;
        }
    block_sum_fast(ne, sh);
    if (t == 0) {
        double ata[64], atb[8], g[9];                   // (this g: the refit's result, handed on through LDS)
        int q = 0;
        for (int i2 = 0; i2 < 8; ++i2)
            for (int j2 = i2; j2 < 8; ++j2) { ata[i2 * 8 + j2] = ne[q]; ata[j2 * 8 + i2] = ne[q]; ++q; }
        for (int i2 = 0; i2 < 8; ++i2) atb[i2] = ne[36 + i2];
        bool refit = rs_solve(ata, atb, 8) != 0;
        if (refit) {
            double hn[9], am[9];
            for (int k = 0; k < 8; ++k) hn[k] = atb[k];
            hn[8] = 1.0;
            for (int r = 0; r < 3; ++r) {
                am[3 * r + 0] = hn[3 * r + 0] * s0;
                am[3 * r + 1] = hn[3 * r + 1] * s0;
                am[3 * r + 2] = hn[3 * r + 2] - s0 * (hn[3 * r + 0] * c0x + hn[3 * r + 1] * c0y);
            }
            for (int c = 0; c < 3; ++c) {
                g[c] = am[c] / s1 + c1x * am[6 + c];
                g[3 + c] = am[3 + c] / s1 + c1y * am[6 + c];
                g[6 + c] = am[6 + c];
            }
            refit = fabs(g[8]) > 1e-12;
            if (refit)
                for (int k = 0; k < 9; ++k) g[k] = g[k] / g[8];
        }
        if (!refit)
            for (int k = 0; k < 9; ++k) g[k] = h[k];
        for (int k = 0; k < 9; ++k) sh_lm[k] = g[k];
    }
    __syncthreads();
    // ---- Levenberg-Marquardt on the inliers' forward transfer error, 8 free entries (h33 = 1): what OpenCV's findHomography
    // appends to its RANSAC (oracle/ransac_oracle.c:lm_refine states the same steps); the mask is not touched
    double g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = sh_lm[k];
    if (a.lm_iters > 0) {
        auto sq_error = [&](const double* hh) {
            double e[1] = {0.0};
            for (int i = t; i < cnt; i += 256)
                if (keep[i]) {
                    const double x = (double)k0[2 * i], y = (double)k0[2 * i + 1];
                    const double iw = 1.0 / (hh[6] * x + hh[7] * y + 1.0);
                    const double du = (hh[0] * x + hh[1] * y + hh[2]) * iw - (double)k1[2 * i], dv = (hh[3] * x + hh[4] * y + hh[5]) * iw - (double)k1[2 * i + 1];
                    e[0] += du * du + dv * dv;
                }
            block_sum_fast(e, sh);
            return e[0];
        };
        double lambda = 1e-3, err = sq_error(g);
        for (int it = 0; it < a.lm_iters; ++it) {
            double ne2[44];                              // upper triangle of J^T J (36) | J^T r (8)
#pragma unroll
            for (int k = 0; k < 44; ++k) ne2[k] = 0.0;
            for (int i = t; i < cnt; i += 256)
                if (keep[i]) {
                    const double x = (double)k0[2 * i], y = (double)k0[2 * i + 1];
                    const double iw = 1.0 / (g[6] * x + g[7] * y + 1.0);
                    const double up = (g[0] * x + g[1] * y + g[2]) * iw, vp = (g[3] * x + g[4] * y + g[5]) * iw;
                    const double ju[8] = {x * iw, y * iw, iw, 0, 0, 0, -(x * up) * iw, -(y * up) * iw};
                    const double jv[8] = {0, 0, 0, x * iw, y * iw, iw, -(x * vp) * iw, -(y * vp) * iw};
                    const double ru = up - (double)k1[2 * i], rv = vp - (double)k1[2 * i + 1];
                    int q = 0;
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) {
#pragma unroll
                        for (int j2 = i2; j2 < 8; ++j2) ne2[q++] += ju[i2] * ju[j2] + jv[i2] * jv[j2];
                    }
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) ne2[36 + i2] += ju[i2] * ru + jv[i2] * rv;
                }
            block_sum_fast(ne2, sh);
            if (t == 0) {
                double A[64], d[8];
                int q = 0;
                for (int i2 = 0; i2 < 8; ++i2)
                    for (int j2 = i2; j2 < 8; ++j2) { A[i2 * 8 + j2] = ne2[q]; A[j2 * 8 + i2] = ne2[q]; ++q; }
                for (int i2 = 0; i2 < 8; ++i2) { A[i2 * 8 + i2] = A[i2 * 8 + i2] + lambda * A[i2 * 8 + i2]; d[i2] = -ne2[36 + i2]; }
                const int ok = rs_solve(A, d, 8);
                sh_lm_ok = ok;
                if (ok) {
                    for (int k = 0; k < 8; ++k) sh_lm[k] = g[k] + d[k];
                    sh_lm[8] = 1.0;
                }
            }
            __syncthreads();
            if (!sh_lm_ok) {
                lambda = lambda * 10.0;
                __syncthreads();
                continue;
            }
            double hn[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) hn[k] = sh_lm[k];
            const double en = sq_error(hn);              // (its barriers also fence sh_lm against the next iteration's writer)
            if (en < err) {                              // (every thread holds the same en / err: the exits are uniform)
                const double gain = err - en;
#pragma unroll
                for (int k = 0; k < 9; ++k) g[k] = hn[k];
                lambda = lambda * 0.1;
                const bool done = gain <= 1e-10 * err;   // converged: the refit's minimum is usually 2-3 steps away
                err = en;
                if (done) break;
            } else {
                lambda = lambda * 10.0;
                if (lambda > 1e6) break;
            }
        }
    }
    if (t == 0) {
        // inverse in fp64 (adjugate), as torch.inverse is applied to the float64 matrix before the cast
        const double det = g[0] * (g[4] * g[8] - g[5] * g[7]) - g[1] * (g[3] * g[8] - g[5] * g[6]) +
                           g[2] * (g[3] * g[7] - g[4] * g[6]);
        double inv[9];
        inv[0] = (g[4] * g[8] - g[5] * g[7]) / det; inv[1] = (g[2] * g[7] - g[1] * g[8]) / det; inv[2] = (g[1] * g[5] - g[2] * g[4]) / det;
        inv[3] = (g[5] * g[6] - g[3] * g[8]) / det; inv[4] = (g[0] * g[8] - g[2] * g[6]) / det; inv[5] = (g[2] * g[3] - g[0] * g[5]) / det;
        inv[6] = (g[3] * g[7] - g[4] * g[6]) / det; inv[7] = (g[1] * g[6] - g[0] * g[7]) / det; inv[8] = (g[0] * g[4] - g[1] * g[3]) / det;
        for (int k = 0; k < 9; ++k) {
            a.M[9 * n + k] = g[k];
            a.Mf[9 * n + k] = (float)g[k];
            a.Minv[9 * n + k] = (float)inv[k];
        }
        a.valid[n] = 1;
    }
}

}   // namespace

extern "C" size_t gf_ransac_workspace_bytes(int N, int iters) {
    if (N <= 0 || iters <= 0) return 0;
    return gf_align_up((size_t)N * iters * 9 * sizeof(double), 256) + gf_align_up((size_t)N * iters * sizeof(int32_t), 256);
}

extern "C" int gf_ransac_homography_v2(const float* mkpts0_c, const float* mkpts1_c, const int32_t* counts, int N,
                                       int capacity, float scale, const float* scale0, const float* scale1,
                                       float thr, int iters, uint32_t seed, int min_points, int integer_keypoints, float* kp0,
                                       float* kp1, double* M,
                                       float* M_f32, float* Minv_f32, int32_t* valid, uint8_t* keep, void* workspace,
                                       size_t workspace_bytes, void* stream, int lm_iters) {
    GF_CHECK_ARG(mkpts0_c && mkpts1_c && counts && kp0 && kp1 && M && M_f32 && Minv_f32 && valid && keep, "null pointer");
    GF_CHECK_ARG(N > 0 && capacity > 0 && iters > 0 && iters % 4 == 0, "need N, capacity > 0 and iters a positive multiple of 4");
    GF_CHECK_ARG((scale0 == nullptr) == (scale1 == nullptr), "scale0/scale1 must both be set or both be NULL");
    if (workspace == nullptr || workspace_bytes < gf_ransac_workspace_bytes(N, iters)) {
        gf_set_error("gf_ransac_homography_v2: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    RsArgs a;
    a.mk0 = mkpts0_c; a.mk1 = mkpts1_c; a.counts = counts; a.N = N; a.iters = iters; a.scale = scale;
    a.scale0 = scale0; a.scale1 = scale1; a.thr2 = (double)thr * (double)thr; a.seed = seed;
    a.kp0 = kp0; a.kp1 = kp1; a.integer_kp = integer_keypoints; a.min_points = min_points < 4 ? 4 : min_points;
    a.lm_iters = lm_iters < 0 ? 0 : lm_iters;
    a.hyp = (double*)workspace;
    a.hyp_cnt = (int32_t*)((char*)workspace + gf_align_up((size_t)N * iters * 9 * sizeof(double), 256));
    a.M = M; a.Mf = M_f32; a.Minv = Minv_f32; a.valid = valid; a.keep = keep;
    hipStream_t st = (hipStream_t)stream;
    const int kb = (capacity / N + 255) / 256;
    ransac_keypoints<<<dim3(kb < 1 ? 1 : (kb > 64 ? 64 : kb), N), 256, 0, st>>>(a);
    ransac_score<<<dim3(iters / 4, N), 256, 0, st>>>(a);
    ransac_final<<<N, 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// the version-1 entry point (rounds 1-2): no refinement behind the refit - callers built against that header keep their M
extern "C" int gf_ransac_homography(const float* mkpts0_c, const float* mkpts1_c, const int32_t* counts, int N,
                                    int capacity, float scale, const float* scale0, const float* scale1,
                                    float thr, int iters, uint32_t seed, int min_points, int integer_keypoints, float* kp0,
                                    float* kp1, double* M,
                                    float* M_f32, float* Minv_f32, int32_t* valid, uint8_t* keep, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return gf_ransac_homography_v2(mkpts0_c, mkpts1_c, counts, N, capacity, scale, scale0, scale1, thr, iters, seed, min_points,
                                   integer_keypoints, kp0, kp1, M, M_f32, Minv_f32, valid, keep, workspace, workspace_bytes, stream, 0);
}
