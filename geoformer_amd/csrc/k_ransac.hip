// On-device homography RANSAC for GeoModule (replaces the cv2.findHomography host round trip of
// model/geo_module.py:45-52).  OpenCV parity is UNPINNED (no OpenCV in the reference tree, no test
// fixing its output); this kernel implements, operation for operation and in fp64 with contraction
// off, the algorithm stated in oracle/ransac_oracle.c, so the inlier mask can be checked bit-exactly.
//
//   ransac_score : grid (T/32, N), 8 hypotheses per wave: each group of 8 lanes draws 4 correspondences with the
//                  counter-based hash and solves its 8x8 system with ONE ROW PER LANE (rs_solve8: the oracle's
//                  elimination, every element operation unchanged, rows exchanged by shuffles); then all 64 lanes
//                  count the inliers of the wave's 8 hypotheses in one pass over the matches.
//                  [Round 3: one hypothesis per wave, lane 0 solving alone on a scratch-memory matrix - the serial
//                  solve, not the scoring, was the kernel's 144 us.]
//   ransac_final : grid N: best hypothesis (most inliers, then smallest t), inlier mask, Hartley-
//                  normalised least-squares refit (block reductions; the 8x8 solves on eight lanes as above),
//                  Levenberg-Marquardt steps, M, M^-1 (fp64 adjugate), fp32 casts.
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
        if (!(best == best)) best = INFINITY;              /* a non-finite entry in the column fails the solve (round 5: NaN keypoints) */
        for (int r = c + 1; r < n; ++r) {
            double v = fabs(a[r * n + c]);
            if (!(v == v)) v = INFINITY;
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 1e-12) || best == INFINITY) return 0;
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

// rs_solve with the 8 rows of the system on 8 consecutive lanes (lane r of the group holds row r: a[0..7] and b in registers):
// the same partial-pivot elimination, element operation for element operation - a row operation a[r][k] - f a[c][k] is the same
// IEEE operation on whichever lane it runs, the pivot is the FIRST row of maximal |a[r][c]| as in the serial loop, and the back
// substitution subtracts a[c][k] x[k] in ascending k - so the solution has the serial code's bits (the inlier mask test against
// oracle/ransac_oracle.c requires it).  Every lane of the group returns the solution x[0..7] and the ok flag.  The caller's
// lanes must all be active (shuffles); groups are independent.
__device__ __forceinline__ double rs_shfl(double v, int src) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __shfl((unsigned)b, src, 64), hi = __shfl((unsigned)(b >> 32), src, 64);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int rs_solve8(double (&a)[8], double& b, double (&x)[8]) {
    const int lane = threadIdx.x & 63, g0 = lane & ~7, r = lane & 7;
    int ok = 1;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // pivot: first row >= c of maximal |a[r][c]|
        // (a NaN compares false both ways: the lanes of a group would disagree on (best, p), shuffle different rows and end with
        // different ok flags - ADVICE r04.  A non-finite entry counts as +inf: it wins the maximum on every lane alike and fails the
        // solve below, as the serial code's NaN solution fails every inlier test.)
        const double av = fabs(a[c]);
        double best = r >= c ? (av == av ? av : INFINITY) : -1.0;
        int p = r;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            const double ob = rs_shfl(best, lane ^ d);
            const int op = __shfl(p, lane ^ d, 64);
            const bool take = ob > best || (ob == best && op < p);
            best = take ? ob : best;
            p = take ? op : p;
        }
        if (!(best > 1e-12) || best == INFINITY) ok = 0;   // (uniform in the group; the remaining steps run on, the result is discarded)
        // swap rows c and p
        const int src = g0 + (r == c ? p : (r == p ? c : r));
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = rs_shfl(a[k], src);
        b = rs_shfl(b, src);
        // eliminate below: the pivot row to every lane
        double pr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pr[k] = k >= c ? rs_shfl(a[k], g0 + c) : 0.0;
        const double pb = rs_shfl(b, g0 + c);
        const double inv = 1.0 / pr[c];
        const double f = a[c] * inv;
        if (r > c && f != 0.0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k >= c) a[k] = a[k] - f * pr[k];
            b = b - f * pb;
        }
    }
#pragma unroll
    for (int c = 7; c >= 0; --c) {
        double s = b;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k > c) s = s - a[k] * x[k];
        x[c] = rs_shfl(s / a[c], g0 + c);                   // lane c holds row c: its value is the one that counts
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) ok &= __shfl(ok, lane ^ d, 64);                  // one flag per group, whatever the values were
    return ok;
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
    const int n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 3, r = lane & 7;
    const int t = (blockIdx.x * 4 + wave) * 8 + grp;                   // this lane group's hypothesis
    int off, cnt;
    rs_range(a, n, off, cnt);
    if (cnt < a.min_points) return;                                    // geo_module.py:46 (uniform per block)
    const float* k0 = a.kp0 + 2 * (size_t)off;
    const float* k1 = a.kp1 + 2 * (size_t)off;
    // ---- the 4 correspondences of hypothesis t (every lane of the group draws the same four) and row r of its 8x8 system
    int idx[4], ok = t < a.iters;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int found = 0;
        for (uint32_t attempt = 0; attempt < 16 && !found && ok; ++attempt) {
            const int c = (int)(draw(a.seed, (uint32_t)n, (uint32_t)t, (uint32_t)k, attempt) % (uint32_t)cnt);
            int dup = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) dup |= (j < k && idx[j] == c);
            if (!dup) { idx[k] = c; found = 1; }
        }
        if (!found) { ok = 0; idx[k] = 0; }
    }
    const int pt = r >> 1;
    const int ip = pt == 0 ? idx[0] : pt == 1 ? idx[1] : pt == 2 ? idx[2] : idx[3];
    const double x = (double)k0[2 * ip], y = (double)k0[2 * ip + 1], u = (double)k1[2 * ip], v = (double)k1[2 * ip + 1];
    double row[8], rhs, h8[8];
    if ((r & 1) == 0) {
        row[0] = x; row[1] = y; row[2] = 1; row[3] = 0; row[4] = 0; row[5] = 0; row[6] = -(u * x); row[7] = -(u * y);
        rhs = u;
    } else {
        row[0] = 0; row[1] = 0; row[2] = 0; row[3] = x; row[4] = y; row[5] = 1; row[6] = -(v * x); row[7] = -(v * y);
        rhs = v;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) h8[k] = 0.0;
    ok = rs_solve8(row, rhs, h8) && ok;
    // ---- the wave's 8 hypotheses to every lane, one pass over the matches
    double H[8][8];
    int okg[8], c[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
        for (int k = 0; k < 8; ++k) H[g][k] = rs_shfl(h8[k], 8 * g);
        okg[g] = __shfl(ok, 8 * g, 64);
        c[g] = 0;
    }
    for (int i = lane; i < cnt; i += 64) {
        const double px = (double)k0[2 * i], py = (double)k0[2 * i + 1], qx = (double)k1[2 * i], qy = (double)k1[2 * i + 1];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const double hh[9] = {H[g][0], H[g][1], H[g][2], H[g][3], H[g][4], H[g][5], H[g][6], H[g][7], 1.0};
            c[g] += rs_inlier(hh, px, py, qx, qy, a.thr2);
        }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c[g] += __shfl_xor(c[g], d, 64);
    // lane 8g + k writes element k of hypothesis g (k = 0..7), lane 8g also the count and h33 = 1
    const int tg = (blockIdx.x * 4 + wave) * 8 + grp;
    if (tg < a.iters) {
        int cg = 0, og = 0;
        double hv = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            cg = grp == g ? c[g] : cg;
            og = grp == g ? okg[g] : og;
#pragma unroll
            for (int k = 0; k < 8; ++k) hv = (grp == g && r == k) ? H[g][k] : hv;
        }
        double* hp = a.hyp + ((size_t)n * a.iters + tg) * 9;
        if (og) hp[r] = hv;
        if (r == 0) {
            if (og) hp[8] = 1.0;
            a.hyp_cnt[(size_t)n * a.iters + tg] = og ? cg : -1;
        }
    }
}

// one DPP exchange of a double (both halves; v_mov_b32 dpp has ALU latency - the ds_bpermute butterfly this replaces spent ~100
// cycles per step and value: 44 values x 6 steps were most of a Levenberg-Marquardt iteration's 20 us)
template <int CTRL>
__device__ __forceinline__ double rs_dpp(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double rs_readlane(double x, int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the 64 lanes, returned in every lane: quad (xor 1, xor 2), half row (mirror within 8), row (mirror within 16) by DPP -
// a + b = b + a bit for bit, so both partners of an exchange hold the same value - then the four row sums through scalar registers
__device__ __forceinline__ double rs_wave_sum(double x) {
    x = x + rs_dpp<0xB1>(x);                          // quad_perm [1,0,3,2]
    x = x + rs_dpp<0x4E>(x);                          // quad_perm [2,3,0,1]
    x = x + rs_dpp<0x141>(x);                         // row_half_mirror
    x = x + rs_dpp<0x140>(x);                         // row_mirror
    return ((rs_readlane(x, 0) + rs_readlane(x, 16)) + rs_readlane(x, 32)) + rs_readlane(x, 48);
}
// sum over the 256 threads with TWO barriers for all NV values: lanes by DPP, the four waves through LDS (added in wave
// order).  `part` = [4][NV] doubles of LDS; the result is returned in every thread.
template <int NV>
__device__ void block_sum_fast(double (&v)[NV], double* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = rs_wave_sum(v[k]);
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
    if (t < 64) {                                     // wave 0: lanes 0..7 hold the rows of the symmetric system (all 64 lanes run the shuffles)
        const int r = t & 7;
        double rowa[8], rb, sol[8], g[9];               // (this g: the refit's result, handed on through LDS)
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) {
            double v = 0.0;                              // ata[r][j2] = ne[tri(min, max)]
            int q = 0;
#pragma unroll
            for (int i3 = 0; i3 < 8; ++i3)
#pragma unroll
                for (int j3 = i3; j3 < 8; ++j3) {
                    const int lo_ = r < j2 ? r : j2, hi_ = r < j2 ? j2 : r;
                    v = (i3 == lo_ && j3 == hi_) ? ne[q] : v;
                    ++q;
                }
            rowa[j2] = v;
        }
        rb = 0.0;
#pragma unroll
        for (int i3 = 0; i3 < 8; ++i3) rb = r == i3 ? ne[36 + i3] : rb;
#pragma unroll
        for (int k = 0; k < 8; ++k) sol[k] = 0.0;
        bool refit = rs_solve8(rowa, rb, sol) != 0;
        if (refit) {
            double hn[9], am[9];
            for (int k = 0; k < 8; ++k) hn[k] = sol[k];
            hn[8] = 1.0;
            for (int rr = 0; rr < 3; ++rr) {
                am[3 * rr + 0] = hn[3 * rr + 0] * s0;
                am[3 * rr + 1] = hn[3 * rr + 1] * s0;
                am[3 * rr + 2] = hn[3 * rr + 2] - s0 * (hn[3 * rr + 0] * c0x + hn[3 * rr + 1] * c0y);
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
        if (t == 0)
            for (int k = 0; k < 9; ++k) sh_lm[k] = g[k];
    }
    __syncthreads();
    // ---- Levenberg-Marquardt on the inliers' forward transfer error, 8 free entries (h33 = 1): what OpenCV's findHomography
    // appends to its RANSAC (oracle/ransac_oracle.c:lm_refine states the same steps); the mask is not touched
    double g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = sh_lm[k];
    if (a.lm_iters > 0) {
        // error and normal equations at hh in ONE pass and ONE block reduction: v[0..35] = upper triangle of J^T J, v[36..43] = J^T r,
        // v[44] = the squared error.  (The oracle recomputes J^T J at the top of every iteration; at an unchanged estimate that is the
        // same matrix, so it is kept here when a step is rejected and taken from the candidate's evaluation when it is accepted.)
        auto evaluate = [&](const double* hh, double (&v)[45]) {
#pragma unroll
            for (int k = 0; k < 45; ++k) v[k] = 0.0;
            for (int i = t; i < cnt; i += 256)
                if (keep[i]) {
                    const double x = (double)k0[2 * i], y = (double)k0[2 * i + 1];
                    const double iw = 1.0 / (hh[6] * x + hh[7] * y + 1.0);
                    const double up = (hh[0] * x + hh[1] * y + hh[2]) * iw, vp = (hh[3] * x + hh[4] * y + hh[5]) * iw;
                    const double ju[8] = {x * iw, y * iw, iw, 0, 0, 0, -(x * up) * iw, -(y * up) * iw};
                    const double jv[8] = {0, 0, 0, x * iw, y * iw, iw, -(x * vp) * iw, -(y * vp) * iw};
                    const double ru = up - (double)k1[2 * i], rv = vp - (double)k1[2 * i + 1];
                    int q = 0;
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) {
#pragma unroll
                        for (int j2 = i2; j2 < 8; ++j2) v[q++] += ju[i2] * ju[j2] + jv[i2] * jv[j2];
                    }
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) v[36 + i2] += ju[i2] * ru + jv[i2] * rv;
                    v[44] += ru * ru + rv * rv;
                }
            block_sum_fast(v, sh);
        };
        double ne2[45], nen[45];
        evaluate(g, ne2);
        double lambda = 1e-3, err = ne2[44];
        // residuals at rounding level already (RMS below 1e-9 px): nothing to refine (exact correspondences - the steps would
        // only chase the last bits of the error sum); stated in oracle/ransac_oracle.c:lm_refine too
        const int iters = err <= 1e-18 * m ? 0 : a.lm_iters;
        for (int it = 0; it < iters; ++it) {
            if (t < 64) {                                // wave 0, rows on lanes 0..7 (see the refit)
                const int r = t & 7;
                double rowa[8], rb, d[8];
#pragma unroll
                for (int j2 = 0; j2 < 8; ++j2) {
                    double v = 0.0;
                    int q = 0;
#pragma unroll
                    for (int i3 = 0; i3 < 8; ++i3)
#pragma unroll
                        for (int j3 = i3; j3 < 8; ++j3) {
                            const int lo_ = r < j2 ? r : j2, hi_ = r < j2 ? j2 : r;
                            v = (i3 == lo_ && j3 == hi_) ? ne2[q] : v;
                            ++q;
                        }
                    rowa[j2] = r == j2 ? v + lambda * v : v;
                }
                rb = 0.0;
#pragma unroll
                for (int i3 = 0; i3 < 8; ++i3) rb = r == i3 ? -ne2[36 + i3] : rb;
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = 0.0;
                const int ok = rs_solve8(rowa, rb, d);
                if (t == 0) {
                    sh_lm_ok = ok;
                    if (ok) {
                        for (int k = 0; k < 8; ++k) sh_lm[k] = g[k] + d[k];
                        sh_lm[8] = 1.0;
                    }
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
            evaluate(hn, nen);                           // (its barriers also fence sh_lm against the next iteration's writer)
            const double en = nen[44];
            if (en < err) {                              // (every thread holds the same en / err: the exits are uniform)
                const double gain = err - en;
#pragma unroll
                for (int k = 0; k < 9; ++k) g[k] = hn[k];
#pragma unroll
                for (int k = 0; k < 45; ++k) ne2[k] = nen[k];
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
    ransac_score<<<dim3((iters + 31) / 32, N), 256, 0, st>>>(a);
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
