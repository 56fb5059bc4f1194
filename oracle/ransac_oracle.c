/* CPU statement of the build's own homography RANSAC (TEST INFRASTRUCTURE - checker only).
 *
 * PARITY UNPINNED with respect to the reference: GeoModule.apply_RANSAC calls OpenCV
 * `cv2.findHomography(kp0, kp1, cv2.RANSAC, 8.0)` (model/geo_module.py:47-48,
 * opencv_python==4.6.0.66 in requirements.txt:4).  OpenCV is neither in the reference tree nor in
 * this image, and the reference has no test that fixes its output.  What IS pinned: the contract
 * `(kp0 int[n,2], kp1 int[n,2]) -> (M float64[3,3] | none, mask uint8[n])` with the >8-matches gate
 * (:46) and the 8 px reprojection threshold, and everything downstream of (M, mask).
 *
 * This file states, in plain C and fp64, the algorithm geoformer_amd/csrc/k_ransac.hip runs on the
 * GPU, operation for operation (build both with fp contraction off), so that tests can require the
 * SAME inlier mask bit for bit and the same M to 1e-9:
 *
 *   T hypotheses; hypothesis t draws 4 distinct correspondences with a counter-based integer hash
 *   (no state, any order of evaluation); 4-point homography by Gaussian elimination with partial
 *   pivoting on the 8x8 system (h33 = 1); score = #points with squared forward transfer error
 *   <= thr^2; best = most inliers, ties -> smallest t; mask = inliers of the best hypothesis;
 *   M = least-squares (normal equations, Hartley-normalised, h33 = 1) refit on those inliers,
 *   falling back to the best hypothesis when the refit is singular; then `lm_iters` Levenberg-Marquardt
 *   steps on the inliers' forward transfer error in the 8 free entries (h33 = 1) - the step OpenCV's
 *   findHomography appends to its RANSAC (LMSolver, 10 iterations; its exact damping schedule is not
*   restated: lambda starts at 1e-3, /10 on an accepted step, x10 on a rejected one, damping on the
 *   diagonal of J^T J; stops early once an accepted step gains less than 1e-10 of the error or lambda
 *   passes 1e6).  The mask is the best hypothesis' and is not touched by either refit.
 *   Fewer than 4 inliers -> none.
 *
 *   gcc -O2 -ffp-contract=off -shared -fPIC -o _build/libransac_oracle.so ransac_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

static uint32_t draw(uint32_t seed, uint32_t sample, uint32_t t, uint32_t k, uint32_t attempt) {
    uint32_t x = seed * 0x9E3779B1u;
    x = mix32(x ^ (sample + 0x7F4A7C15u));
    x = mix32(x ^ (t * 0x85EBCA6Bu + 0x165667B1u));
    x = mix32(x ^ (k * 0xC2B2AE35u + 0x27D4EB2Fu));
    x = mix32(x ^ (attempt * 0x9E3779B1u + 0x61C88647u));
    return x;
}

/* solve the n x n system a x = b in place (row-major a[n][n]); returns 0 when singular */
static int solve(double* a, double* b, int n) {
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

/* 4-point homography (h33 = 1): rows [x y 1 0 0 0 -ux -uy | u], [0 0 0 x y 1 -vx -vy | v] */
static int four_point(const double* p0, const double* p1, const int* idx, double* h) {
    double a[64], b[8];
    for (int k = 0; k < 4; ++k) {
        const double x = p0[2 * idx[k]], y = p0[2 * idx[k] + 1], u = p1[2 * idx[k]], v = p1[2 * idx[k] + 1];
        double* r0 = a + (2 * k) * 8;
        double* r1 = a + (2 * k + 1) * 8;
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -(u * x); r0[7] = -(u * y);
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -(v * x); r1[7] = -(v * y);
        b[2 * k] = u; b[2 * k + 1] = v;
    }
    if (!solve(a, b, 8)) return 0;
    for (int k = 0; k < 8; ++k) h[k] = b[k];
    h[8] = 1.0;
    return 1;
}

static int is_inlier(const double* h, double x, double y, double u, double v, double thr2) {
    const double w = h[6] * x + h[7] * y + h[8];
    if (w == 0.0) return 0;
    const double px = (h[0] * x + h[1] * y + h[2]) / w;
    const double py = (h[3] * x + h[4] * y + h[5]) / w;
    const double dx = px - u, dy = py - v;
    return dx * dx + dy * dy <= thr2;
}

/* sum over the inliers of the squared forward transfer error of h (h[8] = 1) */
static double lm_error(const double* h, const double* p0, const double* p1, const uint8_t* mask, int n) {
    double e = 0;
    for (int i = 0; i < n; ++i) if (mask[i]) {
        const double x = p0[2 * i], y = p0[2 * i + 1];
        const double w = h[6] * x + h[7] * y + 1.0, iw = 1.0 / w;
        const double du = (h[0] * x + h[1] * y + h[2]) * iw - p1[2 * i], dv = (h[3] * x + h[4] * y + h[5]) * iw - p1[2 * i + 1];
        e += du * du + dv * dv;
    }
    return e;
}

/* Levenberg-Marquardt on h[0..7] (h[8] = 1): residuals (u' - u, v' - v) per inlier, u' = (h0 x + h1 y + h2) / w,
 * v' = (h3 x + h4 y + h5) / w, w = h6 x + h7 y + 1;  d u' / d h = [x, y, 1, 0, 0, 0, -x u', -y u'] / w, likewise v'. */
static void lm_refine(double* h, const double* p0, const double* p1, const uint8_t* mask, int n, int lm_iters) {
    double lambda = 1e-3, err = lm_error(h, p0, p1, mask, n);
    int m = 0;
    for (int i = 0; i < n; ++i) m += mask[i];
    /* residuals at rounding level already (RMS below 1e-9 px: exact correspondences): nothing to refine */
    if (err <= 1e-18 * (double)m) return;
    for (int it = 0; it < lm_iters; ++it) {
        double jtj[64], jtr[8];
        memset(jtj, 0, sizeof(jtj)); memset(jtr, 0, sizeof(jtr));
        for (int i = 0; i < n; ++i) if (mask[i]) {
            const double x = p0[2 * i], y = p0[2 * i + 1];
            const double w = h[6] * x + h[7] * y + 1.0, iw = 1.0 / w;
            const double up = (h[0] * x + h[1] * y + h[2]) * iw, vp = (h[3] * x + h[4] * y + h[5]) * iw;
            const double ju[8] = {x * iw, y * iw, iw, 0, 0, 0, -(x * up) * iw, -(y * up) * iw};
            const double jv[8] = {0, 0, 0, x * iw, y * iw, iw, -(x * vp) * iw, -(y * vp) * iw};
            const double ru = up - p1[2 * i], rv = vp - p1[2 * i + 1];
            for (int a = 0; a < 8; ++a) {
                for (int b = 0; b < 8; ++b) jtj[a * 8 + b] += ju[a] * ju[b] + jv[a] * jv[b];
                jtr[a] += ju[a] * ru + jv[a] * rv;
            }
        }
        double A[64], d[8], hn[9];
        for (int k = 0; k < 64; ++k) A[k] = jtj[k];
        for (int k = 0; k < 8; ++k) { A[k * 8 + k] = jtj[k * 8 + k] + lambda * jtj[k * 8 + k]; d[k] = -jtr[k]; }
        if (!solve(A, d, 8)) { lambda = lambda * 10.0; continue; }
        for (int k = 0; k < 8; ++k) hn[k] = h[k] + d[k];
        hn[8] = 1.0;
        const double en = lm_error(hn, p0, p1, mask, n);
        if (en < err) {
            const double gain = err - en;
            for (int k = 0; k < 8; ++k) h[k] = hn[k];
            lambda = lambda * 0.1;
            const int done = gain <= 1e-10 * err;       /* converged: the refit's minimum is usually 2-3 steps away */
            err = en;
            if (done) break;
        } else {
            lambda = lambda * 10.0;
            if (lambda > 1e6) break;                    /* no downhill step left at any damping worth trying */
        }
    }
}

/* The algorithm on fp64 points p0 = [n,2 | n,2] (p1 = p0 + 2n; freed here).  min_points: the caller's gate (geo_module.py:46 demands
 * MORE than 8 matches = 9; the evaluation harness' findHomography needs 4). */
static int ransac_core(double* p0, int n, double thr, int iters, uint32_t seed, uint32_t sample, int lm_iters, double* M,
                       uint8_t* mask) {
    double* p1 = p0 + 2 * (size_t)n;
    const double thr2 = thr * thr;
    int best_cnt = -1, best_t = -1;
    double best_h[9];
    for (int t = 0; t < iters; ++t) {
        int idx[4], ok = 1;
        for (int k = 0; k < 4 && ok; ++k) {
            int found = 0;
            for (uint32_t attempt = 0; attempt < 16 && !found; ++attempt) {
                const int c = (int)(draw(seed, sample, (uint32_t)t, (uint32_t)k, attempt) % (uint32_t)n);
                int dup = 0;
                for (int j = 0; j < k; ++j) dup |= (idx[j] == c);
                if (!dup) { idx[k] = c; found = 1; }
            }
            ok = found;
        }
        double h[9];
        if (!ok || !four_point(p0, p1, idx, h)) continue;
        int cnt = 0;
        for (int i = 0; i < n; ++i) cnt += is_inlier(h, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], thr2);
        if (cnt > best_cnt) { best_cnt = cnt; best_t = t; memcpy(best_h, h, sizeof(h)); }
    }
    if (best_t < 0 || best_cnt < 4) { free(p0); return 0; }
    int m = 0;
    for (int i = 0; i < n; ++i) {
        mask[i] = (uint8_t)is_inlier(best_h, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], thr2);
        m += mask[i];
    }
    /* least-squares refit on the inliers, Hartley normalisation */
    double c0x = 0, c0y = 0, c1x = 0, c1y = 0;
    for (int i = 0; i < n; ++i) if (mask[i]) { c0x += p0[2 * i]; c0y += p0[2 * i + 1]; c1x += p1[2 * i]; c1y += p1[2 * i + 1]; }
    c0x /= m; c0y /= m; c1x /= m; c1y /= m;
    double d0 = 0, d1 = 0;
    for (int i = 0; i < n; ++i) if (mask[i]) {
        d0 += sqrt((p0[2 * i] - c0x) * (p0[2 * i] - c0x) + (p0[2 * i + 1] - c0y) * (p0[2 * i + 1] - c0y));
        d1 += sqrt((p1[2 * i] - c1x) * (p1[2 * i] - c1x) + (p1[2 * i + 1] - c1y) * (p1[2 * i + 1] - c1y));
    }
    const double s0 = d0 > 0 ? sqrt(2.0) * m / d0 : 1.0, s1 = d1 > 0 ? sqrt(2.0) * m / d1 : 1.0;
    double ata[64], atb[8];
    memset(ata, 0, sizeof(ata)); memset(atb, 0, sizeof(atb));
    for (int i = 0; i < n; ++i) if (mask[i]) {
        const double x = (p0[2 * i] - c0x) * s0, y = (p0[2 * i + 1] - c0y) * s0;
        const double u = (p1[2 * i] - c1x) * s1, v = (p1[2 * i + 1] - c1y) * s1;
        const double r0[8] = {x, y, 1, 0, 0, 0, -(u * x), -(u * y)};
        const double r1[8] = {0, 0, 0, x, y, 1, -(v * x), -(v * y)};
        for (int a = 0; a < 8; ++a) {
            for (int b = 0; b < 8; ++b) ata[a * 8 + b] += r0[a] * r0[b] + r1[a] * r1[b];
            atb[a] += r0[a] * u + r1[a] * v;
        }
    }
    double hn[9];
    if (solve(ata, atb, 8)) {
        for (int k = 0; k < 8; ++k) hn[k] = atb[k];
        hn[8] = 1.0;
        /* H = T1^-1 * Hn * T0,  T = [[s,0,-s*cx],[0,s,-s*cy],[0,0,1]] */
        double a[9];   /* Hn * T0 */
        for (int r = 0; r < 3; ++r) {
            a[3 * r + 0] = hn[3 * r + 0] * s0;
            a[3 * r + 1] = hn[3 * r + 1] * s0;
            a[3 * r + 2] = hn[3 * r + 2] - s0 * (hn[3 * r + 0] * c0x + hn[3 * r + 1] * c0y);
        }
        double g[9];   /* T1^-1 * a,  T1^-1 = [[1/s1,0,c1x],[0,1/s1,c1y],[0,0,1]] */
        for (int c = 0; c < 3; ++c) {
            g[c] = a[c] / s1 + c1x * a[6 + c];
            g[3 + c] = a[3 + c] / s1 + c1y * a[6 + c];
            g[6 + c] = a[6 + c];
        }
        if (fabs(g[8]) > 1e-12) {
            for (int k = 0; k < 9; ++k) M[k] = g[k] / g[8];
            if (lm_iters > 0) lm_refine(M, p0, p1, mask, n, lm_iters);
            free(p0);
            return 1;
        }
    }
    memcpy(M, best_h, sizeof(best_h));
    if (lm_iters > 0) lm_refine(M, p0, p1, mask, n, lm_iters);
    free(p0);
    return 1;
}

/* kp0, kp1: int64 [n,2].  Returns 1 and fills M[9], mask[n] when a model is found, else 0 (mask zeroed). */
int gf_oracle_ransac(const int64_t* kp0, const int64_t* kp1, int n, double thr, int iters, uint32_t seed,
                     uint32_t sample, int lm_iters, double* M, uint8_t* mask) {
    memset(mask, 0, (size_t)n);
    if (n <= 8) return 0;                         /* geo_module.py:46 */
    double* p0 = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    double* p1 = p0 + 2 * (size_t)n;
    for (int i = 0; i < 2 * n; ++i) { p0[i] = (double)kp0[i]; p1[i] = (double)kp1[i]; }
    return ransac_core(p0, n, thr, iters, seed, sample, lm_iters, M, mask);
}

/* The same on sub-pixel keypoints (float32 [n,2], widened to fp64 as k_ransac.hip does with integer_keypoints = 0) and a caller-chosen
 * gate: the evaluation harness' homography from the final matches (hpatches_helper.py:185-239 calls cv2.findHomography(..., RANSAC, 3)),
 * geoformer_amd/matcher.py:estimate_homography on the device. */
int gf_oracle_ransac_f32(const float* kp0, const float* kp1, int n, double thr, int iters, uint32_t seed, uint32_t sample,
                         int lm_iters, int min_points, double* M, uint8_t* mask) {
    memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n < min_points || n < 4) return 0;
    double* p0 = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    double* p1 = p0 + 2 * (size_t)n;
    for (int i = 0; i < 2 * n; ++i) { p0[i] = (double)kp0[i]; p1[i] = (double)kp1[i]; }
    return ransac_core(p0, n, thr, iters, seed, sample, lm_iters, M, mask);
}
