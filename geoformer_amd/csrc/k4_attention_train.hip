// K4 (training): FullAttention forward WITH its softmax statistics, and the backward, for the training step's GeoTransformer 'self'
// layers - every cell of an image attends to the projected rows of its inlier cells, no masks
// (geo_transformer/geo_attention.py:72-101 as transformer.py:111-124 calls it; heads of 64 channels, 1/sqrt(64) scale).
// The inference kernel (k4_attention.hip: attn_self_head) keeps no statistics, folds the scale into the 16-bit q operand and gathers
// its rows by token; the training step needs the exact-scale logits, log-sum-exp per query and three gradient products, so these are
// kernels of their own - flash form, nothing of size L x S is ever stored:
//   forward   per 128 queries x head: S^T = K Q^T per 32-key tile, online softmax, O^T += V^T P^T; writes O and lse2 = m + log2(l)
//   delta     per (query, head): sum_d dO . O
//   dq        per 128 queries x head: P^T = exp2(S^T c - lse2), dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T
//   dk, dv    per 32 keys x head: the four waves of a workgroup own the SAME keys and every fourth 32-query tile (wave-private LDS
//             tiles, no workgroup barrier in the loop): dV^T += dO^T P, dK^T += Q^T dS; their partial sums meet in LDS at the end
//             (fixed order: the gradients are bit-reproducible).
// All products are v_mfma_f32_32x32x16 on 16-bit operands with fp32 accumulation; P and dS are rounded to the storage type for the
// second product of each pair (as the library's flash kernels do).  Orientation rule used throughout: the 32 x 32 accumulator
// (column = lane & 31, row = gf_acc_row(r, lane >> 5)) is packed as it stands into the NEXT product's 16-deep operand - registers
// 8 s .. 8 s + 7 are k-step s - so the other operand of that product is the TRANSPOSE of a row tile, read with ds_read_b64_tr_b16 from
// a second row-major image of the tile (its own swizzle; two transposing reads per fragment) in the same row order.
// forward and dq run eight waves per workgroup: waves 4..7 take the odd key tiles of the same 128 queries (two waves per SIMD, half the
// loop) and hand their partial results to waves 0..3 through LDS at the end.
#include <math.h>

#include "gf_common.h"

namespace {

constexpr int CC = 256, HD = 64, NH = CC / HD, KT = 32;
constexpr int RM = KT * 128;               // an image: [32 rows][128 B]; two swizzles:
                                           //   operand image (gf_lds_off): 16-B chunk c at c ^ ((row >> 1) & 7) - ds_read_b128 fragments of rows
                                           //   transposing image (tr_off): 32-B block b at b ^ 2 ((row >> 1) & 1) - ds_read_b64_tr_b16 fragments of channels
constexpr float LOG2E = 1.44269504088896341f;

struct TaArgs {
    const void *q, *k, *v, *o, *dout;
    long ldq, ldk, ldv, ldo, lddo;
    void *out, *dq, *dk, *dv;
    float *lse, *delta;
    int N, L, S;
    float temp;
};

template <typename T>
using Frag8 = typename Mma32<T>::Frag;

__device__ __forceinline__ float half_xor_max(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }
__device__ __forceinline__ int tr_off(int row, int chunk) { return row * 128 + (((((chunk >> 1) ^ (((row >> 1) & 1) << 1)) << 1) | (chunk & 1)) << 4); }

// 16 bytes = 8 channels of a row -> the operand image and / or the transposing image (null = an image the kernel does not read)
template <typename T>
__device__ __forceinline__ void put_row_piece(char* rm, char* tr, int row, int chunk, const Frag8<T>& x) {
    if (rm) *reinterpret_cast<Frag8<T>*>(rm + gf_lds_off(row, chunk)) = x;
    if (tr) *reinterpret_cast<Frag8<T>*>(tr + tr_off(row, chunk)) = x;
}
// fragment of the operand image: lane (row lane & 31, k half lane >> 5), channels 16 g + 8 half ..
template <typename T>
__device__ __forceinline__ Frag8<T> frag_rm(const char* img, int g, int lr, int h) {
    return *reinterpret_cast<const Frag8<T>*>(img + gf_lds_off(lr, g * 2 + h));
}
// fragment of the transposing image: lane (channel 32 b + lane & 31, half), rows 16 s + 8 (i >> 2) + 4 half + (i & 3), i = 0..7.
// A 16-lane group reads 4 rows x 16 channels (lane: row i >> 2 of the group's four, channels 4 (i & 3) ..) and gets them back transposed
// (lane: channel i, the four rows); troff(b) = the lane's byte offset for rows 4 half + (i >> 2) (k4_attention.hip: v_frag).
__device__ __forceinline__ int troff(int b, int lane) {
    const int G = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    return (4 * (G >> 1) + qq) * 128 + (((b * 2 + (G & 1)) ^ (((qq >> 1) & 1) << 1)) << 5) + pp * 8;
}
template <typename T>
__device__ __forceinline__ Frag8<T> frag_tr(const char* img, int off, int s) {
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(img + off + s * 2048));
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(img + off + s * 2048 + 1024));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(Frag8<T>, both);
}
template <typename T>
__device__ __forceinline__ Frag8<T> pack8(const float* x) {
    return Frag8<T>{(T)x[0], (T)x[1], (T)x[2], (T)x[3], (T)x[4], (T)x[5], (T)x[6], (T)x[7]};
}
// 16 bytes of row `row` (zeros behind the last row) of a [rows][ld] map, channels head * 64 + 8 chunk ..
template <typename T>
__device__ __forceinline__ Frag8<T> load_piece(const T* base, long ld, int row, int rows, int head, int chunk) {
    Frag8<T> z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
    if (row < rows) z = *reinterpret_cast<const Frag8<T>*>(base + (size_t)row * ld + head * HD + chunk * 8);
    return z;
}
// a lane's four operand fragments of its own row (query or key), zeros behind the last row
template <typename T>
__device__ __forceinline__ void load_own(Frag8<T> (&f)[4], const T* base, long ld, int row, int rows, int head, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) f[g] = load_piece<T>(base, ld, row, rows, head, g * 2 + h);
}
// the 32 channels x 32 rows accumulator pair (rows = channels 32 b + gf_acc_row(r, h), column = the lane's own row) -> [row][256] map
template <typename T>
__device__ __forceinline__ void store_own(T* base, long ld, int row, int rows, int head, int h, const v16f (&acc)[2], float scale) {
    if (row >= rows) return;
    T* p = base + (size_t)row * ld + head * HD;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<gf_vec<T, 4>*>(p + 32 * b + 8 * r4 + 4 * h) =
                gf_vec<T, 4>{(T)(acc[b][4 * r4] * scale), (T)(acc[b][4 * r4 + 1] * scale), (T)(acc[b][4 * r4 + 2] * scale),
                             (T)(acc[b][4 * r4 + 3] * scale)};
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward.  512 threads: waves 0..3 (group 0) own 32 queries each and the even key tiles, waves 4..7 (group 1) the same queries and
// the odd tiles; iteration `it` holds tile 2 it + group in image [it & 1][group] (K operand image | V transposing image), filled by the
// group's own 256 threads while the previous pair is multiplied: one workgroup barrier per pair.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512) void tattn_fwd(TaArgs a) {
    using M = Mma32<T>;
    using Frag = Frag8<T>;
    constexpr int IMG = 2 * RM;
    __shared__ __attribute__((aligned(16))) char smem[34 * 256 * 4 > 4 * IMG ? 34 * 256 * 4 : 4 * IMG];     // the images; the merge plane at the end
    const int nqb = (a.L + 127) / 128, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, h = lane >> 5, lr = lane & 31;
    int id = blockIdx.x;
    const int n = id / (nqb * NH);
    id -= n * nqb * NH;
    const int head = id / nqb, q0 = (id - head * nqb) * 128 + (wave & 3) * 32, qi = q0 + lr;
    const T* qb = (const T*)a.q + (size_t)n * a.L * a.ldq;
    const T* kb = (const T*)a.k + (size_t)n * a.S * a.ldk;
    const T* vb = (const T*)a.v + (size_t)n * a.S * a.ldv;
    Frag qf[4];
    load_own<T>(qf, qb, a.ldq, qi, a.L, head, h);
    const float c2 = a.temp * LOG2E;
    const int tro[2] = {troff(0, lane), troff(1, lane)};
    v16f o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int ntiles = (a.S + KT - 1) / KT, niter = (ntiles + 1) / 2, frow = (tid & 255) >> 3, fchunk = tid & 7;
    Frag kr = load_piece<T>(kb, a.ldk, grp * KT + frow, a.S, head, fchunk), vr = load_piece<T>(vb, a.ldv, grp * KT + frow, a.S, head, fchunk);
    put_row_piece<T>(smem + grp * IMG, nullptr, frow, fchunk, kr);
    put_row_piece<T>(nullptr, smem + grp * IMG + RM, frow, fchunk, vr);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int t = 2 * it + grp;
        const char* kimg = smem + ((it & 1) * 2 + grp) * IMG;
        const char* vimg = kimg + RM;
        if (it + 1 < niter) {
            kr = load_piece<T>(kb, a.ldk, (t + 2) * KT + frow, a.S, head, fchunk);
            vr = load_piece<T>(vb, a.ldv, (t + 2) * KT + frow, a.S, head, fchunk);
        }
        if (t < ntiles) {                                  // (group 1 has no tile in the last iteration of an odd count)
            v16f sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) M::mma(frag_rm<T>(kimg, g, lr, h), qf[g], sc);        // rows = keys, column = the lane's query
            float x[16], tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = t * KT + gf_acc_row(r, h) < a.S ? sc[r] * c2 : -INFINITY;
                tmax = fmaxf(tmax, x[r]);
            }
            tmax = half_xor_max(tmax);                     // a tile holds at least one key: finite
            const float mn = fmaxf(m, tmax), alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
                x[r] = __builtin_amdgcn_exp2f(x[r] - mn);
                l += x[r];
            }
            const Frag p0 = pack8<T>(x), p1 = pack8<T>(x + 8);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                M::mma(frag_tr<T>(vimg, tro[b], 0), p0, o[b]);                               // rows = channels, column = the lane's query
                M::mma(frag_tr<T>(vimg, tro[b], 1), p1, o[b]);
            }
        }
        if (it + 1 < niter) {
            char* nimg = smem + (((it + 1) & 1) * 2 + grp) * IMG;
            put_row_piece<T>(nimg, nullptr, frow, fchunk, kr);
            put_row_piece<T>(nullptr, nimg + RM, frow, fchunk, vr);
        }
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    // group 1 -> group 0: (m, l, o) of the odd tiles; plane [34][256 lanes]
    float* mg = reinterpret_cast<float*>(smem);
    const int ml = tid & 255;
    if (grp == 1) {
        mg[ml] = m;
        mg[256 + ml] = l;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[(2 + b * 16 + r) * 256 + ml] = o[b][r];
    }
    __syncthreads();
    if (grp == 1) return;
    if (a.S > 0) {
        const float m1 = mg[ml], l1 = mg[256 + ml], mn = fmaxf(m, m1);
        const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);       // m1 = -inf without an odd tile: a1 = 0
        l = l * a0 + l1 * a1;
        m = mn;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][r] = o[b][r] * a0 + mg[(2 + b * 16 + r) * 256 + ml] * a1;
    }
    const float inv = a.S > 0 ? 1.0f / l : 0.f;
    store_own<T>((T*)a.out + (size_t)n * a.L * a.ldo, a.ldo, qi, a.L, head, h, o, inv);
    if (h == 0 && qi < a.L) a.lse[((size_t)n * NH + head) * a.L + qi] = a.S > 0 ? m + __builtin_amdgcn_logf(l) : 0.f;   // v_log_f32 = log2
}

// delta[n][head][query] = sum over the head's channels of dO . O
template <typename T>
__global__ __launch_bounds__(256) void tattn_delta(TaArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)a.N * NH * a.L;
    if (i >= total) return;
    const int qi = (int)(i % a.L), head = (int)((i / a.L) % NH), n = (int)(i / ((size_t)a.L * NH));
    const T* po = (const T*)a.o + ((size_t)n * a.L + qi) * a.ldo + head * HD;
    const T* pd = (const T*)a.dout + ((size_t)n * a.L + qi) * a.lddo + head * HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const Frag8<T> x = *reinterpret_cast<const Frag8<T>*>(po + 8 * c), y = *reinterpret_cast<const Frag8<T>*>(pd + 8 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += gf_to_float(x[j]) * gf_to_float(y[j]);
    }
    a.delta[i] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// dq: the forward's structure (two wave groups over the even / odd key tiles; image = K operand | K transposing | V operand)
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512) void tattn_bwd_dq(TaArgs a) {
    using M = Mma32<T>;
    using Frag = Frag8<T>;
    constexpr int IMG = 3 * RM;
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];                                  // 48 KiB; the merge plane (32 KiB) at the end
    const int nqb = (a.L + 127) / 128, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, h = lane >> 5, lr = lane & 31;
    int id = blockIdx.x;
    const int n = id / (nqb * NH);
    id -= n * nqb * NH;
    const int head = id / nqb, q0 = (id - head * nqb) * 128 + (wave & 3) * 32, qi = q0 + lr;
    const T* kb = (const T*)a.k + (size_t)n * a.S * a.ldk;
    const T* vb = (const T*)a.v + (size_t)n * a.S * a.ldv;
    Frag qf[4], df[4];
    load_own<T>(qf, (const T*)a.q + (size_t)n * a.L * a.ldq, a.ldq, qi, a.L, head, h);
    load_own<T>(df, (const T*)a.dout + (size_t)n * a.L * a.lddo, a.lddo, qi, a.L, head, h);
    const size_t si = ((size_t)n * NH + head) * a.L + min(qi, a.L - 1);
    const float lse = a.lse[si], delta = a.delta[si], c2 = a.temp * LOG2E;
    const int tro[2] = {troff(0, lane), troff(1, lane)};
    v16f dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[0][r] = dq[1][r] = 0.f;
    const int ntiles = (a.S + KT - 1) / KT, niter = (ntiles + 1) / 2, frow = (tid & 255) >> 3, fchunk = tid & 7;
    Frag kr = load_piece<T>(kb, a.ldk, grp * KT + frow, a.S, head, fchunk), vr = load_piece<T>(vb, a.ldv, grp * KT + frow, a.S, head, fchunk);
    put_row_piece<T>(smem + grp * IMG, smem + grp * IMG + RM, frow, fchunk, kr);
    put_row_piece<T>(smem + grp * IMG + 2 * RM, nullptr, frow, fchunk, vr);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int t = 2 * it + grp;
        const char* kimg = smem + ((it & 1) * 2 + grp) * IMG;
        const char* ktr = kimg + RM;
        const char* vimg = ktr + RM;
        if (it + 1 < niter) {
            kr = load_piece<T>(kb, a.ldk, (t + 2) * KT + frow, a.S, head, fchunk);
            vr = load_piece<T>(vb, a.ldv, (t + 2) * KT + frow, a.S, head, fchunk);
        }
        if (t < ntiles) {
            v16f sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = dp[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) M::mma(frag_rm<T>(kimg, g, lr, h), qf[g], sc);        // S^T: rows = keys
#pragma unroll
            for (int g = 0; g < 4; ++g) M::mma(frag_rm<T>(vimg, g, lr, h), df[g], dp);        // dP^T = V dO^T
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = t * KT + gf_acc_row(r, h) < a.S ? __builtin_amdgcn_exp2f(sc[r] * c2 - lse) : 0.f;
                ds[r] = p * (dp[r] - delta);
            }
            const Frag d0 = pack8<T>(ds), d1 = pack8<T>(ds + 8);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                M::mma(frag_tr<T>(ktr, tro[b], 0), d0, dq[b]);                               // dQ^T += K^T dS^T
                M::mma(frag_tr<T>(ktr, tro[b], 1), d1, dq[b]);
            }
        }
        if (it + 1 < niter) {
            char* nimg = smem + (((it + 1) & 1) * 2 + grp) * IMG;
            put_row_piece<T>(nimg, nimg + RM, frow, fchunk, kr);
            put_row_piece<T>(nimg + 2 * RM, nullptr, frow, fchunk, vr);
        }
        __syncthreads();
    }
    float* mg = reinterpret_cast<float*>(smem);
    const int ml = tid & 255;
    if (grp == 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[(b * 16 + r) * 256 + ml] = dq[b][r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[b][r] += mg[(b * 16 + r) * 256 + ml];
    store_own<T>((T*)a.dq + (size_t)n * a.L * CC, CC, qi, a.L, head, h, dq, a.temp);
}

// ---------------------------------------------------------------------------------------------------------------------------
// dk, dv
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int DKV_WAVE = 4 * RM + 256;                   // Q operand | Q transposing | dO operand | dO transposing | lse2[32] | delta[32]
constexpr int DKV_LDS = 4 * DKV_WAVE;                    // 66,560 B (the four 16 KiB partial-sum planes of the epilogue fit inside)

template <typename T>
__global__ __launch_bounds__(256) void tattn_bwd_dkv(TaArgs a) {
    using M = Mma32<T>;
    using Frag = Frag8<T>;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int nkb = (a.S + KT - 1) / KT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    int id = blockIdx.x;
    const int n = id / (nkb * NH);
    id -= n * nkb * NH;
    const int head = id / nkb, k0 = (id - head * nkb) * KT, ki = k0 + lr;
    const T* qb = (const T*)a.q + (size_t)n * a.L * a.ldq;
    const T* db = (const T*)a.dout + (size_t)n * a.L * a.lddo;
    const float* lb = a.lse + ((size_t)n * NH + head) * a.L;
    const float* eb = a.delta + ((size_t)n * NH + head) * a.L;
    Frag kf[4], vf[4];
    load_own<T>(kf, (const T*)a.k + (size_t)n * a.S * a.ldk, a.ldk, ki, a.S, head, h);
    load_own<T>(vf, (const T*)a.v + (size_t)n * a.S * a.ldv, a.ldv, ki, a.S, head, h);
    const float c2 = a.temp * LOG2E;
    const int tro[2] = {troff(0, lane), troff(1, lane)};
    v16f dv[2], dk[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[0][r] = dv[1][r] = dk[0][r] = dk[1][r] = 0.f;
    char* qimg = dsm + wave * DKV_WAVE;
    char* qtr = qimg + RM;
    char* dimg = qtr + RM;
    char* dtr = dimg + RM;
    float* ls = reinterpret_cast<float*>(dtr + RM);
    float* dl = ls + 32;
    const int ntiles = (a.L + KT - 1) / KT, frow = lane >> 3, fchunk = lane & 7;
    Frag qr[4], dr[4];
    float lsr = 0.f, dlr = 0.f;
    auto fetch = [&](int t) {                            // the wave's own 32-query tile: four 8-row passes of q and dout, the statistics on lanes 0..31
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            qr[p] = load_piece<T>(qb, a.ldq, t * KT + 8 * p + frow, a.L, head, fchunk);
            dr[p] = load_piece<T>(db, a.lddo, t * KT + 8 * p + frow, a.L, head, fchunk);
        }
        const int qi = t * KT + lr;
        lsr = qi < a.L ? lb[qi] : INFINITY;              // a row behind the sequence: P = exp2(-inf) = 0
        dlr = qi < a.L ? eb[qi] : 0.f;
    };
    if (wave < ntiles) fetch(wave);
    for (int t = wave; t < ntiles; t += 4) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            put_row_piece<T>(qimg, qtr, 8 * p + frow, fchunk, qr[p]);
            put_row_piece<T>(dimg, dtr, 8 * p + frow, fchunk, dr[p]);
        }
        if (h == 0) {
            ls[lr] = lsr;
            dl[lr] = dlr;
        }
        __builtin_amdgcn_wave_barrier();                 // LDS operations of one wave complete in order: no counter wait needed, only the order
        if (t + 4 < ntiles) fetch(t + 4);
        v16f sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = dp[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) M::mma(frag_rm<T>(qimg, g, lr, h), kf[g], sc);            // S: rows = queries, column = the lane's key
#pragma unroll
        for (int g = 0; g < 4; ++g) M::mma(frag_rm<T>(dimg, g, lr, h), vf[g], dp);            // dP = dO V^T
        float p[16], ds[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v4f l4 = *reinterpret_cast<const v4f*>(ls + 8 * j + 4 * h), d4 = *reinterpret_cast<const v4f*>(dl + 8 * j + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * j + i;
                p[r] = ki < a.S ? __builtin_amdgcn_exp2f(sc[r] * c2 - l4[i]) : 0.f;
                ds[r] = p[r] * (dp[r] - d4[i]);
            }
        }
        const Frag p0 = pack8<T>(p), p1 = pack8<T>(p + 8), d0 = pack8<T>(ds), d1 = pack8<T>(ds + 8);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            M::mma(frag_tr<T>(dtr, tro[b], 0), p0, dv[b]);                                   // dV^T += dO^T P
            M::mma(frag_tr<T>(dtr, tro[b], 1), p1, dv[b]);
            M::mma(frag_tr<T>(qtr, tro[b], 0), d0, dk[b]);                                   // dK^T += Q^T dS
            M::mma(frag_tr<T>(qtr, tro[b], 1), d1, dk[b]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the four waves' partial sums: plane w = [16 registers][64 lanes] fp32 of accumulator w (dv[0], dv[1], dk[0], dk[1]) from every wave,
    // wave w adds the four copies of accumulator w in wave order and stores it
    __syncthreads();
    float* red = reinterpret_cast<float*>(dsm);
#pragma unroll
    for (int acc = 0; acc < 4; ++acc) {
        const v16f& x = acc == 0 ? dv[0] : acc == 1 ? dv[1] : acc == 2 ? dk[0] : dk[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((acc * 4 + wave) * 16 + r) * 64 + lane] = x[r];
    }
    __syncthreads();
    float s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s[r] = red[((wave * 4 + 0) * 16 + r) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) s[r] += red[((wave * 4 + w) * 16 + r) * 64 + lane];
    }
    if (ki < a.S) {
        const bool isk = wave >= 2;
        const float scale = isk ? a.temp : 1.f;
        T* op = (T*)(isk ? a.dk : a.dv) + ((size_t)n * a.S + ki) * CC + head * HD + 32 * (wave & 1);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<gf_vec<T, 4>*>(op + 8 * r4 + 4 * h) =
                gf_vec<T, 4>{(T)(s[4 * r4] * scale), (T)(s[4 * r4 + 1] * scale), (T)(s[4 * r4 + 2] * scale), (T)(s[4 * r4 + 3] * scale)};
    }
}

bool rows_ok(const void* p, long ld) { return ((uintptr_t)p & 15) == 0 && (ld & 7) == 0 && ld >= CC; }

template <typename T>
int launch_fwd(const TaArgs& a, hipStream_t st) {
    const int blocks = (a.L + 127) / 128 * NH * a.N;
    tattn_fwd<T><<<blocks, 512, 0, st>>>(a);
    return 0;
}
template <typename T>
int launch_bwd(const TaArgs& a, hipStream_t st) {
    static std::atomic<uint64_t> done{0};
    if (gf_first_use_on_device(done)) (void)hipFuncSetAttribute((const void*)tattn_bwd_dkv<T>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS);
    const size_t rows = (size_t)a.N * NH * a.L;
    tattn_delta<T><<<(unsigned)((rows + 255) / 256), 256, 0, st>>>(a);
    tattn_bwd_dq<T><<<(a.L + 127) / 128 * NH * a.N, 512, 0, st>>>(a);
    tattn_bwd_dkv<T><<<(a.S + KT - 1) / KT * NH * a.N, 256, DKV_LDS, st>>>(a);
    return 0;
}

}  // namespace

extern "C" int gf_full_attention_train_forward(const void* q, const void* k, const void* v, int dtype, int N, int L, int S, int H, int D, long ldq,
                                               long ldk, long ldv, float softmax_temp, void* out, long ldo, float* lse, void* stream) {
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "16-bit tensors only (the fp32 step keeps the explicit form)");
    GF_CHECK_ARG(H == NH && D == HD, "built for 4 heads of 64 channels (GeoTransformer)");
    GF_CHECK_ARG(N >= 0 && L >= 0 && S >= 0, "negative size");
    if (N == 0 || L == 0) return GF_OK;
    GF_CHECK_ARG(q && out && lse && (S == 0 || (k && v)), "null pointer");
    GF_CHECK_ARG(rows_ok(q, ldq) && rows_ok(out, ldo) && (S == 0 || (rows_ok(k, ldk) && rows_ok(v, ldv))),
                 "rows must be 16-byte aligned with strides that are multiples of 8 elements");
    TaArgs a{};
    a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.out = out; a.ldo = ldo; a.lse = lse;
    a.N = N; a.L = L; a.S = S; a.temp = softmax_temp;
    hipStream_t st = (hipStream_t)stream;
    void* tok = gf_prof_begin("k4_train_forward", st, 4.0 * N * L * (double)S * CC);
    if (dtype == GF_F16) launch_fwd<_Float16>(a, st);
    else launch_fwd<gf_bf16>(a, st);
    gf_prof_end("k4_train_forward", tok, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" size_t gf_full_attention_backward_workspace_bytes(int N, int L, int H) { return gf_align_up((size_t)N * H * L * sizeof(float), 256); }

extern "C" int gf_full_attention_backward(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                                          int dtype, int N, int L, int S, int H, int D, long ldq, long ldk, long ldv, long ldo, long lddo,
                                          float softmax_temp, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "16-bit tensors only");
    GF_CHECK_ARG(H == NH && D == HD, "built for 4 heads of 64 channels (GeoTransformer)");
    GF_CHECK_ARG(N >= 0 && L >= 0 && S >= 0, "negative size");
    if (N == 0) return GF_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = 2;
    if (L == 0 || S == 0) {                                // no keys: the forward wrote zeros, every gradient is zero
        if (L > 0) { GF_CHECK_ARG(dq, "null pointer"); (void)hipMemsetAsync(dq, 0, (size_t)N * L * CC * esz, st); }
        if (S > 0) { GF_CHECK_ARG(dk && dv, "null pointer"); (void)hipMemsetAsync(dk, 0, (size_t)N * S * CC * esz, st); (void)hipMemsetAsync(dv, 0, (size_t)N * S * CC * esz, st); }
        return GF_OK;
    }
    GF_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && workspace, "null pointer");
    GF_CHECK_ARG(workspace_bytes >= gf_full_attention_backward_workspace_bytes(N, L, H), "workspace too small");
    GF_CHECK_ARG(rows_ok(q, ldq) && rows_ok(k, ldk) && rows_ok(v, ldv) && rows_ok(out, ldo) && rows_ok(dout, lddo) && rows_ok(dq, CC) &&
                     rows_ok(dk, CC) && rows_ok(dv, CC),
                 "rows must be 16-byte aligned with strides that are multiples of 8 elements");
    TaArgs a{};
    a.q = q; a.k = k; a.v = v; a.o = out; a.dout = dout; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lse = const_cast<float*>(lse); a.delta = (float*)workspace;
    a.N = N; a.L = L; a.S = S; a.temp = softmax_temp;
    void* tok = gf_prof_begin("k4_train_backward", st, 14.0 * N * L * (double)S * CC);
    if (dtype == GF_F16) launch_bwd<_Float16>(a, st);
    else launch_bwd<gf_bf16>(a, st);
    gf_prof_end("k4_train_backward", tok, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
