// Fine level (CDNA4 / gfx950).
//
//   gf_fine_gather  K7 (a13): FinePreprocess.forward's F.unfold(5x5, stride 4, pad 2) + gather by
//                   (b_ids, i_ids / j_ids) (model/loftr_src/loftr/loftr_module/fine_preprocess.py:41-56),
//                   without materialising the [N, 25*C, L] unfold: each match reads only its own 5x5
//                   window; plus the gather of the two coarse feature rows that feed down_proj (:61).
//   gf_fine_match   K8 (a14/a15): FineMatching2.forward + get_fine_match (model/fine_matching2.py:21-126):
//                   25x25 dual-softmax per match, global arg-max, threshold, ordered compaction and the
//                   fine keypoint arithmetic.
#include <math.h>

#include <type_traits>

#include "gf_common.h"

namespace {

struct FgArgs {
    const void* f0;          // fine maps viewed as [N, C, H, W] with element strides
    const void* f1;
    long s0n, s0c, s0h, s0w, s1n, s1c, s1h, s1w;
    int H0, W0, H1, W1, C;   // fine map sizes
    const void* c0;          // coarse (geo) features [N, L, CC], [N, S, CC]
    const void* c1;
    int L, S, CC;
    const int64_t* b_ids;
    const int64_t* i_ids;
    const int64_t* j_ids;
    int M, w0c, w1c, stride, W;
    void* win;               // [2M][W*W][C]  (image0 windows first, then image1: torch.cat(..., 0))
    void* ccat;              // [2M][CC]
};

// one workgroup per (match, side): thread c < C copies channel c of the 25 window positions
template <typename TF, typename T>
__global__ __launch_bounds__(256) void fine_gather(FgArgs a) {
    const int m = blockIdx.x, side = blockIdx.y, t = threadIdx.x;
    const int b = (int)a.b_ids[m];
    const int cell = (int)(side ? a.j_ids[m] : a.i_ids[m]);
    const int wc = side ? a.w1c : a.w0c;
    const int Hf = side ? a.H1 : a.H0, Wf = side ? a.W1 : a.W0;
    const TF* f = (const TF*)(side ? a.f1 : a.f0);
    const long sn = side ? a.s1n : a.s0n, sc = side ? a.s1c : a.s0c, sh = side ? a.s1h : a.s0h, sw = side ? a.s1w : a.s0w;
    const int cy = (cell / wc) * a.stride - a.W / 2, cx = (cell % wc) * a.stride - a.W / 2;
    T* out = (T*)a.win + ((size_t)side * a.M + m) * a.W * a.W * a.C;
    if (t < a.C) {
        for (int k = 0; k < a.W * a.W; ++k) {
            const int y = cy + k / a.W, x = cx + k % a.W;
            float v = 0.f;                                        // zero padding of F.unfold
            if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = gf_to_float(f[b * sn + t * sc + y * sh + x * sw]);
            out[(size_t)k * a.C + t] = gf_from_float<T>(v);
        }
    }
    const T* cf = (const T*)(side ? a.c1 : a.c0) + ((size_t)b * (side ? a.S : a.L) + cell) * a.CC;
    T* co = (T*)a.ccat + ((size_t)side * a.M + m) * a.CC;
    for (int c = t; c < a.CC; c += blockDim.x) co[c] = cf[c];
}

// The 16-bit inference form (channels-last fine maps, window tensor in the same type): ONE WAVE per (match, side) moves the
// 25 x C window as 16-byte pieces - C / 8 lanes per window position, 64 / (C / 8) positions per instruction - and the coarse
// feature row behind it; out-of-image positions are zeros (F.unfold's padding).  The general kernel above moves 2 bytes per
// lane and instruction with half its threads idle (226 us per 8-pair call at the nominal load against ~100 us of HBM time).
template <typename T>
__global__ __launch_bounds__(256) void fine_gather_rows(FgArgs a) {
    const int lane = threadIdx.x & 63, u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= 2 * a.M) return;
    const int side = u >= a.M, m = side ? u - a.M : u;
    const int b = (int)a.b_ids[m];
    const int cell = (int)(side ? a.j_ids[m] : a.i_ids[m]);
    const int wc = side ? a.w1c : a.w0c;
    const int Hf = side ? a.H1 : a.H0, Wf = side ? a.W1 : a.W0;
    const T* f = (const T*)(side ? a.f1 : a.f0);
    const long sn = side ? a.s1n : a.s0n, sh = side ? a.s1h : a.s0h, sw = side ? a.s1w : a.s0w;
    const int cy = (cell / wc) * a.stride - a.W / 2, cx = (cell % wc) * a.stride - a.W / 2;
    const int ppp = a.C / 8, pieces = a.W * a.W * ppp;                  // 16-byte pieces per position / per window
    T* out = (T*)a.win + (size_t)u * a.W * a.W * a.C;
    const v4u zero{0u, 0u, 0u, 0u};
    for (int e = lane; e < pieces; e += 64) {
        const int k = e / ppp, c8 = (e - k * ppp) * 8, y = cy + k / a.W, x = cx + k % a.W;
        v4u v = zero;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = *reinterpret_cast<const v4u*>(f + b * sn + y * sh + x * sw + c8);
        *reinterpret_cast<v4u*>(out + (size_t)k * a.C + c8) = v;
    }
    const T* cf = (const T*)(side ? a.c1 : a.c0) + ((size_t)b * (side ? a.S : a.L) + cell) * a.CC;
    T* co = (T*)a.ccat + (size_t)u * a.CC;
    for (int c8 = lane * 8; c8 < a.CC; c8 += 512) *reinterpret_cast<v4u*>(co + c8) = *reinterpret_cast<const v4u*>(cf + c8);
}

// ------------------------------------------------------------------------------------------------
struct FmArgs {
    const void* f0;          // [M][WW][C]
    const void* f1;
    int M, C;
    float temperature, thr;
    const int64_t* b_ids;    // coarse match -> sample
    const float* mk0c;       // [M][2]
    const float* mk1c;
    float coarse_scale, c2f, fine_scale;   // hw0_i/hw0_c, hw0_f/hw0_c, hw0_i/hw0_f
    const float* scale0;     // [N][2] or null
    const float* scale1;
    float* fine_matrix;      // [M][WW][WW]
    int32_t* sel;            // [M] flat arg-max index i*WW+j, or -1 when below thr
    int32_t* chunk_cnt;      // [chunks]
    int chunks;
    float* mk0f;             // [M][2] compacted
    float* mk1f;
    float* mconf;
    int64_t* m_bids;
    int32_t* count;          // [1]
};

constexpr int WW = 25;

template <typename T>
__global__ __launch_bounds__(256) void fine_match(FmArgs a) {
    __shared__ float s0[WW][129], s1[WW][129];
    __shared__ float sim[WW][WW + 1];
    __shared__ float rmax[WW], rsum[WW], cmax[WW], csum[WW];
    __shared__ unsigned long long best[256];
    const int m = blockIdx.x, t = threadIdx.x, C = a.C;
    const float rs = sqrtf((float)C);
    const T* p0 = (const T*)a.f0 + (size_t)m * WW * C;
    const T* p1 = (const T*)a.f1 + (size_t)m * WW * C;
    for (int i = t; i < WW * C; i += 256) {
        s0[i / C][i % C] = gf_to_float(p0[i]) / rs;      // feat / C**.5 on both sides (fine_matching2.py:52)
        s1[i / C][i % C] = gf_to_float(p1[i]) / rs;
    }
    __syncthreads();
    for (int o = t; o < WW * WW; o += 256) {
        const int i = o / WW, j = o % WW;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += s0[i][c] * s1[j][c];
        sim[i][j] = acc / a.temperature;
    }
    __syncthreads();
    if (t < WW) {                       // softmax over dim 2 (row statistics)
        float mx = -INFINITY;
        for (int j = 0; j < WW; ++j) mx = fmaxf(mx, sim[t][j]);
        float s = 0.f;
        for (int j = 0; j < WW; ++j) s += expf(sim[t][j] - mx);
        rmax[t] = mx; rsum[t] = s;
    } else if (t >= 64 && t < 64 + WW) {   // softmax over dim 1 (column statistics)
        const int j = t - 64;
        float mx = -INFINITY;
        for (int i = 0; i < WW; ++i) mx = fmaxf(mx, sim[i][j]);
        float s = 0.f;
        for (int i = 0; i < WW; ++i) s += expf(sim[i][j] - mx);
        cmax[j] = mx; csum[j] = s;
    }
    __syncthreads();
    unsigned long long key = 0ull;
    for (int o = t; o < WW * WW; o += 256) {
        const int i = o / WW, j = o % WW;
        const float s = sim[i][j];
        const float cf = (expf(s - cmax[j]) / csum[j]) * (expf(s - rmax[i]) / rsum[i]);
        a.fine_matrix[(size_t)m * WW * WW + o] = cf;
        const unsigned long long k = ((unsigned long long)__float_as_uint(cf) << 32) | (0xFFFFFFFFu - (unsigned)o);
        key = k > key ? k : key;        // arg-max with first-index tie-break (fine_matching2.py:79)
    }
    best[t] = key;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) best[t] = best[t] > best[t + s] ? best[t] : best[t + s];
        __syncthreads();
    }
    if (t == 0) {
        const float v = __uint_as_float((unsigned)(best[0] >> 32));
        const int o = (int)(0xFFFFFFFFu - (unsigned)(best[0] & 0xFFFFFFFFull));
        // the one-hot arg-max is automatically its row's and column's maximum, so the mask of
        // fine_matching2.py:73-82 reduces to the threshold test
        const bool ok = v > a.thr;
        a.sel[m] = ok ? o : -1;
    }
}

// 16-bit storage modes: ONE WAVE per match, the 25 x 25 x C correlation on the matrix cores (32 x 32 tile, rows / columns
// 25..31 re-read window position 24 and are masked out of the statistics), operands straight from global memory (a lane's
// 16 bytes of its window row per k-step).  The tile is computed TWICE, as f0.f1^T and as f1.f0^T (C / 16 MFMAs each, the same
// bits: a product's two factors commute and the k order is the same): a softmax statistic over a tile's ROW index is a
// reduction over the registers of one lane (+ one exchange between the lane halves), so the column statistics come from the
// first tile and the row statistics from the second - no 32-lane butterflies (those were 160 cross-lane operations per
// match); the row statistics reach the first tile's layout by v_readlane.  Hardware exp2 (the arguments are <= 0) and
// reciprocals of the two sums instead of libm's expf and 1250 divisions.  Same arithmetic as fine_match above up to fp32
// rounding (the dot products sum in the matrix cores' order and are scaled once by 1 / (C * temperature) instead of dividing every
// operand by sqrt(C)): 249 us -> HBM time per 8-pair call at the nominal load.
template <typename T>
__global__ __launch_bounds__(256) void fine_match_mfma(FmArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    const int lane = threadIdx.x & 63, h = lane >> 5, lr = lane & 31, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int C = a.C, row = lr < WW ? lr : WW - 1;
    const T* p0 = (const T*)a.f0 + ((size_t)m * WW + row) * C + h * 8;
    const T* p1 = (const T*)a.f1 + ((size_t)m * WW + row) * C + h * 8;
    v16f acc, acct;                     // acc[r]: row i = (r & 3) + 8 (r >> 2) + 4 h of column j = lr;  acct: the transpose
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acct[r] = 0.f; }
    for (int ks = 0; ks < C / 16; ++ks) {
        const Frag x0 = *reinterpret_cast<const Frag*>(p0 + ks * 16), x1 = *reinterpret_cast<const Frag*>(p1 + ks * 16);
        Mm::mma(x0, x1, acc);
        Mm::mma(x1, x0, acct);
    }
    const float scale2 = 1.4426950408889634f / ((float)C * a.temperature);     // logits in log2 units
    const bool lok = lr < WW;           // this lane's column (first tile) / row (second tile) exists
    // statistics over the register index: first tile -> column lr, second tile -> row lr
    float cmx = -INFINITY, rmx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = gf_acc_row(r, h) < WW;
        acc[r] *= scale2;
        acct[r] *= scale2;
        cmx = fmaxf(cmx, ok ? acc[r] : -INFINITY);
        rmx = fmaxf(rmx, ok ? acct[r] : -INFINITY);
    }
    cmx = fmaxf(cmx, __shfl_xor(cmx, 32, 64));
    rmx = fmaxf(rmx, __shfl_xor(rmx, 32, 64));
    float csm = 0.f, rsm = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool ok = gf_acc_row(r, h) < WW;
        csm += ok ? __builtin_amdgcn_exp2f(acc[r] - cmx) : 0.f;
        rsm += ok ? __builtin_amdgcn_exp2f(acct[r] - rmx) : 0.f;
    }
    csm += __shfl_xor(csm, 32, 64);
    rsm += __shfl_xor(rsm, 32, 64);
    const float icsm = 1.0f / csm, irsm = 1.0f / rsm;
    unsigned long long key = 0ull;
    float* fm = a.fine_matrix + (size_t)m * WW * WW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // row i of this register differs between the lane halves: both candidates by v_readlane (wave-uniform lane index)
        const int i0 = gf_acc_row(r, 0), i1 = gf_acc_row(r, 1), i = h ? i1 : i0;
        const int m0 = __builtin_amdgcn_readlane(__float_as_int(rmx), i0), m1 = __builtin_amdgcn_readlane(__float_as_int(rmx), i1);
        const int q0 = __builtin_amdgcn_readlane(__float_as_int(irsm), i0), q1 = __builtin_amdgcn_readlane(__float_as_int(irsm), i1);
        const float rm = __int_as_float(h ? m1 : m0), rq = __int_as_float(h ? q1 : q0);
        if (i < WW && lok) {
            const float cf = (__builtin_amdgcn_exp2f(acc[r] - cmx) * icsm) * (__builtin_amdgcn_exp2f(acc[r] - rm) * rq);
            const int o = i * WW + lr;
            fm[o] = cf;
            const unsigned long long k = ((unsigned long long)__float_as_uint(cf) << 32) | (0xFFFFFFFFu - (unsigned)o);
            key = k > key ? k : key;            // arg-max with first-index tie-break (fine_matching2.py:79)
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)key, d, 64), hi = __shfl_xor((unsigned)(key >> 32), d, 64);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        key = other > key ? other : key;
    }
    if (lane == 0) {
        const float v = __uint_as_float((unsigned)(key >> 32));
        const int o = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        const bool ok = v > a.thr;
        a.sel[m] = ok ? o : -1;
    }
}

// matches that passed the threshold, per 1024-match chunk (one ballot per wave; the per-match atomicAdd this replaces put ~1000
// same-address atomics on each of ~18 counters: 140 of fine_match's 214 us at the nominal load)
__global__ __launch_bounds__(1024) void fine_count(FmArgs a) {
    __shared__ int wave_tot[16];
    const int c = blockIdx.x, tid = threadIdx.x, m = c * 1024 + tid;
    const unsigned long long bal = __ballot(m < a.M && a.sel[m] >= 0);
    if ((tid & 63) == 0) wave_tot[tid >> 6] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += wave_tot[w];
        a.chunk_cnt[c] = t;
    }
}

__global__ __launch_bounds__(1024) void fine_compact(FmArgs a) {
    __shared__ int wave_tot[16];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0, total = 0;
    for (int k = 0; k < a.chunks; ++k) {
        const int v = a.chunk_cnt[k];
        if (k < c) base += v;
        total += v;
    }
    if (c == 0 && tid == 0) a.count[0] = total;
    const int m = c * 1024 + tid;
    const int o = m < a.M ? a.sel[m] : -1;
    const bool f = o >= 0;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wave) ? wave_tot[w] : 0;
    if (!f) return;
    const int pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
    const int i = o / WW, j = o % WW, b = (int)a.b_ids[m];
    const int W = 5;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float cs0 = a.scale0 ? a.coarse_scale * a.scale0[2 * b + d] : a.coarse_scale;
        const float cs1 = a.scale1 ? a.coarse_scale * a.scale1[2 * b + d] : a.coarse_scale;
        const float fs0 = a.scale0 ? a.fine_scale * a.scale0[2 * b + d] : a.fine_scale;
        const float fs1 = a.scale1 ? a.fine_scale * a.scale1[2 * b + d] : a.fine_scale;
        const float c0 = a.mk0c[2 * m + d] / cs0 * a.c2f;             // fine_matching2.py:101-102
        const float c1 = a.mk1c[2 * m + d] / cs1 * a.c2f;
        const int oi = d == 0 ? (i % W - W / 2) : (i / W - W / 2);
        const int oj = d == 0 ? (j % W - W / 2) : (j / W - W / 2);
        a.mk0f[2 * pos + d] = ((float)oi + c0) * fs0;                 // :104-116
        a.mk1f[2 * pos + d] = ((float)oj + c1) * fs1;
    }
    a.mconf[pos] = a.fine_matrix[(size_t)m * WW * WW + o];
    a.m_bids[pos] = b;
}

template <typename TF, typename T>
int fg_launch(const FgArgs& a, hipStream_t st) {
    if constexpr (std::is_same<TF, T>::value && sizeof(T) == 2) {
        const bool rows16 = a.s0c == 1 && a.s1c == 1 && a.C % 8 == 0 && a.CC % 8 == 0 && (uintptr_t)a.f0 % 16 == 0 && (uintptr_t)a.f1 % 16 == 0 &&
                            (uintptr_t)a.c0 % 16 == 0 && (uintptr_t)a.c1 % 16 == 0 && (uintptr_t)a.win % 16 == 0 && (uintptr_t)a.ccat % 16 == 0 &&
                            a.s0n % 8 == 0 && a.s0h % 8 == 0 && a.s0w % 8 == 0 && a.s1n % 8 == 0 && a.s1h % 8 == 0 && a.s1w % 8 == 0;
        if (rows16) {
            fine_gather_rows<T><<<(2 * a.M + 3) / 4, 256, 0, st>>>(a);
            GF_CHECK_LAUNCH();
            return GF_OK;
        }
    }
    fine_gather<TF, T><<<dim3(a.M, 2), 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

}   // namespace

extern "C" int gf_fine_gather(const void* feat_f0, const void* feat_f1, int feat_dtype, const long* strides0,
                              const long* strides1, int H0, int W0, int H1, int W1, int C, const void* feat_c0,
                              const void* feat_c1, int dtype, int L, int S, int CC, const int64_t* b_ids,
                              const int64_t* i_ids, const int64_t* j_ids, int M, int w0c, int w1c, int stride,
                              int window, void* win_out, void* ccat_out, void* stream) {
    GF_CHECK_ARG(feat_f0 && feat_f1 && strides0 && strides1 && feat_c0 && feat_c1 && b_ids && i_ids && j_ids && win_out && ccat_out, "null pointer");
    GF_CHECK_ARG(M > 0, "M must be > 0 (the M == 0 early return of fine_preprocess.py:35-38 is the caller's)");
    GF_CHECK_ARG(C > 0 && C <= 256 && window > 0 && stride > 0 && w0c > 0 && w1c > 0, "bad sizes");
    GF_CHECK_ARG(feat_dtype >= GF_F32 && feat_dtype <= GF_BF16 && dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(feat_dtype == GF_F32 || dtype == GF_F32 || feat_dtype == dtype, "fp16 <-> bf16 conversion is not built");
    FgArgs a{feat_f0, feat_f1, strides0[0], strides0[1], strides0[2], strides0[3], strides1[0], strides1[1], strides1[2],
             strides1[3], H0, W0, H1, W1, C, feat_c0, feat_c1, L, S, CC, b_ids, i_ids, j_ids, M, w0c, w1c, stride, window,
             win_out, ccat_out};
    hipStream_t st = (hipStream_t)stream;
    if (feat_dtype == GF_F32)
        return dtype == GF_F32 ? fg_launch<float, float>(a, st)
                               : dtype == GF_F16 ? fg_launch<float, _Float16>(a, st) : fg_launch<float, gf_bf16>(a, st);
    if (feat_dtype == GF_F16) return dtype == GF_F32 ? fg_launch<_Float16, float>(a, st) : fg_launch<_Float16, _Float16>(a, st);
    return dtype == GF_F32 ? fg_launch<gf_bf16, float>(a, st) : fg_launch<gf_bf16, gf_bf16>(a, st);
}

extern "C" size_t gf_fine_match_workspace_bytes(int M) {
    if (M <= 0) return 0;
    return gf_align_up((size_t)((M + 1023) / 1024) * sizeof(int32_t), 256) + gf_align_up((size_t)M * sizeof(int32_t), 256);
}

extern "C" int gf_fine_match(const void* f0, const void* f1, int dtype, int M, int WWin, int C, float temperature,
                             float thr, const int64_t* b_ids, const float* mkpts0_c, const float* mkpts1_c,
                             float coarse_scale, float c2f_scale, float fine_scale, const float* scale0,
                             const float* scale1, float* fine_matrix, float* mkpts0_f, float* mkpts1_f, float* mconf,
                             int64_t* m_bids, int32_t* count, void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(f0 && f1 && b_ids && mkpts0_c && mkpts1_c && fine_matrix && mkpts0_f && mkpts1_f && mconf && m_bids && count, "null pointer");
    GF_CHECK_ARG(M > 0, "M must be > 0 (the M == 0 early return of fine_matching2.py:34-42 is the caller's)");
    GF_CHECK_ARG(WWin == WW && C > 0 && C <= 128, "built for 5x5 windows and C <= 128");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    if (workspace == nullptr || workspace_bytes < gf_fine_match_workspace_bytes(M)) {
        gf_set_error("gf_fine_match: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    FmArgs a;
    a.f0 = f0; a.f1 = f1; a.M = M; a.C = C; a.temperature = temperature; a.thr = thr; a.b_ids = b_ids;
    a.mk0c = mkpts0_c; a.mk1c = mkpts1_c; a.coarse_scale = coarse_scale; a.c2f = c2f_scale; a.fine_scale = fine_scale;
    a.scale0 = scale0; a.scale1 = scale1; a.fine_matrix = fine_matrix;
    a.chunks = (M + 1023) / 1024;
    a.chunk_cnt = (int32_t*)workspace;
    a.sel = (int32_t*)((char*)workspace + gf_align_up((size_t)a.chunks * sizeof(int32_t), 256));
    a.mk0f = mkpts0_f; a.mk1f = mkpts1_f; a.mconf = mconf; a.m_bids = m_bids; a.count = count;
    hipStream_t st = (hipStream_t)stream;
    const bool mfma = dtype != GF_F32 && C % 16 == 0 && (uintptr_t)f0 % 16 == 0 && (uintptr_t)f1 % 16 == 0;
    if (dtype == GF_F32) fine_match<float><<<M, 256, 0, st>>>(a);
    else if (dtype == GF_F16) {
        if (mfma) fine_match_mfma<_Float16><<<(M + 3) / 4, 256, 0, st>>>(a);
        else fine_match<_Float16><<<M, 256, 0, st>>>(a);
    } else {
        if (mfma) fine_match_mfma<gf_bf16><<<(M + 3) / 4, 256, 0, st>>>(a);
        else fine_match<gf_bf16><<<M, 256, 0, st>>>(a);
    }
    fine_count<<<a.chunks, 1024, 0, st>>>(a);
    fine_compact<<<a.chunks, 1024, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
