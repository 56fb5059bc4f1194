// K1: dual-softmax correlation sweep + mutual-nearest match extraction (CDNA4 / gfx950).
//
// Replaces CoarseMatching.forward / get_coarse_match of the reference
// (model/loftr_src/loftr/utils/coarse_matching.py:90-130, :132-212): there, ~10 full passes over
// the [N,L,S] fp32 matrix (einsum, 2 softmax, >, 2 max, 2 ==, 2 *, max, where).  Here:
//
//   pass A  k1_stats   : 128x128 MFMA tiles of sim = f0.f1^T/(C*tau); per tile the row and column
//                        (max, sum-exp) partials -> workspace.                       (no L*S traffic)
//   reduce  k1_reduce  : combine partials -> row/col softmax statistics.
//   pass B  k1_conf    : recompute the same tile (bit-identical), conf = softmax_col*softmax_row,
//                        ONE streaming write of conf [N,L,S] fp32 (the algorithmic 4*L*S bytes).
//                        Entries with conf > thr are rare (a row holds at most 1/thr of them), so the
//                        row-best (value, first column) key and the column maximum are kept with
//                        atomicMax on those candidates only - no cross-lane reduction in the hot loop
//                        (a dense in-tile reduction is used instead when thr < 0.05).
//   select  k1_select  : per row: best key, equality with the column maximum; exact
//                        first-True-column semantics incl. the tie case (re-reads that ONE row).
//   compact k1_compact : ordered compaction (torch.where order) + keypoint arithmetic.
//
// HBM roofline: pass B is bound by the conf write (L*S*4 B per sample); everything else is O(L*C).
// Measured (8 samples of 6400^2, fp16): the same two-128-B-segment store pattern alone reaches 5.6 TB/s
// (tools/probes/store_pattern.hip); pass B reaches ~3.2 TB/s: ablating the stores or 3 of its 4 K steps
// removes ~195 us each and the two do not overlap.  Non-temporal stores (conf is never re-read by this
// launch) bought 8 %.  The row-panel-persistent form (k1_conf_panel: f0 rows kept as A fragments in
// registers, f1 tiles prefetched a whole tile ahead, 3x fewer operand bytes) is 3-6 % faster (0.344 ms per
// 8 pairs); its phase trace (tools/k1_trace.py, -DK1_TRACE=1) shows per tile ~1000 ticks staging, ~2700
// MFMA + fragment reads, 4000 (thr > 0) to 5200 (dense candidates) epilogue - the epilogue is paced by the
// HBM write drain (32 KiB per tile per workgroup = ~3500 ticks at the measured 5.6 TB/s store ceiling).
#include <math.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gf_common.h"

// This file is compiled twice.  K1_PART 0 (this file's own object): everything except the instantiations of k1_stats_panel.
// K1_PART 1 (k1_stats_noslp.hip, which includes this file): k1_stats_panel and its launcher alone, built with
// -fno-slp-vectorize: packed into v_pk_*_f32 pairs its tile epilogue needs 256 registers + 76 bytes of scratch (a spill reload
// waits for vmcnt(0), i.e. for the LDS-DMA in flight), unpacked 229 and none: 328 -> 285 us per 8-pair call on the same box.
// k1_conf_pipe is 3 % faster WITH the packing, hence the split and not a file-wide flag.
#ifndef K1_PART
#define K1_PART 0
#endif
void gf_k1_stats_panel_launch(const void* k1args, int dtype, int wgs, void* stream);

namespace {

constexpr int BM = 128, BN = 64, NT = 256;      // workgroup tile; 4 waves stacked along M, each 32 rows x 64 cols
constexpr int ROWB = 128;                        // bytes per LDS operand row = one K step
constexpr int STAGE_BYTES = (BM + BN) * ROWB;    // 24 KiB, single stage (the next step's tile waits in registers)
constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f;

struct K1Args {
    const void* f0;
    const void* f1;
    int N, L, S, C;
    const uint8_t* mask0;
    const uint8_t* mask1;
    float inv_c, temperature, mult;   // sim = acc*inv_c/temperature (exact) or acc*mult (fast)
    int tilesM, tilesN;
    int rowparts;      // number of row partials per row: tilesN (tile form) or the number of tile runs (panel form)
    float2* rowpart;   // [N][rowparts][L]  (max, sumexp)
    float2* colpart;   // [N][tilesM][S]
    float2* rstat;     // [N][L]  (max, sum)
    float2* cstat;     // [N][S]
    unsigned long long* rowbest;   // [N][L]  max over candidates of (conf bits << 32 | ~col); 0 = none
    unsigned* colmax;              // [N][S]  max conf bits over candidates; 0 = none
    float* conf;
    float thr;
    int dense;                     // thr so low that candidates are not rare: reduce in the tile first
    unsigned long long* stamp;     // [K1_STAMP_WORDS] or null: what the statistics in this workspace belong to (k1_stamp_words)
    int esize;                     // bytes per feature element (the stamp's content fingerprint samples whole 8-byte words)
};

// ---------------------------------------------------------------------------------------------
// tile engine: acc[ni] = the wave's 32 x 64 block (two MFMA 32x32 tiles) of f0[m0:,:] . f1[n0:,:]^T.
// 24 KiB of LDS and < 128 VGPRs per lane keep 4 workgroups (16 waves) resident per CU: the epilogue of
// one overlaps the loads and MFMAs of the others.  Operand rows are clamped; validity is applied later.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void sim_tile(const T* __restrict__ A, const T* __restrict__ B, int L, int S, int C,
                                         int m0, int n0, char* smem, v16f (&acc)[2]) {
    using M = Mma32<T>;
    using Frag = typename M::Frag;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = ROWB / sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    const int srow = tid >> 3, schunk = tid & 7;
    v4u ra[4], rb[2];
    const T* ga[4];
    const T* gb[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) ga[p] = A + (size_t)min(m0 + srow + 32 * p, L - 1) * C + schunk * EPC;
#pragma unroll
    for (int p = 0; p < 2; ++p) gb[p] = B + (size_t)min(n0 + srow + 32 * p, S - 1) * C + schunk * EPC;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    char* sa = smem;
    char* sb = smem + BM * ROWB;
    const int nk = C / BK;
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const v4u*>(ga[p]);
#pragma unroll
    for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const v4u*>(gb[p]);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<v4u*>(sa + gf_lds_off(srow + 32 * p, schunk)) = ra[p];
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<v4u*>(sb + gf_lds_off(srow + 32 * p, schunk)) = rb[p];
        __syncthreads();
        if (kt + 1 < nk) {
            const int k0 = (kt + 1) * BK;
#pragma unroll
            for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const v4u*>(ga[p] + k0);
#pragma unroll
            for (int p = 0; p < 2; ++p) rb[p] = *reinterpret_cast<const v4u*>(gb[p] + k0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = 2 * g + h;
            const Frag a0 = *reinterpret_cast<const Frag*>(sa + gf_lds_off(wave * 32 + lr, chunk));
            const Frag b0 = *reinterpret_cast<const Frag*>(sb + gf_lds_off(lr, chunk));
            const Frag b1 = *reinterpret_cast<const Frag*>(sb + gf_lds_off(32 + lr, chunk));
            M::mma(a0, b0, acc[0]);
            M::mma(a0, b1, acc[1]);
        }
    }
    __syncthreads();                 // the staging area is reused as scratch by the epilogues
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
// id % 8 labels the XCD.  Giving every XCD a CONTIGUOUS run of tiles (row panel after row panel) keeps
// its f0 row panel and the f1 column tiles in that XCD's L2 instead of re-fetching them 8 times.
// Bijective for any tile count (speed only; correctness never depends on placement).
__device__ __forceinline__ void k1_tile(const K1Args& a, int& bm, int& bn) {
    const int nt = a.tilesM * a.tilesN, id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, q = nt >> 3, r = nt & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bm = t / a.tilesN;
    bn = t % a.tilesN;
}

template <bool EXACT>
__device__ __forceinline__ float k1_exp(float x) {
    if constexpr (EXACT) return expf(x);
    else return __expf(x);
}

// Per-lane view of the wave's 32x64 block after the MFMAs: slot r (0..15) -> row m0 + wave*32 +
// acc_row(r, h); column ni -> n0 + ni*32 + (lane & 31).
struct LaneGeom {
    unsigned row_in, row_ok;   // bit r: row < L ; mask0 true (or no mask)
    unsigned col_in, col_ok;   // bit ni
};

// Branch-free; the packed words are made opaque so that the compiler keeps them as VGPRs instead of
// dozens of live lane masks in SGPRs (which spilled thousands of SGPRs).
__device__ __forceinline__ LaneGeom k1_geom(const K1Args& a, int n, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, lr = lane & 31;
    LaneGeom g;
    const int row0 = m0 + wave * 32, col0 = n0 + lr;
    g.row_in = 0;
    g.row_ok = 0;
    const bool masked = a.mask0 != nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + gf_acc_row(r, h);
        const unsigned in = row < a.L ? 1u : 0u;
        g.row_in |= in << r;
        if (masked) g.row_ok |= (in & (a.mask0[(size_t)n * a.L + min(row, a.L - 1)] != 0 ? 1u : 0u)) << r;
    }
    if (!masked) g.row_ok = g.row_in;
    g.col_in = 0;
    g.col_ok = 0;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = col0 + ni * 32;
        const unsigned in = col < a.S ? 1u : 0u;
        g.col_in |= in << ni;
        if (masked) g.col_ok |= (in & (a.mask1[(size_t)n * a.S + min(col, a.S - 1)] != 0 ? 1u : 0u)) << ni;
    }
    if (!masked) g.col_ok = g.col_in;
    asm volatile("" : "+v"(g.row_in), "+v"(g.row_ok), "+v"(g.col_in), "+v"(g.col_ok));
    return g;
}

// sim values of this lane: sv[ni][r]; out-of-range -> -inf (ignored by every reduction),
// masked pair -> -1e9 exactly as masked_fill does (coarse_matching.py:123-124).
// GUARD=false is the interior, unmasked tile: no predicates at all.
template <bool EXACT, bool GUARD>
__device__ __forceinline__ void k1_sim_values(const K1Args& a, const LaneGeom& g, const v16f (&acc)[2], float (&sv)[2][16]) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float raw = acc[ni][r];
            float s;
            if constexpr (EXACT) s = (raw * a.inv_c) / a.temperature;
            else s = raw * a.mult;
            if constexpr (GUARD) {
                const bool in = ((g.row_in >> r) & (g.col_in >> ni) & 1u) != 0;
                const bool ok = ((g.row_ok >> r) & (g.col_ok >> ni) & 1u) != 0;
                s = in ? (ok ? s : -1e9f) : NEG_INF;
            }
            sv[ni][r] = s;
        }
}

__device__ __forceinline__ bool k1_interior(const K1Args& a, int m0, int n0) {
    return a.mask0 == nullptr && m0 + BM <= a.L && n0 + BN <= a.S;
}

// reduction over the 64 columns of the wave's block for each of its 16 row slots: the 32 (ni, r) values
// go through the 32-lane reduce-scatter (lane c ends with slot c = ni*16 + r), lanes c and c^16 then hold
// the two column halves of row slot r = c & 15
template <typename V, typename Op>
__device__ __forceinline__ V k1_row_reduce(V (&v)[32], Op op) {
    const V x = gf_reduce_scatter32(v, op);
    return op(x, gf_shfl_xor16(x));
}

// ---------------------------------------------------------------------------------------------
// pass A: row / column (max, sum-exp) partials of one tile
// ---------------------------------------------------------------------------------------------
template <typename T, bool GUARD>
__device__ __forceinline__ void k1_stats_epilogue(const K1Args& a, const v16f (&acc)[2], char* smem, int n, int bm, int bn) {
    constexpr bool EXACT = std::is_same<T, float>::value;
    const int m0 = bm * BM, n0 = bn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, lr = lane & 31;
    LaneGeom g;
    if constexpr (GUARD) g = k1_geom(a, n, m0, n0);
    float sv[2][16];
    k1_sim_values<EXACT, GUARD>(a, g, acc, sv);
    float* rowbc = reinterpret_cast<float*>(smem) + wave * 32;             // [4][32] row maxima, per wave
    float2* colx = reinterpret_cast<float2*>(smem + 512);                   // [4 waves][64] column partials
    // ---- columns: lane-local over its 16 rows, then the other lane half
    float cm[2], cl[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        float m = NEG_INF;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sv[ni][r]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float ms = (GUARD && m == NEG_INF) ? 0.f : m;
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) l += k1_exp<EXACT>(sv[ni][r] - ms);
        l += __shfl_xor(l, 32, 64);
        cm[ni] = m;
        cl[ni] = l;
    }
    if (h == 0) {
        colx[wave * 64 + lr] = make_float2(cm[0], cl[0]);
        colx[wave * 64 + 32 + lr] = make_float2(cm[1], cl[1]);
    }
    // ---- rows
    float v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = sv[q >> 4][q & 15];
    const float rmax = k1_row_reduce(v, GfMaxF());             // lanes c, c^16: row slot c & 15
    if (lr < 16) rowbc[h * 16 + lr] = rmax;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const float m = rowbc[h * 16 + (q & 15)];
        const float ms = (GUARD && m == NEG_INF) ? 0.f : m;
        v[q] = k1_exp<EXACT>(sv[q >> 4][q & 15] - ms);
    }
    const float rsum = k1_row_reduce(v, GfAddF());
    if (lr < 16) {
        const int row = m0 + wave * 32 + gf_acc_row(lr, h);
        if (row < a.L) a.rowpart[((size_t)n * a.tilesN + bn) * a.L + row] = make_float2(rmax, rsum);
    }
    // ---- combine the four waves' column partials
    const int t = threadIdx.x;
    if (t < 64) {
        float m = NEG_INF;
#pragma unroll
        for (int w = 0; w < 4; ++w) m = fmaxf(m, colx[w * 64 + t].x);
        const float ms = (m == NEG_INF) ? 0.f : m;
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) l += colx[w * 64 + t].y * k1_exp<EXACT>(colx[w * 64 + t].x - ms);
        const int col = n0 + t;
        if (col < a.S) a.colpart[((size_t)n * a.tilesM + bm) * a.S + col] = make_float2(m, l);
    }
}

template <typename T>
__global__ __launch_bounds__(NT, 4) void k1_stats(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = blockIdx.y;
    int bm, bn;
    k1_tile(a, bm, bn);
    v16f acc[2];
    sim_tile<T>((const T*)a.f0 + (size_t)n * a.L * a.C, (const T*)a.f1 + (size_t)n * a.S * a.C, a.L, a.S, a.C,
                bm * BM, bn * BN, smem, acc);
    if (k1_interior(a, bm * BM, bn * BN)) k1_stats_epilogue<T, false>(a, acc, smem, n, bm, bn);
    else k1_stats_epilogue<T, true>(a, acc, smem, n, bm, bn);
}

// The row / column statistics a gf_dual_softmax_match call leaves in its workspace are stamped with what they were computed
// from (shape, scale, the two feature pointers); gf_dual_softmax_conf_at compares the stamp with its own arguments on the
// device and returns NaN for every entry when they differ (nothing else ties a later conf_at call to that call).
// Round 5: the caching allocator hands the address of a freed feature tensor to the next one of the same shape (the
// second CoarseMatching pass, the next batch), so pointers alone do not tell whose statistics the workspace holds: the LAST word is a
// fingerprint of the CONTENT - 32 eight-byte samples of each tensor, evenly spaced, one per lane of a wave, mixed and XOR-reduced.
// A SAMPLED check (512 bytes): it catches the workspace being reused for other tensors at the same addresses (the case that happens:
// the caching allocator hands the next batch the same blocks), not an in-place edit of a few rows between the two calls - a caller that
// edits f0 / f1 in place re-runs the statistics pass (ops.py says so).
constexpr int K1_STAMP_WORDS = 5;          // the words k1_stamp_word() makes; word K1_STAMP_WORDS is the fingerprint
__device__ __forceinline__ unsigned long long k1_fingerprint(const void* f0, const void* f1, size_t bytes0, size_t bytes1, int lane) {
    const bool second = lane >= 32;
    const unsigned short* p = (const unsigned short*)(second ? f1 : f0);      // two-byte pieces: a 16-bit tensor at an odd storage offset
    const size_t words = (second ? bytes1 : bytes0) / 8;                      // is only 2-byte aligned
    const int k = lane & 31;
    unsigned long long w = 0ull;
    if (words) {
        const unsigned short* q = p + 4 * ((words - 1) * (size_t)k / 31);
        w = (unsigned long long)q[0] | ((unsigned long long)q[1] << 16) | ((unsigned long long)q[2] << 32) | ((unsigned long long)q[3] << 48);
    }
    w = (w + 0x9e3779b97f4a7c15ull * (unsigned long long)(lane + 1)) * 0xff51afd7ed558ccdull;
    w ^= w >> 33;
    unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo ^= (unsigned)__shfl_xor((int)lo, d, 64);
        hi ^= (unsigned)__shfl_xor((int)hi, d, 64);
    }
    return ((unsigned long long)hi << 32) | lo;
}
__host__ __device__ __forceinline__ unsigned long long k1_stamp_word(int k, const void* f0, const void* f1, int N, int L, int S, int C, float mult) {
    union { float f; unsigned u; } m;
    m.f = mult;
    switch (k) {
        case 0: return 0x6b31737461747321ull;                                              // "k1stats!": a stamp has been written at all
        case 1: return ((unsigned long long)(unsigned)N << 32) | (unsigned)L;
        case 2: return ((unsigned long long)(unsigned)S << 32) | (unsigned)C;
        case 3: return (unsigned long long)(uintptr_t)f0 ^ ((unsigned long long)m.u << 40);
        default: return (unsigned long long)(uintptr_t)f1;
    }
}

// combine per-tile (max, sumexp) partials: 32 rows (blockIdx.y==0) or columns (==1) per block,
// 8 threads per row each striding over the partials, merged through LDS
template <bool EXACT>
__global__ __launch_bounds__(256) void k1_reduce_stats(K1Args a) {
    __shared__ float2 sh[8][32];
    const int n = blockIdx.z;
    const bool rows = blockIdx.y == 0;
    if (a.stamp != nullptr && threadIdx.x < 64 && blockIdx.x == 0 && blockIdx.y == 0 && n == 0) {          // wave 0 of one block
        const unsigned long long fp = k1_fingerprint(a.f0, a.f1, (size_t)a.N * a.L * a.C * a.esize, (size_t)a.N * a.S * a.C * a.esize, threadIdx.x);
        if (threadIdx.x < K1_STAMP_WORDS) a.stamp[threadIdx.x] = k1_stamp_word(threadIdx.x, a.f0, a.f1, a.N, a.L, a.S, a.C, a.mult);
        if (threadIdx.x == K1_STAMP_WORDS) a.stamp[K1_STAMP_WORDS] = fp;
    }
    const int len = rows ? a.L : a.S, np = rows ? a.rowparts : a.tilesM;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    if (blockIdx.x * 32 >= len) return;
    const float2* part = (rows ? a.rowpart : a.colpart) + (size_t)n * np * len + min(i, len - 1);
    float m = NEG_INF, l = 0.f;
    for (int p = ty; p < np; p += 8) {
        const float2 v = part[(size_t)p * len];
        const float mn = fmaxf(m, v.x);
        const float ms = (mn == NEG_INF) ? 0.f : mn;
        l = l * k1_exp<EXACT>(m - ms) + v.y * k1_exp<EXACT>(v.x - ms);
        m = mn;
    }
    sh[ty][tx] = make_float2(m, l);
    __syncthreads();
    if (ty == 0 && i < len) {
        float M = NEG_INF;
#pragma unroll
        for (int y = 0; y < 8; ++y) M = fmaxf(M, sh[y][tx].x);
        const float ms = (M == NEG_INF) ? 0.f : M;
        float lt = 0.f;
#pragma unroll
        for (int y = 0; y < 8; ++y) lt += sh[y][tx].y * k1_exp<EXACT>(sh[y][tx].x - ms);
        (rows ? a.rstat : a.cstat)[(size_t)n * len + i] = make_float2(M, lt);
    }
}

// ---------------------------------------------------------------------------------------------
// pass B: confidence tile -> HBM, row-best keys, column maxima
// ---------------------------------------------------------------------------------------------
template <typename T, bool GUARD, bool DENSE>
__device__ __forceinline__ void k1_conf_epilogue(const K1Args& a, const v16f (&acc)[2], char* smem, int n, int bm, int bn) {
    constexpr bool EXACT = std::is_same<T, float>::value;
    const int m0 = bm * BM, n0 = bn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, lr = lane & 31;
    LaneGeom g;
    if constexpr (GUARD) g = k1_geom(a, n, m0, n0);
    // row statistics of the tile -> LDS once: EXACT keeps (max, sum); the fast path stores
    // (-max*log2e, 1/sum) so that conf = exp2(2*s*log2e + A_row + B_col) * (1/sum_row) * (1/sum_col),
    // ONE exponential per element (= softmax_col * softmax_row up to fp32 rounding)
    float2* rst = reinterpret_cast<float2*>(smem);
    if (threadIdx.x < BM) {
        const float2 st = a.rstat[(size_t)n * a.L + min(m0 + (int)threadIdx.x, a.L - 1)];
        rst[threadIdx.x] = EXACT ? st : make_float2(-st.x * LOG2E, __builtin_amdgcn_rcpf(st.y));
    }
    float ca[2], cb[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = GUARD ? min(n0 + ni * 32 + lr, a.S - 1) : n0 + ni * 32 + lr;
        const float2 st = a.cstat[(size_t)n * a.S + col];
        ca[ni] = EXACT ? st.x : -st.x * LOG2E;
        cb[ni] = EXACT ? st.y : __builtin_amdgcn_rcpf(st.y);
    }
    __syncthreads();
    const float k2 = 2.0f * a.mult * LOG2E;
    const int row_base = m0 + __builtin_amdgcn_readfirstlane(wave) * 32;
    float* cbase = a.conf + ((size_t)n * a.L + row_base) * a.S + n0;
    unsigned long long* rbest = a.rowbest + (size_t)n * a.L;
    unsigned* cmax = a.colmax + (size_t)n * a.S;
    const int lane_off = 4 * h * a.S + lr;               // 32-bit per-lane part of the store address
    unsigned long long key[DENSE ? 32 : 1];
    unsigned cbest[2] = {0u, 0u};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 st = rst[wave * 32 + gf_acc_row(r, h)];
        float cf[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if constexpr (EXACT) {
                float s = (acc[ni][r] * a.inv_c) / a.temperature;
                if constexpr (GUARD) {
                    const bool ok = ((g.row_ok >> r) & (g.col_ok >> ni) & 1u) != 0;
                    s = ok ? s : -1e9f;
                }
                cf[ni] = (expf(s - ca[ni]) / cb[ni]) * (expf(s - st.x) / st.y);
            } else {
                float e;
                if constexpr (GUARD) {
                    const bool ok = ((g.row_ok >> r) & (g.col_ok >> ni) & 1u) != 0;
                    const float s2 = ok ? acc[ni][r] * k2 : -2e9f * LOG2E;
                    e = __builtin_amdgcn_exp2f(s2 + (st.x + ca[ni]));
                } else {
                    e = __builtin_amdgcn_exp2f(fmaf(acc[ni][r], k2, st.x + ca[ni]));
                }
                cf[ni] = e * (st.y * cb[ni]);
            }
        }
        // the row of slot r: uniform base + compile-time multiple of S; lanes add their 32-bit offset
        float* rowp = cbase + (size_t)((r & 3) + 8 * (r >> 2)) * a.S;
        bool in0 = true, in1 = true;
        if constexpr (GUARD) {
            in0 = ((g.row_in >> r) & g.col_in & 1u) != 0;
            in1 = ((g.row_in >> r) & (g.col_in >> 1) & 1u) != 0;
        }
        if (in0) __builtin_nontemporal_store(cf[0], rowp + lane_off);
        if (in1) __builtin_nontemporal_store(cf[1], rowp + lane_off + 32);
        if constexpr (!DENSE) {
            // candidates above thr are rare (<= 1/thr per row or column): atomics on them only
            if (fmaxf(cf[0], cf[1]) > a.thr) {
                const int row = row_base + gf_acc_row(r, h);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (cf[ni] > a.thr && (ni == 0 ? in0 : in1)) {
                        const int col = n0 + ni * 32 + lr;
                        const unsigned bits = __float_as_uint(cf[ni]);
                        atomicMax(rbest + row, ((unsigned long long)bits << 32) | (0xFFFFFFFFu - (unsigned)col));
                        atomicMax(cmax + col, bits);
                    }
                }
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bool in = cf[ni] > a.thr && (ni == 0 ? in0 : in1);
                const unsigned bits = in ? __float_as_uint(cf[ni]) : 0u;
                cbest[ni] = max(cbest[ni], bits);
                key[ni * 16 + r] = in ? (((unsigned long long)bits << 32) | (0xFFFFFFFFu - (unsigned)(n0 + ni * 32 + lr))) : 0ull;
            }
        }
    }
    if constexpr (DENSE) {
        const unsigned long long kbest = k1_row_reduce(key, GfMaxU64());
        if (lr < 16 && kbest != 0ull) atomicMax(rbest + row_base + gf_acc_row(lr, h), kbest);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const unsigned c = max(cbest[ni], (unsigned)__shfl_xor((int)cbest[ni], 32, 64));
            if (h == 0 && c != 0u) atomicMax(cmax + n0 + ni * 32 + lr, c);
        }
    }
}

template <typename T, bool DENSE>
__global__ __launch_bounds__(NT, 4) void k1_conf(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = blockIdx.y;
    int bm, bn;
    k1_tile(a, bm, bn);
    v16f acc[2];
    sim_tile<T>((const T*)a.f0 + (size_t)n * a.L * a.C, (const T*)a.f1 + (size_t)n * a.S * a.C, a.L, a.S, a.C,
                bm * BM, bn * BN, smem, acc);
    if (k1_interior(a, bm * BM, bn * BN)) k1_conf_epilogue<T, false, DENSE>(a, acc, smem, n, bm, bn);
    else k1_conf_epilogue<T, true, DENSE>(a, acc, smem, n, bm, bn);
}

// ---------------------------------------------------------------------------------------------
// pass B, row-panel-persistent form (fp16, unmasked, L % 128 == 0, S % 64 == 0): a workgroup keeps its 128
// f0 rows as MFMA A fragments in REGISTERS (16 k-groups x 4 VGPRs per wave) and walks PANEL_TILES
// consecutive column tiles; only the 64 x 512-B f1 tile streams through LDS, and the next tile's rows and
// column statistics are already in flight (registers) while the current tile is multiplied, exponentiated
// and stored - the global-load latency that a one-tile workgroup exposes 5 times is hidden behind a whole
// tile of work.  Same k order as sim_tile, so the sim values are bit-identical to pass A's.
// Units (sample, row panel, run of column tiles) are dealt so that each XCD walks a contiguous range.
// ---------------------------------------------------------------------------------------------
constexpr int PANEL_TILES = 10;
#ifndef K1_TRACE
#define K1_TRACE 0
#endif
#if K1_TRACE
__device__ long long k1_trace[512 * 4 * 32];
#define K1_STAMP(slot) do { if (lane == 0 && ui == slot0 + per_xcd && (slot) < 32) k1_trace[(blockIdx.x * 4 + wave) * 32 + (slot)] = clock64(); } while (0)
#endif
#if K1_TRACE == 1                  // -DK1_TRACE=1: phases of pass B (tools/k1_trace.py);  -DK1_TRACE=2: phases of pass A (tools/k1_trace.py stats)
#define K1_T(slot) K1_STAMP(slot)
#else
#define K1_T(slot)
#endif
#if K1_TRACE == 2
#define K1_TS(slot) K1_STAMP(slot)
#else
#define K1_TS(slot)
#endif
constexpr int PANEL_LDS = BN * 512 + BM * 8;      // f1 tile [64][512 B] + row statistics [128] float2

__device__ __forceinline__ int k1p_off(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

template <typename H, bool DENSE>
__global__ __launch_bounds__(NT, 2) void k1_conf_panel(K1Args a) {
    using V8 = gf_vec<H, 8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* rst = reinterpret_cast<float2*>(smem + BN * 512);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    const int runs = (a.tilesN + PANEL_TILES - 1) / PANEL_TILES;
    const int units = a.N * a.tilesM * runs;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (nwg + 7 - xcd) >> 3;
    const int q = units >> 3, rem = units & 7;
    const int ubeg = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q, ucnt = q + (xcd < rem ? 1 : 0);
    const float k2 = 2.0f * a.mult * LOG2E;
    const int srow = tid >> 5, schunk = tid & 31;             // staging: 8 rows x 32 chunks per pass, 8 passes
    const int slot0 = slot;
    (void)slot0;
    for (int ui = slot; ui < ucnt; ui += per_xcd) {
        const int u = ubeg + ui;
        K1_T(0);
        const int run = u % runs, pm = u / runs, bm = pm % a.tilesM, n = pm / a.tilesM;
        const int m0 = bm * BM, t0 = run * PANEL_TILES, t1 = min(t0 + PANEL_TILES, a.tilesN);
        const H* A = (const H*)a.f0 + ((size_t)n * a.L + m0 + wave * 32 + lr) * a.C + h * 8;
        const H* B = (const H*)a.f1 + (size_t)n * a.S * a.C;
        V8 af[16];
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) af[kg] = *reinterpret_cast<const V8*>(A + kg * 16);
        __syncthreads();                                      // previous unit's rst readers are done
        if (tid < BM) {
            const float2 st = a.rstat[(size_t)n * a.L + m0 + tid];
            rst[tid] = make_float2(-st.x * LOG2E, __builtin_amdgcn_rcpf(st.y));
        }
        v4u rb[8];
        float2 cs[2];
        auto prefetch = [&](int bn) {
            const H* g = B + (size_t)(bn * BN + srow) * a.C + schunk * 8;
#pragma unroll
            for (int p = 0; p < 8; ++p) rb[p] = *reinterpret_cast<const v4u*>(g + (size_t)p * 8 * a.C);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) cs[ni] = a.cstat[(size_t)n * a.S + bn * BN + ni * 32 + lr];
        };
        prefetch(t0);
        K1_T(1);
        const int row_base = m0 + __builtin_amdgcn_readfirstlane(wave) * 32;
        unsigned long long* rbest = a.rowbest + (size_t)n * a.L;
        unsigned* cmax = a.colmax + (size_t)n * a.S;
        const int lane_off = 4 * h * a.S + lr;
        // DENSE (thr ~ 0, every element is a candidate): the best (value, column) of each of the lane's 16 rows is
        // carried in registers over the whole run of tiles and reduced across lanes ONCE per run (a strict > keeps
        // the earliest column on ties: tiles, and ni within a tile, come in increasing column order)
        unsigned runv[DENSE ? 16 : 1], runc[DENSE ? 16 : 1];
        if constexpr (DENSE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { runv[r] = 0u; runc[r] = 0u; }
        }
        for (int bn = t0; bn < t1; ++bn) {
            __syncthreads();                                  // the previous tile's fragments are consumed
#pragma unroll
            for (int p = 0; p < 8; ++p) *reinterpret_cast<v4u*>(smem + k1p_off(srow + 8 * p, schunk)) = rb[p];
            float ca[2], cb[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                ca[ni] = -cs[ni].x * LOG2E;
                cb[ni] = __builtin_amdgcn_rcpf(cs[ni].y);
            }
            __syncthreads();
            K1_T(2 + 4 * (bn - t0));
            if (bn + 1 < t1) prefetch(bn + 1);                // in flight during this tile's MFMAs and epilogue
            v16f acc[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll
            for (int kg = 0; kg < 16; ++kg) {
                const V8 b0 = *reinterpret_cast<const V8*>(smem + k1p_off(lr, 2 * kg + h));
                const V8 b1 = *reinterpret_cast<const V8*>(smem + k1p_off(32 + lr, 2 * kg + h));
                Mma32<H>::mma(af[kg], b0, acc[0]);
                Mma32<H>::mma(af[kg], b1, acc[1]);
            }
            const int n0 = bn * BN;
#if K1_TRACE
            asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[1][15]));      // the stamp below waits for the MFMAs
#endif
            K1_T(3 + 4 * (bn - t0));
            float* cbase = a.conf + ((size_t)n * a.L + row_base) * a.S + n0;
            unsigned cbest[2] = {0u, 0u};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 st = rst[wave * 32 + gf_acc_row(r, h)];
                float cf[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    cf[ni] = __builtin_amdgcn_exp2f(fmaf(acc[ni][r], k2, st.x + ca[ni])) * (st.y * cb[ni]);
                // one v_permlane32_swap puts a row's 64 columns on the 64 lanes: each store instruction then writes
                // 256 contiguous bytes of ONE row (instead of 128 B of two rows 4 apart)
                float* rowp = cbase + (size_t)((r & 3) + 8 * (r >> 2)) * a.S;
                const gf_v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cf[0]), __float_as_uint(cf[1]), false, false);
                __builtin_nontemporal_store(__uint_as_float(sw.x), rowp + lane);
                __builtin_nontemporal_store(__uint_as_float(sw.y), rowp + 4 * a.S + lane);
                if constexpr (!DENSE) {
                    if (fmaxf(cf[0], cf[1]) > a.thr) {
                        const int row = row_base + gf_acc_row(r, h);
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            if (cf[ni] > a.thr) {
                                const int col = n0 + ni * 32 + lr;
                                const unsigned bits = __float_as_uint(cf[ni]);
                                atomicMax(rbest + row, ((unsigned long long)bits << 32) | (0xFFFFFFFFu - (unsigned)col));
                                atomicMax(cmax + col, bits);
                            }
                    }
                } else {
                    const unsigned b0 = cf[0] > a.thr ? __float_as_uint(cf[0]) : 0u;
                    const unsigned b1 = cf[1] > a.thr ? __float_as_uint(cf[1]) : 0u;
                    cbest[0] = max(cbest[0], b0);
                    cbest[1] = max(cbest[1], b1);
                    const bool second = b1 > b0;                      // ni = 1 is the later column: only a strict win
                    const unsigned bv = second ? b1 : b0, bc = (unsigned)(n0 + lr) + (second ? 32u : 0u);
                    const bool better = bv > runv[r];
                    runv[r] = better ? bv : runv[r];
                    runc[r] = better ? bc : runc[r];
                }
            }
            if constexpr (DENSE) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const unsigned c = max(cbest[ni], (unsigned)__shfl_xor((int)cbest[ni], 32, 64));
                    if (h == 0 && c != 0u) atomicMax(cmax + n0 + ni * 32 + lr, c);
                }
            }
            K1_T(4 + 4 * (bn - t0));
        }
        if constexpr (DENSE) {
            unsigned long long key[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                key[r] = runv[r] ? (((unsigned long long)runv[r] << 32) | (0xFFFFFFFFu - runc[r])) : 0ull;
                key[16 + r] = 0ull;
            }
            const unsigned long long kbest = k1_row_reduce(key, GfMaxU64());
            if (lr < 16 && kbest != 0ull) atomicMax(rbest + row_base + gf_acc_row(lr, h), kbest);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass B, pipelined panel form (the default): as k1_conf_panel, but
//   * the f1 tiles arrive by LDS-DMA (global_load_lds, swizzle applied on the source side) into TWO LDS buffers - no
//     staging registers, no ds_write pass, one barrier per tile;
//   * the run's column statistics sit in LDS (read once per unit);
//   * the 32 MFMAs of tile t+1 are interleaved with the exponentials / stores of tile t (two accumulator sets): the
//     store stream of a workgroup no longer pauses for its own K loop.
// The DMA of tile t+1 is waited for with a counted vmcnt: the 32 row-segment stores issued behind it stay in flight.
// Same k order as sim_tile: bit-identical sim values.  Both softmax normalisations are folded into the exponent.
// Measured at 8 pairs (tools/k1_time.py): 385 -> 336 us sparse candidates, 406 -> 385 us dense (bench: 343 -> 334 us).
// What bounds it now is NOT the store stream: with the conf stores compiled out the kernel still takes 308 us (sparse) /
// 341 us (dense), without the exponentials 323 / 353 us - it is paced by the instruction issue of its epilogue (add, fma,
// exp, lane swap, candidate bookkeeping per element) sharing the SIMDs with the 32 MFMAs per tile, at two waves per SIMD.
// ---------------------------------------------------------------------------------------------
constexpr int PIPE_LDS = 2 * BN * 512 + BM * 8 + PANEL_TILES * BN * 8;
constexpr int STATS_LDS = 2 * BN * 512 + 2 * 4 * 64 * 8 + 4 * 32 * 4;         // k1_stats_panel: two tile buffers | column partials | row maxima
struct K1Rsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ K1Rsrc k1_rsrc(const void* p, unsigned bytes) {
    return K1Rsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ void k1_lds_dma(const K1Rsrc& rs, char* dst, int voffset, int soffset) {        // 64 lanes x 16 B -> 1 KiB at dst
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}

template <typename H, bool DENSE, bool STORE>
__global__ __launch_bounds__(NT, 2) void k1_conf_pipe(K1Args a) {
    using V8 = gf_vec<H, 8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* rst = reinterpret_cast<float*>(smem + 2 * BN * 512);              // per row:    -max log2(e) - log2(sum)
    float* cst = reinterpret_cast<float*>(smem + 2 * BN * 512 + BM * 8);     // per column of the run, likewise
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, lr = lane & 31;
    const int runs = (a.tilesN + PANEL_TILES - 1) / PANEL_TILES;
    const int units = a.N * a.tilesM * runs;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (nwg + 7 - xcd) >> 3;
    const int q = units >> 3, rem = units & 7;
    const int ubeg = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q, ucnt = q + (xcd < rem ? 1 : 0);
    const float k2 = 2.0f * a.mult * LOG2E;
    for (int ui = slot; ui < ucnt; ui += per_xcd) {
        const int u = ubeg + ui;
        const int run = u % runs, pm = u / runs, bm = pm % a.tilesM, n = pm / a.tilesM;
        const int m0 = bm * BM, t0 = run * PANEL_TILES, t1 = min(t0 + PANEL_TILES, a.tilesN);
        const H* A = (const H*)a.f0 + ((size_t)n * a.L + m0 + wave * 32 + lr) * a.C + h * 8;
        const char* B = (const char*)((const H*)a.f1 + (size_t)n * a.S * a.C);
        V8 af[16];
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) af[kg] = *reinterpret_cast<const V8*>(A + kg * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (also drains the previous unit's stores: the counted waits below start clean)
        __syncthreads();                                      // previous unit's readers of rst / cst / the tile buffers are done
        // tile bn -> buffer `buf`: 32 pieces of 2 rows x 512 B, 8 per wave; LDS slot j of row r holds chunk j ^ (r & 15)
        // LDS-DMA as buffer_load_dwordx4 ... lds (MUBUF): behind the FLAT form (global_load_lds) the compiler's wait insertion turns
        // every LDS counter wait into lgkmcnt(0) while a request is pending; scalar descriptor + 32-bit offsets besides
        const K1Rsrc brs = k1_rsrc(B, (unsigned)a.S * a.C * (unsigned)sizeof(H));
        auto dma = [&](int bn, int buf) {
            int dl = lane;
            asm volatile("" : "+v"(dl));                     // per-piece source offsets recomputed here, not kept across the tile loop
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = wave * 8 + i, row = 2 * p + (dl >> 5), j = dl & 31;
                k1_lds_dma(brs, smem + buf * (BN * 512) + p * 1024, (row * a.C + ((j ^ (row & 15)) << 3)) * (int)sizeof(H), bn * BN * a.C * (int)sizeof(H));
            }
        };
        dma(t0, 0);
        if (tid < BM) {
            const float2 st = a.rstat[(size_t)n * a.L + m0 + tid];
            rst[tid] = -st.x * LOG2E - __builtin_amdgcn_logf(st.y);       // v_log_f32 = log2: exp2(.. + rst) = e^{-max} / sum
        }
        for (int i = tid; i < (t1 - t0) * BN; i += NT) {
            const float2 cs = a.cstat[(size_t)n * a.S + t0 * BN + i];
            cst[i] = -cs.x * LOG2E - __builtin_amdgcn_logf(cs.y);
        }
        const int row_base = m0 + wave * 32;
        unsigned long long* rbest = a.rowbest + (size_t)n * a.L;
        unsigned* cmax = a.colmax + (size_t)n * a.S;
        unsigned runv[DENSE ? 16 : 1], runc[DENSE ? 16 : 1];
        if constexpr (DENSE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { runv[r] = 0u; runc[r] = 0u; }
        }
        // one row (register r of both column halves) of the epilogue of tile `bn`
        // conf = exp2(2 sim log2e + rst[row] + cst[col]): both softmax normalisations folded into the exponent (one add,
        // one fma and one exponential per element; the two products with 1/sum are gone)
        float ca[2];
        unsigned cbest[2];
#ifndef K1_CG
#define K1_CG 4
#endif
        constexpr int CG = K1_CG;                             // rows per candidate test
        [[maybe_unused]] float hold[CG][2], gmax = 0.f;
        const unsigned thr_bits = __float_as_uint(fmaxf(a.thr, 0.f));
        auto epi_begin = [&](int bn) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                ca[ni] = cst[(bn - t0) * BN + ni * 32 + lr];
                cbest[ni] = 0u;
            }
        };
        auto epi_row = [&](const v16f (&acc)[2], int bn, int r) {
            const int n0 = bn * BN;
            const float st = rst[wave * 32 + gf_acc_row(r, h)];
            float cf[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) cf[ni] = __builtin_amdgcn_exp2f(fmaf(acc[ni][r], k2, st + ca[ni]));
            if constexpr (STORE) {                            // (match-only mode, conf == NULL: the matrix is never written)
                float* rowp = a.conf + ((size_t)n * a.L + row_base + (r & 3) + 8 * (r >> 2)) * a.S + n0;
                const gf_v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cf[0]), __float_as_uint(cf[1]), false, false);
                __builtin_nontemporal_store(__uint_as_float(sw.x), rowp + lane);
                __builtin_nontemporal_store(__uint_as_float(sw.y), rowp + 4 * a.S + lane);
            }
            if constexpr (!DENSE) {
                // candidates (conf > thr) are rare: the test runs once per CG rows on their maximum (one max3 per row; a
                // compare + branch per row cut the MFMA loop into 16 blocks), the rows are looked at one by one only on a hit
                hold[r % CG][0] = cf[0];
                hold[r % CG][1] = cf[1];
                gmax = r % CG == 0 ? fmaxf(cf[0], cf[1]) : fmaxf(gmax, fmaxf(cf[0], cf[1]));
                if (r % CG == CG - 1 && gmax > a.thr) {
#pragma unroll
                    for (int q = 0; q < CG; ++q) {
                        const int row = row_base + gf_acc_row(r - (CG - 1) + q, h);
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            if (hold[q][ni] > a.thr) {
                                const int col = n0 + ni * 32 + lr;
                                const unsigned bits = __float_as_uint(hold[q][ni]);
                                atomicMax(rbest + row, ((unsigned long long)bits << 32) | (0xFFFFFFFFu - (unsigned)col));
                                atomicMax(cmax + col, bits);
                            }
                    }
                }
            } else {
                // (conf >= 0: its bits order like its value; the threshold is applied once, to the maxima)
                const unsigned b0 = __float_as_uint(cf[0]), b1 = __float_as_uint(cf[1]);
                cbest[0] = max(cbest[0], b0);
                cbest[1] = max(cbest[1], b1);
                const bool second = b1 > b0;                          // ni = 1 is the later column: only a strict win
                const unsigned bv = second ? b1 : b0, bc = (unsigned)(n0 + lr) + (second ? 32u : 0u);
                const bool better = bv > runv[r];
                runv[r] = better ? bv : runv[r];
                runc[r] = better ? bc : runc[r];
            }
        };
        auto epi_end = [&](int bn) {
            if constexpr (DENSE) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const unsigned c = max(cbest[ni], (unsigned)__shfl_xor((int)cbest[ni], 32, 64));
                    if (h == 0 && c > thr_bits) atomicMax(cmax + bn * BN + ni * 32 + lr, c);
                }
            }
        };
        v16f prev[2], cur[2];
        for (int bn = t0; bn <= t1; ++bn) {
            const bool mma = bn < t1, epi = bn > t0;
            if (mma) {
                // tile bn has landed: what was issued behind its DMA - the previous iteration's 32 row stores (and possibly
                // candidate atomics) - may stay in flight.  The unit's first two tiles have no stores behind their DMA (tile
                // t0 + 1 is requested in the iteration that only multiplies tile t0): they wait for everything.
                if (bn <= t0 + 1 || !STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
            __syncthreads();                                  // ... for every wave; and every wave is done with tile bn-1's buffer
            if (bn + 1 < t1) dma(bn + 1, (bn + 1 - t0) & 1);
            const char* tb = smem + ((bn - t0) & 1) * (BN * 512);
            if (epi) epi_begin(bn - 1);
            if (mma) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[ni][r] = 0.f;
            }
            if (mma && epi) {
#pragma unroll
                for (int kg = 0; kg < 16; ++kg) {
                    const V8 b0 = *reinterpret_cast<const V8*>(tb + k1p_off(lr, 2 * kg + h));
                    const V8 b1 = *reinterpret_cast<const V8*>(tb + k1p_off(32 + lr, 2 * kg + h));
                    Mma32<H>::mma(af[kg], b0, cur[0]);
                    Mma32<H>::mma(af[kg], b1, cur[1]);
                    epi_row(prev, bn - 1, kg);
                }
            } else if (mma) {
#pragma unroll
                for (int kg = 0; kg < 16; ++kg) {
                    const V8 b0 = *reinterpret_cast<const V8*>(tb + k1p_off(lr, 2 * kg + h));
                    const V8 b1 = *reinterpret_cast<const V8*>(tb + k1p_off(32 + lr, 2 * kg + h));
                    Mma32<H>::mma(af[kg], b0, cur[0]);
                    Mma32<H>::mma(af[kg], b1, cur[1]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) epi_row(prev, bn - 1, r);
            }
            if (epi) epi_end(bn - 1);
            if (mma) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) prev[ni] = cur[ni];
            }
        }
        if constexpr (DENSE) {
            unsigned long long key[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                key[r] = runv[r] > thr_bits ? (((unsigned long long)runv[r] << 32) | (0xFFFFFFFFu - runc[r])) : 0ull;
                key[16 + r] = 0ull;
            }
            const unsigned long long kbest = k1_row_reduce(key, GfMaxU64());
            if (lr < 16 && kbest != 0ull) atomicMax(rbest + row_base + gf_acc_row(lr, h), kbest);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pass A in the same row-panel-persistent form: the f0 panel lives in registers, f1 tiles stream through LDS one tile
// ahead, and - the point - the ROW statistics stay in the lane that owns the row slot for the whole run of tiles
// (online max / rescaled sum per slot, 3 exponentials per slot and tile) and cross the lanes once per run instead
// of two 32-lane reduce-scatters per tile.  Column statistics are lane-local per tile as before.
// ---------------------------------------------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(NT, 2) void k1_stats_panel(K1Args a) {
    using V8 = gf_vec<H, 8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: two f1 tile buffers (LDS-DMA, one tile ahead) | column partials of two tiles [2][4 waves][64] | row maxima of the run
    float2* colx = reinterpret_cast<float2*>(smem + 2 * BN * 512);
    float* rowbc = reinterpret_cast<float*>(smem + 2 * BN * 512 + 2 * 4 * 64 * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, lr = lane & 31;   // (wave as an SGPR:
    // the DMA pieces' LDS targets go through M0 - from a VGPR they were hoisted, spilled and reloaded with a vmcnt(0) between the requests)
    const int runs = (a.tilesN + PANEL_TILES - 1) / PANEL_TILES;
    const int units = a.N * a.tilesM * runs;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (nwg + 7 - xcd) >> 3;
    const int q8 = units >> 3, rem = units & 7;
    const int ubeg = xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8, ucnt = q8 + (xcd < rem ? 1 : 0);
    const int slot0 = slot;
    (void)slot0;
    for (int ui = slot; ui < ucnt; ui += per_xcd) {
        K1_TS(0);
        const int u = ubeg + ui;
        const int run = u % runs, pm = u / runs, bm = pm % a.tilesM, n = pm / a.tilesM;
        const int m0 = bm * BM, t0 = run * PANEL_TILES, t1 = min(t0 + PANEL_TILES, a.tilesN);
        const H* A = (const H*)a.f0 + ((size_t)n * a.L + m0 + wave * 32 + lr) * a.C + h * 8;
        const H* B = (const H*)a.f1 + (size_t)n * a.S * a.C;
        V8 af[16];
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) af[kg] = *reinterpret_cast<const V8*>(A + kg * 16);
        // tile bn -> buffer `buf` by LDS-DMA (as k1_conf_pipe: 32 pieces of 2 rows x 512 B, 8 per wave; LDS slot j of row r holds
        // chunk j ^ (r & 15)): no staging registers, no ds_write pass
        const K1Rsrc brs = k1_rsrc(B, (unsigned)a.S * a.C * (unsigned)sizeof(H));
        auto dma = [&](int bn, int buf) {
            int dl = lane;
            asm volatile("" : "+v"(dl));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = wave * 8 + i, row = 2 * p + (dl >> 5), j = dl & 31;
                k1_lds_dma(brs, smem + buf * (BN * 512) + p * 1024, (row * a.C + ((j ^ (row & 15)) << 3)) * (int)sizeof(H), bn * BN * a.C * (int)sizeof(H));
            }
        };
        // the column partials of tile bn - 1 (parked in LDS by the four waves) are combined behind tile bn's barrier, by wave bn % 4:
        // no barrier of their own
        auto combine = [&](int bn, int par) {
            const float2* cx = colx + par * 256;
            float m = NEG_INF;
#pragma unroll
            for (int w = 0; w < 4; ++w) m = fmaxf(m, cx[w * 64 + lane].x);
            float l = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) l += cx[w * 64 + lane].y * __expf(cx[w * 64 + lane].x - m);
            a.colpart[((size_t)n * a.tilesM + bm) * a.S + bn * BN + lane] = make_float2(m, l);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // the previous unit's readers of the buffers / colx / rowbc are done
        dma(t0, 0);
        // Per-element work of this pass = 1 exponential, not 2.5: everything is referred to ONE lazily updated reference per
        // row slot (a register) of this lane,
        //     e = exp2(s2 - ref[r]),  s2 = acc * mult * log2(e),
        // which feeds the row sum directly (rs[r] += e) and the column sum through a cached per-slot factor
        //     csum += e * f[r],  f[r] = exp2(ref[r] - kappa)   (kappa = the lane's largest ref: csum = sum exp2(s2 - kappa)).
        // ref[r] is a former running maximum of the slot, so the element that set it contributes 1 and whatever flushes to zero is
        // below 2^-126 of the sum; it is moved up (rs rescaled, f recomputed: 'rescale') whenever a tile's maximum exceeds the
        // smallest reference of the lane by more than 2^LAZY, which also bounds e by 2^LAZY.  The column partial is reported
        // against the column's TRUE maximum (one exponential per column and tile); a tile in which some column lies more than
        // 2^LAZY below kappa takes the two-exponential path for its column sums instead ('deep' tiles).  Maxima are taken on the
        // raw accumulators and scaled once: bit-identical to the form this replaces; the sums are exact up to fp32 rounding.
        constexpr float LAZY = 64.f;
        const float mult2 = a.mult * LOG2E;
        float rmx[16], rs[16], ref[16], fcol[16];                         // raw running maximum | sum | reference (log2) | exp2(ref - kappa)
#pragma unroll
        for (int r = 0; r < 16; ++r) { rmx[r] = NEG_INF; rs[r] = 0.f; ref[r] = NEG_INF; fcol[r] = 0.f; }
        float minref = NEG_INF, kappa = NEG_INF;
        // the tile body, instantiated for both buffer parities (the loop below walks the tiles in pairs): with the parity a compile-time
        // constant the 32 fragment reads of a tile are the same 16 lane-constant addresses + an immediate offset - as a run-time
        // term it doubled them, and the spilled ones were reloaded from scratch between the DMA requests
        auto tile = [&](auto par_c, const int bn) {
            constexpr int PAR = decltype(par_c)::value;
            const char* tb = smem + PAR * (BN * 512);
            K1_TS(1 + 6 * (bn - t0));
            // ONE barrier per tile: tile bn has landed (every wave waits for its own pieces; the only other vector-memory operations
            // in flight are 64 column-partial stores of one wave), every wave is past tile bn - 1 (its buffer and colx slot are free)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            K1_TS(2 + 6 * (bn - t0));
            if (bn + 1 < t1) dma(bn + 1, PAR ^ 1);
            if (bn > t0 && wave == ((bn - t0) & 3)) combine(bn - 1, PAR ^ 1);
            v16f acc[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
            // the f1 fragments of k-group kg + 1 are requested in front of the MFMAs of kg (two waves per SIMD do not hide an LDS
            // round trip in front of every MFMA pair: left alone the compiler reads each pair right where it is used)
            V8 b0 = *reinterpret_cast<const V8*>(tb + k1p_off(lr, h)), b1 = *reinterpret_cast<const V8*>(tb + k1p_off(32 + lr, h));
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // issue order: reads(0) | reads(1) MFMAs(0) | reads(2) MFMAs(1) | ...
#pragma unroll
            for (int kg = 0; kg < 16; ++kg) {
                V8 n0 = b0, n1 = b1;
                if (kg < 15) {
                    n0 = *reinterpret_cast<const V8*>(tb + k1p_off(lr, 2 * kg + 2 + h));
                    n1 = *reinterpret_cast<const V8*>(tb + k1p_off(32 + lr, 2 * kg + 2 + h));
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                Mma32<H>::mma(af[kg], b0, acc[0]);
                Mma32<H>::mma(af[kg], b1, acc[1]);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                b0 = n0;
                b1 = n1;
            }
            K1_TS(3 + 6 * (bn - t0));
            // ---- maxima on the raw accumulators: columns over the registers, row slots over the two column halves and the tiles
            float cmr0 = NEG_INF, cmr1 = NEG_INF;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                cmr0 = fmaxf(cmr0, acc[0][r]);
                cmr1 = fmaxf(cmr1, acc[1][r]);
                rmx[r] = fmaxf(rmx[r], fmaxf(acc[0][r], acc[1][r]));
            }
            const float lanemax2 = fmaxf(cmr0, cmr1) * mult2;
            if (__any(lanemax2 - minref > LAZY)) {                       // rescale (always taken by the run's first tile)
                float mn = INFINITY, mxr = NEG_INF;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float nr = rmx[r] * mult2;
                    rs[r] *= __builtin_amdgcn_exp2f(ref[r] - nr);        // 0 * exp2(-inf) = 0 on the first tile
                    ref[r] = nr;
                    mn = fminf(mn, nr);
                    mxr = fmaxf(mxr, nr);
                }
                minref = mn;
                kappa = mxr;
#pragma unroll
                for (int r = 0; r < 16; ++r) fcol[r] = __builtin_amdgcn_exp2f(ref[r] - kappa);
            }
            K1_TS(4 + 6 * (bn - t0));
            // column maxima over the wave's 32 rows (both lane halves), in log2 units
            const float cm0 = fmaxf(cmr0, __shfl_xor(cmr0, 32, 64)), cm1 = fmaxf(cmr1, __shfl_xor(cmr1, 32, 64));
            const float c20 = cm0 * mult2, c21 = cm1 * mult2;
            float cs0 = 0.f, cs1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(acc[0][r], mult2, -ref[r]));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(acc[1][r], mult2, -ref[r]));
                rs[r] += e0 + e1;
                cs0 = fmaf(e0, fcol[r], cs0);
                cs1 = fmaf(e1, fcol[r], cs1);
            }
            if (__any(kappa - fminf(c20, c21) > LAZY)) {                 // a 'deep' tile: column sums against their own maxima
                cs0 = 0.f; cs1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cs0 += __builtin_amdgcn_exp2f(fmaf(acc[0][r], mult2, -c20));
                    cs1 += __builtin_amdgcn_exp2f(fmaf(acc[1][r], mult2, -c21));
                }
            } else {
                cs0 *= __builtin_amdgcn_exp2f(kappa - c20);              // from reference kappa to the column's own maximum
                cs1 *= __builtin_amdgcn_exp2f(kappa - c21);
            }
            cs0 += __shfl_xor(cs0, 32, 64);
            cs1 += __shfl_xor(cs1, 32, 64);
            if (h == 0) {
                float2* cx = colx + PAR * 256;
                cx[wave * 64 + lr] = make_float2(cm0 * a.mult, cs0);
                cx[wave * 64 + 32 + lr] = make_float2(cm1 * a.mult, cs1);
            }
            K1_TS(5 + 6 * (bn - t0));
        };
        for (int bn = t0; bn < t1; bn += 2) {
            tile(std::integral_constant<int, 0>{}, bn);
            if (bn + 1 < t1) tile(std::integral_constant<int, 1>{}, bn + 1);
        }
        __syncthreads();                                      // the last tile's column partials are parked
        if (wave == ((t1 - t0) & 3)) combine(t1 - 1, (t1 - 1 - t0) & 1);
        // ---- end of the run: row maxima across the lanes, sums moved from the lane's reference to them, sums across the lanes
        float v[32];
#pragma unroll
        for (int q = 0; q < 16; ++q) { v[q] = rmx[q]; v[16 + q] = NEG_INF; }
        const float rmax_raw = k1_row_reduce(v, GfMaxF());
        if (lr < 16) rowbc[wave * 32 + h * 16 + lr] = rmax_raw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            v[q] = rs[q] * __builtin_amdgcn_exp2f(ref[q] - rowbc[wave * 32 + h * 16 + q] * mult2);
            v[16 + q] = 0.f;
        }
        const float rsum = k1_row_reduce(v, GfAddF());
        if (lr < 16) a.rowpart[((size_t)n * runs + run) * a.L + m0 + wave * 32 + gf_acc_row(lr, h)] = make_float2(rmax_raw * a.mult, rsum);
    }
}

#if K1_TRACE == 1 + K1_PART        // the stamps of pass B live in part 0's buffer, those of pass A (K1_TRACE 2) in part 1's
}
extern "C" int gf_debug_k1_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k1_trace), sizeof(long long) * 512 * 4 * 32);
}
namespace {
#endif

#if K1_PART == 1
template <typename T>
void k1_stats_panel_launch(const K1Args& a, int wgs, hipStream_t st) {
    static std::atomic<uint64_t> attr{0};                       // 68.5 KiB of dynamic LDS: opt in once per device
    if (gf_first_use_on_device(attr))
        (void)hipFuncSetAttribute((const void*)k1_stats_panel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, STATS_LDS);
    k1_stats_panel<T><<<wgs, NT, STATS_LDS, st>>>(a);
}
}   // namespace

void gf_k1_stats_panel_launch(const void* k1args, int dtype, int wgs, void* stream) {
    K1Args a;
    memcpy(&a, k1args, sizeof(a));
    if (dtype == GF_F16) k1_stats_panel_launch<_Float16>(a, wgs, (hipStream_t)stream);
    else k1_stats_panel_launch<gf_bf16>(a, wgs, (hipStream_t)stream);
}
#else

struct SelArgs {
    const unsigned long long* stamp;   // the workspace stamp (k1_conf_at checks it)
    int N, L, S;
    const unsigned long long* rowbest;   // [N][L]
    const unsigned* colmax;              // [N][S]
    unsigned* colset;                    // [N][SET_WORDS] bitset over hash(colmax value)
    const float* conf;     // null in match-only mode: single entries are recomputed (k1_conf_entry)
    const void* f0;        // (match-only mode) the operands and statistics of the sweep
    const void* f1;
    const float2* rstat;
    const float2* cstat;
    int C;
    float mult;
    int* scanlist;         // [N*L] rows whose first-True column needs the tie rescan
    int* scancnt;          // [1]
    int* selj;             // [N][L]  matched column or -1
    int* samplecnt;        // [N][chunks]  matches per 1024-row chunk
    int chunks;
    int force_one;
    int w0c, w1c;
    float scale;
    const float* scale0;
    const float* scale1;
    int64_t* b_ids;
    int64_t* i_ids;
    int64_t* j_ids;
    float* mconf;
    float* mk0;
    float* mk1;
    int32_t* counts;
};

constexpr int SET_BITS = 1 << 20;                // per-sample bitset of column-maximum values
constexpr int SET_WORDS = SET_BITS / 32;
__device__ __forceinline__ unsigned set_hash(unsigned bits) {
    unsigned x = bits;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x & (SET_BITS - 1);
}

// membership filter: which fp32 values occur as the maximum of SOME column
__global__ void k1_colset(SelArgs a) {
    const int n = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.S) return;
    const unsigned v = a.colmax[(size_t)n * a.S + j];
    if (v == 0u) return;
    const unsigned h = set_hash(v);
    atomicOr(a.colset + (size_t)n * SET_WORDS + (h >> 5), 1u << (h & 31));
}

// one thread per (n, i): coarse_matching.py:161-185 on the candidate statistics instead of the matrix.
// rowbest holds the row maximum only if it exceeds thr (otherwise 0: no candidate, no match).
__global__ void k1_select(SelArgs a) {
    const int n = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.L) return;
    const unsigned long long k = a.rowbest[(size_t)n * a.L + i];
    int sel = -1;
    if (k != 0ull) {
        const unsigned bits = (unsigned)(k >> 32);
        const int j = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        const unsigned* cm = a.colmax + (size_t)n * a.S;
        if (cm[j] == bits) {
            sel = j;
        } else {
            // exact `mask.max(dim=2)` semantics: a LATER column may tie the row maximum AND be its own
            // column's maximum.  That needs some column whose maximum equals this row's maximum bit for
            // bit; the value filter rules that out for almost every row, the rest is queued for k1_rescan.
            const unsigned h = set_hash(bits);
            if ((a.colset[(size_t)n * SET_WORDS + (h >> 5)] >> (h & 31)) & 1u)
                a.scanlist[atomicAdd(a.scancnt, 1)] = n * a.L + i;
        }
    }
    a.selj[(size_t)n * a.L + i] = sel;
    // per-1024-row chunk counts (blockDim = 256: one ballot + one atomic per wave)
    const unsigned long long bal = __ballot(sel >= 0);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&a.samplecnt[n * a.chunks + (i >> 10)], __popcll(bal));
}

// One entry conf[n][i][j] exactly as k1_conf_pipe produces it (16-bit storage, the panel configuration), computed by one wave:
// the 32 x 32 similarity tile that holds (i, j) is multiplied with the same operands in the same k order - an MFMA result
// depends on nothing else - and pushed through the same epilogue arithmetic.  Match-only mode (conf == NULL) uses it where the
// contract mode reads the matrix: the tie rescan, the forced match, and gf_dual_softmax_conf_at.  Returned in every lane.
template <typename H>
__device__ __forceinline__ float k1_conf_entry(const SelArgs& a, int n, int i, int j) {
    using V8 = gf_vec<H, 8>;
    const int lane = threadIdx.x & 63, h = lane >> 5, lr = lane & 31;
    const int i0 = i & ~31, j0 = j & ~31;
    const H* A = (const H*)a.f0 + ((size_t)n * a.L + min(i0 + lr, a.L - 1)) * a.C + h * 8;
    const H* B = (const H*)a.f1 + ((size_t)n * a.S + min(j0 + lr, a.S - 1)) * a.C + h * 8;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kg = 0; kg < a.C / 16; ++kg)
        Mma32<H>::mma(*reinterpret_cast<const V8*>(A + kg * 16), *reinterpret_cast<const V8*>(B + kg * 16), acc);
    // accumulator row ri = (r & 3) + 8 (r >> 2) + 4 h of column lane lr
    const int ri = i - i0, rsel = (ri & 3) + 4 * (ri >> 3), lsel = (j - j0) + 32 * ((ri >> 2) & 1);
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v = r == rsel ? acc[r] : v;
    v = __shfl(v, lsel, 64);
    const float2 rs = a.rstat[(size_t)n * a.L + i], cs = a.cstat[(size_t)n * a.S + j];
    const float st = -rs.x * LOG2E - __builtin_amdgcn_logf(rs.y), ca = -cs.x * LOG2E - __builtin_amdgcn_logf(cs.y);
    return __builtin_amdgcn_exp2f(fmaf(v, 2.0f * a.mult * LOG2E, st + ca));
}

template <typename H>
__global__ __launch_bounds__(256) void k1_conf_at(SelArgs a, const int64_t* b, const int64_t* i, const int64_t* j, int P, float* out) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= P) return;
    // the statistics in the workspace must be those of THESE features at THIS shape, and the entry must exist: NaN otherwise
    bool ok = true;
#pragma unroll
    for (int k = 0; k < K1_STAMP_WORDS; ++k) ok = ok && a.stamp[k] == k1_stamp_word(k, a.f0, a.f1, a.N, a.L, a.S, a.C, a.mult);
    ok = ok && a.stamp[K1_STAMP_WORDS] == k1_fingerprint(a.f0, a.f1, (size_t)a.N * a.L * a.C * sizeof(H), (size_t)a.N * a.S * a.C * sizeof(H),
                                                         threadIdx.x & 63);
    const long long bb = b[e], ii = i[e], jj = j[e];
    ok = ok && bb >= 0 && bb < a.N && ii >= 0 && ii < a.L && jj >= 0 && jj < a.S;
    if (!ok) {                                                              // wave-uniform: e is per wave
        if ((threadIdx.x & 63) == 0) out[e] = __builtin_nanf("");
        return;
    }
    const float c = k1_conf_entry<H>(a, (int)bb, (int)ii, (int)jj);
    if ((threadIdx.x & 63) == 0) out[e] = c;
}

// rows queued by k1_select: one wave per row walks the (L2-resident) column maxima for later columns
// that hold the same value and confirms the tie on the confidence matrix itself
template <typename H, bool MATCH_ONLY>
__global__ __launch_bounds__(256) void k1_rescan(SelArgs a) {
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4, total = *a.scancnt;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < total; e += nwaves) {
        const int row = a.scanlist[e], n = row / a.L, i = row % a.L;
        const unsigned long long k = a.rowbest[row];
        const unsigned bits = (unsigned)(k >> 32);
        const int j = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        const unsigned* cm = a.colmax + (size_t)n * a.S;
        int sel = -1;
        for (int j0 = ((j + 1) / 64) * 64; j0 < a.S && sel < 0; j0 += 64) {
            const int j2 = j0 + lane;
            if constexpr (!MATCH_ONLY) {
                const float* crow = a.conf + (size_t)row * a.S;
                const bool hit = j2 > j && j2 < a.S && cm[j2] == bits && __float_as_uint(crow[j2]) == bits;
                const unsigned long long bal = __ballot(hit);
                if (bal) sel = j0 + __ffsll((long long)bal) - 1;
            } else {
                // candidates by the column maxima alone, then (in column order) the entry itself, recomputed
                unsigned long long bal = __ballot(j2 > j && j2 < a.S && cm[j2] == bits);
                while (bal && sel < 0) {
                    const int jc = j0 + __ffsll((long long)bal) - 1;
                    bal &= bal - 1;
                    if (__float_as_uint(k1_conf_entry<H>(a, n, i, jc)) == bits) sel = jc;
                }
            }
        }
        if (lane == 0 && sel >= 0) {
            a.selj[row] = sel;
            atomicAdd(&a.samplecnt[n * a.chunks + (i >> 10)], 1);
        }
    }
}

// one workgroup per (1024-row chunk, sample): ordered compaction in (n, i) order + keypoints
// (coarse_matching.py:186-201).  Bases come from the chunk counts of everything in front.
template <typename H, bool MATCH_ONLY>
__global__ __launch_bounds__(1024) void k1_compact(SelArgs a) {
    __shared__ int wave_tot[16];
    __shared__ int sh_base, sh_forced;
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int base = 0, forced = 0;
        for (int b = 0; b <= n; ++b) {
            int tot = 0, upto = 0;
            for (int k = 0; k < a.chunks; ++k) {
                const int v = a.samplecnt[b * a.chunks + k];
                tot += v;
                if (k < c) upto += v;
            }
            const bool f = (tot == 0 && a.force_one);
            if (b < n) base += f ? 1 : tot;
            else {
                base += upto;
                forced = f;
                if (c == 0) a.counts[1 + n] = f ? 1 : tot;
                if (c == 0 && n == a.N - 1) a.counts[0] = base + (f ? 1 : tot);
            }
        }
        sh_base = base;
        sh_forced = forced;
    }
    __syncthreads();
    const int base = sh_base;
    const float s0x = a.scale0 ? a.scale * a.scale0[2 * n] : a.scale, s0y = a.scale0 ? a.scale * a.scale0[2 * n + 1] : a.scale;
    const float s1x = a.scale1 ? a.scale * a.scale1[2 * n] : a.scale, s1y = a.scale1 ? a.scale * a.scale1[2 * n + 1] : a.scale;
    auto emit = [&](int pos, int i, int j) {
        a.b_ids[pos] = n;
        a.i_ids[pos] = i;
        a.j_ids[pos] = j;
        // match-only: a selected entry IS its row's best candidate value (k1_select / k1_rescan compared the bits)
        if constexpr (MATCH_ONLY) a.mconf[pos] = __uint_as_float((unsigned)(a.rowbest[(size_t)n * a.L + i] >> 32));
        else a.mconf[pos] = a.conf[((size_t)n * a.L + i) * a.S + j];
        a.mk0[2 * pos] = (float)(i % a.w0c) * s0x;
        a.mk0[2 * pos + 1] = (float)(i / a.w0c) * s0y;
        a.mk1[2 * pos] = (float)(j % a.w1c) * s1x;
        a.mk1[2 * pos + 1] = (float)(j / a.w1c) * s1y;
    };
    if (sh_forced) {
        if (tid == 0 && c == 0) emit(base, 0, 0);
        return;
    }
    const int i = c * 1024 + tid;
    const int j = (i < a.L) ? a.selj[(size_t)n * a.L + i] : -1;
    const bool f = j >= 0;
    const unsigned long long bal = __ballot(f);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wave) ? wave_tot[w] : 0;
    if (f) emit(base + woff + before, i, j);
}

template <typename T>
int k1_launch(K1Args a, SelArgs s, void* zero_begin, size_t zero_bytes, hipStream_t st) {
    constexpr bool EXACT = std::is_same<T, float>::value;
    const dim3 grid(a.tilesN * a.tilesM, a.N);
    (void)hipMemsetAsync(zero_begin, 0, zero_bytes, st);   // rowbest, colmax, samplecnt (contiguous)
    const bool panel = !EXACT && a.mask0 == nullptr && a.L % BM == 0 && a.S % BN == 0 && a.C == 256;
    const bool match_only = a.conf == nullptr;                 // (the entry point admits it in the panel configuration only)
    const int runs = (a.tilesN + PANEL_TILES - 1) / PANEL_TILES;
    const int units = a.N * a.tilesM * runs, wgs = units < 512 ? units : 512;    // two resident workgroups per CU
    a.rowparts = panel ? runs : a.tilesN;
    // the whole call (statistics, reduction, confidence sweep, selection, compaction) against its algorithmic bytes
    void* pu = gf_prof_begin("k1_unit", st, (double)a.N * ((double)(a.L + a.S) * a.C * sizeof(T) + (match_only ? 0.0 : (double)a.L * a.S * 4.0)));
    void* p0 = gf_prof_begin("k1_stats", st, 2.0 * a.N * (double)a.L * a.S * a.C);
    if constexpr (!EXACT) {
        if (panel) {
            gf_k1_stats_panel_launch(&a, std::is_same<T, _Float16>::value ? GF_F16 : GF_BF16, wgs, st);
        }
        else k1_stats<T><<<grid, NT, STAGE_BYTES, st>>>(a);
    } else {
        k1_stats<T><<<grid, NT, STAGE_BYTES, st>>>(a);
    }
    gf_prof_end("k1_stats", p0, st);
    const int mx = a.L > a.S ? a.L : a.S;
    k1_reduce_stats<EXACT><<<dim3((mx + 31) / 32, 2, a.N), 256, 0, st>>>(a);
    void* p1 = gf_prof_begin(match_only ? "k1_conf_matchonly" : "k1_conf", st,
                             (double)a.N * ((double)(a.L + a.S) * a.C * sizeof(T) + (match_only ? 0.0 : (double)a.L * a.S * 4.0)));
    bool done = false;
    if constexpr (!EXACT) {
        if (panel) {
            // GF_K1_CONF=panel selects the unpipelined panel form (A/B measurements, tools/k1_trace.py)
            static const bool old_form = [] { const char* e = getenv("GF_K1_CONF"); return e && e[0] == 'p'; }();
            static std::atomic<uint64_t> attr{0};
            if (gf_first_use_on_device(attr)) {
                (void)hipFuncSetAttribute((const void*)k1_conf_pipe<T, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
                (void)hipFuncSetAttribute((const void*)k1_conf_pipe<T, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
                (void)hipFuncSetAttribute((const void*)k1_conf_pipe<T, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
                (void)hipFuncSetAttribute((const void*)k1_conf_pipe<T, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS);
            }
            if (match_only) {
                if (a.dense) k1_conf_pipe<T, true, false><<<wgs, NT, PIPE_LDS, st>>>(a);
                else k1_conf_pipe<T, false, false><<<wgs, NT, PIPE_LDS, st>>>(a);
            } else if (old_form) {
                if (a.dense) k1_conf_panel<T, true><<<wgs, NT, PANEL_LDS, st>>>(a);
                else k1_conf_panel<T, false><<<wgs, NT, PANEL_LDS, st>>>(a);
            } else if (a.dense) k1_conf_pipe<T, true, true><<<wgs, NT, PIPE_LDS, st>>>(a);
            else k1_conf_pipe<T, false, true><<<wgs, NT, PIPE_LDS, st>>>(a);
            done = true;
        }
    }
    if (done) {
    } else if (a.dense) k1_conf<T, true><<<grid, NT, STAGE_BYTES, st>>>(a);
    else k1_conf<T, false><<<grid, NT, STAGE_BYTES, st>>>(a);
    gf_prof_end(match_only ? "k1_conf_matchonly" : "k1_conf", p1, st);
    k1_colset<<<dim3((a.S + 255) / 256, a.N), 256, 0, st>>>(s);
    k1_select<<<dim3((a.L + 255) / 256, a.N), 256, 0, st>>>(s);
    if constexpr (!EXACT) {
        if (match_only) {
            k1_rescan<T, true><<<256, 256, 0, st>>>(s);
            k1_compact<T, true><<<dim3(s.chunks, a.N), 1024, 0, st>>>(s);
        } else {
            k1_rescan<T, false><<<256, 256, 0, st>>>(s);
            k1_compact<T, false><<<dim3(s.chunks, a.N), 1024, 0, st>>>(s);
        }
    } else {
        k1_rescan<_Float16, false><<<256, 256, 0, st>>>(s);
        k1_compact<_Float16, false><<<dim3(s.chunks, a.N), 1024, 0, st>>>(s);
    }
    gf_prof_end("k1_unit", pu, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

struct K1Workspace {
    float2 *rowpart, *colpart, *rstat, *cstat;
    unsigned long long* rowbest;
    unsigned *colmax, *colset;
    int *samplecnt, *scancnt, *selj, *scanlist;
    unsigned long long* stamp;
    size_t zero_bytes, bytes;
};

K1Workspace k1_carve(void* ws, int N, int L, int S) {
    const int tilesM = (L + BM - 1) / BM, tilesN = (S + BN - 1) / BN;
    GfCarver c(ws);
    K1Workspace w;
    // zeroed every call (one memset): rowbest | colmax | colset | samplecnt | scancnt
    w.rowbest = c.take<unsigned long long>((size_t)N * L);
    w.colmax = c.take<unsigned>((size_t)N * S);
    w.colset = c.take<unsigned>((size_t)N * (1 << 15));
    w.samplecnt = c.take<int>((size_t)N * ((L + 1023) / 1024));
    w.scancnt = c.take<int>(1);
    w.zero_bytes = c.used();
    w.rowpart = c.take<float2>((size_t)N * tilesN * L);
    w.colpart = c.take<float2>((size_t)N * tilesM * S);
    w.rstat = c.take<float2>((size_t)N * L);
    w.cstat = c.take<float2>((size_t)N * S);
    w.selj = c.take<int>((size_t)N * L);
    w.scanlist = c.take<int>((size_t)N * L);
    w.stamp = c.take<unsigned long long>(K1_STAMP_WORDS + 1);
    w.bytes = c.used();
    return w;
}

// =============================================================================================
// Training: the sparse-supervision focal loss on the dual-softmax confidence and its backward,
// WITHOUT materialising conf or dS.  With p_ij = softmax_col(S)_ij * softmax_row(S)_ij and the loss a sum
// over the ground-truth positives (i, j) of l(p_ij)  (loftr_loss.py:246-270, coarse_matching.py:113-125):
//     d log p_ij / dS_kl = 2 d_ik d_jl - d_jl A_kl - d_ik B_kl,   A = softmax over rows, B = softmax over columns
//     dS = 2 G - A o (1 gc^T) - B o (gr 1^T),   g_ij = dL/dlog p_ij,  gr_k = sum_j g_kj,  gc_l = sum_i g_il
//     dF0 = mult * dS . F1,   dF1 = mult * dS^T . F0                    (S = mult * F0 . F1^T)
// k1_grad_panel computes one side's dense part: a workgroup keeps 128 rows of `fa` as MFMA fragments in
// registers (as k1_conf_panel), streams the other side's 64-row tiles through LDS, recomputes the similarity
// tile TRANSPOSED (lane = its panel row, registers = the tile's rows), turns it into
// W = exp(S - m_a)(g_a / l_a) + exp(S - m_b)(g_b / l_b) in registers and feeds it - accumulator as B operand,
// no data movement - to a second MFMA against the tile read column-wise (ds_read_b64_tr_b16), accumulating
// dFa^T for its 128 rows over the whole sweep.  W is scaled into fp16 range by 8192 / max|g|.
// =============================================================================================
struct GrArgs {
    const _Float16* fa;      // [N][La][256] panel side
    const _Float16* fb;      // [N][Lb][256] streamed side
    const float2* sa;        // [N][La] (max, sumexp) of the softmax that normalises over the OTHER side's index
    const float2* sb;        // [N][Lb]
    const float* ga;         // [N][La] g summed per panel row
    const float* gb;         // [N][Lb]
    const uint8_t* ma;       // [N][La] padding masks (0 = padded) or null; a pair with a padded member has sim = -1e9
    const uint8_t* mb;       // [N][Lb]   (coarse_matching.py:123-124) and therefore weight 0
    const unsigned* gmax;    // bits of max |g| (device)
    float mult;
    float* dfa;              // [N][La][256] fp32, written (not accumulated)
    int N, La, Lb, tilesA, tilesB;
};

template <bool MASKED>
__global__ __launch_bounds__(NT, 1) void k1_grad_panel(GrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* bst = reinterpret_cast<float2*>(smem + BN * 512);             // [64] streamed-side (c, w)
    float* bmk = reinterpret_cast<float*>(smem + BN * 512 + BN * 8);      // [64] streamed-side mask (1 / 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    const int bm = blockIdx.x % a.tilesA, n = blockIdx.x / a.tilesA, m0 = bm * BM;
    const int srow = tid >> 5, schunk = tid & 31;
    const float gmax = __uint_as_float(*a.gmax);
    const float sw = gmax > 0.f ? 8192.0f / gmax : 0.f;                   // W * sw fits fp16 comfortably
    const float k2 = a.mult * LOG2E;
    const _Float16* A = a.fa + ((size_t)n * a.La + m0 + wave * 32 + lr) * 256 + h * 8;
    const _Float16* B = a.fb + (size_t)n * a.Lb * 256;
    v8h af[16];
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) af[kg] = *reinterpret_cast<const v8h*>(A + kg * 16);
    // this lane's panel row: -m*log2e and g/l scaled
    const int krow = m0 + wave * 32 + lr;
    const float2 sta = a.sa[(size_t)n * a.La + krow];
    const float ka = (a.ma == nullptr || a.ma[(size_t)n * a.La + krow] != 0) ? 1.0f : 0.0f;
    // a padded row's statistics are those of a row of -1e9 fills: give it an offset that sends its exponentials to 0
    // (its weight is 0 as well); with masks every exponent is also capped, because a padded member's raw similarity
    // is not bounded by the maxima taken over the unpadded ones and inf * 0 would poison the sum
    const float ra = ka != 0.f ? -sta.x * LOG2E : -30000.f, wa = a.ga[(size_t)n * a.La + krow] * __builtin_amdgcn_rcpf(sta.y) * sw * ka;
    v16f dacc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[cb][r] = 0.f;
    v4u rb[8];
    float2 cs = make_float2(0.f, 0.f);
    float cm = 1.f;
    auto prefetch = [&](int bn) {
        const _Float16* g = B + (size_t)(bn * BN + srow) * 256 + schunk * 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) rb[p] = *reinterpret_cast<const v4u*>(g + (size_t)p * 8 * 256);
        if (tid < BN) {
            const size_t l = (size_t)n * a.Lb + bn * BN + tid;
            const float2 st = a.sb[l];
            cm = (a.mb == nullptr || a.mb[l] != 0) ? 1.0f : 0.0f;
            cs = make_float2(cm != 0.f ? -st.x * LOG2E : -30000.f, a.gb[l] * __builtin_amdgcn_rcpf(st.y) * sw * cm);
        }
    };
    prefetch(0);
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    for (int bn = 0; bn < a.tilesB; ++bn) {
        __syncthreads();                                   // the previous tile is consumed
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<v4u*>(smem + k1p_off(srow + 8 * p, schunk)) = rb[p];
        if (tid < BN) {
            bst[tid] = cs;
            bmk[tid] = cm;
        }
        __syncthreads();
        if (bn + 1 < a.tilesB) prefetch(bn + 1);
        // ---- similarity tile, transposed: D[row = tile row l][col = panel row k]
        v16f acc[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll
        for (int kg = 0; kg < 16; ++kg) {
            const v8h b0 = *reinterpret_cast<const v8h*>(smem + k1p_off(lr, 2 * kg + h));
            const v8h b1 = *reinterpret_cast<const v8h*>(smem + k1p_off(32 + lr, 2 * kg + h));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, af[kg], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, af[kg], acc[1], 0, 0, 0);
        }
        // ---- W^T in registers (fp16), then dFa^T[c][k] += sum_l Fb[l][c] W[l][k]
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {                   // 16 tile rows per k-group
            const int ni = gg >> 1, r0 = 8 * (gg & 1);
            v8h wf;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int l0 = ni * 32 + 8 * ((r0 >> 2) + q) + 4 * h;       // rows of registers r0+4q .. r0+4q+3
                const v4f s01 = *reinterpret_cast<const v4f*>(bst + l0);
                const v4f s23 = *reinterpret_cast<const v4f*>(bst + l0 + 2);
                const float cbv[4] = {s01.x, s01.z, s23.x, s23.z}, wbv[4] = {s01.y, s01.w, s23.y, s23.w};
                const v4f mk = *reinterpret_cast<const v4f*>(bmk + l0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float s2 = acc[ni][r0 + 4 * q + j] * k2;
                    // a padded member on either side zeroes the pair: wa carries the panel row's mask, wbv the
                    // streamed row's, and the cross terms take the other side's
                    float e1 = s2 + ra, e2 = s2 + cbv[j];
                    if constexpr (MASKED) { e1 = fminf(e1, 64.f); e2 = fminf(e2, 64.f); }
                    const float w = __builtin_amdgcn_exp2f(e1) * (wa * mk[j]) + __builtin_amdgcn_exp2f(e2) * (wbv[j] * ka);
                    wf[4 * q + j] = (_Float16)w;
                }
            }
            const int G = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
            const int lrow = gg * 16 + 4 * (G >> 1) + q4;                   // + 8 for the second read
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int ch = cb * 32 + 16 * (G & 1) + 4 * p4;             // 4 channels this lane addresses
                const int chunk = ch >> 3, sub = (ch & 7) * 2;
                const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(smem + k1p_off(lrow, chunk) + sub));
                const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(smem + k1p_off(lrow + 8, chunk) + sub));
                typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
                const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                dacc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, both), wf, dacc[cb], 0, 0, 0);
            }
        }
    }
    // ---- dFa = -(mult / sw) * accumulated; lane = panel row, registers = channels
    const float fin = sw > 0.f ? -a.mult / sw : 0.f;
    float* op = a.dfa + ((size_t)n * a.La + krow) * 256;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int c = cb * 32 + 8 * r4 + 4 * h;
            *reinterpret_cast<v4f*>(op + c) = v4f{dacc[cb][4 * r4] * fin, dacc[cb][4 * r4 + 1] * fin, dacc[cb][4 * r4 + 2] * fin,
                                                  dacc[cb][4 * r4 + 3] * fin};
        }
}

struct PosArgs {
    const _Float16* f0;
    const _Float16* f1;
    const float2* rstat;
    const float2* cstat;
    const int64_t* pb;
    const int64_t* pi;
    const int64_t* pj;
    const float* pw;         // per-positive weight or null
    float* conf;             // [P] p_ij
    float* loss;             // [P] focal term (weighted)
    float* grad;             // [P] dL_k / dlog p_k (weighted, unscaled)
    float* gr;               // [N][L]
    float* gc;               // [N][S]
    unsigned* gmax;
    float* d0;               // dF0 / dF1 for the sparse 2G term
    float* d1;
    int P, L, S;
    float mult, alpha, gamma, scale;
    const float* scale_dev;  // optional device scalar multiplied into `scale` (the upstream gradient of the loss sum)
};

// one wave per positive: p = exp(s - mr)/lr * exp(s - mc)/lc, focal term and its derivative
__global__ __launch_bounds__(256) void k1_pos_loss(PosArgs a) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= a.P) return;
    const int b = (int)a.pb[k], i = (int)a.pi[k], j = (int)a.pj[k];
    const v4h x = *reinterpret_cast<const v4h*>(a.f0 + ((size_t)b * a.L + i) * 256 + lane * 4);
    const v4h y = *reinterpret_cast<const v4h*>(a.f1 + ((size_t)b * a.S + j) * 256 + lane * 4);
    float d = (float)x.x * (float)y.x + (float)x.y * (float)y.y + (float)x.z * (float)y.z + (float)x.w * (float)y.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane != 0) return;
    const float s = d * a.mult;
    const float2 r = a.rstat[(size_t)b * a.L + i], c = a.cstat[(size_t)b * a.S + j];
    const float p = (__expf(s - r.x) / r.y) * (__expf(s - c.x) / c.y);
    const float w = a.pw ? a.pw[k] : 1.0f;
    const float pc = fminf(fmaxf(p, 1e-6f), 1.0f - 1e-6f);              // torch.clamp(conf, 1e-6, 1 - 1e-6)
    const float om = 1.0f - pc, lg = __logf(pc), pw_ = powf(om, a.gamma);
    a.conf[k] = p;
    a.loss[k] = -a.alpha * pw_ * lg * w;
    // d/dp of -alpha (1-p)^gamma log p, zero where the clamp is active; times p = d/dlog p
    const float dldp = (p > 1e-6f && p < 1.0f - 1e-6f) ? a.alpha * (a.gamma * powf(om, a.gamma - 1.0f) * lg - pw_ / pc) * w : 0.f;
    a.grad[k] = dldp * pc;
}

// g_k * scale -> row / column sums and the running max |g|
__global__ void k1_pos_scatter(PosArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.P) return;
    const int b = (int)a.pb[k], i = (int)a.pi[k], j = (int)a.pj[k];
    const float g = a.grad[k] * (a.scale_dev ? a.scale * a.scale_dev[0] : a.scale);
    atomicAdd(a.gr + (size_t)b * a.L + i, g);
    atomicAdd(a.gc + (size_t)b * a.S + j, g);
    atomicMax(a.gmax, __float_as_uint(fabsf(g)));
}

// the sparse 2G term: dF0[i] += 2 g mult F1[j], dF1[j] += 2 g mult F0[i]
__global__ __launch_bounds__(256) void k1_pos_grad(PosArgs a) {
    const int k = blockIdx.x, t = threadIdx.x;
    const int b = (int)a.pb[k], i = (int)a.pi[k], j = (int)a.pj[k];
    const float g2 = 2.0f * a.grad[k] * (a.scale_dev ? a.scale * a.scale_dev[0] : a.scale) * a.mult;
    atomicAdd(a.d0 + ((size_t)b * a.L + i) * 256 + t, g2 * (float)a.f1[((size_t)b * a.S + j) * 256 + t]);
    atomicAdd(a.d1 + ((size_t)b * a.S + j) * 256 + t, g2 * (float)a.f0[((size_t)b * a.L + i) * 256 + t]);
}

__global__ void k1_cast_f16(const float* x, _Float16* y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const v4f v = reinterpret_cast<const v4f*>(x)[i];
        reinterpret_cast<v4h*>(y)[i] = v4h{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
}

struct LossWs {
    _Float16 *f0h, *f1h;
    float2 *rowpart, *colpart, *rstat, *cstat;
    float *gr, *gc;
    unsigned* gmax;
    size_t zero_off, zero_bytes, bytes;
};

LossWs loss_carve(void* ws, int N, int L, int S) {
    const int tilesM = (L + BM - 1) / BM, tilesN = (S + BN - 1) / BN;
    GfCarver c(ws);
    LossWs w;
    w.f0h = c.take<_Float16>((size_t)N * L * 256);
    w.f1h = c.take<_Float16>((size_t)N * S * 256);
    w.rowpart = c.take<float2>((size_t)N * tilesN * L);
    w.colpart = c.take<float2>((size_t)N * tilesM * S);
    w.rstat = c.take<float2>((size_t)N * L);
    w.cstat = c.take<float2>((size_t)N * S);
    w.zero_off = c.used();
    w.gr = c.take<float>((size_t)N * L);
    w.gc = c.take<float>((size_t)N * S);
    w.gmax = c.take<unsigned>(1);
    w.zero_bytes = c.used() - w.zero_off;
    w.bytes = c.used();
    return w;
}

}   // namespace

extern "C" size_t gf_dual_softmax_workspace_bytes(int N, int L, int S) {
    if (N <= 0 || L <= 0 || S <= 0) return 0;
    return k1_carve(nullptr, N, L, S).bytes;
}

// match-only mode is built for the configuration the inference path runs (the row-panel form of the sweep)
extern "C" int gf_dual_softmax_match_only_supported(int dtype, int L, int S, int C, int masked, int force_one) {
    return (dtype == GF_F16 || dtype == GF_BF16) && C == 256 && L > 0 && S > 0 && L % BM == 0 && S % BN == 0 && !masked && !force_one;
}

// conf[b[e]][i[e]][j[e]] for e < P, recomputed bit-identically to what gf_dual_softmax_match wrote (or, in match-only mode, would
// have written) from the SAME features and the row / column statistics its last call left in `workspace`
extern "C" int gf_dual_softmax_conf_at(const void* f0, const void* f1, int dtype, int N, int L, int S, int C, float temperature,
                                       const int64_t* b, const int64_t* i, const int64_t* j, int P, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(f0 && f1 && b && i && j && out, "null pointer");
    GF_CHECK_ARG(gf_dual_softmax_match_only_supported(dtype, L, S, C, 0, 0) && N > 0 && P >= 0 && temperature > 0.f,
                 "built for 16-bit features, C = 256, L % 128 == 0, S % 64 == 0");
    if (workspace == nullptr || workspace_bytes < gf_dual_softmax_workspace_bytes(N, L, S)) {
        gf_set_error("gf_dual_softmax_conf_at: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    if (P == 0) return GF_OK;
    const K1Workspace w = k1_carve(workspace, N, L, S);
    SelArgs s{};
    s.N = N; s.L = L; s.S = S; s.f0 = f0; s.f1 = f1; s.rstat = w.rstat; s.cstat = w.cstat; s.C = C; s.mult = (1.0f / (float)C) / temperature;
    s.stamp = w.stamp;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GF_F16) k1_conf_at<_Float16><<<(P + 3) / 4, 256, 0, st>>>(s, b, i, j, P, out);
    else k1_conf_at<gf_bf16><<<(P + 3) / 4, 256, 0, st>>>(s, b, i, j, P, out);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_dual_softmax_match(const void* f0, const void* f1, int dtype, int N, int L, int S, int C,
                                     const uint8_t* mask0, const uint8_t* mask1, float temperature, float thr,
                                     int force_one, int w0c, int w1c, float scale, const float* scale0,
                                     const float* scale1, float* conf, int64_t* b_ids, int64_t* i_ids,
                                     int64_t* j_ids, float* mconf, float* mkpts0_c, float* mkpts1_c,
                                     int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(f0 && f1 && b_ids && i_ids && j_ids && mconf && mkpts0_c && mkpts1_c && counts, "null pointer");
    // conf == NULL: match-only mode (the matrix is not materialised; matches and mconf are bit-identical to the contract mode's)
    GF_CHECK_ARG(conf != nullptr || gf_dual_softmax_match_only_supported(dtype, L, S, C, mask0 != nullptr, force_one),
                 "match-only mode (conf == NULL) needs 16-bit features, C = 256, L % 128 == 0, S % 64 == 0, no masks, no forced match");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(C > 0 && C % (dtype == GF_F32 ? 32 : 64) == 0, "C must be a multiple of 32 (f32) / 64 (f16)");
    GF_CHECK_ARG((mask0 == nullptr) == (mask1 == nullptr), "mask0/mask1 must both be set or both be NULL");
    GF_CHECK_ARG(temperature > 0.f && w0c > 0 && w1c > 0, "bad temperature / grid width");
    GF_CHECK_ARG(thr >= 0.f, "thr must be >= 0 (confidences are probabilities)");
    if (workspace == nullptr || workspace_bytes < gf_dual_softmax_workspace_bytes(N, L, S)) {
        gf_set_error("gf_dual_softmax_match: workspace too small (%zu < %zu)", workspace_bytes,
                     gf_dual_softmax_workspace_bytes(N, L, S));
        return GF_ERR_WORKSPACE;
    }
    const K1Workspace w = k1_carve(workspace, N, L, S);
    K1Args a;
    a.f0 = f0; a.f1 = f1; a.N = N; a.L = L; a.S = S; a.C = C;
    a.mask0 = mask0; a.mask1 = mask1;
    a.inv_c = 1.0f / (float)C;                         // feat/sqrt(C) on both sides (coarse_matching.py:113)
    a.temperature = temperature;
    a.mult = (1.0f / (float)C) / temperature;
    a.tilesM = (L + BM - 1) / BM; a.tilesN = (S + BN - 1) / BN;
    a.rowpart = w.rowpart; a.colpart = w.colpart; a.rstat = w.rstat; a.cstat = w.cstat;
    a.rowbest = w.rowbest; a.colmax = w.colmax; a.conf = conf; a.thr = thr; a.dense = thr < 0.05f;
    a.stamp = w.stamp;
    a.esize = dtype == GF_F32 ? 4 : 2;
    SelArgs s;
    s.N = N; s.L = L; s.S = S;
    s.rowbest = w.rowbest; s.colmax = w.colmax; s.colset = w.colset; s.conf = conf; s.stamp = w.stamp;
    s.f0 = f0; s.f1 = f1; s.rstat = w.rstat; s.cstat = w.cstat; s.C = C; s.mult = a.mult;
    s.selj = w.selj; s.scanlist = w.scanlist; s.scancnt = w.scancnt; s.samplecnt = w.samplecnt; s.chunks = (L + 1023) / 1024; s.force_one = force_one; s.w0c = w0c; s.w1c = w1c;
    s.scale = scale; s.scale0 = scale0; s.scale1 = scale1;
    s.b_ids = b_ids; s.i_ids = i_ids; s.j_ids = j_ids; s.mconf = mconf; s.mk0 = mkpts0_c; s.mk1 = mkpts1_c;
    s.counts = counts;
    hipStream_t st = (hipStream_t)stream;
    return dtype == GF_F32 ? k1_launch<float>(a, s, w.rowbest, w.zero_bytes, st)
                           : dtype == GF_F16 ? k1_launch<_Float16>(a, s, w.rowbest, w.zero_bytes, st)
                                             : k1_launch<gf_bf16>(a, s, w.rowbest, w.zero_bytes, st);
}

extern "C" size_t gf_coarse_loss_workspace_bytes(int N, int L, int S) {
    if (N <= 0 || L <= 0 || S <= 0) return 0;
    return loss_carve(nullptr, N, L, S).bytes;
}

static int coarse_loss_check(const char* fn, int N, int L, int S, int C, int P, void* workspace, size_t workspace_bytes) {
    if (!(N > 0 && L > 0 && S > 0 && P > 0)) { gf_set_error("%s: empty problem", fn); return GF_ERR_INVALID_ARGUMENT; }
    if (C != 256 || L % BM != 0 || S % BN != 0) {
        gf_set_error("%s: built for C = 256, L %% 128 == 0, S %% 64 == 0 (the coarse level)", fn);
        return GF_ERR_INVALID_ARGUMENT;
    }
    if (workspace == nullptr || workspace_bytes < gf_coarse_loss_workspace_bytes(N, L, S)) {
        gf_set_error("%s: workspace too small", fn);
        return GF_ERR_WORKSPACE;
    }
    return GF_OK;
}

extern "C" int gf_coarse_loss_forward(const void* f0, const void* f1, int dtype, int N, int L, int S, int C,
                                      const uint8_t* mask0, const uint8_t* mask1, float temperature, const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j,
                                      int P, const float* pos_weight, float alpha, float gamma, float* pos_conf,
                                      float* pos_loss, float* pos_grad, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    GF_CHECK_ARG(f0 && f1 && pos_b && pos_i && pos_j && pos_conf && pos_loss && pos_grad, "null pointer");
    GF_CHECK_ARG(dtype == GF_F32 || dtype == GF_F16, "dtype must be GF_F32 or GF_F16");
    GF_CHECK_ARG(temperature > 0.f, "bad temperature");
    GF_CHECK_ARG((mask0 == nullptr) == (mask1 == nullptr), "mask0/mask1 must both be set or both be NULL");
    const int rc = coarse_loss_check("gf_coarse_loss_forward", N, L, S, C, P, workspace, workspace_bytes);
    if (rc != GF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const LossWs w = loss_carve(workspace, N, L, S);
    if (dtype == GF_F32) {
        k1_cast_f16<<<2048, 256, 0, st>>>((const float*)f0, w.f0h, (size_t)N * L * 64);
        k1_cast_f16<<<2048, 256, 0, st>>>((const float*)f1, w.f1h, (size_t)N * S * 64);
    } else {
        (void)hipMemcpyAsync(w.f0h, f0, (size_t)N * L * 512, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(w.f1h, f1, (size_t)N * S * 512, hipMemcpyDeviceToDevice, st);
    }
    K1Args a{};
    a.f0 = w.f0h; a.f1 = w.f1h; a.N = N; a.L = L; a.S = S; a.C = C;
    a.mask0 = mask0; a.mask1 = mask1;
    a.inv_c = 1.0f / (float)C; a.temperature = temperature; a.mult = (1.0f / (float)C) / temperature;
    a.tilesM = L / BM; a.tilesN = S / BN;
    a.rowpart = w.rowpart; a.colpart = w.colpart; a.rstat = w.rstat; a.cstat = w.cstat;
    if (mask0 == nullptr) {                                   // panel form (needs no masks)
        const int runs = (a.tilesN + PANEL_TILES - 1) / PANEL_TILES, units = N * a.tilesM * runs;
        a.rowparts = runs;
        gf_k1_stats_panel_launch(&a, GF_F16, units < 512 ? units : 512, st);
    } else {
        a.rowparts = a.tilesN;
        k1_stats<_Float16><<<dim3(a.tilesN * a.tilesM, N), NT, STAGE_BYTES, st>>>(a);
    }
    const int mx = L > S ? L : S;
    k1_reduce_stats<false><<<dim3((mx + 31) / 32, 2, N), 256, 0, st>>>(a);
    PosArgs p{};
    p.f0 = w.f0h; p.f1 = w.f1h; p.rstat = w.rstat; p.cstat = w.cstat; p.pb = pos_b; p.pi = pos_i; p.pj = pos_j;
    p.pw = pos_weight; p.conf = pos_conf; p.loss = pos_loss; p.grad = pos_grad; p.P = P; p.L = L; p.S = S;
    p.mult = a.mult; p.alpha = alpha; p.gamma = gamma;
    k1_pos_loss<<<(P + 3) / 4, 256, 0, st>>>(p);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_coarse_loss_backward(int N, int L, int S, int C, const uint8_t* mask0, const uint8_t* mask1,
                                       float temperature, const int64_t* pos_b,
                                       const int64_t* pos_i, const int64_t* pos_j, int P, const float* pos_grad,
                                       float scale, const float* scale_dev, float* d_f0, float* d_f1, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(pos_b && pos_i && pos_j && pos_grad && d_f0 && d_f1, "null pointer");
    const int rc = coarse_loss_check("gf_coarse_loss_backward", N, L, S, C, P, workspace, workspace_bytes);
    if (rc != GF_OK) return rc;
    GF_CHECK_ARG(S % BM == 0 && L % BN == 0, "S must be a multiple of 128 too");     // before anything is enqueued
    hipStream_t st = (hipStream_t)stream;
    const LossWs w = loss_carve(workspace, N, L, S);
    (void)hipMemsetAsync((char*)workspace + w.zero_off, 0, w.zero_bytes, st);
    PosArgs p{};
    p.f0 = w.f0h; p.f1 = w.f1h; p.pb = pos_b; p.pi = pos_i; p.pj = pos_j; p.grad = const_cast<float*>(pos_grad);
    p.gr = w.gr; p.gc = w.gc; p.gmax = w.gmax; p.d0 = d_f0; p.d1 = d_f1; p.P = P; p.L = L; p.S = S;
    p.mult = (1.0f / (float)C) / temperature; p.scale = scale; p.scale_dev = scale_dev;
    k1_pos_scatter<<<(P + 255) / 256, 256, 0, st>>>(p);
    GrArgs g{};
    g.gmax = w.gmax; g.mult = p.mult; g.N = N;
    // dF0: panel = f0 rows (row statistics, gr), streamed = f1 rows (column statistics, gc)
    g.fa = w.f0h; g.fb = w.f1h; g.sa = w.rstat; g.sb = w.cstat; g.ga = w.gr; g.gb = w.gc; g.dfa = d_f0;
    g.ma = mask0; g.mb = mask1;
    g.La = L; g.Lb = S; g.tilesA = L / BM; g.tilesB = S / BN;
    if (mask0) k1_grad_panel<true><<<N * g.tilesA, NT, PANEL_LDS + BN * 4, st>>>(g);
    else k1_grad_panel<false><<<N * g.tilesA, NT, PANEL_LDS + BN * 4, st>>>(g);
    // dF1: roles swapped (S must then tile by 128 and L by 64: checked above)
    g.fa = w.f1h; g.fb = w.f0h; g.sa = w.cstat; g.sb = w.rstat; g.ga = w.gc; g.gb = w.gr; g.dfa = d_f1;
    g.ma = mask1; g.mb = mask0;
    g.La = S; g.Lb = L; g.tilesA = S / BM; g.tilesB = L / BN;
    if (mask0) k1_grad_panel<true><<<N * g.tilesA, NT, PANEL_LDS + BN * 4, st>>>(g);
    else k1_grad_panel<false><<<N * g.tilesA, NT, PANEL_LDS + BN * 4, st>>>(g);
    k1_pos_grad<<<P, 256, 0, st>>>(p);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
#endif   // K1_PART
