// K4: dense softmax attention of all L query tokens over the K << L tokens at RANSAC-inlier cells
// (GeoTransformer 'self' layers, model/geo_transformer/transformer.py:111-124 ->
//  FullAttention.forward, model/geo_transformer/geo_attention.py:72-101, no masks on this path).
//
// Two launches per call:
//   attn_gather_kv : keys/values of the listed tokens are gathered from the already projected
//                    [L, C] maps into compact, 32-key padded buffers (K row-major; V row-major for
//                    fp32, [channel][key] with the MFMA k-order baked in for fp16).  K is read from
//                    device memory, so no host sync is needed to size anything.
//   attn_self      : flash-style forward.  One workgroup = 32 QB queries x 4 heads (wave = head, QB = 1 or 4 query blocks).
//                    S^T = K.Q^T is computed "swapped" so that the softmax axis (keys) lies in the
//                    accumulator registers of the lane that owns the query: running max / sum /
//                    rescale are lane-local, P^T feeds the P.V MFMA straight from registers
//                    (accumulator-as-B-operand), and O^T is normalised per lane at the end.
#include <math.h>

#include <type_traits>

#include <cstdlib>

#include "gf_common.h"

namespace {

constexpr int HD = 64;      // head dim
constexpr int NH = 4;       // heads  (geo_config.py:12)
constexpr int CC = 256;     // channels
constexpr int KT = 32;      // keys per tile
constexpr float K4_DEFER = 8.0f;   // 16-bit modes: a query's softmax reference moves only when a tile's maximum exceeds it by this much (log2 units)

struct AtArgs {
    const void* q;          // [N][L][ldq]
    const void* kmap;       // [N][L][ldk]   projected keys of every token
    const void* vmap;
    long ldq, ldk, ldv;
    const int32_t* idx;     // [N][idx_stride] ascending token list
    long idx_stride;
    const int32_t* nkeys;   // nkeys[n * nkeys_stride]
    int nkeys_stride;
    void* out;              // [N][L][CC]
    void* kc;               // [N][Kpad][CC]
    void* vc;               // fp32: [N][Kpad][CC]; fp16: [N][CC][Kpad]
    int N, L, Kpad;
    float softmax_temp;
};

// position of key p (0..31) inside its 32-key block of the fp16 V^T image: element j of lane half h
// in k-step s of the P.V MFMA must be key 16s + 8(j>>2) + 4h + (j&3)  ->  swap bits 2 and 3
__device__ __forceinline__ int vt_pos(int p) { return (p & 19) | ((p & 4) << 1) | ((p & 8) >> 1); }

template <typename T>
__global__ __launch_bounds__(256) void attn_gather_kv(AtArgs a) {
    const int n = blockIdx.y, t = threadIdx.x;
    const int K = a.nkeys[(size_t)n * a.nkeys_stride];
    const int32_t* idx = a.idx + (size_t)n * a.idx_stride;
    const T* km = (const T*)a.kmap + (size_t)n * a.L * a.ldk;
    const T* vm = (const T*)a.vmap + (size_t)n * a.L * a.ldv;
    if constexpr (std::is_same<T, float>::value) {
        const int p0 = blockIdx.x * KT;
        if (p0 >= ((K + KT - 1) / KT) * KT) return;
        T* kc = (T*)a.kc + ((size_t)n * a.Kpad + p0) * CC;
        T* vc = (T*)a.vc + ((size_t)n * a.Kpad + p0) * CC;
        for (int p = 0; p < KT; ++p) {
            const bool ok = p0 + p < K;
            const int tok = ok ? idx[p0 + p] : 0;
            kc[(size_t)p * CC + t] = ok ? km[(size_t)tok * a.ldk + t] : 0.f;
            vc[(size_t)p * CC + t] = ok ? vm[(size_t)tok * a.ldv + t] : 0.f;
        }
    } else {
        // 16-bit modes (round 5): EIGHT keys per workgroup instead of 32 - four times the workgroups (the 32-key form ran 608 of
        // them at 1195 keys x 16 images: 2.4 per CU, each a chain of 64 two-byte loads per thread; 22 us for 39 MB) - and the K rows
        // move as 16-byte pieces (thread = row t >> 5, piece t & 31: one load + one store per thread); V^T: thread = channel, its 8
        // keys land as two 8-byte pieces of the channel's 64-byte row of the 32-key block (vt_pos swaps key bits 2 and 3).
        constexpr int KS = 8;
        using V8 = gf_vec<T, 8>;
        using V4 = gf_vec<T, 4>;
        const int kround = ((K + KT - 1) / KT) * KT;                     // the ragged tile's spare key slots are written as zeros
        // (the key count lives on the device: a fixed grid of 64 workgroups per image walks the sub-blocks instead of one workgroup
        // per possible sub-block, of which 4 in 5 would find nothing to do)
        for (int p0 = blockIdx.x * KS; p0 < kround; p0 += gridDim.x * KS) {
            {
                const int row = t >> 5, piece = t & 31;
                const bool ok = p0 + row < K;
                const int tok = ok ? idx[p0 + row] : 0;
                V8 v = *reinterpret_cast<const V8*>(km + (size_t)tok * a.ldk + piece * 8);
                if (!ok) v = V8{(T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
                *reinterpret_cast<V8*>((T*)a.kc + ((size_t)n * a.Kpad + p0 + row) * CC + piece * 8) = v;
            }
            T vt[KS];
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const bool ok = p0 + j < K;
                const int tok = ok ? idx[p0 + j] : 0;
                vt[j] = ok ? vm[(size_t)tok * a.ldv + t] : (T)0;
            }
            // key p = 8 c + j of its 32-key block (c = sub-block): position (j & 3) | (c & 1) << 2 | (j >> 2) << 3 | (c >> 1) << 4
            const int blk = p0 / KT, c = (p0 % KT) / KS;
            T* dst = (T*)a.vc + ((size_t)n * CC + t) * a.Kpad + blk * KT + ((c & 1) << 2) + ((c >> 1) << 4);
            *reinterpret_cast<V4*>(dst) = V4{vt[0], vt[1], vt[2], vt[3]};
            *reinterpret_cast<V4*>(dst + 8) = V4{vt[4], vt[5], vt[6], vt[7]};
        }
    }
}

// LDS images.  K: [32 keys][CC] rows of CC*sizeof(T) bytes, 16-B chunk c stored at c ^ (row & 15)
// (16 distinct slots for the 16 rows of a ds_read_b128 lane group).
template <typename T>
__device__ __forceinline__ int k_off(int row, int chunk) { return row * (CC * (int)sizeof(T)) + ((chunk ^ (row & 15)) << 4); }
// fp16 V^T: [CC channels][32 keys] rows of 64 B, chunk c (0..3) stored at c ^ ((row >> 2) & 3)
__device__ __forceinline__ int vt_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

#ifndef K4_TRACE
#define K4_TRACE 0
#endif
#if K4_TRACE                           // -DK4_TRACE=1: phase stamps of tiles 4..7 of every workgroup's wave 0 (tools/k4_trace.py)
__device__ long long k4_trace[2048 * 32];
#define K4_T(slot) do { if (tid == 0 && tile >= 4 && tile < 8 && blockIdx.y * gridDim.x + blockIdx.x < 2048) k4_trace[(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (tile - 4) * 8 + (slot)] = clock64(); } while (0)
#else
#define K4_T(slot)
#endif

// DIRECT form: 16-byte chunk of V row `row` that LDS slot `s` (0..31) of that row holds: the 32-byte block index XOR 2 (row & 3)
__device__ __forceinline__ int k4_vslot_chunk(int row, int s) { return ((((s >> 1) ^ (2 * (row & 3))) << 1) | (s & 1)); }
// A operand of the P.V product from the ROW-MAJOR V image [32 keys][512 B]: lane (channel blk32 * 16 + 16 (G & 1) + i of its 32-channel block,
// half G >> 1) gets keys 16 s2 + 8 (j >> 2) + 4 (G >> 1) + (j & 3), j = 0..7 - the k order of the packed P^T
template <typename T>
__device__ __forceinline__ typename Mma32<T>::Frag k4_vtr_frag(const char* img, int blk0, int s2, int lane) {
    const int G = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 16 * s2 + 4 * (G >> 1) + q;
    const char* base = img + row * 512 + (((blk0 + (G & 1)) ^ (2 * q)) << 5) + p * 8;
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base));
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(base + 8 * 512));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename Mma32<T>::Frag, both);
}

// LDS-DMA (16-bit modes): 64 lanes x 16 B of a staged tile straight from the compact K / V^T buffers into LDS, no registers.
// MUBUF form: the waits the compiler inserts stay counted (the FLAT form makes every LDS wait lgkmcnt(0)).
struct AtRsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ AtRsrc at_rsrc(const void* p, unsigned bytes) {
    return AtRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ void at_lds_dma(const AtRsrc& rs, char* dst, int voffset, int soffset) {        // -> 1 KiB at dst
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}

// max(x[lane], x[lane ^ 32]) in every lane: one v_permlane32_swap (VALU) instead of a ds_bpermute, whose lgkmcnt(0) wait also
// drained the fragment reads in flight
__device__ __forceinline__ float half_max(float x) {
    const gf_v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
}

// WV = 4: one wave per head, QB query blocks per wave.  WV = 8 (16-bit modes): two waves per head (query groups of 32 QB queries
// each) share every staged tile - half the L2 -> LDS bytes per query, and with QB = 1 the wave fits 128 registers: four waves per
// SIMD (two workgroups of eight) cover each other's LDS / MFMA latencies where two waves per SIMD ran nearly serially.
// WV = 16 (round 5): FOUR waves per head, one 32-query block each - a staged tile serves 128 queries of every head (the L2 -> LDS
// stream, not the matrix pipe, is what paces the two-wave forms: 1.95 GB per 16-image call at 1195 keys = 37 GB/s per CU, half of what
// the LDS-DMA path reaches), 16 waves = four per SIMD, and a THREE-image ring: the tile after next is already requested when a tile is
// computed (counted vmcnt: the wait at a tile's top leaves the next tile's pieces in flight).
// DIRECT (16-bit modes, round 5): no gather pass and no compact K / V^T buffers - the K AND V rows of a tile's 32 keys come straight from the
// projected maps by LDS-DMA with a per-lane source row (the token list entry of the lane's row, read a tile ahead; rows behind the key count are
// out-of-range lanes = zeros in LDS), both row-major in LDS; the P.V product's A operand (channels x keys) is fetched from the row-major V image
// with the transposing read ds_read_b64_tr_b16 in the key order of the packed P^T (k2_linear_attention.hip's la16_tr_frag with the two 4-row
// groups 8 rows apart); the V image's 32-byte blocks are XORed with 2 (row & 3) so that the four rows a 16-lane group reads fall on distinct banks.
// PRE (16-bit modes, round 5; the default): the softmax scale lives in the Q operand - q' = round(q * log2(e) / sqrt(D)) when the query
// rows are loaded - and the query's reference m' enters the S MFMA chain as its C operand (a 16-register tile of -m', rewritten only when
// the reference moves), so the accumulator IS the exponent's argument: per logit one exponential, half a max3, half a pair add and
// half a conversion - no FMA.  The price is one more 16-bit rounding of q (the product q . k is then scaled BEFORE the sum instead of
// after it: the same softmax(Q K^T / sqrt(D)) V; the oracle's storage mode rounds q * c the same way).
// (Round 6: this kernel is the FALLBACK of the head form - fp32 parity mode, unaligned or very wide rows - and is built in one configuration: four waves,
// the compact K / V^T buffers of the gather pass, the prescaled Q operand.  The other settings of the constants below were round-5 experiments;
// their measurements are in DESIGN.md section 3 and profiles/r05_k4_experiments.txt, their launch switches are gone.)
template <typename T, int QB>
__global__ __launch_bounds__(256, QB <= 2 ? 2 : 1) void attn_self(AtArgs a) {
    constexpr int WV = 4;
    constexpr bool MSUM = false, PRE = true, DIRECT = false;
    using M = Mma32<T>;
    using Frag = typename M::Frag;
    constexpr bool F32 = std::is_same<T, float>::value;
    constexpr int KG = M::kGroup;            // head-dim elements per k-group
    constexpr int NG = HD / KG;
    constexpr int EPC = 16 / sizeof(T);      // elements per 16-B chunk
    constexpr int KBYTES = KT * CC * sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroup ids go round-robin over the 8 XCDs: give each XCD whole images (their compact K / V^T, 1.2 MB at 1200 keys,
    // then stay in that XCD's L2 for all the query blocks), not a slice of every image
    const int gx = gridDim.x, total = gx * gridDim.y;
    int id = blockIdx.y * gx + blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int tid = threadIdx.x;
    const int n = id / gx, q0 = (id - n * gx) * (32 * QB * (WV / 4)) + (tid >> 8) * (32 * QB);
    const int head = (tid >> 6) & 3, lane = tid & 63, h = lane >> 5, lr = lane & 31;
    const int K = a.nkeys[(size_t)n * a.nkeys_stride];
    // QB blocks of 32 queries per wave: a staged K / V tile (32 keys x 256 channels, 32 KiB) serves 32 QB queries of every head
    // (with one block a workgroup streams the image's whole K and V for 32 queries: L2-bound at ~1200 keys)
    Frag qf[QB][NG];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + 32 * qb + lr, a.L - 1);
        const T* qp = (const T*)a.q + ((size_t)n * a.L + qrow) * a.ldq + head * HD;
#pragma unroll
        for (int g = 0; g < NG; ++g) qf[qb][g] = *reinterpret_cast<const Frag*>(qp + g * KG + h * (KG / 2));
    }
    constexpr bool PRESCALE = PRE && !F32;
    if constexpr (PRESCALE) {
        const float c2 = a.softmax_temp * 1.44269504088896341f;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j8 = 0; j8 < (int)(sizeof(Frag) / sizeof(T)); ++j8) qf[qb][g][j8] = (T)(gf_to_float(qf[qb][g][j8]) * c2);
    }
    v16f o[QB][2];
    v16f negm[PRESCALE ? QB : 1];                        // PRE: -m' of the lane's query in every register (the S chain's C operand)
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = PRESCALE ? 0.f : -INFINITY;              // PRE: the reference in scaled log2 units, set by the first tile
        if constexpr (PRESCALE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
        }
        l[qb] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][b][r] = 0.f;
    }
    // MSUM (round 5 experiment, GF_K4_MSUM=1; NOT the default): the row sums l = sum_k P from the matrix pipe - P^T (the packed operand
    // of P.V) against a ones operand, two MFMAs per 32 x 32 logits instead of 16 vector adds; every row of the result tile holds the
    // query's sum over BOTH lane halves' keys, l = register 0 (a sum of the ROUNDED probabilities).  Same box, 16 images x 1195 keys:
    // 229-230 us per call against 221-223 with the vector adds - ten MFMAs per tile instead of eight cost more than sixteen adds.
    v16f lacc[QB];
    Frag ones;
    if constexpr (!F32 && MSUM) {
#pragma unroll
        for (int j8 = 0; j8 < (int)(sizeof(Frag) / sizeof(T)); ++j8) ones[j8] = (T)1.0f;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) lacc[qb][r] = 0.f;
    }
    const T* kc = (const T*)a.kc + (size_t)n * a.Kpad * CC;
    const int ntiles = (K + KT - 1) / KT;
    // logits in log2 units: exp(x - m) = exp2(x' - m') with x' = s * (temp * log2 e): one multiply per element instead of two
    constexpr bool FAST = !std::is_same<T, float>::value;              // 16-bit modes: hardware exponential
    const float scale2 = FAST ? a.softmax_temp * 1.44269504088896341f : a.softmax_temp;
    const float defer = K4_DEFER / scale2;                                // the threshold on the unscaled logits (16-bit modes)
    (void)defer;
    auto compute = [&](int tile, const char* ks, const char* vs) {
        const bool ragged = (tile + 1) * KT > K;          // only the last tile can hold key slots beyond K
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // ---- S^T tile: rows = keys (registers), column = query (lane)
            v16f s;
            if constexpr (PRESCALE) {
                s = negm[qb];                                 // the chain starts at -m': s = q' . k - m' is the exponent's argument
            } else {
    #pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
            }
    #pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int chunk = (head * HD + g * KG + h * (KG / 2)) / EPC;
                const Frag kf = *reinterpret_cast<const Frag*>(ks + k_off<T>(lr, chunk));
                M::mma(kf, qf[qb][g], s);
            }
            if (qb == 0) K4_T(3);
            float x[16];
            float tmax = -INFINITY;
            float psum = 0.f;
            if constexpr (PRESCALE) {
                if (ragged) {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = tile * KT + gf_acc_row(r, h) < K ? s[r] : -INFINITY;
                }
    #pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = half_max(tmax);                        // the tile's maximum RELATIVE to the reference (finite: a real key in every tile)
                // deferred reference: it moves up by d = tmax when the tile's maximum is more than K4_DEFER above it, and in the first
                // tile (from 0 to the tile's maximum, whatever its sign); only the queries that need it (d = 0, alpha = 1 for the others)
                const bool need = tmax > K4_DEFER || tile == 0;
                if (__any(need)) {
                    const float d = need ? tmax : 0.f;
                    const float alpha = tile == 0 ? 0.f : __builtin_amdgcn_exp2f(-d);         // (first tile: O = l = 0; no 0 * inf)
                    if constexpr (MSUM) lacc[qb][0] *= alpha;
                    l[qb] *= alpha;
    #pragma unroll
                    for (int b = 0; b < 2; ++b)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][b][r] *= alpha;
                    m[qb] += d;
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        negm[qb][r] = -m[qb];
                        s[r] -= d;                            // this tile's logits against the new reference
                    }
                }
    #pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(s[r]);
                if constexpr (!MSUM) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 ps2{0.f, 0.f};
    #pragma unroll
                    for (int r = 0; r < 16; r += 2) ps2 += f2{x[r], x[r + 1]};
                    psum = ps2.x + ps2.y;
                }
            } else if constexpr (FAST) {
                // 16-bit modes: the running maximum is kept on the UNSCALED logits (scale2 > 0) and the scale rides in the
                // exponent's FMA: exp2(s * scale2 - max * scale2); one max3 per two logits, one FMA, one exp2, one add per
                // logit, all single-issue (the separate multiply was packed into v_pk_mul_f32 pairs: dear beside MFMAs)
                if (ragged) {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = tile * KT + gf_acc_row(r, h);
                        x[r] = key < K ? s[r] : -INFINITY;
                        tmax = fmaxf(tmax, x[r]);
                    }
                } else {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        x[r] = s[r];
                        tmax = fmaxf(tmax, x[r]);
                    }
                }
                tmax = half_max(tmax);
                // Deferred reference (round 5; guide T13): a query's m moves only when its tile maximum exceeds it by more than
                // K4_DEFER = 8 in the exponent's log2 units (always in the first tile: m = -inf), so P <= 2^8 - exact in fp16 / bf16 up
                // to their rounding, the row sum is fp32 - and the rescale of O, which the wave-wide test `some running maximum moved`
                // ran in nearly every one of 38 tiles (a third of the kernel's vector instructions), becomes a rare side path.  Only
                // the queries that need it change m (alpha = 1 for the others): a per-query rule, restated in the oracle.
                const bool need = tmax - m[qb] > defer;         // tmax finite: every tile holds at least one real key
                if (__any(need)) {
                    const float mnew = need ? tmax : m[qb];
                    const float alpha = __builtin_amdgcn_exp2f((m[qb] - mnew) * scale2);      // 0 in the first tile (O = l = 0)
                    if constexpr (MSUM) lacc[qb][0] *= alpha;    // (the tile's other rows are copies nobody reads)
                    l[qb] *= alpha;
    #pragma unroll
                    for (int b = 0; b < 2; ++b)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][b][r] *= alpha;
                    m[qb] = mnew;
                }
                const float nms = -m[qb] * scale2;
    #pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_exp2f(fmaf(x[r], scale2, nms));
                if constexpr (!MSUM) {
                    // the row sum as PAIR adds (v_pk_add_f32: eight instructions instead of sixteen; the two partial sums meet at the end)
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 ps2{0.f, 0.f};
    #pragma unroll
                    for (int r = 0; r < 16; r += 2) ps2 += f2{x[r], x[r + 1]};
                    psum = ps2.x + ps2.y;
                }
            } else {
                if (ragged) {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = tile * KT + gf_acc_row(r, h);
                        x[r] = key < K ? s[r] * scale2 : -INFINITY;
                        tmax = fmaxf(tmax, x[r]);
                    }
                } else {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        x[r] = s[r] * scale2;
                        tmax = fmaxf(tmax, x[r]);
                    }
                }
                tmax = half_max(tmax);
                const float mnew = fmaxf(m[qb], tmax);          // finite: every tile holds at least one real key
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    x[r] = expf(x[r] - mnew);
                    psum += x[r];
                }
                if (__any(mnew != m[qb])) {                     // a running maximum moved somewhere in the wave: rescale (else alpha = 1)
                    const float alpha = expf(m[qb] - mnew);
                    l[qb] *= alpha;
    #pragma unroll
                    for (int b = 0; b < 2; ++b)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][b][r] *= alpha;
                }
                m[qb] = mnew;
            }
            l[qb] += psum;
            if (qb == 0) K4_T(4);
            // ---- O^T += V^T . P^T with P^T taken from the registers as the B operand
            if constexpr (F32) {
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* vrow = reinterpret_cast<const float*>(vs) + gf_acc_row(r, h) * CC + head * HD + lr;
                    o[qb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], x[r], o[qb][0], 0, 0, 0);
                    o[qb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], x[r], o[qb][1], 0, 0, 0);
                }
            } else {
    #pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const Frag pf{(T)x[8 * s2], (T)x[8 * s2 + 1], (T)x[8 * s2 + 2], (T)x[8 * s2 + 3],
                                  (T)x[8 * s2 + 4], (T)x[8 * s2 + 5], (T)x[8 * s2 + 6], (T)x[8 * s2 + 7]};
    #pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        Frag vf;
                        if constexpr (DIRECT) vf = k4_vtr_frag<T>(vs, head * 4 + b * 2, s2, lane);
                        else vf = *reinterpret_cast<const Frag*>(vs + vt_off(head * HD + b * 32 + lr, 2 * s2 + h));
                        M::mma(vf, pf, o[qb][b]);
                    }
                    if constexpr (MSUM) M::mma(ones, pf, lacc[qb]);
                }
            }
            if (qb == 0) K4_T(5);
            if (qb == QB - 1) K4_T(6);
            }
    };
    if constexpr (F32) {
        // fp32 (parity mode): the K / V tile of the NEXT iteration waits in registers (issued before this tile's products,
        // written to LDS after the barrier that ends them)
        constexpr int CPR = CC / EPC;                         // chunks per row
        constexpr int PASSES = KT * CPR / 256;
        v4u rk[PASSES], rv[PASSES];
        auto fetch = [&](int tile) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int e = p * 256 + tid, row = e / CPR, c = e % CPR;
                rk[p] = *reinterpret_cast<const v4u*>(kc + ((size_t)tile * KT + row) * CC + c * EPC);
            }
            const T* vc = (const T*)a.vc + ((size_t)n * a.Kpad + (size_t)tile * KT) * CC;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) rv[p] = *reinterpret_cast<const v4u*>(vc + (size_t)(p * 256 + tid) * EPC);
        };
        if (ntiles > 0) fetch(0);
        for (int tile = 0; tile < ntiles; ++tile) {
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int e = p * 256 + tid, row = e / CPR, c = e % CPR;
                *reinterpret_cast<v4u*>(smem + k_off<T>(row, c)) = rk[p];
            }
#pragma unroll
            for (int p = 0; p < PASSES; ++p) *reinterpret_cast<v4u*>(smem + KBYTES + (size_t)(p * 256 + tid) * 16) = rv[p];
            __syncthreads();
            if (tile + 1 < ntiles) fetch(tile + 1);
            compute(tile, smem, smem + KBYTES);
        }
    } else {
        // 16-bit modes: two LDS images of the (K, V^T) tile, filled by LDS-DMA.  Iteration t: my requests of tile t have
        // landed (vmcnt(0)) -> barrier (everyone's have, and everyone is done reading tile t-1) -> request tile t+1 into the
        // image tile t-1 was read from -> products of tile t.  One barrier per tile, no staging registers, no LDS writes.
        // (round 3, with the tile passing through registers and two barriers: 335 us per 16-image call at 1195 keys, 218 us
        // with the staging compiled out)
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const AtRsrc rk = DIRECT ? at_rsrc((const T*)a.kmap + (size_t)n * a.L * a.ldk, (unsigned)((size_t)a.L * a.ldk * sizeof(T)))
                                 : at_rsrc(kc, (unsigned)((size_t)a.Kpad * CC * sizeof(T)));
        const AtRsrc rv = DIRECT ? at_rsrc((const T*)a.vmap + (size_t)n * a.L * a.ldv, (unsigned)((size_t)a.L * a.ldv * sizeof(T)))
                                 : at_rsrc((const T*)a.vc + (size_t)n * CC * a.Kpad, (unsigned)((size_t)CC * a.Kpad * sizeof(T)));
        // K image: 1 KiB group g = key rows 2g, 2g+1; slot s of row r holds chunk s ^ (r & 15) (k_off)
        // V^T image: group g = channels 16g..16g+15, 4 slots of 16 B; slot s of channel c holds chunk s ^ ((c >> 2) & 3) (vt_off)
        // (DIRECT: the V image is row-major like K's: group g = key rows 2g, 2g+1, slot s of row r holds chunk k4_vslot_chunk(r, s))
        constexpr int PW = 16 / WV;                            // 1-KiB pieces per wave, tile and operand
        int kvo[PW], vvo[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int g = wv * PW + i;
            const int krow = 2 * g + (lane >> 5), kslot = lane & 31;
            if constexpr (DIRECT) {
                kvo[i] = (kslot ^ (krow & 15)) << 4;                       // inside the token's row; the row itself comes from the token list
                vvo[i] = k4_vslot_chunk(krow, kslot) << 4;
            } else {
                kvo[i] = krow * (CC * (int)sizeof(T)) + ((kslot ^ (krow & 15)) << 4);
                const int vrow = 16 * g + (lane >> 2), vslot = lane & 3;
                vvo[i] = vrow * (a.Kpad * (int)sizeof(T)) + ((vslot ^ ((vrow >> 2) & 3)) << 4);
            }
        }
        constexpr int NB = WV == 16 ? 3 : 2;                   // LDS images of the (K, V^T) tile
        // DIRECT: the token of the lane's row of piece i, for the tile whose request comes next (read a tile ahead)
        const int32_t* idxn = a.idx + (size_t)n * a.idx_stride;
        int tokn[PW];
        auto load_tok = [&](int tile) {
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int key = tile * KT + 2 * (wv * PW + i) + (lane >> 5);
                tokn[i] = key < K ? idxn[key] : -1;
            }
        };
        auto request = [&](int tile, int slot) {
            char* img = smem + slot * (2 * KBYTES);
            if constexpr (DIRECT) {
#pragma unroll
                for (int i = 0; i < PW; ++i) {
                    // a key slot behind the count: an offset outside the map = an out-of-range lane: zeros in LDS
                    const int kofs = tokn[i] >= 0 ? tokn[i] * (int)(a.ldk * sizeof(T)) + kvo[i] : 0x7FFFFFF0;
                    at_lds_dma(rk, img + (wv * PW + i) * 1024, kofs, 0);
                }
#pragma unroll
                for (int i = 0; i < PW; ++i) {
                    const int vofs = tokn[i] >= 0 ? tokn[i] * (int)(a.ldv * sizeof(T)) + vvo[i] : 0x7FFFFFF0;
                    at_lds_dma(rv, img + KBYTES + (wv * PW + i) * 1024, vofs, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < PW; ++i) at_lds_dma(rk, img + (wv * PW + i) * 1024, kvo[i], tile * KBYTES);
#pragma unroll
                for (int i = 0; i < PW; ++i) at_lds_dma(rv, img + KBYTES + (wv * PW + i) * 1024, vvo[i], tile * (KT * (int)sizeof(T)));
            }
        };
        if constexpr (NB == 2) {
            if constexpr (DIRECT) {
                if (ntiles > 0) load_tok(0);                      // (waited for by the compiler in front of the first request)
            }
            if (ntiles > 0) request(0, 0);
            if constexpr (DIRECT) {
                if (ntiles > 1) load_tok(1);
            }
            // (round 5: unrolled by the two images so that an image's base is a compile-time constant - the eight fragment addresses of
            // a tile are lane constants + immediates instead of eight v_add3 per tile: the kernel is paced by its instruction count)
            auto iter = [&](auto par_c, int tile) {
                constexpr int PAR = decltype(par_c)::value;
                K4_T(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                K4_T(1);
                __syncthreads();
                K4_T(2);
                if (tile + 1 < ntiles) request(tile + 1, PAR ^ 1);
                if constexpr (DIRECT) {
                    if (tile + 2 < ntiles) load_tok(tile + 2);    // lands under this tile's products; the next top's vmcnt(0) covers it
                }
                const char* img = smem + PAR * (2 * KBYTES);
                compute(tile, img, img + KBYTES);
            };
            for (int tile = 0; tile < ntiles; tile += 2) {
                iter(std::integral_constant<int, 0>{}, tile);
                if (tile + 1 < ntiles) iter(std::integral_constant<int, 1>{}, tile + 1);
            }
        } else {
            // three images: iteration t waits for ITS tile only (the 2 PW youngest requests are tile t + 1's), the barrier says that
            // everyone's pieces of tile t have landed and that everyone is done with tile t - 1, whose image then takes tile t + 2.
            // Requests behind the last tile are made all the same (the counted wait needs a fixed number in flight): out of the
            // descriptor's range they fetch nothing, in range (V^T rows) they fetch bytes nobody reads.
            int slot = 0;
            if (ntiles > 0) {
                request(0, 0);
                request(1, 1);
            }
            for (int tile = 0; tile < ntiles; ++tile) {
                K4_T(0);
                if constexpr (PW == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                K4_T(1);
                __syncthreads();
                K4_T(2);
                const int nslot = slot == 0 ? 2 : slot - 1;             // (slot + 2) % 3
                request(tile + 2, nslot);
                const char* img = smem + slot * (2 * KBYTES);
                compute(tile, img, img + KBYTES);
                slot = slot == 2 ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // nothing of the ring in flight behind the kernel's LDS
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float lsum;
        if constexpr (F32 || !MSUM) lsum = l[qb] + __shfl_xor(l[qb], 32, 64);
        else lsum = lacc[qb][0];
        const int qi = q0 + 32 * qb + lr;
        if (qi < a.L) {
            T* op = (T*)a.out + ((size_t)n * a.L + qi) * CC + head * HD;
            const float inv = K > 0 ? 1.0f / lsum : 0.f;       // K == 0: zeros (the caller skips the layer)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {             // registers 4*r4..4*r4+3 = 4 consecutive channels
                    const int d = b * 32 + 8 * r4 + 4 * h;
                    const v4f v{o[qb][b][4 * r4] * inv, o[qb][b][4 * r4 + 1] * inv, o[qb][b][4 * r4 + 2] * inv, o[qb][b][4 * r4 + 3] * inv};
                    if constexpr (F32) *reinterpret_cast<v4f*>(op + d) = v;
                    else *reinterpret_cast<gf_vec<T, 4>*>(op + d) = gf_vec<T, 4>{(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 5, the HEAD form (the default of the 16-bit modes).  Two measurements decide its structure:
//  * tools/probes/mfma_filler.hip: ONE wave's own stream covers up to five vector instructions (two exponentials) per v_mfma_f32_32x32x16 for nothing
//    (32.3 cycles per MFMA at one or two waves per SIMD), but an MFMA-only wave beside a vector-only wave on the same SIMD both crawl (76 cycles per
//    MFMA): a kernel whose waves alternate an MFMA phase and a softmax phase - attn_self - spends its time in exactly that pairing.  Here every wave's
//    tile is ONE pinned instruction stream in which the softmax of tile t sits between the MFMAs of S(t + 1) and of P(t).V(t):
//        S(t+1): 4 MFMAs, two exponentials of tile t behind each  |  P.V keys 0..15: 2 MFMAs, four exponentials behind each  |  P.V keys 16..31: 2 MFMAs
//    (left alone the compiler puts the sixteen exponentials in front and the eight MFMAs behind them; the order is pinned with sched_barrier);
//  * the four-head workgroup of that stream (one 32 KiB image of 32 keys per 64 queries, eight waves, one barrier domain per CU: 240 us where attn_self
//    takes 203; with four loader waves beside eight computing waves 258 us: the computing waves' tile costs ~1000 cycles, the 32 KiB of rows do not
//    arrive faster than every ~1700).  Here a workgroup owns ONE head: a tile's image is [32 keys][64 channels] of K and of V = 8 KiB, a quarter of the
//    stream per flop, requested as eight 1 KiB pieces - and FOUR waves (128 queries): three workgroups per CU (168 registers) are three barrier
//    domains, whose waves on a SIMD are in different phases of their tiles.  Eight waves x 256 queries (one domain per CU): 206 us; four waves: 183 us
//    (16 images x 6400 queries x 1195 keys, tools/k4_ab.py), 227 us against 257 at 1600 keys.
// NS images form the ring: tile t + NS - 1 is requested while tile t is computed, with the tokens read NS tiles before their request so that the counted
// wait at a tile's top (the counter is in-order) leaves all later requests in flight.  Arithmetic per query: attn_self's PRE / DIRECT form, bit for bit.
//   K image: row = key (128 B), 16-B chunk c at c ^ ((key >> 1) & 7)  (a ds_read_b128 lane group - 16 keys, one chunk - covers the 64 banks)
//   V image: row = key (128 B), 32-B block b (16 channels) at b ^ 2 ((key >> 1) & 1)  (a transposing read's 32 lanes - 4 keys x 2 blocks x 4 pieces -
//            cover the 64 banks); the fragment's second half is 8 keys = 1 KiB further
// Issue order of a tile: it opens with the exponentials of its first half (no LDS operand) under the fragment reads' latency, P.V keys 0..15 before
// the S chain, the requests inside the S chain.  (The variants measured in round 5 and not kept - eight waves, 6- / 8-image rings, the S chain first,
// the ablation instances - are in git history and profiles/r05_k4_experiments.txt.)
template <typename T>
__global__ __launch_bounds__(256, 3) void attn_self_head(AtArgs a) {
    constexpr int NS = 4, NW = 4;                                          // images in the ring, waves per workgroup
    using M = Mma32<T>;
    using Frag = typename M::Frag;
    constexpr int KG = M::kGroup, NG = HD / KG, D = NS - 1, QW = 32 * NW;
    constexpr bool BOTH = NW == 4;                                         // four waves: each requests its eight rows of K AND of V (one token read)
    static_assert(NW == 4 || NW == 8, "a tile's image is eight 1 KiB requests");
    constexpr int KBYTES = KT * HD * sizeof(T), IMG = 2 * KBYTES;                            // 4 KiB + 4 KiB
    static_assert(NG == 4 && sizeof(T) == 2 && KBYTES == 4096 && NS * IMG <= 65536, "written for 64-channel heads in 16-bit storage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NH = CC / HD, nqb = (a.L + QW - 1) / QW, gx = nqb * NH, total = gx * a.N, tid = threadIdx.x;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);           // whole images per XCD (see attn_self)
    const int n = id / gx, head = (id - n * gx) / nqb, q0 = (id - n * gx - head * nqb) * QW + (tid >> 6) * 32;
    const int lane = tid & 63, h = lane >> 5, lr = lane & 31;
    const int K = a.nkeys[(size_t)n * a.nkeys_stride];
    const int ntiles = (K + KT - 1) / KT;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    Frag qf[NG];
    {
        const int qrow = min(q0 + lr, a.L - 1);
        const T* qp = (const T*)a.q + ((size_t)n * a.L + qrow) * a.ldq + head * HD;
        const float c2 = a.softmax_temp * 1.44269504088896341f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            qf[g] = *reinterpret_cast<const Frag*>(qp + g * KG + h * (KG / 2));
#pragma unroll
            for (int j8 = 0; j8 < (int)(sizeof(Frag) / sizeof(T)); ++j8) qf[g][j8] = (T)(gf_to_float(qf[g][j8]) * c2);
        }
    }
    v16f o[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
    float m = 0.f, l = 0.f;
    // the wave's request: waves 0..3 the K rows 8 wv .. 8 wv + 7 of a tile, waves 4..7 the V rows 8 (wv - 4) ..; lane = (row lane >> 3, LDS chunk lane & 7).
    // STRUCTURED buffer over the map: record = one row, the lane's token is the record index (a record index of L or more - the padded keys of the
    // ragged tile get 0x7FFFFFFF - writes zeros), its logical 16-byte chunk of the head's 128 bytes the offset
    const bool isv = !BOTH && wv >= 4;
    const int rrow = 8 * (wv & 3) + (lane >> 3), rs = lane & 7;
    const int roffk = head * (HD * (int)sizeof(T)) + (rs ^ ((rrow >> 1) & 7)) * 16;
    const int roffv = head * (HD * (int)sizeof(T)) + ((((rs >> 1) ^ (((rrow >> 1) & 1) << 1)) << 1) | (rs & 1)) * 16;
    const int roff = isv ? roffv : roffk;
    const __amdgpu_buffer_rsrc_t rvmap =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)a.vmap + (size_t)n * a.L * a.ldv), (short)(a.ldv * sizeof(T)), a.L, 0x00020000);
    const __amdgpu_buffer_rsrc_t rkmap =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)a.kmap + (size_t)n * a.L * a.ldk), (short)(a.ldk * sizeof(T)), a.L, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmap = isv ? rvmap : rkmap;
    // the token list as a raw buffer of K words: tile * 128 bytes is the scalar offset, a key behind the list reads 0
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.idx + (size_t)n * a.idx_stride), 0, K * 4, 0x00020000);
    char* const rdst = smem + (isv ? KBYTES : 0) + (wv & 3) * 1024;
    int tk[NS];                                                            // token of the lane's row in tile j (mod NS), read NS tiles before its request
    auto load_tok = [&](int tile) { return (int)__builtin_amdgcn_raw_buffer_load_b32(ri, rrow * 4, tile * (KT * 4), 0); };
    auto request = [&](int slot, int tile, int t, bool ragged) {
        const int rec = ragged && tile * KT + rrow >= K ? 0x7FFFFFFF : t;
        if constexpr (BOTH) {
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rkmap, (__attribute__((address_space(3))) void*)(rdst + slot * IMG), 16, rec, roffk, 0, 0, 0);
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rvmap, (__attribute__((address_space(3))) void*)(rdst + KBYTES + slot * IMG), 16, rec, roffv, 0, 0, 0);
        } else {
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rmap, (__attribute__((address_space(3))) void*)(rdst + slot * IMG), 16, rec, roff, 0, 0, 0);
        }
    };
    int koff[NG], voff[2];
#pragma unroll
    for (int g = 0; g < NG; ++g) koff[g] = lr * 128 + (((g * 2 + h) ^ ((lr >> 1) & 7)) << 4);
    {
        const int G = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
        for (int b = 0; b < 2; ++b) voff[b] = KBYTES + (4 * (G >> 1) + qq) * 128 + (((b * 2 + (G & 1)) ^ (((qq >> 1) & 1) << 1)) << 5) + pp * 8;
    }
    // A operand of P.V from the row-major V image: keys 16 s2 + 8 (j >> 2) + 4 (lane >> 5) + (j & 3), the k order of the packed P^T (see k4_vtr_frag)
    auto v_frag = [&](const char* img, int b, int s2) {
        typedef __attribute__((address_space(3))) gf_v4s* LP;
        const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(img + voff[b] + s2 * 2048));
        const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(img + voff[b] + s2 * 2048 + 1024));
        typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
        const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(Frag, both);
    };
    v16f sc;
    float tmax = 0.f;
    auto tile_max = [&](v16f& s, int tile, bool ragged) {
        if (ragged && (tile + 1) * KT > K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = tile * KT + gf_acc_row(r, h) < K ? s[r] : -INFINITY;
        }
        float t = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) t = fmaxf(t, s[r]);
        return half_max(t);
    };
    if (ntiles > 0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) tk[j] = load_tok(j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < D; ++j) {
            request(j, j, tk[j], true);
            tk[j] = load_tok(j + NS);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        Frag kf[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) kf[g] = *reinterpret_cast<const Frag*>(smem + koff[g]);
        sc = negm;
#pragma unroll
        for (int g = 0; g < NG; ++g) M::mma(kf[g], qf[g], sc);
        tmax = tile_max(sc, 0, true);
    }
    typedef float f2 __attribute__((ext_vector_type(2)));
    // TAIL: the bodies in which tile t + 1 (its mask is a branch) or the requested tile t + D (its padded keys: a select per request) may be the ragged last tile
    auto body = [&](auto slot_c, auto tail_c, int tile) {
        constexpr bool TAIL = decltype(tail_c)::value;
        constexpr int SLOT = decltype(slot_c)::value, NSLOT = (SLOT + 1) % NS, FSLOT = (SLOT + D) % NS;
        K4_T(0);
        if (tile > 0) {
            // tile + 1's rows (requested D - 1 tiles ago) have landed - this wave's; the requests and token reads issued since stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BOTH ? 3 * D - 5 : 2 * D - 3) : "memory");
            K4_T(1);
            __syncthreads();
        }
        K4_T(2);
        // ---- deferred reference (attn_self's rule): up by d = the tile's maximum when that is more than K4_DEFER above it, and in the first tile
        const bool need = tmax > K4_DEFER || tile == 0;
        if (__any(need)) {
            const float d = need ? tmax : 0.f;
            const float alpha = tile == 0 ? 0.f : __builtin_amdgcn_exp2f(-d);
            l *= alpha;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            m += d;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                negm[r] = -m;
                sc[r] -= d;
            }
        }
        // ---- one basic block, its issue order pinned segment by segment (sched_barrier): every MFMA is followed by the two to four exponentials and
        // the few vector instructions its 32 cycles cover (left alone the compiler puts the sixteen exponentials in front and the eight MFMAs behind)
        K4_T(3);
        const char* cur = smem + SLOT * IMG;
        const char* nxt = smem + NSLOT * IMG;
        Frag kf[NG], vf[2][2];
        float x[16];
        f2 ps2{0.f, 0.f};
        auto ex = [&](int r0, int r1) {
#pragma unroll
            for (int r = r0; r < r1; ++r) x[r] = __builtin_amdgcn_exp2f(sc[r]);
        };
        auto sum = [&](int r0, int r1) {
#pragma unroll
            for (int r = r0; r < r1; r += 2) ps2 += f2{x[r], x[r + 1]};
        };
        auto mm = [&](const Frag& x_, const Frag& y_, v16f& c_) { M::mma(x_, y_, c_); };
        v16f sn;
        {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 2; ++b) vf[0][b] = v_frag(cur, b, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) kf[g] = *reinterpret_cast<const Frag*>(nxt + koff[g]);
#pragma unroll
            for (int b = 0; b < 2; ++b) vf[1][b] = v_frag(cur, b, 1);
            sn = negm;
            __builtin_amdgcn_sched_barrier(0);
            ex(0, 8);
            sum(0, 8);
            const Frag pf0{(T)x[0], (T)x[1], (T)x[2], (T)x[3], (T)x[4], (T)x[5], (T)x[6], (T)x[7]};
            __builtin_amdgcn_sched_barrier(0);
            K4_T(4);
            mm(vf[0][0], pf0, o[0]);
            ex(8, 10);
            __builtin_amdgcn_sched_barrier(0);
            mm(vf[0][1], pf0, o[1]);
            ex(10, 12);
            __builtin_amdgcn_sched_barrier(0);
            mm(kf[0], qf[0], sn);
            ex(12, 14);
            request(FSLOT, tile + D, tk[FSLOT], TAIL);                      // tile + D into tile - 1's image, with the token read NS tiles ago;
            tk[FSLOT] = load_tok(tile + D + NS);                            // then the token of tile + D + NS into its place
            __builtin_amdgcn_sched_barrier(0);
            mm(kf[1], qf[1], sn);
            ex(14, 16);
            __builtin_amdgcn_sched_barrier(0);
            mm(kf[2], qf[2], sn);
            const Frag pf1{(T)x[8], (T)x[9], (T)x[10], (T)x[11], (T)x[12], (T)x[13], (T)x[14], (T)x[15]};
            sum(8, 16);
            __builtin_amdgcn_sched_barrier(0);
            mm(kf[3], qf[3], sn);
            __builtin_amdgcn_sched_barrier(0);
            mm(vf[1][0], pf1, o[0]);
            __builtin_amdgcn_sched_barrier(0);
            mm(vf[1][1], pf1, o[1]);
        }
        K4_T(5);
        l += ps2.x + ps2.y;
        tmax = tile_max(sn, tile + 1, TAIL);             // (behind the last tile: a value nobody uses)
        sc = sn;
        K4_T(6);
    };
    using std::integral_constant;
    using std::false_type;
    using std::true_type;
    auto round = [&](auto tail_c, int tile) {                              // NS bodies, those inside the list
        auto go = [&](auto self, auto j_c) {
            constexpr int J = decltype(j_c)::value;
            if constexpr (J < NS) {
                if (!decltype(tail_c)::value || tile + J < ntiles) body(integral_constant<int, J>{}, tail_c, tile + J);
                self(self, integral_constant<int, J + 1>{});
            }
        };
        go(go, integral_constant<int, 0>{});
    };
    int tile = 0;
    for (; tile + NS + D <= K / KT; tile += NS) round(false_type{}, tile);  // tiles t + 1 .. t + D of these are full tiles
    for (; tile < ntiles; tile += NS) round(true_type{}, tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // nothing of the ring in flight behind the kernel's LDS
    const float lsum = l + __shfl_xor(l, 32, 64);
    const int qi = q0 + lr;
    if (qi < a.L) {
        T* op = (T*)a.out + ((size_t)n * a.L + qi) * CC + head * HD;
        const float inv = K > 0 ? 1.0f / lsum : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = b * 32 + 8 * r4 + 4 * h;
                *reinterpret_cast<gf_vec<T, 4>*>(op + d) = gf_vec<T, 4>{(T)(o[b][4 * r4] * inv), (T)(o[b][4 * r4 + 1] * inv),
                                                                        (T)(o[b][4 * r4 + 2] * inv), (T)(o[b][4 * r4 + 3] * inv)};
            }
    }
}

}   // namespace

#if K4_TRACE
extern "C" int gf_debug_k4_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k4_trace), sizeof(long long) * 2048 * 32);
}
#endif

namespace {
template <typename T>
void k4_launch_head(const AtArgs& a, hipStream_t st) {
    if constexpr (std::is_same<T, float>::value) {
        (void)a; (void)st;
    } else {
        const int blocks = ((a.L + 127) / 128) * (CC / HD) * a.N;
        attn_self_head<T><<<blocks, 256, 4 * 2 * KT * HD * 2, st>>>(a);     // four (K, V) images of 32 keys x 64 channels
    }
}
}   // namespace

extern "C" size_t gf_self_attention_workspace_bytes(int N, int L, int dtype) {
    if (N <= 0 || L <= 0) return 0;
    const size_t kpad = gf_align_up((size_t)L, KT);
    const size_t es = dtype == GF_F32 ? 4 : 2;
    return 2 * gf_align_up((size_t)N * kpad * CC * es, 256);
}

extern "C" int gf_self_attention_gathered(const void* q, const void* kmap, const void* vmap, int dtype, int N, int L,
                                          int H, int D, long ldq, long ldk, long ldv, const int32_t* idx,
                                          long idx_stride, const int32_t* nkeys, int nkeys_stride, void* out,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(q && kmap && vmap && idx && nkeys && out, "null pointer");
    GF_CHECK_ARG(N > 0 && L > 0, "empty problem");
    GF_CHECK_ARG(H == NH && D == HD, "built for nhead=4, head dim 64 (geo_config.py:12, d_model 256)");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(dtype == GF_F32 || ((uintptr_t)kmap % 16 == 0 && ldk % 8 == 0 && (uintptr_t)q % 16 == 0 && ldq % 8 == 0),
                 "16-bit modes move key / query rows as 16-byte pieces: 16-byte aligned maps, row strides that are multiples of 8 elements");
    if (workspace == nullptr || workspace_bytes < gf_self_attention_workspace_bytes(N, L, dtype)) {
        gf_set_error("gf_self_attention_gathered: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    AtArgs a;
    a.q = q; a.kmap = kmap; a.vmap = vmap; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.idx = idx; a.idx_stride = idx_stride; a.nkeys = nkeys; a.nkeys_stride = nkeys_stride; a.out = out;
    a.N = N; a.L = L; a.Kpad = (int)gf_align_up((size_t)L, KT);
    const size_t es = dtype == GF_F32 ? 4 : 2;
    a.kc = workspace;
    a.vc = (char*)workspace + gf_align_up((size_t)N * a.Kpad * CC * es, 256);
    a.softmax_temp = 1.0f / sqrtf((float)D);
    hipStream_t st = (hipStream_t)stream;
    // The head form (16-bit modes): one head and 128 queries per workgroup, K / V rows straight from the projected maps through structured
    // buffers - it needs 16-byte aligned rows whose stride fits the descriptor's 14-bit stride field.  Everything else - the fp32 parity
    // mode, unaligned or very wide rows - takes the gather pass + the four-wave, four-head form (attn_self; QB query blocks per wave once
    // there are enough workgroups to fill the chip), which computes the same arithmetic per query (same bits in the 16-bit modes).
    const bool use_head = dtype != GF_F32 && (uintptr_t)vmap % 16 == 0 && ldv % 8 == 0 && (size_t)(ldk > ldv ? ldk : ldv) * 2 < 16384 &&
                          (size_t)L * (size_t)(ldk > ldv ? ldk : ldv) * 2 < 0x7FFFFFF0ull;
    const int qb = (long)N * ((L + 63) / 64) >= 512 ? 2 : 1;
    const dim3 ggrid(dtype == GF_F32 ? a.Kpad / KT : (a.Kpad / 8 < 64 ? a.Kpad / 8 : 64), N), agrid((L + 32 * qb - 1) / (32 * qb), N);
#define GF_K4_LAUNCH(T, ES)                                                                        \
    do {                                                                                           \
        const size_t LDSB = (ES == 4 ? 2 : 4) * KT * CC * ES;      /* 16-bit: two (K, V^T) images */       \
        if (use_head) { k4_launch_head<T>(a, st); break; }                                         \
        attn_gather_kv<T><<<ggrid, 256, 0, st>>>(a);                                               \
        if (qb == 2) attn_self<T, 2><<<agrid, 256, LDSB, st>>>(a);                              \
        else attn_self<T, 1><<<agrid, 256, LDSB, st>>>(a);                                      \
    } while (0)
    // the key counts live on the device: the caller that knows them (bench.py reads them back) declares the work, 4 L K C flops per sample
    void* pt = gf_prof_begin("k4_self_attention", st, 0.0);
    if (dtype == GF_F32) GF_K4_LAUNCH(float, 4);
    else if (dtype == GF_F16) GF_K4_LAUNCH(_Float16, 2);
    else GF_K4_LAUNCH(gf_bf16, 2);
    gf_prof_end("k4_self_attention", pt, st);
#undef GF_K4_LAUNCH
    GF_CHECK_LAUNCH();
    return GF_OK;
}
