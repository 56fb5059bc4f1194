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
    const int n = blockIdx.y, p0 = blockIdx.x * KT, t = threadIdx.x;
    const int K = a.nkeys[(size_t)n * a.nkeys_stride];
    if (p0 >= ((K + KT - 1) / KT) * KT) return;
    const int32_t* idx = a.idx + (size_t)n * a.idx_stride;
    const T* km = (const T*)a.kmap + (size_t)n * a.L * a.ldk;
    const T* vm = (const T*)a.vmap + (size_t)n * a.L * a.ldv;
    T* kc = (T*)a.kc + ((size_t)n * a.Kpad + p0) * CC;
    if constexpr (std::is_same<T, float>::value) {
        T* vc = (T*)a.vc + ((size_t)n * a.Kpad + p0) * CC;
        for (int p = 0; p < KT; ++p) {
            const bool ok = p0 + p < K;
            const int tok = ok ? idx[p0 + p] : 0;
            kc[(size_t)p * CC + t] = ok ? km[(size_t)tok * a.ldk + t] : 0.f;
            vc[(size_t)p * CC + t] = ok ? vm[(size_t)tok * a.ldv + t] : 0.f;
        }
    } else {
        T vt[KT];
#pragma unroll
        for (int p = 0; p < KT; ++p) {
            const bool ok = p0 + p < K;
            const int tok = ok ? idx[p0 + p] : 0;
            kc[(size_t)p * CC + t] = ok ? km[(size_t)tok * a.ldk + t] : (T)0;
            vt[vt_pos(p)] = ok ? vm[(size_t)tok * a.ldv + t] : (T)0;
        }
        using V8 = gf_vec<T, 8>;
        V8* dst = reinterpret_cast<V8*>((T*)a.vc + ((size_t)n * CC + t) * a.Kpad + p0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            dst[c] = V8{vt[8 * c], vt[8 * c + 1], vt[8 * c + 2], vt[8 * c + 3], vt[8 * c + 4], vt[8 * c + 5], vt[8 * c + 6], vt[8 * c + 7]};
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

template <typename T, int QB>
__global__ __launch_bounds__(256, QB <= 2 ? 2 : 1) void attn_self(AtArgs a) {
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
    const int n = id / gx, q0 = (id - n * gx) * (32 * QB), tid = threadIdx.x;
    const int head = tid >> 6, lane = tid & 63, h = lane >> 5, lr = lane & 31;
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
    v16f o[QB][2];
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = -INFINITY;
        l[qb] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][b][r] = 0.f;
    }
    const T* kc = (const T*)a.kc + (size_t)n * a.Kpad * CC;
    const int ntiles = (K + KT - 1) / KT;
    // logits in log2 units: exp(x - m) = exp2(x' - m') with x' = s * (temp * log2 e): one multiply per element instead of two
    constexpr bool FAST = !std::is_same<T, float>::value;              // 16-bit modes: hardware exponential
    const float scale2 = FAST ? a.softmax_temp * 1.44269504088896341f : a.softmax_temp;
    auto compute = [&](int tile, const char* ks, const char* vs) {
        const bool ragged = (tile + 1) * KT > K;          // only the last tile can hold key slots beyond K
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // ---- S^T tile: rows = keys (registers), column = query (lane)
            v16f s;
    #pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
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
            if constexpr (FAST) {
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
                const float mnew = fmaxf(m[qb], tmax);          // finite: every tile holds at least one real key
                const float nms = -mnew * scale2;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    x[r] = __builtin_amdgcn_exp2f(fmaf(x[r], scale2, nms));
                    psum += x[r];
                }
                if (__any(mnew != m[qb])) {                     // a running maximum moved somewhere in the wave: rescale (else alpha = 1)
                    const float alpha = __builtin_amdgcn_exp2f((m[qb] - mnew) * scale2);
                    l[qb] *= alpha;
    #pragma unroll
                    for (int b = 0; b < 2; ++b)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][b][r] *= alpha;
                }
                m[qb] = mnew;
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
                        const Frag vf = *reinterpret_cast<const Frag*>(vs + vt_off(head * HD + b * 32 + lr, 2 * s2 + h));
                        M::mma(vf, pf, o[qb][b]);
                    }
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
        const AtRsrc rk = at_rsrc(kc, (unsigned)((size_t)a.Kpad * CC * sizeof(T)));
        const AtRsrc rv = at_rsrc((const T*)a.vc + (size_t)n * CC * a.Kpad, (unsigned)((size_t)CC * a.Kpad * sizeof(T)));
        // K image: 1 KiB group g = key rows 2g, 2g+1; slot s of row r holds chunk s ^ (r & 15) (k_off)
        // V^T image: group g = channels 16g..16g+15, 4 slots of 16 B; slot s of channel c holds chunk s ^ ((c >> 2) & 3) (vt_off)
        int kvo[4], vvo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wv * 4 + i;
            const int krow = 2 * g + (lane >> 5), kslot = lane & 31;
            kvo[i] = krow * (CC * (int)sizeof(T)) + ((kslot ^ (krow & 15)) << 4);
            const int vrow = 16 * g + (lane >> 2), vslot = lane & 3;
            vvo[i] = vrow * (a.Kpad * (int)sizeof(T)) + ((vslot ^ ((vrow >> 2) & 3)) << 4);
        }
        auto request = [&](int tile) {
            char* img = smem + (tile & 1) * (2 * KBYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) at_lds_dma(rk, img + (wv * 4 + i) * 1024, kvo[i], tile * KBYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) at_lds_dma(rv, img + KBYTES + (wv * 4 + i) * 1024, vvo[i], tile * (KT * (int)sizeof(T)));
        };
        if (ntiles > 0) request(0);
        for (int tile = 0; tile < ntiles; ++tile) {
            K4_T(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            K4_T(1);
            __syncthreads();
            K4_T(2);
            if (tile + 1 < ntiles) request(tile + 1);
            const char* img = smem + (tile & 1) * (2 * KBYTES);
            compute(tile, img, img + KBYTES);
        }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lsum = l[qb] + __shfl_xor(l[qb], 32, 64);
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

}   // namespace

#if K4_TRACE
extern "C" int gf_debug_k4_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k4_trace), sizeof(long long) * 2048 * 32);
}
#endif

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
    // query blocks per wave (32 QB queries share a staged K / V tile) once there are enough workgroups to fill the chip;
    // GF_K4_QB=1|2|4 overrides (measurements)
    static const int forced = [] { const char* e = getenv("GF_K4_QB"); return e ? atoi(e) : 0; }();
    int qb = forced ? forced : ((long)N * ((L + 63) / 64) >= 512 ? 2 : 1);
    if (qb != 1 && qb != 2 && qb != 4) qb = 1;
    const dim3 ggrid(a.Kpad / KT, N), agrid((L + 32 * qb - 1) / (32 * qb), N);
#define GF_K4_LAUNCH(T, ES)                                                                        \
    do {                                                                                           \
        const size_t LDSB = (ES == 4 ? 2 : 4) * KT * CC * ES;      /* 16-bit: two (K, V^T) images */       \
        attn_gather_kv<T><<<ggrid, 256, 0, st>>>(a);                                               \
        if (qb == 4) attn_self<T, 4><<<agrid, 256, LDSB, st>>>(a);                     \
        else if (qb == 2) attn_self<T, 2><<<agrid, 256, LDSB, st>>>(a);                \
        else attn_self<T, 1><<<agrid, 256, LDSB, st>>>(a);                             \
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
