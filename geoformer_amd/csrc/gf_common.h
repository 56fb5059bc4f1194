// Shared device/host helpers for the GeoFormer HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/geoformer_hip.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __bf16 gf_bf16;                                                     // bfloat16 storage (GF_BF16)
typedef gf_bf16 v8b __attribute__((ext_vector_type(8)));
typedef gf_bf16 v4b __attribute__((ext_vector_type(4)));
template <typename T, int N>
using gf_vec = T __attribute__((ext_vector_type(N)));                        // N elements of a storage type
typedef short gf_v4s __attribute__((__vector_size__(4 * sizeof(short))));   // result of ds_read_b64_tr_b16

// ---------------------------------------------------------------------------------------------
// error plumbing: every extern "C" entry returns 0 or a negative code and never throws
// ---------------------------------------------------------------------------------------------
void gf_set_error(const char* fmt, ...);

#define GF_CHECK_ARG(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            gf_set_error("%s: %s", __func__, msg);                \
            return GF_ERR_INVALID_ARGUMENT;                       \
        }                                                         \
    } while (0)

#define GF_CHECK_LAUNCH()                                                          \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            gf_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return GF_ERR_LAUNCH;                                                  \
        }                                                                          \
    } while (0)

// optional event timing of single kernels (gf_runtime.hip); no-ops unless gf_profile_enable(1)
void* gf_prof_begin(const char* tag, hipStream_t st, double work = 0.0);
void gf_prof_end(const char* tag, void* token, hipStream_t st);

static inline size_t gf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: launch sites that opt a kernel into
// > 64 KiB of LDS do it once per device, not once per process (a process may run the model on cuda:0 and later on cuda:1;
// the bench's pipeline threads race here, hence the atomic).  Returns true exactly once per (flag word, device).
#include <atomic>
static inline bool gf_first_use_on_device(std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;      // unknown device: set the attribute again
    const uint64_t bit = 1ull << dev;
    return (done.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}

// carve helper for caller-provided workspaces
struct GfCarver {
    char* base;
    size_t off;
    explicit GfCarver(void* p) : base((char*)p), off(0) {}
    template <typename T>
    T* take(size_t n) {
        off = gf_align_up(off, 256);
        T* r = (T*)(base + off);
        off += n * sizeof(T);
        return r;
    }
    size_t used() const { return gf_align_up(off, 256); }
};

// ---------------------------------------------------------------------------------------------
// element types.  Kernels are templated on the storage type; arithmetic is always fp32.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
    static constexpr int kDtype = GF_F32;
    static constexpr int kPer16B = 4;
};
template <>
struct ElemTraits<_Float16> {
    static constexpr int kDtype = GF_F16;
    static constexpr int kPer16B = 8;
};
template <>
struct ElemTraits<gf_bf16> {
    static constexpr int kDtype = GF_BF16;
    static constexpr int kPer16B = 8;
};

__device__ __forceinline__ float gf_to_float(float x) { return x; }
__device__ __forceinline__ float gf_to_float(_Float16 x) { return (float)x; }
__device__ __forceinline__ float gf_to_float(gf_bf16 x) { return (float)x; }
template <typename T>
__device__ __forceinline__ T gf_from_float(float x);
template <>
__device__ __forceinline__ float gf_from_float<float>(float x) { return x; }
template <>
__device__ __forceinline__ _Float16 gf_from_float<_Float16>(float x) { return (_Float16)x; }
template <>
__device__ __forceinline__ gf_bf16 gf_from_float<gf_bf16>(float x) { return (gf_bf16)x; }       // round to nearest even (v_cvt_pk_bf16_f32)

// ---------------------------------------------------------------------------------------------
// MFMA 32x32 wrappers.  One "k-group" = the K range covered by one 16-byte fragment per lane:
//   f16: 16 elements, ONE v_mfma_f32_32x32x16_f16   (lane l: row/col l&31, k = 8*(l>>5)+j)
//   f32:  8 elements, FOUR v_mfma_f32_32x32x2_f32   (lane l: row/col l&31, k = l>>5; MFMA t takes
//         element t of both fragments, i.e. k = t and k = 4+t: any k permutation is legal as long
//         as A and B use the same one)
// C/D layout (dtype independent): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Mma32;
template <>
struct Mma32<float> {
    using Frag = v4f;
    static constexpr int kGroup = 8;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, v16f& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    }
};
template <>
struct Mma32<_Float16> {
    using Frag = v8h;
    static constexpr int kGroup = 16;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, v16f& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <>
struct Mma32<gf_bf16> {
    using Frag = v8b;
    static constexpr int kGroup = 16;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, v16f& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// 16x16x32 MFMA (gfx950): A = 16 rows x 32 k, B = 32 k x 16 columns; lane l holds row / column l % 16 and the 8 consecutive
// k values 8 (l / 16) .. + 7; the accumulator: column l % 16, rows 4 (l / 16) + i in register i.  Per FLOP the same cycles and
// operand bytes as the 32x32x16 form, but the chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS item 7).
template <typename T>
struct Mma16;
template <>
struct Mma16<_Float16> {
    using Frag = v8h;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, v4f& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Mma16<gf_bf16> {
    using Frag = v8b;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, v4f& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// row of accumulator register r for lane-half h inside a 32x32 MFMA tile
__device__ __forceinline__ int gf_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------
// LDS operand tile: [rows][128 B], 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7).
// With 128-B rows a ds_read_b128 lane group (16 lanes, distinct rows, same logical chunk) then
// touches 16 distinct 16-B slots of the 256-B bank row: conflict-free (guide T2, adapted to 128 B).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int gf_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------
// cross-lane reduce-scatter over the 32 lanes of each wave half, entirely on the VALU (no LDS
// round trips: the ds_bpermute form measured ~60 dependent LDS latencies per call).
// In: v[q], q = 0..31 (a per-lane array of partials for 32 "slots").  Out: lane c (= lane&31)
// returns the op-reduction over the 32 lanes of its half of slot q = c.  31 exchanges:
//   d = 16 : v_permlane16_swap_b32 exchanges the odd 16-lane rows of one register with the even rows
//            of the other, which IS the keep/send selection - one swap + one op per slot pair;
//   d = 8, 2, 1 : blend (v_bfi) keep/send, fetch the partner with one DPP mov (row_ror:8, quad_perm);
//   d = 4  : two bank-masked DPP movs (row_ror:12 for lanes with bit 2 clear, row_ror:4 for the rest).
// Lane mappings verified on gfx950 with tools/probes/dpp_probe.hip.
// ---------------------------------------------------------------------------------------------
typedef unsigned gf_v2u __attribute__((ext_vector_type(2)));

// Lane-dependent choice between two registers as a bit blend (v_bfi_b32).  A `cond ? v[i+d] : v[i]`
// select gets rewritten by LLVM into a dynamically indexed register array, which it then lowers to
// a 32-way compare/select chain per access (measured: 3,800 v_cmp + 7,500 SGPR spills).
__device__ __forceinline__ unsigned gf_blend_u(unsigned m, unsigned a, unsigned b) { return (a & m) | (b & ~m); }

template <int D>
__device__ __forceinline__ unsigned gf_fetch_xor(unsigned x) {   // value of lane ^ D, D in {1, 2, 4, 8}
    if constexpr (D == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);
    else if constexpr (D == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);
    else if constexpr (D == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);
    else {
        const int lo = __builtin_amdgcn_update_dpp(0, (int)x, 0x12C, 0xF, 0x5, false);     // banks 0,2 <- lane+4
        return (unsigned)__builtin_amdgcn_update_dpp(lo, (int)x, 0x124, 0xF, 0xA, false);   // banks 1,3 <- lane-4
    }
}

struct GfBitsF {      // float <-> register words
    static constexpr int W = 1;
    static __device__ __forceinline__ void unpack(float v, unsigned (&w)[1]) { w[0] = __float_as_uint(v); }
    static __device__ __forceinline__ float pack(const unsigned (&w)[1]) { return __uint_as_float(w[0]); }
};
struct GfBitsU64 {
    static constexpr int W = 2;
    static __device__ __forceinline__ void unpack(unsigned long long v, unsigned (&w)[2]) { w[0] = (unsigned)v; w[1] = (unsigned)(v >> 32); }
    static __device__ __forceinline__ unsigned long long pack(const unsigned (&w)[2]) { return ((unsigned long long)w[1] << 32) | w[0]; }
};
template <typename V> struct GfBits;
template <> struct GfBits<float> : GfBitsF {};
template <> struct GfBits<unsigned long long> : GfBitsU64 {};

template <int D, typename V, typename Op>
__device__ __forceinline__ void gf_rs_stage(V (&v)[32], unsigned up, Op op) {
    using B = GfBits<V>;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        unsigned a[B::W], b[B::W], keep[B::W], recv[B::W];
        B::unpack(v[i], a);
        B::unpack(v[i + D], b);
#pragma unroll
        for (int k = 0; k < B::W; ++k) {
            keep[k] = gf_blend_u(up, b[k], a[k]);
            recv[k] = gf_fetch_xor<D>(gf_blend_u(up, a[k], b[k]));
        }
        v[i] = op(B::pack(keep), B::pack(recv));
    }
}

template <typename V, typename Op>
__device__ __forceinline__ V gf_reduce_scatter32(V (&v)[32], Op op) {
    using B = GfBits<V>;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 16; ++i) {                 // d = 16
        unsigned a[B::W], b[B::W], x[B::W], y[B::W];
        B::unpack(v[i], a);
        B::unpack(v[i + 16], b);
#pragma unroll
        for (int k = 0; k < B::W; ++k) {
            const gf_v2u r = __builtin_amdgcn_permlane16_swap(a[k], b[k], false, false);
            x[k] = r.x;
            y[k] = r.y;
        }
        v[i] = op(B::pack(x), B::pack(y));
    }
    unsigned up8 = (unsigned)(-(int)((lane >> 3) & 1)), up4 = (unsigned)(-(int)((lane >> 2) & 1));
    unsigned up2 = (unsigned)(-(int)((lane >> 1) & 1)), up1 = (unsigned)(-(int)(lane & 1));
    asm volatile("" : "+v"(up8), "+v"(up4), "+v"(up2), "+v"(up1));
    gf_rs_stage<8>(v, up8, op);
    gf_rs_stage<4>(v, up4, op);
    gf_rs_stage<2>(v, up2, op);
    gf_rs_stage<1>(v, up1, op);
    return v[0];
}

// value of lane ^ 16 (the other 16-lane row of the same wave half)
__device__ __forceinline__ float gf_shfl_xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ unsigned long long gf_shfl_xor16(unsigned long long v) {
    const unsigned lo = __shfl_xor((unsigned)v, 16, 64), hi = __shfl_xor((unsigned)(v >> 32), 16, 64);
    return ((unsigned long long)hi << 32) | lo;
}

struct GfMaxF {
    __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};
struct GfAddF {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct GfMaxU64 {
    __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const {
        return a > b ? a : b;
    }
};
