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
void* gf_prof_begin(const char* tag, hipStream_t st);
void gf_prof_end(const char* tag, void* token, hipStream_t st);

static inline size_t gf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

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

__device__ __forceinline__ float gf_to_float(float x) { return x; }
__device__ __forceinline__ float gf_to_float(_Float16 x) { return (float)x; }
template <typename T>
__device__ __forceinline__ T gf_from_float(float x);
template <>
__device__ __forceinline__ float gf_from_float<float>(float x) { return x; }
template <>
__device__ __forceinline__ _Float16 gf_from_float<_Float16>(float x) { return (_Float16)x; }

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

// row of accumulator register r for lane-half h inside a 32x32 MFMA tile
__device__ __forceinline__ int gf_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------
// LDS operand tile: [rows][128 B], 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7).
// With 128-B rows a ds_read_b128 lane group (16 lanes, distinct rows, same logical chunk) then
// touches 16 distinct 16-B slots of the 256-B bank row: conflict-free (guide T2, adapted to 128 B).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int gf_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------
// cross-lane reduce-scatter over the 32 lanes of each wave half.
// In: v[q], q = 0..31 (a per-lane array of partials for 32 "slots").  Out: lane c (= lane&31)
// returns op-reduction over the 32 lanes of its half of slot q = c.  31 exchanges instead of the
// 160 a butterfly per slot would need.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gf_shfl_xor(float v, int d) { return __shfl_xor(v, d, 64); }
__device__ __forceinline__ unsigned long long gf_shfl_xor(unsigned long long v, int d) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, d, 64);
    hi = __shfl_xor(hi, d, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// Lane-dependent choice between two registers as a bit blend (v_bfi_b32).  A `cond ? v[i+d] : v[i]`
// select gets rewritten by LLVM into a dynamically indexed register array, which it then lowers to
// a 32-way compare/select chain per access (measured: 3,800 v_cmp + 7,500 SGPR spills).
__device__ __forceinline__ float gf_blend(unsigned m, float a, float b) {   // m all-ones -> a, zero -> b
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}
__device__ __forceinline__ unsigned long long gf_blend(unsigned m, unsigned long long a, unsigned long long b) {
    const unsigned long long mm = ((unsigned long long)m << 32) | m;
    return (a & mm) | (b & ~mm);
}

template <typename V, typename Op>
__device__ __forceinline__ V gf_reduce_scatter32(V (&v)[32], Op op) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 4; s >= 0; --s) {
        const int d = 1 << s;
        unsigned up = (unsigned)(-(int)((lane >> s) & 1));   // all-ones in the upper lane of each pair
        asm volatile("" : "+v"(up));
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const V keep = gf_blend(up, v[i + d], v[i]);
            const V send = gf_blend(up, v[i], v[i + d]);
            v[i] = op(keep, gf_shfl_xor(send, d));
        }
    }
    return v[0];
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
