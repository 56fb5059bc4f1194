// K9: the encoder layer as TWO launches that touch HBM three times per token (read source, read x, write out) -
// the algorithmic minimum of SURVEY 8d - instead of the eight launches / ~19 token-row transfers of the
// K3 + K2 chain.  16-bit storage modes only (fp16 / bf16 operands, fp32 accumulation and statistics); the fp32 parity
// mode keeps the unfused kernels.
//
// Replaces LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:37-60, ReLU, linear
// attention of linear_attention.py:21-51) and the part of its Geo twin after the attention
// (model/geo_transformer/transformer.py:56-66, Tanh):
//
//   enc_kv_state   k, v = W_k src, W_v src ; KV[n,h] = sum_s phi(k_s)^T v_s ; Ksum[n,h] = sum_s phi(k_s)
//                  per 128-token tile -> one fp32 partial state per tile ; enc_kv_reduce sums the tiles of an image.
//                  k and v never exist in HBM.
//   enc_layer      q = W_q x ; msg = phi(q) KV / (phi(q).Ksum + eps)            (ATTN: linear attention)
//                  or msg = attention output read from HBM                        (Geo layers: K4 / K5 made it)
//                  m = LN1(W_m msg) ; hid = act(W_1 [x | m]) ; out = x + LN2(W_2 hid)
//
// One workgroup = 4 waves = 128 tokens, one wave per SIMD with the whole 512-register file; a wave owns 32 tokens
// for the entire chain.  Products are computed transposed (MFMA A = 32 weight rows, B = 32 tokens), so a result
// tile has the token on the lane and 32 channels in 16 registers per lane half - and that IS the B operand of
// the next product (contraction over channels = over the tile's row index): 8 registers are packed to 16 bits
// per 16-deep k-step and the next weight's A fragment is stored with the matching k order
// (c = 32t + 16s + 8(j>>2) + 4h + (j&3) for element j of lane half h).  Activations therefore never pass through
// LDS between the five GEMMs; LayerNorm, phi, the activation and the residual are lane-local.
//
// Weights are pre-packed on the host (geoformer_amd/fused.py) into the exact sequence of 1-KiB MFMA A fragments
// the kernel consumes (`wstream`), 32 fragments = one 32-KiB block; every workgroup streams the same 1 MiB per
// layer from L2 into a two-block LDS ring with LDS-DMA (global_load_lds_dwordx4: the fragment order makes the
// LDS image lane-linear, so fragment reads are conflict-free ds_read_b128), one barrier per block of 32 MFMAs
// per wave.  The token tile x stays in LDS for the whole kernel (B operand of the q and mlp.0 products, residual).
#include <math.h>

#include <type_traits>

#include "gf_common.h"
#include "k9_args.h"

// round 6: the channel-split wave-pair kernels (k9_encoder_pair.hip) behind the same entry points; GF_K9_PAIR=0 selects this file's
// four-wave kernels (development A/B only - the two forms consume DIFFERENT weight-stream orders, fused.py follows the same switch)
void gf_k9_pair_layer(const GfEncArgs& a, int act, bool attn, int dtype, hipStream_t st);
void gf_k9_pair_state(const GfEncArgs& a, int dtype, hipStream_t st);
void gf_k9_pair_reduce(const float* part, float* fin, int tiles, int len, int n, hipStream_t st);
#include <stdlib.h>
static bool k9_pair() {
    static const bool on = [] { const char* e = getenv("GF_K9_PAIR"); return e == nullptr || e[0] != '0'; }();
    return on;
}

// -DK9_TRACE=1 records s_memtime at the phase boundaries of every workgroup (tools/k9_trace.py reads them); a
// diagnostic build only: the stamps serialise the wave, so only the SHARES of the phases are meaningful.
#ifndef K9_TRACE
#define K9_TRACE 0
#endif
// Round-5 experiments, both bit-identical to the default and both SLOWER (same box, tools/k9_digest.py, 16 images: layer alone 145.8-148.5 us,
// with the state tail 202.6-207.4 us):
//   -DK9_HEADPIPE=1 (+ GF_K9_HEADPIPE=1 at pack time: the weight stream head-major, fused.py) - q projection, attention and merge as one
//       head-pipelined stream (attention of head h inside the q-projection steps of head h + 1): 158.1 / 215.2 us;
//   -DK9_MERGEPIPE=1 - the attention of head h + 1 inside the two merge steps of head h (the stream's order unchanged): 156.0 / 214.9 us.
// The vector work placed inside the ring steps stretches their MFMA gaps by more than the phase it removes.
#ifndef K9_HEADPIPE
#define K9_HEADPIPE 0
#endif
#ifndef K9_MERGEPIPE
#define K9_MERGEPIPE 0
#endif
#if K9_TRACE
__device__ long long k9_trace[4096 * 4 * 16];
#define K9_T(slot) do { if ((slot) < 16 && lane == 0 && blockIdx.x < 4096) k9_trace[(blockIdx.x * 4 + wave) * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int gf_debug_k9_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k9_trace), sizeof(long long) * 4096 * 4 * 16);
}
#else
#define K9_T(slot)
#endif

namespace {

constexpr int C = 256, D = 32, NH = 8;            // coarse level: d_model 256, 8 heads of 32
constexpr int TM = 128;                           // tokens per workgroup
constexpr int WBLK = 32768, FRAG = 1024;          // one weight block = 32 fragments of 64 lanes x 16 B
constexpr int X_OFF = 0;                          // [4 k-chunks][128 rows][128 B], chunk-swizzled (gf_lds_off)
constexpr int W_OFF = 65536;                      // two weight blocks
constexpr int KV_OFF = W_OFF + 2 * WBLK;          // [8 heads][2 k-steps][64 lanes][16 B]: KV/S as 16-bit A fragments
constexpr int KS_OFF = KV_OFF + NH * 2 * FRAG;    // [256] float: Ksum/S rounded to the storage type
constexpr int VEC_OFF = KS_OFF + C * 4;           // gamma1 | beta1 | gamma2 | beta2, [4][256] float
constexpr int LDS_BYTES = VEC_OFF + 4 * C * 4;    // 152,576 B
constexpr int SLAB_RS = 272;                      // epilogue slab row stride (256 B + pad)

using EncArgs = GfEncArgs;

// eight fp32 values -> one 16-byte operand; converted in pairs (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even:
// element by element the compiler sometimes emits the single conversion + a byte permute)
template <typename T>
__device__ __forceinline__ typename Mma32<T>::Frag pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                                          float a7) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const t2 p0 = __builtin_convertvector(f2{a0, a1}, t2), p1 = __builtin_convertvector(f2{a2, a3}, t2);
    const t2 p2 = __builtin_convertvector(f2{a4, a5}, t2), p3 = __builtin_convertvector(f2{a6, a7}, t2);
    return typename Mma32<T>::Frag{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}
// registers 8s .. 8s+7 of an accumulator tile as the operand of k-step s of the next product
template <typename T>
__device__ __forceinline__ typename Mma32<T>::Frag pack_step(const v16f& a, int s) {
    return s == 0 ? pack8<T>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7])
                  : pack8<T>(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
}
// ReLU of 8 packed 16-bit floats: a set sign bit clears the element (-0 -> +0; fp16 and bf16 alike)
template <typename F>
__device__ __forceinline__ F relu_packed(const F& f) {
    typedef short v8s __attribute__((ext_vector_type(8)));
    const v8s x = __builtin_bit_cast(v8s, f);
    return __builtin_bit_cast(F, x & ~(x >> 15));
}
// elu(x) + 1 (linear_attention.py:33-34) = max(x, 0) + exp(min(x, 0)): x + 1 for x > 0 (exp(0) = 1 exactly), exp(x)
// otherwise - branch-free (a conditional exponential compiles to a divergent branch per element), hardware exponential
// (round 5: exp(min(x, 0)) = the exponential CLAMPED to [0, 1] - `v_exp_f32 ... clamp`, the output modifier is free - instead of a
// v_min in front of it: the same bits (for x <= 0 the clamp does nothing, for x > 0 both give exactly 1), one instruction less per value)
__device__ __forceinline__ float phi(float x) {
    return fmaxf(x, 0.f) + __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * 1.44269504088896341f), 0.f, 1.f);
}
__device__ __forceinline__ v16f zero16() { return v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
// the lane id recomputed where a late phase needs it (two VALU instructions; volatile: neither hoisted nor merged): carried in a
// register from the kernel's top it is the 511th live value of enc_layer and gets spilled - and a scratch reload waits for the
// LDS-DMA in flight
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// x[lane] + x[lane ^ 32] in every lane: one v_permlane32_swap (VALU, no address register, no LDS counter) instead of a ds_bpermute -
// the permute's byte-address register stayed live from the first LayerNorm to the state tail and was the value that got spilled
__device__ __forceinline__ float half_sum(float x) {
    const gf_v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(sw.x) + __uint_as_float(sw.y);
}
// sum_j a_j b_j over the 8 sixteen-bit elements of two packed operands, fp32 accumulation (v_dot2c_f32_f16 / _bf16: the products of
// two 16-bit values are exact in fp32)
__device__ __forceinline__ float dot8(const v8h& a, const v8h& b, float c) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_fdot2(h2{a[2 * i], a[2 * i + 1]}, h2{b[2 * i], b[2 * i + 1]}, c, false);
    return c;
}
__device__ __forceinline__ float dot8(const v8b& a, const v8b& b, float c) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_fdot2_f32_bf16(b2{a[2 * i], a[2 * i + 1]}, b2{b[2 * i], b[2 * i + 1]}, c, false);
    return c;
}
template <typename T>
__device__ __forceinline__ float rnd(float x) { return gf_to_float(gf_from_float<T>(x)); }   // round to the storage type

// The weight stream: 32-KiB blocks (32 fragments = 4 STEPS of 8 fragments = 8 MFMAs per wave) through a two-slot LDS ring.
// Wave w moves fragments 8w .. 8w+7 of a block ("pieces" 0..7 of the wave) by LDS-DMA.  Requests behind the stream's end are
// out of the buffer's range: they fetch nothing and leave zeros in the slot that has just been freed, which nobody reads
// (tools/probes/lds_dma_oob.hip) - no conditional code inside the step bodies: one basic block per unrolled phase, so the
// issue order below survives.
struct Ring {
    __amdgpu_buffer_rsrc_t rs;   // the stream as a raw buffer of exactly nblk blocks: a request behind its end is out of range (zeros)
    __amdgpu_buffer_rsrc_t rs2;  // blocks nblk .. nblk + nblk2 - 1 come from a second stream (the state tail's W_k | W_v)
    char* smem;
    int wave, lane, blk, nblk;
};
__device__ __forceinline__ Ring ring_make(const void* wstream, char* smem, int wave, int lane, int nblk, const void* wstream2 = nullptr,
                                          int nblk2 = 0) {
    return Ring{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wstream), 0, nblk * WBLK, 0x00020000),
                __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wstream2 ? wstream2 : wstream), 0, wstream2 ? nblk2 * WBLK : 0, 0x00020000),
                smem, wave, lane, 0, nblk};
}
// LDS-DMA as buffer_load_dwordx4 ... lds (MUBUF), not global_load_lds: behind a FLAT-encoded LDS-DMA the compiler's wait
// insertion treats every LDS counter wait as lgkmcnt(0) for as long as the request is pending (it may touch both address
// spaces), i.e. for the whole kernel here; behind the MUBUF form it counts (lgkmcnt(7) in front of every MFMA below).  The
// descriptor and the block offset are scalar: no 64-bit address arithmetic per piece.
__device__ __forceinline__ void dma_piece(const Ring& g, int b, int i) {
#if !defined(K9_ABLATE) || K9_ABLATE != 1
    // b is wave-uniform: the descriptor select is scalar
    const bool second = b >= g.nblk;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? g.rs2 : g.rs, (__attribute__((address_space(3))) void*)(g.smem + W_OFF + (b & 1) * WBLK + (g.wave * 8 + i) * FRAG), 16,
                                             g.lane * 16 + g.wave * 8 * FRAG, (second ? b - g.nblk : b) * WBLK + i * FRAG, 0, 0);
#endif
}
// the next block has landed (every wave waits for its own pieces) and every wave is done reading the current one
__device__ __forceinline__ void ring_turn() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if !defined(K9_ABLATE) || K9_ABLATE != 2
    __builtin_amdgcn_s_barrier();
#endif
}
template <typename Frag>
__device__ __forceinline__ Frag ring_read(const char* p) {
#if defined(K9_ABLATE) && K9_ABLATE == 3
    Frag f; asm volatile("" : "=v"(f)); return f;
#else
    return *reinterpret_cast<const Frag*>(p);
#endif
}
// One STEP = the 8 MFMAs `mma(0..7)` on the fragments the previous step fetched, and the fetch of the next step's 8
// fragments into `nxt`: ONE ds_read_b128 behind each MFMA, waited for by the counted lgkmcnt the compiler derives (a fragment
// is used 8 reads = 8 MFMAs = 256 cycles after its request).  [Before: two reads per gap in the first half of a step and
// lgkmcnt(0) at its top - the four waves of a workgroup run in lock-step behind the ring barrier, so their read bursts met
// at the LDS (2 x 4 waves x 1 KiB per 32-cycle gap = its whole 256 B/clk) and the stalled read issue held back the MFMAs:
// with the fragment reads compiled out the layer took 120 us instead of 166.]
// Step 3 of a block turns the ring after its first two MFMAs: block b+1 has landed, every wave holds block b's last fragments
// in registers; pieces 0..3 of block b+2 are requested behind MFMAs 2..5 of this step, pieces 4..7 behind MFMAs 0..3 of the
// next (a burst of 8 requests held the MFMA pipe idle for ~270 cycles per block).  xread() issues the step's EXTRA operand
// reads (token-tile fragments of the next step).
// the fragment MFMA j has just consumed stays allocated until the next read is out: the read lands in the register of the
// MFMA BEFORE it (issued a gap earlier, operands long read) instead of overwriting the operand of an MFMA still in the queue
#if defined(K9_NOKEEP)
#define K9_KEEP(f)
#else
#define K9_KEEP(f) asm volatile("" ::"v"(f))
#endif
template <int EXTRA, typename Frag, typename MF, typename XF>
__device__ __forceinline__ void ring_step(Ring& g, Frag (&cur)[8], Frag (&nxt)[8], int st, MF mma, XF xread) {
    if (st != 3) {
        const char* p = g.smem + W_OFF + (g.blk & 1) * WBLK + (st + 1) * 8 * FRAG + g.lane * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mma(j);
            if (st == 0 && j < 4) dma_piece(g, g.blk + 1, 4 + j);
            nxt[j] = ring_read<Frag>(p + j * FRAG);
            K9_KEEP(cur[j]);
            if (j == 0) xread();
        }
        // issue order: MFMA, [DMA request,] fragment read(s) - eight times
#define K9_GAP(DMA, READS)                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
        if (DMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
        K9_GAP(st == 0, 1 + EXTRA) K9_GAP(st == 0, 1) K9_GAP(st == 0, 1) K9_GAP(st == 0, 1)
        K9_GAP(false, 1) K9_GAP(false, 1) K9_GAP(false, 1) K9_GAP(false, 1)
        __builtin_amdgcn_sched_barrier(0);
    } else {
        mma(0);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
        ring_turn();
        const char* p = g.smem + W_OFF + ((g.blk + 1) & 1) * WBLK + g.lane * 16;
#pragma unroll
        for (int j = 2; j < 8; ++j) {
            mma(j);
            if (j < 6) dma_piece(g, g.blk + 2, j - 2);
            if (j < 4) {
                nxt[2 * (j - 2)] = ring_read<Frag>(p + 2 * (j - 2) * FRAG);
                nxt[2 * (j - 2) + 1] = ring_read<Frag>(p + (2 * (j - 2) + 1) * FRAG);
            } else {
                nxt[j] = ring_read<Frag>(p + j * FRAG);
            }
            K9_KEEP(cur[j]);
            if (j == 2) xread();
        }
        K9_GAP(true, 2 + EXTRA) K9_GAP(true, 2) K9_GAP(true, 1) K9_GAP(true, 1) K9_GAP(false, 1) K9_GAP(false, 1)
#undef K9_GAP
        __builtin_amdgcn_sched_barrier(0);
        ++g.blk;
    }
}
// ring_step with ONE extra operand read per MFMA gap (`xr(j)` behind MFMA j): the head-pipelined q projection, whose eight MFMAs of a
// step take eight DIFFERENT token-tile fragments (the k-steps of one head) - two ds_read_b128 per gap, the LDS's limit beside MFMAs
template <typename Frag, typename MF, typename XF, typename VF>
__device__ __forceinline__ void ring_step_q(Ring& g, Frag (&cur)[8], Frag (&nxt)[8], int st, MF mma, XF xr, VF valu) {
#define K9_GAPQ(DMA, READS)                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
        if (DMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);
    if (st != 3) {
        const char* p = g.smem + W_OFF + (g.blk & 1) * WBLK + (st + 1) * 8 * FRAG + g.lane * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mma(j);
            if (st == 0 && j < 4) dma_piece(g, g.blk + 1, 4 + j);
            nxt[j] = ring_read<Frag>(p + j * FRAG);
            xr(j);
            K9_KEEP(cur[j]);
            if (j == 0) valu();
        }
        K9_GAPQ(st == 0, 2) K9_GAPQ(st == 0, 2) K9_GAPQ(st == 0, 2) K9_GAPQ(st == 0, 2)
        K9_GAPQ(false, 2) K9_GAPQ(false, 2) K9_GAPQ(false, 2) K9_GAPQ(false, 2)
        __builtin_amdgcn_sched_barrier(0);
    } else {
        mma(0);
        xr(0);
        mma(1);
        xr(1);
        __builtin_amdgcn_sched_barrier(0);
        ring_turn();
        const char* p = g.smem + W_OFF + ((g.blk + 1) & 1) * WBLK + g.lane * 16;
#pragma unroll
        for (int j = 2; j < 8; ++j) {
            mma(j);
            if (j < 6) dma_piece(g, g.blk + 2, j - 2);
            if (j < 4) {
                nxt[2 * (j - 2)] = ring_read<Frag>(p + 2 * (j - 2) * FRAG);
                nxt[2 * (j - 2) + 1] = ring_read<Frag>(p + (2 * (j - 2) + 1) * FRAG);
            } else {
                nxt[j] = ring_read<Frag>(p + j * FRAG);
            }
            xr(j);
            K9_KEEP(cur[j]);
            if (j == 2) valu();
        }
        K9_GAPQ(true, 3) K9_GAPQ(true, 3) K9_GAPQ(true, 2) K9_GAPQ(true, 2) K9_GAPQ(false, 2) K9_GAPQ(false, 2)
        __builtin_amdgcn_sched_barrier(0);
        ++g.blk;
    }
#undef K9_GAPQ
}
// prologue: block 0 whole and the first half of block 1 (its second half goes out in step 0, like every later block's)
__device__ __forceinline__ void ring_start(const Ring& g) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(g, 0, i);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void ring_start2(const Ring& g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(g, 1, i);
    __builtin_amdgcn_sched_barrier(0);
}
template <typename Frag>
__device__ __forceinline__ void load_step0(const Ring& g, Frag (&f)[8]) {
    const char* p = g.smem + W_OFF + g.lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = ring_read<Frag>(p + i * FRAG);
}

// x tile -> LDS: [kc][row][128 B] with the 16-B chunk index XORed with (row>>1)&7 (conflict-free ds_read_b128)
template <typename T>
__device__ __forceinline__ void load_tile(const T* base, long ld, int row0, int rows_valid, char* smem, int off, int tid) {
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int e = p * 256 + tid, row = e >> 5, c32 = e & 31;
        const int r = min(row0 + row, rows_valid - 1);
        const v4u v = *reinterpret_cast<const v4u*>(base + (size_t)r * ld + c32 * 8);
        *reinterpret_cast<v4u*>(smem + off + (c32 >> 3) * 16384 + gf_lds_off(row, c32 & 7)) = v;
    }
}

template <typename T>
__device__ __forceinline__ void kv_tail(Ring& ring, typename Mma32<T>::Frag (&fa)[8], typename Mma32<T>::Frag (&fb)[8], char* smem, unsigned valid,
                                        float* dst, int wave, int lane, int tid, int trace_base);

template <typename T, int ACT, bool ATTN>
__global__ __launch_bounds__(256, 1) void enc_layer(EncArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, lr = lane & 31;
    const int n = blockIdx.x / a.tiles, tile = blockIdx.x - n * a.tiles;
    const int t0 = tile * TM;                                         // first token of the tile inside image n
    const T* xg = (const T*)a.x + (size_t)n * a.L * a.ldx;
    constexpr int NBLK = ATTN ? 32 : 28;
    float* vec = reinterpret_cast<float*>(smem + VEC_OFF);
    K9_T(0);
    // state tail (see EncArgs): this image's finished rows also leave their linear-attention state; the tail's 8 weight blocks
    // simply follow the layer's in the ring (requests behind the layer's last block are out of range without a tail: zeros)
    const bool tail = ATTN && a.wstream_tail != nullptr && n >= a.tail_first;
    Ring ring = ring_make(a.wstream, smem, wave, lane, NBLK, tail ? a.wstream_tail : nullptr, 8);
    ring_start(ring);
#if !defined(K9_ABLATE) || K9_ABLATE != 7
    load_tile<T>(xg, a.ldx, t0, a.L, smem, X_OFF, tid);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) vec[i * C + tid] = a.ln[i * C + tid];
    if constexpr (ATTN) {
        // KV / S and Ksum / S of image n in the storage type (the reference's "prevent fp16 overflow" scaling,
        // linear_attention.py:45-49: out = (Q.KV/S) / (Q.Ksum/S + eps/S)), KV as A fragments in the k order of a
        // packed accumulator: lane (v = lr, h2), element j  <->  d = 16s + 8(j>>2) + 4 h2 + (j&3)
        const float* kvf = a.kvfinal + (size_t)n * (C * D + C);
        const float inv_s = 1.0f / (float)a.S;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int slot = p * 256 + tid, ln = slot & 63, hs = slot >> 6, hh = hs >> 1, s = hs & 1;
            const float* src = kvf + (size_t)(hh * D + 16 * s + 4 * (ln >> 5)) * D + (ln & 31);       // KV[c = (hh, d)][v]
            *reinterpret_cast<Frag*>(smem + KV_OFF + hs * FRAG + ln * 16) =
                pack8<T>(src[0] * inv_s, src[D] * inv_s, src[2 * D] * inv_s, src[3 * D] * inv_s, src[8 * D] * inv_s, src[9 * D] * inv_s,
                         src[10 * D] * inv_s, src[11 * D] * inv_s);
        }
        {
            // Ksum / S rounded to the storage type, as 16-byte operands in the k order of the packed phi(q): entry (hh, s, h2),
            // element j = channel 32 hh + 16 s + 8 (j >> 2) + 4 h2 + (j & 3) - the denominator is then 4 dot-pair instructions per k-step
            const int j = tid & 7, hx = (tid >> 3) & 1, sx = (tid >> 4) & 1, hh = tid >> 5;
            reinterpret_cast<T*>(smem + KS_OFF)[tid] = gf_from_float<T>(kvf[C * D + hh * D + 16 * sx + 8 * (j >> 2) + 4 * hx + (j & 3)] * inv_s);
        }
    }
    Frag mfrag[8][2];                      // the B operand of the merge product, then of the second half of mlp.0
    if constexpr (!ATTN) {
        // the attention output of K4 / K5 comes from HBM: stage the 128 x 256 tile row-contiguously in the (still
        // empty) second ring slot + the KV area ... it is 64 KiB, the ring is 64 KiB, block 0 is in slot 0: use the
        // x-tile layout at W_OFF + WBLK is too small, so the tile goes through in two 64-token halves
        const T* mg = (const T*)a.msg + (size_t)n * a.L * a.ldm;
        __syncthreads();                   // nothing pending on slot 1; slot 0 holds block 0 (DMA may still be in flight: different bytes)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // 64 rows x 512 B = 32 KiB into slot 1, image [kc][64 rows][128 B]
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int e = p * 256 + tid, row = e >> 5, c32 = e & 31;
                const int r = min(t0 + half * 64 + row, a.L - 1);
                const v4u v = *reinterpret_cast<const v4u*>(mg + (size_t)r * a.ldm + c32 * 8);
                *reinterpret_cast<v4u*>(smem + W_OFF + WBLK + (c32 >> 3) * 8192 + gf_lds_off(row, c32 & 7)) = v;
            }
            __syncthreads();
            if ((wave >> 1) == half) {     // waves 0,1 own rows 0..63, waves 2,3 rows 64..127
                const int row = (wave & 1) * 32 + lr;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        // channels 32t + 16s + 4h2 + {0..3} and + 8: two 8-byte pieces of chunk (32t + 16s) / 8 and the next
                        const int c0 = 32 * t + 16 * s + 4 * h2, ch = c0 >> 3;           // c0 % 8 is 0 or 4
                        const char* p0 = smem + W_OFF + WBLK + (ch >> 3) * 8192 + gf_lds_off(row, ch & 7) + (c0 & 7) * 2;
                        const char* p1 = smem + W_OFF + WBLK + ((ch + 1) >> 3) * 8192 + gf_lds_off(row, (ch + 1) & 7) + (c0 & 7) * 2;
                        typedef short v4s __attribute__((ext_vector_type(4)));
                        typedef short v8s __attribute__((ext_vector_type(8)));
                        const v4s lo = *reinterpret_cast<const v4s*>(p0), hi = *reinterpret_cast<const v4s*>(p1);
                        const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        mfrag[t][s] = __builtin_bit_cast(Frag, both);
                    }
            }
            __syncthreads();
        }
    }
    // block 0 and the first half of block 1 are requested; wait for block 0 only (the 4 youngest requests are block 1's)
    ring_start2(ring);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    K9_T(1);
    const int tok = t0 + wave * 32 + lr;                              // this lane's token (accumulator column)
    const char* xrow = smem + X_OFF;
    const int myrow = wave * 32 + lr;
    Frag fa[8], fb[8];                                                // fragments of the current / next step, alternating
    load_step0(ring, fa);
    // token-tile operand of 16-deep k-step ks (0..15)
    auto xfrag = [&](int ks) { return *reinterpret_cast<const Frag*>(xrow + (ks >> 2) * 16384 + gf_lds_off(myrow, 2 * (ks & 3) + h2)); };

#if K9_HEADPIPE
    v16f m[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) m[nb] = zero16();
    if constexpr (ATTN) {
        // Round 5: q projection, attention and merge as ONE head-pipelined instruction stream.  The weight stream is head-major
        // (fused.py: Q0 Q1 M0 Q2 M1 ... Q7 M6 M7; Q(h) = the 16 k-steps of channel tile h into ONE accumulator, M(h) = the merge's
        // k-steps 2h, 2h + 1 over tiles 0..7) and the attention of head h - phi, packing, the normaliser, its two state MFMAs,
        // the normalisation - rides in the two steps of Q(h + 1) (AB7 in M6): its ~160 vector instructions issue under the 16 MFMAs of
        // the neighbouring head instead of in a phase of their own with the matrix pipe idle (one wave per SIMD: nothing else overlaps
        // them).  Every accumulator still adds its k-steps in ascending order - q[h]: 0..15, m[nb]: 0..15 - so the layer's output
        // keeps its bits.  The token tile's 16 operand fragments are the B operands of EVERY head's projection: read once.
        // The token tile's operand fragments: MFMA G of the phase (G = 16 head + k-step) takes k-step G % 16; a four-deep rotating
        // window of them lives in registers (the fragment of MFMA G + 3 is requested behind MFMA G: three gaps ahead of its use)
        Frag xq[4];
        xq[0] = xfrag(0);
        xq[1] = xfrag(1);
        xq[2] = xfrag(2);
        v16f qh[2];
        const float qmul = (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + min(t0 + wave * 32 + (fresh_lane() & 31), a.L - 1)] != 0) ? 1.f : 0.f;
        const float eps_s = a.attn_eps / (float)a.S;
        struct HeadOps { Frag ks0, ks1; Frag kv0, kv1; };
        auto head_ops = [&](int hh) {
            HeadOps o;
            o.ks0 = *reinterpret_cast<const Frag*>(smem + KS_OFF + ((hh * 2 + 0) * 2 + h2) * 16);
            o.ks1 = *reinterpret_cast<const Frag*>(smem + KS_OFF + ((hh * 2 + 1) * 2 + h2) * 16);
            o.kv0 = *reinterpret_cast<const Frag*>(smem + KV_OFF + (hh * 2) * FRAG + lane * 16);
            o.kv1 = *reinterpret_cast<const Frag*>(smem + KV_OFF + (hh * 2 + 1) * FRAG + lane * 16);
            return o;
        };
        v16f num;
        float den = 0.f;
        HeadOps hop;
        // stage A(h): phi(q[h]) rounded by its packing, the normaliser from the PACKED operands, the two state MFMAs
        auto stage_a = [&](int hh) {
            float pq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pq[r] = phi(qh[hh & 1][r]);
            const Frag p0 = pack8<T>(pq[0], pq[1], pq[2], pq[3], pq[4], pq[5], pq[6], pq[7]);
            const Frag p1 = pack8<T>(pq[8], pq[9], pq[10], pq[11], pq[12], pq[13], pq[14], pq[15]);
            den = dot8(p1, hop.ks1, dot8(p0, hop.ks0, 0.f));
            num = zero16();
            Mm::mma(hop.kv0, p0, num);
            Mm::mma(hop.kv1, p1, num);
        };
        // stage B(h): normalise and pack the message of head h = the B operand of M(h)
        auto stage_b = [&](int hh) {
            const float z = __builtin_amdgcn_rcpf(half_sum(den) + eps_s) * qmul;
#pragma unroll
            for (int r = 0; r < 16; ++r) num[r] *= z;
            mfrag[hh][0] = pack_step<T>(num, 0);
            mfrag[hh][1] = pack_step<T>(num, 1);
        };
        int gs = 0;                                                       // step counter of the phase (everything below is unrolled)
        auto q_slot = [&](auto hh_c, auto ab_c) {                          // Q(hh), with A(hh - 1) / B(hh - 1) riding in its two steps
            constexpr int HH = decltype(hh_c)::value;
            constexpr bool AB = decltype(ab_c)::value;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                Frag (&cur)[8] = (gs & 1) ? fb : fa;
                Frag (&nxt)[8] = (gs & 1) ? fa : fb;
                if (hf == 0) qh[HH & 1] = zero16();
                ring_step_q(ring, cur, nxt, gs & 3,
                            [&](int j) { Mm::mma(cur[j], xq[(8 * hf + j) & 3], qh[HH & 1]); },          // (16 HH + 8 hf + j) % 4
                            [&](int j) { if (HH < 7 || 8 * hf + j + 3 < 16) xq[(8 * hf + j + 3) & 3] = xfrag((8 * hf + j + 3) & 15); },
                            [&] {
                                if constexpr (AB) {
                                    if (hf == 0) stage_a(HH - 1);
                                    else stage_b(HH - 1);
                                }
                            });
                ++gs;
            }
        };
        auto m_slot = [&](int hh, auto ab_c) {                             // M(hh), with A(7) / B(7) riding in M(6)
            constexpr bool AB = decltype(ab_c)::value;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                Frag (&cur)[8] = (gs & 1) ? fb : fa;
                Frag (&nxt)[8] = (gs & 1) ? fa : fb;
                const Frag bf = mfrag[hh][sx];
                ring_step<0>(ring, cur, nxt, gs & 3, [&](int nb) { Mm::mma(cur[nb], bf, m[nb]); },
                             [&] {
                                 if constexpr (AB) {
                                     if (sx == 0) stage_a(7);
                                     else stage_b(7);
                                 }
                             });
                ++gs;
            }
        };
        using std::true_type;
        using std::false_type;
        using std::integral_constant;
        q_slot(integral_constant<int, 0>{}, false_type{});
        hop = head_ops(0);
        q_slot(integral_constant<int, 1>{}, true_type{});
        m_slot(0, false_type{}); hop = head_ops(1); q_slot(integral_constant<int, 2>{}, true_type{});
        m_slot(1, false_type{}); hop = head_ops(2); q_slot(integral_constant<int, 3>{}, true_type{});
        m_slot(2, false_type{}); hop = head_ops(3); q_slot(integral_constant<int, 4>{}, true_type{});
        m_slot(3, false_type{}); hop = head_ops(4); q_slot(integral_constant<int, 5>{}, true_type{});
        m_slot(4, false_type{}); hop = head_ops(5); q_slot(integral_constant<int, 6>{}, true_type{});
        m_slot(5, false_type{}); hop = head_ops(6); q_slot(integral_constant<int, 7>{}, true_type{});
        hop = head_ops(7);
        m_slot(6, true_type{});
        m_slot(7, false_type{});
        K9_T(2);
        K9_T(3);
    } else {
        K9_T(3);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            const Frag bf = mfrag[st >> 1][st & 1];
            ring_step<0>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(cur[nb], bf, m[nb]); }, [] {});
        }
    }
#else
#if K9_MERGEPIPE
    v16f m[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) m[nb] = zero16();
#endif
    if constexpr (ATTN) {
        // ---------------- q = W_q x : 16 steps (k-step ks = step, tiles nb = 0..7)
        v16f q[8];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) q[nb] = zero16();
        Frag tf = xfrag(0);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            Frag tn;
            ring_step<1>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(cur[nb], tf, q[nb]); }, [&] { tn = xfrag(st < 15 ? st + 1 : 15); });
            tf = tn;
        }
        K9_T(2);
        // ---------------- linear attention per head: tile h of q is head h (32 channels)
        const float qmul = (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + min(t0 + wave * 32 + (fresh_lane() & 31), a.L - 1)] != 0) ? 1.f : 0.f;
        const float eps_s = a.attn_eps / (float)a.S;
        // Two-stage pipeline over the heads (same arithmetic, same order per head): stage A(h) = phi, denominator and the two
        // state MFMAs of head h, stage B(h) = normalise and pack; B(h-1) runs behind A(h), under A(h)'s MFMAs, and the LDS operands
        // of head h+1 (Ksum rows, state fragments) are requested a head ahead - one wave per SIMD: nothing else hides their latency.
        struct HeadOps { Frag ks0, ks1; Frag kv0, kv1; };
        auto head_ops = [&](int hh) {
            HeadOps o;
            o.ks0 = *reinterpret_cast<const Frag*>(smem + KS_OFF + ((hh * 2 + 0) * 2 + h2) * 16);
            o.ks1 = *reinterpret_cast<const Frag*>(smem + KS_OFF + ((hh * 2 + 1) * 2 + h2) * 16);
            o.kv0 = *reinterpret_cast<const Frag*>(smem + KV_OFF + (hh * 2) * FRAG + lane * 16);
            o.kv1 = *reinterpret_cast<const Frag*>(smem + KV_OFF + (hh * 2 + 1) * FRAG + lane * 16);
            return o;
        };
#if K9_MERGEPIPE
#define K9_HX(hh) 0
#else
#define K9_HX(hh) ((hh) & 1)
#endif
        v16f num[2];
        float den[2];
        // phi(q) is rounded by its packing (one conversion per pair) and the denominator phi(q) . Ksum / S is summed from the PACKED
        // operands (the 16-bit products are exact in fp32) - rounds 2-3 rounded every value by a conversion there and back, multiplied
        // in fp32 and converted again for the packing; a masked query (phi(q) = 0 in linear_attention.py:35-36, message 0 / eps = 0)
        // is zeroed through its normaliser instead of a multiply per value
        auto stage_a = [&](int hh, const HeadOps& ho) {
            float pq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pq[r] = phi(q[hh][r]);
            const Frag p0 = pack8<T>(pq[0], pq[1], pq[2], pq[3], pq[4], pq[5], pq[6], pq[7]);
            const Frag p1 = pack8<T>(pq[8], pq[9], pq[10], pq[11], pq[12], pq[13], pq[14], pq[15]);
            den[K9_HX(hh)] = dot8(p1, ho.ks1, dot8(p0, ho.ks0, 0.f));
            num[K9_HX(hh)] = zero16();
            Mm::mma(ho.kv0, p0, num[K9_HX(hh)]);
            Mm::mma(ho.kv1, p1, num[K9_HX(hh)]);
        };
        auto stage_b = [&](int hh) {
            float d = den[K9_HX(hh)];
            d = half_sum(d);
            const float z = __builtin_amdgcn_rcpf(d + eps_s) * qmul;
            v16f& nm = num[K9_HX(hh)];
#pragma unroll
            for (int r = 0; r < 16; ++r) nm[r] *= z;
            mfrag[hh][0] = pack_step<T>(nm, 0);
            mfrag[hh][1] = pack_step<T>(nm, 1);
        };
        HeadOps hops[2];
        hops[0] = head_ops(0);
#if K9_MERGEPIPE
        // Round 5 (-DK9_MERGEPIPE=1, an experiment): the merge's k-steps are head-major already (steps 2h, 2h + 1 = head h), so the
        // attention of head h + 1 can ride in the two merge steps of head h - only head 0's attention stays a phase of its own.
        // Same sums in the same order.
        stage_a(0, hops[0]);
        hops[0] = head_ops(1);
        stage_b(0);
        K9_T(3);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            const Frag bf = mfrag[st >> 1][st & 1];
            const int hn = (st >> 1) + 1;                                   // the head whose attention rides in this step
            ring_step<0>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(cur[nb], bf, m[nb]); },
                         [&] {
                             if (hn < 8) {
                                 if ((st & 1) == 0) {
                                     stage_a(hn, hops[0]);                  // (one set of operands and one numerator: A and B of a head are a step apart)
                                 } else {
                                     stage_b(hn);
                                     if (hn < 7) hops[0] = head_ops(hn + 1);
                                 }
                             }
                         });
        }
    } else {
        K9_T(3);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            const Frag bf = mfrag[st >> 1][st & 1];
            ring_step<0>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(cur[nb], bf, m[nb]); }, [] {});
        }
    }
#else
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
            if (hh < 7) hops[(hh + 1) & 1] = head_ops(hh + 1);
            stage_a(hh, hops[hh & 1]);
            if (hh > 0) stage_b(hh - 1);
        }
        stage_b(7);
    }
#endif

#if !K9_MERGEPIPE
    K9_T(3);
    // ---------------- m = LN1(W_m msg) : 16 steps, operand (tile t = step / 2, k-step s = step % 2) from registers
    v16f m[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) m[nb] = zero16();
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        Frag (&cur)[8] = (st & 1) ? fb : fa;
        Frag (&nxt)[8] = (st & 1) ? fa : fb;
        const Frag bf = mfrag[st >> 1][st & 1];
        ring_step<0>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(cur[nb], bf, m[nb]); }, [] {});
    }
#endif
#endif
    // nn.LayerNorm over the 256 channels of the lane's token, statistics in fp32; the other lane half holds the other
    // 128 channels.  One pass over the accumulators (they live in AGPRs: every use is a register move): sum and sum of
    // squares, var = E[x^2] - mean^2 (|x| = O(1) after a 256..512-deep product of O(1) operands: no cancellation issue).
    auto ln_stats = [&](const v16f (&t)[8], float eps, float& mean, float& rstd) {
        float s = 0.f, qd = 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = t[nb][r];
                s += v;
                qd = fmaf(v, v, qd);
            }
        s = half_sum(s);
        qd = half_sum(qd);
        mean = s * (1.0f / C);
        rstd = 1.0f / sqrtf(fmaxf(qd - s * mean, 0.f) * (1.0f / C) + eps);
    };
    // gamma / beta of tile nb for this lane half (channels nb*32 + 8g + 4 h2 + {0..3}); requested a tile ahead of their use
    struct LnOps { v4f ga[4], be[4]; };
    auto ln_ops = [&](const float* gamma, const float* beta, int nb, int hl) {
        LnOps o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            o.ga[g] = *reinterpret_cast<const v4f*>(gamma + nb * 32 + 8 * g + 4 * hl);
            o.be[g] = *reinterpret_cast<const v4f*>(beta + nb * 32 + 8 * g + 4 * hl);
        }
        return o;
    };
    // the four normalised values of registers 4g .. 4g+3 of a tile
    // (t - mean) rstd gamma + beta as two FMAs per value: t rstd - mean rstd, then times gamma plus beta (`nmr` = -mean rstd)
    auto ln_apply = [&](const v16f& t, int g, const LnOps& p, float nmr, float rstd) {
        return v4f{fmaf(fmaf(t[4 * g], rstd, nmr), p.ga[g].x, p.be[g].x), fmaf(fmaf(t[4 * g + 1], rstd, nmr), p.ga[g].y, p.be[g].y),
                   fmaf(fmaf(t[4 * g + 2], rstd, nmr), p.ga[g].z, p.be[g].z), fmaf(fmaf(t[4 * g + 3], rstd, nmr), p.ga[g].w, p.be[g].w)};
    };
    K9_T(4);
    {
        float mean, rstd;
        ln_stats(m, a.eps1, mean, rstd);
        const float nmr = -mean * rstd;
        {
            LnOps lp[2];
            lp[0] = ln_ops(vec, vec + C, 0, h2);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                if (nb < 7) lp[(nb + 1) & 1] = ln_ops(vec, vec + C, nb + 1, h2);
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const v4f lo = ln_apply(m[nb], 2 * sx, lp[nb & 1], nmr, rstd), hi = ln_apply(m[nb], 2 * sx + 1, lp[nb & 1], nmr, rstd);
                    mfrag[nb][sx] = pack8<T>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
                }
            }
        }
    }

    K9_T(5);
    // ---------------- hid = act(W_1 [x | m]) in four 128-wide slices, each consumed at once by out += W_2[:, slice] hid
    // per slice 24 steps: 8 (x half: tiles hb = 0..3 x k-steps 2j, 2j+1) + 8 (m half: tile j of m) + 8 (W_2: tiles nb, k-step u)
    v16f o[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
#pragma unroll 1
    for (int sl = 0; sl < 4; ++sl) {
        v16f hd[4];
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) hd[hb] = zero16();            // constant: the first MFMA of each chain takes C = 0 inline
        if (sl == 1) K9_T(6);
        Frag t0f = xfrag(0), t1f = xfrag(1);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            Frag n0, n1;
            ring_step<2>(ring, cur, nxt, st & 3, [&](int j) { Mm::mma(cur[j], j < 4 ? t0f : t1f, hd[j & 3]); },
                         [&] { n0 = xfrag(st < 7 ? 2 * st + 2 : 14); n1 = xfrag(st < 7 ? 2 * st + 3 : 15); });
            t0f = n0;
            t1f = n1;
        }
        if (sl == 0) K9_T(10);
        if (sl == 1) K9_T(13);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            ring_step<0>(ring, cur, nxt, st & 3, [&](int j) { Mm::mma(cur[j], mfrag[st][j >> 2], hd[j & 3]); }, [] {});
        }
        if (sl == 0) K9_T(11);
        if (sl == 1) K9_T(14);
        // activation + packing of hidden tile hb (32 channels): for ReLU on the PACKED operand - relu(round(x)) = round(relu(x)),
        // and on 16-bit floats it is 'sign bit set -> 0' (one v_pk_max_i16 per pair).  Tile 0 in front of the W_2 steps, tile
        // hb + 1 inside steps 2 hb, 2 hb + 1 (a handful of VALU instructions per MFMA gap instead of ~350 in front of the loop).
        Frag hfrag[4][2];
        auto act_pack = [&](int hb, int sx) {
            if constexpr (ACT == 0) {
                hfrag[hb][sx] = relu_packed(pack_step<T>(hd[hb], sx));
            } else {
#pragma unroll
                for (int r = 8 * sx; r < 8 * sx + 8; ++r) hd[hb][r] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * hd[hb][r]) + 1.0f);     // tanh, ~1e-6 abs
                hfrag[hb][sx] = pack_step<T>(hd[hb], sx);
            }
        };
        act_pack(0, 0);
        act_pack(0, 1);
        if (sl == 0) K9_T(12);
        if (sl == 1) K9_T(15);
#pragma unroll
        for (int st = 0; st < 8; ++st) {                                // out += W_2[:, 128 sl + 16 st .. + 15] hid
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            const Frag bf = hfrag[st >> 1][st & 1];
            ring_step<0>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(cur[nb], bf, o[nb]); }, [&] { if (st < 6) act_pack((st >> 1) + 1, st & 1); });
        }
    }

    K9_T(7);
    // ---------------- out = x + LN2(.) (one rounding), per-sample skip predicate, row-contiguous stores through a slab
    float mean2, rstd2;
#if defined(K9_ABLATE) && K9_ABLATE == 6
    {
        float sacc = 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) sacc += o[nb][0] + o[nb][15];
        if (sacc == 1.2345f) ((float*)a.out)[threadIdx.x] = sacc;
        return;
    }
#endif
    ln_stats(o, a.eps2, mean2, rstd2);
    const float nmr2 = -mean2 * rstd2;
    K9_T(8);
    // the predicate is per sample (flag_rows is a multiple of L in every caller): read it once, wave-uniformly
    const bool keep = a.flag == nullptr || __builtin_amdgcn_readfirstlane(a.flag[((size_t)n * a.L + t0) / a.flag_rows]) != 0;
    // The finished rows replace the x tile IN PLACE (a lane overwrites exactly the residual bytes it has just read) and leave for
    // HBM from there, two whole 512-byte rows per store instruction.  [Rounds 2-3 went through a slab in the ring's LDS: the ring had
    // to be drained first and could not run on into the state tail.]  Only the wave's own 32 rows are touched: no workgroup barrier.
    T* og = (T*)a.out + (size_t)n * a.L * a.ldo;
    // one straight-line body per (layer kept?, tile inside the sequence?): as run-time branches inside the loops every join
    // costs a conservative wait on the stores of the first channel half
    auto finish = [&](auto keep_c, auto full_c) {
        constexpr bool KEEP = decltype(keep_c)::value, FULL = decltype(full_c)::value;
        typedef T v4t __attribute__((ext_vector_type(4)));
        const int el = fresh_lane();                                    // row offsets recomputed here, not carried through the layer
        const int er = el & 31, eh2 = el >> 5, erow = wave * 32 + er;
        // residual x[row][nb*32 + 8g + 4 eh2 ..+3]: 16-B chunk 4 nb + g of the row = k-chunk nb >> 1, slot (4 (nb & 1) + g) ^ swizzle:
        // eight lane-constant slot addresses, everything else is an immediate offset
        int xs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[k] = X_OFF + gf_lds_off(erow, k) + eh2 * 8;
        struct FinOps { v4t x[4]; LnOps ln; };
        auto fin_ops = [&](int nb) {
            FinOps o;
#pragma unroll
            for (int g = 0; g < 4; ++g) o.x[g] = *reinterpret_cast<const v4t*>(smem + xs[4 * (nb & 1) + g] + (nb >> 1) * 16384);
            if constexpr (KEEP) o.ln = ln_ops(vec + 2 * C, vec + 3 * C, nb, eh2);
            return o;
        };
        if constexpr (KEEP) {
            FinOps fo[2];
            fo[0] = fin_ops(0);                                         // the operands of tile nb + 1 are requested before tile nb is computed
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                if (nb < 7) fo[(nb + 1) & 1] = fin_ops(nb + 1);
                const FinOps& p = fo[nb & 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f y = ln_apply(o[nb], g, p.ln, nmr2, rstd2);
                    v4t ov;
                    ov[0] = gf_from_float<T>(gf_to_float(p.x[g][0]) + y.x);
                    ov[1] = gf_from_float<T>(gf_to_float(p.x[g][1]) + y.y);
                    ov[2] = gf_from_float<T>(gf_to_float(p.x[g][2]) + y.z);
                    ov[3] = gf_from_float<T>(gf_to_float(p.x[g][3]) + y.w);
                    *reinterpret_cast<v4t*>(smem + xs[4 * (nb & 1) + g] + (nb >> 1) * 16384) = ov;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // (layer skipped: out = x, the tile is the result as it stands)
        // 16-B piece c32 of a row = plane c32 >> 3, slot c32 & 7: lanes 0..31 one row, lanes 32..63 the next
        const int c32 = el & 31, rsel = el >> 5;
        v4u rows[16];
#pragma unroll
        for (int it = 0; it < 16; ++it)
            rows[it] = *reinterpret_cast<const v4u*>(smem + X_OFF + (c32 >> 3) * 16384 + gf_lds_off(wave * 32 + 2 * it + rsel, c32 & 7));
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int tg = t0 + wave * 32 + 2 * it + rsel;
            if (FULL || tg < a.L) *reinterpret_cast<v4u*>(og + (size_t)tg * a.ldo + c32 * 8) = rows[it];
        }
    };
    using std::integral_constant;
    const bool full = t0 + TM <= a.L;
    if (keep) {
        if (full) finish(integral_constant<bool, true>{}, integral_constant<bool, true>{});
        else finish(integral_constant<bool, true>{}, integral_constant<bool, false>{});
    } else {
        if (full) finish(integral_constant<bool, false>{}, integral_constant<bool, true>{});
        else finish(integral_constant<bool, false>{}, integral_constant<bool, false>{});
    }
    K9_T(9);
    if constexpr (ATTN) {
        if (tail) {
            // the finished rows (storage type, in the x tile's place) are the source tile of the state: masked or out-of-sequence
            // tokens do not count (linear_attention.py:37-39).  The ring runs on: its next blocks are the tail's W_k | W_v.
            const int tl = fresh_lane();                                // the tail's lane-derived addresses are computed here, not carried through the layer
            ring.lane = tl;
            const int ttok = t0 + wave * 32 + (tl & 31);
            const bool ok = ttok < a.L && (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + min(ttok, a.L - 1)] != 0);
            const unsigned valid = (unsigned)__ballot(ok && (tl >> 5) == 0);
            kv_tail<T>(ring, fa, fb, smem, valid, a.part + ((size_t)(n - a.tail_first) * a.tiles + tile) * (C * D + C), wave, tl, wave * 64 + tl, 16);
        }
    }
}

// The body both enc_kv_state and enc_layer's state tail run: the token tile is in LDS at X_OFF (x-tile layout), the ring's
// current block is the first block of a W_k | W_v stream with its step-0 fragments in `fa`; `valid` = bit mask of the wave's 32
// tokens that count.  Ends with the tile's partial state in `dst`; uses ALL of the workgroup's LDS at the end (the tile and the
// ring are dead by then).
template <typename T>
__device__ __forceinline__ void kv_tail(Ring& ring, typename Mma32<T>::Frag (&fa)[8], typename Mma32<T>::Frag (&fb)[8], char* smem, unsigned valid,
                                        float* dst, int wave, int lane, int tid, int trace_base) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    const int h2 = lane >> 5, lr = lane & 31;
    const int myrow = wave * 32 + lr;
    auto xfrag = [&](int ks) { return *reinterpret_cast<const Frag*>(smem + X_OFF + (ks >> 2) * 16384 + gf_lds_off(myrow, 2 * (ks & 3) + h2)); };
    auto project = [&](v16f (&acc)[8], auto between) {                  // acc = src tile x W^T: 16 steps (k-step = step)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        Frag tf = xfrag(0);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            Frag (&cur)[8] = (st & 1) ? fb : fa;
            Frag (&nxt)[8] = (st & 1) ? fa : fb;
            Frag tn;
            ring_step<1>(ring, cur, nxt, st & 3, [&](int nb) { Mm::mma(tf, cur[nb], acc[nb]); }, [&] { tn = xfrag(st < 15 ? st + 1 : 15); between(st); });
            tf = tn;
        }
    };
    // k first; phi(k) = elu + 1, its sum and the packing of head hh as MFMA operands then ride in the gaps of the v projection's
    // steps 2 hh, 2 hh + 1 (eight values per lane and step: ~60 VALU instructions beside 8 MFMAs) - as a phase of its own between the
    // two projections they were ~5 of the tail's ~20 thousand cycles with the matrix pipe idle
    Frag kf[8][2];
    float ksum[8];
    float vm[16];                                                       // 1 / 0 per accumulator row (token) of this lane half
#pragma unroll
    for (int r = 0; r < 16; ++r) vm[r] = ((valid >> gf_acc_row(r, h2)) & 1u) ? 1.f : 0.f;
    v16f k[8];
    project(k, [](int) {});
    K9_T(trace_base + 0);
    float srun = 0.f;
    auto phi_half = [&](int st) {
        const int hh = st >> 1, sx = st & 1;
        float s = sx == 0 ? 0.f : srun;
#pragma unroll
        for (int r = 8 * sx; r < 8 * sx + 8; ++r) {
            const float p = phi(k[hh][r]) * vm[r];
            k[hh][r] = p;                                               // rounded when packed as the MFMA operand
            s += p;                                                     // Ksum adds the fp32 values, rows in order
        }
        kf[hh][sx] = pack_step<T>(k[hh], sx);
        if (sx == 0) srun = s;
        else ksum[hh] = half_sum(s);                                  // lane lr = channel d
    };
    // state of the wave's 32 tokens: head h = channel tile h; rows = d, lane = v
    v16f kv[8];
    {
        v16f v[8];
        K9_T(trace_base + 1);
        project(v, phi_half);
        K9_T(trace_base + 2);
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) kv[hh][r] = 0.f;
            Mm::mma(kf[hh][0], pack_step<T>(v[hh], 0), kv[hh]);
            Mm::mma(kf[hh][1], pack_step<T>(v[hh], 1), kv[hh]);
        }
    }
    K9_T(trace_base + 3);
    // sum of the four waves, then one partial per tile: every wave parks its partial in LDS (4 x 34 KiB: the tile and the ring
    // are dead by now), and all 256 threads add the four copies - in the order (w0 + w2) + (w1 + w3) - and store the partial
    // (a tree that ended with ONE wave adding and storing 33 KB took 9 of the kernel's 35 thousand cycles)
    constexpr int KVN = 8 * 16 * 64, SLOT = KVN + 8 * 64;              // floats per wave: kv[hh][r][lane] | ksum[hh][lane]
    float* red = reinterpret_cast<float*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the requests behind the stream's end (out of range: zeros) have landed
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * SLOT + (hh * 16 + r) * 64 + lane] = kv[hh][r];
        red[wave * SLOT + KVN + hh * 64 + lane] = ksum[hh];
    }
    __syncthreads();
    K9_T(trace_base + 4);
#pragma unroll 4
    for (int e = tid; e < KVN; e += 256) {
        const float sum = (red[e] + red[2 * SLOT + e]) + (red[SLOT + e] + red[3 * SLOT + e]);
        const int hh = e >> 10, r = (e >> 6) & 15, ln = e & 63;
        dst[(size_t)(hh * D + gf_acc_row(r, ln >> 5)) * D + (ln & 31)] = sum;                               // [c][v]: 128-B runs
    }
    for (int e = tid; e < 8 * 64; e += 256) {
        const int ln = e & 63;
        if (ln < 32) {
            const int o = KVN + e;
            dst[C * D + (e >> 6) * D + ln] = (red[o] + red[2 * SLOT + o]) + (red[SLOT + o] + red[3 * SLOT + o]);
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// enc_kv_state: k, v projections of a 128-token source tile and its linear-attention state, in registers.
// Here the products are NOT transposed (A = 32 token rows from the LDS tile, B = 32 weight rows): the result has
// the channel on the lane and the tokens in registers, so the state KV[d][v] = sum_tok phi(k)[tok][d] v[tok][v] - a
// contraction over the tiles' ROW index - takes both accumulators as MFMA operands directly (phi(k) as A gives
// phi(k)^T . v).  Stream: the four 64-deep blocks of W_k, then the four of W_v.
// -------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 1) void enc_kv_state(EncArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, lr = lane & 31;
    const int n = blockIdx.x / a.tiles, tile = blockIdx.x - n * a.tiles;
    const int t0 = tile * TM;
    const T* xg = (const T*)a.x + (size_t)n * a.S * a.ldx;
    constexpr int NBLK = 8;
    K9_T(0);
    Ring ring = ring_make(a.wstream, smem, wave, lane, NBLK);
    ring_start(ring);
    load_tile<T>(xg, a.ldx, t0, a.S, smem, X_OFF, tid);
    // validity of the wave's 32 tokens as a bit mask (tail of the image, padding mask of linear_attention.py:37-39)
    const int mytok = t0 + wave * 32 + lr;
    const bool ok = mytok < a.S && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + min(mytok, a.S - 1)] != 0);
    const unsigned valid = (unsigned)__ballot(ok && h2 == 0);
    ring_start2(ring);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // block 0 and the tile; block 1's first half stays in flight
    __builtin_amdgcn_s_barrier();
    K9_T(1);
    Frag fa[8], fb[8];
    load_step0(ring, fa);
    kv_tail<T>(ring, fa, fb, smem, valid, a.part + ((size_t)n * a.tiles + tile) * (C * D + C), wave, lane, tid, 2);
    K9_T(7);
}

__global__ void enc_kv_reduce(const float* part, float* fin, int tiles, int len) {
    const int n = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const float* p = part + (size_t)n * tiles * len + i;
    float s = 0.f;
    int c = 0;
    for (; c + 10 <= tiles; c += 10) {                                 // ten loads in flight, added in tile order
        float v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = p[(size_t)(c + j) * len];
#pragma unroll
        for (int j = 0; j < 10; ++j) s += v[j];
    }
    for (; c < tiles; ++c) s += p[(size_t)c * len];
    fin[(size_t)n * len + i] = s;
}

template <typename T>
int enc_launch(const EncArgs& a, int act, bool attn, hipStream_t st) {
    const dim3 grid(a.N * a.tiles);
    if (attn) {
        if (act == 0) enc_layer<T, 0, true><<<grid, 256, LDS_BYTES, st>>>(a);
        else enc_layer<T, 1, true><<<grid, 256, LDS_BYTES, st>>>(a);
    } else {
        if (act == 0) enc_layer<T, 0, false><<<grid, 256, LDS_BYTES, st>>>(a);
        else enc_layer<T, 1, false><<<grid, 256, LDS_BYTES, st>>>(a);
    }
    return 0;
}

std::atomic<uint64_t> enc_attr_done{0};
template <typename K>
void enc_allow_lds(K kern) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); }
void enc_init() {
    if (!gf_first_use_on_device(enc_attr_done)) return;
    enc_allow_lds(enc_layer<_Float16, 0, true>); enc_allow_lds(enc_layer<_Float16, 1, true>);
    enc_allow_lds(enc_layer<_Float16, 0, false>); enc_allow_lds(enc_layer<_Float16, 1, false>);
    enc_allow_lds(enc_kv_state<_Float16>);
    enc_allow_lds(enc_layer<gf_bf16, 0, true>); enc_allow_lds(enc_layer<gf_bf16, 1, true>);
    enc_allow_lds(enc_layer<gf_bf16, 0, false>); enc_allow_lds(enc_layer<gf_bf16, 1, false>);
    enc_allow_lds(enc_kv_state<gf_bf16>);
}

}   // namespace

extern "C" size_t gf_encoder_kv_workspace_bytes(int N, int S) {
    if (N <= 0 || S <= 0) return 0;
    const size_t tiles = (S + TM - 1) / TM, len = (size_t)C * D + C;
    return gf_align_up(sizeof(float) * N * tiles * len, 256);
}

// k/v projection + linear-attention state of `src` [N, S, 256]: kv_state [N][256*32 + 256] fp32 (KV as [c][v], then Ksum)
extern "C" int gf_encoder_kv_state(const void* src, long ld, int dtype, int N, int S, const uint8_t* kv_mask,
                                   const void* wstream_kv, float* kv_state, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    GF_CHECK_ARG(src && wstream_kv && kv_state, "null pointer");
    GF_CHECK_ARG(N > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "the fused encoder kernels are built for 16-bit storage (GF_F16 / GF_BF16)");
    GF_CHECK_ARG((ld * 2) % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)wstream_kv % 16 == 0, "operands must be 16-byte aligned");
    if (workspace == nullptr || workspace_bytes < gf_encoder_kv_workspace_bytes(N, S)) {
        gf_set_error("gf_encoder_kv_state: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    enc_init();
    EncArgs a{};
    a.x = src; a.ldx = ld; a.N = N; a.L = S; a.S = S; a.tiles = (S + TM - 1) / TM; a.kv_mask = kv_mask; a.wstream = wstream_kv;
    a.part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int len = C * D + C;
    void* pt = gf_prof_begin("enc_kv_state", st, 2.0 * N * (double)S * C * (2.0 * C + 2.0 * D));
    if (k9_pair()) {
        gf_k9_pair_state(a, dtype, st);
        gf_k9_pair_reduce(a.part, kv_state, a.tiles, len, N, st);
    } else {
        if (dtype == GF_F16) enc_kv_state<_Float16><<<N * a.tiles, 256, W_OFF + 2 * WBLK + 8192, st>>>(a);
        else enc_kv_state<gf_bf16><<<N * a.tiles, 256, W_OFF + 2 * WBLK + 8192, st>>>(a);
        enc_kv_reduce<<<dim3((len + 255) / 256, N), 256, 0, st>>>(a.part, kv_state, a.tiles, len);
    }
    gf_prof_end("enc_kv_state", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// the encoder layer with its state tail: the layer of gf_encoder_layer (linear attention form) whose images tail_first .. N-1 also
// leave kv_state_out[n - tail_first] = the linear-attention state of their OUTPUT rows under wstream_tail (the W_k | W_v stream of
// the layer call that will read those rows as its source): no second pass over the features, no second launch
extern "C" int gf_encoder_layer_kv(const void* x, long ldx, const float* kv_state, int S, const uint8_t* q_mask, float attn_eps,
                                   const void* wstream, const float* ln_params, float eps1, float eps2, int activation, void* out,
                                   long ldo, int dtype, int N, int L, const void* wstream_tail, int tail_first, float* kv_state_out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(x && wstream && ln_params && out && kv_state, "null pointer");
    GF_CHECK_ARG(wstream_tail && kv_state_out && tail_first >= 0 && tail_first < N, "the state tail needs its weight stream, its output and 0 <= tail_first < N");
    GF_CHECK_ARG(N > 0 && L > 0 && S > 0, "empty problem");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "the fused encoder kernels are built for 16-bit storage (GF_F16 / GF_BF16)");
    GF_CHECK_ARG(activation == 0 || activation == 1, "activation: 0 = ReLU, 1 = Tanh");
    GF_CHECK_ARG((ldx * 2) % 16 == 0 && (ldo * 2) % 16 == 0, "rows must be 16-byte aligned");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)wstream % 16 == 0 && (uintptr_t)wstream_tail % 16 == 0,
                 "tensors must be 16-byte aligned");
    const int nt = N - tail_first;
    if (workspace == nullptr || workspace_bytes < gf_encoder_kv_workspace_bytes(nt, L)) {
        gf_set_error("gf_encoder_layer_kv: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    enc_init();
    EncArgs a{};
    a.x = x; a.ldx = ldx; a.kvfinal = kv_state; a.q_mask = q_mask; a.wstream = wstream; a.ln = ln_params;
    a.eps1 = eps1; a.eps2 = eps2; a.attn_eps = attn_eps; a.out = out; a.ldo = ldo; a.N = N; a.L = L; a.S = S;
    a.tiles = (L + TM - 1) / TM;
    a.wstream_tail = wstream_tail; a.tail_first = tail_first; a.part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const double per_tok = 2.0 * C * C + 2.0 * C * (D + 1) + 2.0 * C * C + 8.0 * C * C + 4.0 * C * C;
    const double tail_tok = 2.0 * C * (2.0 * C + 2.0 * D);                // the flops gf_encoder_kv_state declares per source token
    void* pt = gf_prof_begin("enc_layer", st, per_tok * N * (double)L + tail_tok * nt * (double)L);
    const int len = C * D + C;
    if (k9_pair()) {
        gf_k9_pair_layer(a, activation, true, dtype, st);
        gf_k9_pair_reduce(a.part, kv_state_out, a.tiles, len, nt, st);
    } else {
        if (dtype == GF_F16) enc_launch<_Float16>(a, activation, true, st);
        else enc_launch<gf_bf16>(a, activation, true, st);
        enc_kv_reduce<<<dim3((len + 255) / 256, nt), 256, 0, st>>>(a.part, kv_state_out, a.tiles, len);
    }
    gf_prof_end("enc_layer", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

// the encoder layer after (attn = 0) or including (attn = 1) the attention; see the header for the arguments
extern "C" int gf_encoder_layer(const void* x, long ldx, const void* msg, long ldm, const float* kv_state, int S,
                                const uint8_t* q_mask, float attn_eps, const void* wstream, const float* ln_params, float eps1,
                                float eps2, int activation, const int32_t* row_flag, int flag_rows, void* out, long ldo, int dtype,
                                int N, int L, void* stream) {
    GF_CHECK_ARG(x && wstream && ln_params && out, "null pointer");
    GF_CHECK_ARG((msg != nullptr) != (kv_state != nullptr), "exactly one of msg (attention output) and kv_state (linear attention) is given");
    GF_CHECK_ARG(N > 0 && L > 0 && (kv_state == nullptr || S > 0), "empty problem");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "the fused encoder kernels are built for 16-bit storage (GF_F16 / GF_BF16)");
    GF_CHECK_ARG(activation == 0 || activation == 1, "activation: 0 = ReLU, 1 = Tanh");
    GF_CHECK_ARG((ldx * 2) % 16 == 0 && (ldo * 2) % 16 == 0 && (msg == nullptr || (ldm * 2) % 16 == 0), "rows must be 16-byte aligned");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)msg % 16 == 0 && (uintptr_t)wstream % 16 == 0,
                 "tensors must be 16-byte aligned");
    GF_CHECK_ARG(row_flag == nullptr || flag_rows > 0, "flag_rows must be > 0");
    enc_init();
    EncArgs a{};
    a.x = x; a.ldx = ldx; a.msg = msg; a.ldm = ldm; a.kvfinal = kv_state; a.q_mask = q_mask; a.wstream = wstream; a.ln = ln_params;
    a.eps1 = eps1; a.eps2 = eps2; a.attn_eps = attn_eps; a.out = out; a.ldo = ldo; a.N = N; a.L = L; a.S = S > 0 ? S : 1;
    a.tiles = (L + TM - 1) / TM; a.flag = row_flag; a.flag_rows = flag_rows;
    hipStream_t st = (hipStream_t)stream;
    const bool attn = kv_state != nullptr;
    // flops per token: [q 2C^2 + apply 2C(D+1)] + merge 2C^2 + mlp.0 2(2C)(2C) + mlp.2 2(2C)C
    const double per_tok = (attn ? 2.0 * C * C + 2.0 * C * (D + 1) : 0.0) + 2.0 * C * C + 8.0 * C * C + 4.0 * C * C;
    void* pt = gf_prof_begin("enc_layer", st, per_tok * N * (double)L);
    if (k9_pair()) gf_k9_pair_layer(a, activation, attn, dtype, st);
    else if (dtype == GF_F16) enc_launch<_Float16>(a, activation, attn, st);
    else enc_launch<gf_bf16>(a, activation, attn, st);
    gf_prof_end("enc_layer", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
