// K9: the encoder layer as ONE launch (plus a state pass for the sources no earlier launch produced) that touches HBM three times per token
// (read source state, read x, write out) - the algorithmic minimum of SURVEY 8d - instead of the eight launches / ~19 token-row transfers of
// the K3 + K2 chain.  16-bit storage modes only (fp16 / bf16 operands, fp32 accumulation and statistics); the fp32 parity mode keeps the
// unfused kernels.  Round 6: rebuilt on CHANNEL-SPLIT WAVE PAIRS - eight waves (two per SIMD, <= 256 registers each) per 128-token
// workgroup instead of four waves with the whole 512-register file (rounds 2-5; git history).
//
// Replaces LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:37-60, ReLU, linear attention of
// linear_attention.py:21-51) and the part of its Geo twin after the attention (model/geo_transformer/transformer.py:56-66, Tanh):
//
//   enc_kv_state  k, v = W_k src, W_v src ; KV[n,h] = sum_s phi(k_s)^T v_s ; Ksum[n,h] = sum_s phi(k_s)   (one partial per tile)
//   enc_layer        q = W_q x ; msg = phi(q) KV / (phi(q).Ksum + eps)            (ATTN: linear attention)
//                   or msg = attention output read from HBM                        (Geo layers: K4 / K5 made it)
//                   m = LN1(W_m msg) ; hid = act(W_1 [x | m]) ; out = x + LN2(W_2 hid)   [+ the state of `out` for its consumer]
//
// Why pairs.  Rounds 2-5 ran one wave per SIMD: a wave owned 32 tokens x all 256 channels, so every LayerNorm / phi / packing phase
// (~3300 vector instructions per tile) ran with the matrix pipe idle and every LDS-DMA piece or fragment read stretched an MFMA gap
// (0.27 of the MFMA peak for three rounds).  Here a 32-token group belongs to TWO waves, wave half w = 0 / 1 owning the OUTPUT channels
// [128 w, 128 w + 128) of every product: 64 accumulator registers per product instead of 128, no accumulator in AGPRs (no
// v_accvgpr moves), and a SIMD always has a second wave whose MFMAs cover the first one's reads, requests and waits.
// The contraction of the next product runs over ALL 256 channels, so the two waves of a pair exchange their packed 16-bit operands
// through a 32-KiB LDS region between the products (msg, LN1 output, hidden slices: 4 KiB per wave and round); LayerNorm exchanges
// two partial sums per token.  Every accumulator still adds its k-steps in ascending order.
//
// Weights: the host packs them (geoformer_amd/fused.py:pack_layer_stream) as a stream of 1-KiB MFMA A fragments in 16-KiB BLOCKS:
// fragments 0..7 of a block belong to wave half 0, 8..15 to half 1, each half in the order its waves multiply them.  Every workgroup
// streams the 1 MiB from L2 through a THREE-slot LDS ring by LDS-DMA (two 1-KiB pieces per wave and block, issued two blocks = ~1000
// cycles ahead), one workgroup barrier per block of 8 MFMAs per wave.  The token tile x stays in LDS for the whole kernel (B operand
// of the q and mlp.0 products, residual); the finished rows replace it in place and leave for HBM from there.
#include <math.h>

#include <type_traits>

#include "gf_common.h"
#include "k9_args.h"

// -DK9P_TRACE=1 records s_memtime at the phase boundaries of every wave of the first 1024 workgroups (tools/k9p_trace.py); a diagnostic
// build only.  -DK9P_ABLATE=n (wrong results by construction, timing only): 1 = no ring requests behind the prologue's, 2 = no barrier
// in the ring's turn, 3 = no fragment reads, 4 = no MFMAs.
#ifndef K9P_TRACE
#define K9P_TRACE 0
#endif
#ifndef K9P_ABLATE
#define K9P_ABLATE 0
#endif
#if K9P_TRACE
__device__ long long k9p_trace[1024 * 8 * 24];
#define K9P_T(slot) do { if (lane == 0 && blockIdx.x < 1024) k9p_trace[(blockIdx.x * 8 + wave) * 24 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int gf_debug_k9p_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k9p_trace), sizeof(long long) * 1024 * 8 * 24);
}
#else
#define K9P_T(slot)
#endif

namespace {

constexpr int C = 256, D = 32, NH = 8;
constexpr int TM = 128;                            // tokens per workgroup
constexpr int FRAG = 1024, BLK = 16 * FRAG;        // a weight block = 16 fragments of 64 lanes x 16 B (8 per wave half)
constexpr int X_OFF = 0;                           // [4 k-chunks][128 rows][128 B], chunk-swizzled (gf_lds_off)
constexpr int W_OFF = 65536;                       // three weight blocks
constexpr int E_OFF = W_OFF + 3 * BLK;             // exchange region: 4 pairs x 8 fragment slots (slots 0-3 wave half 0, 4-7 half 1)
constexpr int VEC_OFF = E_OFF + 32768;             // gamma1 | beta1 | gamma2 | beta2, [4][256] float
constexpr int KS_OFF = VEC_OFF + 4 * C * 4;        // [256] T: Ksum/S rounded to the storage type, in operand order
constexpr int ST_OFF = KS_OFF + 512;               // [8 waves][64 lanes] float2: LayerNorm partial sums
constexpr int LDS_BYTES = ST_OFF + 8 * 64 * 8;     // 156,160 B
constexpr int NB_Q = 8, NB_M = 8, NB_SLICE = 12, NB_KV = 16;   // blocks per phase (all even, NB_SLICE % 3 == 0)

using EncArgs = GfEncArgs;

// eight fp32 values -> one 16-byte operand; converted in pairs (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even)
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
// elu(x) + 1 (linear_attention.py:33-34) = max(x, 0) + exp(min(x, 0)): branch-free, hardware exponential with the minimum as its
// CLAMP output modifier (for x <= 0 the clamp does nothing, for x > 0 both give exactly 1)
__device__ __forceinline__ float phi(float x) {
    return fmaxf(x, 0.f) + __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * 1.44269504088896341f), 0.f, 1.f);
}
__device__ __forceinline__ v16f zero16() { return v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
// the lane id recomputed where a late phase needs it (two VALU instructions; volatile: neither hoisted nor merged)
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// x[lane] + x[lane ^ 32] in every lane: one v_permlane32_swap
__device__ __forceinline__ float half_sum(float x) {
    const gf_v2u sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(sw.x) + __uint_as_float(sw.y);
}
// sum_j a_j b_j over the 8 sixteen-bit elements of two packed operands, fp32 accumulation (the products of two 16-bit values are
// exact in fp32)
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
struct PMma {                     // the ring products' MFMA (ablation 4: compiled out, operands kept alive)
    static __device__ __forceinline__ void mma(const typename Mma32<T>::Frag& a, const typename Mma32<T>::Frag& b, v16f& c) {
#if K9P_ABLATE == 4
        asm volatile("" : "+v"(c) : "v"(a), "v"(b));
#else
        Mma32<T>::mma(a, b, c);
#endif
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// The weight ring.  Block b lives in slot b % 3.  While block b is multiplied, block b + 1 has landed (it is read into registers
// behind the MFMAs of block b) and block b + 2 is in flight; the turn inside block b (a workgroup barrier after every wave has
// ALL of block b's fragments in registers) frees slot b % 3 for block b + 3.  Wave v moves fragments 2v, 2v + 1 of every block.
// Requests behind the stream's end are out of the buffer's range: they fetch nothing and leave zeros in a slot nobody reads.
// LDS-DMA in the MUBUF form (buffer_load_dwordx4 ... lds): the compiler counts LDS waits behind it (behind the FLAT form every
// wait becomes lgkmcnt(0)); descriptor and block offset are scalar.
// ---------------------------------------------------------------------------------------------------------------------------
struct Ring {
    __amdgpu_buffer_rsrc_t rs;    // the stream as a raw buffer of exactly nblk blocks
    __amdgpu_buffer_rsrc_t rs2;   // blocks nblk .. come from a second stream (the state tail's W_k | W_v)
    char* smem;
    int voff;                     // lane * 16 + wave * 2 KiB: this lane's bytes inside a block
    int wave;
    int rbase;                    // W_OFF + (wave & 1) * 8 KiB + lane * 16: this lane's fragment reads inside a slot
    int blk, nblk;
};
__device__ __forceinline__ Ring ring_make(const void* wstream, char* smem, int wave, int lane, int nblk, const void* wstream2, int nblk2) {
    return Ring{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wstream), 0, nblk * BLK, 0x00020000),
                __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wstream2 ? wstream2 : wstream), 0, wstream2 ? nblk2 * BLK : 0, 0x00020000),
                smem, lane * 16 + wave * 2 * FRAG, wave, W_OFF + (wave & 1) * 8 * FRAG + lane * 16, 0, nblk};
}
// piece i (0, 1) of this wave of block b into slot `slot`
__device__ __forceinline__ void dma_piece(const Ring& g, int slot, int b, int i) {
#if K9P_ABLATE == 1
    if (b >= 3) return;
#endif
    const bool second = b >= g.nblk;                  // wave-uniform: the descriptor select is scalar
    __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? g.rs2 : g.rs,
                                             (__attribute__((address_space(3))) void*)(g.smem + W_OFF + slot * BLK + (g.wave * 2 + i) * FRAG), 16,
                                             g.voff, (second ? b - g.nblk : b) * BLK + i * FRAG, 0, 0);
}
// all but the VM youngest vector-memory operations of this wave are done (its pieces of the next block have landed), every LDS
// read it issued has returned; then the workgroup's barrier
template <int VM>
__device__ __forceinline__ void turn() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
#if K9P_ABLATE != 2
    __builtin_amdgcn_s_barrier();
#endif
}
// exchange barrier: my LDS writes / reads are done, then everybody's
__device__ __forceinline__ void xbar() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
template <typename Frag>
__device__ __forceinline__ Frag lds_frag(const char* p) { return *reinterpret_cast<const Frag*>(p); }
template <typename Frag>
__device__ __forceinline__ Frag ring_frag(const char* p) {
#if K9P_ABLATE == 3
    Frag f; asm volatile("" : "=v"(f)); return f;
#else
    return *reinterpret_cast<const Frag*>(p);
#endif
}

// One BLOCK = the wave's 8 MFMAs `mma(0..7)` on the fragments F[0..7] of the block in slot `slot`.  F is ONE rotating set of eight
// operands: fragments 0..5 of a block are read behind MFMAs 4..7 of the block before it (behind the ring's turn), fragments 6, 7 behind
// its own MFMAs 0, 1 - their registers hold the previous block's 6, 7 until then:
//   MFMA 0 + read 6 (+ h0's operand reads) | MFMA 1 + read 7 | MFMA 2 | MFMA 3 | wait + barrier |
//   MFMA 4 + piece 0 + reads 0', 1' (+ h4's) | MFMA 5 + piece 1 + reads 2', 3' | MFMA 6 + read 4' | MFMA 7 + read 5'
// (x' = fragment x of the next block).  The turn after MFMA 3: every wave has ALL of this block's fragments in registers, so its slot is
// free for block b + 3; the four MFMAs behind it cover the first reads of the next block.  NX0 / NX4 = LDS reads issued by the hooks
// h0 / h4 (pinned behind MFMA 0 / MFMA 4); VM: see turn().
#define K9_GAP(DMA, READS, VALU)                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
    if (DMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);             \
    if (READS) __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);       \
    if (VALU) __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
// NV0 / NV4: vector instructions of the hooks to place behind each of MFMAs 0..3 / 4..7 (the hooks' arithmetic rides in the gaps)
template <int NX0, int NX4, int VM, int NV0 = 0, int NV4 = 0, typename Frag, typename MF, typename H0, typename H4>
__device__ __forceinline__ void pblock(Ring& g, Frag (&F)[8], int slot, MF mma, H0 h0, H4 h4) {
    const char* pc = g.smem + g.rbase + slot * BLK;
    mma(0);
    F[6] = ring_frag<Frag>(pc + 6 * FRAG);
    h0();
    mma(1);
    F[7] = ring_frag<Frag>(pc + 7 * FRAG);
    mma(2);
    mma(3);
    K9_GAP(false, 1 + NX0, NV0) K9_GAP(false, 1, NV0) K9_GAP(false, 0, NV0) K9_GAP(false, 0, NV0)
    __builtin_amdgcn_sched_barrier(0);
    turn<VM>();
    const char* p = g.smem + g.rbase + ((slot + 1) % 3) * BLK;
    mma(4);
    dma_piece(g, slot, g.blk + 3, 0);
    F[0] = ring_frag<Frag>(p);
    F[1] = ring_frag<Frag>(p + FRAG);
    h4();
    mma(5);
    dma_piece(g, slot, g.blk + 3, 1);
    F[2] = ring_frag<Frag>(p + 2 * FRAG);
    F[3] = ring_frag<Frag>(p + 3 * FRAG);
    mma(6);
    F[4] = ring_frag<Frag>(p + 4 * FRAG);
    mma(7);
    F[5] = ring_frag<Frag>(p + 5 * FRAG);
    K9_GAP(true, 2 + NX4, NV4) K9_GAP(true, 2, NV4) K9_GAP(false, 1, NV4) K9_GAP(false, 1, NV4)
    __builtin_amdgcn_sched_barrier(0);
    ++g.blk;
}
#undef K9_GAP
// prologue: blocks 0, 1, 2 requested
__device__ __forceinline__ void ring_start(const Ring& g) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        dma_piece(g, b, b, 0);
        dma_piece(g, b, b, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <typename Frag>
__device__ __forceinline__ void load_block0(const Ring& g, Frag (&f)[8]) {      // fragments 0..5 of block 0 (6, 7: inside the block)
    const char* p = g.smem + g.rbase;
#pragma unroll
    for (int i = 0; i < 6; ++i) f[i] = ring_frag<Frag>(p + i * FRAG);
}

// token tile -> LDS by LDS-DMA: [kc][row][128 B] with the 16-B chunk index XORed with (row >> 1) & 7 (conflict-free ds_read_b128).
// A piece = 8 rows of one 64-channel plane (1 KiB, lane-linear): lane l fills row 8 p + (l >> 3), slot l & 7 with the row's chunk
// (l & 7) ^ swizzle - the swizzle is applied on the SOURCE side.  `pieces` per wave; rows past the sequence are clamped to its last.
template <typename T>
__device__ __forceinline__ void tile_dma(const T* base, long ld, int rows_total, int row0, int nrows, char* smem, int off, int wave, int lane,
                                         int waves) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, 0x7FFFFFFF, 0x00020000);
    const int planes_pieces = nrows / 8;                                   // pieces per 64-channel plane
    const int total = 4 * planes_pieces, per = total / waves;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < per) {
            const int q = wave * per + i, kc = q / planes_pieces, rp = q - kc * planes_pieces;
            const int row = rp * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            const int r = min(row0 + row, rows_total - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + off + kc * (nrows * 128) + rp * 1024), 16,
                                                     (int)(r * ld * 2) + kc * 128 + c * 16, 0, 0, 0);
        }
    }
}

template <typename T>
__device__ __forceinline__ void kv_tail(Ring& ring, typename Mma32<T>::Frag (&F)[8], char* smem, unsigned valid,
                                        float* dst, int wave, int lane, int tid, int slot0);

template <typename T, int ACT, bool ATTN>
__global__ __launch_bounds__(512) void enc_layer(EncArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, lr = lane & 31;
    const int w = wave & 1, grp = wave >> 1;                           // channel half, 32-token group
    const int n = blockIdx.x / a.tiles, tile = blockIdx.x - n * a.tiles;
    const int t0 = tile * TM;                                         // first token of the tile inside image n
    const T* xg = (const T*)a.x + (size_t)n * a.L * a.ldx;
    constexpr int NBLK = (ATTN ? NB_Q : 0) + NB_M + 4 * NB_SLICE;     // 64 / 56
    float* vec = reinterpret_cast<float*>(smem + VEC_OFF);
    const bool tail = ATTN && a.wstream_tail != nullptr && n >= a.tail_first;
    K9P_T(0);
    Ring ring = ring_make(a.wstream, smem, wave, lane, NBLK, tail ? a.wstream_tail : nullptr, NB_KV);
    const int mytok = t0 + grp * 32 + lr;                             // this lane's token (accumulator column)
    const int myrow = grp * 32 + lr;
    // the prologue's plain loads go out FIRST: the waits of their consumers then do not cover the LDS-DMA requests behind them
    const float lnv0 = a.ln[tid], lnv1 = a.ln[512 + tid];
    unsigned char qm = 1;
    if (a.q_mask != nullptr) qm = a.q_mask[(size_t)n * a.L + min(mytok, a.L - 1)];
    float kvv[2][8], ksv = 0.f;
    if constexpr (ATTN) {
        const float* kvf = a.kvfinal + (size_t)n * (C * D + C);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int sl = p * 512 + tid, ln = sl & 63, hs = sl >> 6, hh = hs >> 1, s = hs & 1;
            const float* src = kvf + (size_t)(hh * D + 16 * s + 4 * (ln >> 5)) * D + (ln & 31);       // KV[c = (hh, d)][v]
#pragma unroll
            for (int j = 0; j < 8; ++j) kvv[p][j] = src[((j >> 2) * 8 + (j & 3)) * D];
        }
        {
            const int tq = tid & 255, j = tq & 7, hx = (tq >> 3) & 1, sx = (tq >> 4) & 1, hh = tq >> 5;
            ksv = kvf[C * D + hh * D + 16 * sx + 8 * (j >> 2) + 4 * hx + (j & 3)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the token tile and weight block 0 only: a CU's LDS-DMA moves ~30 B per cycle, so the 112 KiB of "tile + three blocks" took ~3.9 thousand
    // cycles to REQUEST (round 6 trace); blocks 1 and 2 go out behind the prologue's barrier, in front of the first MFMAs
    tile_dma<T>(xg, a.ldx, a.L, t0, TM, smem, X_OFF, wave, lane, 8);
    dma_piece(ring, 0, 0, 0);
    dma_piece(ring, 0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    K9P_T(13);
    vec[tid] = lnv0;
    vec[512 + tid] = lnv1;
    const float qmul = qm != 0 ? 1.f : 0.f;                            // masked query -> 0
    if constexpr (ATTN) {
        // KV / S and Ksum / S of image n in the storage type (the reference's "prevent fp16 overflow" scaling,
        // linear_attention.py:45-49: out = (Q.KV/S) / (Q.Ksum/S + eps/S)), KV as A fragments in the k order of a
        // packed accumulator: lane (v = lr, h2), element j  <->  d = 16s + 8(j>>2) + 4 h2 + (j&3); parked in the exchange region
        // (free until the first exchange), every wave then takes the fragments of its four heads into registers
        const float inv_s = 1.0f / (float)a.S;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int sl = p * 512 + tid, ln = sl & 63, hs = sl >> 6;
            *reinterpret_cast<Frag*>(smem + E_OFF + hs * FRAG + ln * 16) =
                pack8<T>(kvv[p][0] * inv_s, kvv[p][1] * inv_s, kvv[p][2] * inv_s, kvv[p][3] * inv_s, kvv[p][4] * inv_s, kvv[p][5] * inv_s,
                         kvv[p][6] * inv_s, kvv[p][7] * inv_s);
        }
        // Ksum / S rounded to the storage type, as 16-byte operands in the k order of the packed phi(q): entry (hh, s, h2),
        // element j = channel 32 hh + 16 s + 8 (j >> 2) + 4 h2 + (j & 3)   (both halves of the workgroup write the same 256 values)
        reinterpret_cast<T*>(smem + KS_OFF)[tid & 255] = gf_from_float<T>(ksv * inv_s);
    }
    // the B operands of the merge product (k-step ks = 2 head + s), then of the second half of mlp.0: ALL 16 k-steps in both waves
    Frag ma[16];
    if constexpr (!ATTN) {
        // the attention output of K4 / K5 comes from HBM: the 128 x 256 tile goes through the exchange region in two 64-token
        // halves ([kc][64 rows][128 B], swizzled like the x tile); both waves of a pair take all 16 operands of their 32 tokens
        const T* mg = (const T*)a.msg + (size_t)n * a.L * a.ldm;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            tile_dma<T>(mg, a.ldm, a.L, t0 + half * 64, 64, smem, E_OFF, wave, lane, 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if ((grp >> 1) == half) {      // groups 0, 1 own rows 0..63, groups 2, 3 rows 64..127
                const int row = (grp & 1) * 32 + lr;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        // channels 32t + 16s + 4h2 + {0..3} and + 8: two 8-byte pieces of chunk (32t + 16s) / 8 and the next
                        const int c0 = 32 * t + 16 * s + 4 * h2, ch = c0 >> 3;           // c0 % 8 is 0 or 4
                        const char* p0 = smem + E_OFF + (ch >> 3) * 8192 + gf_lds_off(row, ch & 7) + (c0 & 7) * 2;
                        const char* p1 = smem + E_OFF + ((ch + 1) >> 3) * 8192 + gf_lds_off(row, (ch + 1) & 7) + (c0 & 7) * 2;
                        typedef short v4s __attribute__((ext_vector_type(4)));
                        typedef short v8s __attribute__((ext_vector_type(8)));
                        const v4s lo = *reinterpret_cast<const v4s*>(p0), hi = *reinterpret_cast<const v4s*>(p1);
                        const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        ma[2 * t + s] = __builtin_bit_cast(Frag, both);
                    }
            }
            xbar();
        }
    }
    // block 0 and the tile have landed
    K9P_T(14);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    K9P_T(15);
    __builtin_amdgcn_s_barrier();
    dma_piece(ring, 1, 1, 0);
    dma_piece(ring, 1, 1, 1);
    dma_piece(ring, 2, 2, 0);
    dma_piece(ring, 2, 2, 1);
    __builtin_amdgcn_sched_barrier(0);

    K9P_T(1);
    Frag F[8];                                                        // the rotating weight fragments (see pblock)
    load_block0(ring, F);
    // token-tile operand of 16-deep k-step ks (0..15)
    auto xfrag = [&](int ks) { return lds_frag<Frag>(smem + X_OFF + (ks >> 2) * 16384 + gf_lds_off(myrow, 2 * (ks & 3) + h2)); };
    // the pair's exchange slots (8 x 1 KiB): wave half 0 writes slots 0-3, half 1 slots 4-7, both read all eight
    char* const ex = smem + E_OFF + grp * 8192 + lane * 16;
    auto ex_put = [&](int i, const Frag& f) { *reinterpret_cast<Frag*>(ex + (4 * w + i) * FRAG) = f; };
    auto ex_get = [&](int s) { return lds_frag<Frag>(ex + s * FRAG); };
    int slot = 0;                                                     // slot of the current block (compile-time after unrolling)

    // ---------------- linear attention (ATTN): q = W_q x HEAD-MAJOR - block b multiplies k-steps 8 (b & 1) .. + 7 of head t = b >> 1 (the wave's
    // head 4 w + t), its token-tile operands X16 in registers - so that the attention of head t (phi, packing, normaliser, two state MFMAs,
    // normalisation: ~110 vector instructions) rides in the MFMA gaps of head t + 1's two blocks (head 3's in the first two merge blocks)
    // instead of a phase of its own with the matrix pipe idle.  The packed message of head t crosses the pair through slot set t & 1 of the
    // exchange region (written behind the turn of block 2 t + 3, read behind the turn of block 2 t + 4: no barrier of its own); the merge
    // product's block h multiplies head h, whose operands both waves hold by then.
    Frag kvf[4][2];
    v16f qacc[2];                                                     // q of the head being multiplied / of the head in the attention
    v16f num;
    float den = 0.f;
    Frag p0, p1, msg0, msg1;
    const float eps_s = ATTN ? a.attn_eps / (float)a.S : 0.f;
    auto att_a0 = [&](int t) {                                        // phi(q) of registers 0..7: rounded by its packing; the normaliser from the PACKED operands
        const int hh = 4 * w + t;
        const Frag ks0 = lds_frag<Frag>(smem + KS_OFF + ((hh * 2 + 0) * 2 + h2) * 16);
        const v16f& qh = qacc[t & 1];
        p0 = pack8<T>(phi(qh[0]), phi(qh[1]), phi(qh[2]), phi(qh[3]), phi(qh[4]), phi(qh[5]), phi(qh[6]), phi(qh[7]));
        den = dot8(p0, ks0, 0.f);
    };
    auto att_a1 = [&](int t) {
        const int hh = 4 * w + t;
        const Frag ks1 = lds_frag<Frag>(smem + KS_OFF + ((hh * 2 + 1) * 2 + h2) * 16);
        const v16f& qh = qacc[t & 1];
        p1 = pack8<T>(phi(qh[8]), phi(qh[9]), phi(qh[10]), phi(qh[11]), phi(qh[12]), phi(qh[13]), phi(qh[14]), phi(qh[15]));
        den = dot8(p1, ks1, den);
        num = zero16();
        Mm::mma(kvf[t][0], p0, num);
        Mm::mma(kvf[t][1], p1, num);
    };
    auto att_b0 = [&] {                                                // a masked query (phi(q) = 0 in linear_attention.py:35-36) is zeroed through its normaliser
        const float z = __builtin_amdgcn_rcpf(half_sum(den) + eps_s) * qmul;
#pragma unroll
        for (int r = 0; r < 16; ++r) num[r] *= z;
        msg0 = pack_step<T>(num, 0);
        msg1 = pack_step<T>(num, 1);
    };
    auto att_put = [&](int t) {                                        // slot set t & 1: slots 4 w + 2 (t & 1) + {0, 1}
        *reinterpret_cast<Frag*>(ex + (4 * w + 2 * (t & 1)) * FRAG) = msg0;
        *reinterpret_cast<Frag*>(ex + (4 * w + 2 * (t & 1) + 1) * FRAG) = msg1;
    };
    auto att_get = [&](int t) {                                        // heads t (wave half 0) and 4 + t (half 1): k-steps 2 t, 2 t + 1 and 8 + 2 t, 9 + 2 t
        ma[2 * t] = ex_get(2 * (t & 1));
        ma[2 * t + 1] = ex_get(2 * (t & 1) + 1);
        ma[8 + 2 * t] = ex_get(4 + 2 * (t & 1));
        ma[9 + 2 * t] = ex_get(5 + 2 * (t & 1));
    };
    if constexpr (ATTN) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) kvf[t][s] = lds_frag<Frag>(smem + E_OFF + ((4 * w + t) * 2 + s) * FRAG + lane * 16);
        Frag X16[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) X16[ks] = xfrag(ks);
#pragma unroll
        for (int b = 0; b < NB_Q; ++b) {
            const int t = b >> 1, hf = b & 1;
            if (hf == 0) qacc[t & 1] = zero16();
            auto mm = [&](int j) { PMma<T>::mma(F[j], X16[8 * hf + j], qacc[t & 1]); };
            if (b < 2) pblock<0, 0, 2>(ring, F, slot, mm, [] {}, [] {});
            else if (hf == 0) pblock<1, 5, 2, 10, 10>(ring, F, slot, mm, [&] { att_a0(t - 1); }, [&] { att_a1(t - 1); if (t >= 2) att_get(t - 2); });
            else pblock<0, 0, 2, 8, 0>(ring, F, slot, mm, [&] { att_b0(); }, [&] { att_put(t - 1); });
            slot = (slot + 1) % 3;
        }
        K9P_T(2);
        K9P_T(3);
    }

    // ---------------- m = W_m msg for the wave's 128 channels: 8 blocks of (2 k-steps x 4 tiles), operands from registers
    v16f m[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) m[t] = zero16();
    Frag X[4];                                                        // token-tile operands of a W_1 block: X[kk] for MFMAs 2 kk, 2 kk + 1
#pragma unroll
    for (int b = 0; b < NB_M; ++b) {
        auto mm = [&](int j) { PMma<T>::mma(F[j], ma[2 * b + (j >> 2)], m[j & 3]); };
        if (ATTN && b == 0) pblock<1, 5, 2, 10, 10>(ring, F, slot, mm, [&] { att_a0(3); }, [&] { att_a1(3); att_get(2); });
        else if (ATTN && b == 1) pblock<0, 0, 2, 8, 0>(ring, F, slot, mm, [&] { att_b0(); }, [&] { att_put(3); });
        else if (ATTN && b == 2) pblock<0, 4, 2>(ring, F, slot, mm, [] {}, [&] { att_get(3); });
        else if (b < NB_M - 1) pblock<0, 0, 2>(ring, F, slot, mm, [] {}, [] {});
        else pblock<0, 2, 2>(ring, F, slot, mm, [] {}, [&] { X[0] = xfrag(0); X[1] = xfrag(1); });      // mlp.0's first operands
        slot = (slot + 1) % 3;
    }
    K9P_T(4);
    // nn.LayerNorm over the 256 channels of the lane's token, statistics in fp32: this lane half holds 64 of them, the other half 64,
    // the pair's other wave 128: sum and sum of squares per wave (one pass, var = E[x^2] - mean^2), the two waves' partials through LDS
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto stats_add = [&](const v16f& t, float& s, float& qd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = t[r];
            s += v;
            qd = fmaf(v, v, qd);
        }
    };
    auto stats_put = [&](float s, float qd) { *reinterpret_cast<f2*>(smem + ST_OFF + (wave * 64 + lane) * 8) = f2{s, qd}; };
    auto stats_finish = [&](float s, float qd, float eps, float& mean, float& rstd) {
        const f2 o = *reinterpret_cast<const f2*>(smem + ST_OFF + ((wave ^ 1) * 64 + lane) * 8);
        s += o.x;                                                      // a + b == b + a: both waves of the pair get the same bits
        qd += o.y;
        mean = s * (1.0f / C);
        rstd = 1.0f / sqrtf(fmaxf(qd - s * mean, 0.f) * (1.0f / C) + eps);
    };
    // gamma / beta of the 8 channels nb*32 + 8g + 4 h2 + {0..3}, g = 2 sx, 2 sx + 1 (one packed operand = half a channel tile)
    struct LnOps { v4f ga[2], be[2]; };
    auto ln_ops = [&](const float* gamma, const float* beta, int nb, int sx) {
        LnOps o;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            o.ga[g] = *reinterpret_cast<const v4f*>(gamma + nb * 32 + 8 * (2 * sx + g) + 4 * h2);
            o.be[g] = *reinterpret_cast<const v4f*>(beta + nb * 32 + 8 * (2 * sx + g) + 4 * h2);
        }
        return o;
    };
    // (t - mean) rstd gamma + beta as two FMAs per value: t rstd - mean rstd, then times gamma plus beta (`nmr` = -mean rstd)
    auto ln_apply = [&](const v16f& t, int g, const v4f& ga, const v4f& be, float nmr, float rstd) {
        return v4f{fmaf(fmaf(t[4 * g], rstd, nmr), ga.x, be.x), fmaf(fmaf(t[4 * g + 1], rstd, nmr), ga.y, be.y),
                   fmaf(fmaf(t[4 * g + 2], rstd, nmr), ga.z, be.z), fmaf(fmaf(t[4 * g + 3], rstd, nmr), ga.w, be.w)};
    };
    // LN1 of tile t, operand sx -> one packed B operand of mlp.0's second half
    auto ln1_pack = [&](int t, int sx, float nmr, float rstd) {
        const LnOps p = ln_ops(vec, vec + C, 4 * w + t, sx);
        const v4f lo = ln_apply(m[t], 2 * sx, p.ga[0], p.be[0], nmr, rstd), hi = ln_apply(m[t], 2 * sx + 1, p.ga[1], p.be[1], nmr, rstd);
        return pack8<T>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
    };

    // ---------------- hid = act(W_1 [x | m]) in four 128-wide slices (the wave: 64 of them = 2 tiles), each consumed by
    // out += W_2[:, slice] hid (the wave's 4 output tiles) - SOFTWARE-PIPELINED so that no product waits for an exchange:
    //   stage 0: W_1x(0) [LN1 and the exchange of its output ride in these four blocks] W_1m(0)
    //   stage s = 1..3: W_1x(s) W_2(s - 1) W_1m(s)        (slice s - 1's packed hidden operands cross the pair behind W_1x(s)'s turns)
    //   last: W_2(3)
    // W_1x / W_1m block: 4 k-steps x the wave's 2 hidden tiles; W_2 block: 2 k-steps x its 4 output tiles.
    v16f o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = zero16();
    v16f hd[2];
    auto w1x = [&](int j) { PMma<T>::mma(F[j], X[j >> 1], hd[j & 1]); };
    auto xh0 = [&](int b) { X[2] = xfrag(4 * b + 2); X[3] = xfrag(4 * b + 3); };
    auto xh4 = [&](int b) { X[0] = xfrag(4 * b + 4); X[1] = xfrag(4 * b + 5); };
    // activation + packing of the wave's two hidden tiles into the pair's exchange slots: for ReLU on the PACKED operand
    // (relu(round(x)) = round(relu(x)))
    auto act_put = [&] {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                Frag f;
                if constexpr (ACT == 0) {
                    f = relu_packed(pack_step<T>(hd[t], sx));
                } else {
#pragma unroll
                    for (int r = 8 * sx; r < 8 * sx + 8; ++r) hd[t][r] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * hd[t][r]) + 1.0f);   // tanh, ~1e-6 abs
                    f = pack_step<T>(hd[t], sx);
                }
                ex_put(2 * t + sx, f);
            }
    };
    {
        // ----- stage 0
        hd[0] = zero16();
        hd[1] = zero16();
        float ls = 0.f, lq = 0.f, nmr = 0.f, rstd = 0.f;
        Frag mo[4];                                                   // the wave's LN1 operands of tiles 2, 3 until their round
        pblock<2, 2, 2, 8, 10>(ring, F, slot, w1x, [&] { xh0(0); stats_add(m[0], ls, lq); stats_add(m[1], ls, lq); }, [&] {
            xh4(0);
            stats_add(m[2], ls, lq);
            stats_add(m[3], ls, lq);
            ls = half_sum(ls);
            lq = half_sum(lq);
            stats_put(ls, lq);
        });
        slot = (slot + 1) % 3;
        pblock<2, 3, 2, 0, 12>(ring, F, slot, w1x, [&] { xh0(1); }, [&] {
            xh4(1);
            float mean;
            stats_finish(ls, lq, a.eps1, mean, rstd);
            nmr = -mean * rstd;
#pragma unroll
            for (int i = 0; i < 4; ++i) ex_put(i, ln1_pack(i >> 1, i & 1, nmr, rstd));        // round A: the wave's k-steps 0..3
        });
        slot = (slot + 1) % 3;
        pblock<2, 10, 2, 12, 0>(ring, F, slot, w1x, [&] {
            xh0(2);
#pragma unroll
            for (int i = 0; i < 4; ++i) mo[i] = ln1_pack(2 + (i >> 1), i & 1, nmr, rstd);
        }, [&] {
            xh4(2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ma[i] = ex_get(i);
                ma[8 + i] = ex_get(4 + i);
            }
        });
        slot = (slot + 1) % 3;
        pblock<2, 0, 2>(ring, F, slot, w1x, [&] { xh0(3); }, [&] {
#pragma unroll
            for (int i = 0; i < 4; ++i) ex_put(i, mo[i]);                                       // round B: its k-steps 4..7
        });
        slot = (slot + 1) % 3;
#if K9P_TRACE
        K9P_T(6);
#endif
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            auto mm = [&](int j) { PMma<T>::mma(F[j], ma[4 * b + (j >> 1)], hd[j & 1]); };
            if (b == 0) {
                pblock<0, 8, 2>(ring, F, slot, mm, [] {}, [&] {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ma[4 + i] = ex_get(i);
                        ma[12 + i] = ex_get(4 + i);
                    }
                });
            } else if (b < 3) {
                pblock<0, 0, 2>(ring, F, slot, mm, [] {}, [] {});
            } else {
                pblock<0, 2, 2>(ring, F, slot, mm, [] {}, [&] { X[0] = xfrag(0); X[1] = xfrag(1); });
            }
            slot = (slot + 1) % 3;
        }
        act_put();
#if K9P_TRACE
        K9P_T(7);
#endif
    }
    K9P_T(5);
    constexpr int STAGE_SLOT = ((ATTN ? NB_Q : 0) + NB_M + 8) % 3;   // 12 blocks per stage: every stage starts in the same slot
    Frag H[2];                                                        // hidden operands of a W_2 block: H[0] for MFMAs 0..3, H[1] for 4..7
#pragma unroll 1
    for (int st = 1; st < 4; ++st) {
        int slot = STAGE_SLOT;
        v16f hn[2];                                                   // this stage's hidden accumulators (hd: consumed by act_put above)
        hn[0] = zero16();
        hn[1] = zero16();
        auto w1xn = [&](int j) { PMma<T>::mma(F[j], X[j >> 1], hn[j & 1]); };
#pragma unroll
        for (int b = 0; b < 4; ++b) {                                   // W_1x(st); behind its first turn the hidden operands of st - 1 are visible
            if (b < 3) pblock<2, 2, 2>(ring, F, slot, w1xn, [&] { xh0(b); }, [&] { xh4(b); });
            else pblock<2, 1, 2>(ring, F, slot, w1xn, [&] { xh0(b); }, [&] { H[0] = ex_get(0); });
            slot = (slot + 1) % 3;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {                                   // out += W_2[:, 128 (st - 1) + 32 b .. + 31] hid(st - 1)
            pblock<1, 1, 2>(ring, F, slot, [&](int j) { PMma<T>::mma(F[j], H[j >> 2], o[j & 3]); }, [&] { H[1] = ex_get(2 * b + 1); },
                            [&] { if (b < 3) H[0] = ex_get(2 * b + 2); });
            slot = (slot + 1) % 3;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {                                   // W_1m(st)
            auto mm = [&](int j) { PMma<T>::mma(F[j], ma[4 * b + (j >> 1)], hn[j & 1]); };
            if (b < 3) pblock<0, 0, 2>(ring, F, slot, mm, [] {}, [] {});
            else pblock<0, 2, 2>(ring, F, slot, mm, [] {}, [&] { X[0] = xfrag(0); X[1] = xfrag(1); });
            slot = (slot + 1) % 3;
        }
        hd[0] = hn[0];
        hd[1] = hn[1];
        act_put();
    }
    K9P_T(8);
    {
        // ----- last: W_2(3).  Its operands cross the pair behind one exchange barrier (no ring turn in between)
        xbar();
        int slot = STAGE_SLOT;
        H[0] = ex_get(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            pblock<1, 1, 2>(ring, F, slot, [&](int j) { PMma<T>::mma(F[j], H[j >> 2], o[j & 3]); }, [&] { H[1] = ex_get(2 * b + 1); },
                            [&] { if (b < 3) H[0] = ex_get(2 * b + 2); });
            slot = (slot + 1) % 3;
        }
    }
    K9P_T(10);

    // ---------------- out = x + LN2(.) (one rounding), per-sample skip predicate.  The finished values replace the x tile IN PLACE
    // (a lane overwrites exactly the residual bytes it has just read) and leave for HBM from there: the wave's 32 rows x 256 B
    // (its channel half), four rows per store instruction.  Only the wave's own bytes are touched: no workgroup barrier.
    float mean2, rstd2;
    {
        float s2 = 0.f, q2 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) stats_add(o[t], s2, q2);
        s2 = half_sum(s2);
        q2 = half_sum(q2);
        stats_put(s2, q2);
        xbar();
        stats_finish(s2, q2, a.eps2, mean2, rstd2);
    }
    const float nmr2 = -mean2 * rstd2;
    K9P_T(11);
    // the predicate is per sample (flag_rows is a multiple of L in every caller): read it once, wave-uniformly
    const bool keep = a.flag == nullptr || __builtin_amdgcn_readfirstlane(a.flag[((size_t)n * a.L + t0) / a.flag_rows]) != 0;
    {
        typedef T v4t __attribute__((ext_vector_type(4)));
        if (keep) {
            // residual x[row][nb*32 + 8g + 4 h2 ..+3], nb = 4 w + t: 16-B chunk 4 nb + g of the row = plane nb >> 1, slot
            // (4 (nb & 1) + g) ^ swizzle: eight lane-constant slot addresses, the rest immediates
            int xs[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xs[k] = X_OFF + w * 32768 + gf_lds_off(myrow, k) + h2 * 8;
            struct FinOps { v4t x[4]; LnOps ln[2]; };
            auto fin_ops = [&](int t) {
                FinOps fo;
#pragma unroll
                for (int g = 0; g < 4; ++g) fo.x[g] = *reinterpret_cast<const v4t*>(smem + xs[4 * (t & 1) + g] + (t >> 1) * 16384);
                fo.ln[0] = ln_ops(vec + 2 * C, vec + 3 * C, 4 * w + t, 0);
                fo.ln[1] = ln_ops(vec + 2 * C, vec + 3 * C, 4 * w + t, 1);
                return fo;
            };
            FinOps fo[2];
            fo[0] = fin_ops(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < 3) fo[(t + 1) & 1] = fin_ops(t + 1);
                const FinOps& p = fo[t & 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4f y = ln_apply(o[t], g, p.ln[g >> 1].ga[g & 1], p.ln[g >> 1].be[g & 1], nmr2, rstd2);
                    v4t ov;
                    ov[0] = gf_from_float<T>(gf_to_float(p.x[g][0]) + y.x);
                    ov[1] = gf_from_float<T>(gf_to_float(p.x[g][1]) + y.y);
                    ov[2] = gf_from_float<T>(gf_to_float(p.x[g][2]) + y.z);
                    ov[3] = gf_from_float<T>(gf_to_float(p.x[g][3]) + y.w);
                    *reinterpret_cast<v4t*>(smem + xs[4 * (t & 1) + g] + (t >> 1) * 16384) = ov;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        K9P_T(17);
        // (layer skipped: out = x, the tile is the result as it stands)
        // 16-B piece c16 of the wave's half row = piece 16 w + c16 of the row: plane 2 w + (c16 >> 3), slot c16 & 7; lanes 16 r .. 16 r + 15
        // one row.  Buffer stores: rows past the sequence are out of the descriptor's range and dropped (always 8 store instructions:
        // the ring's counted waits behind them stay exact)
        T* og = (T*)a.out + (size_t)n * a.L * a.ldo;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(og, 0, (int)min((long)a.L * a.ldo * 2, 0x7FFFFFFFL), 0x00020000);
        const int c16 = lane & 15, rsel = lane >> 4;
        v4u rows[8];
#pragma unroll
        for (int it = 0; it < 8; ++it)
            rows[it] = *reinterpret_cast<const v4u*>(smem + X_OFF + (2 * w + (c16 >> 3)) * 16384 + gf_lds_off(grp * 32 + 4 * it + rsel, c16 & 7));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int tg = t0 + grp * 32 + 4 * it + rsel;
            __builtin_amdgcn_raw_buffer_store_b128(rows[it], ors, (int)(tg * a.ldo * 2) + (16 * w + c16) * 16, 0, 0);
        }
    }
    K9P_T(12);
    if constexpr (ATTN) {
        if (tail) {
            // the finished rows (storage type, in the x tile's place) are the source tile of the state: masked or out-of-sequence
            // tokens do not count (linear_attention.py:37-39).  The ring runs on: its next blocks are the tail's W_k | W_v.
            xbar();                                                    // the pair's other half of the rows is in place
            const int tl = fresh_lane();                               // the tail's lane-derived values are computed here, not carried through the layer
            const int ttok = t0 + grp * 32 + (tl & 31);
            const bool ok = ttok < a.L && (a.q_mask == nullptr || a.q_mask[(size_t)n * a.L + min(ttok, a.L - 1)] != 0);
            const unsigned valid = (unsigned)__ballot(ok && (tl >> 5) == 0);
            kv_tail<T>(ring, F, smem, valid, a.part + ((size_t)(n - a.tail_first) * a.tiles + tile) * (C * D + C), wave, tl, wave * 64 + tl, NBLK % 3);
        }
    }
}

// The body both enc_kv_state and enc_layer's state tail run: the token tile is in LDS at X_OFF (x-tile layout), the ring's
// current block (slot `slot0`) is the first block of a W_k | W_v stream with its fragments in `fa`; `valid` = bit mask of the
// wave's 32 tokens that count.  Here the products are NOT transposed (A = 32 token rows from the LDS tile, B = 32 weight rows): the
// result has the channel on the lane and the tokens in registers, so the state KV[d][v] = sum_tok phi(k)[tok][d] v[tok][v] - a
// contraction over the tiles' ROW index - takes both accumulators as MFMA operands directly (phi(k) as A gives phi(k)^T . v).  A wave
// computes k and v of its four heads (head = channel tile: k_h and v_h belong to the same wave, no exchange).  Ends with the tile's
// partial state in `dst`; uses ALL of the workgroup's LDS at the end (the tile and the ring are dead by then).
template <typename T>
__device__ __forceinline__ void kv_tail(Ring& ring, typename Mma32<T>::Frag (&F)[8], char* smem, unsigned valid,
                                        float* dst, int wave, int lane, int tid, int slot0) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    const int h2 = lane >> 5, lr = lane & 31, w = wave & 1, grp = wave >> 1;
    const int myrow = grp * 32 + lr;
    auto xfrag = [&](int ks) { return lds_frag<Frag>(smem + X_OFF + (ks >> 2) * 16384 + gf_lds_off(myrow, 2 * (ks & 3) + h2)); };
    int slot = slot0;
    auto project = [&](v16f (&acc)[4], auto between) {                  // acc = src tile x W^T: 8 blocks of (2 k-steps x 4 tiles)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = zero16();
        Frag X[2];
        X[0] = xfrag(0);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            pblock<1, 1, 2>(ring, F, slot, [&](int j) { PMma<T>::mma(X[j >> 2], F[j], acc[j & 3]); }, [&] { X[1] = xfrag(2 * b + 1); },
                            [&] { X[0] = xfrag(b < 7 ? 2 * b + 2 : 15); between(b); });
            slot = (slot + 1) % 3;
        }
    };
    // k first; phi(k) = elu + 1, its sum and the packing of head t as MFMA operands then ride in the v projection's blocks 2 t, 2 t + 1
    Frag kf[4][2];
    float ksum[4];
    float vm[16];                                                       // 1 / 0 per accumulator row (token) of this lane half
#pragma unroll
    for (int r = 0; r < 16; ++r) vm[r] = ((valid >> gf_acc_row(r, h2)) & 1u) ? 1.f : 0.f;
    v16f k[4];
    project(k, [](int) {});
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
    // state of the wave's 32 tokens and four heads: rows = d, lane = v
    v16f kv[4];
    {
        v16f v[4];
        project(v, phi_half);
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            kv[hh] = zero16();
            Mm::mma(kf[hh][0], pack_step<T>(v[hh], 0), kv[hh]);
            Mm::mma(kf[hh][1], pack_step<T>(v[hh], 1), kv[hh]);
        }
    }
    // sum over the four token groups, one partial per tile: every wave parks its partial in LDS (8 x 17 KiB: the tile and the ring
    // are dead by now), all 512 threads add the four copies of a half - in the order (g0 + g2) + (g1 + g3) - and store the partial
    constexpr int KVN = 4 * 16 * 64, SLOT = KVN + 4 * 64;               // floats per wave: kv[hh][r][lane] | ksum[hh][lane]
    float* red = reinterpret_cast<float*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the requests behind the stream's end (out of range: zeros) have landed
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * SLOT + (hh * 16 + r) * 64 + lane] = kv[hh][r];
        red[wave * SLOT + KVN + hh * 64 + lane] = ksum[hh];
    }
    __syncthreads();
#pragma unroll 4
    for (int e2 = tid; e2 < 2 * KVN; e2 += 512) {
        const int wh = e2 >> 12, e = e2 & (KVN - 1);                    // wave half, element of its partial
        const float* r0 = red + wh * SLOT + e;                          // wave 2 g + wh: group g's copy is 2 SLOT further per group
        const float sum = (r0[0] + r0[4 * SLOT]) + (r0[2 * SLOT] + r0[6 * SLOT]);
        const int hh = 4 * wh + (e >> 10), r = (e >> 6) & 15, ln = e & 63;
        dst[(size_t)(hh * D + gf_acc_row(r, ln >> 5)) * D + (ln & 31)] = sum;                               // [c][v]: 128-B runs
    }
    if (tid < 256) {
        const int wh = tid >> 7, e = tid & 127, hh = e >> 5, ln = e & 31;
        const float* r0 = red + wh * SLOT + KVN + hh * 64 + ln;
        dst[C * D + (4 * wh + hh) * D + ln] = (r0[0] + r0[4 * SLOT]) + (r0[2 * SLOT] + r0[6 * SLOT]);
    }
}

// -------------------------------------------------------------------------------------------------------------
// enc_kv_state: k, v projections of a 128-token source tile and its linear-attention state (a pass of its own: the first
// layer's sources and image 0's rows in front of a 'self' layer).  Stream: 8 blocks of W_k, then 8 of W_v.
// -------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(512) void enc_kv_state(EncArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h2 = lane >> 5, lr = lane & 31;
    const int n = blockIdx.x / a.tiles, tile = blockIdx.x - n * a.tiles;
    const int t0 = tile * TM;
    const T* xg = (const T*)a.x + (size_t)n * a.S * a.ldx;
    Ring ring = ring_make(a.wstream, smem, wave, lane, NB_KV, nullptr, 0);
    tile_dma<T>(xg, a.ldx, a.S, t0, TM, smem, X_OFF, wave, lane, 8);
    dma_piece(ring, 0, 0, 0);
    dma_piece(ring, 0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    // validity of the wave's 32 tokens as a bit mask (tail of the image, padding mask of linear_attention.py:37-39)
    const int mytok = t0 + (wave >> 1) * 32 + lr;
    const bool ok = mytok < a.S && (a.kv_mask == nullptr || a.kv_mask[(size_t)n * a.S + min(mytok, a.S - 1)] != 0);
    const unsigned valid = (unsigned)__ballot(ok && h2 == 0);
    dma_piece(ring, 1, 1, 0);
    dma_piece(ring, 1, 1, 1);
    dma_piece(ring, 2, 2, 0);
    dma_piece(ring, 2, 2, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // block 0 and the tile; blocks 1, 2 stay in flight
    __builtin_amdgcn_s_barrier();
    Frag F[8];
    load_block0(ring, F);
    kv_tail<T>(ring, F, smem, valid, a.part + ((size_t)n * a.tiles + tile) * (C * D + C), wave, lane, tid, 0);
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
void pair_launch(const EncArgs& a, int act, bool attn, hipStream_t st) {
    const dim3 grid(a.N * a.tiles);
    if (attn) {
        if (act == 0) enc_layer<T, 0, true><<<grid, 512, LDS_BYTES, st>>>(a);
        else enc_layer<T, 1, true><<<grid, 512, LDS_BYTES, st>>>(a);
    } else {
        if (act == 0) enc_layer<T, 0, false><<<grid, 512, LDS_BYTES, st>>>(a);
        else enc_layer<T, 1, false><<<grid, 512, LDS_BYTES, st>>>(a);
    }
}

std::atomic<uint64_t> pair_attr_done{0};
template <typename K>
void pair_allow_lds(K kern) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); }
void pair_init() {
    if (!gf_first_use_on_device(pair_attr_done)) return;
    pair_allow_lds(enc_layer<_Float16, 0, true>); pair_allow_lds(enc_layer<_Float16, 1, true>);
    pair_allow_lds(enc_layer<_Float16, 0, false>); pair_allow_lds(enc_layer<_Float16, 1, false>);
    pair_allow_lds(enc_kv_state<_Float16>);
    pair_allow_lds(enc_layer<gf_bf16, 0, true>); pair_allow_lds(enc_layer<gf_bf16, 1, true>);
    pair_allow_lds(enc_layer<gf_bf16, 0, false>); pair_allow_lds(enc_layer<gf_bf16, 1, false>);
    pair_allow_lds(enc_kv_state<gf_bf16>);
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
    pair_init();
    EncArgs a{};
    a.x = src; a.ldx = ld; a.N = N; a.L = S; a.S = S; a.tiles = (S + TM - 1) / TM; a.kv_mask = kv_mask; a.wstream = wstream_kv;
    a.part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int len = C * D + C;
    void* pt = gf_prof_begin("enc_kv_state", st, 2.0 * N * (double)S * C * (2.0 * C + 2.0 * D));
    if (dtype == GF_F16) enc_kv_state<_Float16><<<N * a.tiles, 512, LDS_BYTES, st>>>(a);
    else enc_kv_state<gf_bf16><<<N * a.tiles, 512, LDS_BYTES, st>>>(a);
    enc_kv_reduce<<<dim3((len + 255) / 256, N), 256, 0, st>>>(a.part, kv_state, a.tiles, len);
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
    pair_init();
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
    if (dtype == GF_F16) pair_launch<_Float16>(a, activation, true, st);
    else pair_launch<gf_bf16>(a, activation, true, st);
    enc_kv_reduce<<<dim3((len + 255) / 256, nt), 256, 0, st>>>(a.part, kv_state_out, a.tiles, len);
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
    pair_init();
    EncArgs a{};
    a.x = x; a.ldx = ldx; a.msg = msg; a.ldm = ldm; a.kvfinal = kv_state; a.q_mask = q_mask; a.wstream = wstream; a.ln = ln_params;
    a.eps1 = eps1; a.eps2 = eps2; a.attn_eps = attn_eps; a.out = out; a.ldo = ldo; a.N = N; a.L = L; a.S = S > 0 ? S : 1;
    a.tiles = (L + TM - 1) / TM; a.flag = row_flag; a.flag_rows = flag_rows;
    hipStream_t st = (hipStream_t)stream;
    const bool attn = kv_state != nullptr;
    // flops per token: [q 2C^2 + apply 2C(D+1)] + merge 2C^2 + mlp.0 2(2C)(2C) + mlp.2 2(2C)C
    const double per_tok = (attn ? 2.0 * C * C + 2.0 * C * (D + 1) : 0.0) + 2.0 * C * C + 8.0 * C * C + 4.0 * C * C;
    void* pt = gf_prof_begin("enc_layer", st, per_tok * N * (double)L);
    if (dtype == GF_F16) pair_launch<_Float16>(a, activation, attn, st);
    else pair_launch<gf_bf16>(a, activation, attn, st);
    gf_prof_end("enc_layer", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
