// hipcc-flags: -fno-slp-vectorize
// K11: one encoder layer of the FINE-level transformer (loftr_fine: d_model 128, 8 heads of 16, windows of W*W = 25 tokens,
// 16-bit storage) as ONE launch per layer call: a window's tokens never leave the CU between the six GEMMs.
//
// Replaces LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:37-60) with LinearAttention.forward
// (linear_attention.py:21-51) as LocalFeatureTransformer.forward (:82-104) calls it for the fine level (full_model.py:97-98:
// [M, 25, 128] window tensors, no masks), i.e. the K3 x 5 + K2 + torch.cat launch chain of round 2 (per 8-pair step at the
// nominal load: 37 k window-layers through 320 KiB of weights).
//
//   one WAVE owns one window for the whole layer (32 token slots, 25 used), 8 waves = 8 windows per workgroup, two waves per
//   SIMD (256 registers each) so that one wave's LayerNorm / phi / packing VALU work runs beside the other's MFMAs;
//   workgroups are persistent and walk the groups of 8 windows.
//
//   k, v = W_k src, W_v src     NOT transposed (A = token rows from the LDS tile, B = weight fragments): tokens in registers,
//                               channel on the lane, so the state contracts over the accumulators' ROW index:
//   KV  = phi(k)^T v            phi(k) packed as A operand (X^T.B form), v packed as B operand: 32 x 32 tile per channel tile =
//                               the two 16 x 16 head blocks on its diagonal (the off-diagonal blocks are zeroed);
//   Ksum                        the same A operand against a ones operand (every column holds Ksum[d])
//   q = W_q x                   transposed (A = weight fragments, B = token tile): channel in registers, token on the lane
//   msg = KV^T phi(q) / (Ksum^T phi(q) + eps/S)     packed KV / Ksum tiles as A operands, phi(q) as B: lane-local division
//   m = LN1(W_m msg) ; hid = relu(W_1 [x | m]) ; out = x + LN2(W_2 hid)      chained through registers as in K9
//
// Weights are pre-packed on the host (geoformer_amd/fused.py:pack_fine_layer_stream) into the sequence of 1-KiB MFMA fragments
// the kernel consumes: 10 blocks of 32 fragments, TILE-major so that only one 32-channel result tile is live at a time
// (per channel tile nb: W_k[nb], W_v[nb]; then W_q[nb]; then W_m[nb]; then per 32-wide hidden tile hb: W_1[hb, :128],
// W_1[hb, 128:], W_2[:, hb]), streamed L2 -> LDS by LDS-DMA through a two-slot ring that runs on across window groups; every
// fragment is read once by each of the 8 waves.  A step = 4 fragments (four 16-deep k-steps of one tile, or one k-step of the
// four output tiles for W_2) = 4 MFMAs per wave; the staged window itself sits in registers as 8 operand fragments.
//
// Rounding points = those of the K3 / K2 chain it replaces (oracle: encoder_layer_chain + linear_attention_window): q, k, v
// rounded to the storage type, phi(q), phi(k) rounded, KV / S and Ksum / S rounded, message, LN1 output, hidden activations
// and the LN2 output rounded, the residual sum rounded once more; all accumulation and statistics in fp32.
#include <math.h>

#include <type_traits>

#include "gf_common.h"

// -DK11_TRACE=1 records s_memtime at the phase boundaries of the first groups of every workgroup (tools/k11_trace.py); a
// diagnostic build only.
#ifndef K11_TRACE
#define K11_TRACE 0
#endif
#if K11_TRACE
__device__ long long k11_trace[256 * 8 * 4 * 8];                        // [workgroup][wave][group iteration < 4][slot]
#define K11_T(slot) do { if ((ln & 63) == 0 && blockIdx.x < 256 && git < 4) k11_trace[((blockIdx.x * 8 + wave) * 4 + git) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int gf_debug_k11_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(k11_trace), sizeof(long long) * 256 * 8 * 4 * 8);
}
#else
#define K11_T(slot)
#endif

namespace {

constexpr int FC = 128;                           // d_model of the fine level
// -DK11_FW=4 (experiment, round 5): FOUR waves = four windows per workgroup and weight blocks of 16 fragments - 66 KiB of LDS, TWO workgroups
// (two barrier domains) per CU instead of one of eight waves; the stream and every wave's arithmetic are unchanged
#ifndef K11_FW
#define K11_FW 8
#endif
constexpr int FW = K11_FW;                        // waves (= windows) per workgroup
static_assert(FW == 8 || FW == 4, "a weight block is 4 FW fragments: every wave moves four of them");
constexpr int FRAG = 1024, WBLK = 4 * FW * FRAG;  // one weight block = 4 FW fragments of 64 lanes x 16 B = FW steps
constexpr int NBLK = 320 / (4 * FW);              // blocks per layer (320 fragments)
constexpr int TILE = 8192;                        // one window tile: [2 planes][32 rows][128 B], chunk-swizzled (gf_lds_off)
constexpr int X_OFF = 0;
constexpr int W_OFF = FW * TILE;                  // 65,536: two weight blocks
constexpr int VEC_OFF = W_OFF + 2 * WBLK;         // gamma1 | beta1 | gamma2 | beta2, [4][128] float
constexpr int LDS_BYTES = VEC_OFF + 4 * FC * 4;   // 133,120 B

struct FlArgs {
    const void* x;          // [Nw][Lw][128]
    const void* src;        // [Nw][Lw][128] (may equal x)
    void* out;              // [Nw][Lw][128]
    const void* wstream;    // 320 KiB packed fragments
    const float* ln;        // gamma1 | beta1 | gamma2 | beta2
    float eps1, eps2, attn_eps;
    int Nw, Lw, groups;
};

template <typename T>
__device__ __forceinline__ typename Mma32<T>::Frag fl_pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    typename Mma32<T>::Frag f;
    f[0] = gf_from_float<T>(a0); f[1] = gf_from_float<T>(a1); f[2] = gf_from_float<T>(a2); f[3] = gf_from_float<T>(a3);
    f[4] = gf_from_float<T>(a4); f[5] = gf_from_float<T>(a5); f[6] = gf_from_float<T>(a6); f[7] = gf_from_float<T>(a7);
    return f;
}
// registers 8s .. 8s+7 of an accumulator tile as the operand of k-step s of the next product
template <typename T>
__device__ __forceinline__ typename Mma32<T>::Frag fl_pack_step(const v16f& a, int s) {
    return s == 0 ? fl_pack8<T>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7])
                  : fl_pack8<T>(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
}
// elu(x) + 1 = max(x, 0) + exp(min(x, 0)), branch-free, hardware exponential
// (round 5: exp(min(x, 0)) = the exponential CLAMPED to [0, 1] - `v_exp_f32 ... clamp`, the output modifier is free - instead of a
// v_min in front of it: the same bits (for x <= 0 the clamp does nothing, for x > 0 both give exactly 1), one instruction less per value)
__device__ __forceinline__ float fl_phi(float x) {
    return fmaxf(x, 0.f) + __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * 1.44269504088896341f), 0.f, 1.f);
}
template <typename T>
__device__ __forceinline__ float fl_rnd(float x) { return gf_to_float(gf_from_float<T>(x)); }

// a zero accumulator as a CONSTANT operand: the first MFMA of a chain takes C = 0 inline instead of 16 v_mov per tile
// (44 tiles per window: 2.8 k issue cycles)
__device__ __forceinline__ v16f fl_zero() { return v16f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }

// the weight ring: block g (counted over the whole kernel) sits in slot g & 1 and holds stream block g % NBLK
struct FlRsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ FlRsrc fl_rsrc(const void* p, unsigned bytes) {
    return FlRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
// LDS-DMA as buffer_load_dwordx4 ... lds (MUBUF): behind the FLAT form (global_load_lds) the compiler turns every LDS counter
// wait into lgkmcnt(0) while a request is pending - which is always, here (see k9_encoder_fused.hip); scalar descriptor, 32-bit offsets
__device__ __forceinline__ void fl_lds_dma(const FlRsrc& rs, char* dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}
struct FlRing {
    FlRsrc ws;
    char* smem;
    int wave, lane;
    int blk;        // block being consumed (global count)
    int sblk;       // its index in the layer's stream (0 .. NBLK-1)
    int total;      // blocks this workgroup consumes in all
};
// stream block `sb` into ring slot `g & 1`: wave w moves fragments 4w .. 4w+3
__device__ __forceinline__ void fl_dma_block(const FlRing& r, int g, int sb) {
    char* dst = r.smem + W_OFF + (g & 1) * WBLK + r.wave * 4 * FRAG;
#pragma unroll
    for (int i = 0; i < 4; ++i) fl_lds_dma(r.ws, dst + i * FRAG, r.lane * 16, sb * WBLK + (r.wave * 4 + i) * FRAG);
    __builtin_amdgcn_sched_barrier(0);
}
template <typename Frag>
__device__ __forceinline__ void fl_load_step(const FlRing& r, Frag (&f)[4], int g, int st) {
    const char* p = r.smem + W_OFF + (g & 1) * WBLK + st * 4 * FRAG + r.lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const Frag*>(p + i * FRAG);
}
// before the MFMAs of step st (0..7) of the current block: request the fragments of the following step.  The ring turns in
// front of the LAST step of a block: by then this wave holds the block's last fragments in registers; the wait + barrier
// say "my part of block g+1 has landed" and "everybody is done reading block g", so block g+2 can be requested into g's slot.
// Invariant of the wait: every vector-memory operation of this wave older than the turn has completed (vmcnt(0)): the DMA
// of block g+1 (requested one block ago), tile loads and output stores of earlier phases.
template <typename Frag>
__device__ __forceinline__ void fl_fetch_next(FlRing& r, Frag (&nx)[4], int st) {
    if (st == FW - 1) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (r.blk + 2 < r.total) {
            int sb = r.sblk + 2;
            sb = sb >= NBLK ? sb - NBLK : sb;
            fl_dma_block(r, r.blk + 2, sb);
        }
        if (r.blk + 1 < r.total) fl_load_step(r, nx, r.blk + 1, 0);
        ++r.blk;
        r.sblk = r.sblk + 1 == NBLK ? 0 : r.sblk + 1;
    } else {
        fl_load_step(r, nx, r.blk, st + 1);
    }
}

// window tile <- global rows by LDS-DMA: instruction i covers plane i >> 2, rows 8 (i & 3) .. + 7; the chunk swizzle of
// gf_lds_off sits on the SOURCE side (the LDS image of a DMA is lane-linear).  Rows >= Lw re-read row Lw - 1 (finite values;
// they only feed token slots that are masked or never stored).
__device__ __forceinline__ void fl_tile_dma(const FlRsrc& wins, int win_byte_offset, int Lw, char* tile, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 8 * (i & 3) + (lane >> 3), slot = lane & 7, c = slot ^ ((row >> 1) & 7);
        const int r = row < Lw ? row : Lw - 1;
        fl_lds_dma(wins, tile + i * FRAG, r * (FC * 2) + (i >> 2) * 128 + c * 16, win_byte_offset);
    }
}

template <typename T>
__global__ __launch_bounds__(64 * FW, 2) void fine_layer(FlArgs a) {
    using Mm = Mma32<T>;
    using Frag = typename Mm::Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* vec = reinterpret_cast<float*>(smem + VEC_OFF);
    char* tile = smem + X_OFF + wave * TILE;
#ifdef K11_PRIO       // -DK11_PRIO (experiment, round 5): the second-dispatched half of the waves at static priority 1, as in K10
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int my_groups = (a.groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    FlRing ring{fl_rsrc(a.wstream, (unsigned)NBLK * WBLK), smem, wave, lane, 0, 0, my_groups * NBLK};
    const FlRsrc xrs = fl_rsrc(a.x, (unsigned)a.Nw * a.Lw * (FC * 2)), srs = fl_rsrc(a.src, (unsigned)a.Nw * a.Lw * (FC * 2));
    fl_dma_block(ring, 0, 0);
    fl_dma_block(ring, 1, 1);
    for (int i = tid; i < 4 * FC; i += 64 * FW) vec[i] = a.ln[i];       // 4 x 128 floats
    const bool same_src = a.src == a.x;
    const float inv_s = 1.0f / (float)a.Lw, eps_s = a.attn_eps * inv_s;
    // token-slot validity of accumulator row (r&3) + 8 (r>>2) + 4 h2 in the non-transposed products: row < Lw  <=>  its
    // lane-independent part < lim
    Frag ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = gf_from_float<T>(1.0f);
    Frag fa[4], fb[4];
    bool first = true;

    // One STEP of the weight stream = 4 fragments = 4 MFMAs of this wave; `gs` is the step's index inside the layer (every loop
    // below is unrolled, so it is a constant): its parity picks the fragment registers, gs % 8 == 7 is the ring turn.
#define FL_STEP_BEGIN(gs)                                               \
    Frag (&cur)[4] = ((gs) & 1) ? fb : fa;                               \
    Frag (&nxt)[4] = ((gs) & 1) ? fa : fb;                               \
    fl_fetch_next(ring, nxt, (gs) & (FW - 1))

    int git = -1;
    (void)git;
    for (int g = blockIdx.x; g < a.groups; g += gridDim.x) {
        ++git;
        // per-lane offsets are RECOMPUTED every group from an opaque copy of the lane id: hoisted out of the loop they stay live
        // through the whole layer and spill (77 registers in the first build)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int h2 = ln >> 5, lr = ln & 31;
        // token slots >= Lw of the non-transposed products (row = token) are zeroed on the PACKED operands: element pair p of k-step s
        // holds rows 16 s + 8 (p >> 1) + 4 h2 + 2 (p & 1) and the next one.  (A select per element in front of phi compiles into a
        // divergent branch per element.)
        v4u pm0, pm1;
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
            const int ra = 8 * (p4 >> 1) + 4 * h2 + 2 * (p4 & 1);
            pm0[p4] = (ra < a.Lw ? 0xFFFFu : 0u) | (ra + 1 < a.Lw ? 0xFFFF0000u : 0u);
            pm1[p4] = (ra + 16 < a.Lw ? 0xFFFFu : 0u) | (ra + 17 < a.Lw ? 0xFFFF0000u : 0u);
        }
        auto tok_mask = [](const Frag& f, const v4u& m) { return __builtin_bit_cast(Frag, __builtin_bit_cast(v4u, f) & m); };
        const int win = g * FW + wave;
        K11_T(0);
        const int wclamp = win < a.Nw ? win : a.Nw - 1;                 // a tail group's spare waves recompute the last window (never stored)
        const int win_off = (int)((unsigned)wclamp * (unsigned)(a.Lw * (FC * 2)));   // byte offset of the window in x / src: unsigned 32-bit (the entry point bounds a launch's tensors to < 2^31 BYTES)
        // the wave's own earlier reads of its tile (the previous group's output rows) are in registers by now
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fl_tile_dma(srs, win_off, a.Lw, tile, ln);
        if (first) {
            // blocks 0 and 1 are requested; block 0 must have landed everywhere before its first fragment is read
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            fl_load_step(ring, fa, 0, 0);
            first = false;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tile (wave-private: no barrier needed)
        }
        // the staged window as MFMA operand fragments (A of the k / v products, B of the q and mlp.0 products): k-step ks =
        // channels 16 ks + 8 h2 .. + 7 of token row lr
        K11_T(1);
        Frag xf[8];
        auto load_xf = [&]() {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int ch = 2 * ks + h2;
                xf[ks] = *reinterpret_cast<const Frag*>(tile + (ch >> 3) * 4096 + gf_lds_off(lr, ch & 7));
            }
        };
        load_xf();
        if (!same_src) {
            // the source window sits in registers now: the query window takes its place in the tile (the whole k / v phase to land)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fl_tile_dma(xrs, win_off, a.Lw, tile, ln);
        }

        // ---------------- per channel tile nb (= two heads): k, v = src W^T (tokens in registers, channel on the lane), state
        Frag kvA[4][2], ksA[4][2];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            v16f k = fl_zero(), v = fl_zero();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                {
                    FL_STEP_BEGIN(4 * nb + hf);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mm::mma(xf[4 * hf + i], cur[i], k);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) k[r] = fl_phi(fl_rnd<T>(k[r]));
            const Frag k0 = tok_mask(fl_pack_step<T>(k, 0), pm0), k1 = tok_mask(fl_pack_step<T>(k, 1), pm1);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                {
                    FL_STEP_BEGIN(4 * nb + 2 + hf);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mm::mma(xf[4 * hf + i], cur[i], v);
                }
            const Frag v0 = tok_mask(fl_pack_step<T>(v, 0), pm0), v1 = tok_mask(fl_pack_step<T>(v, 1), pm1);
            // KV tile = phi(k)^T v, Ksum tile = phi(k)^T 1 (every column holds Ksum[d]).  The tile's off-head blocks (row d and
            // column v in different heads) are NOT zeroed: the apply below contracts head by head (one 16-deep k-step = the
            // 16 d of one head) and reads only the result rows of that head.
            v16f kv = fl_zero(), ks = fl_zero();
            Mm::mma(k0, v0, kv);
            Mm::mma(k1, v1, kv);
            Mm::mma(k0, ones, ks);
            Mm::mma(k1, ones, ks);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                kv[r] *= inv_s;
                ks[r] *= inv_s;
            }
            kvA[nb][0] = fl_pack_step<T>(kv, 0); kvA[nb][1] = fl_pack_step<T>(kv, 1);
            ksA[nb][0] = fl_pack_step<T>(ks, 0); ksA[nb][1] = fl_pack_step<T>(ks, 1);
        }
        if (!same_src) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the query window has landed (wave-private tile)
            load_xf();
        }

        K11_T(2);
        // ---------------- per channel tile: q = W_q x (channel in registers, token on the lane), attention of its two heads
        Frag mB[4][2];                                                  // message, then LN1 output: B operand of the next product
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            v16f q = fl_zero();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                {
                    FL_STEP_BEGIN(16 + 2 * nb + hf);
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mm::mma(cur[i], xf[4 * hf + i], q);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] = fl_phi(fl_rnd<T>(q[r]));
            const Frag q0 = fl_pack_step<T>(q, 0), q1 = fl_pack_step<T>(q, 1);
            // head by head: k-step s holds the 16 d of head s of this tile, and only result rows v of the same head (registers
            // 8 s .. 8 s + 7) are meaningful; every row of a den tile holds the token's normaliser of that head
            v16f num0 = fl_zero(), num1 = fl_zero(), den0 = fl_zero(), den1 = fl_zero();
            Mm::mma(kvA[nb][0], q0, num0);
            Mm::mma(kvA[nb][1], q1, num1);
            Mm::mma(ksA[nb][0], q0, den0);
            Mm::mma(ksA[nb][1], q1, den1);
            const float z0 = __builtin_amdgcn_rcpf(den0[0] + eps_s), z1 = __builtin_amdgcn_rcpf(den1[8] + eps_s);
            v16f num;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                num[r] = num0[r] * z0;
                num[8 + r] = num1[8 + r] * z1;
            }
            mB[nb][0] = fl_pack_step<T>(num, 0);
            mB[nb][1] = fl_pack_step<T>(num, 1);
        }

        K11_T(3);
        // LayerNorm over the 128 channels of the lane's token: this lane half holds 64 of them (4 tiles x 16 registers)
        auto ln_stats = [&](const v16f (&t)[4], float eps, float& mean, float& rstd) {
            float s = 0.f, qd = 0.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = t[nb][r];
                    s += v;
                    qd = fmaf(v, v, qd);
                }
            // x[lane] + x[lane ^ 32] by one v_permlane32_swap each (VALU; the ds_bpermute form waits on the LDS counter)
            const gf_v2u ss = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
            const gf_v2u sq = __builtin_amdgcn_permlane32_swap(__float_as_uint(qd), __float_as_uint(qd), false, false);
            s = __uint_as_float(ss.x) + __uint_as_float(ss.y);
            qd = __uint_as_float(sq.x) + __uint_as_float(sq.y);
            mean = s * (1.0f / FC);
            rstd = 1.0f / sqrtf(fmaxf(qd - s * mean, 0.f) * (1.0f / FC) + eps);
        };
        auto ln_apply = [&](const v16f& t, int nb, int g4, const float* gamma, const float* beta, float mean, float rstd) {
            const int c = nb * 32 + 8 * g4 + 4 * h2;
            const v4f ga = *reinterpret_cast<const v4f*>(gamma + c), be = *reinterpret_cast<const v4f*>(beta + c);
            const float nmr = -mean * rstd;                  // (t - mean) rstd gamma + beta as two FMAs per value
            return v4f{fmaf(fmaf(t[4 * g4], rstd, nmr), ga.x, be.x), fmaf(fmaf(t[4 * g4 + 1], rstd, nmr), ga.y, be.y),
                       fmaf(fmaf(t[4 * g4 + 2], rstd, nmr), ga.z, be.z), fmaf(fmaf(t[4 * g4 + 3], rstd, nmr), ga.w, be.w)};
        };

        // ---------------- m = LN1(W_m msg)
        {
            v16f m[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                m[nb] = fl_zero();
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    {
                    FL_STEP_BEGIN(24 + 2 * nb + hf);
#pragma unroll
                        for (int i = 0; i < 4; ++i) Mm::mma(cur[i], mB[2 * hf + (i >> 1)][i & 1], m[nb]);
                    }
            }
            float mean, rstd;
            ln_stats(m, a.eps1, mean, rstd);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    const v4f lo = ln_apply(m[nb], nb, 2 * sx, vec, vec + FC, mean, rstd), hi = ln_apply(m[nb], nb, 2 * sx + 1, vec, vec + FC, mean, rstd);
                    mB[nb][sx] = fl_pack8<T>(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
                }
        }

        K11_T(4);
        // ---------------- per 32-wide hidden tile hb: hid = relu(W_1[hb] [x | m]) (4 steps), consumed at once by out += W_2[:, hb] hid
        // (2 steps); 6 steps per tile, so 4 tiles = 24 steps = 3 whole blocks per iteration of the outer loop
        v16f o[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) o[nb] = fl_zero();
#pragma unroll 1
        for (int hq = 0; hq < 2; ++hq) {
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                v16f hd = fl_zero();
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    {
                    FL_STEP_BEGIN(6 * hb + hf);
#pragma unroll
                        for (int i = 0; i < 4; ++i) Mm::mma(cur[i], xf[4 * hf + i], hd);
                    }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    {
                    FL_STEP_BEGIN(6 * hb + 2 + hf);
#pragma unroll
                        for (int i = 0; i < 4; ++i) Mm::mma(cur[i], mB[2 * hf + (i >> 1)][i & 1], hd);
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) hd[r] = fmaxf(hd[r], 0.f);
                const Frag h0 = fl_pack_step<T>(hd, 0), h1 = fl_pack_step<T>(hd, 1);
                {
                    FL_STEP_BEGIN(6 * hb + 4);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) Mm::mma(cur[nb], h0, o[nb]);
                }
                {
                    FL_STEP_BEGIN(6 * hb + 5);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) Mm::mma(cur[nb], h1, o[nb]);
                }
            }
        }

        K11_T(5);
        // ---------------- out = x + LN2(.), written into the wave's own tile (token rows), then stored in 16-byte row segments
        {
            float mean2, rstd2;
            ln_stats(o, a.eps2, mean2, rstd2);
            typedef T v4t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c = nb * 32 + 8 * g4 + 4 * h2, ch = c >> 3;
                    char* p = tile + (ch >> 3) * 4096 + gf_lds_off(lr, ch & 7) + (c & 7) * 2;
                    const v4t xv = *reinterpret_cast<const v4t*>(p);
                    const v4f y = ln_apply(o[nb], nb, g4, vec + 2 * FC, vec + 3 * FC, mean2, rstd2);
                    v4t ov;
                    ov[0] = gf_from_float<T>(gf_to_float(xv[0]) + fl_rnd<T>(y.x));
                    ov[1] = gf_from_float<T>(gf_to_float(xv[1]) + fl_rnd<T>(y.y));
                    ov[2] = gf_from_float<T>(gf_to_float(xv[2]) + fl_rnd<T>(y.z));
                    ov[3] = gf_from_float<T>(gf_to_float(xv[3]) + fl_rnd<T>(y.w));
                    *reinterpret_cast<v4t*>(p) = ov;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (win < a.Nw) {
                char* og = (char*)a.out + (size_t)win * a.Lw * (FC * 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = ln + 64 * i, row = e >> 4, ch = e & 15;         // 16 chunks of 16 B per 256-B row
                    if (row < a.Lw)
                        *reinterpret_cast<v4u*>(og + row * (FC * 2) + ch * 16) =
                            *reinterpret_cast<const v4u*>(tile + (ch >> 3) * 4096 + gf_lds_off(row, ch & 7));
                }
            }
        }
        K11_T(6);
    }
#undef FL_STEP_BEGIN
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}   // namespace

// one fine-level encoder layer over Nw windows of Lw (<= 32) tokens x 128 channels; see include/geoformer_hip.h
extern "C" int gf_fine_layer(const void* x, const void* src, void* out, int dtype, int Nw, int Lw, const void* wstream,
                             const float* ln_params, float eps1, float eps2, float attn_eps, void* stream) {
    GF_CHECK_ARG(x && src && out && wstream && ln_params, "null pointer");
    GF_CHECK_ARG(Nw > 0, "empty problem");
    GF_CHECK_ARG(Lw > 0 && Lw <= 32, "windows of 1 .. 32 tokens");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "the fused fine-level layer is built for 16-bit storage (GF_F16 / GF_BF16)");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)wstream % 16 == 0,
                 "tensors must be 16-byte aligned");
    static std::atomic<uint64_t> attr{0};
    if (gf_first_use_on_device(attr)) {
        (void)hipFuncSetAttribute((const void*)fine_layer<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)fine_layer<gf_bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    hipStream_t st = (hipStream_t)stream;
    // flops per window token: q, k, v, merge 4 x 2 C^2 + state and apply 4 x 2 C D (D = 16) + mlp 2 (2C)(2C) + 2 (2C) C
    const double per_tok = 8.0 * FC * FC + 8.0 * FC * 16 + 8.0 * FC * FC + 4.0 * FC * FC;
    void* pt = gf_prof_begin("fine_layer", st, per_tok * (double)Nw * Lw);
    // the kernel addresses a window through a 32-bit byte offset into a buffer descriptor: a launch takes at most
    // GF_FINE_LAYER_MAX_BYTES of windows (2^31 - 64 KiB; 335 k windows of 25 tokens), more windows = more launches
    const long win_bytes = (long)Lw * FC * 2;
    const int per_launch = (int)(((1l << 31) - 65536) / win_bytes);
    for (long w0 = 0; w0 < Nw; w0 += per_launch) {
        const int nw = (int)((Nw - w0) < per_launch ? (Nw - w0) : per_launch);
        FlArgs a{(const char*)x + w0 * win_bytes, (const char*)src + w0 * win_bytes, (char*)out + w0 * win_bytes, wstream, ln_params,
                 eps1, eps2, attn_eps, nw, Lw, (nw + FW - 1) / FW};
        const int slots = 256 * (8 / FW), grid = a.groups < slots ? a.groups : slots;
        if (dtype == GF_F16) fine_layer<_Float16><<<grid, 64 * FW, LDS_BYTES, st>>>(a);
        else fine_layer<gf_bf16><<<grid, 64 * FW, LDS_BYTES, st>>>(a);
    }
    gf_prof_end("fine_layer", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
