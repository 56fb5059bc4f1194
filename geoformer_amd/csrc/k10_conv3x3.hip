// K10: 3x3 / stride 1 / pad 1 convolution of channels-last 16-bit maps with the BatchNorm shift, the BasicBlock
// shortcut and the activation in its epilogue - the backbone's dominant shapes (SURVEY 8f rank 4:
// model/loftr_src/loftr/backbone/resnet_fpn.py:9-40 BasicBlock, :60-83 the FPN heads), as an implicit GEMM on the
// matrix cores:   out[n,y,x,:] = act( sum_{ky,kx} W[:, :, ky, kx] . x[n, y+ky-1, x+kx-1, :] + shift + shortcut[n,y,x,:] )
//
// Same machinery as K9 (k9_encoder_fused.hip): products transposed (MFMA A = output channels, B = pixels, so a pixel's
// channels sit in one lane's registers; v_mfma_f32_16x16x32: 16 channels x 16 pixels x the 32 input channels of a chunk),
// weights pre-packed on the host into the exact sequence of 1-KiB MFMA A fragments the kernel consumes and streamed from L2
// through a two-block LDS ring by LDS-DMA; the fragments of the next sub-step are requested in front of this one's MFMAs.
//   workgroup = 8 waves (two per SIMD, 256 registers each: one wave's DMA requests, ring turns and epilogue fill the
//               other's MFMA gaps) = a 16-row x 32-column pixel tile of one image, a wave owns two rows = four 16-pixel
//               blocks, 4 x COUT/16 accumulator tiles of four registers (outputs wider than 128 channels: 8-row tiles, one
//               row per wave);
//   K loop     = input channels in chunks of 32 x 9 taps x 2 halves of the output channels: one SUB-STEP = COUT/32 weight
//               fragments (16 channels x 32 input channels each) times the tap's 2 PB pixel fragments (16 pixels x 32
//               channels = the 64 bytes of a patch pixel); a weight block = 6 sub-steps, so the ring turns at fixed
//               places of a chunk;
//   pixels     = per chunk the halo patch of the tile (64 B per pixel) sits in LDS, double-buffered: the next chunk's
//               (or the next tile's first) patch arrives by LDS-DMA while the current one is multiplied; pixels outside
//               the image are read from a page of zeros (per-lane DMA source address); the 16-byte slot index is XORed
//               with (pixel >> 1) & 3 on the source side: the ds_read_b128 fragment reads of 16 consecutive pixels x 4
//               k groups are then conflict-free for every patch offset;
//   persistent = min(tiles, 256) workgroups walk the tiles (13 rounds at 16 x 320 x 320); the ring and the patch
//               buffers run on across tiles (the next tile's first blocks and patch are requested during the last
//               chunk), and the epilogue is wave-private: no workgroup barrier besides the ring turns;
//   epilogue   = from registers (wave-private): the packing deals the output channels to the fragment rows so that two accumulator
//               tiles of a lane are 8 consecutive channels; accumulators (started at the BatchNorm shift) -> storage type -> neighbouring
//               pixels exchange pieces by DPP so that an instruction writes 8 pixels x 128 B -> + shortcut (every piece requested
//               before the first store) -> activation -> 16-byte buffer stores that nobody waits for (pixels outside the image: an
//               out-of-range offset); one straight-line body per (shortcut?, activation form).
#include <type_traits>

#include "gf_common.h"

namespace {

constexpr int TW = 32, PW = TW + 2;               // tile / halo-patch width
constexpr int C10_FRAG = 1024;
constexpr int P_OFF = 0;                          // two patches
enum { C10_NONE = 0, C10_RELU = 1, C10_LEAKY = 2 };

struct ConvArgs {
    const void* x;          // [N][H][W][CIN]
    const void* wstream;    // packed fragments (fused.py:pack_conv3x3_stream)
    const float* shift;     // [COUT] or null
    const void* res;        // [N][H][W][COUT] or null
    void* out;              // [N][H][W][COUT]
    const void* zeros;      // >= 64 bytes of zeros (source of out-of-image pixels)
    int N, H, W, act;       // H x W: the OUTPUT map (= the input map at stride 1)
    float slope;
    int tiles_x, tiles_y, ntiles;
    int Hi, Wi;             // the input map (stride 2: H = (Hi - 1) / 2 + 1)
};

template <int NT, int NW, bool REM = false, bool S2 = false>
struct ConvGeo {
    // output widths up to 128 channels: a wave owns two 32-pixel rows (every weight fragment feeds two MFMAs, 2 x NT
    // accumulator tiles); wider outputs: one row per wave (the accumulators of two would not fit the register file).
    // NW = 4 waves (one per SIMD, 512 registers each) or 8 (two per SIMD, 256 each: one wave's issue gaps - DMA requests,
    // ring turns, the epilogue - are filled by the other's MFMAs).
    static constexpr int PB = NT <= 4 ? 2 : 1;
    // S2 (stride 2, wide outputs only): the halo patch of a chunk is held as its four PARITY PLANES (rows / columns of equal parity
    // relative to the patch origin: a tap reads ONE plane, at stride 1 - plane pixel (yl + ky / 2, xl + kx / 2) of plane (ky & 1, kx & 1));
    // a plane = (TH + 1) x PW pixels; ONE set of planes (80 KB: no room for two), refilled plane by plane as the taps leave them
    static constexpr int TH = NW * PB, PH = S2 ? TH + 1 : TH + 2;  // tile / halo-patch (plane) height
    static constexpr int PIECES = (PH * PW + 15) / 16;            // DMA pieces of 16 pixels x 64 B (per patch; S2: per plane)
    static constexpr int PLANE_BYTES = PIECES * 1024;
    static constexpr int PATCH_BYTES = S2 ? 4 * PLANE_BYTES : PIECES * 1024;
    static constexpr int W10_OFF = S2 ? PATCH_BYTES : 2 * PATCH_BYTES;   // two weight blocks behind the two patches (S2: the one set of planes)
    static_assert(!S2 || PB == 1, "the stride-2 form is built for the wide outputs (one row per wave)");
    // a weight block = 6 sub-steps (a third of a chunk's 18); K10_BS=3 builds the short blocks of rounds 2-3 for the wide outputs
#ifndef K10_BS
#define K10_BS 6
#endif
    static constexpr int BLOCK_STEPS = S2 ? 3 : (NW == 8 && PB == 1) ? K10_BS : 6;   // (S2: 80 KB of planes leave room for short blocks only)
    static constexpr int FR = BLOCK_STEPS * NT;                   // fragments per block
    static constexpr int WBLK = FR * C10_FRAG;
    static constexpr int SHIFT_OFF = W10_OFF + 2 * WBLK;
    // REM (GF_CONV_REM8): the patch of the 8-channel remainder chunk (16 B per pixel) has a buffer of its own, in pieces of 64 pixels
    static constexpr int RPIECES = (PH * PW + 63) / 64;
    static constexpr int R_OFF = SHIFT_OFF + NT * 32 * 4 + 16;   // (behind the shifts: the LeakyReLU slope)
    static constexpr int LDS = R_OFF + (REM ? RPIECES * 1024 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

// LDS-DMA as buffer_load_dwordx4 ... lds (MUBUF) rather than global_load_lds: behind a FLAT-encoded LDS-DMA the compiler's wait
// insertion turns every LDS counter wait into lgkmcnt(0) while the request is pending (it may touch both address spaces) - i.e.
// always, here; behind the MUBUF form it counts.  Scalar descriptor + 32-bit offsets: no 64-bit address arithmetic per piece,
// and a lane whose offset is outside the buffer's range gets ZEROS in LDS (tools/probes/lds_dma_oob.hip): out-of-image halo
// pixels need no page of zeros and no pointer select.
struct ConvRsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ ConvRsrc conv_rsrc(const void* p, unsigned bytes) {
    return ConvRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ void conv_lds_dma(const ConvRsrc& rs, char* dst, int voffset, int soffset) {      // 64 lanes x 16 B -> 1 KiB at dst
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}
// request weight block b of the stream into ring slot `slot`; the waves share its fragments round-robin
template <int NT, int NW, bool S2 = false>
__device__ __forceinline__ void conv_dma_block(const ConvRsrc& ws, char* smem, int b, int slot, int wave, int lane, int i0 = 0, int i1 = 1 << 20) {
    using G = ConvGeo<NT, NW, false, S2>;
    asm volatile("" : "+v"(lane));      // (as in conv_dma_patch)
    char* dst = smem + G::W10_OFF + slot * G::WBLK;
    // the wave's pieces i0 .. i1 - 1 (the requests of a block are dealt to sub-steps; constant loop bounds: the range folds once the
    // sub-step loop is unrolled)
#pragma unroll
    for (int i = 0; i < (G::FR + NW - 1) / NW; ++i) {
        const int f = wave + NW * i;
        // (f < FR holds for every wave where NW (i + 1) <= FR: no scalar branch for those pieces)
        if (i >= i0 && i < i1 && (NW * (i + 1) <= G::FR || f < G::FR)) conv_lds_dma(ws, dst + f * C10_FRAG, lane * 16, b * G::WBLK + f * C10_FRAG);
    }
}

// request the halo patch of (tile, channel chunk c) into patch buffer `buf`: pieces of 16 pixels x 64 B over the waves
template <typename T, int CIN, int NT, int NW>
__device__ __forceinline__ void conv_dma_patch(const ConvArgs& a, const ConvRsrc& xs, char* smem, int buf, int n, int y0, int x0, int c, int wave,
                                               int lane, int i0 = 0, int i1 = 1 << 20) {
    using G = ConvGeo<NT, NW>;
    asm volatile("" : "+v"(lane));      // recompute the per-lane source offsets here: hoisted out of the K loop they cost
                                        // live registers per piece, and a spilled one a vmcnt(0) reload between requests
    char* dst = smem + P_OFF + buf * G::PATCH_BYTES;
    // element offset of patch pixel (0, 0), channel 32 c (may be negative; 32-bit: the entry point bounds the tensor)
    const int sbase = ((n * a.Hi + y0 - 1) * a.Wi + x0 - 1) * CIN + 32 * c;
#pragma unroll
    for (int i = 0; i < (G::PIECES + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i;
        if (i >= i0 && i < i1 && (NW * (i + 1) <= G::PIECES || piece < G::PIECES)) {
            const int q = piece * 16 + (lane >> 2), slot = (lane & 3) ^ ((q >> 1) & 3);
            const int pr = q / PW, pc = q - pr * PW;
            const bool in = (unsigned)(y0 - 1 + pr) < (unsigned)a.Hi && (unsigned)(x0 - 1 + pc) < (unsigned)a.Wi && q < G::PH * PW;
            int off = (sbase + (pr * a.Wi + pc) * CIN + 8 * slot) * (int)sizeof(T);
            asm volatile("" : "+v"(off));       // (computed for every lane: left to the compiler the select becomes an exec-masked branch per piece)
            conv_lds_dma(xs, dst + piece * 1024, in ? off : 0x7FFFFFF0, 0);                // outside the image: out of range -> zeros
        }
    }
}

// stride 2: request parity plane `plane` = (row parity, column parity) of the halo patch of (tile at OUTPUT pixel (y0, x0), channel chunk c):
// plane pixel (pr, pc) = input pixel (2 (y0 + pr) - 1 + plane / 2, 2 (x0 + pc) - 1 + plane % 2); pieces of 16 plane pixels x 64 B
template <typename T, int CIN, int NT, int NW>
__device__ __forceinline__ void conv_dma_plane(const ConvArgs& a, const ConvRsrc& xs, char* smem, int plane, int n, int y0, int x0, int c, int wave,
                                               int lane) {
    using G = ConvGeo<NT, NW, false, true>;
    asm volatile("" : "+v"(lane));
    char* dst = smem + P_OFF + plane * G::PLANE_BYTES;
    const int py = plane >> 1, px = plane & 1;
    const int iy0 = 2 * y0 - 1 + py, ix0 = 2 * x0 - 1 + px;
    const int sbase = ((n * a.Hi + iy0) * a.Wi + ix0) * CIN + 32 * c;
#pragma unroll
    for (int i = 0; i < (G::PIECES + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i;
        if (piece < G::PIECES) {
            const int q = piece * 16 + (lane >> 2), slot = (lane & 3) ^ ((q >> 1) & 3);
            const int pr = q / PW, pc = q - pr * PW;
            const bool in = (unsigned)(iy0 + 2 * pr) < (unsigned)a.Hi && (unsigned)(ix0 + 2 * pc) < (unsigned)a.Wi && q < G::PH * PW;
            int off = (sbase + 2 * (pr * a.Wi + pc) * CIN + 8 * slot) * (int)sizeof(T);
            asm volatile("" : "+v"(off));
            conv_lds_dma(xs, dst + piece * 1024, in ? off : 0x7FFFFFF0, 0);
        }
    }
}

// request the REMAINDER patch of a tile (channels 192 .. 199 of every halo pixel: 16 B per pixel, pixel q at q * 16): pieces of 64 pixels
template <typename T, int CIN, int NT, int NW>
__device__ __forceinline__ void conv_dma_patch_rem(const ConvArgs& a, const ConvRsrc& xs, char* smem, int n, int y0, int x0, int wave, int lane) {
    using G = ConvGeo<NT, NW, true>;
    asm volatile("" : "+v"(lane));
    char* dst = smem + G::R_OFF;
    const int sbase = ((n * a.Hi + y0 - 1) * a.Wi + x0 - 1) * CIN + 192;
#pragma unroll
    for (int i = 0; i < (G::RPIECES + NW - 1) / NW; ++i) {
        const int piece = wave + NW * i;
        if (piece < G::RPIECES) {
            const int q = piece * 64 + lane;
            const int pr = q / PW, pc = q - pr * PW;
            const bool in = (unsigned)(y0 - 1 + pr) < (unsigned)a.Hi && (unsigned)(x0 - 1 + pc) < (unsigned)a.Wi && q < G::PH * PW;
            const int off = sbase + (pr * a.Wi + pc) * CIN;
            conv_lds_dma(xs, dst + piece * 1024, in ? off * (int)sizeof(T) : 0x7FFFFFF0, 0);
        }
    }
}

// -DK10_TRACE=1 records s_memtime at the phase boundaries of the first 8 tiles of every wave (tools/k10_trace.py)
#ifndef K10_TRACE
#define K10_TRACE 0
#endif
#ifdef K10_EXP_NOSTORE
#define K10_STORE_OK (a.slope == 12345.f)
#else
#define K10_STORE_OK true
#endif
#if K10_TRACE
__device__ long long k10_trace[256 * 8 * 8 * 16];
#define K10_T(slot) do { if (lane == 0 && it < 8) k10_trace[((blockIdx.x * 8 + it) * 8 + wave) * 16 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
// sub-step stamps of chunk 1 of the workgroup's fifth tile: 0..17 sub-step starts, 18 chunk end, 20+b vmcnt wait done at turn b, 24+b barrier passed
__device__ long long k10_trace2[256 * 8 * 32];
#define K10_T2(slot) do { if (lane == 0 && it == 4 && c == 1) k10_trace2[(blockIdx.x * 8 + wave) * 32 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define K10_T(slot)
#define K10_T2(slot)
#endif

// PADL: the last 16 output channels are padding (zero weights - act | GF_CONV_PAD16): their fragment (the last one of every odd
// sub-step) is neither read from the ring nor multiplied; the accumulators keep the shift and the epilogue writes act(shift +
// residual) as for every other channel
// REM (act | GF_CONV_REM8, CIN = 224): the input channels 200 .. 223 carry zero weights (the 196-channel pyramid level padded for the
// matrix cores): six full chunks, then channels 192 .. 199 as a REMAINDER chunk whose 9 taps x 8 channels fill three 32-deep k-steps
// (lane k group g4 of k-step s reads tap 4 s + g4 of its pixel: 16 bytes) - 6 sub-steps instead of the 18 of a seventh chunk, i.e. 114
// instead of 126 sub-steps per tile
// S2 (act | GF_CONV_S2): stride 2 (the first convolution of layer2 / layer3: resnet_fpn.py:14-17 with stride 2).  Same sub-steps, ring and
// epilogue; the weight stream lists the taps plane by plane ((0,0) (0,2) (2,0) (2,2) | (0,1) (2,1) | (1,0) (1,2) | (1,1)), a chunk's patch is
// its four parity planes (ConvGeo), and plane p of the NEXT chunk is requested at the turn that follows the last tap of plane p
// (turns 8, 11, 14, 17 of a chunk's 18 sub-steps): one set of planes does what two patch buffers do at stride 1
template <typename T, int CIN, int COUT, int NW, bool PADL, bool REM = false, bool S2 = false>
__global__ __launch_bounds__(NW * 64) void conv3x3_kernel(ConvArgs a) {
    using Mm = Mma16<T>;
    using Frag = typename Mm::Frag;
    using G = ConvGeo<COUT / 32, NW, REM, S2>;
    static_assert(!(S2 && REM), "no remainder form at stride 2");
    using V4 = gf_vec<T, 4>;
    using V8 = gf_vec<T, 8>;
    static_assert(!REM || CIN == 224, "the remainder form is built for 224-channel inputs");
    constexpr int NT = COUT / 32, NCH = REM ? 6 : CIN / 32, BS = G::BLOCK_STEPS, BPC = 18 / BS, RBPC = 6 / BS;
    constexpr int NBLK = NCH * BPC + (REM ? RBPC : 0);
    constexpr int PB = G::PB, TH = G::TH;
    constexpr int NSTORE = PB * 2 * NT;                             // 16-byte output stores per lane and tile
    // DMA pieces per wave: WP of a weight block, PP of a patch (every wave at least PPMIN).  They are requested in the turn's own
    // sub-step (K10_DEAL=1 builds the round-4 experiment that deals them to the sub-steps behind the turn - weight pieces to the
    // first WS, patch pieces to the other BS - WS, waves 0-3 in front of a sub-step's MFMAs, waves 4-7 behind them: a piece costs its
    // wave 180-300 cycles of issue wherever it stands, and requested later it lands later - an L2 hit takes 2 thousand cycles, a
    // patch piece from HBM 3.2 thousand, against the 4.2 thousand of a block: the turns waited 400-600 cycles on vmcnt, +4 % per tile)
    constexpr int WP = (G::FR + NW - 1) / NW, PP = (G::PIECES + NW - 1) / NW, PPMIN = G::PIECES / NW;    // (S2: PIECES = a plane's)
#ifndef K10_DEAL
#define K10_DEAL 0
#endif
    constexpr bool DEAL = K10_DEAL != 0;
    constexpr int WS = !DEAL ? 1 : WP <= 3 ? WP : 4, PS0 = DEAL ? WS : 0, PS = DEAL ? BS - WS : 1;
    auto wbeg = [](int k) { return k >= WS ? WP : k * (WP / WS) + (k < WP % WS ? k : WP % WS); };
    auto pbeg = [](int k) { return k >= PS ? PP : k * (PP / PS) + (k < PP % PS ? k : PP % PS); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, g4 = lane >> 4;                       // MFMA 16x16x32: row / column of the lane, its k group
    const ConvRsrc ws = conv_rsrc(a.wstream, (unsigned)NBLK * G::WBLK);
    const ConvRsrc xs = conv_rsrc(a.x, (unsigned)a.N * a.Hi * a.Wi * CIN * (unsigned)sizeof(T));
    const ConvRsrc rres = conv_rsrc(a.res, (unsigned)a.N * a.H * a.W * COUT * (unsigned)sizeof(T));
    const ConvRsrc rout = conv_rsrc(a.out, (unsigned)a.N * a.H * a.W * COUT * (unsigned)sizeof(T));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    float* shiftv = reinterpret_cast<float*>(smem + G::SHIFT_OFF);
    for (int i = tid; i < COUT; i += NW * 64) shiftv[i] = a.shift ? a.shift[i] : 0.f;
    if (tid == 0) shiftv[COUT] = a.act == C10_LEAKY ? a.slope : 1.f;     // (read per tile: as a kernel-long live value it is one register too many)
    // the second-dispatched half of an 8-wave workgroup loses every arbitration on its SIMD: one static priority step for it
    // (no per-phase flips) takes 1-2 % off every shape
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);

    auto decode = [&](int t, int& n, int& y0, int& x0) {
        const int per = a.tiles_x * a.tiles_y;
        n = t / per;
        const int r = t - n * per, ty = r / a.tiles_x;
        y0 = ty * TH;
        x0 = (r - ty * a.tiles_x) * TW;
    };
    // A sub-step = (tap, half of the output channels): NT weight fragments (16 channels x the chunk's 32 input channels each)
    // times the 2 PB pixel fragments of the tap (16 pixels x 32 channels = the 64 bytes of a patch pixel: lane (pixel, k
    // group) reads the 16-byte slot k group ^ ((q >> 1) & 3) - conflict-free for every patch offset).
    const int n_none = a.N;                                            // (an image index that is out of range: a request nobody reads)
    const int xq0 = PB * wave * PW + lp;
    // (see load_x) the slot table of the lane: entry d = g4 ^ (((xq0 + d) >> 1) & 3), and the byte offset of its pixel xq0
    int swz = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) swz |= (g4 ^ (((xq0 + d) >> 1) & 3)) << (2 * d);
    const int xbase = xq0 * 64;
    Frag wa[NT], wb[NT], xa[2 * PB], xb[2 * PB];
    auto load_w = [&](Frag (&f)[NT], int slot, int s, bool second_half) {   // weight fragments of sub-step s of the block in `slot`
        const char* p = smem + G::W10_OFF + slot * G::WBLK + s * NT * C10_FRAG + lane * 16;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (!(PADL && second_half && t == NT - 1)) f[t] = *reinterpret_cast<const Frag*>(p + t * C10_FRAG);
    };
    auto load_x = [&](Frag (&f)[2 * PB], int buf, int tap) {       // pixel fragments of `tap` of the chunk in `buf`
        // stride 2: tap index -> (plane, row / column offset inside the plane) in the stream's plane-by-plane order
        constexpr int s2_plane[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3}, s2_dy[9] = {0, 0, 1, 1, 0, 1, 0, 0, 0}, s2_dx[9] = {0, 1, 0, 1, 0, 0, 0, 1, 0};
        const int ky = S2 ? s2_dy[tap] : tap / 3, kx = S2 ? s2_dx[tap] : tap - 3 * (tap / 3);
        const char* p = smem + P_OFF + (S2 ? s2_plane[tap] * G::PLANE_BYTES : buf * G::PATCH_BYTES);
#ifndef K10_SWZ_TABLE
#define K10_SWZ_TABLE 1
#endif
        if constexpr (K10_SWZ_TABLE != 0) {
            // pixel q = xq0 + c, c a compile-time constant of (fragment, tap): its byte offset c * 64 goes into the read's immediate, and the
            // 16-byte slot g4 ^ ((q >> 1) & 3) depends on (xq0 + c) % 8 only - eight 2-bit entries in ONE register per lane (swz):
            // a bit-field extract and a shift-add per fragment instead of six instructions
            int tb = swz, xb = xbase;
            asm volatile("" : "+v"(tb), "+v"(xb));
#pragma unroll
            for (int bb = 0; bb < 2 * PB; ++bb) {
                const int c = ((bb >> 1) + ky) * PW + 16 * (bb & 1) + kx;
                const int slot = __builtin_amdgcn_ubfe(tb, 2 * (c & 7), 2);
                f[bb] = *reinterpret_cast<const Frag*>(p + (xb + (slot << 4)) + c * 64);
            }
        } else {
        int xq = xq0;
        asm volatile("" : "+v"(xq));        // the 36 fragment offsets of a chunk are recomputed (4 VALU each), not kept in registers
#pragma unroll
        for (int bb = 0; bb < 2 * PB; ++bb) {
            const int q = xq + ((bb >> 1) + ky) * PW + 16 * (bb & 1) + kx;
            f[bb] = *reinterpret_cast<const Frag*>(p + q * 64 + ((g4 ^ ((q >> 1) & 3)) << 4));
        }
        }
    };

    auto load_xr = [&](Frag (&f)[2 * PB], int ks) {               // remainder: k group g4 of k-step ks = tap 4 ks + g4 (9 .. 11: zero weights)
        const char* p = smem + G::R_OFF;
        int xq = xq0;
        asm volatile("" : "+v"(xq));
        int tap = 4 * ks + g4;
        tap = tap > 8 ? 8 : tap;                                    // (a pixel that exists: finite values against zero weights)
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;         // tap / 3 for 0 .. 8
#pragma unroll
        for (int bb = 0; bb < 2 * PB; ++bb) {
            const int q = xq + ((bb >> 1) + ky) * PW + 16 * (bb & 1) + kx;
            f[bb] = *reinterpret_cast<const Frag*>(p + q * 16);
        }
    };

    // XCD-aware walk (workgroups are dealt to the 8 XCDs round-robin): XCD x owns the contiguous tile range [xbeg, xend) and
    // its workgroups walk it side by side, so that neighbouring tiles - which share halo rows - meet in one L2
    const int nx8 = gridDim.x >= 8 ? 8 : 1;
    const int xcd = nx8 == 8 ? (int)(blockIdx.x & 7) : 0, xslot = nx8 == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int xper = nx8 == 8 ? (int)((gridDim.x + 7 - xcd) >> 3) : (int)gridDim.x;      // workgroups on this XCD
    const int tq = a.ntiles / nx8, tr = a.ntiles % nx8;
    const int xbeg = xcd * tq + (xcd < tr ? xcd : tr), xend = xbeg + tq + (xcd < tr ? 1 : 0);
    const int tile0 = xbeg + xslot;
    int pbuf = 0, wslot = 0;                                        // patch buffer / ring slot being multiplied
    if (tile0 < xend) {
        int n, y0, x0;
        decode(tile0, n, y0, x0);
        if constexpr (S2) {
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) conv_dma_plane<T, CIN, NT, NW>(a, xs, smem, pl, n, y0, x0, 0, wave, lane);
        } else {
            conv_dma_patch<T, CIN, NT, NW>(a, xs, smem, 0, n, y0, x0, 0, wave, lane);
        }
        conv_dma_block<NT, NW, S2>(ws, smem, 0, 0, wave, lane);
        conv_dma_block<NT, NW, S2>(ws, smem, 1, 1, wave, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    int it = -1;
    for (int tile = tile0; tile < xend; tile += xper) {
        int n, y0, x0;
        decode(tile, n, y0, x0);
        const int nxt_tile = tile + xper;
        const bool has_next = nxt_tile < xend;
        int n2 = n_none, y2 = 0, x2 = 0;                              // (no next tile: its patch requests name an image that is out of range)
        if (has_next) decode(nxt_tile, n2, y2, x2);
        ++it;
        K10_T(0);
        load_w(wa, wslot, 0, false);               // block 0 and the patch of chunk 0 landed before the previous tile's last turn
        load_x(xa, pbuf, 0);
        // acc[bb][ct]: pixel block bb (row bb / 2 of the wave, pixels 16 (bb % 2) .. + 15), channels 32 (ct / 2) + 8 g4 + 4 (ct % 2) + {0..3}
        // (the packing deals the output channels to the fragment rows so that the tiles 2 j and 2 j + 1 of a lane are EIGHT CONSECUTIVE
        // channels of its pixel: 16 bytes of the NHWC output - the epilogue stores from registers); accumulators start at the shift
        v4f acc[2 * PB][2 * NT];
        int il = lane;
        asm volatile("" : "+v"(il));
#pragma unroll
        for (int ct = 0; ct < 2 * NT; ++ct) {
            const float4 sh = *reinterpret_cast<const float4*>(shiftv + 32 * (ct >> 1) + 4 * (ct & 1) + 8 * (il >> 4));
#pragma unroll
            for (int bb = 0; bb < 2 * PB; ++bb) acc[bb][ct] = v4f{sh.x, sh.y, sh.z, sh.w};
        }
        // one chunk of the K loop: a FULL chunk c (32 input channels: 9 taps x 2 halves of the output channels = 18 sub-steps) or the
        // REMAINDER chunk (REM: 8 channels: 3 k-steps x 2 halves = 6 sub-steps; its patch has its own buffer, so the two patch buffers keep
        // alternating over the full chunks and the next tile's first patch is requested during the last FULL chunk, as without it)
        auto chunk = [&](auto rem_c, int c) {
            constexpr bool R = decltype(rem_c)::value;
            constexpr int NTS = R ? 6 : 18;
            const int blk0 = R ? NCH * BPC : c * BPC;                       // this chunk's first weight block
            const bool tile_ends = R || (!REM && c + 1 == NCH);              // no chunk of this tile behind this one
            const bool rem_next = !R && REM && c + 1 == NCH;                 // the remainder follows
#pragma unroll
            for (int ts = 0; ts < NTS; ++ts) {                               // sub-step (tap or k-step ts / 2, channel half ts % 2)
                const int tap = ts >> 1, hf = ts & 1;
                Frag (&cw)[NT] = (ts & 1) ? wb : wa;
                Frag (&nw)[NT] = (ts & 1) ? wa : wb;
                Frag (&cx)[2 * PB] = (tap & 1) ? xb : xa;
                Frag (&nx)[2 * PB] = (tap & 1) ? xa : xb;
                // (this sub-step's fragments were requested a sub-step ago; the compiler's counted lgkmcnt waits retire them)
                const int kslot = (ts + 1) % BS;                               // sub-steps since the last ring turn (0: the turn is in this one)
                if constexpr (!R) K10_T2(ts);
                // the DMA requests behind a turn: the weight pieces of the block after the landed one, then (a full chunk's first turn)
                // the pieces of the next patch - in this order: the chunk's second turn waits for the weights only
                auto dma_slot = [&]() {
                    if (kslot < WS && !(ts == NTS - 1 && tile_ends)) {            // (a tile's last turn: issued whole at the turn)
                        if (ts >= BS - 1 || R || c > 0) {                        // (a tile's first sub-steps: nothing is pending)
                            const int nb = blk0 + (ts + 1) / BS + 1;             // = block being finished + 2 at the turn, current block + 1 behind it
                            // (one call, no branch: behind the last tile's last blocks the request names a block outside the stream - out of
                            // range: zeros into a slot nobody reads)
                            conv_dma_block<NT, NW, S2>(ws, smem, nb < NBLK ? nb : has_next ? nb - NBLK : NBLK, wslot ^ 1, wave, lane, wbeg(kslot), wbeg(kslot + 1));
                        }
                    }
                    if constexpr (S2) {
                        // plane (ts - 8) / 3 of the next chunk: its last tap's fragments are in registers, every wave is past the barrier
                        if (kslot == 0 && ts >= 8) {
                            const bool same = c + 1 < NCH;
                            conv_dma_plane<T, CIN, NT, NW>(a, xs, smem, (ts - 8) / 3, same ? n : n2, same ? y0 : y2, same ? x0 : x2, same ? c + 1 : 0, wave, lane);
                        }
                    } else if (!R && ts >= BS - 1 + PS0 && ts < BS - 1 + PS0 + PS) {
                        // behind the first turn of a full chunk every wave is past the previous chunk: its patch buffer is free
                        const int kp = ts - (BS - 1 + PS0);
                        // (one call: the next chunk of this tile, or the first of the next tile; behind the last tile's last chunk: image N, out of range)
                        const bool same = c + 1 < NCH;
                        conv_dma_patch<T, CIN, NT, NW>(a, xs, smem, pbuf ^ 1, same ? n : n2, same ? y0 : y2, same ? x0 : x2, same ? c + 1 : 0,
                                                       wave, lane, pbeg(kp), pbeg(kp + 1));
                        // the tile's remainder patch: its buffer is free since every wave left the previous tile's remainder chunk
                        if constexpr (REM) {
                            if (c == 0 && kp == PS - 1) conv_dma_patch_rem<T, CIN, NT, NW>(a, xs, smem, n, y0, x0, wave, lane);
                        }
                    }
                };
                if (kslot == 0) {
                    // ring turn before the last sub-step of a block: the next block (and a patch requested a turn ago) has
                    // landed, every wave holds this block's last fragments in registers: its slot takes the block after next.
                    // What must have landed: the weight pieces, requested in the sub-steps behind the previous turn; a chunk's SECOND
                    // turn leaves the patch pieces requested behind them in flight (every wave has issued at least PPMIN of them);
                    // the first turn of a tile only needs the block requested before the previous tile's epilogue: its NSTORE output
                    // stores - a store outside the image is issued too, and dropped - may stay in flight
                    const bool first_turn = !R && ts == BS - 1 && c == 0;
                    if (first_turn) K10_T(12);
                    if (first_turn && it > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
                    else if (!S2 && !R && ts == 2 * BS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPMIN) : "memory");
                    // (S2: behind a turn that requested a plane - weights first, then the plane - the plane's pieces stay in flight; it is
                    // needed nine sub-steps after its request at the earliest, and two turns later all of it has landed)
                    else if (S2 && (ts >= 11 || (ts == BS - 1 && c > 0))) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPMIN) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (first_turn) K10_T(13);
                    if constexpr (!R) K10_T2(20 + ts / BS);
                    __builtin_amdgcn_s_barrier();
#ifdef K10_STAGGER     // -DK10_STAGGER=n (experiment, round 5; guide: 'two waves that run the same program with one barrier per block: try a stagger'):
                    // the second-dispatched half of the waves sleeps 64 n cycles behind every ring turn, so that the two waves of a SIMD do not reach
                    // their fragment-read bursts and MFMA runs together
                    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_sleep(K10_STAGGER);
#endif
                    if constexpr (!R) K10_T2(24 + ts / BS);
                    if (first_turn) K10_T(14);
                    wslot ^= 1;                                                 // the landed block; the other slot is the one to fill
                    if (ts == NTS - 1 && tile_ends) {
                        // last turn of a tile: the whole request at once (dealt to the next tile's first sub-steps its pieces would
                        // queue behind the epilogue's stores, and the next first turn would have to wait for those)
                        const int nb = blk0 + ts / BS + 2;
                        conv_dma_block<NT, NW, S2>(ws, smem, has_next ? nb - NBLK : NBLK, wslot ^ 1, wave, lane);      // (no next tile: out of range)
                    }
                    if (!DEAL) {
                        dma_slot();
                        if (first_turn) K10_T(15);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (in front of the fragment reads, behind the MFMAs: where the sub-step's 32-64 fragment registers of one buffer are dead)
                if (DEAL && wave < 4) {
                    dma_slot();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // next sub-step's weight fragments, and behind a tap's second half the next tap's pixel fragments (a tile's
                // first ones are read at its start: held across the epilogue they spill)
                if (ts < NTS - 1 || !tile_ends) {
                    load_w(nw, wslot, (ts + 1) % BS, ((ts + 1) & 1) != 0);
                    if (hf) {
                        if constexpr (R) {
                            if (ts < NTS - 1) load_xr(nx, tap + 1);
                        } else {
                            if (ts < NTS - 1) load_x(nx, pbuf, tap + 1);
                            else if (rem_next) load_xr(nx, 0);
                            else load_x(nx, pbuf ^ 1, 0);
                        }
                    }
                }
                // (no issue-order hints: with 128 accumulators in four-register tuples the allocator gives an MFMA's result
                // other registers than its addend, and every constraint on the order - sched_group_barrier pipelines, a
                // barrier between requests and MFMAs, opaque in-order instructions - measured slower or spilled)
#pragma unroll
                for (int bb = 0; bb < 2 * PB; ++bb)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (!(PADL && hf == 1 && t == NT - 1)) Mm::mma(cw[t], cx[bb], acc[bb][hf * NT + t]);
                if (DEAL && wave >= 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma_slot();
                }
                // issue order: two reads, then one read behind each of the first MFMAs (left alone the compiler sinks the
                // reads behind the sub-step's last MFMA and the next one waits out the whole LDS latency)
                __builtin_amdgcn_sched_barrier(0);
            }
            // a chunk has an odd number of taps / k-steps (9, 3): the next chunk's first fragments were prefetched into xb, and its
            // first tap reads xa
            if (!tile_ends) {
#pragma unroll
                for (int bb = 0; bb < 2 * PB; ++bb) xa[bb] = xb[bb];
            }
            if constexpr (!R) pbuf ^= 1;
            if constexpr (!R) K10_T2(18);
        };
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            chunk(std::integral_constant<bool, false>{}, c);
            K10_T(2 + c);
        }
        if constexpr (REM) chunk(std::integral_constant<bool, true>{}, NCH);
        // ---------------- epilogue (wave-private, registers only): T(acc) + shortcut -> act -> 16-byte NHWC stores
        K10_T(10);
        // one straight-line body per (shortcut?, activation form): with these as run-time branches inside the body every join
        // waits for vmcnt(0), i.e. for the stores in flight
        auto epilogue = [&](auto has_res_c, auto mode_c) {
            constexpr bool HAS_RES = decltype(has_res_c)::value;
            constexpr int MODE = decltype(mode_c)::value;       // 0 packed-half ReLU, 1 packed-half none, 2 fp32 ReLU, 3 fp32 max(f, k f)
            int el = lane;                  // (recompute the lane's pixel offsets per tile: kept across the K loop they spill)
            asm volatile("" : "+v"(el));
            const float neg_k = MODE == 3 ? shiftv[COUT + (el >> 6)] : 1.f;
            // Accumulator layout: lane (pixel lp = el % 16 of block bb, k group g4 = el / 16) holds the 16-byte pieces j = 0 .. NT - 1 of its
            // pixel: channels 32 j + 8 g4 .. + 7 - stored as they are, an instruction would write 16 pixels x 64 bytes, and half-line
            // writes run at 0.4 of the rate of whole 128-byte lines (tools/probes/epi_store.hip: 6900 against 2600 cycles per tile).
            // So neighbouring pixels exchange pieces first (one DPP move per register): instruction A of (bb, m) writes the block's EVEN
            // pixels, instruction B the odd ones, 128 bytes each - the even lane of a pair brings piece 2 m, the odd lane piece 2 m + 1.
            // Buffer addressing: a 32-bit byte offset per (block, instruction) - out of range for a pixel outside the image (its load
            // returns zeros, its store is dropped) - and 128 m in the instruction's immediate.
            const int odd = el & 1, ey = y0 + PB * wave;
            const int exa = x0 + (el & 14), exo = x0 + (el & 15);
            const int ebase = ((n * a.H + ey) * a.W + x0) * COUT * (int)sizeof(T) + 16 * (el >> 4);        // (32-bit: the entry point bounds the maps)
#ifndef K10_XCHG_WIDE
#define K10_XCHG_WIDE 1
#endif
            constexpr int NB = 2 * PB, NM = (PB == 2 || K10_XCHG_WIDE) ? NT / 2 : 0, NL = NT - 2 * NM;
            constexpr bool LONE = NL != 0;                    // pieces 2 NM .. NT - 1 are stored in the accumulator layout (224 channels: piece 6 has no partner)
            int voa[NB], vob[NB], vol[LONE ? NB : 1];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const int rowoff = ((bb >> 1) * a.W + 16 * (bb & 1)) * COUT * (int)sizeof(T);
                const bool iny = ey + (bb >> 1) < a.H;
                voa[bb] = iny && exa + 16 * (bb & 1) < a.W ? ebase + rowoff + (el & 14) * COUT * (int)sizeof(T) + 64 * odd : 0x7FFFFFF0;
                vob[bb] = iny && exa + 1 + 16 * (bb & 1) < a.W ? ebase + rowoff + ((el & 14) + 1) * COUT * (int)sizeof(T) + 64 * odd : 0x7FFFFFF0;
                if constexpr (LONE) vol[bb] = iny && exo + 16 * (bb & 1) < a.W ? ebase + rowoff + (el & 15) * COUT * (int)sizeof(T) + 64 * (2 * NM) : 0x7FFFFFF0;
            }
            auto pack8 = [](const v4f& lo, const v4f& hi) {
                return V8{(T)lo[0], (T)lo[1], (T)lo[2], (T)lo[3], (T)hi[0], (T)hi[1], (T)hi[2], (T)hi[3]};
            };
            // pieces p0 (2 m) and p1 (2 m + 1) of the lane's pixel -> what the lane stores in instruction A / B
            auto exchange = [&](const V8& p0, const V8& p1, V8& da, V8& db) {
                const v4u u0 = __builtin_bit_cast(v4u, p0), u1 = __builtin_bit_cast(v4u, p1);
                v4u ua, ub;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned send = odd ? u0[i] : u1[i];                                               // the piece the neighbour stores
                    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                    ua[i] = odd ? recv : u0[i];
                    ub[i] = odd ? u1[i] : recv;
                }
                da = __builtin_bit_cast(V8, ua);
                db = __builtin_bit_cast(V8, ub);
            };
            auto finish = [&](const V8& v, const V8& r) {                  // + shortcut, activation (v: the sum rounded to the storage type)
                V8 o;
                if constexpr (MODE <= 1) {
                    // packed half arithmetic: the sum of two halves rounded to half is what the fp32 route gives
                    // (up to a double rounding when the exponents are > 13 apart), max is exact
                    o = v;
                    if constexpr (HAS_RES) o = o + r;
                    if constexpr (MODE == 0) o = __builtin_elementwise_max(o, V8{0, 0, 0, 0, 0, 0, 0, 0});
                } else {
                    float f[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
                    if constexpr (HAS_RES) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] += (float)r[i];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (T)(MODE == 2 ? fmaxf(f[i], 0.f) : fmaxf(f[i], f[i] * neg_k));
                }
                return o;
            };
            auto load8 = [&](int vo, int imm) { return __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rres.r, vo + imm, 0, 0)); };
            auto store8 = [&](const V8& o, int vo, int imm) {
                if (K10_STORE_OK) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rout.r, vo + imm, 0, 0);
            };
            // every shortcut piece is requested before the first output store: a load waited for behind a store (the memory
            // counter is in issue order) would expose the store's round trip (the LeakyReLU-with-shortcut body, which no layer of the
            // backbone uses, goes pixel block by pixel block instead: its fp32 route would spill beside all the shortcut rows)
            constexpr int BATCH = (HAS_RES && MODE == 3) ? 1 : NB;
#pragma unroll
            for (int b0 = 0; b0 < NB; b0 += BATCH) {
                V8 ra[HAS_RES ? BATCH : 1][HAS_RES ? NM : 1], rb[HAS_RES ? BATCH : 1][HAS_RES ? NM : 1], rlone[HAS_RES && LONE ? BATCH : 1][LONE ? NL : 1];
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int bq = 0; bq < BATCH; ++bq) {
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            ra[bq][m] = load8(voa[b0 + bq], 128 * m);
                            rb[bq][m] = load8(vob[b0 + bq], 128 * m);
                        }
                        if constexpr (LONE) {
#pragma unroll
                            for (int q = 0; q < NL; ++q) rlone[bq][q] = load8(vol[b0 + bq], 64 * q);
                        }
                    }
                }
#pragma unroll
                for (int bq = 0; bq < BATCH; ++bq) {
                    const int bb = b0 + bq;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        V8 da, db;
                        exchange(pack8(acc[bb][4 * m], acc[bb][4 * m + 1]), pack8(acc[bb][4 * m + 2], acc[bb][4 * m + 3]), da, db);
                        store8(finish(da, ra[HAS_RES ? bq : 0][HAS_RES ? m : 0]), voa[bb], 128 * m);
                        store8(finish(db, rb[HAS_RES ? bq : 0][HAS_RES ? m : 0]), vob[bb], 128 * m);
                        // (left alone the scheduler runs the arithmetic of every piece before the first store: 100+ more live registers)
                        if constexpr (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (LONE) {
#pragma unroll
                        for (int q = 0; q < NL; ++q) {
                            store8(finish(pack8(acc[bb][4 * NM + 2 * q], acc[bb][4 * NM + 2 * q + 1]), rlone[HAS_RES ? bq : 0][q]), vol[bb], 64 * q);
                            if constexpr (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        };
        using std::integral_constant;
        auto by_res = [&](auto mode_c) {
            if (a.res) epilogue(integral_constant<bool, true>{}, mode_c);
            else epilogue(integral_constant<bool, false>{}, mode_c);
        };
        if constexpr (std::is_same<T, _Float16>::value) {
            if (a.act == C10_RELU) by_res(integral_constant<int, 0>{});
            else if (a.act == C10_NONE) by_res(integral_constant<int, 1>{});
            else by_res(integral_constant<int, 3>{});
        } else {
            if (a.act == C10_RELU) by_res(integral_constant<int, 2>{});
            else by_res(integral_constant<int, 3>{});
        }
        K10_T(11);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

#if K10_TRACE
}
extern "C" int gf_debug_k10_trace(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k10_trace), sizeof(k10_trace)) == hipSuccess ? 0 : -1;
}
extern "C" int gf_debug_k10_trace2(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k10_trace2), sizeof(k10_trace2)) == hipSuccess ? 0 : -1;
}
namespace {
#endif

template <typename T, int CIN, int COUT, int NW, bool PADL, bool REM = false, bool S2 = false>
int conv_launch(ConvArgs a, hipStream_t st) {
    using G = ConvGeo<COUT / 32, NW, REM, S2>;
    static std::atomic<uint64_t> attr{0};
    if (gf_first_use_on_device(attr))
        (void)hipFuncSetAttribute((const void*)conv3x3_kernel<T, CIN, COUT, NW, PADL, REM, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    const long nt = (long)a.N * a.tiles_x * a.tiles_y;
    if (nt >= (1l << 31)) return -2;
    a.ntiles = (int)nt;
    conv3x3_kernel<T, CIN, COUT, NW, PADL, REM, S2><<<a.ntiles < 256 ? a.ntiles : 256, NW * 64, G::LDS, st>>>(a);
    return 0;
}

// 8 waves per workgroup: measured 1.1-1.3x faster than 4 at every shape (tools/k10_time.py); the 4-wave form stays
// instantiable (conv_launch<..., 4>) for experiments
template <typename T, int CIN, int COUT, bool PADL = false, bool REM = false, bool S2 = false>
int conv_launch_w(const ConvArgs& a, hipStream_t st) {
    return conv_launch<T, CIN, COUT, 8, PADL, REM, S2>(a, st);
}

template <typename T>
int conv_dispatch(const ConvArgs& a, int cin, int cout, bool padl, bool rem, bool s2, hipStream_t st) {
    if (s2) {                                            // the stride-2 convolutions of the (128, 196 -> 224, 256) pyramid
        if (cin == 128 && cout == 224 && padl) return conv_launch_w<T, 128, 224, true, false, true>(a, st);
        if (cin == 128 && cout == 224) return conv_launch_w<T, 128, 224, false, false, true>(a, st);
        if (cin == 224 && cout == 256) return conv_launch_w<T, 224, 256, false, false, true>(a, st);
        return -1;
    }
    if (rem) {                                           // the 196-channel level as input: 6 chunks + the 8-channel remainder
        if (cin == 224 && cout == 224 && padl) return conv_launch_w<T, 224, 224, true, true>(a, st);
        if (cin == 224 && cout == 128 && !padl) return conv_launch_w<T, 224, 128, false, true>(a, st);
        return -1;
    }
    if (cin == 128 && cout == 128) return conv_launch_w<T, 128, 128>(a, st);
    if (cin == 224 && cout == 224 && padl) return conv_launch_w<T, 224, 224, true>(a, st);
    if (cin == 256 && cout == 224 && padl) return conv_launch_w<T, 256, 224, true>(a, st);
    if (cin == 224 && cout == 224) return conv_launch_w<T, 224, 224>(a, st);
    if (cin == 224 && cout == 128) return conv_launch_w<T, 224, 128>(a, st);
    if (cin == 256 && cout == 256) return conv_launch_w<T, 256, 256>(a, st);
    if (cin == 256 && cout == 224) return conv_launch_w<T, 256, 224>(a, st);
    // the transposed widths of the two FPN heads: their backward-data passes as forward convolutions (train/hip_autograd.py:Conv3x3Function)
    if (cin == 128 && cout == 224) return conv_launch_w<T, 128, 224>(a, st);
    if (cin == 224 && cout == 256) return conv_launch_w<T, 224, 256>(a, st);
    return -1;
}

}   // namespace

// 1 if gf_conv3x3_nhwc has a stride-2 kernel (act | GF_CONV_S2) for these channel counts
extern "C" int gf_conv3x3s2_supported(int cin, int cout) { return (cin == 128 && cout == 224) || (cin == 224 && cout == 256); }

// 1 if gf_conv3x3_nhwc has a kernel for these channel counts
extern "C" int gf_conv3x3_supported(int cin, int cout) {
    return (cin == 128 && (cout == 128 || cout == 224)) || (cin == 224 && (cout == 224 || cout == 128 || cout == 256)) || (cin == 256 && (cout == 256 || cout == 224));
}

// out = act(conv3x3(x, w) + shift + residual), channels-last 16-bit maps; wstream = fused.py:pack_conv3x3_stream(w);
// zeros = >= 64 bytes of device memory holding zeros (the source of out-of-image pixels)
extern "C" int gf_conv3x3_nhwc(const void* x, const void* wstream, const float* shift, const void* residual, void* out,
                               const void* zeros, int N, int H, int W, int cin, int cout, int act, float slope, int dtype,
                               void* stream) {
    GF_CHECK_ARG(x && wstream && out && zeros, "null pointer");
    GF_CHECK_ARG(N > 0 && H > 0 && W > 0, "empty problem");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit maps");
    const bool padl = (act & GF_CONV_PAD16) != 0;     // the output channels 196.. are padding (zero weights): their own tile is not multiplied
    const bool rem = (act & GF_CONV_REM8) != 0;        // the input channels 200.. carry zero weights; wstream is the rem8 packing
    const bool s2 = (act & GF_CONV_S2) != 0;           // stride 2: H x W is the INPUT map, out / residual are [N][(H-1)/2+1][(W-1)/2+1][cout]
    act &= ~(GF_CONV_PAD16 | GF_CONV_REM8 | GF_CONV_S2);
    GF_CHECK_ARG(s2 ? gf_conv3x3s2_supported(cin, cout) : gf_conv3x3_supported(cin, cout),
                 "no kernel for these channel counts (see gf_conv3x3_supported / gf_conv3x3s2_supported)");
    GF_CHECK_ARG(!rem || (!s2 && cin == 224 && ((cout == 224 && padl) || (cout == 128 && !padl))),
                 "GF_CONV_REM8 is built for stride 1, Cin = 224 with Cout = 224 | GF_CONV_PAD16 or Cout = 128");
    GF_CHECK_ARG(!padl || cout == 224, "GF_CONV_PAD16 belongs to 224-wide outputs");
    GF_CHECK_ARG(act >= C10_NONE && act <= C10_LEAKY, "unknown activation");
    GF_CHECK_ARG(act != C10_LEAKY || (slope >= 0.f && slope <= 1.f), "LeakyReLU slope must lie in [0, 1]");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)residual % 16 == 0 && (uintptr_t)wstream % 16 == 0 &&
                     (uintptr_t)zeros % 16 == 0, "tensors must be 16-byte aligned");
    GF_CHECK_ARG((long)N * H * W * (cin > cout ? cin : cout) * 2 < 0x7FFFFFF0l, "maps of 2 GiB or more are not supported (32-bit buffer offsets)");
    const int Ho = s2 ? (H - 1) / 2 + 1 : H, Wo = s2 ? (W - 1) / 2 + 1 : W;
    ConvArgs a{x, wstream, shift, residual, out, zeros, N, Ho, Wo, act, slope, 0, 0, 0, H, W};
    hipStream_t st = (hipStream_t)stream;
    // declared work = the reference's convolution: a 224-wide operand is the zero-padded form of the backbone's 196-channel maps
    // (resnet_fpn.py block_dims (128, 196, 256); model/backbone.py pads them for the matrix cores) and the padding is not work
    const double cin_w = cin == 224 ? 196.0 : cin, cout_w = cout == 224 ? 196.0 : cout;
    void* pt = gf_prof_begin("conv3x3", st, 2.0 * N * (double)Ho * Wo * cin_w * cout_w * 9.0);
    const int rc = dtype == GF_F16 ? conv_dispatch<_Float16>(a, cin, cout, padl, rem, s2, st) : conv_dispatch<gf_bf16>(a, cin, cout, padl, rem, s2, st);
    gf_prof_end("conv3x3", pt, st);
    GF_CHECK_ARG(rc == 0, "dispatch failed");
    GF_CHECK_LAUNCH();
    return GF_OK;
}
