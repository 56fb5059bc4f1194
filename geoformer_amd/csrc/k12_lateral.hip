// K12: the FPN lateral of the 1/2-scale level with its top-down merge as ONE streaming kernel (backbone glue, SURVEY 8 f4:
// model/loftr_src/loftr/backbone/resnet_fpn.py:104-111  x1_out = layer1_outconv(x1) + interpolate(x2_out, scale 2, bilinear, align_corners)):
//     out[n,y,x,:] = W . x[n,y,x,:] + bilinear(lo -> H x W, align_corners=True)[n,y,x,:]            (16-bit channels-last maps)
// An HBM-bound op - 67 flop per byte: x + out + the coarser map once = 1.34 GB at 16 x 320 x 320 - that ran at 0.33-0.40 of its bound on
// the K3 tile engine (gf_conv1x1_upsample_add_nhwc: rows through registers into LDS, one tile per workgroup, the taps of the merge
// gathered row group by row group).  Here:
//   weights    = all CIN x COUT of them resident in LDS for the life of the (persistent) workgroup, as the 16x16x32 MFMA's A fragments
//                (fused.py:pack_lateral_frags: the output channels dealt to the fragment rows as in K10, so that two accumulator tiles
//                of a lane are 8 consecutive channels);
//   pixels     = a tile is 128 consecutive pixels, a wave owns 16 of them for the whole tile and brings ITS OWN rows in by LDS-DMA
//                (4 pieces of 16 pixels x 64 B per tile, two wave-private buffers): there is no workgroup barrier behind the
//                prologue - the eight waves of a workgroup drift apart, one's epilogue under another's MFMAs;
//   merge      = the 4 x 7 tap pieces of a lane are requested BEFORE the tile's MFMAs (they do not depend on them) in the layout the
//                result is stored in, and consumed as four v_fma_mix_f32 per value (fp16; conversions + FMAs for bf16);
//   stores     = neighbouring pixels exchange pieces by DPP (as in K10's epilogue): 8 pixels x 128 B per instruction; an iteration requests
//                the NEXT tile's taps (and the rows of the one after) in front of ITS OWN stores, so no load is waited for behind a store.
// Measured (16 x 320 x 320, fp16): 407 us against 491 of the K3 form (586 at the start of round 4).  What bounds it is the vector-memory
// issue path: per tile a wave issues 28 tap loads + 4 LDS-DMA pieces + 7 stores of 1 KB and spends 45 % of its cycles doing so (-DK12_TRACE=1:
// product 32 %, merge 10 %, requests 45 %, stores 14 %) - 313 KB per tile pass the CU's texture path, 224 KB of them taps (with every tap
// piece read from one address: 253 us; with the stores compiled out: 68 us; 734 MB of pure stores: 135 us, tools/probes/row448.hip).
// The STAGED form below (widths that are multiples of 16, scale factor 2: the FPN case) cuts the taps' share of that path: 326 us.
#include <type_traits>

#include "gf_common.h"

namespace {

constexpr int L12_TP = 128;          // pixels per tile
constexpr int L12_NW = 8;            // waves per workgroup

struct LatArgs {
    const void* x;          // [P][CIN]
    const void* wfrag;      // packed A fragments [CIN/32][COUT/16][64][8]
    const void* lo;         // [N][h][w][COUT]
    void* out;              // [P][COUT]
    int N, H, W, h, w, P, ntiles;
    float ry, rx;
};

// -DK12_TRACE=1: per wave, the cycles spent in the four phases of an iteration, summed over its tiles (tools/k3_upadd_trace.py)
#ifndef K12_TRACE
#define K12_TRACE 0
#endif
#if K12_TRACE
__device__ long long k12_trace[256 * 8 * 8];
#define K12_T(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); ph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define K12_T(i)
#endif

struct LatRsrc {
    __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ LatRsrc lat_rsrc(const void* p, unsigned bytes) {
    return LatRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ void lat_lds_dma(const LatRsrc& rs, char* dst, int voffset, int soffset) {      // 64 lanes x 16 B -> 1 KiB at dst
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.r, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}

template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(L12_NW * 64) void lateral_kernel(LatArgs a) {
    using Mm = Mma16<T>;
    using Frag = typename Mm::Frag;
    using V8 = gf_vec<T, 8>;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int KC = CIN / 32, NCT = COUT / 16, NP = COUT / 32, NM = NP / 2;
    constexpr bool LONE = (NP & 1) != 0;
    constexpr int W_BYTES = KC * NCT * 1024, XW = KC * 1024;                 // a wave's x buffer: KC pieces of 16 pixels x 64 B
    constexpr int X_OFF = W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, g4 = lane >> 4, odd = lane & 1;
    const LatRsrc xs = lat_rsrc(a.x, (unsigned)a.P * CIN * (unsigned)sizeof(T));
    const LatRsrc wsr = lat_rsrc(a.wfrag, (unsigned)W_BYTES);
    const LatRsrc los = lat_rsrc(a.lo, (unsigned)a.N * a.h * a.w * COUT * (unsigned)sizeof(T));
    const LatRsrc outs = lat_rsrc(a.out, (unsigned)a.P * COUT * (unsigned)sizeof(T));

    // XCD-aware walk (workgroups are dealt to the 8 XCDs round-robin): XCD x owns a contiguous range of tiles, so the rows of the
    // coarser map a run of tiles shares meet in one L2
    const int nx8 = gridDim.x >= 8 ? 8 : 1;
    const int xcd = nx8 == 8 ? (int)(blockIdx.x & 7) : 0, xslot = nx8 == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int xper = nx8 == 8 ? (int)((gridDim.x + 7 - xcd) >> 3) : (int)gridDim.x;
    const int tq = a.ntiles / nx8, tr = a.ntiles % nx8;
    const int xbeg = xcd * tq + (xcd < tr ? xcd : tr), xend = xbeg + tq + (xcd < tr ? 1 : 0);
    const int tile0 = xbeg + xslot;
    if (tile0 >= xend) return;

    // the wave's own 16 pixels of tile t, channel chunk kc: lane (pixel q = lane / 4, 16-byte slot lane % 4 ^ (q / 2) % 4 on the source
    // side: the fragment reads of 16 pixels x 4 k groups are then conflict-free); pixels behind the last one are out of range = zeros
    auto dma_x = [&](int t, int buf) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int q = l >> 2, slot = (l & 3) ^ ((q >> 1) & 3);
        const int off = ((t * L12_TP + wave * 16 + q) * CIN + 8 * slot) * (int)sizeof(T);
        char* dst = smem + X_OFF + (buf * L12_NW + wave) * XW;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) lat_lds_dma(xs, dst + kc * 1024, off, kc * 64);
    };
    // prologue: the weights (once per workgroup), the first tile
#pragma unroll
    for (int i = 0; i < (KC * NCT + L12_NW - 1) / L12_NW; ++i) {
        const int f = wave + L12_NW * i;
        if (f < KC * NCT) lat_lds_dma(wsr, smem + f * 1024, lane * 16, f * 1024);
    }
    dma_x(tile0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- per-tile state of the lane: instruction A of a channel-pair pair writes the wave's EVEN pixels, B the odd ones, the lone piece
    // (224 channels: piece 6) the lane's own pixel; W is even (entry point), so a pixel pair shares its row.  tA / tB / tL: the four tap
    // pieces (y0,x0) (y0,x1) (y1,x0) (y1,x1) of every piece the lane stores, wA / wB / wL their weights, sA / sB / sL the store offsets
    // (out of range for pixels behind the last one: loads return zeros, stores are dropped)
    V8 tA[NM][4], tB[NM][4], tL[LONE ? 4 : 1];
    float wA[4], wB[4], wL[4];
    int sA = 0, sB = 0, sL = 0;
    auto request_taps = [&](int tile) {
        int el = lane;
        asm volatile("" : "+v"(el));
        const int pa = tile * L12_TP + wave * 16 + (el & 14);
        const unsigned qy = (unsigned)pa / (unsigned)a.W;
        const int xa = pa - (int)qy * a.W, n = (int)(qy / (unsigned)a.H), y = (int)qy - n * a.H;
        const bool live = pa < a.P;
        const float fy = a.ry * y;
        const int y0 = (int)fy, y1 = y0 + (y0 < a.h - 1);
        const float wy1 = fy - y0, wy0 = 1.f - wy1;
        const int cb = 64 * (el & 1) + 16 * (el >> 4);                           // the lane's bytes inside a 128-byte channel-pair pair
        const int rb0 = ((n * a.h + y0) * a.w) * COUT * (int)sizeof(T), rb1 = ((n * a.h + y1) * a.w) * COUT * (int)sizeof(T);
        auto taps = [&](int x, int colbyte, int (&vo)[4], float (&wt)[4]) {
            const float fx = a.rx * x;
            const int x0 = (int)fx, x1 = x0 + (x0 < a.w - 1);
            const float wx1 = fx - x0, wx0 = 1.f - wx1;
            vo[0] = live ? rb0 + x0 * COUT * (int)sizeof(T) + colbyte : 0x7FFFFFF0;
            vo[1] = live ? rb0 + x1 * COUT * (int)sizeof(T) + colbyte : 0x7FFFFFF0;
            vo[2] = live ? rb1 + x0 * COUT * (int)sizeof(T) + colbyte : 0x7FFFFFF0;
            vo[3] = live ? rb1 + x1 * COUT * (int)sizeof(T) + colbyte : 0x7FFFFFF0;
            wt[0] = wy0 * wx0; wt[1] = wy0 * wx1; wt[2] = wy1 * wx0; wt[3] = wy1 * wx1;
        };
        int voA[4], voB[4], voL[4];
        taps(xa, cb, voA, wA);
        taps(xa + 1, cb, voB, wB);
        if constexpr (LONE) taps(xa + (el & 1), 128 * NM + 16 * (el >> 4), voL, wL);
        auto load8 = [&](int vo, int imm) { return __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(los.r, vo + imm, 0, 0)); };
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tA[m][k] = load8(voA[k], 128 * m);
                tB[m][k] = load8(voB[k], 128 * m);
            }
        if constexpr (LONE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tL[k] = load8(voL[k], 0);
        }
        sA = live ? pa * COUT * (int)sizeof(T) + cb : 0x7FFFFFF0;
        sB = live && pa + 1 < a.P ? (pa + 1) * COUT * (int)sizeof(T) + cb : 0x7FFFFFF0;
        const int pl = pa + (el & 1);
        sL = pl < a.P ? pl * COUT * (int)sizeof(T) + 128 * NM + 16 * (el >> 4) : 0x7FFFFFF0;
    };
    auto pack8 = [](const v4f& lo_, const v4f& hi_) {
        return V8{(T)lo_[0], (T)lo_[1], (T)lo_[2], (T)lo_[3], (T)hi_[0], (T)hi_[1], (T)hi_[2], (T)hi_[3]};
    };
    auto exchange = [&](const V8& p0, const V8& p1, V8& da, V8& db) {
        const v4u u0 = __builtin_bit_cast(v4u, p0), u1 = __builtin_bit_cast(v4u, p1);
        v4u ua, ub;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned send = odd ? u0[i] : u1[i];
            const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
            ua[i] = odd ? recv : u0[i];
            ub[i] = odd ? u1[i] : recv;
        }
        da = __builtin_bit_cast(V8, ua);
        db = __builtin_bit_cast(V8, ub);
    };
    auto merge = [&](const V8& v, const V8 (&t)[4], const float (&wt)[4]) {
        V8 o;
        if constexpr (std::is_same<T, _Float16>::value) {
            const v4u uv = __builtin_bit_cast(v4u, v), u0 = __builtin_bit_cast(v4u, t[0]), u1 = __builtin_bit_cast(v4u, t[1]),
                      u2 = __builtin_bit_cast(v4u, t[2]), u3 = __builtin_bit_cast(v4u, t[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo_, hi_;
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(lo_) : "v"(u0[i]), "v"(wt[0]), "v"(uv[i]));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hi_) : "v"(u0[i]), "v"(wt[0]), "v"(uv[i]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u1[i]), "v"(wt[1]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u1[i]), "v"(wt[1]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u2[i]), "v"(wt[2]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u2[i]), "v"(wt[2]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u3[i]), "v"(wt[3]));
                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u3[i]), "v"(wt[3]));
                o[2 * i] = (T)lo_;
                o[2 * i + 1] = (T)hi_;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = (T)(fmaf(wt[3], (float)t[3][i], fmaf(wt[2], (float)t[2][i], fmaf(wt[1], (float)t[1][i], fmaf(wt[0], (float)t[0][i], (float)v[i])))));
        }
        return o;
    };
    auto store8 = [&](const V8& o, int vo, int imm) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), outs.r, vo + imm, 0, 0); };

    // prologue, second part: the second tile's rows and the first tile's taps
    if (tile0 + xper < xend) dma_x(tile0 + xper, 1);
    request_taps(tile0);
    int buf = 0;
    // The memory counter retires in issue order, so a load waited for BEHIND a store waits for the store's round trip (the first form
    // of this kernel - taps of tile t requested after the stores of tile t - 1 - took 16 thousand cycles per tile and wave: 400 us; with
    // the stores compiled out 68 us, and 734 MB of pure stores take 135 us: tools/probes/row448.hip).  Hence the order of an iteration:
    // product -> merge into registers (the taps are consumed) -> request the rows of tile t + 2 and the taps of tile t + 1 -> ONLY THEN
    // the 7 stores of tile t: the next iteration's wait for its taps leaves exactly those stores in flight.
#if K12_TRACE
    long long ph[5] = {0, 0, 0, 0, 0}, tlast = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int tile = tile0; tile < xend; tile += xper) {
        // ---- the product: acc[ct] = channels 32 (ct / 2) + 8 g4 + 4 (ct % 2) + {0..3} of pixel lp (the rows of this tile were requested two
        // iterations ago, in front of taps that have been consumed since: they have landed)
        v4f acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4f{0.f, 0.f, 0.f, 0.f};
        const char* xb = smem + X_OFF + (buf * L12_NW + wave) * XW + lp * 64 + ((g4 ^ ((lp >> 1) & 3)) << 4);
        const char* wb = smem + lane * 16;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const Frag xf = *reinterpret_cast<const Frag*>(xb + kc * 1024);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) Mm::mma(*reinterpret_cast<const Frag*>(wb + (kc * NCT + ct) * 1024), xf, acc[ct]);
        }
        K12_T(0);
        // ---- storage type -> exchange with the neighbouring pixel -> + the merge: the tile's 2 NM (+ 1) output pieces, in registers
        V8 oA[NM], oB[NM], oL;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            V8 da, db;
            exchange(pack8(acc[4 * m], acc[4 * m + 1]), pack8(acc[4 * m + 2], acc[4 * m + 3]), da, db);
            oA[m] = merge(da, tA[m], wA);
            oB[m] = merge(db, tB[m], wB);
        }
        if constexpr (LONE) oL = merge(pack8(acc[4 * NM], acc[4 * NM + 1]), tL, wL);
        const int stA = sA, stB = sB, stL = sL;
        __builtin_amdgcn_sched_barrier(0);
        K12_T(1);
        // ---- requests of the following tiles, in front of this tile's stores
        if (tile + 2 * xper < xend) dma_x(tile + 2 * xper, buf);
        if (tile + xper < xend) request_taps(tile + xper);
        __builtin_amdgcn_sched_barrier(0);
        K12_T(2);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            store8(oA[m], stA, 128 * m);
            store8(oB[m], stB, 128 * m);
        }
        if constexpr (LONE) store8(oL, stL, 0);
        __builtin_amdgcn_sched_barrier(0);
        K12_T(3);
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if K12_TRACE
    K12_T(4);
    if (lane == 0)
        for (int i = 0; i < 5; ++i) k12_trace[(blockIdx.x * 8 + wave) * 8 + i] = ph[i];
#endif
}

// The STAGED form (W a multiple of 16 and exactly twice the coarser map's width: the FPN case): a wave's 16 pixels lie in one row, and
// the taps of all of them come from TWO rows x TEN pixels of the coarser map - 8960 bytes, brought into a wave-private LDS area by ten
// LDS-DMA pieces per tile instead of the 28 gathers of 1 KB the form above issues (the vector-memory issue path is what bounds it:
// 39 -> 21 instructions per tile and wave); the merge reads its tap pieces from there (ds_read_b128).  LDS: weights 56 KB + 8 x (4 KB of
// rows + 8960 B of the coarser map) = 158 KB - ONE buffer each: the rows of tile t + 1 are requested behind the product of tile t, its
// piece of the coarser map behind the merge of tile t, both in front of tile t's stores (the memory counter retires in issue order).
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(L12_NW * 64) void lateral_staged_kernel(LatArgs a) {
    using Mm = Mma16<T>;
    using Frag = typename Mm::Frag;
    using V8 = gf_vec<T, 8>;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int KC = CIN / 32, NCT = COUT / 16, NP = COUT / 32, NM = NP / 2;
    constexpr bool LONE = (NP & 1) != 0;
    constexpr int ROWB = COUT * (int)sizeof(T);                              // bytes of a pixel of the coarser map / of the output
    constexpr int LOC = 10, LOROW = LOC * ROWB, LOP = (LOROW + 1023) / 1024;  // staged pixels per row, their bytes, DMA pieces per row
    constexpr int W_BYTES = KC * NCT * 1024, XW = KC * 1024, WV = XW + 2 * LOROW;   // a wave's area: its rows, then two rows of the coarser map
    constexpr int NST = 2 * NM + (LONE ? 1 : 0);                              // stores per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, g4 = lane >> 4, odd = lane & 1;
    const LatRsrc xs = lat_rsrc(a.x, (unsigned)a.P * CIN * (unsigned)sizeof(T));
    const LatRsrc wsr = lat_rsrc(a.wfrag, (unsigned)W_BYTES);
    const LatRsrc los = lat_rsrc(a.lo, (unsigned)a.N * a.h * a.w * COUT * (unsigned)sizeof(T));
    const LatRsrc outs = lat_rsrc(a.out, (unsigned)a.P * COUT * (unsigned)sizeof(T));
    const int nx8 = gridDim.x >= 8 ? 8 : 1;
    const int xcd = nx8 == 8 ? (int)(blockIdx.x & 7) : 0, xslot = nx8 == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int xper = nx8 == 8 ? (int)((gridDim.x + 7 - xcd) >> 3) : (int)gridDim.x;
    const int tq = a.ntiles / nx8, tr = a.ntiles % nx8;
    const int xbeg = xcd * tq + (xcd < tr ? xcd : tr), xend = xbeg + tq + (xcd < tr ? 1 : 0);
    const int tile0 = xbeg + xslot;
    if (tile0 >= xend) return;
    char* const wv = smem + W_BYTES + wave * WV;                               // the wave's area

    auto dma_x = [&](int t) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int q = l >> 2, slot = (l & 3) ^ ((q >> 1) & 3);
        const int off = ((t * L12_TP + wave * 16 + q) * CIN + 8 * slot) * (int)sizeof(T);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) lat_lds_dma(xs, wv + kc * 1024, off, kc * 64);
    };
    // ---- per-tile state.  Wave-uniform: the row of the wave's 16 pixels (image n, row y), its two rows y0 / y1 of the coarser map and
    // xl0, the first staged column.  Per lane: LDS offsets of the four taps of its A / B / lone pieces, their weights, the store offsets.
    int toA[4], toB[4], toL[4];
    float wA[4], wB[4], wL[4];
    int sA = 0, sB = 0, sL = 0;
    auto stage = [&](int tile) {
        const int p0 = tile * L12_TP + wave * 16;                              // (uniform)
        const unsigned qy = (unsigned)p0 / (unsigned)a.W;
        const int xw = p0 - (int)qy * a.W, n = (int)(qy / (unsigned)a.H), y = (int)qy - n * a.H;
        const float fy = a.ry * y;
        const int y0 = (int)fy, y1 = y0 + (y0 < a.h - 1);
        const float wy1 = fy - y0, wy0 = 1.f - wy1;
        const int xl0 = (int)(a.rx * xw);
        // the two rows, LOC pixels from column xl0: contiguous bytes of the map (columns behind the row's end belong to the next row or
        // are out of range: no tap reads them); tiles behind the last pixel: image n is out of range, zeros
        int l = lane;
        asm volatile("" : "+v"(l));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rbase = ((n * a.h + (r ? y1 : y0)) * a.w + xl0) * ROWB;
#pragma unroll
            for (int i = 0; i < LOP; ++i) {
                // (the last piece of a row is partial: its lanes behind the row's end are masked off - an out-of-range lane would write
                // zeros over the start of the next row's area)
                const int b = i * 1024 + l * 16;
                if (b < LOROW) lat_lds_dma(los, wv + XW + r * LOROW + i * 1024, p0 < a.P ? rbase + b : 0x7FFFFFF0, 0);
            }
        }
        int el = lane;
        asm volatile("" : "+v"(el));
        const int xa = xw + (el & 14), pa = p0 + (el & 14);
        const int cb = 64 * (el & 1) + 16 * (el >> 4);
        auto taps = [&](int x, int colbyte, int (&to)[4], float (&wt)[4]) {
            const float fx = a.rx * x;
            const int x0 = (int)fx, x1 = x0 + (x0 < a.w - 1);
            const float wx1 = fx - x0, wx0 = 1.f - wx1;
            to[0] = XW + (x0 - xl0) * ROWB + colbyte;
            to[1] = XW + (x1 - xl0) * ROWB + colbyte;
            to[2] = to[0] + LOROW;
            to[3] = to[1] + LOROW;
            wt[0] = wy0 * wx0; wt[1] = wy0 * wx1; wt[2] = wy1 * wx0; wt[3] = wy1 * wx1;
        };
        taps(xa, cb, toA, wA);
        taps(xa + 1, cb, toB, wB);
        if constexpr (LONE) taps(xa + (el & 1), 128 * NM + 16 * (el >> 4), toL, wL);
        sA = pa < a.P ? pa * ROWB + cb : 0x7FFFFFF0;
        sB = pa + 1 < a.P ? (pa + 1) * ROWB + cb : 0x7FFFFFF0;
        const int pl = pa + (el & 1);
        sL = pl < a.P ? pl * ROWB + 128 * NM + 16 * (el >> 4) : 0x7FFFFFF0;
    };
    auto pack8 = [](const v4f& lo_, const v4f& hi_) {
        return V8{(T)lo_[0], (T)lo_[1], (T)lo_[2], (T)lo_[3], (T)hi_[0], (T)hi_[1], (T)hi_[2], (T)hi_[3]};
    };
    auto exchange = [&](const V8& p0, const V8& p1, V8& da, V8& db) {
        const v4u u0 = __builtin_bit_cast(v4u, p0), u1 = __builtin_bit_cast(v4u, p1);
        v4u ua, ub;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned send = odd ? u0[i] : u1[i];
            const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
            ua[i] = odd ? recv : u0[i];
            ub[i] = odd ? u1[i] : recv;
        }
        da = __builtin_bit_cast(V8, ua);
        db = __builtin_bit_cast(V8, ub);
    };
    // v + sum_k wt[k] * tap k, the taps read from the wave's staged rows at to[k] + imm
    auto merge = [&](const V8& v, const int (&to)[4], int imm, const float (&wt)[4]) {
        v4u u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const v4u*>(wv + to[k] + imm);
        const v4u uv = __builtin_bit_cast(v4u, v);
        V8 o;
        if constexpr (std::is_same<T, _Float16>::value) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo_, hi_;
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(lo_) : "v"(u[0][i]), "v"(wt[0]), "v"(uv[i]));
                asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(hi_) : "v"(u[0][i]), "v"(wt[0]), "v"(uv[i]));
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo_) : "v"(u[k][i]), "v"(wt[k]));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi_) : "v"(u[k][i]), "v"(wt[k]));
                }
                o[2 * i] = (T)lo_;
                o[2 * i + 1] = (T)hi_;
            }
        } else {
            // bf16: a value is the high half of its fp32 form (a shift / a mask), same sums in the same order
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo_ = __uint_as_float(uv[i] << 16), hi_ = __uint_as_float(uv[i] & 0xFFFF0000u);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lo_ = fmaf(__uint_as_float(u[k][i] << 16), wt[k], lo_);
                    hi_ = fmaf(__uint_as_float(u[k][i] & 0xFFFF0000u), wt[k], hi_);
                }
                o[2 * i] = (T)lo_;
                o[2 * i + 1] = (T)hi_;
            }
        }
        return o;
    };
    auto store8 = [&](const V8& o, int vo, int imm) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), outs.r, vo + imm, 0, 0); };

    // prologue: the weights (once per workgroup), the first tile's rows and its piece of the coarser map
#pragma unroll
    for (int i = 0; i < (KC * NCT + L12_NW - 1) / L12_NW; ++i) {
        const int f = wave + L12_NW * i;
        if (f < KC * NCT) lat_lds_dma(wsr, smem + f * 1024, lane * 16, f * 1024);
    }
    dma_x(tile0);
    stage(tile0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool first = true;
    for (int tile = tile0; tile < xend; tile += xper) {
        const bool has_next = tile + xper < xend;
        // ---- the product.  In flight, oldest first: rows(t) | map(t) | stores(t - 1)
        if (!first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOP + NST) : "memory");
        v4f acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4f{0.f, 0.f, 0.f, 0.f};
        const char* xb = wv + lp * 64 + ((g4 ^ ((lp >> 1) & 3)) << 4);
        const char* wb = smem + lane * 16;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const Frag xf = *reinterpret_cast<const Frag*>(xb + kc * 1024);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) Mm::mma(*reinterpret_cast<const Frag*>(wb + (kc * NCT + ct) * 1024), xf, acc[ct]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the rows of the next tile take the place of this one's (every fragment read above has returned: the MFMAs consumed them)
        if (has_next) dma_x(tile + xper);
        // ---- the merge.  In flight: map(t) | stores(t - 1) | rows(t + 1)
        if (!first) {
            if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST + KC) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        }
        V8 oA[NM], oB[NM], oL;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            V8 da, db;
            exchange(pack8(acc[4 * m], acc[4 * m + 1]), pack8(acc[4 * m + 2], acc[4 * m + 3]), da, db);
            oA[m] = merge(da, toA, 128 * m, wA);
            oB[m] = merge(db, toB, 128 * m, wB);
        }
        if constexpr (LONE) oL = merge(pack8(acc[4 * NM], acc[4 * NM + 1]), toL, 0, wL);
        const int stA = sA, stB = sB, stL = sL;
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's piece of the coarser map takes the place of this one's (its tap reads have returned), then this tile's stores
        if (has_next) stage(tile + xper);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            store8(oA[m], stA, 128 * m);
            store8(oB[m], stB, 128 * m);
        }
        if constexpr (LONE) store8(oL, stL, 0);
        __builtin_amdgcn_sched_barrier(0);
        first = false;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int CIN, int COUT>
int lat_launch(LatArgs a, hipStream_t st) {
    constexpr int LDS = (CIN / 32) * (COUT / 16) * 1024 + 2 * L12_NW * (CIN / 32) * 1024;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr{0};
    if (gf_first_use_on_device(attr))
        (void)hipFuncSetAttribute((const void*)lateral_kernel<T, CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    lateral_kernel<T, CIN, COUT><<<a.ntiles < 256 ? a.ntiles : 256, L12_NW * 64, LDS, st>>>(a);
    return 0;
}

template <typename T, int CIN, int COUT>
int lat_launch_staged(LatArgs a, hipStream_t st) {
    constexpr int LDS = (CIN / 32) * (COUT / 16) * 1024 + L12_NW * ((CIN / 32) * 1024 + 2 * 10 * COUT * (int)sizeof(T));
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static std::atomic<uint64_t> attr{0};
    if (gf_first_use_on_device(attr))
        (void)hipFuncSetAttribute((const void*)lateral_staged_kernel<T, CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    lateral_staged_kernel<T, CIN, COUT><<<a.ntiles < 256 ? a.ntiles : 256, L12_NW * 64, LDS, st>>>(a);
    return 0;
}

}   // namespace

#if K12_TRACE
extern "C" int gf_debug_k12_trace(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(k12_trace), sizeof(k12_trace)) == hipSuccess ? 0 : -1;
}
#endif

// 1 if gf_lateral_upsample_add_nhwc has a kernel for these channel counts
extern "C" int gf_lateral_supported(int cin, int cout) { return cin == 128 && cout == 224; }

// out[n,y,x,:] = W . x[n,y,x,:] + bilinear(lo -> H x W, align_corners=True)[n,y,x,:]; x [N,H,W,cin], lo [N,h,w,cout], out [N,H,W,cout]
// (GF_F16; GF_BF16 where the staged form applies: W % 16 == 0 and W == 2 w; channels-last); wfrag = fused.py:pack_lateral_frags(w [cout, cin]); W must be even
extern "C" int gf_lateral_upsample_add_nhwc(const void* x, const void* wfrag, const void* lo, void* out, int N, int h, int wl, int H, int W,
                                            int cin, int cout, int dtype, void* stream) {
    GF_CHECK_ARG(x && wfrag && lo && out, "null pointer");
    GF_CHECK_ARG(N > 0 && h > 0 && wl > 0 && H > 0 && W > 0, "empty problem");
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "built for 16-bit maps (the inference backbone)");
    const bool staged = W % 16 == 0 && W == 2 * wl;
    GF_CHECK_ARG(dtype == GF_F16 || staged, "bf16 runs the staged form only (W a multiple of 16 and twice the coarser map's width): without v_fma_mix the "
                                            "gather form's merge needs more registers than two waves per SIMD leave; use gf_conv1x1_upsample_add_nhwc");
    GF_CHECK_ARG(gf_lateral_supported(cin, cout), "no kernel for these channel counts (see gf_lateral_supported)");
    GF_CHECK_ARG(W % 2 == 0, "W must be even (a pixel pair shares its row)");
    GF_CHECK_ARG((long)N * H * W * 224 * 2 < 0x7FFFFFF0l, "maps of 2 GiB or more are not supported (32-bit buffer offsets)");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)lo % 16 == 0 && (uintptr_t)wfrag % 16 == 0,
                 "tensors must be 16-byte aligned");
    LatArgs a{};
    a.x = x; a.wfrag = wfrag; a.lo = lo; a.out = out; a.N = N; a.H = H; a.W = W; a.h = h; a.w = wl; a.P = N * H * W;
    a.ntiles = (a.P + L12_TP - 1) / L12_TP;
    a.ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    a.rx = W > 1 ? (float)(wl - 1) / (float)(W - 1) : 0.f;
    hipStream_t st = (hipStream_t)stream;
    // same tag and declared work as the K3 form it replaces: algorithmic bytes (x + out + the coarser map + weights once)
    void* pt = gf_prof_begin("k3_upadd", st, 2.0 * ((double)a.P * (cin + cout) + (double)N * h * wl * cout + (double)cin * cout));
    // the staged form where a wave's 16 pixels share a row and 10 columns of the coarser map cover their taps (scale factor 2)
    if (staged) {
        if (dtype == GF_F16) lat_launch_staged<_Float16, 128, 224>(a, st);
        else lat_launch_staged<gf_bf16, 128, 224>(a, st);
    } else {
        lat_launch<_Float16, 128, 224>(a, st);
    }
    gf_prof_end("k3_upadd", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
