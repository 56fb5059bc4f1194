// K10 (training): weight gradient of the backbone's 3x3 / stride 1 / pad 1 convolutions on channels-last 16-bit maps
// (model/loftr_src/loftr/backbone/resnet_fpn.py:9-40,60-83 under autograd):
//     dW[co][ci][ky][kx] = sum over images and pixels of dY[n, y, x, co] * X[n, y + ky - 1, x + kx - 1, ci]          (fp32)
// An implicit GEMM whose contraction runs over PIXELS, so both MFMA operands are transposes of channels-last rows: a row segment of
// 32 pixels sits in LDS pixel-major ([pixel][channels], padded stride) and the fragments are read with ds_read_b64_tr_b16 - the
// scheme of the linear layers' weight gradient (k_train.hip: wgrad_kernel), extended by the nine taps:
//  * workgroup = 128 output channels x 64 input channels x 9 taps; four waves, one per SIMD: wave (wm, wn) owns 64 x 32 x 9 =
//    2 x 9 accumulator tiles (288 registers of the 512 a single wave per SIMD may hold).  Per 16 pixels a wave reads 2 fragments
//    of dY and 9 of X (one per tap: the tap's shift is a pixel offset of the read) for 18 MFMAs.
//  * a workgroup walks DOWN a 32-pixel column strip of an image: the three X rows a dY row needs live in a ring of four row images
//    (34 pixels: the strip and its two neighbours), so every X row is fetched once per strip and every dY row once; the next row of
//    each is in flight in registers while the current one is multiplied; one barrier per row.
//  * the strips' rows (image, segment, row) are dealt to `chunks` workgroups per output block in contiguous runs; every workgroup
//    leaves its partial sums [tap][co][ci] (coalesced), a second kernel adds the chunks in order and writes dW[co][ci][3][3]:
//    no atomics, the gradient is bit-reproducible.
// Channels: the maps may carry padding channels (the 196-channel level is stored 224 wide): cx / cy = stored widths (multiples of 8),
// cin / cout = the real ones; blocks beyond the stored width read zeros, waves beyond the real width do nothing.
#include "gf_common.h"

namespace {

constexpr int RSY = 320, RSX = 192;        // LDS strides per pixel: 128 co x 2 B + 64 / 64 ci x 2 B + 64 (both = 16 dwords mod 64: the four
                                           // pixel rows of a transposing read and the two 16-channel halves tile the 64 banks)
constexpr int DYIMG = 32 * RSY, XIMG = 34 * RSX;

struct CwArgs {
    const void* x;
    const void* dy;
    int N, H, W, cx, cy, cin, cout, nseg, chunks, rpc, coP, ciP;
    long total;          // N * nseg * H strip rows
    float* part;         // [chunks][9][coP][ciP]
    float* dw;           // [cout][cin][3][3]
};

template <typename T>
__device__ __forceinline__ gf_vec<T, 8> tr_frag(const char* p, int rs) {
    typedef __attribute__((address_space(3))) gf_v4s* LP;
    const gf_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)p);
    const gf_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LP)(p + 4 * rs));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(gf_vec<T, 8>, both);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad(CwArgs a) {
    using M = Mma32<T>;
    using Frag = typename M::Frag;
    __shared__ __attribute__((aligned(16))) char dys[2 * DYIMG];
    __shared__ __attribute__((aligned(16))) char xs[4 * XIMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int ib = blockIdx.x, cb = blockIdx.y, chunk = blockIdx.z;
    const bool active = cb * 128 + wm * 64 < a.cout && ib * 64 + wn * 32 < a.cin;
    const T* xb = (const T*)a.x;
    const T* yb = (const T*)a.dy;
    v16f acc[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    // the lane's fragment offsets: pixel 8 (G >> 1) + q of a 16-pixel step, channels 16 (G & 1) + 4 p .. of a 32-channel block
    const int G = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;
    const int offa = (8 * (G >> 1) + q) * RSY + (wm * 64 + 16 * (G & 1) + 4 * p4) * 2;
    const int offb = (8 * (G >> 1) + q) * RSX + (wn * 32 + 16 * (G & 1) + 4 * p4) * 2;
    const v4u zero{0u, 0u, 0u, 0u};
    // row pieces: dY 32 pixels x 16 pieces (two per thread), X 34 pixels x 8 pieces (one per thread, a second for threads 0..15)
    auto load_dy = [&](int n, int y, int x0, int k) {
        const int e = k * 256 + tid, px = e >> 4, c = cb * 128 + (e & 15) * 8;
        return (x0 + px < a.W && c < a.cy) ? *reinterpret_cast<const v4u*>(yb + (((size_t)n * a.H + y) * a.W + x0 + px) * a.cy + c) : zero;
    };
    auto put_dy = [&](char* img, int k, const v4u& v) {
        const int e = k * 256 + tid;
        *reinterpret_cast<v4u*>(img + (e >> 4) * RSY + (e & 15) * 16) = v;
    };
    auto load_x = [&](int n, int yy, int x0, int k) {
        const int e = k * 256 + tid, px = e >> 3, c = ib * 64 + (e & 7) * 8, xx = x0 - 1 + px;
        return (e < 272 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && c < a.cx)
                   ? *reinterpret_cast<const v4u*>(xb + (((size_t)n * a.H + yy) * a.W + xx) * a.cx + c) : zero;
    };
    auto put_x = [&](char* img, int k, const v4u& v) {
        const int e = k * 256 + tid;
        if (e < 272) *reinterpret_cast<v4u*>(img + (e >> 3) * RSX + (e & 7) * 16) = v;
    };
    long R = (long)chunk * a.rpc;
    const long Rend = R + a.rpc < a.total ? R + a.rpc : a.total;
    while (R < Rend) {
        const int strip = (int)(R / a.H);
        int y = (int)(R - (long)strip * a.H);
        const int n = strip / a.nseg, x0 = (strip - n * a.nseg) * 32;
        const int yend = (long)(a.H - y) < Rend - R ? a.H : y + (int)(Rend - R);
        R += yend - y;
        // prime the strip: X rows y - 1, y, y + 1 and dY row y
#pragma unroll
        for (int d = -1; d <= 1; ++d)
#pragma unroll
            for (int k = 0; k < 2; ++k) put_x(xs + ((y + d) & 3) * XIMG, k, load_x(n, y + d, x0, k));
#pragma unroll
        for (int k = 0; k < 2; ++k) put_dy(dys + (y & 1) * DYIMG, k, load_dy(n, y, x0, k));
        __syncthreads();
        for (; y < yend; ++y) {
            const bool more = y + 1 < yend;
            v4u rx[2], ry[2];
            if (more) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    rx[k] = load_x(n, y + 2, x0, k);
                    ry[k] = load_dy(n, y + 1, x0, k);
                }
            }
            if (active) {
                const char* dimg = dys + (y & 1) * DYIMG + offa;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const Frag a0 = tr_frag<T>(dimg + s2 * 16 * RSY, RSY), a1 = tr_frag<T>(dimg + s2 * 16 * RSY + 64, RSY);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const char* ximg = xs + ((y + ky - 1) & 3) * XIMG + offb + s2 * 16 * RSX;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const Frag b = tr_frag<T>(ximg + kx * RSX, RSX);
                            M::mma(a0, b, acc[0][ky * 3 + kx]);
                            M::mma(a1, b, acc[1][ky * 3 + kx]);
                        }
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    put_x(xs + ((y + 2) & 3) * XIMG, k, rx[k]);
                    put_dy(dys + ((y + 1) & 1) * DYIMG, k, ry[k]);
                }
            }
            __syncthreads();
        }
    }
    if (!active) return;
    // accumulator: row = output channel, lane = input channel: 128-byte runs along ci
    const int h2 = lane >> 5, lr = lane & 31;
    float* dst = a.part + ((size_t)chunk * 9 * a.coP + cb * 128 + wm * 64) * a.ciP + ib * 64 + wn * 32 + lr;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((size_t)t * a.coP + i * 32 + gf_acc_row(r, h2)) * a.ciP] = acc[i][t][r];
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce(CwArgs a) {
    const size_t n = (size_t)9 * a.cout * a.cin, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int ci = (int)(e % a.cin), co = (int)((e / a.cin) % a.cout), t = (int)(e / ((size_t)a.cin * a.cout));
    const size_t plane = (size_t)9 * a.coP * a.ciP;
    const float* p = a.part + ((size_t)t * a.coP + co) * a.ciP + ci;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= a.chunks; c += 8) {                   // eight partials in flight, added in chunk order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(c + k) * plane];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; c < a.chunks; ++c) s += p[(size_t)c * plane];
    a.dw[((size_t)co * a.cin + ci) * 9 + t] = s;
}

#ifndef CW_WG_TARGET
#define CW_WG_TARGET 256
#endif
void plan(CwArgs& a) {
    a.nseg = (a.W + 31) / 32;
    a.total = (long)a.N * a.nseg * a.H;
    a.coP = (a.cout + 127) / 128 * 128;
    a.ciP = (a.cin + 63) / 64 * 64;
    const long blocks = (long)(a.coP / 128) * (a.ciP / 64);
    long chunks = (CW_WG_TARGET + blocks - 1) / blocks;   // one workgroup per CU (its registers allow no second): 256 measured 3-10 % faster than 512 / 768 / 1024
    const long most = (a.total + 7) / 8;                  // at least eight rows per chunk
    if (chunks > most) chunks = most;
    if (chunks < 1) chunks = 1;
    a.rpc = (int)((a.total + chunks - 1) / chunks);
    a.chunks = (int)((a.total + a.rpc - 1) / a.rpc);
}

}  // namespace

extern "C" size_t gf_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int cin, int cout) {
    if (N <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return 0;
    CwArgs a{};
    a.N = N; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    plan(a);
    return gf_align_up(sizeof(float) * (size_t)a.chunks * 9 * a.coP * a.ciP, 256);
}

extern "C" int gf_conv3x3_wgrad_nhwc(const void* x, const void* dy, int dtype, int N, int H, int W, int cx, int cy, int cin, int cout, float* dw,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    GF_CHECK_ARG(dtype == GF_F16 || dtype == GF_BF16, "16-bit maps only");
    GF_CHECK_ARG(N > 0 && H > 0 && W > 0 && cin > 0 && cout > 0 && cin <= cx && cout <= cy, "sizes");
    GF_CHECK_ARG(cx % 8 == 0 && cy % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "stored widths must be multiples of 8 channels, 16-byte aligned maps");
    GF_CHECK_ARG(x && dy && dw, "null pointer");
    CwArgs a{};
    a.x = x; a.dy = dy; a.N = N; a.H = H; a.W = W; a.cx = cx; a.cy = cy; a.cin = cin; a.cout = cout; a.dw = dw;
    plan(a);
    if (workspace == nullptr || workspace_bytes < gf_conv3x3_wgrad_workspace_bytes(N, H, W, cin, cout)) {
        gf_set_error("gf_conv3x3_wgrad_nhwc: workspace too small");
        return GF_ERR_WORKSPACE;
    }
    a.part = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    void* tok = gf_prof_begin("conv3x3_wgrad", st, 18.0 * N * H * W * (double)cin * cout);
    const dim3 grid(a.ciP / 64, a.coP / 128, a.chunks);
    if (dtype == GF_F16) conv_wgrad<_Float16><<<grid, 256, 0, st>>>(a);
    else conv_wgrad<gf_bf16><<<grid, 256, 0, st>>>(a);
    conv_wgrad_reduce<<<(unsigned)(((size_t)9 * cout * cin + 255) / 256), 256, 0, st>>>(a);
    gf_prof_end("conv3x3_wgrad", tok, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}
