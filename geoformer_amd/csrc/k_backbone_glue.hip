// Backbone glue: the element-wise passes BETWEEN the MIOpen convolutions of the
// fp16 inference backbone, fused so that each activation map is read and written once.
//
//   gf_bias_act_nhwc      y = act(x + bias[c] (+ residual))     folded-BatchNorm shift + ReLU / LeakyReLU and the
//                                                               BasicBlock shortcut add
//                                                               (backbone/resnet_fpn.py:20-40, :92-95, :106-115)
//   gf_upsample_add_nhwc  y = hi + bilinear_x2(lo)              the FPN top-down merge, align_corners=True
//                                                               (backbone/resnet_fpn.py:104-105, :110-111)
//
// Tensors are channels-last ([N,H,W,C] in memory).  Both kernels are pure HBM streams: 16-byte (C % 8 == 0)
// or 8-byte (C % 4 == 0) vectors per lane along the channel axis; the low-resolution source of the
// upsample is 4x smaller than the output and is served by L2.
#include "gf_common.h"

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

template <typename T, int V> struct VecOf;
template <> struct VecOf<_Float16, 8> { typedef v8h type; };
template <> struct VecOf<_Float16, 4> { typedef v4h type; };
template <> struct VecOf<gf_bf16, 8> { typedef v8b type; };
template <> struct VecOf<gf_bf16, 4> { typedef v4b type; };
template <> struct VecOf<float, 4> { typedef v4f type; };

struct BaArgs {
    const void* x;
    const float* bias;
    const void* res;
    void* out;
    long nvec;          // P * C / V
    int C, act;
    float slope;
};

template <typename T, int V>
__global__ __launch_bounds__(256) void bias_act(BaArgs a) {
    typedef typename VecOf<T, V>::type Vec;
    const int cv = a.C / V;
    const long stride = (long)gridDim.x * 256;
    const int smod = (int)(stride % cv);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int c = (int)(i % cv);                                   // channel-vector of element i, advanced incrementally
    const Vec* x = reinterpret_cast<const Vec*>(a.x);
    const Vec* res = reinterpret_cast<const Vec*>(a.res);
    Vec* out = reinterpret_cast<Vec*>(a.out);
    for (; i < a.nvec; i += stride) {
        Vec v = __builtin_nontemporal_load(x + i);
        Vec r;
        if (res) r = __builtin_nontemporal_load(res + i);
        float bv[V];
        if (a.bias) {
#pragma unroll
            for (int k = 0; k < V; k += 4) {
                const v4f b4 = *reinterpret_cast<const v4f*>(a.bias + c * V + k);
                bv[k] = b4.x; bv[k + 1] = b4.y; bv[k + 2] = b4.z; bv[k + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) bv[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float f = (float)v[k] + bv[k];
            if (res) f += (float)r[k];
            if (a.act == ACT_RELU) f = fmaxf(f, 0.f);
            else if (a.act == ACT_LEAKY) f = f > 0.f ? f : f * a.slope;
            v[k] = (T)f;
        }
        out[i] = v;
        c += smod;
        if (c >= cv) c -= cv;
    }
}

struct UaArgs {
    const void* lo;
    const void* hi;
    void* out;
    int N, h, w, H, W, C;
    float ry, rx;       // (h-1)/(H-1), (w-1)/(W-1)
};

// grid: x = 256-thread chunks of one output row (W * C/V vectors), y = output row, z = sample
template <typename T, int V>
__global__ __launch_bounds__(256) void upsample_add(UaArgs a) {
    typedef typename VecOf<T, V>::type Vec;
    const unsigned cv = a.C / V;
    const unsigned xc = blockIdx.x * 256 + threadIdx.x;
    if (xc >= (unsigned)a.W * cv) return;
    const unsigned x = xc / cv, c = xc - x * cv;
    const int y = blockIdx.y, n = blockIdx.z;
    const float fy = a.ry * y, fx = a.rx * x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < a.h - 1), x1 = x0 + (x0 < a.w - 1);
    const float wy1 = fy - y0, wx1 = fx - x0, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    const Vec* lo = reinterpret_cast<const Vec*>(a.lo) + (size_t)n * a.h * a.w * cv + c;
    const Vec v00 = lo[((size_t)y0 * a.w + x0) * cv], v01 = lo[((size_t)y0 * a.w + x1) * cv];
    const Vec v10 = lo[((size_t)y1 * a.w + x0) * cv], v11 = lo[((size_t)y1 * a.w + x1) * cv];
    const size_t i = ((size_t)n * a.H + y) * a.W * cv + xc;
    Vec o = __builtin_nontemporal_load(reinterpret_cast<const Vec*>(a.hi) + i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float up = wy0 * (wx0 * (float)v00[k] + wx1 * (float)v01[k]) + wy1 * (wx0 * (float)v10[k] + wx1 * (float)v11[k]);
        o[k] = (T)((float)o[k] + up);
    }
    reinterpret_cast<Vec*>(a.out)[i] = o;
}

// Backward of the bilinear upsampling (align_corners=True) of the FPN merge, for the training step: dlo[n, y, x, :] = sum over the output pixels
// (Y, X) whose interpolation touches (y, x) of their weight times dhi[n, Y, X, :].  A GATHER - one thread per low-resolution pixel and channel
// vector walks the <= 5 x 5 output pixels whose source position lies within one pixel of it, with the forward's own index arithmetic (ry * Y in
// float, the clamped second tap) - instead of the scatter with atomics the library runs (3.3 ms per call at 16 x 196 x 320 x 320).
struct UbArgs {
    const void* dhi;
    void* dlo;
    int N, h, w, H, W, C;
    float ry, rx;       // (h-1)/(H-1), (w-1)/(W-1)
};

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample_backward(UbArgs a) {
    typedef typename VecOf<T, V>::type Vec;
    const unsigned cv = a.C / V;
    const unsigned xc = blockIdx.x * 256 + threadIdx.x;
    if (xc >= (unsigned)a.w * cv) return;
    const int x = (int)(xc / cv), c = (int)(xc - (unsigned)x * cv);
    const int y = blockIdx.y, n = blockIdx.z;
    // output rows / columns whose source coordinate may fall into (y - 1, y + 1) / (x - 1, x + 1): one more on either side than the exact bounds,
    // the weight test below decides
    const int Y0 = a.ry > 0.f ? max(0, (int)floorf((y - 1) / a.ry) - 1) : 0, Y1 = a.ry > 0.f ? min(a.H - 1, (int)ceilf((y + 1) / a.ry) + 1) : a.H - 1;
    const int X0 = a.rx > 0.f ? max(0, (int)floorf((x - 1) / a.rx) - 1) : 0, X1 = a.rx > 0.f ? min(a.W - 1, (int)ceilf((x + 1) / a.rx) + 1) : a.W - 1;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    const Vec* hi = reinterpret_cast<const Vec*>(a.dhi) + (size_t)n * a.H * a.W * cv + c;
    for (int Y = Y0; Y <= Y1; ++Y) {
        const float fy = a.ry * Y;
        const int y0 = (int)fy, y1 = y0 + (y0 < a.h - 1);
        const float wy1 = fy - y0, wy = (y0 == y ? 1.f - wy1 : 0.f) + (y1 == y ? wy1 : 0.f);
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            const float fx = a.rx * X;
            const int x0 = (int)fx, x1 = x0 + (x0 < a.w - 1);
            const float wx1 = fx - x0, wx = (x0 == x ? 1.f - wx1 : 0.f) + (x1 == x ? wx1 : 0.f);
            if (wx == 0.f) continue;
            const Vec g = hi[((size_t)Y * a.W + X) * cv];
            const float wgt = wy * wx;
#pragma unroll
            for (int k = 0; k < V; ++k) acc[k] += wgt * (float)g[k];
        }
    }
    Vec o;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (T)acc[k];
    reinterpret_cast<Vec*>(a.dlo)[((size_t)n * a.h + y) * a.w * cv + xc] = o;
}

// ---------------------------------------------------------------------------------------------
// Stem: 7x7 stride-2 pad-3 convolution of the 1-channel image to 128 channels + shift + ReLU, written
// NHWC.  An implicit GEMM on the matrix cores: K = (ky, kx) padded 7x7 -> 8x8, so that the k-group of one
// v_mfma_f32_32x32x16_f16 is two kernel rows and a lane's 8 consecutive k are 8 consecutive image
// pixels (one 16-byte LDS fragment, 4-byte aligned because the stride is 2).  A wave owns 32 consecutive
// output pixels of one row x all 128 channels; the weights live in registers as 16 MFMA fragments for
// the whole (persistent) kernel; the product is transposed (lane = pixel) so shift + ReLU are lane-local,
// and the result goes through an LDS slab to 16-byte row-contiguous stores.  HBM-bound on the output.
// ---------------------------------------------------------------------------------------------
struct StArgs {
    const void* img;        // [N][H][W]
    const float* w;         // [128][7][7], BatchNorm scale folded in
    const float* shift;     // [128]
    void* out;              // [N][Ho][Wo][128] of the compute type (fp16 / bf16)
    int N, H, W, Ho, Wo;
    int tiles_x, tiles_y;   // 32-pixel x 4-row output tiles
};

template <typename TI, typename T>
__global__ __launch_bounds__(256, 2) void stem_conv(StArgs a) {
    using Frag = typename Mma32<T>::Frag;
    typedef T v4t __attribute__((ext_vector_type(4)));
    constexpr int TW = 72, TH = 14, RS = 272;
    __shared__ __attribute__((aligned(16))) T tile[TH * TW];
    __shared__ __attribute__((aligned(16))) char slab[4 * 32 * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, lr = lane & 31;
    Frag wf[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ky = 2 * g + h;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                wf[nb][g][j] = (ky < 7 && j < 7) ? (T)a.w[(nb * 32 + lr) * 49 + ky * 7 + j] : (T)0.f;
        }
    float sh[4][16];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[nb][r] = a.shift[nb * 32 + gf_acc_row(r, h)];

    const long ntiles = (long)a.N * a.tiles_y * a.tiles_x;
    // the image pixels of a tile: requested into registers one tile AHEAD, in front of the previous tile's output stores (the memory
    // counter retires in issue order: requested behind them, every tile waited for its predecessor's stores to come back)
    constexpr int NPRE = (TH * TW + 255) / 256;
    T pre[NPRE];
    auto request = [&](long t) {
        const int tx = (int)(t % a.tiles_x);
        const long q = t / a.tiles_x;
        const int ty = (int)(q % a.tiles_y), n = (int)(q / a.tiles_y);
        const TI* img = (const TI*)a.img + (long)n * a.H * a.W;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + 256 * i, r = e / TW, c = e % TW;
            const int y = 8 * ty - 3 + r, x = 64 * tx - 3 + c;
            pre[i] = (e < TH * TW && y >= 0 && y < a.H && x >= 0 && x < a.W) ? (T)gf_to_float(img[(long)y * a.W + x]) : (T)0.f;
        }
    };
    if ((long)blockIdx.x < ntiles) request(blockIdx.x);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % a.tiles_x);
        const long q = t / a.tiles_x;
        const int ty = (int)(q % a.tiles_y), n = (int)(q / a.tiles_y);
        const int ox0 = tx * 32, oy0 = ty * 4;
        __syncthreads();                                   // previous tile's fragments are consumed
#pragma unroll
        for (int i = 0; i < NPRE; ++i)
            if (tid + 256 * i < TH * TW) tile[tid + 256 * i] = pre[i];
        __syncthreads();
        if (t + gridDim.x < ntiles) request(t + gridDim.x);
        v16f acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(tile + (2 * wave + 2 * g + h) * TW + 2 * lr);
            v4u raw{src[0], src[1], src[2], src[3]};
            const Frag tf = __builtin_bit_cast(Frag, raw);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) Mma32<T>::mma(wf[nb][g], tf, acc[nb]);
        }
        // shift + ReLU, lane = pixel ox0 + lr of row oy0 + wave; channels nb*32 + acc_row(r, h)
        char* ot = slab + wave * 32 * RS;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c = nb * 32 + 8 * r4 + 4 * h;
                v4t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (T)fmaxf(acc[nb][4 * r4 + k] + sh[nb][4 * r4 + k], 0.f);
                *reinterpret_cast<v4t*>(ot + lr * RS + c * 2) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int oy = oy0 + wave;
        if (oy < a.Ho) {
            T* orow = (T*)a.out + (((long)n * a.Ho + oy) * a.Wo + ox0) * 128;
            const int prow = lane >> 4, pch = lane & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int px = it * 4 + prow;
                if (ox0 + px < a.Wo)
                    *reinterpret_cast<Frag*>(orow + px * 128 + pch * 8) = *reinterpret_cast<const Frag*>(ot + px * RS + pch * 16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

inline int glue_blocks(long nvec) {
    const long b = (nvec + 255) / 256;
    return (int)(b < 256 * 16 ? b : 256 * 16);          // 16 workgroups per CU, grid-stride beyond that
}

}   // namespace

extern "C" int gf_bias_act_nhwc(const void* x, const float* bias, const void* residual, void* out, long pixels, int C,
                                int act, float slope, int dtype, void* stream) {
    GF_CHECK_ARG(x && out, "null pointer");
    GF_CHECK_ARG(pixels > 0 && C > 0, "empty problem");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(act >= ACT_NONE && act <= ACT_LEAKY, "unknown activation");
    GF_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    GF_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)residual % 16 == 0 && (uintptr_t)bias % 16 == 0,
                 "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    BaArgs a{x, bias, residual, out, 0, C, act, slope};
    void* pt = gf_prof_begin("bias_act", st, (double)pixels * C * (dtype == GF_F32 ? 4 : 2) * (residual ? 3.0 : 2.0));
    if (dtype == GF_F16 && C % 8 == 0) {
        a.nvec = pixels * C / 8;
        bias_act<_Float16, 8><<<glue_blocks(a.nvec), 256, 0, st>>>(a);
    } else if (dtype == GF_F16) {
        a.nvec = pixels * C / 4;
        bias_act<_Float16, 4><<<glue_blocks(a.nvec), 256, 0, st>>>(a);
    } else if (dtype == GF_BF16 && C % 8 == 0) {
        a.nvec = pixels * C / 8;
        bias_act<gf_bf16, 8><<<glue_blocks(a.nvec), 256, 0, st>>>(a);
    } else if (dtype == GF_BF16) {
        a.nvec = pixels * C / 4;
        bias_act<gf_bf16, 4><<<glue_blocks(a.nvec), 256, 0, st>>>(a);
    } else {
        a.nvec = pixels * C / 4;
        bias_act<float, 4><<<glue_blocks(a.nvec), 256, 0, st>>>(a);
    }
    gf_prof_end("bias_act", pt, st);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_upsample_add_nhwc(const void* lo, const void* hi, void* out, int N, int h, int w, int H, int W, int C,
                                    int dtype, void* stream) {
    GF_CHECK_ARG(lo && hi && out, "null pointer");
    GF_CHECK_ARG(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "empty problem");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    GF_CHECK_ARG((uintptr_t)lo % 16 == 0 && (uintptr_t)hi % 16 == 0 && (uintptr_t)out % 16 == 0, "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    UaArgs a{lo, hi, out, N, h, w, H, W, C, H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f,
             W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f};
    GF_CHECK_ARG(H <= 65535 && N <= 65535, "H and N must fit the launch grid");
    const int V = (dtype != GF_F32 && C % 8 == 0) ? 8 : 4;
    const dim3 grid((unsigned)((W * (C / V) + 255) / 256), (unsigned)H, (unsigned)N);
    if (dtype == GF_F16 && V == 8) upsample_add<_Float16, 8><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_F16) upsample_add<_Float16, 4><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_BF16 && V == 8) upsample_add<gf_bf16, 8><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_BF16) upsample_add<gf_bf16, 4><<<grid, 256, 0, st>>>(a);
    else upsample_add<float, 4><<<grid, 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_upsample_bilinear_backward_nhwc(const void* dhi, void* dlo, int N, int h, int w, int H, int W, int C, int dtype, void* stream) {
    GF_CHECK_ARG(dhi && dlo, "null pointer");
    GF_CHECK_ARG(N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "empty problem");
    GF_CHECK_ARG(dtype >= GF_F32 && dtype <= GF_BF16, "bad dtype");
    GF_CHECK_ARG(C % 4 == 0, "C must be a multiple of 4");
    GF_CHECK_ARG((uintptr_t)dhi % 16 == 0 && (uintptr_t)dlo % 16 == 0, "tensors must be 16-byte aligned");
    GF_CHECK_ARG(h <= 65535 && N <= 65535, "h and N must fit the launch grid");
    hipStream_t st = (hipStream_t)stream;
    UbArgs a{dhi, dlo, N, h, w, H, W, C, H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f};
    const int V = (dtype != GF_F32 && C % 8 == 0) ? 8 : 4;
    const dim3 grid((unsigned)((w * (C / V) + 255) / 256), (unsigned)h, (unsigned)N);
    if (dtype == GF_F16 && V == 8) upsample_backward<_Float16, 8><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_F16) upsample_backward<_Float16, 4><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_BF16 && V == 8) upsample_backward<gf_bf16, 8><<<grid, 256, 0, st>>>(a);
    else if (dtype == GF_BF16) upsample_backward<gf_bf16, 4><<<grid, 256, 0, st>>>(a);
    else upsample_backward<float, 4><<<grid, 256, 0, st>>>(a);
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_stem_conv7x7_dt(const void* image, int image_dtype, const float* weight, const float* shift, void* out, int out_dtype,
                                  int N, int H, int W, int C, void* stream) {
    GF_CHECK_ARG(image && weight && shift && out, "null pointer");
    GF_CHECK_ARG(N > 0 && H > 0 && W > 0, "empty problem");
    GF_CHECK_ARG(image_dtype == GF_F32 || image_dtype == out_dtype, "image dtype must be GF_F32 or the output's");
    GF_CHECK_ARG(out_dtype == GF_F16 || out_dtype == GF_BF16, "output dtype must be GF_F16 or GF_BF16");
    GF_CHECK_ARG(C == 128, "the stem kernel is built for 128 output channels (resnetfpn.initial_dim)");
    GF_CHECK_ARG((uintptr_t)out % 16 == 0, "output must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H + 2 * 3 - 7) / 2 + 1, Wo = (W + 2 * 3 - 7) / 2 + 1;
    StArgs a{image, weight, shift, out, N, H, W, Ho, Wo, (Wo + 31) / 32, (Ho + 3) / 4};
    const long ntiles = (long)N * a.tiles_x * a.tiles_y;
    const int blocks = (int)(ntiles < 512 ? ntiles : 512);               // persistent: two workgroups per CU
    if (out_dtype == GF_F16) {
        if (image_dtype == GF_F32) stem_conv<float, _Float16><<<blocks, 256, 0, st>>>(a);
        else stem_conv<_Float16, _Float16><<<blocks, 256, 0, st>>>(a);
    } else {
        if (image_dtype == GF_F32) stem_conv<float, gf_bf16><<<blocks, 256, 0, st>>>(a);
        else stem_conv<gf_bf16, gf_bf16><<<blocks, 256, 0, st>>>(a);
    }
    GF_CHECK_LAUNCH();
    return GF_OK;
}

extern "C" int gf_stem_conv7x7(const void* image, int image_dtype, const float* weight, const float* shift, void* out,
                               int N, int H, int W, int C, void* stream) {
    return gf_stem_conv7x7_dt(image, image_dtype, weight, shift, out, GF_F16, N, H, W, C, stream);
}
