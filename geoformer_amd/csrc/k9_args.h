// Arguments of the fused encoder-layer kernels (K9), shared by the kernel file and its launch sites.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GfEncArgs {
    const void* x;          // [N*L][ldx] tokens
    long ldx;
    const void* msg;        // [N*L][ldm] attention output (ATTN = false)
    long ldm;
    const float* kvfinal;   // [N][C*D + C] fp32: KV as [c][v] then Ksum[c]   (ATTN = true)
    const uint8_t* q_mask;  // [N*L] or null: masked query rows -> phi(q) = 0 (linear_attention.py:35-36)
    const void* wstream;    // packed fragments (see fused.py)
    const float* ln;        // gamma1 | beta1 | gamma2 | beta2
    float eps1, eps2, attn_eps;
    void* out;
    long ldo;
    int N, L, S, tiles;     // tiles = ceil(L / 128) per image
    const int32_t* flag;    // [N*L / flag_rows] or null: 0 -> out = x (GeoTransformer's "layer skipped")
    int flag_rows;
    // the state kernel
    const uint8_t* kv_mask; // [N*S] or null
    float* part;            // [N][tiles][C*D + C]
    // the layer's state tail: the images n >= tail_first ALSO leave the linear-attention state of their OUTPUT rows for
    // the layer call that reads them as its source: k / v projection with `wstream_tail` (that consumer's W_k | W_v stream)
    // straight from the finished tile in LDS, per-tile partials in `part` ([N - tail_first][tiles][C*D + C]); q_mask doubles as
    // the source mask (the rows are the same tokens).  null = no tail.
    const void* wstream_tail;
    int tail_first;
};
