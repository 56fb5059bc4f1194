/* geoformer_hip.h - C ABI of libgeoformer_hip.so (MI355X / gfx950).
 *
 * The reference (ruc-aimc-lab/GeoFormer) is pure Python on torch and has no FFI of its own; the
 * path this library replaces sits behind torch.nn.Module.forward() calls.  Each entry point below
 * replaces the sequence of ATen ops of ONE reference function (cited as file:line, relative to the
 * reference checkout) and is what a ctypes binding inside that function would call
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless a name ends in _host;
 *   - the caller allocates and owns every buffer, including the workspace
 *     (size from the matching gf_*_workspace_bytes query);
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*), never
 *     synchronises, never allocates; safe to capture in a hipGraph;
 *   - returns 0 (GF_OK) or a negative gf_status; gf_last_error() gives the message;
 *   - dtype: GF_F32 (parity mode, exact-fp32 MFMA), GF_F16 or GF_BF16 (16-bit storage, fp32 accumulate);
 *   - data-dependent sizes (match counts) are produced in device memory; entry points that
 *     consume them read them from device memory too, so no host round trip is forced.
 */
#ifndef GEOFORMER_HIP_H_
#define GEOFORMER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { GF_OK = 0, GF_ERR_INVALID_ARGUMENT = -1, GF_ERR_WORKSPACE = -2, GF_ERR_LAUNCH = -3 } gf_status;
typedef enum { GF_F32 = 0, GF_F16 = 1, GF_BF16 = 2 } gf_dtype;

/* Bumped whenever an existing entry point changes its signature or its defaults; a binding compares gf_abi_version() with the
 * GF_ABI_VERSION of the header it was written against at load time (geoformer_amd/_lib.py does) instead of calling shifted
 * arguments.  History: 1 = rounds 1-2; 2 = round 3 (gf_ransac_homography had gained `lm_iters` in the MIDDLE of its list: a
 * caller built against version 1 would have passed min_points as lm_iters); 3 = round 4: gf_ransac_homography is back to its
 * version-1 signature (no refinement), new arguments live in gf_ransac_homography_v2, appended at the END. */
/* 4 (round 4): the fragment stream of gf_conv3x3_nhwc deals the output channels differently (fused.py:pack_conv3x3_stream: a stream packed for
 * version 3 gives wrong channels) and GF_CONV_PAD16 now states that channels 196 .. 223 are padding; new: GF_CONV_S2, gf_lateral_upsample_add_nhwc. */
#define GF_ABI_VERSION 4
int gf_abi_version(void);
const char* gf_last_error(void);

/* Optional per-kernel timing with HIP events on the launch stream (used by bench.py's `roofline`).
 * Tags: "k1_stats" (work = flops), "k1_conf" (work = algorithmic bytes), "k3_linear" (work = flops).
 * gf_profile_collect synchronises on the recorded events and returns their summed time, count and work. */
void gf_profile_enable(int on);
void gf_profile_filter(const char* tag);   /* record only this tag (NULL: all): keeps the event count inside a timed region small */
int gf_profile_collect(const char* tag, double* total_ms, int* count, double* work);

/* ------------------------------------------------------------------------------------------
 * K1  dual-softmax correlation + mutual-nearest match extraction
 * replaces CoarseMatching.forward + get_coarse_match
 *          (model/loftr_src/loftr/utils/coarse_matching.py:90-130, :132-212)
 *
 *   sim  = <f0[n,i,:], f1[n,j,:]> / C / temperature       (-1e9 where !(mask0[n,i] & mask1[n,j]))
 *   conf = softmax(sim, dim=1) * softmax(sim, dim=2)       -> conf [N,L,S] fp32 (written unless conf == NULL, below)
 *   keep (n,i,j) iff conf > thr and conf is the maximum of its row and of its column; per row the
 *   first such column; rows emitted in (n,i) order, exactly like torch.where (:185-188).
 *   force_one != 0 reproduces the 'dataset_name' branch (:182-184): a sample without any match
 *   contributes (i=0, j=0).
 *
 *   f0 [N,L,C], f1 [N,S,C] of `dtype`, C a multiple of 64 (f16) / 32 (f32);
 *   mask0 [N,L], mask1 [N,S] uint8 (both NULL or both set);  scale0/scale1 [N,2] fp32 or NULL;
 *   w0c/w1c = coarse grid widths, scale = hw0_i[0]/hw0_c[0] (:193);
 *   outputs have capacity N*min(L,S) (+N when force_one): b/i/j_ids int64, mconf fp32,
 *   mkpts0_c/mkpts1_c [cap,2] fp32 (x,y);  counts int32[1+N]: total, then per sample.
 *
 *   MATCH-ONLY MODE (conf == NULL): the [N,L,S] matrix is not materialised - what inference.py:51-75 and
 *   eval_tool/immatch/modules/geoformer.py consume are the matches - and 2 x 164 MB of writes per 640x640 pair are
 *   saved; ids, mconf and keypoints are bit-identical to the contract mode's (entries the selection needs are
 *   recomputed with the sweep's own arithmetic).  Built for the configuration inference runs, see
 *   gf_dual_softmax_match_only_supported: 16-bit features, C = 256, L % 128 == 0, S % 64 == 0, no masks,
 *   force_one == 0; anything else with conf == NULL is GF_ERR_INVALID_ARGUMENT.
 *   gf_dual_softmax_conf_at returns single entries conf[b,i,j] afterwards (same features, the workspace of the
 *   last gf_dual_softmax_match call), bit-identical to what the contract mode writes.  gf_dual_softmax_match stamps
 *   the workspace with (N, L, S, C, temperature, f0, f1); conf_at compares the stamp with its own arguments ON THE
 *   DEVICE (no host synchronisation) and writes NaN for every entry if they differ, and for any b/i/j out of range.
 * ------------------------------------------------------------------------------------------ */
size_t gf_dual_softmax_workspace_bytes(int N, int L, int S);
int gf_dual_softmax_match_only_supported(int dtype, int L, int S, int C, int masked, int force_one);
int gf_dual_softmax_conf_at(const void* f0, const void* f1, int dtype, int N, int L, int S, int C, float temperature,
                            const int64_t* b, const int64_t* i, const int64_t* j, int P, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);
int gf_dual_softmax_match(const void* f0, const void* f1, int dtype, int N, int L, int S, int C,
                          const uint8_t* mask0, const uint8_t* mask1, float temperature, float thr,
                          int force_one, int w0c, int w1c, float scale, const float* scale0,
                          const float* scale1, float* conf, int64_t* b_ids, int64_t* i_ids,
                          int64_t* j_ids, float* mconf, float* mkpts0_c, float* mkpts1_c,
                          int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3 (training)  backward of the linear + LayerNorm / activation chain, mixed-16-bit step
 * what torch autograd derives for LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:45-60), the Geo
 *          layers (model/geo_transformer/transformer.py:49-66) and FinePreprocess' linears (fine_preprocess.py:61-72) when
 *          lightning/train_depth_geoformer.py:117-119 runs the step under 16-bit autocast; fp32 master weights / gradients.
 *   y = x W^T:   dX = dY W is gf_linear with the transposed weight;  dW = dY^T X is gf_linear_wgrad (fp32 [cout, cin], row
 *   stride lddw, optionally accumulated): the contraction runs over the T token rows of both operands (MFMA operands by
 *   transpose reads), split over token chunks, partials added in chunk order (deterministic).  cout, cin multiples of 8 (round 6; 128 before).
 *   gf_layernorm_forward keeps stats[t] = (mean, rstd); gf_layernorm_backward returns dy and dgamma / dbeta (fp32 [C],
 *   optionally accumulated); C in {128, 256, 512}.  gf_activation_backward: dz = dh * act'(z) from the OUTPUT h of the
 *   activation (kind 0 ReLU, 1 Tanh).  All activations GF_F16 or GF_BF16, statistics and parameter gradients fp32.
 * ------------------------------------------------------------------------------------------ */
size_t gf_linear_wgrad_workspace_bytes(long T, int cout, int cin);
int gf_linear_wgrad(const void* dy, long lddy, const void* x, long ldx, int dtype, long T, int cout, int cin, float* dw,
                    long lddw, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int gf_layernorm_forward(const void* y, int dtype, long T, int C, const float* gamma, const float* beta, float eps, void* out,
                         float* stats, void* stream);
size_t gf_layernorm_backward_workspace_bytes(int C);
int gf_layernorm_backward(const void* dout, const void* y, const float* stats, int dtype, long T, int C, const float* gamma, void* dy,
                          float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int gf_activation_backward(const void* dh, const void* h, void* dz, size_t n, int kind, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2 (training)  backward of LinearAttention.forward (model/loftr_src/loftr/loftr_module/linear_attention.py:21-51), heads of 32
 *   given dout [N, L, H, 32]: dq [N, L, H*32], dk, dv [N, S, H*32] (contiguous) of out = phi(q) KV S / (phi(q) . Ksum + eps),
 *   KV = sum_s phi(k_s)^T (v_s / S), Ksum = sum_s phi(k_s), with the q / kv padding masks of :35-39.  q, k, v, dout: [N, L|S, H, 32]
 *   views with row strides in elements (GF_F16 / GF_BF16); fp32 arithmetic; the (image, head) states are sums of chunk partials
 *   in chunk order (deterministic).
 * ------------------------------------------------------------------------------------------ */
/* K8 (training): df0, df1 [M, 25, C] (`dtype`: GF_F32 / GF_F16 / GF_BF16) of FineMatching2.forward's confidence (model/fine_matching2.py:52-63:
 * sim = f0 f1^T / (C temperature), conf = softmax(sim, 1) * softmax(sim, 2)) given dconf fp32 [M, 25, 25]; fp32 arithmetic. */
int gf_fine_match_backward(const void* f0, const void* f1, int dtype, int M, int WW, int C, float temperature, const float* dconf,
                           void* df0, void* df1, void* stream);
size_t gf_linear_attention_backward_workspace_bytes(int N, int L, int S, int H);
int gf_linear_attention_backward(const void* q, const void* k, const void* v, const void* dout, int dtype, int N, int L, int S, int H,
                                 int D, long ldq, long ldk, long ldv, long ldo, const uint8_t* q_mask, const uint8_t* kv_mask, float eps,
                                 void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes, void* stream);
/* the same backward for the fine level's windows: q, k, v, dout, dq, dk, dv contiguous [Nw, Lw <= 32, 8 heads x 16] tensors, no masks
 * (the forward is gf_linear_attention's window form); one workgroup per window, fp32 arithmetic on the 16-bit operands. */
int gf_window_linear_attention_backward(const void* q, const void* k, const void* v, const void* dout, int dtype, int Nw, int Lw,
                                        float eps, void* dq, void* dk, void* dv, void* stream);

/* ------------------------------------------------------------------------------------------
 * K10 (training)  weight gradient of the backbone's 3x3 / stride 1 / pad 1 convolutions
 * replaces, for the training step, autograd's weight gradient of the BasicBlock / FPN-head convolutions
 *          (model/loftr_src/loftr/backbone/resnet_fpn.py:9-40,60-83; the library call is aten::convolution_backward, weights only).
 *   x  channels-last [N, H, W, cx], dy channels-last [N, H, W, cy] (GF_F16 / GF_BF16; cx, cy = STORED widths, multiples of 8 - the
 *   196-channel level is stored 224 wide), cin <= cx, cout <= cy the real widths;
 *   dw fp32 [cout][cin][3][3] (overwritten) = sum_n,y,x dy[n,y,x,co] * x[n, y+ky-1, x+kx-1, ci]; fp32 accumulation, partial sums
 *   per run of strip rows added in a fixed order (bit-reproducible).  workspace: gf_conv3x3_wgrad_workspace_bytes.
 * ------------------------------------------------------------------------------------------ */
size_t gf_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int cin, int cout);
int gf_conv3x3_wgrad_nhwc(const void* x, const void* dy, int dtype, int N, int H, int W, int cx, int cy, int cin, int cout, float* dw,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4 (training)  FullAttention.forward with saved softmax statistics, and its backward (flash form)
 * replaces, for the training step, model/geo_transformer/geo_attention.py:72-101 as GeoTransformer's 'self' branch calls it
 *          (model/geo_transformer/transformer.py:111-124: every cell of an image against the projected rows of its inlier cells,
 *          no masks) and the autograd backward through it.  Nothing of size L x S is stored.
 *   q [N, L, 4 x 64], k, v [N, S, 4 x 64]: row-strided views (ld* in elements, multiples of 8, 16-byte aligned rows; batch stride =
 *   rows x row stride), GF_F16 / GF_BF16; softmax_temp = 1 / sqrt(64) is applied to the fp32 logits (not folded into q).
 *   forward:  out [N, L, ldo] = softmax(q k^T softmax_temp) v, lse fp32 [N, 4, L] = log2 of the row's sum of exponentials (base 2,
 *             scaled logits: m + log2 l);  S == 0: out = 0.
 *   backward: dq [N, L, 256], dk, dv [N, S, 256] (contiguous, `dtype`) given out, dout [N, L, ld] and the forward's lse;
 *             P and dS are rounded to `dtype` for the second product of each pair, accumulation fp32, partial sums in a fixed
 *             order (bit-reproducible).  workspace: gf_full_attention_backward_workspace_bytes (the rows' dO . O).
 * ------------------------------------------------------------------------------------------ */
int gf_full_attention_train_forward(const void* q, const void* k, const void* v, int dtype, int N, int L, int S, int H, int D, long ldq,
                                    long ldk, long ldv, float softmax_temp, void* out, long ldo, float* lse, void* stream);
size_t gf_full_attention_backward_workspace_bytes(int N, int L, int H);
int gf_full_attention_backward(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                               int dtype, int N, int L, int S, int H, int D, long ldq, long ldk, long ldv, long ldo, long lddo,
                               float softmax_temp, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * K1 (training)  sparse-supervision focal loss on the dual-softmax confidence, forward and backward
 * replaces, for the training step, CoarseMatching.forward's conf_matrix (coarse_matching.py:113-125) as consumed by
 *          GeoLoss.compute_coarse_loss, focal / sparse_spvs / dual_softmax branch (loftr_loss.py:246-270), and the
 *          autograd backward through both.  Neither conf nor its gradient is materialised.
 *   f0 [N,L,256], f1 [N,S,256] (GF_F32 or GF_F16; the arithmetic is fp16 operands / fp32 accumulation),
 *   positives (pos_b, pos_i, pos_j)[P] = torch.where(conf_matrix_gt == 1), optional per-positive weight,
 *   optional padding masks uint8 [N,L] / [N,S] (pairs with a padded member: sim = -1e9, coarse_matching.py:123-124).
 *   forward:  pos_conf[k] = conf[b,i,j], pos_loss[k] = -alpha (1-p)^gamma log p * w  (p clamped to [1e-6, 1-1e-6]),
 *             pos_grad[k] = d pos_loss[k] / d log p.   loss_c = c_pos_w * mean_k pos_loss[k] is formed by the caller.
 *   backward: d_f0 [N,L,256], d_f1 [N,S,256] fp32 (overwritten) for  loss = scale * scale_dev[0] * sum_k pos_loss[k]
 *             (scale_dev: optional DEVICE scalar, NULL = 1; lets autograd hand the upstream gradient over without a
 *             host synchronisation);
 *             the workspace of the forward call must be passed unchanged (fp16 features and softmax statistics).
 *   L and S multiples of 128, C = 256.
 * ------------------------------------------------------------------------------------------ */
size_t gf_coarse_loss_workspace_bytes(int N, int L, int S);
int gf_coarse_loss_forward(const void* f0, const void* f1, int dtype, int N, int L, int S, int C, const uint8_t* mask0,
                           const uint8_t* mask1, float temperature, const int64_t* pos_b, const int64_t* pos_i, const int64_t* pos_j, int P,
                           const float* pos_weight, float alpha, float gamma, float* pos_conf, float* pos_loss,
                           float* pos_grad, void* workspace, size_t workspace_bytes, void* stream);
int gf_coarse_loss_backward(int N, int L, int S, int C, const uint8_t* mask0, const uint8_t* mask1, float temperature,
                            const int64_t* pos_b, const int64_t* pos_i,
                            const int64_t* pos_j, int P, const float* pos_grad, float scale, const float* scale_dev,
                            float* d_f0, float* d_f1, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * a1  position encoding add + flatten
 * replaces PositionEncodingSine.forward (model/loftr_src/loftr/utils/position_encoding.py:37-42)
 *          and the permute/reshape of model/full_model.py:69-77
 *   out[n, y*W+x, c] = x[n,c,y,x] + pe[y,x,c];  x viewed as [N,C,H,W] through element strides
 *   (NCHW or channels_last), pe fp32 [H,W,C] (host table built as position_encoding.py:22-35).
 * ------------------------------------------------------------------------------------------ */
int gf_pos_encode(const void* x, int x_dtype, long sn, long sc, long sh, long sw, const float* pe,
                  void* out, int out_dtype, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  linear attention
 * replaces LinearAttention.forward (model/loftr_src/loftr/loftr_module/linear_attention.py:21-51)
 *   q [N,L,H,D], k,v [N,S,H,D] with row strides ldq/ldk/ldv (elements), masks uint8 or NULL,
 *   out [N,L,H*D] contiguous.  D in {16,32,64}, H*D in {64,128,192,256}.
 *   Rounding points in the 16-bit modes: the coarse shape (8 heads of 32) and the fine level's windows (8 heads of 16,
 *   L, S <= 32, 16-byte aligned rows) run on the matrix cores: phi(q), phi(k), KV / S and Ksum / S are rounded to the
 *   storage type, sums and the division are fp32 (oracle: linear_attention_fused / linear_attention_window); every other
 *   shape evaluates phi, the state and the normaliser in fp32 from the stored q, k, v.
 * ------------------------------------------------------------------------------------------ */
size_t gf_linear_attention_workspace_bytes(int N, int S, int H, int D);
int gf_linear_attention(const void* q, const void* k, const void* v, int dtype, int N, int L, int S, int H,
                        int D, long ldq, long ldk, long ldv, const uint8_t* q_mask, const uint8_t* kv_mask,
                        float eps, void* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9  the encoder layer in two launches (16-bit storage modes)
 * replaces LoFTREncoderLayer.forward as a whole (model/loftr_src/loftr/loftr_module/transformer.py:37-60 with
 *          LinearAttention.forward, linear_attention.py:21-51) and the part of the Geo layer after its attention
 *          (model/geo_transformer/transformer.py:56-66); d_model 256, 8 heads of 32 (linear attention).
 *   gf_encoder_kv_state: kv_state[n] = { KV[c][v] = sum_s phi(k)[s,c] v[s, head(c)*32 + v],  Ksum[c] = sum_s phi(k)[s,c] },
 *          k = W_k src, v = W_v src, phi = elu + 1, masked / out-of-range rows excluded; fp32 [N][256*32 + 256]
 *          (KV from the 16-bit rounded phi(k) and v, Ksum from the fp32 phi(k)).
 *          wstream_kv = W_k and W_v packed by geoformer_amd/fused.py:pack_kv_stream (256 KiB).
 *   gf_encoder_layer: out = x + LN2(W_2 act(W_1 [x | LN1(W_m msg)])) with
 *          msg = phi(W_q x) KV / (phi(W_q x) . Ksum + attn_eps)       when kv_state is given (S = source length), or
 *          msg = the given attention output [N*L, 256]                 when msg is given;
 *          wstream = fused.py:pack_layer_stream (1 MiB with W_q, 896 KiB without); ln_params = gamma1|beta1|gamma2|beta2
 *          fp32 [4][256]; activation 0 = ReLU, 1 = Tanh; row_flag / flag_rows as in gf_linear (0 -> out = x).
 *   Rounding points (what oracle/geoformer_oracle.py's storage mode mirrors): every MFMA operand is rounded to the
 *   storage type (phi(q), phi(k), v, KV/S, Ksum/S, msg, LN1 output, hidden activations); accumulation, LayerNorm,
 *   phi, the activation and the residual sum are fp32; out is rounded once.
 * ------------------------------------------------------------------------------------------ */
size_t gf_encoder_kv_workspace_bytes(int N, int S);
int gf_encoder_kv_state(const void* src, long ld, int dtype, int N, int S, const uint8_t* kv_mask,
                        const void* wstream_kv, float* kv_state, void* workspace, size_t workspace_bytes,
                        void* stream);
int gf_encoder_layer(const void* x, long ldx, const void* msg, long ldm, const float* kv_state, int S,
                     const uint8_t* q_mask, float attn_eps, const void* wstream, const float* ln_params, float eps1,
                     float eps2, int activation, const int32_t* row_flag, int flag_rows, void* out, long ldo, int dtype,
                     int N, int L, void* stream);
/*   gf_encoder_layer_kv (round 4): gf_encoder_layer in its linear-attention form (kv_state given) whose launch ALSO produces,
 *          for the images tail_first .. N-1, kv_state_out[n - tail_first] = gf_encoder_kv_state of their OUTPUT rows under
 *          wstream_tail - the W_k | W_v stream of the layer call that reads those rows as its source (transformer.py:95-100:
 *          the next layer's source is this layer's output).  The finished 128-token tile is still in the CU's LDS when the
 *          tail's weights arrive through the same ring: no second read of the features, no second launch; masked rows
 *          (q_mask) do not count, as in gf_encoder_kv_state with kv_mask = q_mask.  workspace: gf_encoder_kv_workspace_bytes(
 *          N - tail_first, L).  The states are bit-identical to gf_encoder_kv_state(out[tail_first:], ...)'s. */
int gf_encoder_layer_kv(const void* x, long ldx, const float* kv_state, int S, const uint8_t* q_mask, float attn_eps,
                        const void* wstream, const float* ln_params, float eps1, float eps2, int activation, void* out,
                        long ldo, int dtype, int N, int L, const void* wstream_tail, int tail_first, float* kv_state_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K11 one encoder layer of the fine-level transformer in ONE launch (16-bit storage modes)
 * replaces LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:37-60) with LinearAttention.forward
 *          (linear_attention.py:21-51) as LocalFeatureTransformer.forward (:82-104) runs it on the fine windows
 *          (model/full_model.py:97-98: [M, W*W = 25, 128] tensors, 8 heads of 16, no masks).
 *   out[w] = x[w] + LN2(W_2 relu(W_1 [x[w] | LN1(W_m msg[w])])),
 *   msg[w] = phi(W_q x[w]) KV[w] / (phi(W_q x[w]) . Ksum[w] + attn_eps),  KV[w], Ksum[w] from phi(W_k src[w]), W_v src[w]
 *   x, src, out: [Nw][Lw][128] contiguous windows of `dtype` (src may be x: the 'self' layers), Lw <= 32;
 *   wstream = geoformer_amd/fused.py:pack_fine_layer_stream (320 KiB); ln_params = gamma1|beta1|gamma2|beta2 fp32 [4][128].
 *   Rounding points = those of the gf_linear / gf_linear_attention chain it replaces (q, k, v, phi(q), phi(k), KV/S, Ksum/S,
 *   the message, LN1 output, hidden activations and LN2 output rounded to the storage type; fp32 accumulation).
 * ------------------------------------------------------------------------------------------ */
int gf_fine_layer(const void* x, const void* src, void* out, int dtype, int Nw, int Lw, const void* wstream,
                  const float* ln_params, float eps1, float eps2, float attn_eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * K10 3x3 / stride 1 / pad 1 convolution of channels-last 16-bit maps with a fused epilogue (backbone)
 * replaces conv3x3 + BatchNorm(eval) [+ shortcut] + ReLU / LeakyReLU of BasicBlock.forward and of the FPN heads
 *          (model/loftr_src/loftr/backbone/resnet_fpn.py:9-40, :60-83, :100-116), BatchNorm folded into the weights
 *   out[n,y,x,:] = act( sum_{ky,kx} w[:, :, ky, kx] . x[n, y+ky-1, x+kx-1, :] + shift + residual[n,y,x,:] )
 *   x [N,H,W,cin], out / residual [N,H,W,cout] (GF_F16 or GF_BF16); fp32 accumulation starting from the shift; GF_F16 with
 *   act 0 / 1: the sum is rounded to half before the residual is added in packed half arithmetic and once more at the output (a
 *   separate convolution followed by gf_bias_act_nhwc rounds twice as well); GF_BF16 and LeakyReLU: residual and activation on
 *   the fp32 accumulators, ONE rounding at the output; act: 0 none, 1 ReLU, 2 LeakyReLU(slope in [0,1]), optionally
 *   | GF_CONV_PAD16 (cout = 224): the caller states that the output channels 196 .. 223 are zero padding (zero weights: the 196
 *   real channels of the reference's middle pyramid level in a 224-wide map) - 16 of them are not multiplied (the accumulator tile
 *   that holds only padding), their outputs are act(shift + residual) as everywhere;
 *   shift fp32 [cout] or NULL; residual or NULL; maps must hold fewer than 2^31 elements;
 *   wstream = geoformer_amd/fused.py:pack_conv3x3_stream(w); zeros = >= 64 bytes of zeroed device memory.
 *   Channel counts: gf_conv3x3_supported(cin, cout).
 * ------------------------------------------------------------------------------------------ */
#define GF_CONV_PAD16 0x100
/*   | GF_CONV_REM8 (round 4; Cin = 224 with Cout = 224 | GF_CONV_PAD16, or Cout = 128): the caller states that the INPUT channels
 *     200 .. 223 carry zero weights (196 real channels: the other side of the same padding) and passes the rem8 packing of the
 *     weights (fused.py:pack_conv3x3_stream(w, rem8=True)): channels 0 .. 191 run as six 32-channel chunks, channels 192 .. 199 as a
 *     remainder whose 9 taps x 8 channels fill three MFMA k-steps instead of a seventh chunk's nine - 114 instead of 126 sub-steps per
 *     tile, the result differs from the plain call only by the order of the fp32 sums. */
#define GF_CONV_REM8 0x200
/*   | GF_CONV_S2 (round 4; cin -> cout in gf_conv3x3s2_supported: 128 -> 224, 224 -> 256): STRIDE 2 - the first convolution of layer2 /
 *     layer3 (resnet_fpn.py:14-17 with stride = 2).  H x W is then the INPUT map; out (and residual) are [N][(H-1)/2+1][(W-1)/2+1][cout];
 *     wstream = fused.py:pack_conv3x3_stream(w, s2=True) (the taps listed parity plane by parity plane).  Replaces the last two
 *     F.conv2d / MIOpen calls + gf_bias_act_nhwc of the 16-bit backbone. */
#define GF_CONV_S2 0x400
int gf_conv3x3_supported(int cin, int cout);
int gf_conv3x3s2_supported(int cin, int cout);
int gf_conv3x3_nhwc(const void* x, const void* wstream, const float* shift, const void* residual, void* out,
                    const void* zeros, int N, int H, int W, int cin, int cout, int act, float slope, int dtype,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  encoder-layer linears with fused epilogues
 * replaces the nn.Linear / LayerNorm / activation / concat / residual sequence of
 * LoFTREncoderLayer.forward (model/loftr_src/loftr/loftr_module/transformer.py:45-60 and
 * model/geo_transformer/transformer.py:49-66) and the two linears of FinePreprocess.forward
 * (model/loftr_src/loftr/loftr_module/fine_preprocess.py:61-72)
 *   out[m,n] = epi( sum_k [a1|a2][m,k] * w[n,k] + bias[n] + rowgroup_bias[m / rowgroup_rows, n] )
 *   a1 [M,k1] row stride lda1, a2 [M,k2] (k2 may be 0) - the two halves of torch.cat([x, message], 2);
 *   w [N, k1+k2] row-major (nn.Linear.weight); bias fp32 [N] or NULL; rowgroup_bias [M/rows, N] or NULL;
 *   epilogue: 0 none, 1 ReLU, 2 Tanh, 3 LayerNorm(gamma, beta, eps) over N,
 *             4 residual + LayerNorm(...) with optional predicate row_flag[m / flag_rows]
 *               (0 -> out = residual: the GeoTransformer "layer skipped for this sample" case);
 *   LayerNorm epilogues need N in {128, 256}.
 * ------------------------------------------------------------------------------------------ */
int gf_linear(const void* a1, long lda1, int k1, const void* a2, long lda2, int k2, const void* w,
              const float* bias, const void* rowgroup_bias, int rowgroup_rows, int epilogue,
              const float* ln_gamma, const float* ln_beta, float ln_eps, const void* residual, long ldres,
              const int32_t* row_flag, int flag_rows, void* out, long ldo, int dtype, int M, int N,
              void* stream);

/* ------------------------------------------------------------------------------------------
 * homography RANSAC on the device
 * replaces the cv2.findHomography(kp0, kp1, cv2.RANSAC, 8.0) host round trip and the inlier
 * filtering of GeoModule.apply_RANSAC (model/geo_module.py:38-52).  OpenCV parity is unpinned;
 * the algorithm is the one stated in oracle/ransac_oracle.c (bit-exact inlier mask).
 *   mkpts0_c/mkpts1_c [cap,2] fp32 and counts int32[1+N] as written by gf_dual_softmax_match;
 *   gf_ransac_homography_v2 only - lm_iters: Levenberg-Marquardt steps on the inliers' forward transfer error behind the
 *   least-squares refit (OpenCV's findHomography appends 10 to its RANSAC; 0 = none = gf_ransac_homography); the inlier mask
 *   is the best hypothesis' either way;
 *   min_points: samples with fewer matches get no model (GeoModule passes 9: `len(kp0) > 8`, :46);
 *   integer_keypoints = 1: keypoints are truncated like the reference's .long() (GeoModule);
 *   0: sub-pixel keypoints are used as given (homography estimation from fine matches in the
 *   evaluation harness, eval_tool/immatch/utils/hpatches_helper.py:216);
 *   outputs: kp0/kp1 fp32 [cap,2] (the keypoints RANSAC saw), M fp64 [N,9], M_f32 / Minv_f32 [N,9]
 *   (the casts of :58 and :67), valid int32 [N] (0 = "M is None"), keep uint8 [cap] (inlier, or 1
 *   for every match of a sample without model: what feeds the occupancy maps of :82-94).
 * ------------------------------------------------------------------------------------------ */
size_t gf_ransac_workspace_bytes(int N, int iters);
int gf_ransac_homography(const float* mkpts0_c, const float* mkpts1_c, const int32_t* counts, int N,
                         int capacity, float scale, const float* scale0, const float* scale1, float thr,
                         int iters, uint32_t seed, int min_points, int integer_keypoints, float* kp0, float* kp1, double* M,
                         float* M_f32, float* Minv_f32, int32_t* valid, uint8_t* keep, void* workspace,
                         size_t workspace_bytes, void* stream);
int gf_ransac_homography_v2(const float* mkpts0_c, const float* mkpts1_c, const int32_t* counts, int N,
                            int capacity, float scale, const float* scale0, const float* scale1, float thr,
                            int iters, uint32_t seed, int min_points, int integer_keypoints, float* kp0, float* kp1, double* M,
                            float* M_f32, float* Minv_f32, int32_t* valid, uint8_t* keep, void* workspace,
                            size_t workspace_bytes, void* stream, int lm_iters);

/* ------------------------------------------------------------------------------------------
 * a8  window geometry
 * replaces get_map_keypoints (utils/common_utils.py:137-144) + warp_points_batch
 * (utils/homography.py:86-105) + generate_window (utils/common_utils.py:65-91) + the cell lookup of
 * sample_descriptors (utils/common_utils.py:171-181)
 *   H fp32 [N,9] maps the (hq x wq) coarse grid of the query image into the other image
 *   (Himg x Wimg pixels, coarse width wk); win int32 [N, hq*wq, ws*ws] = coarse cell sampled by each
 *   window position or -1 when out of bounds (or when valid[n] == 0).
 *   Optional outputs for tests: kps int32 [N,L,ws*ws,2] (the .long() window coordinates, 0 when
 *   out of bounds) and warped fp32 [N,L,2].
 * ------------------------------------------------------------------------------------------ */
int gf_window_geometry(const float* H, const int32_t* valid, int N, int hq, int wq, int Himg, int Wimg,
                       int wk, int scale, int window_size, const float* window_scale, int32_t* win,
                       int32_t* kps_or_null, float* warped_or_null, void* stream);

/* ------------------------------------------------------------------------------------------
 * a7  inlier occupancy maps + token lists
 * replaces model/geo_module.py:82-94 and the boolean-mask gathers feat[mask] of
 * model/geo_transformer/transformer.py:118,121
 *   map0 uint8 [N,L], map1 uint8 [N,S]; idx0 int32 [N,L], idx1 int32 [N,S] ascending cell lists;
 *   nidx int32 [N,2] their lengths.
 * ------------------------------------------------------------------------------------------ */
int gf_inlier_index(const float* kp0, const float* kp1, const uint8_t* keep, const int32_t* counts, int N,
                    int L, int S, int w0, int w1, int scale, uint8_t* map0, uint8_t* map1, int32_t* idx0,
                    int32_t* idx1, int32_t* nidx, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  self attention over the tokens at inlier cells
 * replaces FullAttention.forward (model/geo_transformer/geo_attention.py:72-101) as called from the
 * 'self' branch of GeoTransformer.forward (model/geo_transformer/transformer.py:111-124)
 *   q [N,L,256], kmap/vmap [N,L,256] = k_proj/v_proj of EVERY token (row strides ld*), idx/nkeys =
 *   the token list of gf_inlier_index; out [N,L,256]; nkeys == 0 -> zeros (layer skipped by caller).
 *   16-bit modes: q and kmap 16-byte aligned with row strides that are multiples of 8 elements (rows move as 16-byte pieces); with
 *   vmap aligned the same way the key / value rows are read from the maps directly (no gather pass; the workspace is then unused):
 *   with row strides below 8192 elements by the head form - one head and 128 queries per workgroup, rows through structured buffer
 *   descriptors - otherwise by the four-head form; both run the arithmetic below bit for bit.
 *   Arithmetic of the 16-bit modes (flash form): the softmax scale lives in the query operand, q' = round(q * log2(e) / sqrt(D)) to the
 *   storage type (one more 16-bit rounding of q), logits q' . k in fp32; per query a softmax reference that starts at the first key
 *   tile's maximum and moves up only when a tile's maximum exceeds it by more than 8 (log2 units); probabilities rounded to the
 *   storage type for P.V, their sum in fp32; mathematically softmax(Q K^T / sqrt(D)) V.  GF_F32: the exact running maximum, the
 *   scale applied to the fp32 logits, fp32 throughout.
 * ------------------------------------------------------------------------------------------ */
size_t gf_self_attention_workspace_bytes(int N, int L, int dtype);
int gf_self_attention_gathered(const void* q, const void* kmap, const void* vmap, int dtype, int N, int L, int H,
                               int D, long ldq, long ldk, long ldv, const int32_t* idx, long idx_stride,
                               const int32_t* nkeys, int nkeys_stride, void* out, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  windowed cross attention
 * replaces sample_descriptors (utils/common_utils.py:166-208) + FullAttention.forward with kv_mask
 * (model/geo_transformer/geo_attention.py:72-101) as called from the 'cross' branch of
 * GeoTransformer.forward (model/geo_transformer/transformer.py:125-139)
 *   q [N,L,256]; kmap/vmap [N,S,256] = k_proj/v_proj of every token of the OTHER image;
 *   win int32 [N,L,25] from gf_window_geometry; out [N,L,256] (zeros where all 25 are masked).
 * ------------------------------------------------------------------------------------------ */
int gf_window_cross_attention(const void* q, const void* kmap, const void* vmap, int dtype, int N, int L, int S,
                              int H, int D, long ldq, long ldk, long ldv, const int32_t* win, int WW,
                              const int32_t* valid, void* out, void* stream);
/* Backward of gf_window_cross_attention for the training step: dq [N,L,256] of `dtype`; dk, dv fp32 [N,S,256],
 * ZEROED by the caller - overlapping windows make them scatter-adds (fp32 atomics; the sums are rounded to the storage type once,
 * by the caller).  dout [N,L,256] contiguous.  Masked window positions and queries without a valid key get no gradient. */
int gf_window_cross_attention_backward(const void* q, const void* kmap, const void* vmap, const void* dout, int dtype, int N, int L,
                                       int S, int H, int D, long ldq, long ldk, long ldv, const int32_t* win, int WW, void* dq,
                                       float* dk, float* dv, void* stream);
/* the same gradients without atomics (round 6): dk, dv [N, S, 256] of `dtype` gathered cell by cell along the caller's inverse index of
 * `win` - entries int32 [N * L * WW] = l * WW + k of every (query, window position), sorted (stable) by global cell n * S + cell with the masked
 * positions (cell < 0) last; offsets int32 [N * S + 1] - so the sums run in a fixed order (bit-reproducible) and no fp32 maps are needed.
 * workspace: gf_window_cross_attention_backward_workspace_bytes (the per-(query, position, head) dlogit and p). */
size_t gf_window_cross_attention_backward_workspace_bytes(int N, int L, int WW);
int gf_window_cross_attention_backward_gather(const void* q, const void* kmap, const void* vmap, const void* dout, int dtype, int N, int L,
                                              int S, int H, int D, long ldq, long ldk, long ldv, const int32_t* win, int WW,
                                              const int32_t* entries, const int32_t* offsets, void* dq, void* dk, void* dv, void* workspace,
                                              size_t workspace_bytes, void* stream);
/* The same operation when the query map is hq x wq cells and the key map hk x wk cells (L = hq*wq, S = hk*wk, cells row-major as
 * gf_window_geometry numbers them).  16-bit storage: one workgroup per tile of 8 x 4 query cells and head; the windows of a
 * tile overlap, so the key / value rows of their bounding rectangle are staged in LDS once (a tile whose rectangle exceeds 144
 * cells reads its rows from global memory).  fp32 storage: forwards to gf_window_cross_attention.  Rows must be 16-byte
 * aligned (ld % 8 == 0). */
int gf_window_cross_attention_tiled(const void* q, const void* kmap, const void* vmap, int dtype, int N, int hq, int wq,
                                    int hk, int wk, int H, int D, long ldq, long ldk, long ldv, const int32_t* win,
                                    int WW, const int32_t* valid, void* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7  fine window extraction
 * replaces F.unfold + gather + coarse-row gather of FinePreprocess.forward
 * (model/loftr_src/loftr/loftr_module/fine_preprocess.py:41-61)
 *   feat_f0/1 viewed as [N,C,H,W] through strides (4 longs each); feat_c0 [N,L,CC], feat_c1 [N,S,CC];
 *   win_out [2M, W*W, C] (image0 windows then image1), ccat_out [2M, CC].
 * ------------------------------------------------------------------------------------------ */
int gf_fine_gather(const void* feat_f0, const void* feat_f1, int feat_dtype, const long* strides0,
                   const long* strides1, int H0, int W0, int H1, int W1, int C, const void* feat_c0,
                   const void* feat_c1, int dtype, int L, int S, int CC, const int64_t* b_ids,
                   const int64_t* i_ids, const int64_t* j_ids, int M, int w0c, int w1c, int stride, int window,
                   void* win_out, void* ccat_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backbone glue (fp16 inference backbone; the convolutions themselves stay on PyTorch-ROCm / MIOpen)
 * replaces the element-wise passes of BasicBlock.forward and the FPN merges
 *          (model/loftr_src/loftr/backbone/resnet_fpn.py:20-40 bn/relu/shortcut, :92-95 stem,
 *           :104-105 and :110-111 F.interpolate(..., align_corners=True) + add, :106-115 BN + LeakyReLU)
 *   channels-last tensors ([pixels, C] in memory), C % 4 == 0, 16-byte aligned, in-place allowed.
 *   gf_bias_act_nhwc:     out = act(x + bias[c] + residual);  bias fp32 [C] or NULL, residual or NULL;
 *                         act 0 none, 1 ReLU, 2 LeakyReLU(slope)
 *   gf_upsample_add_nhwc: out = hi + bilinear(lo -> HxW, align_corners=True);  lo [N,h,w,C], hi/out [N,H,W,C]
 * ------------------------------------------------------------------------------------------ */
int gf_bias_act_nhwc(const void* x, const float* bias, const void* residual, void* out, long pixels, int C, int act,
                     float slope, int dtype, void* stream);
int gf_upsample_add_nhwc(const void* lo, const void* hi, void* out, int N, int h, int w, int H, int W, int C,
                         int dtype, void* stream);
/*   gf_upsample_bilinear_backward_nhwc (training): dlo [N,h,w,C] = the gradient of bilinear(lo -> HxW, align_corners=True) with respect to lo
 *                         given dhi [N,H,W,C] (F.interpolate of resnet_fpn.py:104-105,:110-111 under autograd); a gather, no atomics. */
int gf_upsample_bilinear_backward_nhwc(const void* dhi, void* dlo, int N, int h, int w, int H, int W, int C, int dtype, void* stream);
/*   gf_conv1x1_upsample_add_nhwc: out = conv1x1(x; w [Cout,Cin]) + bilinear(lo -> HxW, align_corners=True): the FPN lateral
 *                         convolution with the top-down merge as its epilogue (resnet_fpn.py:109-111); fp16, Cin % 64 == 0 */
int gf_conv1x1_upsample_add_nhwc(const void* x, const void* w, const void* lo, void* out, int N, int h, int wl, int H,
                                 int W, int Cin, int Cout, int dtype, void* stream);

/* K12 (round 4): the same operation as gf_conv1x1_upsample_add_nhwc for the 1/2-scale lateral (cin 128 -> cout 224: layer1_outconv +
 * the merge with the upsampled 1/4-scale map, resnet_fpn.py:109-111) as a streaming kernel of its own - weights resident in LDS as MFMA
 * fragments (wfrag = geoformer_amd/fused.py:pack_lateral_frags(w)), pixel rows by LDS-DMA, the merge's taps requested in front of the
 * tile's MFMAs, 128-byte stores; W must be even.  gf_lateral_supported(cin, cout) names the built widths. */
int gf_lateral_supported(int cin, int cout);
int gf_lateral_upsample_add_nhwc(const void* x, const void* wfrag, const void* lo, void* out, int N, int h, int w, int H, int W,
                                 int cin, int cout, int dtype, void* stream);
/* 1x1 convolution of a channels-last 16-bit map, stride 1 or 2 (H, W even), no bias: out [N, H/s, W/s, Cout] = W x[n, s y, s x, :]
 * (resnet_fpn.py:23-27 the BasicBlock downsample shortcut conv1x1(stride 2), :69-71 layer3_outconv / layer2_outconv, with the
 * BatchNorm scale folded into w [Cout, Cin]); Cin, Cout multiples of 32. */
int gf_conv1x1_nhwc(const void* x, const void* w, void* out, int N, int H, int W, int Cin, int Cout, int stride, int dtype,
                    void* stream);
/*   gf_stem_conv7x7:      out[n,oy,ox,c] = relu(sum_{ky,kx} image[n, 2oy-3+ky, 2ox-3+kx] * weight[c,ky,kx] + shift[c])
 *                         = conv1 (7x7, stride 2, pad 3, 1 input channel) + bn1 (folded) + relu, resnet_fpn.py:60-62, :92;
 *                         image [N,H,W] fp32 or fp16, weight fp32 [C,7,7], shift fp32 [C], out fp16 [N,Ho,Wo,C], C = 128 */
int gf_stem_conv7x7(const void* image, int image_dtype, const float* weight, const float* shift, void* out, int N,
                    int H, int W, int C, void* stream);
/*   gf_stem_conv7x7_dt:   the same with the output / compute type as an argument (GF_F16 or GF_BF16: the operands of the matrix product
 *                         are rounded to it, accumulation fp32); image fp32 or of that type. */
int gf_stem_conv7x7_dt(const void* image, int image_dtype, const float* weight, const float* shift, void* out, int out_dtype, int N,
                       int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * K8  fine matching
 * replaces FineMatching2.forward + get_fine_match (model/fine_matching2.py:21-126), M > 0 branch
 *   f0, f1 [M,25,C]; fine_matrix fp32 [M,25,25]; compacted mkpts0_f/mkpts1_f [Mf,2], mconf [Mf],
 *   m_bids int64 [Mf]; count int32 [1] = Mf (device).
 * ------------------------------------------------------------------------------------------ */
size_t gf_fine_match_workspace_bytes(int M);
int gf_fine_match(const void* f0, const void* f1, int dtype, int M, int WW, int C, float temperature, float thr,
                  const int64_t* b_ids, const float* mkpts0_c, const float* mkpts1_c, float coarse_scale,
                  float c2f_scale, float fine_scale, const float* scale0, const float* scale1,
                  float* fine_matrix, float* mkpts0_f, float* mkpts1_f, float* mconf, int64_t* m_bids,
                  int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOFORMER_HIP_H_ */
