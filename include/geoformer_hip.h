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
 *   - dtype: GF_F32 (parity mode, exact-fp32 MFMA) or GF_F16 (fp16 storage, fp32 accumulate);
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
typedef enum { GF_F32 = 0, GF_F16 = 1 } gf_dtype;

int gf_abi_version(void);
const char* gf_last_error(void);

/* ------------------------------------------------------------------------------------------
 * K1  dual-softmax correlation + mutual-nearest match extraction
 * replaces CoarseMatching.forward + get_coarse_match
 *          (model/loftr_src/loftr/utils/coarse_matching.py:90-130, :132-212)
 *
 *   sim  = <f0[n,i,:], f1[n,j,:]> / C / temperature       (-1e9 where !(mask0[n,i] & mask1[n,j]))
 *   conf = softmax(sim, dim=1) * softmax(sim, dim=2)       -> conf [N,L,S] fp32 (always written)
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
 * ------------------------------------------------------------------------------------------ */
size_t gf_dual_softmax_workspace_bytes(int N, int L, int S);
int gf_dual_softmax_match(const void* f0, const void* f1, int dtype, int N, int L, int S, int C,
                          const uint8_t* mask0, const uint8_t* mask1, float temperature, float thr,
                          int force_one, int w0c, int w1c, float scale, const float* scale0,
                          const float* scale1, float* conf, int64_t* b_ids, int64_t* i_ids,
                          int64_t* j_ids, float* mconf, float* mkpts0_c, float* mkpts1_c,
                          int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOFORMER_HIP_H_ */
