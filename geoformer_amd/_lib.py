"""ctypes binding of libgeoformer_hip.so (the C ABI declared in include/geoformer_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `python -m geoformer_amd.build`.
There is NO fallback: if the shared object is missing or a symbol is absent, importing an op
raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GF_LIB_PATH') or os.path.join(_HERE, 'libgeoformer_hip.so')   # GF_LIB_PATH: A/B timing of two builds (tools/)

GF_F32, GF_F16, GF_BF16 = 0, 1, 2

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
c_long, c_uint32 = ctypes.c_long, ctypes.c_uint32

# name -> (restype, argtypes).  Kept in the order of include/geoformer_hip.h;
# tests/test_abi.py checks this table against the header.
SIGNATURES = {
    'gf_abi_version': (c_int, []),
    'gf_last_error': (ctypes.c_char_p, []),
    'gf_profile_filter': (None, [ctypes.c_char_p]),
    'gf_profile_enable': (None, [c_int]),
    'gf_profile_collect': (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int),
                                   ctypes.POINTER(ctypes.c_double)]),
    'gf_dual_softmax_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'gf_dual_softmax_match_only_supported': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    'gf_dual_softmax_conf_at': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_dual_softmax_match': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_float, c_float, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    'gf_linear_wgrad_workspace_bytes': (c_size_t, [c_long, c_int, c_int]),
    'gf_linear_wgrad': (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_long, c_int, c_int, c_void_p, c_long, c_int, c_void_p, c_size_t,
                                c_void_p]),
    'gf_layernorm_forward': (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    'gf_layernorm_backward_workspace_bytes': (c_size_t, [c_int]),
    'gf_layernorm_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_size_t, c_void_p]),
    'gf_activation_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'gf_fine_match_backward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'gf_linear_attention_backward_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'gf_window_linear_attention_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                                    c_void_p, c_void_p]),
    'gf_linear_attention_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long,
                                             c_long, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_conv3x3_wgrad_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'gf_conv3x3_wgrad_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    'gf_full_attention_train_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_long,
                                                c_float, c_void_p, c_long, c_void_p, c_void_p]),
    'gf_full_attention_backward_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'gf_full_attention_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_long, c_long, c_long, c_long, c_long, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    'gf_coarse_loss_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'gf_coarse_loss_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_coarse_loss_backward': (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_pos_encode': (c_int, [c_void_p, c_int, c_long, c_long, c_long, c_long, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_int, c_int, c_void_p]),
    'gf_linear_attention_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'gf_linear_attention': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_long,
                                    c_long, c_long, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    'gf_encoder_kv_workspace_bytes': (c_size_t, [c_int, c_int]),
    'gf_encoder_kv_state': (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_encoder_layer': (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_float,
                                 c_float, c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_int, c_int, c_void_p]),
    'gf_encoder_layer_kv': (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_float, c_void_p, c_void_p, c_float, c_float, c_int,
                                    c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_fine_layer': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p]),
    'gf_conv3x3_supported': (c_int, [c_int, c_int]),
    'gf_conv3x3s2_supported': (c_int, [c_int, c_int]),
    'gf_lateral_supported': (c_int, [c_int, c_int]),
    'gf_lateral_upsample_add_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'gf_conv3x3_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_float, c_int, c_void_p]),
    'gf_linear': (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                          c_void_p, c_void_p, c_float, c_void_p, c_long, c_void_p, c_int, c_void_p, c_long, c_int, c_int,
                          c_int, c_void_p]),
    'gf_ransac_workspace_bytes': (c_size_t, [c_int, c_int]),
    'gf_ransac_homography': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_float,
                                     c_int, c_uint32, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    'gf_ransac_homography_v2': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_float,
                                        c_int, c_uint32, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    'gf_window_geometry': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'gf_inlier_index': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'gf_self_attention_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'gf_self_attention_gathered': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_long,
                                           c_long, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p,
                                           c_size_t, c_void_p]),
    'gf_window_cross_attention': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_long, c_long, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'gf_window_cross_attention_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                                   c_long, c_long, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'gf_window_cross_attention_backward_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'gf_window_cross_attention_backward_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_long,
                                                          c_long, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                          c_void_p, c_size_t, c_void_p]),
    'gf_window_cross_attention_tiled': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                c_int, c_long, c_long, c_long, c_void_p, c_int, c_void_p, c_void_p,
                                                c_void_p]),
    'gf_fine_gather': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                               c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'gf_bias_act_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_int, c_void_p]),
    'gf_upsample_add_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'gf_upsample_bilinear_backward_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'gf_conv1x1_upsample_add_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_int, c_void_p]),
    'gf_conv1x1_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'gf_stem_conv7x7': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'gf_stem_conv7x7_dt': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'gf_fine_match_workspace_bytes': (c_size_t, [c_int]),
    'gf_fine_match': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                              c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}


ABI_VERSION = 4          # include/geoformer_hip.h GF_ABI_VERSION (4: the conv3x3 stream deals the output channels for the register epilogue)


class GeoFormerHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GeoFormerHipError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  geoformer_amd has no CPU / PyTorch fallback.')
        # torch first: its wheel ships its own HIP runtime (libamdhip64), and whichever copy is loaded FIRST is the one this library's
        # launches go through - loaded before torch, the library binds /opt/rocm's runtime and its first launch fails with "no
        # ROCm-capable device is detected" once torch has initialised the device through the other copy (seen with
        # `python __graft_entry__.py smoke`, where build() loads the library before anything imports torch)
        import torch  # noqa: F401
        # PyDLL: the interpreter lock is NOT released around a call.  Every entry point enqueues and returns in microseconds (the ones that wait -
        # gf_profile_collect, the gf_debug_* trace readers - are diagnostics); with CDLL each of the ~113 launches of a batch-1 forward released and
        # re-took the lock, and two host threads feeding two streams fell into a convoy: 230 pairs/s on two pipelines where one stream alone gives
        # 318 and the lock-holding form 530 (tools/b1_pipeline_probe.py, round 6).  A thread now gives the interpreter up where torch does - at its
        # device synchronisations - which is exactly when the other pipeline should be enqueueing.
        h = ctypes.PyDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)       # AttributeError if the symbol is missing: fail loudly
            fn.restype, fn.argtypes = res, args
        if h.gf_abi_version() != ABI_VERSION:
            raise GeoFormerHipError(f'{LIB_PATH} has ABI version {h.gf_abi_version()}, this binding was written against {ABI_VERSION} '
                                    '(include/geoformer_hip.h GF_ABI_VERSION): rebuild the library')
        _lib = h
    return _lib


def check(status, what):
    if status != 0:
        raise GeoFormerHipError(f'{what} failed ({status}): {lib().gf_last_error().decode()}')
