"""Thin torch-tensor front end of the C ABI: pointer plumbing, workspace caching, stream passing.

Every function enqueues on torch's current stream and returns device tensors; data-dependent
counts stay on the device (`counts` tensors) until a caller needs their value.
"""
import ctypes

import torch

from . import _lib
from ._lib import GF_BF16, GF_F16, GF_F32, check

_DTYPES = {torch.float32: GF_F32, torch.float16: GF_F16, torch.bfloat16: GF_BF16}


def _dt(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f'geoformer_amd kernels take float32, float16 or bfloat16 tensors, got {t.dtype}') from None


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream_handle(device=None):
    """torch's current HIP stream (of `device`, default: the current device) as an integer handle.  Through the two C-level getters when
    this torch build has them: torch.cuda.current_stream() costs ~8 us of Python per call - a millisecond per batch-1 pair at 113 launches."""
    if _raw_stream is not None and _raw_device is not None:
        idx = _raw_device() if device is None else (device.index if isinstance(device, torch.device) else device)
        if isinstance(idx, int):
            return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def _stream():
    return ctypes.c_void_p(_stream_handle())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GeoFormerHipError('geoformer_amd ops need CUDA(HIP) tensors; there is no CPU path')


class _Workspaces:
    """One growing byte buffer per (device, tag, stream)."""

    def __init__(self):
        self.bufs = {}

    def get(self, tag, nbytes, device):
        key = (device, tag, _stream_handle(device))
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self.bufs[key] = b
        return b


_ws = _Workspaces()


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


def match_only_supported(f0, f1, masked=False, force_one=False):
    """Is the match-only mode of K1 (conf_matrix not materialised) built for these operands?"""
    return bool(_lib.lib().gf_dual_softmax_match_only_supported(_dt(f0), f0.shape[1], f1.shape[1], f0.shape[2], int(masked), int(bool(force_one))))


def dual_softmax_conf_at(f0, f1, temperature, b, i, j):
    """Single entries conf[b, i, j] (fp32 [P]) recomputed bit-identically to the matrix gf_dual_softmax_match writes, from the same
    features and the statistics its last call on this device and stream left in the workspace (match-only mode's way to a few
    values).  The workspace is stamped by that call with (N, L, S, C, temperature, the two feature pointers): when the stamp
    does not belong to THESE arguments - another pair of feature tensors was matched since (GeoFormer.forward runs two
    CoarseMatching passes on one workspace), the shape changed - or an index is out of range, the entry comes back NaN.  The stamp also
    carries a fingerprint of the features' CONTENT (64 sampled 8-byte words, 512 bytes in all): a new tensor that the caching allocator
    placed at a freed tensor's address does not pass for it.  It is a SAMPLED check: an in-place edit of the stamped tensors that misses
    the sampled words (one masked row, say) still passes - after editing the features in place, call dual_softmax_match again before
    asking for entries.  The features must be contiguous (the very tensors the match call got): a copy made here
    would never carry the stamped pointers."""
    _need_cuda(f0, f1)
    if not (f0.is_contiguous() and f1.is_contiguous()):
        raise ValueError('dual_softmax_conf_at: pass the contiguous feature tensors gf_dual_softmax_match was called with')
    N, L, C = f0.shape
    S = f1.shape[1]
    b, i, j = (_contig(t.to(device=f0.device, dtype=torch.int64)) for t in (b, i, j))
    out = torch.empty(b.numel(), dtype=torch.float32, device=f0.device)
    L_ = _lib.lib()
    ws = _ws.get('k1', L_.gf_dual_softmax_workspace_bytes(N, L, S), f0.device)
    check(L_.gf_dual_softmax_conf_at(_p(f0), _p(f1), _dt(f0), N, L, S, C, float(temperature), _p(b), _p(i), _p(j), b.numel(), _p(out),
                                     _p(ws), ws.numel(), _stream()), 'gf_dual_softmax_conf_at')
    return out


def dual_softmax_match(f0, f1, temperature, thr, hw0_c, hw1_c, scale, mask0=None, mask1=None, scale0=None,
                       scale1=None, force_one=False, materialize=True):
    """K1.  f0 [N,L,C], f1 [N,S,C] -> dict(conf [N,L,S] fp32, b_ids/i_ids/j_ids int64 [cap], mconf [cap],
    mkpts0_c/mkpts1_c [cap,2], counts int32 [1+N]) - all on the device, match arrays at capacity.
    materialize=False: match-only mode (conf_matrix is None; everything else bit-identical), where the library is built for it
    (match_only_supported) - otherwise the matrix is materialised as usual."""
    _need_cuda(f0, f1)
    f0, f1 = _contig(f0), _contig(f1)
    N, L, C = f0.shape
    S = f1.shape[1]
    dev = f0.device
    cap = N * min(L, S) + (N if force_one else 0)
    match_only = not materialize and match_only_supported(f0, f1, mask0 is not None, force_one)
    conf = None if match_only else torch.empty(N, L, S, dtype=torch.float32, device=dev)
    ids = torch.empty(3, cap, dtype=torch.int64, device=dev)
    mconf = torch.empty(cap, dtype=torch.float32, device=dev)
    mk = torch.empty(2, cap, 2, dtype=torch.float32, device=dev)
    counts = torch.empty(1 + N, dtype=torch.int32, device=dev)
    m0 = m1 = None
    if mask0 is not None:
        m0 = _contig(mask0.reshape(N, L).to(torch.uint8))
        m1 = _contig(mask1.reshape(N, S).to(torch.uint8))
    s0 = None if scale0 is None else _contig(scale0.to(device=dev, dtype=torch.float32))
    s1 = None if scale1 is None else _contig(scale1.to(device=dev, dtype=torch.float32))
    L_ = _lib.lib()
    nbytes = L_.gf_dual_softmax_workspace_bytes(N, L, S)
    ws = _ws.get('k1', nbytes, dev)
    check(L_.gf_dual_softmax_match(_p(f0), _p(f1), _dt(f0), N, L, S, C, _p(m0), _p(m1), float(temperature), float(thr),
                                   int(bool(force_one)), int(hw0_c[1]), int(hw1_c[1]), float(scale), _p(s0), _p(s1),
                                   _p(conf), _p(ids[0]), _p(ids[1]), _p(ids[2]), _p(mconf), _p(mk[0]), _p(mk[1]),
                                   _p(counts), _p(ws), ws.numel(), _stream()), 'gf_dual_softmax_match')
    return {'conf_matrix': conf, 'b_ids': ids[0], 'i_ids': ids[1], 'j_ids': ids[2], 'mconf': mconf,
            'mkpts0_c': mk[0], 'mkpts1_c': mk[1], 'counts': counts}


def _i32p(t):
    return _p(t)


def pos_encode(x, pe_hwc, out_dtype, out=None):
    """a1.  x [N,C,H,W] (any strides, fp32/fp16), pe_hwc fp32 [H,W,C] on the device -> [N, H*W, C] (into `out` if given)."""
    _need_cuda(x, pe_hwc)
    N, C, H, W = x.shape
    if out is None:
        out = torch.empty(N, H * W, C, dtype=out_dtype, device=x.device)
    elif out.shape != (N, H * W, C) or out.dtype != out_dtype or not out.is_contiguous():
        raise ValueError('out must be a contiguous [N, H*W, C] tensor of out_dtype')
    sn, sc, sh, sw = x.stride()
    check(_lib.lib().gf_pos_encode(_p(x), _dt(x), sn, sc, sh, sw, _p(pe_hwc), _p(out), _DTYPES[out_dtype], N, C, H, W,
                                   _stream()), 'gf_pos_encode')
    return out


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _nhwc(t):
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('expected a channels_last [N,C,H,W] tensor')
    return t


def bias_act_(x, bias=None, residual=None, act=ACT_NONE, slope=0.01):
    """Backbone glue, in place: x <- act(x + bias[c] + residual) on channels_last [N,C,H,W] tensors."""
    _need_cuda(x)
    _nhwc(x)
    if residual is not None and (_nhwc(residual).shape != x.shape or residual.dtype != x.dtype):
        raise ValueError('residual must match x')
    N, C, H, W = x.shape
    check(_lib.lib().gf_bias_act_nhwc(_p(x), _p(bias), _p(residual), _p(x), N * H * W, C, int(act), float(slope), _dt(x),
                                      _stream()), 'gf_bias_act_nhwc')
    return x


def upsample_add_(hi, lo):
    """Backbone glue, in place: hi <- hi + bilinear(lo -> hi's size, align_corners=True); channels_last."""
    _need_cuda(hi, lo)
    _nhwc(hi), _nhwc(lo)
    N, C, H, W = hi.shape
    if lo.shape[:2] != (N, C) or lo.dtype != hi.dtype:
        raise ValueError('lo must have the batch, channels and dtype of hi')
    check(_lib.lib().gf_upsample_add_nhwc(_p(lo), _p(hi), _p(hi), N, lo.shape[2], lo.shape[3], H, W, C, _dt(hi),
                                          _stream()), 'gf_upsample_add_nhwc')
    return hi


def upsample_bilinear_backward(dhi, h, w):
    """Training: the gradient of F.interpolate(lo, size=(H, W), mode='bilinear', align_corners=True) with respect to lo [N, C, h, w], given
    dhi [N, C, H, W] channels_last (fp32 / fp16 / bf16, C % 4 == 0); a gather, bit-reproducible."""
    _need_cuda(dhi)
    _nhwc(dhi)
    N, C, H, W = dhi.shape
    dlo = torch.empty(N, C, int(h), int(w), dtype=dhi.dtype, device=dhi.device, memory_format=torch.channels_last)
    check(_lib.lib().gf_upsample_bilinear_backward_nhwc(_p(dhi), _p(dlo), N, int(h), int(w), H, W, C, _dt(dhi), _stream()),
          'gf_upsample_bilinear_backward_nhwc')
    return dlo


def conv1x1_upsample_add(x, weight, lo):
    """Backbone glue: 1x1 convolution of channels_last x [N,Cin,H,W] with weight [Cout,Cin(,1,1)] plus the bilinear
    (align_corners=True) upsampling of channels_last lo [N,Cout,h,w], in one K3 launch -> channels_last [N,Cout,H,W]."""
    _need_cuda(x, weight, lo)
    _nhwc(x), _nhwc(lo)
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    w2 = _contig(weight.reshape(Cout, Cin))
    if lo.shape[:2] != (N, Cout) or lo.dtype != x.dtype or w2.dtype != x.dtype:
        raise ValueError('lo must be [N, Cout, h, w] of the dtype of x and weight')
    out = torch.empty(N, Cout, H, W, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(_lib.lib().gf_conv1x1_upsample_add_nhwc(_p(x), _p(w2), _p(lo), _p(out), N, lo.shape[2], lo.shape[3], H, W, Cin, Cout,
                                                  _dt(x), _stream()), 'gf_conv1x1_upsample_add_nhwc')
    return out


def conv1x1(x, weight, stride=1):
    """Backbone glue: 1x1 convolution (no bias) of channels_last x [N,Cin,H,W] with weight [Cout,Cin(,1,1)], stride 1 or 2 ->
    channels_last [N,Cout,H/stride,W/stride]; one K3 launch on the (strided) pixel rows (gf_conv1x1_nhwc)."""
    _need_cuda(x, weight)
    _nhwc(x)
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    w2 = _contig(weight.reshape(Cout, Cin))
    if w2.dtype != x.dtype:
        raise ValueError('weight must have the dtype of x')
    out = torch.empty(N, Cout, H // stride, W // stride, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(_lib.lib().gf_conv1x1_nhwc(_p(x), _p(w2), _p(out), N, H, W, Cin, Cout, int(stride), _dt(x), _stream()), 'gf_conv1x1_nhwc')
    return out


def stem_conv7x7(image, weight, shift, dtype=torch.float16):
    """Backbone stem: image [N,1,H,W] (fp32 or `dtype`, contiguous), weight fp32 [128,1,7,7] (BN folded), shift fp32 [128]
    -> relu(conv 7x7 / stride 2 / pad 3 + shift), `dtype` (fp16 / bf16) channels_last [N,128,Ho,Wo]."""
    _need_cuda(image, weight, shift)
    N, one, H, W = image.shape
    if one != 1 or not image.is_contiguous():
        raise ValueError('stem_conv7x7 needs a contiguous [N,1,H,W] image')
    C = weight.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty(N, C, Ho, Wo, dtype=dtype, device=image.device, memory_format=torch.channels_last)
    check(_lib.lib().gf_stem_conv7x7_dt(_p(image), _dt(image), _p(weight), _p(shift), _p(out), _DTYPES[dtype], N, H, W, C, _stream()),
          'gf_stem_conv7x7_dt')
    return out


def _rows(t):
    """[..., C] tensor whose last dim is contiguous -> (tensor, row stride in elements)."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    return t, t.stride(-2)


def linear_attention(q, k, v, nhead, q_mask=None, kv_mask=None, eps=1e-6):
    """K2.  q [N,L,C], k,v [N,S,C] (row-strided views allowed) -> [N,L,C]."""
    _need_cuda(q, k, v)
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    q, ldq = _rows(q)
    k, ldk = _rows(k)
    v, ldv = _rows(v)
    for t, ld, n in ((q, ldq, L), (k, ldk, S), (v, ldv, S)):
        if t.stride(0) != ld * n:
            raise ValueError('linear_attention needs batch stride == rows * row stride')
    out = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    qm = None if q_mask is None else _contig(q_mask.reshape(N, L).to(torch.uint8))
    km = None if kv_mask is None else _contig(kv_mask.reshape(N, S).to(torch.uint8))
    L_ = _lib.lib()
    nbytes = L_.gf_linear_attention_workspace_bytes(N, S, nhead, D)
    ws = _ws.get('k2', nbytes, q.device)
    check(L_.gf_linear_attention(_p(q), _p(k), _p(v), _dt(q), N, L, S, nhead, D, ldq, ldk, ldv, _p(qm), _p(km),
                                 float(eps), _p(out), _p(ws), ws.numel(), _stream()), 'gf_linear_attention')
    return out


RANSAC_ITERS = 1024
RANSAC_SEED = 0x5EED
RANSAC_LM_ITERS = 10            # Levenberg-Marquardt steps behind the refit: what cv2.findHomography(..., cv2.RANSAC, ...) appends


def ransac_homography(mkpts0_c, mkpts1_c, counts, N, scale, scale0=None, scale1=None, thr=8.0, iters=RANSAC_ITERS,
                      seed=RANSAC_SEED, integer_keypoints=True, min_points=9, lm_iters=RANSAC_LM_ITERS):
    """Device RANSAC on the first-pass coarse matches.  Returns dict(kp0, kp1 fp32 [cap,2], M fp64 [N,3,3],
    M_f32, Minv_f32 [N,3,3], valid int32 [N], keep uint8 [cap])."""
    _need_cuda(mkpts0_c, mkpts1_c, counts)
    dev = mkpts0_c.device
    cap = mkpts0_c.shape[0]
    kp = torch.empty(2, max(cap, 1), 2, dtype=torch.float32, device=dev)
    M = torch.empty(N, 3, 3, dtype=torch.float64, device=dev)
    Mf = torch.empty(2, N, 3, 3, dtype=torch.float32, device=dev)
    valid = torch.empty(N, dtype=torch.int32, device=dev)
    keep = torch.empty(max(cap, 1), dtype=torch.uint8, device=dev)
    s0 = None if scale0 is None else _contig(scale0.to(device=dev, dtype=torch.float32))
    s1 = None if scale1 is None else _contig(scale1.to(device=dev, dtype=torch.float32))
    L_ = _lib.lib()
    nbytes = L_.gf_ransac_workspace_bytes(N, iters)
    ws = _ws.get('ransac', nbytes, dev)
    check(L_.gf_ransac_homography_v2(_p(mkpts0_c), _p(mkpts1_c), _p(counts), N, max(cap, 1), float(scale), _p(s0), _p(s1),
                                     float(thr), int(iters), int(seed), int(min_points), int(bool(integer_keypoints)), _p(kp[0]), _p(kp[1]),
                                     _p(M), _p(Mf[0]), _p(Mf[1]), _p(valid), _p(keep), _p(ws), ws.numel(), _stream(), int(lm_iters)),
          'gf_ransac_homography_v2')
    return {'kp0': kp[0], 'kp1': kp[1], 'M': M, 'M_f32': Mf[0], 'Minv_f32': Mf[1], 'valid': valid, 'keep': keep}


def window_geometry(H_f32, valid, grid_hw, img_hw, key_grid_w, scale=8, window_size=5, window_scale=None,
                    debug=False):
    """a8.  H_f32 [N,3,3] maps the query grid (grid_hw cells) into the image of size img_hw (pixels).
    -> win int32 [N, L, ws*ws] (+ kps int32 [N,L,ww,2], warped fp32 [N,L,2] when debug)."""
    _need_cuda(H_f32)
    N = H_f32.shape[0]
    hq, wq = grid_hw
    L = hq * wq
    ww = window_size * window_size
    dev = H_f32.device
    win = torch.empty(N, L, ww, dtype=torch.int32, device=dev)
    kps = torch.empty(N, L, ww, 2, dtype=torch.int32, device=dev) if debug else None
    warped = torch.empty(N, L, 2, dtype=torch.float32, device=dev) if debug else None
    wsc = None if window_scale is None else _contig(window_scale.to(device=dev, dtype=torch.float32))
    Hc = _contig(H_f32.reshape(N, 9).float())
    check(_lib.lib().gf_window_geometry(_p(Hc), _p(valid), N, hq, wq, int(img_hw[0]), int(img_hw[1]), int(key_grid_w),
                                        int(scale), int(window_size), _p(wsc), _p(win), _p(kps), _p(warped), _stream()),
          'gf_window_geometry')
    return (win, kps, warped) if debug else win


def inlier_index(kp0, kp1, keep, counts, N, L, S, w0, w1, scale=8):
    """a7.  -> dict(map0 uint8 [N,L], map1 [N,S], idx0 int32 [N,L], idx1 [N,S], nidx int32 [N,2])."""
    dev = kp0.device
    map0 = torch.empty(N, L, dtype=torch.uint8, device=dev)
    map1 = torch.empty(N, S, dtype=torch.uint8, device=dev)
    idx0 = torch.empty(N, L, dtype=torch.int32, device=dev)
    idx1 = torch.empty(N, S, dtype=torch.int32, device=dev)
    nidx = torch.empty(N, 2, dtype=torch.int32, device=dev)
    check(_lib.lib().gf_inlier_index(_p(kp0), _p(kp1), _p(keep), _p(counts), N, L, S, int(w0), int(w1), int(scale),
                                     _p(map0), _p(map1), _p(idx0), _p(idx1), _p(nidx), _stream()), 'gf_inlier_index')
    return {'map0': map0, 'map1': map1, 'idx0': idx0, 'idx1': idx1, 'nidx': nidx}


def self_attention_gathered(q, kmap, vmap, idx, nkeys, nhead=4):
    """K4.  q, kmap, vmap [N,L,256]; idx int32 [N,>=L]; nkeys int32 view with one entry per sample (any stride) -> [N,L,256].
    Row-strided views are taken as they are when the library can move their rows as 16-byte pieces (16-bit modes: a 16-byte aligned
    base and a row stride that is a multiple of 8 elements - e.g. the halves of a [N, L, 512] k|v projection); any other view is
    copied to a contiguous tensor first."""
    _need_cuda(q, kmap, vmap, idx, nkeys)
    N, L, C = q.shape

    def rows16(t):
        t, ld = _rows(t)
        if t.element_size() == 2 and (t.data_ptr() % 16 != 0 or ld % 8 != 0):
            t = t.contiguous()
            ld = t.shape[-1]
        return t, ld
    q, ldq = rows16(q)
    kmap, ldk = rows16(kmap)
    vmap, ldv = rows16(vmap)
    for t, ld in ((q, ldq), (kmap, ldk), (vmap, ldv)):
        if t.stride(0) != ld * L:
            raise ValueError('self_attention_gathered needs batch stride == L * row stride')
    out = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    L_ = _lib.lib()
    nbytes = L_.gf_self_attention_workspace_bytes(N, L, _dt(q))
    ws = _ws.get('k4', nbytes, q.device)
    check(L_.gf_self_attention_gathered(_p(q), _p(kmap), _p(vmap), _dt(q), N, L, nhead, C // nhead, ldq, ldk, ldv,
                                        _p(idx), idx.stride(0), _p(nkeys), nkeys.stride(0) if nkeys.dim() else 1,
                                        _p(out), _p(ws), ws.numel(), _stream()), 'gf_self_attention_gathered')
    return out


def window_cross_attention(q, kmap, vmap, win, valid=None, nhead=4, hw_q=None, hw_k=None):
    """K5.  q [N,L,256]; kmap, vmap [N,S,256]; win int32 [N,L,25] -> [N,L,256].  With the map shapes hw_q = (hq, wq),
    hw_k = (hk, wk) (L = hq*wq, S = hk*wk) the 16-bit modes run the tiled form (key / value rows of a query tile's windows
    staged in LDS once)."""
    _need_cuda(q, kmap, vmap, win)
    N, L, C = q.shape
    S = kmap.shape[1]
    q, ldq = _rows(q)
    kmap, ldk = _rows(kmap)
    vmap, ldv = _rows(vmap)
    if q.stride(0) != ldq * L or kmap.stride(0) != ldk * S or vmap.stride(0) != ldv * S:
        raise ValueError('window_cross_attention needs batch stride == rows * row stride')
    out = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    if hw_q is not None and hw_k is not None and ldq % 8 == 0 and ldk % 8 == 0 and ldv % 8 == 0:
        if hw_q[0] * hw_q[1] != L or hw_k[0] * hw_k[1] != S:
            raise ValueError('window_cross_attention: map shapes do not match the token counts')
        check(_lib.lib().gf_window_cross_attention_tiled(_p(q), _p(kmap), _p(vmap), _dt(q), N, hw_q[0], hw_q[1], hw_k[0],
                                                         hw_k[1], nhead, C // nhead, ldq, ldk, ldv, _p(win), win.shape[-1],
                                                         _p(valid), _p(out), _stream()), 'gf_window_cross_attention_tiled')
        return out
    check(_lib.lib().gf_window_cross_attention(_p(q), _p(kmap), _p(vmap), _dt(q), N, L, S, nhead, C // nhead, ldq, ldk,
                                               ldv, _p(win), win.shape[-1], _p(valid), _p(out), _stream()),
          'gf_window_cross_attention')
    return out


def window_inverse_index(win, S):
    """The inverse of the window table win int32 [N, L, WW] (cell of the other image per window position, < 0 = masked) for the gather form
    of K5's backward: (entries int32 [N*L*WW] = l*WW + k sorted - stably - by global cell n*S + cell, masked positions last;
    offsets int32 [N*S + 1]).  torch ops only, no host synchronisation; one sort of N*L*WW keys."""
    N, L, WW = win.shape
    base = (torch.arange(N, device=win.device, dtype=torch.int64) * S)[:, None, None]
    key = torch.where(win >= 0, win.to(torch.int64) + base, N * S).reshape(-1)
    key, order = torch.sort(key, stable=True)
    entries = (order % (L * WW)).to(torch.int32)
    counts = torch.bincount(key, minlength=N * S + 1)[:N * S]
    offsets = torch.zeros(N * S + 1, dtype=torch.int32, device=win.device)
    offsets[1:] = counts.cumsum(0).to(torch.int32)
    return entries, offsets


def window_cross_attention_backward_gather(q, kmap, vmap, dout, win, index, nhead=4):
    """(dq [N,L,256], dk, dv [N,S,256], all of q's dtype) of window_cross_attention given dout [N,L,256], the cell sums gathered along
    `index` = window_inverse_index(win, S): no atomics, bit-reproducible."""
    _need_cuda(q, kmap, vmap, dout, win)
    N, L, C = q.shape
    S = kmap.shape[1]
    q, ldq = _rows(q)
    kmap, ldk = _rows(kmap)
    vmap, ldv = _rows(vmap)
    if q.stride(0) != ldq * L or kmap.stride(0) != ldk * S or vmap.stride(0) != ldv * S:
        raise ValueError('window_cross_attention_backward needs batch stride == rows * row stride')
    dout = _contig(dout)
    entries, offsets = index
    dq = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    dkv = torch.empty(2, N, S, C, dtype=q.dtype, device=q.device)
    L_ = _lib.lib()
    ws = _ws.get('k5bwd', L_.gf_window_cross_attention_backward_workspace_bytes(N, L, win.shape[-1]), q.device)
    check(L_.gf_window_cross_attention_backward_gather(_p(q), _p(kmap), _p(vmap), _p(dout), _dt(q), N, L, S, nhead, C // nhead, ldq, ldk, ldv,
                                                       _p(win), win.shape[-1], _p(entries), _p(offsets), _p(dq), _p(dkv[0]), _p(dkv[1]),
                                                       _p(ws), ws.numel(), _stream()), 'gf_window_cross_attention_backward_gather')
    return dq, dkv[0], dkv[1]


def window_cross_attention_backward(q, kmap, vmap, dout, win, nhead=4):
    """(dq [N,L,256] of q's dtype, dk, dv fp32 [N,S,256]) of window_cross_attention given dout [N,L,256]."""
    _need_cuda(q, kmap, vmap, dout, win)
    N, L, C = q.shape
    S = kmap.shape[1]
    q, ldq = _rows(q)
    kmap, ldk = _rows(kmap)
    vmap, ldv = _rows(vmap)
    if q.stride(0) != ldq * L or kmap.stride(0) != ldk * S or vmap.stride(0) != ldv * S:
        raise ValueError('window_cross_attention_backward needs batch stride == rows * row stride')
    dout = _contig(dout)
    dq = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    dkv = torch.zeros(2, N, S, C, dtype=torch.float32, device=q.device)
    check(_lib.lib().gf_window_cross_attention_backward(_p(q), _p(kmap), _p(vmap), _p(dout), _dt(q), N, L, S, nhead, C // nhead, ldq, ldk, ldv,
                                                        _p(win), win.shape[-1], _p(dq), _p(dkv[0]), _p(dkv[1]), _stream()),
          'gf_window_cross_attention_backward')
    return dq, dkv[0], dkv[1]


def fine_gather(feat_f0, feat_f1, feat_c0, feat_c1, b_ids, i_ids, j_ids, w0c, w1c, stride, window, out_dtype):
    """K7.  feat_f* [N,Cf,H,W] any strides; feat_c* [N,L,CC] contiguous of out_dtype; ids int64 [M] (M > 0)
    -> (win [2M, W*W, Cf], ccat [2M, CC])."""
    _need_cuda(feat_f0, feat_f1, feat_c0, feat_c1, b_ids)
    M = b_ids.shape[0]
    Cf = feat_f0.shape[1]
    CC = feat_c0.shape[-1]
    dev = feat_f0.device
    win = torch.empty(2 * M, window * window, Cf, dtype=out_dtype, device=dev)
    ccat = torch.empty(2 * M, CC, dtype=out_dtype, device=dev)
    s0 = (ctypes.c_long * 4)(*feat_f0.stride())
    s1 = (ctypes.c_long * 4)(*feat_f1.stride())
    fc0, fc1 = _contig(feat_c0), _contig(feat_c1)
    if fc0.dtype != out_dtype or fc1.dtype != out_dtype or feat_f0.dtype != feat_f1.dtype:
        raise TypeError('fine_gather: coarse features must already be in out_dtype; fine maps must share a dtype')
    check(_lib.lib().gf_fine_gather(_p(feat_f0), _p(feat_f1), _dt(feat_f0), ctypes.cast(s0, ctypes.c_void_p),
                                    ctypes.cast(s1, ctypes.c_void_p), feat_f0.shape[2], feat_f0.shape[3], feat_f1.shape[2],
                                    feat_f1.shape[3], Cf, _p(fc0), _p(fc1), _DTYPES[out_dtype], fc0.shape[1], fc1.shape[1],
                                    CC, _p(_contig(b_ids)), _p(_contig(i_ids)), _p(_contig(j_ids)), M, int(w0c), int(w1c),
                                    int(stride), int(window), _p(win), _p(ccat), _stream()), 'gf_fine_gather')
    return win, ccat


def fine_match(f0, f1, temperature, thr, b_ids, mkpts0_c, mkpts1_c, coarse_scale, c2f_scale, fine_scale, scale0=None,
               scale1=None):
    """K8.  f0, f1 [M,25,C] (M > 0) -> dict(fine_matrix [M,25,25], mkpts0_f/mkpts1_f [M,2] (first count rows
    valid), mconf [M], m_bids int64 [M], count int32 [1])."""
    _need_cuda(f0, f1)
    f0, f1 = _contig(f0), _contig(f1)
    M, WW, C = f0.shape
    dev = f0.device
    fm = torch.empty(M, WW, WW, dtype=torch.float32, device=dev)
    mk = torch.empty(2, M, 2, dtype=torch.float32, device=dev)
    mconf = torch.empty(M, dtype=torch.float32, device=dev)
    mb = torch.empty(M, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    s0 = None if scale0 is None else _contig(scale0.to(device=dev, dtype=torch.float32))
    s1 = None if scale1 is None else _contig(scale1.to(device=dev, dtype=torch.float32))
    L_ = _lib.lib()
    nbytes = L_.gf_fine_match_workspace_bytes(M)
    ws = _ws.get('k8', nbytes, dev)
    check(L_.gf_fine_match(_p(f0), _p(f1), _dt(f0), M, WW, C, float(temperature), float(thr), _p(_contig(b_ids)),
                           _p(_contig(mkpts0_c)), _p(_contig(mkpts1_c)), float(coarse_scale), float(c2f_scale),
                           float(fine_scale), _p(s0), _p(s1), _p(fm), _p(mk[0]), _p(mk[1]), _p(mconf), _p(mb), _p(count),
                           _p(ws), ws.numel(), _stream()), 'gf_fine_match')
    return {'fine_matrix': fm, 'mkpts0_f': mk[0], 'mkpts1_f': mk[1], 'mconf': mconf, 'm_bids': mb, 'count': count}


EPI_NONE, EPI_RELU, EPI_TANH, EPI_LN, EPI_LN_RES = 0, 1, 2, 3, 4


def _rows2d(t):
    """[..., K] -> (tensor viewed as rows, row stride); last dim contiguous and rows uniformly strided."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    lead = t.shape[:-1]
    ld = t.stride(-2) if t.dim() > 1 else t.shape[-1]
    # every leading dim must collapse onto one row stride
    expect = ld
    for size, stride in zip(reversed(lead), reversed(t.stride()[:-1])):
        if size != 1 and stride != expect:
            t = t.contiguous()
            ld = t.shape[-1]
            break
        expect *= size
    return t, ld


def linear(a1, w, a2=None, bias=None, rowgroup_bias=None, rowgroup_rows=0, epilogue=EPI_NONE, ln=None, eps=1e-5,
           residual=None, row_flag=None, flag_rows=0, out=None):
    """K3.  out[..., n] = epi([a1|a2] @ w.T + bias + rowgroup_bias); see include/geoformer_hip.h (gf_linear).
    ln = (gamma fp32 [N], beta fp32 [N]) for the LayerNorm epilogues."""
    _need_cuda(a1, w)
    a1, lda1 = _rows2d(a1)
    k1 = a1.shape[-1]
    k2, lda2 = 0, 0
    if a2 is not None:
        a2, lda2 = _rows2d(a2)
        k2 = a2.shape[-1]
    N = w.shape[0]
    M = a1.numel() // k1
    if out is None:
        out = torch.empty(*a1.shape[:-1], N, dtype=a1.dtype, device=a1.device)
    elif out.shape != (*a1.shape[:-1], N) or out.dtype != a1.dtype or not out.is_contiguous():
        raise ValueError('out must be a contiguous tensor of the result shape and dtype')
    res, ldres = (None, 0) if residual is None else _rows2d(residual)
    g, b = (None, None) if ln is None else ln
    check(_lib.lib().gf_linear(_p(a1), lda1, k1, _p(a2), lda2, k2, _p(w), _p(bias), _p(rowgroup_bias), int(rowgroup_rows),
                               int(epilogue), _p(g), _p(b), float(eps), _p(res), ldres, _p(row_flag), int(flag_rows), _p(out),
                               N, _dt(a1), M, N, _stream()), 'gf_linear')
    return out


# ---------------------------------------------------------------------------------------------
# training: fused dual-softmax focal loss (forward + backward in HIP, nothing L x S is materialised)
# ---------------------------------------------------------------------------------------------
class CoarseFocalLoss(torch.autograd.Function):
    """sum_k -alpha (1 - p_k)^gamma log p_k * w_k over the positives k = (b, i, j), p = the dual-softmax confidence
    of (f0, f1) at temperature `temperature` (loftr_loss.py:246-270 on coarse_matching.py:113-125).  Returns
    (loss_sum, p [P]); gradients flow to f0 and f1 only."""

    @staticmethod
    def forward(ctx, f0, f1, pos_b, pos_i, pos_j, temperature, alpha, gamma, weight, mask0=None, mask1=None):
        _need_cuda(f0, f1, pos_b)
        N, L, C = f0.shape
        S = f1.shape[1]
        P = pos_b.numel()
        ctx_dtype = f0.dtype                       # the gradients go back in the caller's dtype
        if f0.dtype == torch.bfloat16:           # the mixed-bf16 training step: the kernels take fp16 operands (same 16-bit
            f0, f1 = f0.to(torch.float16), f1.to(torch.float16)       # storage, 3 more mantissa bits; features are O(1))
        f0c, f1c = _contig(f0), _contig(f1)
        if f0c.dtype not in (torch.float32, torch.float16) or f1c.dtype != f0c.dtype:
            raise ValueError('f0/f1 must both be fp32, fp16 or bf16')
        L_ = _lib.lib()
        ws = torch.empty(L_.gf_coarse_loss_workspace_bytes(N, L, S), dtype=torch.uint8, device=f0.device)
        conf = torch.empty(P, dtype=torch.float32, device=f0.device)
        loss, grad = torch.empty_like(conf), torch.empty_like(conf)
        pb, pi, pj = _contig(pos_b.long()), _contig(pos_i.long()), _contig(pos_j.long())
        w = None if weight is None else _contig(weight.float())
        m0 = None if mask0 is None else _contig(mask0.reshape(N, L).to(torch.uint8))
        m1 = None if mask1 is None else _contig(mask1.reshape(N, S).to(torch.uint8))
        ctx.masks = (m0, m1)
        check(L_.gf_coarse_loss_forward(_p(f0c), _p(f1c), _dt(f0c), N, L, S, C, _p(m0), _p(m1), float(temperature), _p(pb), _p(pi), _p(pj), P, _p(w),
                                        float(alpha), float(gamma), _p(conf), _p(loss), _p(grad), _p(ws), ws.numel(), _stream()),
              'gf_coarse_loss_forward')
        ctx.save_for_backward(pb, pi, pj, grad, ws)
        ctx.meta = (N, L, S, C, float(temperature), ctx_dtype)
        ctx.mark_non_differentiable(conf)
        return loss.sum(), conf

    @staticmethod
    def backward(ctx, g_loss, _g_conf):
        pb, pi, pj, grad, ws = ctx.saved_tensors
        N, L, S, C, temperature, dtype = ctx.meta
        d0 = torch.empty(N, L, C, dtype=torch.float32, device=grad.device)
        d1 = torch.empty(N, S, C, dtype=torch.float32, device=grad.device)
        m0, m1 = ctx.masks
        gl = _contig(g_loss.detach().reshape(1).to(device=grad.device, dtype=torch.float32))   # stays on the device: no host sync
        check(_lib.lib().gf_coarse_loss_backward(N, L, S, C, _p(m0), _p(m1), temperature, _p(pb), _p(pi), _p(pj), pb.numel(), _p(grad),
                                                 1.0, _p(gl), _p(d0), _p(d1), _p(ws), ws.numel(), _stream()),
              'gf_coarse_loss_backward')
        return d0.to(dtype), d1.to(dtype), None, None, None, None, None, None, None, None, None


def coarse_focal_loss(f0, f1, pos_b, pos_i, pos_j, temperature, alpha=0.25, gamma=2.0, weight=None, mask0=None, mask1=None):
    return CoarseFocalLoss.apply(f0, f1, pos_b, pos_i, pos_j, temperature, alpha, gamma, weight, mask0, mask1)


# ---------------------------------------------------------------------------------------------
# training: backward of the K3 chain (linear, LayerNorm, activation) - see csrc/k_train.hip
# ---------------------------------------------------------------------------------------------
def linear_wgrad(dy, x, out=None, accumulate=False):
    """dW [cout, cin] fp32 (+)= dy^T x over all leading dimensions; dy [..., cout], x [..., cin] 16-bit (row-strided views allowed).
    `out` may be a column block of a wider gradient (row stride > cin): the two halves of torch.cat([x, m]) write theirs."""
    _need_cuda(dy, x)
    dy, lddy = _rows2d(dy)
    x, ldx = _rows2d(x)
    cout, cin = dy.shape[-1], x.shape[-1]
    T = dy.numel() // cout
    if out is None:
        out = torch.empty(cout, cin, dtype=torch.float32, device=dy.device)
    L_ = _lib.lib()
    ws = _ws.get('wgrad', L_.gf_linear_wgrad_workspace_bytes(T, cout, cin), dy.device)
    check(L_.gf_linear_wgrad(_p(dy), lddy, _p(x), ldx, _dt(dy), T, cout, cin, _p(out), out.stride(0), int(bool(accumulate)), _p(ws), ws.numel(),
                             _stream()), 'gf_linear_wgrad')
    return out


def layernorm_forward(y, gamma, beta, eps=1e-5):
    """(LayerNorm(y) in y's 16-bit dtype, stats fp32 [T, 2] = (mean, rstd)); gamma, beta fp32."""
    _need_cuda(y, gamma)
    y = _contig(y)
    C = y.shape[-1]
    T = y.numel() // C
    out = torch.empty_like(y)
    stats = torch.empty(T, 2, dtype=torch.float32, device=y.device)
    check(_lib.lib().gf_layernorm_forward(_p(y), _dt(y), T, C, _p(gamma), _p(beta), float(eps), _p(out), _p(stats), _stream()),
          'gf_layernorm_forward')
    return out, stats


def layernorm_backward(dout, y, stats, gamma):
    """(dy in y's dtype, dgamma fp32 [C], dbeta fp32 [C])."""
    _need_cuda(dout, y)
    dout, y = _contig(dout), _contig(y)
    C = y.shape[-1]
    T = y.numel() // C
    dy = torch.empty_like(y)
    dg = torch.empty(2, C, dtype=torch.float32, device=y.device)
    L_ = _lib.lib()
    ws = _ws.get('lnbwd', L_.gf_layernorm_backward_workspace_bytes(C), y.device)
    check(L_.gf_layernorm_backward(_p(dout), _p(y), _p(stats), _dt(y), T, C, _p(gamma), _p(dy), _p(dg[0]), _p(dg[1]), 0, _p(ws), ws.numel(),
                                   _stream()), 'gf_layernorm_backward')
    return dy, dg[0], dg[1]


def activation_backward(dh, h, kind):
    """dz = dh * act'(z) from the activation's output h (kind 'relu' | 'tanh'); 16-bit tensors."""
    _need_cuda(dh, h)
    dh, h = _contig(dh), _contig(h)
    dz = torch.empty_like(h)
    check(_lib.lib().gf_activation_backward(_p(dh), _p(h), _p(dz), h.numel(), {'relu': 0, 'tanh': 1}[kind], _dt(h), _stream()),
          'gf_activation_backward')
    return dz


def linear_attention_backward(q, k, v, dout, nhead, q_mask=None, kv_mask=None, eps=1e-6):
    """(dq [N,L,C], dk, dv [N,S,C]) of K2's linear attention given dout [N,L,C]; q [N,L,C], k, v [N,S,C] (row-strided views allowed),
    heads of 32 channels, 16-bit tensors."""
    _need_cuda(q, k, v, dout)
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    q, ldq = _rows(q)
    k, ldk = _rows(k)
    v, ldv = _rows(v)
    dout, ldo = _rows(dout)
    for t, ld, n in ((q, ldq, L), (dout, ldo, L), (k, ldk, S), (v, ldv, S)):
        if t.stride(0) != ld * n:
            raise ValueError('linear_attention_backward needs batch stride == rows * row stride')
    dq = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    dkv = torch.empty(2, N, S, C, dtype=q.dtype, device=q.device)
    qm = None if q_mask is None else _contig(q_mask.reshape(N, L).to(torch.uint8))
    km = None if kv_mask is None else _contig(kv_mask.reshape(N, S).to(torch.uint8))
    L_ = _lib.lib()
    ws = _ws.get('k2bwd', L_.gf_linear_attention_backward_workspace_bytes(N, L, S, nhead), q.device)
    check(L_.gf_linear_attention_backward(_p(q), _p(k), _p(v), _p(dout), _dt(q), N, L, S, nhead, D, ldq, ldk, ldv, ldo, _p(qm), _p(km), float(eps),
                                          _p(dq), _p(dkv[0]), _p(dkv[1]), _p(ws), ws.numel(), _stream()), 'gf_linear_attention_backward')
    return dq, dkv[0], dkv[1]


def _rows16(t, rows):
    """A [N, rows, C] 16-bit tensor as the training attention kernels take it: last dim contiguous, 16-byte aligned rows with a stride
    that is a multiple of 8 elements, batch stride = rows x row stride; anything else is copied."""
    if (t.stride(-1) != 1 or t.data_ptr() % 16 or t.stride(-2) % 8 or t.stride(-2) < t.shape[-1] or
            (t.shape[0] > 1 and t.stride(0) != rows * t.stride(-2))):
        t = t.contiguous()
    return t, t.stride(-2)


def full_attention_train_forward(q, k, v, nhead):
    """K4 (training): softmax(q k^T / sqrt(D)) v of FullAttention (geo_attention.py:72-101, no masks) on q [N,L,256], k, v [N,S,256] 16-bit
    tensors with 4 heads of 64 -> (out [N,L,256], lse fp32 [N,4,L] = base-2 log-sum-exp of the scaled logits, kept for the backward)."""
    _need_cuda(q, k, v)
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    q, ldq = _rows16(q, L)
    k, ldk = _rows16(k, S)
    v, ldv = _rows16(v, S)
    out = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    lse = torch.empty(N, nhead, L, dtype=torch.float32, device=q.device)
    check(_lib.lib().gf_full_attention_train_forward(_p(q), _p(k), _p(v), _dt(q), N, L, S, nhead, D, ldq, ldk, ldv, 1.0 / D ** 0.5, _p(out), C,
                                                     _p(lse), _stream()), 'gf_full_attention_train_forward')
    return out, lse


def full_attention_backward(q, k, v, out, dout, lse, nhead):
    """(dq [N,L,256], dk, dv [N,S,256]) of full_attention_train_forward given its out / lse and dout [N,L,256]."""
    _need_cuda(q, k, v, out, dout, lse)
    N, L, C = q.shape
    S = k.shape[1]
    D = C // nhead
    q, ldq = _rows16(q, L)
    k, ldk = _rows16(k, S)
    v, ldv = _rows16(v, S)
    out, ldo = _rows16(out, L)
    dout, lddo = _rows16(dout, L)
    dq = torch.empty(N, L, C, dtype=q.dtype, device=q.device)
    dkv = torch.empty(2, N, S, C, dtype=q.dtype, device=q.device)
    L_ = _lib.lib()
    ws = _ws.get('k4bwd', L_.gf_full_attention_backward_workspace_bytes(N, L, nhead), q.device)
    check(L_.gf_full_attention_backward(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(_contig(lse)), _dt(q), N, L, S, nhead, D, ldq, ldk, ldv, ldo, lddo,
                                        1.0 / D ** 0.5, _p(dq), _p(dkv[0]), _p(dkv[1]), _p(ws), ws.numel(), _stream()),
          'gf_full_attention_backward')
    return dq, dkv[0], dkv[1]


def window_linear_attention_backward(q, k, v, dout, eps=1e-6):
    """(dq, dk, dv) [Nw, Lw, 128] of the fine level's window linear attention (8 heads of 16, Lw <= 32, no masks) given dout."""
    _need_cuda(q, k, v, dout)
    q, k, v, dout = _contig(q), _contig(k), _contig(v), _contig(dout)
    Nw, Lw, C = q.shape
    if C != 128 or Lw > 32 or k.shape != q.shape or v.shape != q.shape or dout.shape != q.shape:
        raise ValueError('window_linear_attention_backward: [Nw, Lw <= 32, 128] tensors of one shape')
    d = torch.empty(3, Nw, Lw, C, dtype=q.dtype, device=q.device)
    if Nw:
        check(_lib.lib().gf_window_linear_attention_backward(_p(q), _p(k), _p(v), _p(dout), _dt(q), Nw, Lw, float(eps), _p(d[0]), _p(d[1]),
                                                             _p(d[2]), _stream()), 'gf_window_linear_attention_backward')
    return d[0], d[1], d[2]


def fine_match_backward(f0, f1, temperature, dconf):
    """(df0, df1) [M,25,C] of K8's fine_matrix given dconf fp32 [M,25,25]; f0, f1 [M,25,C] (M > 0) of one dtype."""
    _need_cuda(f0, f1, dconf)
    f0, f1, dconf = _contig(f0), _contig(f1), _contig(dconf.float())
    M, WW, C = f0.shape
    df = torch.empty(2, M, WW, C, dtype=f0.dtype, device=f0.device)
    check(_lib.lib().gf_fine_match_backward(_p(f0), _p(f1), _dt(f0), M, WW, C, float(temperature), _p(dconf), _p(df[0]), _p(df[1]), _stream()),
          'gf_fine_match_backward')
    return df[0], df[1]
