"""Thin torch-tensor front end of the C ABI: pointer plumbing, workspace caching, stream passing.

Every function enqueues on torch's current stream and returns device tensors; data-dependent
counts stay on the device (`counts` tensors) until a caller needs their value.
"""
import ctypes

import torch

from . import _lib
from ._lib import GF_F16, GF_F32, check

_DTYPES = {torch.float32: GF_F32, torch.float16: GF_F16}


def _dt(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f'geoformer_amd kernels take float32 or float16 tensors, got {t.dtype}') from None


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GeoFormerHipError('geoformer_amd ops need CUDA(HIP) tensors; there is no CPU path')


class _Workspaces:
    """One growing byte buffer per (device, tag, stream)."""

    def __init__(self):
        self.bufs = {}

    def get(self, tag, nbytes, device):
        key = (device, tag, torch.cuda.current_stream(device).cuda_stream)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self.bufs[key] = b
        return b


_ws = _Workspaces()


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


def dual_softmax_match(f0, f1, temperature, thr, hw0_c, hw1_c, scale, mask0=None, mask1=None, scale0=None,
                       scale1=None, force_one=False):
    """K1.  f0 [N,L,C], f1 [N,S,C] -> dict(conf [N,L,S] fp32, b_ids/i_ids/j_ids int64 [cap], mconf [cap],
    mkpts0_c/mkpts1_c [cap,2], counts int32 [1+N]) - all on the device, match arrays at capacity."""
    _need_cuda(f0, f1)
    f0, f1 = _contig(f0), _contig(f1)
    N, L, C = f0.shape
    S = f1.shape[1]
    dev = f0.device
    cap = N * min(L, S) + (N if force_one else 0)
    conf = torch.empty(N, L, S, dtype=torch.float32, device=dev)
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
