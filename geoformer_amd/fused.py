"""Host side of the fused encoder-layer kernels (csrc/k9_encoder_fused.hip): weight pre-packing and the two ops.

The kernels consume weights as a linear stream of 1-KiB MFMA A fragments (64 lanes x 8 sixteen-bit elements) in
exactly the order the kernel multiplies them (K9: 16-KiB blocks of 16 fragments, 8 per wave half; K11: 32-KiB blocks).  Two fragment
orders exist:

  standard   element j of lane l = W[32 nb + (l & 31)][16 ks + 8 (l >> 5) + j]
             (the other operand comes from the token tile in LDS, natural k order)
  permuted   element j of lane l = W[32 nb + (l & 31)][32 t + 16 s + 8 (j >> 2) + 4 (l >> 5) + (j & 3)],  ks = 2 t + s
             (the other operand is a packed accumulator tile: its k order is the accumulator's row order)

Packing happens once per layer and dtype (cached by the modules), with plain torch gathers.
"""
import ctypes

import torch

from . import _lib
from ._lib import check
from .ops import _DTYPES, _contig, _dt, _need_cuda, _p, _rows2d, _stream, _ws

_IDX = {}


def _frag_index(order, device):
    """[ks, lane, j] -> k offset inside the 16-deep k-step ... returned as the absolute k = f(ks, lane, j) table builder."""
    key = (order, str(device))
    if key not in _IDX:
        lane = torch.arange(64, device=device)[:, None]
        j = torch.arange(8, device=device)[None, :]
        h = lane >> 5
        if order == 'std':
            koff = 8 * h + j                                        # within a 16-deep step
        else:
            koff = 8 * (j >> 2) + 4 * h + (j & 3)
        _IDX[key] = (koff, (lane & 31).expand(64, 8))
    return _IDX[key]


def fragments(w, order):
    """w [N, K] -> [N/32, K/16, 64, 8]: fragment (nb, ks) of the chosen order."""
    N, K = w.shape
    koff, row = _frag_index(order, w.device)
    nb = torch.arange(N // 32, device=w.device)[:, None, None, None]
    ks = torch.arange(K // 16, device=w.device)[None, :, None, None]
    return w[(32 * nb + row[None, None]), (16 * ks + koff[None, None])]


def _steps(f, tiles, ksteps):
    """f [NB, KS, 64, 8] -> stream piece: for each k-step in `ksteps` (a list of lists: the k-steps of one step) the fragments of the
    tiles `tiles`, k-step-major inside the step (the fine-level layer's stream, pack_fine_layer_stream)."""
    return torch.cat([f[tiles, ks].reshape(-1) for group in ksteps for ks in group])


def _pair_block(f, tiles_w0, tiles_w1, ksteps):
    """One 16-fragment block of the wave-pair kernels (csrc/k9_encoder_fused.hip): fragments 0..7 for wave half 0, 8..15 for half 1;
    each half = its tiles x the block's k-steps, k-step-major."""
    return torch.cat([f[torch.as_tensor(tiles, device=f.device), ks].reshape(-1) for tiles in (tiles_w0, tiles_w1) for ks in ksteps])


def pack_layer_stream(wq, wm, w1, w2):
    """Stream of gf_encoder_layer: 16-KiB blocks of 16 fragments, 8 per wave half (half w owns the output channels 128 w .. 128 w + 127 of
    every product = channel tiles 4 w .. 4 w + 3; of a 128-wide hidden slice the two tiles 2 w, 2 w + 1):
       [W_q: 8 blocks, head-major (block b: k-steps 8 (b & 1) .. + 7 of the half's head b >> 1)] + W_m: 8 blocks (same, permuted order) + the MLP's blocks, per 128-wide
       hidden slice sl: W_1x(sl) = W_1[:, :256]: 4 blocks (k-steps 4 b .. 4 b + 3 x the half's 2 hidden tiles), W_1m(sl) = W_1[:, 256:]:
       4 blocks (same, permuted), W_2(sl) = W_2[:, slice]: 4 blocks (k-steps 2 b, 2 b + 1 x the half's 4 output tiles, permuted), in the
       kernel's software-pipelined order (below).   wq may be None (attention computed elsewhere).  1 MiB with W_q (64 blocks), 896 KiB
       without (56)."""
    c = wm.shape[0]
    lo, hi = [0, 1, 2, 3], [4, 5, 6, 7]
    parts = []
    if wq is not None:
        # HEAD-MAJOR: block b = k-steps 8 (b & 1) .. + 7 of the half's head b >> 1 (one accumulator tile per head: its attention then rides in
        # the next head's blocks)
        fq = fragments(wq, 'std')
        parts += [torch.cat([fq[4 * wh + (b >> 1), 8 * (b & 1):8 * (b & 1) + 8].reshape(-1) for wh in range(2)]) for b in range(8)]
    fm = fragments(wm, 'perm')
    parts += [_pair_block(fm, lo, hi, [2 * b, 2 * b + 1]) for b in range(8)]
    f1x, f1m = fragments(w1[:, :c], 'std'), fragments(w1[:, c:], 'perm')   # [16, 16, 64, 8]

    def w1_blocks(f, sl):
        return [_pair_block(f, [4 * sl, 4 * sl + 1], [4 * sl + 2, 4 * sl + 3], range(4 * b, 4 * b + 4)) for b in range(4)]

    def w2_blocks(sl):
        f2 = fragments(w2[:, 128 * sl:128 * sl + 128], 'perm')                # [8, 8, 64, 8]
        return [_pair_block(f2, lo, hi, [2 * b, 2 * b + 1]) for b in range(4)]
    # the MLP is software-pipelined (the exchange of slice s - 1's hidden operands hides behind W_1x(s)):
    #   W_1x(0) W_1m(0) | W_1x(s) W_2(s - 1) W_1m(s), s = 1..3 | W_2(3)
    parts += w1_blocks(f1x, 0) + w1_blocks(f1m, 0)
    for sl in range(1, 4):
        parts += w1_blocks(f1x, sl) + w2_blocks(sl - 1) + w1_blocks(f1m, sl)
    parts += w2_blocks(3)
    return torch.cat(parts).contiguous()


def pack_kv_stream(wk, wv):
    """Stream of gf_encoder_kv_state (and of the layer's state tail): W_k in 8 blocks (k-steps 2 b, 2 b + 1 x the half's 4 tiles), then W_v."""
    lo, hi = [0, 1, 2, 3], [4, 5, 6, 7]
    return torch.cat([_pair_block(fragments(w, 'std'), lo, hi, [2 * b, 2 * b + 1]) for w in (wk, wv) for b in range(8)]).contiguous()


def _state_out(out, n, C, device):
    if out is None:
        return torch.empty(n, C * 32 + C, dtype=torch.float32, device=device)
    if out.shape != (n, C * 32 + C) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError('the state output must be a contiguous fp32 [N, 256*32 + 256] tensor (a row range of a larger one is fine)')
    return out


def encoder_kv_state(src, wstream_kv, kv_mask=None, out=None):
    """src [N, S, 256] (16-bit) -> kv_state fp32 [N, 256*32 + 256]."""
    _need_cuda(src, wstream_kv)
    N, S, C = src.shape
    if C != 256:
        raise ValueError('the fused encoder kernels are built for d_model = 256')
    src, ld = _rows2d(src)
    L_ = _lib.lib()
    ws = _ws.get('k6', L_.gf_encoder_kv_workspace_bytes(N, S), src.device)
    out = _state_out(out, N, C, src.device)
    km = None if kv_mask is None else _contig(kv_mask.reshape(N, S).to(torch.uint8))
    check(L_.gf_encoder_kv_state(_p(src), ld, _dt(src), N, S, _p(km), _p(wstream_kv), _p(out), _p(ws), ws.numel(), _stream()),
          'gf_encoder_kv_state')
    return out


def encoder_layer(x, wstream, ln_params, eps1, eps2, activation, msg=None, kv_state=None, source_len=0, q_mask=None,
                  attn_eps=1e-6, row_flag=None, flag_rows=0, out=None, tail_stream=None, tail_first=0, tail_out=None):
    """x [N, L, 256] -> out [N, L, 256]; give either `msg` (attention output) or `kv_state` (+ source_len).
    tail_stream (a pack_kv_stream, linear-attention form only): returns (out, kv_state_out) with kv_state_out [N - tail_first,
    256*32 + 256] = encoder_kv_state(out[tail_first:], tail_stream, q_mask[tail_first:]) computed inside the same launch."""
    _need_cuda(x, wstream, ln_params)
    N, L, C = x.shape
    x, ldx = _rows2d(x)
    ldm = 0
    if msg is not None:
        msg, ldm = _rows2d(msg)
    if out is None:
        out = torch.empty(N, L, C, dtype=x.dtype, device=x.device)
    elif out.shape != (N, L, C) or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError('out must be a contiguous tensor of the result shape and dtype')
    qm = None if q_mask is None else _contig(q_mask.reshape(N, L).to(torch.uint8))
    if tail_stream is not None:
        if kv_state is None or row_flag is not None:
            raise ValueError('the state tail belongs to the linear-attention form (kv_state given, no row_flag)')
        L_ = _lib.lib()
        nt = N - int(tail_first)
        ws = _ws.get('k6', L_.gf_encoder_kv_workspace_bytes(nt, L), x.device)
        st_out = _state_out(tail_out, nt, C, x.device)
        check(L_.gf_encoder_layer_kv(_p(x), ldx, _p(kv_state), int(source_len), _p(qm), float(attn_eps), _p(wstream), _p(ln_params),
                                     float(eps1), float(eps2), int(activation), _p(out), C, _dt(x), N, L, _p(tail_stream), int(tail_first),
                                     _p(st_out), _p(ws), ws.numel(), _stream()), 'gf_encoder_layer_kv')
        return out, st_out
    check(_lib.lib().gf_encoder_layer(_p(x), ldx, _p(msg), ldm, _p(kv_state), int(source_len), _p(qm), float(attn_eps), _p(wstream),
                                      _p(ln_params), float(eps1), float(eps2), int(activation), _p(row_flag), int(flag_rows), _p(out), C,
                                      _dt(x), N, L, _stream()), 'gf_encoder_layer')
    return out


# ---------------------------------------------------------------------------------------------------------------
# K11: fused fine-level encoder layer (csrc/k11_fine_layer.hip)
# ---------------------------------------------------------------------------------------------------------------
def pack_fine_layer_stream(wq, wk, wv, wm, w1, w2):
    """Stream of gf_fine_layer (d_model 128): 80 STEPS of 4 fragments (10 blocks of 32), tile-major:
       per channel tile nb { W_k[nb]: k-steps 0-3, 4-7 ; W_v[nb]: 0-3, 4-7 }  (standard order: the other operand is the window),
       per nb { W_q[nb]: 0-3, 4-7 } (standard), per nb { W_m[nb]: 0-3, 4-7 } (permuted: the other operand is a packed accumulator),
       per 32-wide hidden tile hb { W_1[hb, :128]: 0-3, 4-7 (standard) ; W_1[hb, 128:]: 0-3, 4-7 (permuted) ;
                                    W_2[tiles 0-3, k-step 2 hb] ; W_2[tiles 0-3, k-step 2 hb + 1] (permuted) }."""
    c = wm.shape[0]
    if c != 128 or w1.shape != (256, 256) or w2.shape != (128, 256):
        raise ValueError('the fused fine-level layer is built for d_model = 128')
    dev = wm.device
    halves = [[0, 1, 2, 3], [4, 5, 6, 7]]
    one = lambda nb: torch.tensor([nb], device=dev)                                   # noqa: E731
    all4 = torch.arange(4, device=dev)
    fk, fv, fq, fm = fragments(wk, 'std'), fragments(wv, 'std'), fragments(wq, 'std'), fragments(wm, 'perm')
    parts = []
    for nb in range(4):
        for f in (fk, fv):
            parts += [_steps(f, one(nb), [h]) for h in halves]
    for f in (fq, fm):
        for nb in range(4):
            parts += [_steps(f, one(nb), [h]) for h in halves]
    f1x, f1m, f2 = fragments(w1[:, :c], 'std'), fragments(w1[:, c:], 'perm'), fragments(w2, 'perm')   # [8,8] [8,8] [4,16]
    for hb in range(8):
        parts += [_steps(f1x, one(hb), [h]) for h in halves]
        parts += [_steps(f1m, one(hb), [h]) for h in halves]
        parts += [_steps(f2, all4, [[2 * hb + s]]) for s in range(2)]
    out = torch.cat(parts).contiguous()
    assert out.numel() == 10 * 32 * 64 * 8                      # 10 blocks of 32 fragments of 64 lanes x 8 elements
    return out


def fine_layer(x, src, wstream, ln_params, eps1, eps2, attn_eps=1e-6, out=None):
    """x, src [Nw, Lw <= 32, 128] (16-bit, contiguous; src may be x) -> out [Nw, Lw, 128]: one fine-level encoder layer."""
    _need_cuda(x, src, wstream, ln_params)
    Nw, Lw, C = x.shape
    if C != 128 or src.shape != x.shape or x.dtype != src.dtype:
        raise ValueError('gf_fine_layer: x and src must be [Nw, Lw, 128] tensors of one 16-bit dtype')
    x = _contig(x)
    src = x if src is x else _contig(src)
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError('out must be a contiguous tensor of the result shape and dtype')
    if Nw == 0:
        return out
    check(_lib.lib().gf_fine_layer(_p(x), _p(src), _p(out), _dt(x), Nw, Lw, _p(wstream), _p(ln_params), float(eps1), float(eps2),
                                   float(attn_eps), _stream()), 'gf_fine_layer')
    return out


# ---------------------------------------------------------------------------------------------------------------
# K10: 3x3 convolution with fused epilogue (csrc/k10_conv3x3.hip)
# ---------------------------------------------------------------------------------------------------------------
_ZEROS = {}


def pack_conv3x3_stream(w, rem8=False, s2=False):
    """w [Cout, Cin, 3, 3] (BatchNorm already folded, 16-bit) -> the fragment stream of gf_conv3x3_nhwc: for each 32-channel
    input chunk c, tap t = 3 ky + kx and half hf of the output channels, Cout/32 fragments of the 16x16x32 MFMA's A operand:
    fragment tt = accumulator tile ct = hf Cout/32 + tt, lane l -> row l % 16, input channels 32 c + 8 (l / 16) .. + 7; row r of
    tile ct is output channel 32 (ct / 2) + 8 (r / 4) + 4 (ct % 2) + r % 4 - the four accumulator rows a lane owns in the tiles
    2 j and 2 j + 1 are eight consecutive channels, i.e. 16 bytes of the NHWC output, stored from registers
    (18 sub-steps per chunk; the kernel's weight blocks are 6 consecutive sub-steps).
    rem8 (Cin = 224 whose channels 200.. carry zero weights - the 196-channel pyramid level): six full chunks (channels 0 .. 191)
    and a REMAINDER of channels 192 .. 199 whose 9 taps x 8 channels = 72 contraction elements fill three 32-deep MFMA k-steps
    (lane k group kg of k-step s = tap 4 s + kg, taps 9 .. 11 zero) instead of the nine k-steps of a seventh chunk: 6 sub-steps
    (k-step s, half hf) of Cout/32 fragments behind the full chunks' (GF_CONV_REM8).
    s2 (GF_CONV_S2, stride 2): the taps of a chunk are listed parity plane by parity plane - (ky, kx) = (0,0) (0,2) (2,0) (2,2) |
    (0,1) (2,1) | (1,0) (1,2) | (1,1) - the order in which the kernel walks (and refills) the four planes of its halo patch."""
    cout, cin = w.shape[:2]
    nt = cout // 32
    dev = w.device
    lane = torch.arange(64, device=dev)
    row, kg = lane % 16, lane // 16
    # out[c, t, hf, tt, lane, j] = wt[channel(hf nt + tt, row), c, 8 kg + j, t]
    ct = torch.arange(2, device=dev)[:, None, None] * nt + torch.arange(nt, device=dev)[None, :, None]                              # [2, nt, 1]
    co = 32 * (ct // 2) + 4 * (ct % 2) + (8 * (row // 4) + row % 4)[None, None, :]                                                 # [2, nt, 64]
    kk = 8 * kg[:, None] + torch.arange(8, device=dev)[None, :]                      # [64, 8]
    full = 192 if rem8 else cin
    if rem8 and (cin != 224 or bool(w[:, 200:].any())):
        raise ValueError('rem8 packs a 224-channel convolution whose input channels 200.. have zero weights')
    wt = w[:, :full].reshape(cout, full // 32, 32, 9)                                # [cout, chunk, k, tap]
    if s2:
        if rem8:
            raise ValueError('no remainder form at stride 2')
        wt = wt[..., torch.tensor([0, 2, 6, 8, 1, 7, 3, 5, 4], device=dev)]
    g = wt[co[:, :, :, None], :, kk[None, None, :, :], :]                            # [2, nt, 64, 8, chunk, tap]
    out = g.permute(4, 5, 0, 1, 2, 3).contiguous().reshape(-1)
    if not rem8:
        return out
    wr = torch.zeros(cout, 8, 12, dtype=w.dtype, device=dev)                         # [cout, channel 192 + j, tap (9 real + 3 zero)]
    wr[:, :, :9] = w[:, 192:200].reshape(cout, 8, 9)
    j = torch.arange(8, device=dev)
    tap = 4 * torch.arange(3, device=dev)[:, None] + kg[None, :]                      # [3 k-steps, 64 lanes]
    # rem[s, hf, tt, lane, j] = wr[channel(hf nt + tt, row), j, 4 s + kg]
    r = wr[co[None, :, :, :, None], j[None, None, None, None, :], tap[:, None, None, :, None]]        # [3, 2, nt, 64, 8]
    return torch.cat([out, r.contiguous().reshape(-1)])


def pack_lateral_frags(w):
    """w [Cout, Cin(,1,1)] (16-bit) -> the A fragments of gf_lateral_upsample_add_nhwc (K12): [Cin/32][Cout/16][64 lanes][8]: fragment
    (kc, ct), lane l = row l % 16 of accumulator tile ct - output channel 32 (ct / 2) + 8 (row / 4) + 4 (ct % 2) + row % 4, the deal of
    pack_conv3x3_stream - and input channels 32 kc + 8 (l / 16) .. + 7."""
    w = w.reshape(w.shape[0], w.shape[1])
    cout, cin = w.shape
    dev = w.device
    lane = torch.arange(64, device=dev)
    row, kg = lane % 16, lane // 16
    ct = torch.arange(cout // 16, device=dev)[:, None]
    co = 32 * (ct // 2) + 4 * (ct % 2) + (8 * (row // 4) + row % 4)[None, :]                       # [nct, 64]
    kk = 8 * kg[:, None] + torch.arange(8, device=dev)[None, :]                                      # [64, 8]
    kc = torch.arange(cin // 32, device=dev)
    return w[co[None, :, :, None], (32 * kc[:, None, None, None] + kk[None, None, :, :])].contiguous().reshape(-1)


def lateral_supported(cin, cout):
    return bool(_lib.lib().gf_lateral_supported(int(cin), int(cout)))


def lateral_upsample_add(x, wfrag, cout, lo):
    """K12: 1x1 convolution of channels_last x [N, Cin, H, W] (weights as pack_lateral_frags) + bilinear (align_corners=True) upsampling of
    channels_last lo [N, cout, h, w] -> channels_last [N, cout, H, W]; W even."""
    _need_cuda(x, wfrag, lo)
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or not lo.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('lateral_upsample_add expects channels_last [N, C, H, W] tensors')
    N, cin, H, W = x.shape
    if lo.shape[:2] != (N, cout) or lo.dtype != x.dtype or wfrag.dtype != x.dtype:
        raise ValueError('lo must be [N, cout, h, w] of the dtype of x and of the fragments')
    out = torch.empty(N, cout, H, W, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    check(_lib.lib().gf_lateral_upsample_add_nhwc(_p(x), _p(wfrag), _p(lo), _p(out), N, lo.shape[2], lo.shape[3], H, W, cin, cout,
                                                  _dt(x), _stream()), 'gf_lateral_upsample_add_nhwc')
    return out


def conv3x3_supported(cin, cout):
    return bool(_lib.lib().gf_conv3x3_supported(int(cin), int(cout)))


CONV_PAD16 = 0x100          # GF_CONV_PAD16 (include/geoformer_hip.h)
CONV_REM8 = 0x200           # GF_CONV_REM8
CONV_S2 = 0x400             # GF_CONV_S2


def conv3x3s2_supported(cin, cout):
    return bool(_lib.lib().gf_conv3x3s2_supported(int(cin), int(cout)))


def conv3x3(x, wstream, cout, shift=None, residual=None, act=0, slope=0.01, pad16=False, rem8=False, stride=1):
    """x channels_last [N, Cin, H, W] (16-bit) -> channels_last [N, cout, H, W] = act(conv3x3(x) + shift + residual).
    pad16 (cout = 224): the output channels 196 .. 223 are zero padding (zero weights): the all-padding accumulator tile is skipped.
    rem8: wstream = pack_conv3x3_stream(w, rem8=True) (Cin = 224, input channels 200.. have zero weights: not multiplied).
    stride 2: wstream = pack_conv3x3_stream(w, s2=True); the result is [N, cout, (H-1)//2+1, (W-1)//2+1]."""
    _need_cuda(x, wstream)
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise ValueError('conv3x3 expects a channels_last [N, C, H, W] tensor')
    N, cin, H, W = x.shape
    if stride not in (1, 2):
        raise ValueError('stride 1 or 2')
    Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride == 2 else (H, W)
    out = torch.empty(N, cout, Ho, Wo, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (residual.shape != out.shape or residual.dtype != x.dtype or
                                 not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('residual must be a channels_last tensor of the output shape and dtype')
    z = _ZEROS.get(x.device)
    if z is None:
        z = _ZEROS[x.device] = torch.zeros(256, dtype=torch.uint8, device=x.device)
    check(_lib.lib().gf_conv3x3_nhwc(_p(x), _p(wstream), _p(shift), _p(residual), _p(out), _p(z), N, H, W, cin, cout,
                                     int(act) | (CONV_PAD16 if pad16 else 0) | (CONV_REM8 if rem8 else 0) | (CONV_S2 if stride == 2 else 0),
                                     float(slope), _dt(x), _stream()), 'gf_conv3x3_nhwc')
    return out


def conv3x3_wgrad(x, dy, cin=None, cout=None):
    """K10 (training): dW fp32 [cout, cin, 3, 3] of y = conv3x3(x, W, stride 1, pad 1) from channels_last 16-bit x [N, cx, H, W] and
    dy [N, cy, H, W]; cin / cout = the real widths when the maps carry padding channels (default: the stored ones)."""
    _need_cuda(x, dy)
    for t in (x, dy):
        if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
            raise ValueError('conv3x3_wgrad expects channels_last [N, C, H, W] tensors')
    N, cx, H, W = x.shape
    cy = dy.shape[1]
    if dy.shape[0] != N or tuple(dy.shape[2:]) != (H, W) or dy.dtype != x.dtype:
        raise ValueError('x and dy must agree in batch, size and dtype')
    cin, cout = int(cin or cx), int(cout or cy)
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=x.device)
    L_ = _lib.lib()
    ws = _ws.get('k10wgrad', L_.gf_conv3x3_wgrad_workspace_bytes(N, H, W, cin, cout), x.device)
    check(L_.gf_conv3x3_wgrad_nhwc(_p(x), _p(dy), _dt(x), N, H, W, cx, cy, cin, cout, _p(dw), _p(ws), ws.numel(), _stream()), 'gf_conv3x3_wgrad_nhwc')
    return dw
