"""Differentiable (autograd) form of the GeoFormer forward for the training step.

The inference path of this package is forward-only HIP.  Training needs gradients through every block, and
the backward kernels are not written yet, so the training step runs the SAME arithmetic as
`model/full_model.py:39-123` in plain torch ops on the model's own parameters (names = state-dict keys) and
lets autograd differentiate it; the data-dependent, gradient-free pieces keep their device implementations
(RANSAC homography on the GPU: `ops.ransac_homography`).  Each function cites the reference lines it follows.
Nothing here is used by `GeoFormer.forward` (inference).
"""
import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- a1 position encoding
def position_encoding_table(d_model, h, w, temp_bug_fix, device):
    """position_encoding.py:22-35 (incl. the `//2` precedence quirk of the non-fixed variant)."""
    ypos = torch.arange(1, h + 1, dtype=torch.float32, device=device).view(1, h, 1).expand(1, h, w)
    xpos = torch.arange(1, w + 1, dtype=torch.float32, device=device).view(1, 1, w).expand(1, h, w)
    k2 = torch.arange(0, d_model // 2, 2, device=device).float()
    if temp_bug_fix:
        freq = torch.exp(k2 * (-math.log(10000.0) / (d_model // 2)))
    else:
        freq = torch.exp(k2 * (-math.log(10000.0) / d_model // 2))
    freq = freq.view(-1, 1, 1)
    pe = torch.zeros(d_model, h, w, device=device)
    pe[0::4] = torch.sin(xpos * freq)
    pe[1::4] = torch.cos(xpos * freq)
    pe[2::4] = torch.sin(ypos * freq)
    pe[3::4] = torch.cos(ypos * freq)
    return pe


def add_pe(x, temp_bug_fix=False):
    _, c, h, w = x.shape
    return x + position_encoding_table(c, h, w, temp_bug_fix, x.device)[None]


# ----------------------------------------------------------------------------- attention + encoder layers
def linear_attention(q, k, v, q_mask=None, kv_mask=None, eps=1e-6):
    """linear_attention.py:21-51.  Under the mixed-bf16 step the feature maps phi and the normaliser are evaluated in fp32
    (the reference's own comment: the /S ... *S scaling exists "to prevent fp16 overflow"); the two contractions take
    whatever the autocast context gives them (bf16 operands, fp32 accumulation)."""
    if q.dtype in (torch.float16, torch.bfloat16):
        out_dtype = q.dtype
        Q, K = F.elu(q.float()) + 1, F.elu(k.float()) + 1
        if q_mask is not None:
            Q = Q * q_mask[:, :, None, None]
        if kv_mask is not None:
            K = K * kv_mask[:, :, None, None]
            v = v * kv_mask[:, :, None, None]
        s_len = v.size(1)
        KV = torch.einsum('nshd,nshv->nhdv', K.to(out_dtype), v / s_len).float()
        with torch.autocast(device_type=q.device.type, enabled=False):
            Z = 1 / (torch.einsum('nlhd,nhd->nlh', Q, K.sum(dim=1)) + eps)
            out = torch.einsum('nlhd,nhdv->nlhv', Q, KV) * (Z * s_len)[..., None]
        return out.to(out_dtype)
    Q, K = F.elu(q) + 1, F.elu(k) + 1
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    s_len = v.size(1)
    v = v / s_len
    KV = torch.einsum('nshd,nshv->nhdv', K, v)
    Z = 1 / (torch.einsum('nlhd,nhd->nlh', Q, K.sum(dim=1)) + eps)
    return torch.einsum('nlhd,nhdv,nlh->nlhv', Q, KV, Z) * s_len


def full_attention(q, k, v, kv_mask=None):
    """geo_attention.py:53-101: masked logits -1e8 before the 1/sqrt(D) scale, all-masked rows zeroed.
    The windowed cross layers call it with ONE query per 'sample' (q [L, 1, H, D] against 25 gathered keys, L = 6400 'samples'):
    as einsum that is a batch of 25,600 products of 1 x 64 by 64 x 25 - the library bmm spent 33 ms of GPU time and 90 ms of host
    time per training step on them - so that case is written as broadcast multiply + reduce (same sums, other order)."""
    temp = 1.0 / q.size(3) ** .5
    if q.size(1) == 1:
        qk = (q * k).sum(-1, dtype=torch.float32)                          # [n, S, H]
        if kv_mask is not None:
            qk = qk.masked_fill(~kv_mask[:, :, None], -1e8)
        a = torch.softmax(qk * temp, dim=1)
        out = (a.unsqueeze(-1) * v).sum(1, keepdim=True).to(v.dtype)       # [n, 1, H, D]
    elif kv_mask is None and q.is_cuda and q.dtype in (torch.float16, torch.bfloat16):
        # GeoTransformer's 'self' layers in the mixed-16-bit step (all L queries against the gathered inlier keys, no mask): the fused
        # attention of the library (flash form, forward AND backward) instead of materialised [L, K, H] logits + softmax under autograd -
        # at batch 8 / 640 x 640 with thousands of inlier cells per image those tensors were 300 of the step's 440 ms of GPU time
        # (aten::copy_ 121, bmm + its backward 165, softmax + its backward 75; tools/train_profile.py --mega).  Same arithmetic up to
        # the 16-bit rounding of P (fp32 softmax statistics, scale 1 / sqrt(D)); the fp32 step keeps the explicit form below.
        return F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    else:
        qk = torch.einsum('nlhd,nshd->nlsh', q, k)
        if kv_mask is not None:
            qk = qk.masked_fill(~kv_mask[:, None, :, None], -1e8)
        a = torch.softmax(qk * temp, dim=2)
        out = torch.einsum('nlsh,nshd->nlhd', a, v)
    if kv_mask is not None:
        out = out * (kv_mask.sum(-1) != 0)[:, None, None, None].to(out.dtype)
    return out


_HIP_BACKWARD = [False]
_FP16_MATCHING = [True]           # the fused coarse stage's matching (no gradient) on fp16 copies of the features (False: fp32, for A/B)
_BATCHED_GEO = [True]             # the Geo layers' per-token operations once per layer on all images of the batch (False: image by image, for A/B)
_OWN_FULL_ATTENTION = [True]      # the Geo 'self' layers of the HIP step on csrc/k4_attention_train.hip (False: the library's fused attention, for A/B)


def set_hip_backward(on: bool):
    """Route the linears / LayerNorms / activations of the encoder layers through the HIP forward + backward kernels
    (train/hip_autograd.py) inside a 16-bit autocast region; everything else (attention cores, losses) stays on autograd."""
    _HIP_BACKWARD[0] = bool(on)


def _encoder_layer_hip(P, prefix, x, source, nhead, kind, x_mask, source_mask):
    """The same layer with the K3 chain in HIP, forward and backward.  Rounding points = those of the autocast path (16-bit
    operands, fp32 accumulation and LayerNorm statistics), plus ONE more: the LayerNorm-2 output is rounded to the 16-bit type
    before the fp32 residual add (autocast's layer_norm returns fp32)."""
    from . import hip_autograd as HA
    dt = torch.get_autocast_dtype('cuda')
    n, _, c = x.shape
    d = c // nhead
    x16 = x.to(dt)
    s16 = x16 if source is x else source.to(dt)
    q = HA.linear(x16, P[prefix + 'q_proj.weight'])
    k = HA.linear(s16, P[prefix + 'k_proj.weight'])
    v = HA.linear(s16, P[prefix + 'v_proj.weight'])
    if kind == 'loftr' and d == 32:                      # the coarse level: K2 forward and backward in HIP
        msg = HA.linear_attention(q, k, v, nhead, x_mask, source_mask)
    elif (kind == 'loftr' and d == 16 and c == 128 and x.shape[1] <= 32 and source.shape[1] == x.shape[1] and x_mask is None
          and source_mask is None):                      # the fine level's 25-token windows: K2's window form and its backward
        msg = HA.window_linear_attention(q, k, v, nhead)
    elif kind == 'geo' and d == 64 and c == 256 and source_mask is None and _OWN_FULL_ATTENTION[0]:
        msg = HA.full_attention(q, k, v, nhead)          # GeoTransformer's 'self' layers: K4's training kernels, forward and backward
    else:
        q, k, v = (t.view(n, -1, nhead, d) for t in (q, k, v))
        msg = linear_attention(q, k, v, x_mask, source_mask) if kind == 'loftr' else full_attention(q, k, v, source_mask)
    return _finish_layer_hip(P, prefix, x, x16, msg.reshape(n, -1, c).to(dt), kind)


def _finish_layer_hip(P, prefix, x, x16, msg, kind):
    """merge -> LayerNorm -> MLP on [x | message] -> LayerNorm -> + x on the HIP Functions (the part of an encoder layer behind its
    attention; transformer.py:53-60 / geo_transformer/transformer.py:59-66)."""
    from . import hip_autograd as HA
    msg = HA.linear(msg, P[prefix + 'merge.weight'])
    msg = HA.layer_norm(msg, P[prefix + 'norm1.weight'], P[prefix + 'norm1.bias'])
    hid = HA.linear(x16, P[prefix + 'mlp.0.weight'], msg, 'relu' if kind == 'loftr' else 'tanh')
    msg = HA.linear(hid, P[prefix + 'mlp.2.weight'])
    msg = HA.layer_norm(msg, P[prefix + 'norm2.weight'], P[prefix + 'norm2.bias'])
    return x + msg


def _geo_cross_layer_hip(P, prefix, x, kmap, vmap, win, nhead):
    """One side of GeoTransformer's 'cross' layer (geo_transformer/transformer.py:125-139) on the HIP Functions: x [L, C] (or a batch
    [A, L, C]) attends to the 25 window positions `win` [A, L, 25] (cells of the other image, -1 = masked) of the other image's PROJECTED
    maps kmap / vmap [A, S, C] (project-then-gather: k_proj / v_proj have no bias, so gathering their outputs equals projecting the
    gathered rows)."""
    from . import hip_autograd as HA
    dt = torch.get_autocast_dtype('cuda')
    single = x.dim() == 2
    xb = x[None] if single else x
    x16 = xb.to(dt)
    q = HA.linear(x16, P[prefix + 'q_proj.weight'])
    msg = HA.window_cross_attention(q, kmap, vmap, win, nhead)
    out = _finish_layer_hip(P, prefix, xb, x16, msg, 'geo')
    return out[0] if single else out


def _geo_self_layers_batched(P, prefix, feats, idxs, nhead):
    """GeoTransformer's 'self' layer (transformer.py:111-124) for ALL images of the batch in one chain of HIP Functions: the per-token
    operations (q / k / v projections, merge, LayerNorms, MLP) run once on the concatenated rows of every image that has inlier cells -
    per image they were 6400-row GEMMs at a tenth of the batched rate, 16 launches each where one does - and only the attention core runs
    per image (its key sets differ in length).  Same arithmetic per token as _encoder_layer_hip; images without inliers pass through."""
    from . import hip_autograd as HA
    dt = torch.get_autocast_dtype('cuda')
    act = [i for i, ix in enumerate(idxs) if ix.numel()]
    if not act:
        return feats
    lens = [feats[i].shape[0] for i in act]
    X = torch.cat([feats[i] for i in act], 0)[None]                     # [1, T, C] fp32 residual stream
    X16 = X.to(dt)
    rows, o = [], 0
    for i, l in zip(act, lens):
        rows.append(X16[0, o:o + l].index_select(0, idxs[i]))
        o += l
    src = torch.cat(rows, 0)[None]
    q = HA.linear(X16, P[prefix + 'q_proj.weight'])
    k = HA.linear(src, P[prefix + 'k_proj.weight'])
    v = HA.linear(src, P[prefix + 'v_proj.weight'])
    msgs, o, so = [], 0, 0
    for i, l in zip(act, lens):
        S = idxs[i].numel()
        msgs.append(HA.full_attention(q[:, o:o + l], k[:, so:so + S], v[:, so:so + S], nhead))
        o, so = o + l, so + S
    out = _finish_layer_hip(P, prefix, X, X16, torch.cat(msgs, 1), 'geo')[0].split(lens, 0)
    feats = list(feats)
    for j, i in enumerate(act):
        feats[i] = out[j]
    return feats


def encoder_layer(P, prefix, x, source, nhead, kind, x_mask=None, source_mask=None):
    """loftr_module/transformer.py:37-60 (ReLU, linear attention) / geo_transformer/transformer.py:39-66 (Tanh, full)."""
    n, _, c = x.shape
    if _HIP_BACKWARD[0] and x.is_cuda and torch.is_autocast_enabled() and c % 128 == 0 and x.shape[1] > 0 and source.shape[1] > 0:
        return _encoder_layer_hip(P, prefix, x, source, nhead, kind, x_mask, source_mask)
    d = c // nhead
    q = F.linear(x, P[prefix + 'q_proj.weight']).view(n, -1, nhead, d)
    k = F.linear(source, P[prefix + 'k_proj.weight']).view(n, -1, nhead, d)
    v = F.linear(source, P[prefix + 'v_proj.weight']).view(n, -1, nhead, d)
    msg = linear_attention(q, k, v, x_mask, source_mask) if kind == 'loftr' else full_attention(q, k, v, source_mask)
    msg = F.linear(msg.reshape(n, -1, c), P[prefix + 'merge.weight'])
    msg = F.layer_norm(msg, (c,), P[prefix + 'norm1.weight'], P[prefix + 'norm1.bias'])
    hid = F.linear(torch.cat([x, msg], dim=2), P[prefix + 'mlp.0.weight'])
    hid = torch.relu(hid) if kind == 'loftr' else torch.tanh(hid)
    msg = F.linear(hid, P[prefix + 'mlp.2.weight'])
    msg = F.layer_norm(msg, (c,), P[prefix + 'norm2.weight'], P[prefix + 'norm2.bias'])
    return x + msg


def local_feature_transformer(P, prefix, layer_names, nhead, f0, f1, m0=None, m1=None):
    """loftr_module/transformer.py:82-104 ('cross': f1 attends to the UPDATED f0)."""
    for idx, name in enumerate(layer_names):
        lp = f'{prefix}layers.{idx}.'
        if name == 'self':
            f0 = encoder_layer(P, lp, f0, f0, nhead, 'loftr', m0, m0)
            f1 = encoder_layer(P, lp, f1, f1, nhead, 'loftr', m1, m1)
        elif name == 'cross':
            f0 = encoder_layer(P, lp, f0, f1, nhead, 'loftr', m0, m1)
            f1 = encoder_layer(P, lp, f1, f0, nhead, 'loftr', m1, m0)
        else:
            raise KeyError(name)
    return f0, f1


# ----------------------------------------------------------------------------- coarse matching
def dual_softmax(f0, f1, temperature, m0=None, m1=None):
    """coarse_matching.py:113-125.  Always in fp32, also inside the mixed-bf16 step: a bf16 similarity (8 significant bits
    at magnitudes of 10-30, i.e. steps of 0.1-0.25 in the exponent) would move every probability by tens of per cent."""
    c = f0.shape[-1]
    with torch.autocast(device_type=f0.device.type, enabled=False):
        if f0.dtype in (torch.float16, torch.bfloat16):
            f0, f1 = f0.float(), f1.float()
        sim = torch.einsum('nlc,nsc->nls', f0 / c ** .5, f1 / c ** .5) / temperature
        if m0 is not None:
            sim = sim.masked_fill(~(m0[..., None] * m1[:, None]).bool(), -1e9)
        return F.softmax(sim, 1) * F.softmax(sim, 2)


@torch.no_grad()
def coarse_match(conf, data, thr):
    """coarse_matching.py:132-212 (border_rm forced to 0; 'dataset_name' -> forced (0,0) match for empty samples)."""
    mask = conf > thr
    mask = mask & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
    if 'dataset_name' in data:
        empty = mask.flatten(1).sum(-1) == 0
        mask[empty, 0, 0] = True
    mask_v, all_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    scale = data['hw0_i'][0] / data['hw0_c'][0]
    scale0 = scale * data['scale0'][b_ids] if 'scale0' in data else scale
    scale1 = scale * data['scale1'][b_ids] if 'scale1' in data else scale
    w0c, w1c = int(data['hw0_c'][1]), int(data['hw1_c'][1])
    mk0 = torch.stack([i_ids % w0c, i_ids // w0c], dim=1) * scale0
    mk1 = torch.stack([j_ids % w1c, j_ids // w1c], dim=1) * scale1
    return {'b_ids': b_ids, 'i_ids': i_ids, 'j_ids': j_ids, 'm_bids': b_ids, 'mkpts0_c': mk0, 'mkpts1_c': mk1}


# ----------------------------------------------------------------------------- GeoModule
def _map_keypoints(h, w, scale, device):
    ys, xs = torch.meshgrid(torch.arange(h // scale, device=device), torch.arange(w // scale, device=device), indexing='ij')
    return torch.stack([xs.reshape(-1), ys.reshape(-1)], -1) * scale


def _warp_points(points, hmat):
    """utils/homography.py:86-105 for one sample."""
    homog = torch.cat([points, torch.ones(points.shape[0], 1, device=points.device)], dim=-1)
    out = torch.bmm(hmat[None].to(homog.dtype), homog[None].permute(0, 2, 1)).permute(0, 2, 1)[0]
    w = out[:, 2:].clone()
    w[w == 0] = 1e-6
    return out[:, :2] / w


def _make_windows(kps, img_hw, window_size, scale):
    """utils/common_utils.py:65-91."""
    h, w = img_hw
    r = torch.arange(window_size, device=kps.device) - window_size // 2
    dy, dx = torch.meshgrid(r, r, indexing='ij')
    off = torch.stack([dx, dy], -1).float().view(1, window_size * window_size, 2) * scale
    p = kps[:, None, :] + off
    oob = (p[..., 0] < 0) | (p[..., 1] < 0) | (p[..., 0] >= w) | (p[..., 1] >= h)
    return p.masked_fill(oob[..., None], 0).long(), ~oob


def _sample_windows(kps, tokens, w, s):
    """utils/common_utils.py:166-181: kps [L,ww,2] pixel coords, tokens [H*W, C] (the map, token-major) -> [L,ww,C].
    index_select instead of fmap[:, y, x]: same values, but its backward is an atomic index_add instead of the sort-based
    index_put (1.9 ms per call at 80x80x25 windows: 10 % of the training step)."""
    cell = (kps.float() // s).long()
    lin = cell[..., 1] * w + cell[..., 0]
    return tokens.index_select(0, lin.reshape(-1)).view(lin.shape[0], lin.shape[1], tokens.shape[1])


def geo_module(P, cnn0, cnn1, data, geo_cfg, homography_fn: Callable):
    """model/geo_module.py:23-116 + geo_transformer/transformer.py:89-146.  The geometry (homography, inlier maps,
    window tables) carries no gradient; the attention layers do.  homography_fn(b, kp0, kp1) -> (M float64 [3,3]
    tensor | None, inlier mask bool [n])."""
    n, c, hh0, ww0 = cnn0.shape
    _, _, hh1, ww1 = cnn1.shape
    dev = cnn0.device
    f0 = add_pe(cnn0).flatten(2).transpose(1, 2)
    f1 = add_pe(cnn1).flatten(2).transpose(1, 2)
    H0, W0 = data['image0'].shape[2:]
    H1, W1 = data['image1'].shape[2:]
    scale = int(data['hw0_i'][0] // data['hw0_c'][0])
    wsz = geo_cfg['window_size']
    per_sample_scale = 'scale0' in data
    win0, win1, msk0, msk1, map0, map1 = [], [], [], [], [], []
    with torch.no_grad():
        for b in range(n):
            sel = data['m_bids'] == b
            kp0, kp1 = data['mkpts0_c'][sel].long(), data['mkpts1_c'][sel].long()
            if per_sample_scale:
                kp0 = (kp0 / (scale * data['scale0'][b]) * scale).long()
                kp1 = (kp1 / (scale * data['scale1'][b]) * scale).long()
            M = None
            if len(kp0) > 8:
                M, keep = homography_fn(b, kp0, kp1)
            if M is not None:
                kp0, kp1 = kp0[keep], kp1[keep]
                Md = M.to(device=dev, dtype=torch.float64)
                s0 = scale * data['scale0'][b] if per_sample_scale else scale
                s1 = scale * data['scale1'][b] if per_sample_scale else scale
                p1 = _warp_points(_map_keypoints(H0, W0, scale, dev), Md.to(f0.dtype))
                k1w, m1w = _make_windows(p1, (H1, W1), wsz, s1)
                p0 = _warp_points(_map_keypoints(H1, W1, scale, dev), torch.inverse(Md[None])[0].to(f0.dtype))
                k0w, m0w = _make_windows(p0, (H0, W0), wsz, s0)
                win0.append(k0w); win1.append(k1w); msk0.append(m0w); msk1.append(m1w)
            else:
                win0.append(None); win1.append(None); msk0.append(None); msk1.append(None)
            m0 = torch.zeros(hh0 * ww0, dtype=torch.bool, device=dev)
            m1 = torch.zeros(hh1 * ww1, dtype=torch.bool, device=dev)
            m0[(kp0[:, 1] // scale) * ww0 + kp0[:, 0] // scale] = True
            m1[(kp1[:, 1] // scale) * ww1 + kp1[:, 0] // scale] = True
            map0.append(m0); map1.append(m1)

    nhead = geo_cfg['nhead']
    idx0 = [m.nonzero().flatten() for m in map0]          # feat[mask] (transformer.py:118,121) as index_select: ascending order
    idx1 = [m.nonzero().flatten() for m in map1]
    f0 = [f0[b] for b in range(n)]          # per-sample lists: no in-place writes into autograd inputs
    f1 = [f1[b] for b in range(n)]
    cells = [None] * n                      # HIP path: (win1 cells, win0 cells) int32 [1, L, 25] per sample, -1 = masked
    cat_cells = None
    for idx, name in enumerate(geo_cfg['layer_names']):
        lp = f'geo_module.des_transformer.layers.{idx}.'
        hip = _HIP_BACKWARD[0] and cnn0.is_cuda and torch.is_autocast_enabled() and c == 256 and nhead == 4
        if name == 'self' and hip and _OWN_FULL_ATTENTION[0] and _BATCHED_GEO[0]:
            both = _geo_self_layers_batched(P, lp, f0 + f1, idx0 + idx1, nhead)
            f0, f1 = both[:n], both[n:]
        elif name == 'cross' and hip and wsz == 5 and _BATCHED_GEO[0]:
            # K5 forward and backward in HIP on the projected maps (both images' keys / values from the PRE-update features, :126-129), all
            # samples that have a homography in one batch per side
            from . import hip_autograd as HA
            dt = torch.get_autocast_dtype('cuda')
            act = [b for b in range(n) if win1[b] is not None]
            if act:
                for b in act:
                    if cells[b] is None:
                        with torch.no_grad():
                            cells[b] = tuple(torch.where(m, (k[..., 1] // scale) * wk_ + k[..., 0] // scale, -1).to(torch.int32)[None].contiguous()
                                             for k, m, wk_ in ((win1[b], msk1[b], ww1), (win0[b], msk0[b], ww0)))
                X0, X1 = torch.stack([f0[b] for b in act]), torch.stack([f1[b] for b in act])
                s0, s1 = X0.to(dt), X1.to(dt)
                k0, v0 = HA.linear(s0, P[lp + 'k_proj.weight']), HA.linear(s0, P[lp + 'v_proj.weight'])
                k1, v1 = HA.linear(s1, P[lp + 'k_proj.weight']), HA.linear(s1, P[lp + 'v_proj.weight'])
                if cat_cells is None:                    # one table per side for all 'cross' layers of the step (its inverse index is kept on it)
                    cat_cells = (torch.cat([cells[b][0] for b in act], 0), torch.cat([cells[b][1] for b in act], 0))
                o0 = _geo_cross_layer_hip(P, lp, X0, k1, v1, cat_cells[0], nhead)
                o1 = _geo_cross_layer_hip(P, lp, X1, k0, v0, cat_cells[1], nhead)
                for j, b in enumerate(act):
                    f0[b], f1[b] = o0[j], o1[j]
        elif name == 'self':
            for b in range(n):
                if idx0[b].numel():
                    f0[b] = encoder_layer(P, lp, f0[b][None], f0[b].index_select(0, idx0[b])[None], nhead, 'geo')[0]
                if idx1[b].numel():
                    f1[b] = encoder_layer(P, lp, f1[b][None], f1[b].index_select(0, idx1[b])[None], nhead, 'geo')[0]
        elif name == 'cross' and _HIP_BACKWARD[0] and cnn0.is_cuda and torch.is_autocast_enabled() and c == 256 and nhead == 4 and wsz == 5:
            # K5 forward and backward in HIP on the projected maps (both images' keys / values from the PRE-update features, :126-129)
            from . import hip_autograd as HA
            dt = torch.get_autocast_dtype('cuda')
            for b in range(n):
                if win1[b] is None:
                    continue
                if cells[b] is None:
                    with torch.no_grad():
                        cells[b] = tuple(torch.where(m, (k[..., 1] // scale) * wk_ + k[..., 0] // scale, -1).to(torch.int32)[None].contiguous()
                                         for k, m, wk_ in ((win1[b], msk1[b], ww1), (win0[b], msk0[b], ww0)))
                s0, s1 = f0[b].to(dt)[None], f1[b].to(dt)[None]
                k0, v0 = HA.linear(s0, P[lp + 'k_proj.weight']), HA.linear(s0, P[lp + 'v_proj.weight'])
                k1, v1 = HA.linear(s1, P[lp + 'k_proj.weight']), HA.linear(s1, P[lp + 'v_proj.weight'])
                f0[b] = _geo_cross_layer_hip(P, lp, f0[b], k1, v1, cells[b][0], nhead)
                f1[b] = _geo_cross_layer_hip(P, lp, f1[b], k0, v0, cells[b][1], nhead)
        elif name == 'cross':
            g0 = [None if win0[b] is None else _sample_windows(win0[b], f0[b], ww0, scale) for b in range(n)]
            g1 = [None if win1[b] is None else _sample_windows(win1[b], f1[b], ww1, scale) for b in range(n)]
            for b in range(n):
                if g1[b] is None:
                    continue
                f0[b] = encoder_layer(P, lp, f0[b][:, None], g1[b], nhead, 'geo', None, msk1[b])[:, 0]
                f1[b] = encoder_layer(P, lp, f1[b][:, None], g0[b], nhead, 'geo', None, msk0[b])[:, 0]
        else:
            raise KeyError(name)
    return torch.stack(f0), torch.stack(f1)


# ----------------------------------------------------------------------------- fine level
def _fine_windows(feat_f, b_ids, cell_ids, w_c, stride, W):
    """fine_preprocess.py:41-56 (unfold + gather) without materialising the unfold."""
    pad = W // 2
    fp = F.pad(feat_f, (pad, pad, pad, pad)).permute(0, 2, 3, 1)           # [N, Hp, Wp, C]
    n, hp, wp, c = fp.shape
    cy, cx = (cell_ids // w_c) * stride, (cell_ids % w_c) * stride
    r = torch.arange(W, device=feat_f.device)
    yy = (cy[:, None, None] + r[None, :, None]).expand(-1, W, W)
    xx = (cx[:, None, None] + r[None, None, :]).expand(-1, W, W)
    lin = (b_ids[:, None, None] * hp + yy) * wp + xx
    return fp.reshape(n * hp * wp, c).index_select(0, lin.reshape(-1)).view(-1, W * W, c)      # (index_add backward)


def fine_preprocess(P, feat_f0, feat_f1, feat_c0, feat_c1, data, W):
    """fine_preprocess.py:30-74."""
    stride = int(data['hw0_f'][0] // data['hw0_c'][0])
    b, i, j = data['b_ids'], data['i_ids'], data['j_ids']
    cf = feat_f0.shape[1]
    if b.shape[0] == 0:
        e = torch.empty(0, W * W, cf, device=feat_f0.device)
        return e, e
    w0 = _fine_windows(feat_f0, b, i, int(data['hw0_c'][1]), stride, W)
    w1 = _fine_windows(feat_f1, b, j, int(data['hw1_c'][1]), stride, W)
    cc = torch.cat([feat_c0[b, i], feat_c1[b, j]], 0)
    hip = _HIP_BACKWARD[0] and cc.is_cuda and torch.is_autocast_enabled() and cf == 128 and cc.shape[1] == 256

    def lin(x, name):
        # the two projections of FinePreprocess (fine_preprocess.py:23-24, 66-72): on the K3 engine in the HIP step (the library's GEMM of the
        # 2 M x 25 window rows ran at 80 TFLOP/s: 1.7 ms per call), bias added in the storage type as autocast's linear does
        if hip:
            from . import hip_autograd as HA
            dt = torch.get_autocast_dtype('cuda')
            return HA.linear(x.to(dt), P[name + '.weight']) + P[name + '.bias'].to(dt)
        return F.linear(x, P[name + '.weight'], P[name + '.bias'])
    cwin = lin(cc, 'fine_preprocess.down_proj')
    both = torch.cat([torch.cat([w0, w1], 0).to(cwin.dtype), cwin[:, None].expand(-1, W * W, -1)], -1)
    both = lin(both, 'fine_preprocess.merge_feat')
    return torch.chunk(both, 2, dim=0)


def fine_match(f0, f1, data, temperature, thr):
    """fine_matching2.py:21-126; `fine_matrix` carries the gradient, the keypoints do not."""
    M, WW, C = f0.shape
    if M == 0:
        return {'fine_matrix': torch.empty(0, WW, WW, device=f0.device), 'mkpts0_f': data['mkpts0_c'],
                'mkpts1_f': data['mkpts1_c'], 'W': int(math.sqrt(WW))}
    W = int(math.sqrt(WW))
    if _HIP_BACKWARD[0] and f0.is_cuda and C <= 128 and WW == 25:
        # K8 forward (confidence, arg-max, threshold, keypoints) and its backward in HIP; the confidence stays fp32
        from . import hip_autograd as HA
        hi, hc, hf = float(data['hw0_i'][0]), float(data['hw0_c'][0]), float(data['hw0_f'][0])
        conf, fine_b, mk0, mk1, mconf = HA.HipFineMatch.apply(f0.float().contiguous(), f1.float().contiguous(), temperature, thr, data['b_ids'],
                                                               data['mkpts0_c'], data['mkpts1_c'], hi / hc, hf / hc, hi / hf,
                                                               data.get('scale0'), data.get('scale1'))
        return {'fine_matrix': conf, 'm_bids': fine_b, 'mkpts0_f': mk0, 'mkpts1_f': mk1, 'mconf': mconf, 'W': W}
    conf = dual_softmax(f0, f1, temperature)
    with torch.no_grad():
        mask = conf > thr
        mask = mask & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
        top = conf.reshape(M, -1).argmax(1)
        onehot = torch.zeros(M, WW * WW, dtype=torch.bool, device=f0.device)
        onehot[torch.arange(M, device=f0.device), top] = True
        mask = mask & onehot.view(M, WW, WW)
        fine_b = data['b_ids'][:, None, None].expand(-1, WW, WW)[mask]
        mask_v, all_j = mask.max(dim=2)
        m_ids, i_ids = torch.where(mask_v)
        j_ids = all_j[m_ids, i_ids]
        mconf = conf[m_ids, i_ids, j_ids]
        cscale = data['hw0_i'][0] / data['hw0_c'][0]
        cscale0 = cscale * data['scale0'][data['b_ids']] if 'scale0' in data else cscale
        cscale1 = cscale * data['scale1'][data['b_ids']] if 'scale1' in data else cscale
        c2f = data['hw0_f'][0] / data['hw0_c'][0]
        c0 = data['mkpts0_c'] / cscale0 * c2f
        c1 = data['mkpts1_c'] / cscale1 * c2f
        mk0 = torch.stack([i_ids % W - W // 2, i_ids // W - W // 2], dim=1) + c0[m_ids]
        mk1 = torch.stack([j_ids % W - W // 2, j_ids // W - W // 2], dim=1) + c1[m_ids]
        fscale = data['hw0_i'][0] / data['hw0_f'][0]
        fscale0 = fscale * data['scale0'][fine_b] if 'scale0' in data else fscale
        fscale1 = fscale * data['scale1'][fine_b] if 'scale1' in data else fscale
    return {'fine_matrix': conf, 'm_bids': fine_b, 'mkpts0_f': mk0 * fscale0, 'mkpts1_f': mk1 * fscale1, 'mconf': mconf,
            'W': W}


# ----------------------------------------------------------------------------- device RANSAC adaptor
def device_homography_fn(data, scale):
    """The forward's `cv2.findHomography` (geo_module.py:47-48) on the GPU: one batched `ops.ransac_homography`
    call on the first-pass matches; returns the per-sample callback `geo_module` expects."""
    from .. import ops
    n = int(data['bs'])
    counts = torch.zeros(1 + n, dtype=torch.int32, device=data['m_bids'].device)
    per = torch.bincount(data['m_bids'], minlength=n).to(torch.int32)
    counts[0] = per.sum()
    counts[1:] = per
    rs = ops.ransac_homography(data['mkpts0_c'].float().contiguous(), data['mkpts1_c'].float().contiguous(), counts, n,
                               float(scale), data.get('scale0'), data.get('scale1'))
    valid = rs['valid'].cpu()
    starts = torch.cumsum(per, 0) - per

    def fn(b, kp0, kp1):
        if int(valid[b]) == 0:
            return None, None
        s = int(starts[b])
        return rs['M'][b], rs['keep'][s:s + len(kp0)].bool()
    return fn


# ----------------------------------------------------------------------------- the forward
def _fused_coarse_stage(f0, f1, data, temp, thr, focal, m0=None, m1=None):
    """One coarse-matching stage of the training forward on the HIP path: matches from K1 (no gradient), and the
    sparse-supervision focal term + its gradient from the fused loss kernels (ops.coarse_focal_loss), so that no
    L x S tensor enters the autograd graph.  Returns (conf_matrix without grad, match dict, (loss_sum, count))."""
    from .. import ops
    scale = float(data['hw0_i'][0]) / float(data['hw0_c'][0])
    # the matches carry no gradient: K1 reads fp16 copies of the features (its 16-bit kernels: 0.5 ms per 8-pair stage where the exact-fp32
    # matrix path took 4 ms).  11 significant bits - the reference's own mixed-precision run forms this similarity under bf16 autocast (8 bits).
    h0, h1 = (t.detach().to(torch.float16) if (_FP16_MATCHING[0] and t.dtype == torch.float32) else t.detach() for t in (f0, f1))
    r = ops.dual_softmax_match(h0, h1, temp, thr, data['hw0_c'], data['hw1_c'], scale, mask0=m0, mask1=m1,
                               scale0=data.get('scale0'), scale1=data.get('scale1'), force_one='dataset_name' in data)
    m = int(r['counts'][0])
    match = {k: r[k][:m] for k in ('b_ids', 'i_ids', 'j_ids', 'mkpts0_c', 'mkpts1_c')}
    match['m_bids'] = match['b_ids']
    n_gt = data['spv_num_gt']
    if n_gt > 0:
        pb, pi, pj = data['spv_b_ids'], data['spv_i_ids'], data['spv_j_ids']
        w = None if m0 is None else (m0[pb, pi] * m1[pb, pj]).float()      # GeoLoss.compute_c_weight at the positives
        loss_sum, _ = ops.coarse_focal_loss(f0, f1, pb, pi, pj, temp, focal[0], focal[1], w, m0, m1)
    else:                                       # loftr_loss.py:220-224: a dummy positive with zero weight
        loss_sum = f0.sum() * 0.0
    return r['conf_matrix'], match, (loss_sum, max(n_gt, 1))


def fused_coarse_loss_applicable(model, data):
    """The fused HIP loss covers the training configuration of the reference: dual-softmax, sparse supervision, focal
    loss (padding masks included), coarse grids that tile (L, S multiples of 128, C = 256)."""
    h0, w0 = data['image0'].shape[2] // 8, data['image0'].shape[3] // 8
    h1, w1 = data['image1'].shape[2] // 8, data['image1'].shape[3] // 8
    return (data['image0'].is_cuda and model.config['coarse']['d_model'] == 256
            and (h0 * w0) % 128 == 0 and (h1 * w1) % 128 == 0 and 'spv_num_gt' in data)


def forward_train(model, data: Dict[str, torch.Tensor], homography_fn: Optional[Callable] = None,
                  fused_coarse_loss=None, backbone_features=None, channels_last=False):
    """`GeoFormer.forward` (model/full_model.py:39-123) under autograd.  `model` is `geoformer_amd.GeoFormer` in fp32
    precision; its backbone runs as the nn.Module (train-mode BatchNorm / SyncBatchNorm).
    fused_coarse_loss = (alpha, gamma): both coarse stages take matches from K1 and their focal term from the fused
    HIP loss (writes data['loss_d_fused'], data['loss_c_fused'] = (sum over positives, count)); None = autograd on
    the materialised confidence matrices.
    backbone_features = ((cnn0, ff0), (cnn1, ff1)): skip the backbone and run the matching path on these maps (gradient checks of the
    matching path alone, tests/test_train_gpu.py - an explicit argument: nothing in a batch dict can switch the backbone off).
    channels_last: the backbone's input in NHWC memory format (TrainStep(channels_last=True))."""
    cfg, gcfg = model.config, model.geo_cfg
    P = dict(model.named_parameters())
    img0, img1 = data['image0'], data['image1']
    n = img0.size(0)
    data.update({'bs': torch.tensor(n), 'hw0_i': torch.tensor(img0.shape[2:]), 'hw1_i': torch.tensor(img1.shape[2:])})
    if backbone_features is not None:
        # ((cnn0, ff0), (cnn1, ff1)) given as leaf tensors, e.g. planted-correspondence maps - a regime with decisive confidences that
        # random-init weights on images never reach
        (cnn0, ff0), (cnn1, ff1) = backbone_features
    elif img0.shape[2:] == img1.shape[2:]:
        both = torch.cat([img0, img1], dim=0)
        if channels_last:                                          # TrainStep(channels_last=True): NHWC convolutions (MIOpen / CK)
            both = both.contiguous(memory_format=torch.channels_last)
        feats_c, feats_f = model.backbone(both)
        (cnn0, cnn1), (ff0, ff1) = feats_c.split(n), feats_f.split(n)
    else:
        (cnn0, ff0), (cnn1, ff1) = model.backbone(img0), model.backbone(img1)
    data.update({'hw0_c': torch.tensor(cnn0.shape[2:]), 'hw1_c': torch.tensor(cnn1.shape[2:]),
                 'hw0_f': torch.tensor(ff0.shape[2:]), 'hw1_f': torch.tensor(ff1.shape[2:])})
    tbf = cfg['coarse']['temp_bug_fix']
    f0 = add_pe(cnn0, tbf).flatten(2).transpose(1, 2)
    f1 = add_pe(cnn1, tbf).flatten(2).transpose(1, 2)
    m0 = m1 = None
    if 'mask0' in data:
        m0, m1 = data['mask0'].flatten(-2), data['mask1'].flatten(-2)
    f0, f1 = local_feature_transformer(P, 'loftr_coarse.', cfg['coarse']['layer_names'], cfg['coarse']['nhead'], f0, f1, m0, m1)
    temp, thr = cfg['match_coarse']['dsmax_temperature'], cfg['match_coarse']['thr']
    if fused_coarse_loss is not None:
        conf, match, data['loss_d_fused'] = _fused_coarse_stage(f0, f1, data, temp, thr, fused_coarse_loss, m0, m1)
        data.update(conf_matrix=conf, **match)
    else:
        conf = dual_softmax(f0, f1, temp, m0, m1)
        data.update(conf_matrix=conf, **coarse_match(conf, data, thr))
    data['dect_conf_matrix'] = data['conf_matrix']
    if homography_fn is None:
        if not cnn0.is_cuda:
            raise RuntimeError('forward_train on the CPU needs a homography_fn (the device RANSAC is HIP only)')
        homography_fn = device_homography_fn(data, int(data['hw0_i'][0] // data['hw0_c'][0]))
    g0, g1 = geo_module(P, cnn0, cnn1, data, gcfg, homography_fn)
    if fused_coarse_loss is not None:
        conf, match, data['loss_c_fused'] = _fused_coarse_stage(g0, g1, data, temp, thr, fused_coarse_loss, m0, m1)
        data.update(conf_matrix=conf, **match)
    else:
        conf = dual_softmax(g0, g1, temp, m0, m1)
        data.update(conf_matrix=conf, **coarse_match(conf, data, thr))
    W = cfg['fine_window_size']
    u0, u1 = fine_preprocess(P, ff0, ff1, g0, g1, data, W)
    if u0.size(0) != 0:
        u0, u1 = local_feature_transformer(P, 'loftr_fine.', cfg['fine']['layer_names'], cfg['fine']['nhead'], u0, u1)
    data.update(fine_match(u0, u1, data, gcfg['fine_temperature'], gcfg['fine_thr']))
    return data
