"""CPU oracle for the GeoFormer coarse-to-fine matching path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this module; the product package
(geoformer_amd/) never does and fails loudly without its HIP library.

What it is: a plain-PyTorch fp32 restatement (CPU) of the reference's forward
pass (plus, further down, the same arithmetic with the product's 16-bit storage
round trips: geoformer_forward_storage), written from the behaviour of the files cited per function (paths are
relative to the reference checkout).  It is functional: weights come in as a
flat dict keyed by the reference's state-dict names.

Parity status
  * pinned:   every function below is checked in tests/test_oracle_golden.py
    against golden vectors produced by importing the reference itself in the
    build container (oracle/gen_golden.py -> tests/golden/*.npz).
  * UNPINNED: the homography RANSAC.  The reference calls OpenCV
    `cv2.findHomography(kp0, kp1, cv2.RANSAC, 8.0)` (model/geo_module.py:47-48;
    opencv_python==4.6.0.66 per requirements.txt:4), which is neither in the
    reference tree nor in this image.  Everything here takes the homography as
    an injected function `homography_fn(kp0, kp1) -> (M float64[3,3] | None,
    mask uint8[n,1])`; fixtures carry the (M, mask) they were generated with.
    oracle/ransac_oracle.c is the CPU statement of the build's own RANSAC.
"""
import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# configs (same keys/values as model/loftr_src/loftr/utils/cvpr_ds_config.py:10-48 and
# model/geo_config.py:10-17, lower-cased)
# --------------------------------------------------------------------------------------


def default_loftr_config():
    return {
        'backbone_type': 'ResNetFPN', 'resolution': (8, 2), 'fine_window_size': 5,
        'fine_concat_coarse_feat': True,
        'resnetfpn': {'initial_dim': 128, 'block_dims': [128, 196, 256]},
        'coarse': {'d_model': 256, 'd_ffn': 256, 'nhead': 8, 'layer_names': ['self', 'cross'] * 4,
                   'attention': 'linear', 'temp_bug_fix': False},
        'match_coarse': {'thr': 0.4, 'border_rm': 2, 'match_type': 'dual_softmax', 'dsmax_temperature': 0.1,
                         'skh_iters': 3, 'skh_init_bin_score': 1.0, 'skh_prefilter': True,
                         'train_coarse_percent': 0.4, 'train_pad_num_gt_min': 200},
        'fine': {'d_model': 128, 'd_ffn': 128, 'nhead': 8, 'layer_names': ['self', 'cross'], 'attention': 'linear'},
    }


def default_geo_config():
    return {'layer_names': ['self', 'cross'] * 2, 'nhead': 4, 'coarse_thr': 0.2, 'fine_temperature': 0.1,
            'fine_thr': 0.1, 'window_size': 5, 'topk': 1}


# --------------------------------------------------------------------------------------
# a1  position encoding   (model/loftr_src/loftr/utils/position_encoding.py:22-42)
# --------------------------------------------------------------------------------------


def position_encoding_table(d_model: int, h: int, w: int, temp_bug_fix: bool = False) -> torch.Tensor:
    """[d_model, h, w] table.  Positions start at 1 (cumsum of ones, :23-24).  With
    temp_bug_fix=False the exponent scale is `(-ln(1e4) / d_model) // 2` == -1.0 (:28), so the
    frequencies are exp(-2k); the fixed variant uses -ln(1e4)/(d_model//2) (:26)."""
    ypos = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(1, h, w)
    xpos = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(1, h, w)
    k2 = torch.arange(0, d_model // 2, 2).float()
    if temp_bug_fix:
        freq = torch.exp(k2 * (-math.log(10000.0) / (d_model // 2)))
    else:
        freq = torch.exp(k2 * (-math.log(10000.0) / d_model // 2))
    freq = freq.view(-1, 1, 1)
    pe = torch.zeros(d_model, h, w)
    pe[0::4] = torch.sin(xpos * freq)
    pe[1::4] = torch.cos(xpos * freq)
    pe[2::4] = torch.sin(ypos * freq)
    pe[3::4] = torch.cos(ypos * freq)
    return pe


def add_position_encoding(x: torch.Tensor, temp_bug_fix: bool = False) -> torch.Tensor:
    """x [N,C,H,W] -> x + pe[:, :H, :W]   (position_encoding.py:37-42)."""
    _, c, h, w = x.shape
    return x + position_encoding_table(c, h, w, temp_bug_fix)[None]


# --------------------------------------------------------------------------------------
# a2  linear attention   (model/loftr_src/loftr/loftr_module/linear_attention.py:21-51)
# --------------------------------------------------------------------------------------


def linear_attention(q, k, v, q_mask=None, kv_mask=None, eps: float = 1e-6):
    """q [N,L,H,D], k,v [N,S,H,D] -> [N,L,H,D].  V is divided by S and the result multiplied
    back (:45-49); eps joins the denominator before the reciprocal (:48)."""
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
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


# --------------------------------------------------------------------------------------
# a11 full attention (Geo flavour)   (model/geo_transformer/geo_attention.py:53-101)
# --------------------------------------------------------------------------------------


def full_attention(q, k, v, q_mask=None, kv_mask=None):
    """Masked logits are filled with -1e8 BEFORE the 1/sqrt(D) scale (:83,92); query rows whose
    kv_mask is all False are zeroed afterwards (:98-100)."""
    qk = torch.einsum('nlhd,nshd->nlsh', q, k)
    mask = None
    if q_mask is not None:
        mask = q_mask[:, :, None, None]
    if kv_mask is not None:
        mask = kv_mask[:, None, :, None] if mask is None else mask * kv_mask[:, None, :, None]
    if mask is not None:
        qk = qk.masked_fill(~mask, -1e8)
    a = torch.softmax(qk * (1.0 / q.size(3) ** .5), dim=2)
    out = torch.einsum('nlsh,nshd->nlhd', a, v)
    if kv_mask is not None:
        out = out * (kv_mask.sum(-1) != 0)[:, None, None, None].to(out.dtype)
    return out


# --------------------------------------------------------------------------------------
# a3 / a10  encoder layer
#   LoFTR: model/loftr_src/loftr/loftr_module/transformer.py:37-60  (ReLU MLP, linear attention)
#   Geo:   model/geo_transformer/transformer.py:39-66               (Tanh MLP, full attention)
# All projections and MLP linears are bias-free; the two LayerNorms are affine.
# --------------------------------------------------------------------------------------


def encoder_layer(P: Dict[str, torch.Tensor], prefix: str, x, source, nhead: int, kind: str,
                  x_mask=None, source_mask=None):
    n, _, c = x.shape
    d = c // nhead
    q = F.linear(x, P[prefix + 'q_proj.weight']).view(n, -1, nhead, d)
    k = F.linear(source, P[prefix + 'k_proj.weight']).view(n, -1, nhead, d)
    v = F.linear(source, P[prefix + 'v_proj.weight']).view(n, -1, nhead, d)
    if kind == 'loftr':
        msg = linear_attention(q, k, v, x_mask, source_mask)
    elif kind == 'geo':
        msg = full_attention(q, k, v, x_mask, source_mask)
    else:
        raise KeyError(kind)
    msg = F.linear(msg.reshape(n, -1, c), P[prefix + 'merge.weight'])
    msg = F.layer_norm(msg, (c,), P[prefix + 'norm1.weight'], P[prefix + 'norm1.bias'])
    hid = F.linear(torch.cat([x, msg], dim=2), P[prefix + 'mlp.0.weight'])
    hid = torch.relu(hid) if kind == 'loftr' else torch.tanh(hid)
    msg = F.linear(hid, P[prefix + 'mlp.2.weight'])
    msg = F.layer_norm(msg, (c,), P[prefix + 'norm2.weight'], P[prefix + 'norm2.bias'])
    return x + msg


# --------------------------------------------------------------------------------------
# 16-bit STORAGE MODE of the encoder layers.  The product's fast modes keep tensors in fp16 (or bf16) and
# accumulate in fp32; the functions below restate the same reference arithmetic with a round trip through
# the storage type at exactly the points where the HIP kernels round (include/geoformer_hip.h, K9 / K3 / K2):
#   'fused'   csrc/k9_encoder_fused.hip (coarse LoFTR layers; the part of a Geo layer after its attention):
#             every MFMA operand is rounded (phi(q), phi(k), v, KV/S, Ksum/S, msg, LN1 output, hidden
#             activations), everything else is fp32, the output x + LN2(.) is rounded once;
#   'chain'   the K3 + K2 kernel chain (fine level): every tensor a kernel writes is rounded (q, k, v, message,
#             LN1 output, hidden activations, LN2 output, and then the residual sum once more).
# `st` is the storage dtype.  Not a new algorithm: with st = torch.float32 both reduce to encoder_layer above.
# --------------------------------------------------------------------------------------


def rt(x: torch.Tensor, st) -> torch.Tensor:
    """Round trip through the storage type."""
    return x if st is None or st == torch.float32 else x.to(st).float()


def _phi(x):
    """elu(x) + 1 as the kernels evaluate it: x + 1 for x > 0, exp(x) otherwise."""
    return torch.where(x > 0, x + 1, torch.exp(torch.clamp(x, max=0)))


def linear_attention_fused(q, k, v, st, q_mask=None, kv_mask=None, eps: float = 1e-6):
    """q [N,L,H,D], k, v [N,S,H,D] un-rounded fp32 projections -> message [N,L,H,D] (rounded)."""
    s_len = v.size(1)
    inv_s = torch.tensor(1.0, dtype=torch.float32) / float(s_len)
    Kf = _phi(k)                                     # fp32: Ksum is not an MFMA operand, it adds the unrounded values
    if kv_mask is not None:
        Kf = Kf * kv_mask[:, :, None, None]
    Q, K, V = rt(_phi(q), st), rt(Kf, st), rt(v, st)
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    KV = rt(torch.einsum('nshd,nshv->nhdv', K, V) * inv_s, st)
    Ks = rt(Kf.sum(dim=1) * inv_s, st)
    num = torch.einsum('nlhd,nhdv->nlhv', Q, KV)
    den = torch.einsum('nlhd,nhd->nlh', Q, Ks) + torch.tensor(eps, dtype=torch.float32) / float(s_len)
    return rt(num * (1.0 / den)[..., None], st)


def _finish_fused(P, prefix, x, msg, kind, st, round_out=True):
    """x, msg [N,L,C] (rounded) -> x + LN2(W_2 act(W_1 [x | LN1(W_m msg)])).  round_out=False keeps the fp32 sum (the rounding
    ablation of tools/rounding_ablation.py: what the features cost the dual-softmax by being STORED in 16 bits)."""
    c = x.shape[-1]
    W = lambda name: rt(P[prefix + name], st)                                    # noqa: E731
    m = F.linear(msg, W('merge.weight'))
    m = rt(F.layer_norm(m, (c,), P[prefix + 'norm1.weight'], P[prefix + 'norm1.bias']), st)
    hid = F.linear(torch.cat([x, m], dim=2), W('mlp.0.weight'))
    hid = rt(torch.relu(hid) if kind == 'loftr' else torch.tanh(hid), st)
    o = F.layer_norm(F.linear(hid, W('mlp.2.weight')), (c,), P[prefix + 'norm2.weight'], P[prefix + 'norm2.bias'])
    return rt(x + o, st) if round_out else x + o


def encoder_layer_fused(P, prefix, x, source, nhead, st, x_mask=None, source_mask=None, round_out=True):
    """LoFTR (linear attention, ReLU) layer in the 'fused' storage mode; x, source already rounded."""
    n, _, c = x.shape
    d = c // nhead
    W = lambda name: rt(P[prefix + name], st)                                    # noqa: E731
    q = F.linear(x, W('q_proj.weight')).view(n, -1, nhead, d)
    k = F.linear(source, W('k_proj.weight')).view(n, -1, nhead, d)
    v = F.linear(source, W('v_proj.weight')).view(n, -1, nhead, d)
    msg = linear_attention_fused(q, k, v, st, x_mask, source_mask).reshape(n, -1, c)
    return _finish_fused(P, prefix, x, msg, 'loftr', st, round_out)


def linear_attention_window(q, k, v, st, q_mask=None, kv_mask=None, eps: float = 1e-6):
    """q [N,L,H,D], k, v [N,S,H,D] as STORED (already rounded) -> message [N,L,H,D] (rounded): the rounding points of
    la_window_mfma (csrc/k2_linear_attention.hip, the fine level's windows in the 16-bit modes): phi(q), phi(k) are MFMA
    operands (rounded), KV / S and Ksum / S are rounded (Ksum adds the ROUNDED phi(k)), accumulation and the division
    are fp32."""
    s_len = v.size(1)
    inv_s = torch.tensor(1.0, dtype=torch.float32) / float(s_len)
    Q, K = rt(_phi(q), st), rt(_phi(k), st)
    if q_mask is not None:
        Q = Q * q_mask[:, :, None, None]
    if kv_mask is not None:
        K = K * kv_mask[:, :, None, None]
        v = v * kv_mask[:, :, None, None]
    KV = rt(torch.einsum('nshd,nshv->nhdv', K, v) * inv_s, st)
    Ks = rt(K.sum(dim=1) * inv_s, st)
    num = torch.einsum('nlhd,nhdv->nlhv', Q, KV)
    den = torch.einsum('nlhd,nhd->nlh', Q, Ks) + torch.tensor(eps, dtype=torch.float32) * inv_s
    return rt(num / den[..., None], st)


def encoder_layer_chain(P, prefix, x, source, nhead, st, x_mask=None, source_mask=None):
    """LoFTR layer in the 'chain' storage mode (K3 linears + K2 attention of short sequences, the fine level).  The
    attention: 8 heads of 16 over <= 32 tokens in a 16-bit mode runs la_window_mfma (linear_attention_window above); any
    other shape runs la_small, which evaluates phi, the state and the normaliser in fp32 from the stored q, k, v."""
    n, _, c = x.shape
    d = c // nhead
    W = lambda name: rt(P[prefix + name], st)                                    # noqa: E731
    q = rt(F.linear(x, W('q_proj.weight')), st).view(n, -1, nhead, d)
    k = rt(F.linear(source, W('k_proj.weight')), st).view(n, -1, nhead, d)
    v = rt(F.linear(source, W('v_proj.weight')), st).view(n, -1, nhead, d)
    if st is not None and st != torch.float32 and c == 128 and d == 16 and q.shape[1] <= 32 and k.shape[1] <= 32:
        msg = linear_attention_window(q, k, v, st, x_mask, source_mask).reshape(n, -1, c)
    else:
        Q, K = _phi(q), _phi(k)
        if x_mask is not None:
            Q = Q * x_mask[:, :, None, None]
        if source_mask is not None:
            K = K * source_mask[:, :, None, None]
            v = v * source_mask[:, :, None, None]
        s_len = v.size(1)
        KV = torch.einsum('nshd,nshv->nhdv', K, v / s_len)
        Z = 1 / (torch.einsum('nlhd,nhd->nlh', Q, K.sum(dim=1)) + 1e-6)
        msg = rt(torch.einsum('nlhd,nhdv,nlh->nlhv', Q, KV, Z) * s_len, st).reshape(n, -1, c)
    m = rt(F.layer_norm(F.linear(msg, W('merge.weight')), (c,), P[prefix + 'norm1.weight'], P[prefix + 'norm1.bias']), st)
    hid = rt(torch.relu(F.linear(torch.cat([x, m], dim=2), W('mlp.0.weight'))), st)
    o = rt(F.layer_norm(F.linear(hid, W('mlp.2.weight')), (c,), P[prefix + 'norm2.weight'], P[prefix + 'norm2.bias']), st)
    return rt(x + o, st)


# --------------------------------------------------------------------------------------
# a4  LoFTR layer schedule   (loftr_module/transformer.py:82-104)
# --------------------------------------------------------------------------------------


def local_feature_transformer(P, prefix, layer_names, nhead, f0, f1, m0=None, m1=None):
    """'cross': f0 is updated first and f1 then attends to the UPDATED f0 (:99-100)."""
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


# --------------------------------------------------------------------------------------
# a5  dual-softmax confidence   (model/loftr_src/loftr/utils/coarse_matching.py:90-130)
# --------------------------------------------------------------------------------------


def dual_softmax(f0, f1, temperature: float, m0=None, m1=None):
    """Both feature sets are divided by sqrt(C) (:113), sim = f0.f1^T / temperature (:118-119),
    masked pairs filled with -1e9 (:123-124), conf = softmax(dim=1) * softmax(dim=2) (:125)."""
    c = f0.shape[-1]
    f0, f1 = f0 / c ** .5, f1 / c ** .5
    sim = torch.einsum('nlc,nsc->nls', f0, f1) / temperature
    if m0 is not None:
        sim = sim.masked_fill(~(m0[..., None] * m1[:, None]).bool(), -1e9)
    return F.softmax(sim, 1) * F.softmax(sim, 2)


# --------------------------------------------------------------------------------------
# a6  coarse match extraction   (coarse_matching.py:132-212)
# --------------------------------------------------------------------------------------


def coarse_match(conf, data, thr: float):
    """mask = conf>thr & row-max & col-max (:161,176-178; the border mask is a no-op because
    border_rm is forced to 0, :32,78-79); with 'dataset_name' present an empty sample gets
    mask[b,0,0]=True (:182-184); first True column per row (:185); torch.where row-major order
    (:186); keypoints (x,y) = (id % w, id // w) * scale (:193-201)."""
    mask = conf > thr
    mask = mask & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
    if 'dataset_name' in data:
        empty = mask.flatten(1).sum(-1) == 0
        mask[empty, 0, 0] = True
    mask_v, all_j = mask.max(dim=2)
    b_ids, i_ids = torch.where(mask_v)
    j_ids = all_j[b_ids, i_ids]
    mconf = conf[b_ids, i_ids, j_ids]
    scale = data['hw0_i'][0] / data['hw0_c'][0]
    scale0 = scale * data['scale0'][b_ids] if 'scale0' in data else scale
    scale1 = scale * data['scale1'][b_ids] if 'scale1' in data else scale
    w0c, w1c = data['hw0_c'][1], data['hw1_c'][1]
    mk0 = torch.stack([i_ids % w0c, i_ids // w0c], dim=1) * scale0
    mk1 = torch.stack([j_ids % w1c, j_ids // w1c], dim=1) * scale1
    return {'b_ids': b_ids, 'i_ids': i_ids, 'j_ids': j_ids, 'm_bids': b_ids,
            'mkpts0_c': mk0, 'mkpts1_c': mk1, 'mconf': mconf}


# --------------------------------------------------------------------------------------
# a8  window geometry
#   get_map_keypoints  utils/common_utils.py:137-144
#   warp_points_batch  utils/homography.py:86-105
#   generate_window    utils/common_utils.py:65-91
# --------------------------------------------------------------------------------------


def map_keypoints(h: int, w: int, scale: int = 8) -> torch.Tensor:
    """[(h/scale)*(w/scale), 2] int64 (x, y) pixel coordinates of the coarse cells, row-major."""
    ys, xs = torch.meshgrid(torch.arange(h // scale), torch.arange(w // scale), indexing='ij')
    return torch.stack([xs.reshape(-1), ys.reshape(-1)], -1) * scale


def warp_points(points: torch.Tensor, hmat: torch.Tensor) -> torch.Tensor:
    """points [L,2] (any dtype) warped by hmat [3,3]; an exactly-zero w becomes 1e-6 (:101-103)."""
    ones = torch.ones(points.shape[0], 1)
    homog = torch.cat([points, ones], dim=-1)                      # type promotion as torch.cat does
    out = torch.bmm(hmat[None].to(homog.dtype) if hmat.dtype != homog.dtype else hmat[None],
                    homog[None].permute(0, 2, 1)).permute(0, 2, 1)[0]
    w = out[:, 2:].clone()
    w[w == 0] = 1e-6
    return out[:, :2] / w


def make_windows(kps: torch.Tensor, img_hw, window_size: int, scale):
    """kps [L,2] float -> (kps [L,ww,2] int64, mask [L,ww] bool).  Offsets (c-2, r-2)*scale with
    x fastest (:71-78); OOB test on the float coordinates (:84); OOB entries zeroed then .long()
    (:88-89)."""
    h, w = img_hw
    r = torch.arange(window_size) - window_size // 2
    dy, dx = torch.meshgrid(r, r, indexing='ij')
    off = torch.stack([dx, dy], -1).float().view(1, window_size * window_size, 2) * scale
    p = kps[:, None, :] + off
    oob = (p[..., 0] < 0) | (p[..., 1] < 0) | (p[..., 0] >= w) | (p[..., 1] >= h)
    p = p.masked_fill(oob[..., None], 0)
    return p.long(), ~oob


def sample_windows(kps: torch.Tensor, fmap: torch.Tensor, s: int = 8) -> torch.Tensor:
    """a12: kps [L,ww,2] int64 pixel coords, fmap [C,H,W] -> [L,ww,C]; cell = float(kps)//s
    (utils/common_utils.py:166-181; no 1/sqrt(C) in this branch)."""
    cell = (kps.float() // s).long()
    return fmap[:, cell[..., 1], cell[..., 0]].permute(1, 2, 0)


# --------------------------------------------------------------------------------------
# a7 + a9  GeoModule / GeoTransformer
#   model/geo_module.py:23-116, model/geo_transformer/transformer.py:89-146
# --------------------------------------------------------------------------------------


def geo_module(P, cnn0, cnn1, data, geo_cfg, homography_fn: Callable, record: Optional[dict] = None):
    """cnn0/cnn1 are the RAW backbone coarse maps [N,C,h,w] (full_model.py:60-61,89); position
    encoding is re-added here with the buggy variant (geo_module.py:19,28-29)."""
    n, c, hh0, ww0 = cnn0.shape
    _, _, hh1, ww1 = cnn1.shape
    f0 = add_position_encoding(cnn0).flatten(2).transpose(1, 2).contiguous()
    f1 = add_position_encoding(cnn1).flatten(2).transpose(1, 2).contiguous()
    H0, W0 = data['image0'].shape[2:]
    H1, W1 = data['image1'].shape[2:]
    scale = int(data['hw0_i'][0] // data['hw0_c'][0])
    wsz = geo_cfg['window_size']
    per_sample_scale = 'scale0' in data
    win0, win1, msk0, msk1 = [], [], [], []
    map0 = torch.zeros(n, hh0 * ww0, dtype=torch.bool)
    map1 = torch.zeros(n, hh1 * ww1, dtype=torch.bool)
    for b in range(n):
        sel = data['m_bids'] == b
        kp0, kp1 = data['mkpts0_c'][sel].long(), data['mkpts1_c'][sel].long()   # geo_module.py:110-111
        if per_sample_scale:  # geo_module.py:38-43 (intended per-sample scale)
            kp0 = (kp0 / (scale * data['scale0'][b]) * scale).long()
            kp1 = (kp1 / (scale * data['scale1'][b]) * scale).long()
        M = None
        if len(kp0) > 8:
            M, inl = homography_fn(kp0.numpy(), kp1.numpy())
        if record is not None:
            record.setdefault('homographies', []).append(None if M is None else (M.copy(), inl.copy()))
        if M is not None:
            keep = torch.from_numpy(inl[:, 0] == 1)
            kp0, kp1 = kp0[keep], kp1[keep]
            Md = torch.from_numpy(M)
            s0 = scale * data['scale0'][b] if per_sample_scale else scale
            s1 = scale * data['scale1'][b] if per_sample_scale else scale
            # forward warp uses M cast to the feature dtype (:58); the inverse is taken in
            # float64 and cast afterwards (:67)
            p1 = warp_points(map_keypoints(H0, W0, scale), Md.to(f0.dtype))
            k1w, m1w = make_windows(p1, (H1, W1), wsz, s1)
            p0 = warp_points(map_keypoints(H1, W1, scale), torch.inverse(Md[None])[0].to(f0.dtype))
            k0w, m0w = make_windows(p0, (H0, W0), wsz, s0)
            win0.append(k0w); win1.append(k1w); msk0.append(m0w); msk1.append(m1w)
        else:
            win0.append(None); win1.append(None); msk0.append(None); msk1.append(None)
        map0[b, (kp0[:, 1] // scale) * ww0 + kp0[:, 0] // scale] = True
        map1[b, (kp1[:, 1] // scale) * ww1 + kp1[:, 0] // scale] = True
    if record is not None:
        record.update(map0=map0, map1=map1, win0=win0, win1=win1, msk0=msk0, msk1=msk1)

    nhead = geo_cfg['nhead']
    f0, f1 = f0.clone(), f1.clone()
    for idx, name in enumerate(geo_cfg['layer_names']):
        lp = f'geo_module.des_transformer.layers.{idx}.'
        if name == 'self':   # transformer.py:111-124: keys/values = tokens at inlier cells
            for b in range(n):
                if map0[b].any():
                    f0[b] = encoder_layer(P, lp, f0[b][None], f0[b][map0[b]][None], nhead, 'geo')[0]
                if map1[b].any():
                    f1[b] = encoder_layer(P, lp, f1[b][None], f1[b][map1[b]][None], nhead, 'geo')[0]
        elif name == 'cross':  # transformer.py:125-139: both gathers happen before either update
            g0 = [None if win0[b] is None else sample_windows(win0[b], f0[b].T.reshape(c, hh0, ww0), scale)
                  for b in range(n)]
            g1 = [None if win1[b] is None else sample_windows(win1[b], f1[b].T.reshape(c, hh1, ww1), scale)
                  for b in range(n)]
            for b in range(n):
                if g1[b] is None:
                    continue
                f0[b] = encoder_layer(P, lp, f0[b][:, None], g1[b], nhead, 'geo', None, msk1[b])[:, 0]
                f1[b] = encoder_layer(P, lp, f1[b][:, None], g0[b], nhead, 'geo', None, msk0[b])[:, 0]
        else:
            raise KeyError(name)
    return f0, f1   # the trailing LayerNorm `norm` exists in the state dict but is not applied (:144-145)


# --------------------------------------------------------------------------------------
# a13  fine window extraction   (model/loftr_src/loftr/loftr_module/fine_preprocess.py:30-74)
# --------------------------------------------------------------------------------------


def fine_windows(feat_f: torch.Tensor, b_ids, cell_ids, w_c: int, stride: int, W: int):
    """Equivalent of F.unfold(kernel W, stride, padding W//2) + gather: [M, W*W, C_f] with window
    index ky*W+kx and zero padding outside the map (:41-56)."""
    n, c, hf, wf = feat_f.shape
    pad = W // 2
    fp = F.pad(feat_f, (pad, pad, pad, pad))
    cy, cx = (cell_ids // w_c) * stride, (cell_ids % w_c) * stride
    r = torch.arange(W)
    yy = (cy[:, None, None] + r[None, :, None]).expand(-1, W, W)
    xx = (cx[:, None, None] + r[None, None, :]).expand(-1, W, W)
    out = fp[b_ids[:, None, None], :, yy, xx]          # [M, W, W, C]
    return out.reshape(-1, W * W, c)


def fine_preprocess(P, feat_f0, feat_f1, feat_c0, feat_c1, data, W: int = 5):
    stride = int(data['hw0_f'][0] // data['hw0_c'][0])
    b, i, j = data['b_ids'], data['i_ids'], data['j_ids']
    cf = feat_f0.shape[1]
    if b.shape[0] == 0:
        return torch.empty(0, W * W, cf), torch.empty(0, W * W, cf)
    w0 = fine_windows(feat_f0, b, i, int(data['hw0_c'][1]), stride, W)
    w1 = fine_windows(feat_f1, b, j, int(data['hw1_c'][1]), stride, W)
    cwin = F.linear(torch.cat([feat_c0[b, i], feat_c1[b, j]], 0),
                    P['fine_preprocess.down_proj.weight'], P['fine_preprocess.down_proj.bias'])
    both = torch.cat([torch.cat([w0, w1], 0), cwin[:, None].expand(-1, W * W, -1)], -1)
    both = F.linear(both, P['fine_preprocess.merge_feat.weight'], P['fine_preprocess.merge_feat.bias'])
    return torch.chunk(both, 2, dim=0)


# --------------------------------------------------------------------------------------
# a14 + a15  fine matching   (model/fine_matching2.py:21-126)
# --------------------------------------------------------------------------------------


def fine_match(f0, f1, data, temperature: float, thr: float):
    """Returns the dict entries FineMatching2 writes.  M==0: fine_matrix empty and mkpts*_f =
    mkpts*_c, mconf/m_bids untouched (:34-42).  Otherwise 25x25 dual-softmax (:52-60); the kept
    cell pair is the global arg-max of each match if it exceeds thr (:73-91); coordinates (:93-116)."""
    M, WW, C = f0.shape
    if M == 0:
        return {'fine_matrix': torch.empty(0, WW, WW), 'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': data['mkpts1_c']}
    W = int(math.sqrt(WW))
    conf = dual_softmax(f0, f1, temperature)
    mask = conf > thr
    mask = mask & (conf == conf.max(dim=2, keepdim=True)[0]) & (conf == conf.max(dim=1, keepdim=True)[0])
    top = conf.view(M, -1).argmax(1)
    onehot = torch.zeros(M, WW * WW, dtype=torch.bool)
    onehot[torch.arange(M), top] = True
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
    return {'fine_matrix': conf, 'm_bids': fine_b, 'mkpts0_f': mk0 * fscale0, 'mkpts1_f': mk1 * fscale1,
            'mconf': mconf}


# --------------------------------------------------------------------------------------
# backbone (out of HIP scope; restated so the oracle is end-to-end)
#   model/loftr_src/loftr/backbone/resnet_fpn.py:43-118, eval-mode BatchNorm
# --------------------------------------------------------------------------------------


def _bn(P, name, x):
    return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'], P[name + '.weight'],
                        P[name + '.bias'], training=False, eps=1e-5)


def _basic_block(P, name, x, stride):
    y = F.relu(_bn(P, name + '.bn1', F.conv2d(x, P[name + '.conv1.weight'], stride=stride, padding=1)))
    y = _bn(P, name + '.bn2', F.conv2d(y, P[name + '.conv2.weight'], padding=1))
    if stride != 1:
        x = _bn(P, name + '.downsample.1', F.conv2d(x, P[name + '.downsample.0.weight'], stride=stride))
    return F.relu(x + y)


def backbone(P, x, prefix='backbone.'):
    Q = {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}
    x0 = F.relu(_bn(Q, 'bn1', F.conv2d(x, Q['conv1.weight'], stride=2, padding=3)))
    x1 = _basic_block(Q, 'layer1.1', _basic_block(Q, 'layer1.0', x0, 1), 1)
    x2 = _basic_block(Q, 'layer2.1', _basic_block(Q, 'layer2.0', x1, 2), 1)
    x3 = _basic_block(Q, 'layer3.1', _basic_block(Q, 'layer3.0', x2, 2), 1)
    x3o = F.conv2d(x3, Q['layer3_outconv.weight'])
    x2o = F.conv2d(x2, Q['layer2_outconv.weight'])
    up = F.interpolate(x3o, size=x2o.shape[2:], mode='bilinear', align_corners=True)
    t = F.conv2d(x2o + up, Q['layer2_outconv2.0.weight'], padding=1)
    t = F.leaky_relu(_bn(Q, 'layer2_outconv2.1', t), 0.01)
    x2o = F.conv2d(t, Q['layer2_outconv2.3.weight'], padding=1)
    x1o = F.conv2d(x1, Q['layer1_outconv.weight'])
    up = F.interpolate(x2o, size=x1o.shape[2:], mode='bilinear', align_corners=True)
    t = F.conv2d(x1o + up, Q['layer1_outconv2.0.weight'], padding=1)
    t = F.leaky_relu(_bn(Q, 'layer1_outconv2.1', t), 0.01)
    x1o = F.conv2d(t, Q['layer1_outconv2.3.weight'], padding=1)
    return x3o, x1o


# --------------------------------------------------------------------------------------
# a16  full forward   (model/full_model.py:39-123)
# --------------------------------------------------------------------------------------


def geoformer_forward(P, data, loftr_cfg=None, geo_cfg=None, homography_fn: Callable = None,
                      record: Optional[dict] = None, feats=None):
    """Mutates and returns `data` like the reference.  `feats` = optional precomputed backbone
    outputs ((c0, f0), (c1, f1)) so that the post-backbone path can be driven with identical
    inputs on both sides of a parity test."""
    loftr_cfg = loftr_cfg or default_loftr_config()
    geo_cfg = geo_cfg or default_geo_config()
    thr = geo_cfg['coarse_thr']                      # full_model.py:31
    temp = loftr_cfg['match_coarse']['dsmax_temperature']
    img0, img1 = data['image0'], data['image1']
    data.update(bs=torch.tensor(img0.size(0)), hw0_i=torch.tensor(img0.shape[2:]), hw1_i=torch.tensor(img1.shape[2:]))
    if feats is None:
        if img0.shape[2:] == img1.shape[2:]:
            c, f = backbone(P, torch.cat([img0, img1], 0))
            (c0, c1), (ff0, ff1) = c.split(img0.size(0)), f.split(img0.size(0))
        else:
            (c0, ff0), (c1, ff1) = backbone(P, img0), backbone(P, img1)
    else:
        (c0, ff0), (c1, ff1) = feats
    data.update(hw0_c=torch.tensor(c0.shape[2:]), hw1_c=torch.tensor(c1.shape[2:]),
                hw0_f=torch.tensor(ff0.shape[2:]), hw1_f=torch.tensor(ff1.shape[2:]))
    tbf = loftr_cfg['coarse']['temp_bug_fix']
    f0 = add_position_encoding(c0, tbf).flatten(2).transpose(1, 2)
    f1 = add_position_encoding(c1, tbf).flatten(2).transpose(1, 2)
    m0 = m1 = None
    if 'mask0' in data:
        m0, m1 = data['mask0'].flatten(-2), data['mask1'].flatten(-2)
    f0, f1 = local_feature_transformer(P, 'loftr_coarse.', loftr_cfg['coarse']['layer_names'],
                                       loftr_cfg['coarse']['nhead'], f0, f1, m0, m1)
    if record is not None:
        record.update(loftr_f0=f0, loftr_f1=f1)
    conf = dual_softmax(f0, f1, temp, m0, m1)
    data['conf_matrix'] = conf
    data.update(coarse_match(conf, data, thr))
    data['dect_conf_matrix'] = data['conf_matrix']
    g0, g1 = geo_module(P, c0, c1, data, geo_cfg, homography_fn, record)
    if record is not None:
        record.update(geo_f0=g0, geo_f1=g1)
    conf = dual_softmax(g0, g1, temp, m0, m1)
    data['conf_matrix'] = conf
    data.update(coarse_match(conf, data, thr))
    W = loftr_cfg['fine_window_size']
    data['W'] = torch.tensor(W)
    u0, u1 = fine_preprocess(P, ff0, ff1, g0, g1, data, W)
    if u0.size(0) != 0:
        u0, u1 = local_feature_transformer(P, 'loftr_fine.', loftr_cfg['fine']['layer_names'],
                                           loftr_cfg['fine']['nhead'], u0, u1)
    if record is not None:
        record.update(fine_f0=u0, fine_f1=u1)
    data.update(fine_match(u0, u1, data, geo_cfg['fine_temperature'], geo_cfg['fine_thr']))
    return data


# --------------------------------------------------------------------------------------
# a16 in 16-bit STORAGE MODE: the same forward with the round trips of the product's fast modes (fp16 / bf16
# storage, fp32 accumulation) at the points where its kernels round.  Used by the end-to-end parity tests of those
# modes: coarse indices are then compared bit for bit, not by overlap with the fp32 run.
#   position encoding      out = rt(x + pe)                                       (k_pos_encode.hip)
#   coarse LoFTR layers    encoder_layer_fused                                    (k9_encoder_fused.hip)
#   dual softmax           fp32 arithmetic on the rounded features                (k1_dual_softmax.hip)
#   Geo layers             q, k, v = rt(W x) (K3); self: flash attention over 32-key tiles with the probabilities
#                          rounded for the P.V product (k4_attention.hip); cross: fp32 softmax over the 25 window keys
#                          (k_geo.hip); message rounded; rest = _finish_fused
#   fine level             FinePreprocess linears rounded per kernel (K3), loftr_fine = encoder_layer_chain,
#                          FineMatching2 in fp32 on the rounded features (k_fine.hip)
# --------------------------------------------------------------------------------------


K4_DEFER_LOG2 = 8.0      # geoformer_amd/csrc/k4_attention.hip:K4_DEFER


def _flash_self_attention(q, k, v, st, tile: int = 32):
    """q [L,H,D], k, v [K,H,D] (rounded) -> [L,H,D]: online softmax over key tiles exactly as attn_self runs it in the 16-bit
    modes (round 5) - the softmax scale lives in the query operand, q' = round(q * log2(e) / sqrt(D)) to the storage type, so the
    logits x = q' . k are the exponent's log2 argument; a reference m per query that starts at the first tile's maximum and then
    moves up only when a tile's maximum exceeds it by more than 8 (the kernel's deferred maximum, a per-query rule);
    probabilities 2^(x - m) rounded to the storage type for the P.V product while their sum stays fp32, rescale by 2^(m_old - m_new)
    where m moved, one division at the end.  (Mathematically the reference's softmax(QK^T / sqrt(D)) V, geo_attention.py:72-101,
    whatever the threshold; the one extra rounding is that of q * c.  With st = float32 nothing is rounded.)"""
    L, H, D = q.shape
    K = k.shape[0]
    c = math.log2(math.e) / D ** .5
    qs = rt(q * c, st)
    m = torch.zeros(L, H)
    l = torch.zeros(L, H)
    o = torch.zeros(L, H, D)
    for t0 in range(0, K, tile):
        kt, vt = k[t0:t0 + tile], v[t0:t0 + tile]
        x = torch.einsum('lhd,shd->lhs', qs, kt)          # log2 units
        tmax = x.max(dim=2)[0] - m
        need = (tmax > K4_DEFER_LOG2) if t0 > 0 else torch.ones_like(tmax, dtype=torch.bool)
        d = torch.where(need, tmax, torch.zeros_like(tmax))
        alpha = torch.exp2(-d) if t0 > 0 else torch.zeros_like(d)      # first tile: O = l = 0
        m = m + d
        p = torch.exp2(x - m[..., None])
        l = l * alpha + p.sum(dim=2)
        o = o * alpha[..., None] + torch.einsum('lhs,shd->lhd', rt(p, st), vt)
    return rt(o / l[..., None], st)


def _geo_layer_storage(P, prefix, x, source, nhead, st, kv_mask=None, flash=False, round_out=True):
    """Geo encoder layer: x [n,L,C], source [n,S,C] (rounded)."""
    n, _, c = x.shape
    d = c // nhead
    W = lambda name: rt(P[prefix + name], st)                                    # noqa: E731
    q = rt(F.linear(x, W('q_proj.weight')), st).view(n, -1, nhead, d)
    k = rt(F.linear(source, W('k_proj.weight')), st).view(n, -1, nhead, d)
    v = rt(F.linear(source, W('v_proj.weight')), st).view(n, -1, nhead, d)
    if flash:
        msg = torch.stack([_flash_self_attention(q[i], k[i], v[i], st) for i in range(n)])
    else:
        msg = rt(full_attention(q, k, v, None, kv_mask), st)
    return _finish_fused(P, prefix, x, msg.reshape(n, -1, c), 'geo', st, round_out)


def geo_module_storage(P, f0, f1, hw0, hw1, data, geo_cfg, homography_fn: Callable, st, round_last=True):
    """geo_module above with f0/f1 = the rounded position-encoded maps [N,L,C] / [N,S,C].  round_last=False: the LAST layer's outputs stay
    fp32 (tools/rounding_ablation.py)."""
    n, _, c = f0.shape
    (hh0, ww0), (hh1, ww1) = hw0, hw1
    H0, W0 = data['image0'].shape[2:]
    H1, W1 = data['image1'].shape[2:]
    scale = int(data['hw0_i'][0] // data['hw0_c'][0])
    wsz = geo_cfg['window_size']
    per_sample_scale = 'scale0' in data
    win0, win1, msk0, msk1 = [], [], [], []
    map0 = torch.zeros(n, hh0 * ww0, dtype=torch.bool)
    map1 = torch.zeros(n, hh1 * ww1, dtype=torch.bool)
    for b in range(n):
        sel = data['m_bids'] == b
        kp0, kp1 = data['mkpts0_c'][sel].long(), data['mkpts1_c'][sel].long()
        if per_sample_scale:
            kp0 = (kp0 / (scale * data['scale0'][b]) * scale).long()
            kp1 = (kp1 / (scale * data['scale1'][b]) * scale).long()
        M = None
        if len(kp0) > 8:
            M, inl = homography_fn(kp0.numpy(), kp1.numpy())
        if M is not None:
            keep = torch.from_numpy(inl[:, 0] == 1)
            kp0, kp1 = kp0[keep], kp1[keep]
            Md = torch.from_numpy(M)
            s0 = scale * data['scale0'][b] if per_sample_scale else scale
            s1 = scale * data['scale1'][b] if per_sample_scale else scale
            p1 = warp_points(map_keypoints(H0, W0, scale), Md.to(torch.float32))
            k1w, m1w = make_windows(p1, (H1, W1), wsz, s1)
            p0 = warp_points(map_keypoints(H1, W1, scale), torch.inverse(Md[None])[0].to(torch.float32))
            k0w, m0w = make_windows(p0, (H0, W0), wsz, s0)
            win0.append(k0w); win1.append(k1w); msk0.append(m0w); msk1.append(m1w)
        else:
            win0.append(None); win1.append(None); msk0.append(None); msk1.append(None)
        map0[b, (kp0[:, 1] // scale) * ww0 + kp0[:, 0] // scale] = True
        map1[b, (kp1[:, 1] // scale) * ww1 + kp1[:, 0] // scale] = True
    nhead = geo_cfg['nhead']
    f0, f1 = f0.clone(), f1.clone()
    for idx, name in enumerate(geo_cfg['layer_names']):
        lp = f'geo_module.des_transformer.layers.{idx}.'
        ro = round_last or idx != len(geo_cfg['layer_names']) - 1
        if name == 'self':
            for b in range(n):
                if map0[b].any():
                    f0[b] = _geo_layer_storage(P, lp, f0[b][None], f0[b][map0[b]][None], nhead, st, flash=True, round_out=ro)[0]
                if map1[b].any():
                    f1[b] = _geo_layer_storage(P, lp, f1[b][None], f1[b][map1[b]][None], nhead, st, flash=True, round_out=ro)[0]
        else:
            g0 = [None if win0[b] is None else sample_windows(win0[b], f0[b].T.reshape(c, hh0, ww0), scale) for b in range(n)]
            g1 = [None if win1[b] is None else sample_windows(win1[b], f1[b].T.reshape(c, hh1, ww1), scale) for b in range(n)]
            for b in range(n):
                if g1[b] is None:
                    continue
                f0[b] = _geo_layer_storage(P, lp, f0[b][:, None], g1[b], nhead, st, kv_mask=msk1[b], round_out=ro)[:, 0]
                f1[b] = _geo_layer_storage(P, lp, f1[b][:, None], g0[b], nhead, st, kv_mask=msk0[b], round_out=ro)[:, 0]
    return f0, f1


def fine_preprocess_storage(P, feat_f0, feat_f1, feat_c0, feat_c1, data, st, W: int = 5):
    """fine_preprocess above as the product evaluates it: merge_feat(cat([win, down_proj(c)])) = W_win win + ctx with
    ctx = W_ctx down_proj(c) + b computed once per match; every linear's output is rounded."""
    stride = int(data['hw0_f'][0] // data['hw0_c'][0])
    b, i, j = data['b_ids'], data['i_ids'], data['j_ids']
    cf = feat_f0.shape[1]
    if b.shape[0] == 0:
        return torch.empty(0, W * W, cf), torch.empty(0, W * W, cf)
    win = torch.cat([fine_windows(feat_f0, b, i, int(data['hw0_c'][1]), stride, W),
                     fine_windows(feat_f1, b, j, int(data['hw1_c'][1]), stride, W)], 0)
    ccat = torch.cat([feat_c0[b, i], feat_c1[b, j]], 0)
    mw = rt(P['fine_preprocess.merge_feat.weight'], st)
    down = rt(F.linear(ccat, rt(P['fine_preprocess.down_proj.weight'], st), P['fine_preprocess.down_proj.bias']), st)
    ctx = rt(F.linear(down, mw[:, cf:], P['fine_preprocess.merge_feat.bias']), st)
    both = rt(F.linear(win, mw[:, :cf]) + ctx[:, None], st)
    return torch.chunk(both, 2, dim=0)


def geoformer_forward_storage(P, data, st, loftr_cfg=None, geo_cfg=None, homography_fn: Callable = None, feats=None,
                              record: Optional[dict] = None):
    """geoformer_forward in the 16-bit storage mode `st` (torch.float16 / torch.bfloat16), from backbone features
    `feats` = ((c0, f0), (c1, f1)) - the backbone (MIOpen) is outside the HIP path, its outputs are the inputs here."""
    loftr_cfg = loftr_cfg or default_loftr_config()
    geo_cfg = geo_cfg or default_geo_config()
    thr, temp = geo_cfg['coarse_thr'], loftr_cfg['match_coarse']['dsmax_temperature']
    img0, img1 = data['image0'], data['image1']
    data.update(bs=torch.tensor(img0.size(0)), hw0_i=torch.tensor(img0.shape[2:]), hw1_i=torch.tensor(img1.shape[2:]))
    (c0, ff0), (c1, ff1) = [(rt(c, st), rt(f, st)) for c, f in feats]
    data.update(hw0_c=torch.tensor(c0.shape[2:]), hw1_c=torch.tensor(c1.shape[2:]),
                hw0_f=torch.tensor(ff0.shape[2:]), hw1_f=torch.tensor(ff1.shape[2:]))
    tbf = loftr_cfg['coarse']['temp_bug_fix']
    pe0 = rt(add_position_encoding(c0, tbf), st).flatten(2).transpose(1, 2).contiguous()
    pe1 = rt(add_position_encoding(c1, tbf), st).flatten(2).transpose(1, 2).contiguous()
    m0 = m1 = None
    if 'mask0' in data:
        m0, m1 = data['mask0'].flatten(-2), data['mask1'].flatten(-2)
    f0, f1 = pe0, pe1
    for idx, name in enumerate(loftr_cfg['coarse']['layer_names']):
        lp = f'loftr_coarse.layers.{idx}.'
        nh = loftr_cfg['coarse']['nhead']
        if name == 'self':
            f0 = encoder_layer_fused(P, lp, f0, f0, nh, st, m0, m0)
            f1 = encoder_layer_fused(P, lp, f1, f1, nh, st, m1, m1)
        else:
            f0 = encoder_layer_fused(P, lp, f0, f1, nh, st, m0, m1)
            f1 = encoder_layer_fused(P, lp, f1, f0, nh, st, m1, m0)
    conf = dual_softmax(f0, f1, temp, m0, m1)
    data['conf_matrix'] = conf
    data.update(coarse_match(conf, data, thr))
    data['dect_conf_matrix'] = data['conf_matrix']
    if tbf:
        raise NotImplementedError('storage mode assumes the GeoModule shares the position-encoded maps (temp_bug_fix False)')
    g0, g1 = geo_module_storage(P, pe0, pe1, tuple(c0.shape[2:]), tuple(c1.shape[2:]), data, geo_cfg, homography_fn, st)
    conf = dual_softmax(g0, g1, temp, m0, m1)
    data['conf_matrix'] = conf
    data.update(coarse_match(conf, data, thr))
    W = loftr_cfg['fine_window_size']
    data['W'] = torch.tensor(W)
    u0, u1 = fine_preprocess_storage(P, ff0, ff1, g0, g1, data, st, W)
    if u0.size(0) != 0:
        nh = loftr_cfg['fine']['nhead']
        for idx, name in enumerate(loftr_cfg['fine']['layer_names']):
            lp = f'loftr_fine.layers.{idx}.'
            if name == 'self':
                u0 = encoder_layer_chain(P, lp, u0, u0, nh, st)
                u1 = encoder_layer_chain(P, lp, u1, u1, nh, st)
            else:
                u0 = encoder_layer_chain(P, lp, u0, u1, nh, st)
                u1 = encoder_layer_chain(P, lp, u1, u0, nh, st)
    if record is not None:
        record.update(loftr_f0=f0, loftr_f1=f1, geo_f0=g0, geo_f1=g1, fine_f0=u0, fine_f1=u1)
    data.update(fine_match(u0, u1, data, geo_cfg['fine_temperature'], geo_cfg['fine_thr']))
    return data


# --------------------------------------------------------------------------------------
# deterministic closed-form weights (fixtures hold inputs/outputs only, never 57 MB of weights)
# --------------------------------------------------------------------------------------


def _hash_uniform(n: int, salt: int):
    """n deterministic pseudo-random doubles in [0,1): murmur3 finaliser of (index, salt) in pure
    integer arithmetic (numpy uint64), so any re-implementation can reproduce them bit-exactly."""
    import numpy as np
    x = (np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(16))) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(13))) * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return x.astype(np.float64) / 4294967296.0


def closed_form_fill(state_dict: Dict[str, torch.Tensor], gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Fills every tensor of a GeoFormer state dict in place from its NAME and shape only, so that
    the reference model and any re-implementation get bit-identical weights without shipping
    57 MB of them.  Matrices/convs: hash-uniform with Xavier-uniform bound; LayerNorm/BatchNorm
    affine near identity; BN running stats well-conditioned."""
    import zlib
    for name, t in state_dict.items():
        if t.dtype not in (torch.float32, torch.float64):
            t.zero_()   # num_batches_tracked
            continue
        u = torch.from_numpy(_hash_uniform(t.numel(), zlib.crc32(name.encode()))) * 2.0 - 1.0
        if name.endswith('running_var'):
            val = 1.0 + 0.1 * u.abs()
        elif name.endswith('running_mean'):
            val = 0.02 * u
        elif t.dim() == 1 and name.endswith('weight'):
            val = 1.0 + 0.05 * u
        elif t.dim() == 1:
            val = 0.02 * u
        else:
            rf = t[0][0].numel() if t.dim() > 2 else 1
            fan_in, fan_out = t.shape[1] * rf, t.shape[0] * rf
            val = gain * u * math.sqrt(6.0 / (fan_in + fan_out))
        t.copy_(val.to(t.dtype).view_as(t))
    return state_dict


def state_dict_schema(loftr_cfg=None, geo_cfg=None) -> Dict[str, tuple]:
    """name -> shape of the 253-entry GeoFormer state dict (model/full_model.py:19-37 and the
    modules it builds), used to make weights without instantiating anything."""
    loftr_cfg = loftr_cfg or default_loftr_config()
    geo_cfg = geo_cfg or default_geo_config()
    S: Dict[str, tuple] = {}

    def bn(name, c):
        S[name + '.weight'] = (c,); S[name + '.bias'] = (c,)
        S[name + '.running_mean'] = (c,); S[name + '.running_var'] = (c,)
        S[name + '.num_batches_tracked'] = ()

    d0 = loftr_cfg['resnetfpn']['initial_dim']
    b1, b2, b3 = loftr_cfg['resnetfpn']['block_dims']
    S['backbone.conv1.weight'] = (d0, 1, 7, 7); bn('backbone.bn1', d0)
    cin = d0
    for li, (dim, stride) in enumerate([(b1, 1), (b2, 2), (b3, 2)], start=1):
        for bi in range(2):
            p = f'backbone.layer{li}.{bi}'
            st = stride if bi == 0 else 1
            S[p + '.conv1.weight'] = (dim, cin, 3, 3); S[p + '.conv2.weight'] = (dim, dim, 3, 3)
            bn(p + '.bn1', dim); bn(p + '.bn2', dim)
            if st != 1:
                S[p + '.downsample.0.weight'] = (dim, cin, 1, 1); bn(p + '.downsample.1', dim)
            cin = dim
    S['backbone.layer3_outconv.weight'] = (b3, b3, 1, 1)
    S['backbone.layer2_outconv.weight'] = (b3, b2, 1, 1)
    S['backbone.layer2_outconv2.0.weight'] = (b3, b3, 3, 3); bn('backbone.layer2_outconv2.1', b3)
    S['backbone.layer2_outconv2.3.weight'] = (b2, b3, 3, 3)
    S['backbone.layer1_outconv.weight'] = (b2, b1, 1, 1)
    S['backbone.layer1_outconv2.0.weight'] = (b2, b2, 3, 3); bn('backbone.layer1_outconv2.1', b2)
    S['backbone.layer1_outconv2.3.weight'] = (b1, b2, 3, 3)

    def enc(prefix, c):
        for nm in ('q_proj', 'k_proj', 'v_proj', 'merge'):
            S[f'{prefix}{nm}.weight'] = (c, c)
        S[prefix + 'mlp.0.weight'] = (2 * c, 2 * c); S[prefix + 'mlp.2.weight'] = (c, 2 * c)
        for nm in ('norm1', 'norm2'):
            S[f'{prefix}{nm}.weight'] = (c,); S[f'{prefix}{nm}.bias'] = (c,)

    cc, cf = loftr_cfg['coarse']['d_model'], loftr_cfg['fine']['d_model']
    for i in range(len(loftr_cfg['coarse']['layer_names'])):
        enc(f'loftr_coarse.layers.{i}.', cc)
    S['fine_preprocess.down_proj.weight'] = (cf, cc); S['fine_preprocess.down_proj.bias'] = (cf,)
    S['fine_preprocess.merge_feat.weight'] = (cf, 2 * cf); S['fine_preprocess.merge_feat.bias'] = (cf,)
    for i in range(len(loftr_cfg['fine']['layer_names'])):
        enc(f'loftr_fine.layers.{i}.', cf)
    for i in range(len(geo_cfg['layer_names'])):
        enc(f'geo_module.des_transformer.layers.{i}.', cc)
    S['geo_module.des_transformer.norm.weight'] = (cc,); S['geo_module.des_transformer.norm.bias'] = (cc,)
    return S


def make_weights(loftr_cfg=None, geo_cfg=None, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    sd = {}
    for k, shp in state_dict_schema(loftr_cfg, geo_cfg).items():
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
    return closed_form_fill(sd, gain)
