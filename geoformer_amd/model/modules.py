"""Host-side mirror of the reference's module interfaces for the matching path.  Same class names,
constructor arguments, forward signatures and state-dict keys as the reference; the arithmetic runs
in the HIP kernels of libgeoformer_hip.so (geoformer_amd.ops).  PyTorch supplies device memory,
streams, the plain library GEMMs (nn.Linear -> hipBLASLt) and LayerNorm.

There is no CPU path: every forward here needs CUDA(HIP) tensors and the built library.
"""
import copy
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
from .. import fused, ops


# ---------------------------------------------------------------------------------------------
# position encoding  (reference: model/loftr_src/loftr/utils/position_encoding.py:6-42)
# ---------------------------------------------------------------------------------------------
class PositionEncodingSine(nn.Module):
    """x [N,C,H,W] -> x + pe, returned FLATTENED as [N, H*W, C] (the reference's callers permute and
    reshape right after, model/full_model.py:69-77, model/geo_module.py:28-29).  The sin/cos table is
    built on the host with the reference's own formula - including its operator-precedence quirk when
    temp_bug_fix is False (:28) - cached per (H, W) in [H, W, C] layout, and added by gf_pos_encode."""

    def __init__(self, d_model, max_shape=(256, 256), temp_bug_fix=False):
        super().__init__()
        self.d_model, self.max_shape, self.temp_bug_fix = d_model, tuple(max_shape), temp_bug_fix
        self._tables = {}

    def table(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._tables:
            if h > self.max_shape[0] or w > self.max_shape[1]:
                raise ValueError(f'feature map {h}x{w} exceeds max_shape {self.max_shape}')
            c = self.d_model
            ypos = torch.ones(h, w).cumsum(0).float().unsqueeze(0)
            xpos = torch.ones(h, w).cumsum(1).float().unsqueeze(0)
            k2 = torch.arange(0, c // 2, 2).float()
            if self.temp_bug_fix:
                div = torch.exp(k2 * (-math.log(10000.0) / (c // 2)))
            else:
                div = torch.exp(k2 * (-math.log(10000.0) / c // 2))
            div = div[:, None, None]
            pe = torch.zeros(c, h, w)
            pe[0::4], pe[1::4] = torch.sin(xpos * div), torch.cos(xpos * div)
            pe[2::4], pe[3::4] = torch.sin(ypos * div), torch.cos(ypos * div)
            self._tables[key] = pe.permute(1, 2, 0).contiguous().to(device)
        return self._tables[key]

    def forward(self, x, out_dtype=None, out=None):
        _, _, h, w = x.shape
        return ops.pos_encode(x, self.table(h, w, x.device), out_dtype or x.dtype, out)


# ---------------------------------------------------------------------------------------------
# encoder layer  (reference: model/loftr_src/loftr/loftr_module/transformer.py:9-60 with ReLU and
# linear attention; model/geo_transformer/transformer.py:9-66 with Tanh and full attention)
# ---------------------------------------------------------------------------------------------
class LoFTREncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, attention='linear', activation='relu'):
        super().__init__()
        self.dim, self.nhead, self.d_model = d_model // nhead, nhead, d_model
        self.attention_kind, self.activation = attention, activation
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(nn.Linear(d_model * 2, d_model * 2, bias=False),
                                 nn.ReLU(True) if activation == 'relu' else nn.Tanh(),
                                 nn.Linear(d_model * 2, d_model, bias=False))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self._cache = {}

    # weights in compute dtype; k_proj|v_proj fused so one GEMM projects keys and values; LayerNorm
    # affine terms stay fp32 (the K3 epilogue normalises in fp32)
    def weights(self, dtype):
        w = self._cache.get(dtype)
        if w is None:
            f32 = torch.float32
            w = {'q': self.q_proj.weight.detach().to(dtype).contiguous(),
                 'kv': torch.cat([self.k_proj.weight, self.v_proj.weight], 0).detach().to(dtype).contiguous(),
                 'qkv': torch.cat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], 0).detach().to(dtype).contiguous(),
                 'merge': self.merge.weight.detach().to(dtype).contiguous(),
                 'w1': self.mlp[0].weight.detach().to(dtype).contiguous(),
                 'w2': self.mlp[2].weight.detach().to(dtype).contiguous(),
                 'n1': (self.norm1.weight.detach().to(f32).contiguous(), self.norm1.bias.detach().to(f32).contiguous()),
                 'n2': (self.norm2.weight.detach().to(f32).contiguous(), self.norm2.bias.detach().to(f32).contiguous())}
            if self.fusable_window(dtype, 1):
                # K11: the fine level's layer in one launch (csrc/k11_fine_layer.hip)
                w['ln'] = torch.cat([w['n1'][0], w['n1'][1], w['n2'][0], w['n2'][1]]).contiguous()
                c = self.d_model
                w['stream_fine'] = fused.pack_fine_layer_stream(w['q'], w['kv'][:c], w['kv'][c:], w['merge'], w['w1'], w['w2'])
            if self.fusable(dtype):
                # K9: the weights as the fragment streams the fused kernels consume (fused.py), packed once
                c = self.d_model
                w['ln'] = torch.cat([w['n1'][0], w['n1'][1], w['n2'][0], w['n2'][1]]).contiguous()
                w['stream_finish'] = fused.pack_layer_stream(None, w['merge'], w['w1'], w['w2'])
                if self.attention_kind == 'linear':
                    w['stream'] = fused.pack_layer_stream(w['q'], w['merge'], w['w1'], w['w2'])
                    w['stream_kv'] = fused.pack_kv_stream(w['kv'][:c], w['kv'][c:])
            self._cache[dtype] = w
        return w

    def fusable(self, dtype):
        """The fused two-launch form (csrc/k9_encoder_fused.hip) exists for 16-bit storage at d_model 256, with
        8 heads of 32 when the layer's own (linear) attention is part of it."""
        return dtype != torch.float32 and self.d_model == 256 and (self.attention_kind != 'linear' or self.nhead == 8)

    def fusable_window(self, dtype, seq_len):
        """The one-launch fine-level form (csrc/k11_fine_layer.hip): 16-bit storage, d_model 128, 8 heads of 16, linear
        attention, ReLU, sequences (windows) of at most 32 tokens."""
        return (dtype != torch.float32 and self.d_model == 128 and self.nhead == 8 and self.attention_kind == 'linear'
                and self.activation == 'relu' and seq_len <= 32)

    def invalidate(self):
        self._cache = {}

    def project_q(self, x):
        return ops.linear(x, self.weights(x.dtype)['q'])

    def project_kv(self, source):
        kv = ops.linear(source, self.weights(source.dtype)['kv'])
        c = self.d_model
        return kv[..., :c], kv[..., c:]

    def project_qkv(self, x):
        """q, k and v of the SAME tokens in one GEMM (x read once, one launch): row-strided views of a [..., 3C] tensor."""
        qkv = ops.linear(x, self.weights(x.dtype)['qkv'])
        c = self.d_model
        return qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]

    def finish(self, x, message, row_flag=None, flag_rows=0, out=None):
        """x + norm2(mlp([x, norm1(merge(message))])) in three K3 launches: merge+LN, mlp.0 on the
        two-part operand (no concat) + activation, mlp.2+LN+residual.  row_flag (int32 per `flag_rows`
        rows) == 0 leaves x unchanged - GeoTransformer's per-sample 'layer skipped' cases."""
        w = self.weights(x.dtype)
        if self.fusable(x.dtype):
            return fused.encoder_layer(x, w['stream_finish'], w['ln'], self.norm1.eps, self.norm2.eps,
                                       0 if self.activation == 'relu' else 1, msg=message, row_flag=row_flag,
                                       flag_rows=flag_rows, out=out)
        act = ops.EPI_RELU if self.activation == 'relu' else ops.EPI_TANH
        message = ops.linear(message, w['merge'], epilogue=ops.EPI_LN, ln=w['n1'], eps=self.norm1.eps)
        hid = ops.linear(x, w['w1'], a2=message, epilogue=act)
        return ops.linear(hid, w['w2'], epilogue=ops.EPI_LN_RES, ln=w['n2'], eps=self.norm2.eps, residual=x,
                          row_flag=row_flag, flag_rows=flag_rows, out=out)

    def forward(self, x, source, x_mask: Optional[torch.Tensor] = None, source_mask: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None):
        """x [N,L,C], source [N,S,C] -> [N,L,C]  (linear-attention flavour; the geometry-guided flavours
        are driven by GeoTransformer, which owns the token lists / windows)."""
        if self.attention_kind != 'linear':
            raise NotImplementedError('full attention layers are driven by GeoTransformer')
        if (x_mask is None and source_mask is None and x.shape == source.shape and self.fusable_window(x.dtype, x.shape[1])
                and x.is_contiguous() and source.is_contiguous()):
            # one launch: the window's tokens stay on the CU through all six GEMMs (the fine level: [M, 25, 128] windows)
            w = self.weights(x.dtype)
            return fused.fine_layer(x, source, w['stream_fine'], w['ln'], self.norm1.eps, self.norm2.eps, out=out)
        if self.fusable(x.dtype):
            # two launches, three token-row transfers: source -> state (k, v never reach HBM), x -> out
            w = self.weights(x.dtype)
            state = fused.encoder_kv_state(source, w['stream_kv'], source_mask)
            return fused.encoder_layer(x, w['stream'], w['ln'], self.norm1.eps, self.norm2.eps,
                                       0 if self.activation == 'relu' else 1, kv_state=state, source_len=source.shape[1],
                                       q_mask=x_mask, out=out)
        q = self.project_q(x)
        k, v = self.project_kv(source)
        message = ops.linear_attention(q, k, v, self.nhead, x_mask, source_mask)
        return self.finish(x, message, out=out)

    def kv_state(self, source, source_mask=None, out=None):
        """The linear-attention state of `source` under this layer's k / v projections (fused 16-bit form)."""
        return fused.encoder_kv_state(source, self.weights(source.dtype)['stream_kv'], source_mask, out=out)

    def forward_state(self, x, state, source_len, x_mask=None, out=None, tail_layer=None, tail_first=0, tail_out=None):
        """The layer on x given the source's state (fused 16-bit form, one launch).  tail_layer: the layer that will read THIS
        call's output rows as its source - the images tail_first.. also leave their state under that layer's k / v projections
        (returned second; csrc/k9_encoder_fused.hip's state tail), so the consumer needs no pass of its own over the features."""
        w = self.weights(x.dtype)
        tail = None if tail_layer is None else tail_layer.weights(x.dtype)['stream_kv']
        return fused.encoder_layer(x, w['stream'], w['ln'], self.norm1.eps, self.norm2.eps, 0 if self.activation == 'relu' else 1,
                                   kv_state=state, source_len=source_len, q_mask=x_mask, out=out, tail_stream=tail,
                                   tail_first=tail_first, tail_out=tail_out)


def _adjacent_halves(a, b):
    """[2n, ...] view over a and b when they are the first and the second half of one contiguous buffer (e.g. out[:M], out[M:]), else None."""
    if (a.shape != b.shape or a.dtype != b.dtype or not a.is_contiguous() or not b.is_contiguous() or a.numel() == 0
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.storage_offset() != a.storage_offset() + a.numel()):
        return None
    return a.as_strided((2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())


class LocalFeatureTransformer(nn.Module):
    """reference: model/loftr_src/loftr/loftr_module/transformer.py:63-104"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.d_model, self.nhead, self.layer_names = config['d_model'], config['nhead'], config['layer_names']
        if config['attention'] != 'linear':
            raise NotImplementedError("only attention='linear' (the GeoFormer configuration) is built")
        layer = LoFTREncoderLayer(config['d_model'], config['nhead'], 'linear', 'relu')
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in self.layer_names])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, feat0, feat1, mask0: Optional[torch.Tensor] = None, mask1: Optional[torch.Tensor] = None,
                both: Optional[torch.Tensor] = None):
        """both: optional [2N, L, C] buffer whose halves ARE feat0 and feat1 (read only here): saves the concat."""
        assert self.d_model == feat0.size(2), 'the feature number of src and transformer must be equal'
        same = feat0.shape == feat1.shape
        n = feat0.shape[0]
        # same-shape pairs live in ONE [2N, L, C] buffer: self layers run both images in one batched launch set
        # (same weights, independent rows) and cross layers write their halves in place of a concat
        if not same:
            both = None
        elif both is None:
            both = _adjacent_halves(feat0, feat1)              # FinePreprocess hands out the two halves of ONE buffer: no concat
            if both is None:
                both = torch.cat([feat0, feat1], 0)
        mboth = torch.cat([mask0, mask1], 0) if (same and mask0 is not None) else None
        if same and both.is_cuda and all(ly.fusable(both.dtype) for ly in self.layers):
            return self._forward_fused(both, n, mask0, mask1, mboth)
        if (not same and feat0.is_cuda and feat0.dim() == 3 and feat0.shape[0] == feat1.shape[0]
                and all(ly.fusable(feat0.dtype) for ly in self.layers)):
            return self._forward_fused_unequal(feat0, feat1, mask0, mask1)
        for layer, name in zip(self.layers, self.layer_names):
            if name == 'self':
                if same:
                    both = layer(both, both, mboth, mboth)
                else:
                    feat0 = layer(feat0, feat0, mask0, mask0)
                    feat1 = layer(feat1, feat1, mask1, mask1)
            elif name == 'cross':  # feat1 attends to the UPDATED feat0 (transformer.py:99-100)
                if same:
                    nxt = torch.empty_like(both)
                    layer(both[:n], both[n:], mask0, mask1, out=nxt[:n])
                    layer(both[n:], nxt[:n], mask1, mask0, out=nxt[n:])
                    both = nxt
                else:
                    feat0 = layer(feat0, feat1, mask0, mask1)
                    feat1 = layer(feat1, feat0, mask1, mask0)
            else:
                raise KeyError
        return (both[:n], both[n:]) if same else (feat0, feat1)

    def _forward_fused_unequal(self, f0, f1, m0, m1):
        """_forward_fused for a pair whose images have DIFFERENT token counts (the HPatches loop: 480x640 against 480x608, data_io.py:16-26;
        BASELINE configs[1]) - the two images cannot share a launch, but the state tails work per launch: the call that produces an image's
        rows leaves their state for the call that reads them as its source.  Own gf_encoder_kv_state passes are left for the first layer's
        two sources and for image 0's rows in front of every later 'self' layer: 5 instead of 16 per forward (round 6; bit-identical
        features: a tail's state equals a separate pass's, tests/test_encoder_fused.py)."""
        layers, names = list(self.layers), list(self.layer_names)
        L0, L1 = f0.shape[1], f1.shape[1]
        have = {}                                                # image -> state of its current rows under the layer about to run
        for idx, (layer, name) in enumerate(zip(layers, names)):
            nxt_layer = layers[idx + 1] if idx + 1 < len(layers) else None
            nxt_name = names[idx + 1] if nxt_layer is not None else None
            if name == 'self':
                st0 = have[0] if 0 in have else layer.kv_state(f0, m0)
                st1 = have[1] if 1 in have else layer.kv_state(f1, m1)
                have = {}
                if nxt_name == 'self':                            # the next layer reads both images' new rows
                    f0, have[0] = layer.forward_state(f0, st0, L0, m0, tail_layer=nxt_layer, tail_first=0)
                else:
                    f0 = layer.forward_state(f0, st0, L0, m0)
                if nxt_layer is None:
                    f1 = layer.forward_state(f1, st1, L1, m1)
                else:                                             # 'cross': its first call reads image 1's rows
                    f1, have[1] = layer.forward_state(f1, st1, L1, m1, tail_layer=nxt_layer, tail_first=0)
            elif name == 'cross':                                 # feat1 attends to the UPDATED feat0 (transformer.py:99-100)
                st1 = have[1] if 1 in have else layer.kv_state(f1, m1)
                f0, st0 = layer.forward_state(f0, st1, L1, m0, tail_layer=layer, tail_first=0)
                have = {}
                if nxt_layer is None:
                    f1 = layer.forward_state(f1, st0, L0, m1)
                else:                                             # 'self' and 'cross' alike read image 1's new rows (first)
                    f1, have[1] = layer.forward_state(f1, st0, L0, m1, tail_layer=nxt_layer, tail_first=0)
            else:
                raise KeyError
        return f0, f1

    def _forward_fused(self, both, n, mask0, mask1, mboth):
        """The same schedule (transformer.py:82-104) on the fused 16-bit layer kernels, one launch per layer call: the state a
        call needs - phi(K)^T V of its source - is left behind by the call that PRODUCED the source rows (its state tail),
        under the consumer's k / v weights.  Only what no earlier call of this transformer produced (the first layer's
        sources, and image 0's rows in front of a 'self' layer - its producer's tail is busy with the state the second cross
        call needs) takes a gf_encoder_kv_state pass of its own: 40 image-states per 8-pair step instead of 128."""
        L = both.shape[1]
        C = self.d_model
        layers, names = list(self.layers), list(self.layer_names)
        have = {}                                                # half (0 / 1) -> state rows [n, 8448] under the NEXT layer's k / v
        for idx, (layer, name) in enumerate(zip(layers, names)):
            nxt_layer = layers[idx + 1] if idx + 1 < len(layers) else None
            nxt_name = names[idx + 1] if nxt_layer is not None else None
            if name == 'self':
                st = have.get('buf')                              # a producer's tail wrote its half straight into this buffer
                if st is None:
                    st = torch.empty(2 * n, C * 32 + C, dtype=torch.float32, device=both.device)
                    for half in (0, 1):
                        if half in have:
                            st[half * n:(half + 1) * n].copy_(have[half])
                for half, (rows, msk) in enumerate(((both[:n], mask0), (both[n:], mask1))):
                    if half not in have:
                        layer.kv_state(rows, msk, out=st[half * n:(half + 1) * n])
                have = {}
                if nxt_layer is None:
                    both = layer.forward_state(both, st, L, mboth)
                elif nxt_name == 'cross':                         # its first call reads image 1's rows
                    both, t = layer.forward_state(both, st, L, mboth, tail_layer=nxt_layer, tail_first=n)
                    have = {1: t}
                else:                                             # another 'self' layer reads both
                    both, t = layer.forward_state(both, st, L, mboth, tail_layer=nxt_layer, tail_first=0)
                    have = {0: t[:n], 1: t[n:], 'buf': t}
            elif name == 'cross':                                 # feat1 attends to the UPDATED feat0 (transformer.py:99-100)
                st1 = have[1] if 1 in have else layer.kv_state(both[n:], mask1)
                new = torch.empty_like(both)
                _, st0 = layer.forward_state(both[:n], st1, L, mask0, out=new[:n], tail_layer=layer, tail_first=0)
                have = {}
                if nxt_layer is None:
                    layer.forward_state(both[n:], st0, L, mask1, out=new[n:])
                elif nxt_name == 'self':                          # its state buffer: image 1's half from this call's tail
                    buf = torch.empty(2 * n, C * 32 + C, dtype=torch.float32, device=both.device)
                    layer.forward_state(both[n:], st0, L, mask1, out=new[n:], tail_layer=nxt_layer, tail_first=0, tail_out=buf[n:])
                    have = {1: buf[n:], 'buf': buf}
                else:                                             # another 'cross' layer: its first call reads image 1's new rows
                    _, t = layer.forward_state(both[n:], st0, L, mask1, out=new[n:], tail_layer=nxt_layer, tail_first=0)
                    have = {1: t}
                both = new
            else:
                raise KeyError
        return both[:n], both[n:]


# ---------------------------------------------------------------------------------------------
# coarse matching  (reference: model/loftr_src/loftr/utils/coarse_matching.py:25-212)
# ---------------------------------------------------------------------------------------------
class CoarseMatching(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.thr = config['thr']
        self.border_rm = 0                        # forced to 0 in the reference (:31-32)
        self.train_coarse_percent = config['train_coarse_percent']
        self.train_pad_num_gt_min = config['train_pad_num_gt_min']
        self.match_type = config['match_type']
        if self.match_type != 'dual_softmax':
            raise NotImplementedError("only match_type='dual_softmax' is built (sinkhorn needs superglue.py, "
                                      'which the reference does not ship either)')
        self.temperature = config['dsmax_temperature']
        # False: match-only mode of K1 - conf_matrix / dect_conf_matrix are None in `data` (inference.py:51-75 and the evaluation
        # harness never read them), matches bit-identical; set from geoformer_cfg['materialize_conf'] (optional key, default True)
        self.materialize_conf = True

    def forward(self, feat_c0, feat_c1, data: Dict[str, torch.Tensor], mask_c0: Optional[torch.Tensor] = None,
                mask_c1: Optional[torch.Tensor] = None, lazy: bool = False):
        """Updates data with conf_matrix, b_ids, i_ids, j_ids, m_bids, mkpts0_c, mkpts1_c, mconf.
        lazy=True keeps the match arrays at capacity with their count on the device (data['_coarse_dev'])
        and does not synchronise; GeoModule consumes that form directly."""
        scale = float(data['hw0_i'][0]) / float(data['hw0_c'][0])
        raw = ops.dual_softmax_match(feat_c0, feat_c1, self.temperature, self.thr, data['hw0_c'], data['hw1_c'], scale,
                                     mask_c0, mask_c1, data.get('scale0'), data.get('scale1'),
                                     force_one='dataset_name' in data, materialize=self.materialize_conf)
        data['conf_matrix'] = raw['conf_matrix']
        data['_coarse_dev'] = raw
        if not lazy:
            data.update(materialize_matches(raw))
        return raw


def materialize_matches(raw):
    m = int(raw['counts'][0])              # the one host sync of the coarse stage
    out = {k: raw[k][:m] for k in ('b_ids', 'i_ids', 'j_ids', 'mkpts0_c', 'mkpts1_c', 'mconf')}
    out['m_bids'] = out['b_ids']
    return out


# ---------------------------------------------------------------------------------------------
# GeoModule / GeoTransformer  (reference: model/geo_module.py, model/geo_transformer/transformer.py)
# ---------------------------------------------------------------------------------------------
class GeoTransformer(nn.Module):
    def __init__(self, config, layer_names, d_model, linear=True):
        super().__init__()
        if linear:
            raise NotImplementedError('GeoModule builds its transformer with linear=False (geo_module.py:20-21)')
        self.config, self.d_model, self.layer_names, self.nhead = config, d_model, layer_names, config['nhead']
        layer = LoFTREncoderLayer(d_model, self.nhead, 'full', 'tanh')
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in self.layer_names])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.norm = nn.LayerNorm(d_model)         # present in the checkpoint, never applied (:144-145)

    def forward(self, feat0, feat1, geo, both: Optional[torch.Tensor] = None):
        """feat0 [N,L,C], feat1 [N,S,C]; geo = dict(idx0, idx1, nidx, win0, win1, valid) on the device:
        idx*/nidx = tokens at inlier cells (self layers), win1 = cells of image1 seen from each cell of
        image0 and win0 the converse (cross layers), valid = per-sample 'homography found'.
        both: optional [2N, L, C] buffer whose halves ARE feat0 and feat1 (read only here): saves the concat."""
        assert self.d_model == feat0.size(2), 'the feature number of src and transformer must be equal'
        n, L, S = feat0.shape[0], feat0.shape[1], feat1.shape[1]
        same = feat0.shape == feat1.shape
        nk = geo['nidx']
        if same:                     # one [2N, L, C] buffer, as in LocalFeatureTransformer
            if both is None:
                both = torch.cat([feat0, feat1], 0)
            feat0, feat1 = both[:n], both[n:]
        for layer, name in zip(self.layers, self.layer_names):
            if name == 'self':       # keys/values = tokens at inlier cells; a sample without any keeps its features
                if same:
                    q, k, v = layer.project_qkv(both)
                    msg = ops.self_attention_gathered(q, k, v, geo['idx_both'], geo['nidx_both'], self.nhead)
                    both = layer.finish(both, msg, geo['nidx_both'], L)
                    feat0, feat1 = both[:n], both[n:]
                else:
                    k, v = layer.project_kv(feat0)
                    m0 = ops.self_attention_gathered(layer.project_q(feat0), k, v, geo['idx0'], nk[:, 0], self.nhead)
                    k, v = layer.project_kv(feat1)
                    m1 = ops.self_attention_gathered(layer.project_q(feat1), k, v, geo['idx1'], nk[:, 1], self.nhead)
                    feat0 = layer.finish(feat0, m0, geo['nidx_t'][0], L)
                    feat1 = layer.finish(feat1, m1, geo['nidx_t'][1], S)
            elif name == 'cross':
                # keys/values of BOTH images come from the pre-update features (the reference gathers
                # feat0_cross and feat1_cross before either update, :126-129); project, then gather.
                if same:             # q, k, v of both images in one GEMM over the [2N, L, C] buffer
                    q, k, v = layer.project_qkv(both)
                    q0, q1, k0, k1, v0, v1 = q[:n], q[n:], k[:n], k[n:], v[:n], v[n:]
                else:
                    k0, v0 = layer.project_kv(feat0)
                    k1, v1 = layer.project_kv(feat1)
                    q0, q1 = layer.project_q(feat0), layer.project_q(feat1)
                m0 = ops.window_cross_attention(q0, k1, v1, geo['win1'], geo['valid'], self.nhead, geo.get('hw0'), geo.get('hw1'))
                m1 = ops.window_cross_attention(q1, k0, v0, geo['win0'], geo['valid'], self.nhead, geo.get('hw1'), geo.get('hw0'))
                if same:
                    nxt = torch.empty_like(both)
                    layer.finish(feat0, m0, geo['valid'], L, out=nxt[:n])
                    layer.finish(feat1, m1, geo['valid'], S, out=nxt[n:])
                    both = nxt
                    feat0, feat1 = both[:n], both[n:]
                else:
                    feat0 = layer.finish(feat0, m0, geo['valid'], L)
                    feat1 = layer.finish(feat1, m1, geo['valid'], S)
            else:
                raise KeyError
        return feat0, feat1


class GeoModule(nn.Module):
    def __init__(self, config, d_model):
        super().__init__()
        self.d_model = d_model
        self.window_size = config['window_size']
        self.pos_encoding = PositionEncodingSine(d_model)
        self.des_transformer = GeoTransformer(config, config['layer_names'], d_model, linear=False)
        self.ransac_thr = 8.0                     # cv2.findHomography(..., cv2.RANSAC, 8.0) (geo_module.py:48)
        self.ransac_iters, self.ransac_seed, self.ransac_lm_iters = ops.RANSAC_ITERS, ops.RANSAC_SEED, ops.RANSAC_LM_ITERS
        # Optional HOST callback with cv2.findHomography's contract, `fn(kp0 [n,2] int64 ndarray, kp1) ->
        # (M float64 [3,3] | None, mask uint8 [n,1])`, called per sample with > 8 matches exactly like
        # geo_module.py:45-48 (e.g. lambda a, b: cv2.findHomography(a, b, cv2.RANSAC, 8.0), or a replay of
        # recorded homographies in parity tests).  None (default) = the device RANSAC, no host round trip.
        self.homography_fn = None

    def _host_homographies(self, rs, counts, n, dev):
        cnt = counts.cpu().tolist()                      # host sync (only on this optional path)
        kp0, kp1 = rs['kp0'].cpu().numpy(), rs['kp1'].cpu().numpy()
        M = torch.zeros(n, 3, 3, dtype=torch.float64)
        valid = torch.zeros(n, dtype=torch.int32)
        keep = torch.ones(max(cnt[0], 1), dtype=torch.uint8)
        off = 0
        for b in range(n):
            c = cnt[1 + b]
            if c > 8:
                Mb, mask = self.homography_fn(kp0[off:off + c].astype('int64'), kp1[off:off + c].astype('int64'))
                if Mb is not None:
                    M[b] = torch.as_tensor(Mb, dtype=torch.float64)
                    valid[b] = 1
                    keep[off:off + c] = torch.as_tensor(mask).reshape(-1).to(torch.uint8)
            off += c
        rs['M'], rs['valid'] = M.to(dev), valid.to(dev)
        rs['keep'][:keep.numel()] = keep.to(dev)
        rs['M_f32'] = M.float().to(dev)                                   # cast of geo_module.py:58
        safe = torch.where(valid.bool()[:, None, None], M, torch.eye(3, dtype=torch.float64))
        rs['Minv_f32'] = torch.inverse(safe).float().to(dev)              # inverse in fp64, then cast (:67)

    def geometry(self, batch, n, hw0c, hw1c, dev):
        """RANSAC -> inlier token lists -> window cell tables, all on the device (apply_RANSAC, :23-94)."""
        raw = batch.get('_coarse_dev')
        if raw is None:                            # plain tensors: rebuild the device-side form
            mb = batch['m_bids']
            counts = torch.cat([torch.tensor([mb.numel()], device=dev), torch.bincount(mb, minlength=n)]).to(torch.int32)
            raw = {'mkpts0_c': batch['mkpts0_c'].contiguous(), 'mkpts1_c': batch['mkpts1_c'].contiguous(), 'counts': counts}
        H0, W0 = batch['image0'].shape[2:]
        H1, W1 = batch['image1'].shape[2:]
        scale = int(batch['hw0_i'][0]) // int(batch['hw0_c'][0])
        s0, s1 = batch.get('scale0'), batch.get('scale1')
        rs = ops.ransac_homography(raw['mkpts0_c'], raw['mkpts1_c'], raw['counts'], n, scale, s0, s1, self.ransac_thr,
                                   self.ransac_iters, self.ransac_seed, lm_iters=self.ransac_lm_iters)
        if self.homography_fn is not None:
            self._host_homographies(rs, raw['counts'], n, dev)
        L, S = hw0c[0] * hw0c[1], hw1c[0] * hw1c[1]
        geo = ops.inlier_index(rs['kp0'], rs['kp1'], rs['keep'], raw['counts'], n, L, S, hw0c[1], hw1c[1], scale)
        # windows in image1 for the cells of image0 (M), and in image0 for the cells of image1 (M^-1)
        geo['win1'] = ops.window_geometry(rs['M_f32'], rs['valid'], hw0c, (H1, W1), hw1c[1], scale, self.window_size, s1)
        geo['win0'] = ops.window_geometry(rs['Minv_f32'], rs['valid'], hw1c, (H0, W0), hw0c[1], scale, self.window_size, s0)
        geo['valid'] = rs['valid']
        geo['hw0'], geo['hw1'] = tuple(hw0c), tuple(hw1c)
        geo['nidx_t'] = geo['nidx'].t().contiguous()          # [2, N]: per-side key counts, also the 'skip' predicates
        if L == S:
            geo['idx_both'] = torch.cat([geo['idx0'], geo['idx1']], 0)
            geo['nidx_both'] = geo['nidx_t'].view(-1)
        geo['ransac'] = rs
        return geo

    def forward(self, cnn_desc0, cnn_desc1, batch, desc_map0=None, desc_map1=None, dtype=None, desc_both=None):
        """cnn_desc0/1: raw backbone coarse maps [N,C,h,w]; position encoding is added here
        (geo_module.py:28-29) unless the caller passes the already encoded [N,L,C] maps (desc_both: the [2N,L,C] buffer
        whose halves they are, if there is one)."""
        n = cnn_desc0.shape[0]
        hw0c, hw1c = tuple(cnn_desc0.shape[2:]), tuple(cnn_desc1.shape[2:])
        if desc_map0 is None:
            desc_map0 = self.pos_encoding(cnn_desc0, dtype)
            desc_map1 = self.pos_encoding(cnn_desc1, dtype)
        geo = self.geometry(batch, n, hw0c, hw1c, cnn_desc0.device)
        batch['_geo_dev'] = geo
        return self.des_transformer(desc_map0, desc_map1, geo, both=desc_both)


# ---------------------------------------------------------------------------------------------
# fine level  (reference: loftr_module/fine_preprocess.py, model/fine_matching2.py)
# ---------------------------------------------------------------------------------------------
class FinePreprocess(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.cat_c_feat = config['fine_concat_coarse_feat']
        self.W = config['fine_window_size']
        d_c, d_f = config['coarse']['d_model'], config['fine']['d_model']
        self.d_model_f = d_f
        if not self.cat_c_feat:
            raise NotImplementedError('GeoFormer always concatenates the coarse context (cvpr_ds_config.py:13)')
        self.down_proj = nn.Linear(d_c, d_f, bias=True)
        self.merge_feat = nn.Linear(2 * d_f, d_f, bias=True)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')
        self._cache = {}

    def invalidate(self):
        self._cache = {}

    def weights(self, dtype):
        w = self._cache.get(dtype)
        if w is None:
            d = self.d_model_f
            mw = self.merge_feat.weight.detach()
            w = {'dw': self.down_proj.weight.detach().to(dtype).contiguous(),
                 'db': self.down_proj.bias.detach().float().contiguous(),
                 'mw_win': mw[:, :d].to(dtype).contiguous(), 'mw_ctx': mw[:, d:].to(dtype).contiguous(),
                 'mb': self.merge_feat.bias.detach().float().contiguous()}
            self._cache[dtype] = w
        return w

    def forward(self, feat_f0, feat_f1, feat_c0, feat_c1, data: Dict[str, torch.Tensor]):
        W = self.W
        stride = int(data['hw0_f'][0]) // int(data['hw0_c'][0])
        data.update({'W': torch.tensor(W)})
        dtype = feat_c0.dtype
        M = data['b_ids'].shape[0]
        if M == 0:
            e = torch.empty(0, W ** 2, self.d_model_f, device=feat_f0.device, dtype=dtype)
            return e, e.clone()
        win, ccat = ops.fine_gather(feat_f0, feat_f1, feat_c0, feat_c1, data['b_ids'], data['i_ids'], data['j_ids'],
                                    int(data['hw0_c'][1]), int(data['hw1_c'][1]), stride, W, dtype)
        w = self.weights(dtype)
        # merge_feat(cat([win, down_proj(c)])) = W_win.win + (W_ctx.down_proj(c) + b): the context term is
        # computed once per match and enters the window GEMM as a row-group bias (25 rows per match)
        ctx = ops.linear(ops.linear(ccat, w['dw'], bias=w['db']), w['mw_ctx'], bias=w['mb'])
        out = ops.linear(win, w['mw_win'], rowgroup_bias=ctx, rowgroup_rows=W * W)
        return out[:M], out[M:]


class FineMatching2(nn.Module):
    def __init__(self, temperature=0.1, thr=0.1):
        super().__init__()
        self.temperature, self.thr = temperature, thr

    def forward(self, feat_f0, feat_f1, data: Dict[str, torch.Tensor]):
        M, WW, C = feat_f0.shape
        if M == 0:        # fine_matching2.py:34-42: mconf / m_bids keep their coarse values
            data.update({'fine_matrix': torch.empty(0, WW, WW, device=feat_f0.device),
                         'mkpts0_f': data['mkpts0_c'], 'mkpts1_f': data['mkpts1_c']})
            return
        hi, hc, hf = float(data['hw0_i'][0]), float(data['hw0_c'][0]), float(data['hw0_f'][0])
        out = ops.fine_match(feat_f0, feat_f1, self.temperature, self.thr, data['b_ids'], data['mkpts0_c'],
                             data['mkpts1_c'], hi / hc, hf / hc, hi / hf, data.get('scale0'), data.get('scale1'))
        mf = int(out['count'][0])           # final host sync: number of fine matches
        data.update({'fine_matrix': out['fine_matrix'], 'm_bids': out['m_bids'][:mf], 'mkpts0_f': out['mkpts0_f'][:mf],
                     'mkpts1_f': out['mkpts1_f'][:mf], 'mconf': out['mconf'][:mf]})
