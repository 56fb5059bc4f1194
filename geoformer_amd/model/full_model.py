"""GeoFormer - drop-in for the reference's model/full_model.py:18-129.

Same constructor `GeoFormer(loftr_config, geoformer_cfg=default_cfg)`, same `forward(data) -> data`
contract (keys written: bs, hw0_i, hw1_i, hw0_c, hw1_c, hw0_f, hw1_f, conf_matrix, dect_conf_matrix,
b_ids, i_ids, j_ids, m_bids, mkpts0_c, mkpts1_c, mconf, W, fine_matrix, mkpts0_f, mkpts1_f), same
state-dict keys (253 tensors) and the same `matcher.` prefix stripping in load_state_dict.

What differs is where the arithmetic runs: the backbone stays on PyTorch-ROCm/MIOpen; everything after
it goes through the HIP kernels of libgeoformer_hip.so.  The forward synchronises with the host twice
per batch (number of coarse matches after the second coarse matching, number of fine matches at the
end); the RANSAC homography of GeoModule runs on the device, so nothing synchronises in between.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .backbone import build_backbone
from .geo_config import default_cfg
from .modules import (CoarseMatching, FineMatching2, FinePreprocess, GeoModule, LocalFeatureTransformer,
                      PositionEncodingSine, materialize_matches)

_PRECISIONS = {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}


_CONCURRENT_BACKBONES = [True]    # unequal-shape pairs: the two backbone calls on two streams (False: one after the other, for A/B)


class GeoFormer(nn.Module):
    def __init__(self, loftr_config, geoformer_cfg=default_cfg):
        super().__init__()
        self.config = loftr_config
        self.geo_cfg = geoformer_cfg
        self.backbone = build_backbone(loftr_config)
        self.loftr_coarse = LocalFeatureTransformer(loftr_config['coarse'])
        self.pos_encoding = PositionEncodingSine(loftr_config['coarse']['d_model'],
                                                 temp_bug_fix=loftr_config['coarse']['temp_bug_fix'])
        loftr_config['match_coarse']['thr'] = geoformer_cfg['coarse_thr']     # same side effect as the reference (:31)
        self.coarse_matching = CoarseMatching(loftr_config['match_coarse'])
        self.coarse_matching.materialize_conf = bool(geoformer_cfg.get('materialize_conf', True))     # False: match-only K1 (opt-in)
        self.fine_preprocess = FinePreprocess(loftr_config)
        self.loftr_fine = LocalFeatureTransformer(loftr_config['fine'])
        self.fine_matching = FineMatching2(geoformer_cfg['fine_temperature'], geoformer_cfg['fine_thr'])
        self.geo_module = GeoModule(geoformer_cfg, loftr_config['coarse']['d_model'])
        self._fused = [None]     # holder list: keeps the folded inference copy out of the module tree / state dict
        self._side_streams = {}  # (device, current stream) -> the stream image 1's backbone runs on (unequal-shape pairs)
        # unequal-shape pairs: image 1's backbone beside image 0's on a second stream.  Shortens ONE pair (2.8 instead of 3.1 ms at the
        # HPatches shapes); a host that already feeds the GPU from several threads / streams gets MORE pairs per second with it off
        # (two pipelines: 560 against 420 pairs/s) - bench.py measures latency with it on and throughput with it off
        self.concurrent_backbones = True
        self.set_precision(geoformer_cfg.get('precision', 'fp32'))

    # -- precision of the matching path: 'fp32' (parity mode), 'fp16' or 'bf16' (16-bit storage, fp32 accumulate)
    def set_precision(self, precision: str, backbone_dtype: Optional[torch.dtype] = None):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self.precision = precision
        self.compute_dtype = _PRECISIONS[precision]
        self.backbone_dtype = backbone_dtype or self.compute_dtype
        self.backbone.to(self.backbone_dtype)
        self._fused[0] = None
        self._drop_graphs()
        return self

    def _drop_graphs(self):
        # captured graphs read the packed weight caches (K9/K10 fragment streams, folded backbone) by address: once
        # those are dropped a replay would read freed or re-used memory - capture again on the next forward
        if getattr(self, '_graphs', None):
            self._graphs = {}

    def _invalidate(self):
        self._fused[0] = None
        self._drop_graphs()
        for m in self.modules():
            if hasattr(m, 'invalidate'):
                m.invalidate()

    def _inference_backbone(self):
        """fp16 inference form of the backbone (BN folded, channels_last, MIOpen convolutions + HIP glue
        kernels; model/backbone.py FusedInferenceBackbone).  The fp32 parity mode keeps the unfused module
        (bit-for-bit the reference's op sequence)."""
        if self._fused[0] is None:
            from .backbone import FusedInferenceBackbone
            self._fused[0] = FusedInferenceBackbone(self.backbone, self.backbone_dtype)
        return self._fused[0]

    def _backbone(self, x):
        x = x.to(self.backbone_dtype)
        if x.is_cuda and self.backbone_dtype != torch.float32 and not self.training:
            return self._inference_backbone()(x)
        return self.backbone(x)

    def _backbone_unequal(self, img0, img1):
        """The two backbone calls of a pair whose images differ in shape (full_model.py:58-59; the HPatches loop, batch 1).  On the
        16-bit inference path they run CONCURRENTLY, image 1 on a side stream: at batch 1 a convolution launches 150-300 workgroups on
        256 CUs for ~40 us, two of them side by side fill the chip (round 6: -0.5 ms of a 3.3 ms pair).  Sequential under graph
        capture, in training and in the fp32 mode (the library's handles are per stream)."""
        if (not img0.is_cuda or self.training or self.backbone_dtype == torch.float32 or torch.cuda.is_current_stream_capturing()
                or not _CONCURRENT_BACKBONES[0] or not self.concurrent_backbones):
            return self._backbone(img0), self._backbone(img1)
        cur = torch.cuda.current_stream(img0.device)
        key = (img0.device.index, cur.cuda_stream)
        side = self._side_streams.get(key)
        if side is None:
            side = self._side_streams[key] = torch.cuda.Stream(device=img0.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out1 = self._backbone(img1)
        out0 = self._backbone(img0)
        cur.wait_stream(side)
        # out1 lives in the side stream's pool and is consumed on the current stream.  No record_stream (it makes the allocator hold the
        # blocks back until an event completes: with two pipelines the light-load leg fell from 560 to 330 pairs/s): the side stream is
        # used by this function only, and every use starts with side.wait_stream(cur) - issued after the consumers of the previous pair
        # were enqueued on cur -, so a block of out1 that returns to the side pool is not touched again before they have run
        return out0, out1

    # -- hipGraph replay of the static part.  Everything before the first host synchronisation has
    # data-independent launches (the device RANSAC and the device-side counts keep it that way), so per (shapes, stream)
    # it is captured once with torch.cuda.graph - MIOpen convolutions and the library's own launches alike - and
    # replayed on inputs copied into the capture's static buffers.  On one GPU the eager path is already GPU-bound
    # (17.15 ms eager vs 17.13 ms replayed per 8 pairs), so this is OFF by default; it takes ~250 launches per batch
    # off the host, which matters when one host feeds many GPUs.
    def enable_graphs(self, on: bool = True):
        self._graphs = {} if on else None
        return self

    def _forward_graphed(self, data):
        img0, img1 = data['image0'], data['image1']
        key = (tuple(img0.shape), tuple(img1.shape), img0.dtype, torch.cuda.current_stream().cuda_stream,
               float(self.coarse_matching.thr), float(self.coarse_matching.temperature), self.precision, self.coarse_matching.materialize_conf)
        entry = self._graphs.get(key)
        if entry is None:
            # warm-up ON the stream the capture will use: MIOpen keeps a handle (and its algorithm picks / workspaces) per
            # stream and this library a workspace per stream - their first use must not fall inside the capture
            cur, side = torch.cuda.current_stream(), torch.cuda.Stream()
            s0, s1 = img0.clone(), img1.clone()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.forward_static({'image0': s0, 'image1': s1})
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            cap = {'image0': s0, 'image1': s1}
            with torch.cuda.graph(g, stream=side):
                self.forward_static(cap)
            entry = self._graphs[key] = (g, s0, s1, cap, side)
        g, s0, s1, cap = entry[:4]
        s0.copy_(img0)
        s1.copy_(img1)
        g.replay()
        data.update({k: v for k, v in cap.items() if k not in ('image0', 'image1')})
        data['_static'] = dict(cap['_static'])                   # the tensors are the capture's static outputs: consumed below
        data['_backbone_feats'] = tuple(cap['_static'][k] for k in ('feat_c0', 'feat_f0', 'feat_c1', 'feat_f1'))
        data = self.forward_dynamic(data)
        for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids', 'mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f', 'mconf'):
            data[k] = data[k].clone()                    # views of the capture's capacity arrays otherwise
        data['_aliases_static_buffers'] = ('conf_matrix', 'dect_conf_matrix', '_feat_dev', '_backbone_feats')
        return data

    def forward(self, data: Dict[str, torch.Tensor]):
        img0, img1 = data['image0'], data['image1']
        if not img0.is_cuda:
            raise RuntimeError('geoformer_amd.GeoFormer runs on an MI355X (CUDA/HIP device) only; there is no CPU path')
        if (getattr(self, '_graphs', None) is not None and not self.training and 'mask0' not in data and 'scale0' not in data
                and self.geo_module.homography_fn is None):
            return self._forward_graphed(data)
        data.update({'bs': torch.tensor(img0.size(0)), 'hw0_i': torch.tensor(img0.shape[2:]),
                     'hw1_i': torch.tensor(img1.shape[2:])})
        n = img0.size(0)
        # 1. backbone (PyTorch-ROCm).  Same-shape pairs go through it as one batch (:55-57)
        if img0.shape[2:] == img1.shape[2:]:
            feats_c, feats_f = self._backbone(torch.cat([img0, img1], dim=0))
            (feat_c0, feat_c1), (feat_f0, feat_f1) = feats_c.split(n), feats_f.split(n)
        else:
            (feat_c0, feat_f0), (feat_c1, feat_f1) = self._backbone_unequal(img0, img1)
        return self.forward_features(data, feat_c0, feat_f0, feat_c1, feat_f1)

    def forward_static(self, data):
        """The part of the forward whose launches do not depend on data: backbone -> ... -> second coarse matching, with
        the match arrays at capacity and their counts on the device.  No host synchronisation: it can be captured into a
        hipGraph (torch.cuda.graph) and replayed; `forward_dynamic` finishes the forward from its result."""
        img0, img1 = data['image0'], data['image1']
        data.update({'bs': torch.tensor(img0.size(0)), 'hw0_i': torch.tensor(img0.shape[2:]), 'hw1_i': torch.tensor(img1.shape[2:])})
        n = img0.size(0)
        if img0.shape[2:] == img1.shape[2:]:
            feats_c, feats_f = self._backbone(torch.cat([img0, img1], dim=0))
            (feat_c0, feat_c1), (feat_f0, feat_f1) = feats_c.split(n), feats_f.split(n)
        else:
            (feat_c0, feat_f0), (feat_c1, feat_f1) = self._backbone_unequal(img0, img1)
        return self.forward_features(data, feat_c0, feat_f0, feat_c1, feat_f1, static_only=True)

    def forward_dynamic(self, data):
        """Host sync #1 (number of coarse matches), fine level, fine matching (host sync #2)."""
        st = data.pop('_static')
        data.update(materialize_matches(st['raw']))
        f0u, f1u = self.fine_preprocess(st['feat_f0'], st['feat_f1'], st['geo0'], st['geo1'], data)
        if f0u.size(0) != 0:
            f0u, f1u = self.loftr_fine(f0u, f1u)
        self.fine_matching(f0u, f1u, data)
        data['_feat_dev'] = {'loftr_f0': st['feat0'], 'loftr_f1': st['feat1'], 'geo_f0': st['geo0'], 'geo_f1': st['geo1'],
                             'fine_f0': f0u, 'fine_f1': f1u}
        return data

    def forward_features(self, data, feat_c0, feat_f0, feat_c1, feat_f1, static_only=False):
        """Everything after the backbone.  Public so that parity tests and benchmarks can drive the
        matching path with given feature maps ([N,256,h,w] coarse, [N,128,4h,4w] fine)."""
        dt = self.compute_dtype
        data.update({'hw0_c': torch.tensor(feat_c0.shape[2:]), 'hw1_c': torch.tensor(feat_c1.shape[2:]),
                     'hw0_f': torch.tensor(feat_f0.shape[2:]), 'hw1_f': torch.tensor(feat_f1.shape[2:])})
        if 'bs' not in data:
            data.update({'bs': torch.tensor(data['image0'].size(0)), 'hw0_i': torch.tensor(data['image0'].shape[2:]),
                         'hw1_i': torch.tensor(data['image1'].shape[2:])})
        # 2. position encoding + flatten, coarse LoFTR transformer
        pe_both = None
        if feat_c0.shape == feat_c1.shape:
            # both position-encoded maps in ONE [2N, L, C] buffer: the two transformers batch the images of a pair and would
            # otherwise concatenate them (52 MB per 8 pairs, twice per step); nothing writes into it afterwards
            n, c, h, w = feat_c0.shape
            pe_both = torch.empty(2 * n, h * w, c, dtype=dt, device=feat_c0.device)
            pe0 = self.pos_encoding(feat_c0, dt, out=pe_both[:n])
            pe1 = self.pos_encoding(feat_c1, dt, out=pe_both[n:])
        else:
            pe0 = self.pos_encoding(feat_c0, dt)
            pe1 = self.pos_encoding(feat_c1, dt)
        mask_c0 = mask_c1 = None
        if 'mask0' in data:
            mask_c0, mask_c1 = data['mask0'].flatten(-2), data['mask1'].flatten(-2)
        feat0, feat1 = self.loftr_coarse(pe0, pe1, mask_c0, mask_c1, both=pe_both)
        # 3. first coarse matching (detector) -> geometry-guided transformer -> second coarse matching
        self.coarse_matching(feat0, feat1, data, mask_c0=mask_c0, mask_c1=mask_c1, lazy=True)
        data['dect_conf_matrix'] = data['conf_matrix']
        same_pe = self.pos_encoding.temp_bug_fix == self.geo_module.pos_encoding.temp_bug_fix
        geo0, geo1 = self.geo_module(feat_c0, feat_c1, data, pe0 if same_pe else None, pe1 if same_pe else None, dt,
                                     desc_both=pe_both if same_pe else None)
        raw = self.coarse_matching(geo0, geo1, data, mask_c0=mask_c0, mask_c1=mask_c1, lazy=True)
        data['_static'] = {'raw': raw, 'feat_c0': feat_c0, 'feat_c1': feat_c1, 'feat_f0': feat_f0, 'feat_f1': feat_f1, 'geo0': geo0, 'geo1': geo1,
                           'feat0': feat0, 'feat1': feat1}
        if static_only:
            return data
        # 4. host sync #1 (M), fine level, 5. fine matching (host sync #2: Mf)
        return self.forward_dynamic(data)

    def load_state_dict(self, state_dict, *args, **kwargs):
        for k in list(state_dict.keys()):
            if k.startswith('matcher.'):
                state_dict[k.replace('matcher.', '', 1)] = state_dict.pop(k)
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self._invalidate()
        self.backbone.to(self.backbone_dtype)
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out
