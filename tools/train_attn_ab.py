"""The training step of BASELINE configs[2] / [3] (bench.py: train_measurements) with round 6's changes switched on / off, same box:
own training kernels (K4 forward / backward of the Geo 'self' layers, K10's weight gradient) and the batched Geo layers.
   python tools/train_attn_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from geoformer_amd import miopen
from geoformer_amd.train import functional as TF
from geoformer_amd.train import hip_autograd as HA

miopen.use_shipped_find_db()
dev = torch.device('cuda:0')
bench.train_measurements(dev, lambda *a: None, steps=3, warmup=2)          # (process warm-up)
for attn, wgrad, batched in ((True, True, True), (True, True, False), (False, False, False), (True, True, True), (True, True, False), (False, False, False)):
    TF._OWN_FULL_ATTENTION[0], HA._OWN_CONV_WGRAD[0], TF._BATCHED_GEO[0] = attn, wgrad, batched
    r = bench.train_measurements(dev, lambda *a: None, steps=8, warmup=2)
    print(f'attention {"own    " if attn else "library"} | conv wgrad {"own    " if wgrad else "library"} | Geo layers {"batched       " if batched else "image by image"}',
          {k: round(v['ms_per_step'], 1) for k, v in r.items()}, flush=True)
