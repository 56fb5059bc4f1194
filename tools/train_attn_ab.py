"""The training step of BASELINE configs[2] / [3] (bench.py: train_measurements) with round 6's own training kernels (K4 forward / backward of
the Geo 'self' layers, K10's weight gradient) against the library's, same box:   python tools/train_attn_ab.py"""
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
for attn, wgrad in ((True, True), (False, False), (True, False), (False, True), (True, True), (False, False)):
    TF._OWN_FULL_ATTENTION[0] = attn
    HA._OWN_CONV_WGRAD[0] = wgrad
    r = bench.train_measurements(dev, lambda *a: None, steps=8, warmup=2)
    print(f'attention {"own    " if attn else "library"} | conv wgrad {"own    " if wgrad else "library"}', {k: round(v['ms_per_step'], 1) for k, v in r.items()}, flush=True)
