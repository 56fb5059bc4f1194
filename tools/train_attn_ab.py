"""The training step of BASELINE configs[2] / [3] (bench.py: train_measurements) with the Geo 'self' layers on K4's own training kernels
against the library's fused attention, same box:   python tools/train_attn_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from geoformer_amd import miopen
from geoformer_amd.train import functional as TF

miopen.use_shipped_find_db()
dev = torch.device('cuda:0')
for own in (True, False, True, False):
    TF._OWN_FULL_ATTENTION[0] = own
    r = bench.train_measurements(dev, lambda *a: None, steps=8, warmup=2)
    print('own kernels' if own else 'library    ', {k: round(v['ms_per_step'], 1) for k, v in r.items()}, flush=True)
