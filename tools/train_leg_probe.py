"""Why bench.py's MegaDepth-style training leg reads 213-278 ms where the same step alone takes 187-197: the leg alone, then after the
hpatches_b1 leg (graphs, pipelines), per-step times.   python tools/train_leg_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from geoformer_amd import _lib, miopen
miopen.use_shipped_find_db()
dev = torch.device('cuda:0')
L = _lib.lib()


def leg(tag):
    r = bench.train_measurements(dev, lambda *a: None)
    print(tag, {k: round(v['ms_per_step'], 1) for k, v in r.items()}, 'reserved GB', round(torch.cuda.memory_reserved() / 2**30, 1),
          'device allocs', torch.cuda.memory_stats()['num_device_alloc'], flush=True)


leg('alone            ')
leg('alone again      ')
bench.hpatches_b1_measurements(dev, lambda *a: None, L)
leg('after hpatches_b1')
leg('again            ')
