"""Latency of GeoFormer.forward on HPatches-like shapes (one pair at a time, unequal sizes), fp16 mode:
   python tools/matcher_latency.py [--search]     (--search: let MIOpen search per shape first)"""
import sys, time, torch
sys.path.insert(0, '.')
from geoformer_amd import miopen; miopen.use_shipped_find_db()
from geoformer_amd.model.cvpr_ds_config import get_default_cfg
from geoformer_amd.model.full_model import GeoFormer
from geoformer_amd.model.geo_config import get_cfg_model
from geoformer_amd.weights import deterministic_init_
torch.backends.cudnn.benchmark = '--search' in sys.argv
g = get_cfg_model(); g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp16')
model = deterministic_init_(GeoFormer(get_default_cfg(), g)).cuda().eval()
for hw0, hw1 in (((480, 640), (480, 608)), ((480, 640), (480, 640)), ((480, 720), (640, 480))):
    a, b = torch.rand(1, 1, *hw0, device='cuda'), torch.rand(1, 1, *hw1, device='cuda')
    with torch.no_grad():
        for _ in range(3):
            model({'image0': a, 'image1': b})
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            out = model({'image0': a, 'image1': b})
        torch.cuda.synchronize()
    print(hw0, hw1, '%.2f ms/pair' % ((time.perf_counter() - t) * 100), 'matches', len(out['mkpts0_f']), flush=True)
