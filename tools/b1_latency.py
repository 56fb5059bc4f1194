"""Per-pair latency of an HPatches-shaped pair (480x640 / 480x608, batch 1, bf16, one stream, a host synchronisation per pair) with the two
backbone calls concurrent (round 6) and one after the other.  (The two images' 'self' layer calls on two streams as well: 2.82 -> 2.80 ms,
not kept - at batch 1 the pair is bound by the host's enqueue time by then.)   python tools/b1_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from geoformer_amd.model import full_model as FM
dev = torch.device('cuda:0')
ml, _ = bench.build_model('bf16', 0.0, 0.0, dev)
j = bench.synth_rect_pair((480, 640), (480, 608), 77, dev)
for conc in (True, False, True, False):
    FM._CONCURRENT_BACKBONES[0] = conc
    with torch.no_grad():
        lat = []
        for i in range(70):
            torch.cuda.synchronize(); t = time.perf_counter()
            ml({'image0': j[0], 'image1': j[1]})
            torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
        lat = sorted(lat[10:])
    print(f'backbones {"concurrent" if conc else "sequential"}: latency p50 {lat[30] * 1e3:.3f} ms, p90 {lat[54] * 1e3:.3f} ms')
