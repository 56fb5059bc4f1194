"""Two-pipeline throughput of the light-load HPatches-shaped pair with the pair's two backbone calls on two streams (GeoFormer.concurrent_backbones)
and on one, alternating regions on one box.   python tools/b1_concurrency_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device('cuda:0')
ml, _ = bench.build_model('bf16', 0.0, 0.0, dev)
j1 = bench.synth_rect_pair((480, 640), (480, 608), 77, dev)
step = lambda i: ml({'image0': j1[0], 'image1': j1[1]})
pipes = None
res = {True: [], False: []}
with torch.no_grad():
    for rep in range(8):
        for on in (True, False):
            ml.concurrent_backbones = on
            el, pipes = bench.measure_fn(step, 60, 6, 2, dev, pipes, stagger=0.0015)
            res[on].append(round(60 / el, 1))
pipes.close()
for on in (True, False):
    r = sorted(res[on])
    print(f'backbones on {"two streams" if on else "one stream "}: two-pipeline pairs/s {res[on]}  median {r[len(r) // 2]}')
