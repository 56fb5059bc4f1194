"""Why the two-pipeline batch-1 throughput of bench.py's hpatches_b1 leg is bimodal (≈ 500 or ≈ 250 pairs/s): rates of repeated regions of the
light-load step, device allocations inside the regions, the interpreter's switch interval.   python tools/b1_pipeline_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda:0')
ml, _ = bench.build_model('bf16', 0.0, 0.0, dev)
j1 = bench.synth_rect_pair((480, 640), (480, 608), 77, dev)
step = lambda i: ml({'image0': j1[0], 'image1': j1[1]})
pipes = None
with torch.no_grad():
    for interval in (0.005, 0.005, 0.0005, 0.00005, 0.005):
        sys.setswitchinterval(interval)
        rates, allocs = [], []
        for _ in range(5):
            a0 = torch.cuda.memory_stats()['num_device_alloc']
            el, pipes = bench.measure_fn(step, 60, 6, 2, dev, pipes)
            rates.append(round(60 / el, 1)); allocs.append(torch.cuda.memory_stats()['num_device_alloc'] - a0)
        print(f'switch interval {interval}: two pipelines {rates} pairs/s, device allocations per region {allocs}', flush=True)
    sys.setswitchinterval(0.005)
    for stag in (0.0016, 0.0016, 0.0008, 0.0032):
        rates = []
        for _ in range(6):
            el, pipes = bench.measure_fn(step, 60, 6, 2, dev, pipes, stagger=stag)
            rates.append(round(60 / el, 1))
        print(f'stagger {stag * 1e3:.1f} ms: two pipelines {rates} pairs/s', flush=True)
    pipes.serial = True
    el, pipes = bench.measure_fn(step, 60, 6, 2, dev, pipes)
    print(f'one stream, no threads: {60 / el:.1f} pairs/s')
pipes.close()
