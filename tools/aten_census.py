"""Which ATen ops (not our HIP kernels) one bench-shaped forward launches, by Python call site.
python tools/aten_census.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse_args(['--no-extras', '--no-cpu-baseline'])
from geoformer_amd import miopen
miopen.use_shipped_find_db()
dev = torch.device('cuda:0')
model, _ = bench.build_model('fp16', 0.0, 0.0, dev)
i0, i1 = bench.synth_pairs(8, 0, 640, dev)
with torch.no_grad():
    for _ in range(2):
        model({'image0': i0, 'image1': i1})
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model({'image0': i0, 'image1': i1})
        torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if ev.name.startswith('aten::') and not any(c.name.startswith('aten::') for c in ev.cpu_children) and \
            any('aunch' in c.name for c in ev.cpu_children):                      # leaf ATen ops that launch a kernel
        site = next((f for f in ev.stack if 'geoformer_amd' in f or 'bench.py' in f), ev.stack[0] if ev.stack else '?')
        rows[(site.split('/')[-1][:80], ev.name)] += 1
for (site, name), n in rows.most_common(60):
    print(f'{n:5d}  {name:30s} {site}')
print('total kernel-launching aten ops', sum(rows.values()))
