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
planted = '--homography' not in sys.argv              # default: the bench's nominal-load step (planted feature maps)
model, _ = bench.build_model('fp16', 0.2 if planted else 0.0, 0.1 if planted else 0.0, dev)
i0, i1 = bench.synth_pairs(8, 0, 640, dev, kind='shift' if planted else 'homography')
feats = bench.planted_features(8, 60000, 80, device=dev, dtype=torch.float16) if planted else None


def step():
    if feats is None:
        return model({'image0': i0, 'image1': i1})
    model._backbone(torch.cat([i0, i1], dim=0))
    return model.forward_features({'image0': i0, 'image1': i1}, *feats)


with torch.no_grad():
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
rows = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.name.startswith('aten::') and not any(c.name.startswith('aten::') for c in ev.cpu_children) and \
            any('aunch' in c.name for c in ev.cpu_children):                      # leaf ATen ops that launch a kernel
        site = next((f for f in ev.stack if 'geoformer_amd' in f or 'bench.py' in f), ev.stack[0] if ev.stack else '?')
        rows[(site.split('/')[-1][:80], ev.name)] += 1
        dur[(site.split('/')[-1][:80], ev.name)] += ev.device_time_total if hasattr(ev, 'device_time_total') else ev.cuda_time_total
for (site, name), n in rows.most_common(60):
    print(f'{n:5d}  {dur[(site, name)]:9.1f} us  {name:30s} {site}')
print('total kernel-launching aten ops', sum(rows.values()))
# device kernels of the profiled step by name (everything, ours included)
kc = collections.Counter()
kd = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kc[ev.name[:80]] += 1
        kd[ev.name[:80]] += ev.device_time if hasattr(ev, 'device_time') else ev.cuda_time
print('device kernels of the step:', sum(kc.values()))
for name, n in kc.most_common(25):
    print(f'{n:5d}  {kd[name]:9.1f} us  {name}')
par = collections.Counter()
for ev in prof.events():
    if ev.name.startswith('aten::') and any('gather' in k.name or 'index' in k.name for k in getattr(ev, 'kernels', [])):
        site = next((f for f in ev.stack if 'geoformer_amd' in f or 'bench.py' in f), ev.stack[0] if ev.stack else '?')
        par[(ev.name, site.split('/')[-1][:90])] += 1
for (name, site), n in par.most_common(20):
    print(f'{n:5d}  {name:28s} {site}')
