"""ATen operators inside one steady-state inference step of the bench workload (8 planted pairs, fp16), by operand shapes - what is left of
torch in the step besides the HIP library's launches.   python tools/infer_profile_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model, W = bench.build_model('fp16', 0.2, 0.1, dev)
i0, i1 = bench.synth_pairs(8, seed=0, size=640, device=dev, kind='shift')
pl = bench.planted_maps(8, 60000, 80, dev, model.compute_dtype)


def step():
    return bench.planted_step(model, i0, i1, pl, True)


with torch.no_grad():
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
dt = lambda e: getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0))
rows = [e for e in prof.key_averages(group_by_input_shape=True) if dt(e) > 0 and e.key.startswith('aten::')]
rows.sort(key=dt, reverse=True)
print(f'aten operators with device time in one step: {sum(dt(e) for e in rows) * 1e-3:.3f} ms in {sum(e.count for e in rows)} calls')
for e in rows[:40]:
    print(f'  {dt(e) * 1e-3:7.3f} ms  x{e.count:4d}  {e.key:28s} {str(e.input_shapes)[:120]}')
