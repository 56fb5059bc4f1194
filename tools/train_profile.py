"""Where the training step spends its time (torch profiler, top ops by device time).
   python tools/train_profile.py [H W] [--no-fused] [--no-prof] [--bf16] [--hip] [--batch B] [--long] [--mega] [--cl] [--hipconv] [--stacks]
   --stacks: additionally the call sites (python stacks) of the elementwise copy / cast kernels by device time"""
import sys, time, torch
HW = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 and sys.argv[1].isdigit() else (480, 640)
FUSED = '--no-fused' not in sys.argv
B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 2
sys.path.insert(0, '.')
from geoformer_amd import miopen; miopen.use_shipped_find_db()
if '--search' in sys.argv:
    torch.backends.cudnn.benchmark = True          # MIOpen find mode: searches the convolution algorithms of the training shapes once
from geoformer_amd.model.cvpr_ds_config import get_default_cfg
from geoformer_amd.model.full_model import GeoFormer
from geoformer_amd.model.geo_config import get_cfg_model
from geoformer_amd.weights import deterministic_init_
from geoformer_amd.train import TrainStep, synthetic_homography_batch, synthetic_megadepth_batch
if '--mega' in sys.argv:            # BASELINE configs[3]'s data contract: depth + pose supervision, padding masks, per-image scales
    synthetic_homography_batch = synthetic_megadepth_batch
g = get_cfg_model(); g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
model = deterministic_init_(GeoFormer(get_default_cfg(), g)).cuda()
PREC = 'bf16' if '--bf16' in sys.argv else 'fp32'
step = TrainStep(model, batch_size=B, fused_coarse_loss=FUSED, precision=PREC, hip_backward='--hip' in sys.argv, channels_last='--cl' in sys.argv, hip_conv='--hipconv' in sys.argv)
nsteps = 10 if "--long" in sys.argv else 4
batches = [synthetic_homography_batch(B, HW, seed=it, device='cuda') for it in range(nsteps)]      # (made outside the timed steps)
torch.cuda.synchronize()
for it in range(nsteps):
    t = time.perf_counter(); step(batches[it]); torch.cuda.synchronize()
    print('step', it, '%.3f s' % (time.perf_counter() - t), 'fused' if FUSED else 'autograd', PREC, HW, 'batch', B, flush=True)
if '--no-prof' in sys.argv:
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes='--stacks' in sys.argv) as prof:
    step(synthetic_homography_batch(B, HW, seed=9, device='cuda')); torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=50))
dev = sum(getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0)) for e in ka) * 1e-3
ncalls = sum(e.count for e in ka if getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0)) > 0)
print(f'device time (sum over kernels) {dev:.1f} ms in {ncalls} device-side calls; host ops recorded {sum(e.count for e in ka)}')
print(ka.table(sort_by='self_cpu_time_total', row_limit=15, max_name_column_width=50))

if '--stacks' in sys.argv:          # (python stacks are not recorded on this build: the operand shapes tell the call sites apart)
    dt = lambda e: getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0))
    for op in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::fill_', 'aten::cat', 'aten::convolution_backward', 'aten::conv2d',
               'aten::miopen_convolution', 'aten::_convolution'):
        rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key == op and (dt(e) > 0 or op.startswith('aten::conv'))]
        key = (lambda e: e.device_time_total) if op.startswith('aten::conv') else dt
        rows.sort(key=key, reverse=True)
        print(f'== {op}: {sum(key(e) for e in rows) * 1e-3:.2f} ms in {sum(e.count for e in rows)} calls; by operand shapes')
        for e in rows[:16]:
            print(f'   {key(e) * 1e-3:7.3f} ms  x{e.count:4d}  {str(e.input_shapes)[:150]}')
