"""K1 (gf_dual_softmax_match) at the bench shape, per kernel tag via the library's HIP-event profile:
python tools/k1_time.py [pairs=8] [thr=0.0]     (GF_K1_CONF=panel selects the unpipelined pass-B form)"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
L = _lib.lib()
torch.manual_seed(0)
f0 = torch.randn(N, 6400, 256, device='cuda').half()
f1 = (f0[:, torch.randperm(6400, device='cuda')] + 0.3 * torch.randn(N, 6400, 256, device='cuda').half()).contiguous()
for _ in range(3):
    out = ops.dual_softmax_match(f0, f1, 0.1, thr, (80, 80), (80, 80), 8.0)
torch.cuda.synchronize()
L.gf_profile_filter(None)
L.gf_profile_enable(1)
for _ in range(10):
    out = ops.dual_softmax_match(f0, f1, 0.1, thr, (80, 80), (80, 80), 8.0)
torch.cuda.synchronize()
for tag in ('k1_stats', 'k1_conf'):
    ms, cnt, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
    L.gf_profile_collect(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(work))
    t = ms.value / max(cnt.value, 1)
    unit = work.value / max(cnt.value, 1) / (t * 1e-3) / (1e12 if tag == 'k1_stats' else 1e9)
    print(f'{tag}: {t * 1e3:.1f} us per launch of {N} pairs -> {unit:.0f} {"TFLOP/s" if tag == "k1_stats" else "GB/s (algorithmic)"}')
L.gf_profile_enable(0)
