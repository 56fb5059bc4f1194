"""K4 (gf_self_attention_gathered) at the nominal load: python tools/k4_time.py [keys=1195]   (GF_K4_QB=1|2|4)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1195
N, L = 16, 6400
q = torch.randn(N, L, 256, device='cuda').half()
kv = torch.randn(N, L, 512, device='cuda').half()
idx = torch.stack([torch.randperm(L, device='cuda')[:L].sort()[0] for _ in range(N)]).int()
nk = torch.full((N,), K, device='cuda', dtype=torch.int32)
f = lambda: ops.self_attention_gathered(q, kv[..., :256], kv[..., 256:], idx, nk)
for _ in range(3):
    f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f'K = {K}: {us:.1f} us per call of {N} images  -> {4.0 * N * L * K * 256 / us / 1e6:.0f} TFLOP/s')
