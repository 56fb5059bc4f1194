"""Fine-level linear attention (windows of 25 tokens, 8 heads of 16) at the nominal load: python tools/la_fine_time.py [windows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 37120
q = torch.randn(N, 25, 128, device='cuda').half()
kv = torch.randn(N, 25, 256, device='cuda').half()
f = lambda: ops.linear_attention(q, kv[..., :128], kv[..., 128:], 8)
for _ in range(3):
    f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f'{N} windows: {us:.1f} us  -> {4 * N * 25 * 128 * 2 / us / 1e3:.0f} GB/s (q, k, v in + message out)')
