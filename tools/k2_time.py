"""Times K2 (linear attention) at the coarse shape: 16 images x 6400 tokens x 256 channels, fp16 (whole call:
la16_kv + la_kv_final + la16_apply), with torch events over back-to-back calls."""
import sys, torch
sys.path.insert(0, '.')
from geoformer_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
q = torch.randn(N, 6400, 256, device='cuda').half()
kv = torch.randn(N, 6400, 512, device='cuda').half()
k, v = kv[..., :256], kv[..., 256:]
for _ in range(5):
    ops.linear_attention(q, k, v, 8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.linear_attention(q, k, v, 8)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f'K2 N={N}: {ms * 1e3:.1f} us/call, {4 * N * 6400 * 256 * 2 / ms / 1e6:.0f} GB/s algorithmic')
