"""K5 (gf_window_cross_attention) at the bench shape with spatially coherent windows: python tools/k5_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import ops
N, H, W = (8 if '--n8' in sys.argv else 16), 80, 80
L = H * W
q = torch.randn(N, L, 256, device='cuda').half()
kv = torch.randn(N, L, 512, device='cuda').half()
ys, xs = torch.meshgrid(torch.arange(H, device='cuda'), torch.arange(W, device='cuda'), indexing='ij')
dy, dx = torch.meshgrid(torch.arange(-2, 3, device='cuda'), torch.arange(-2, 3, device='cuda'), indexing='ij')
wy = (ys.reshape(-1, 1) + 1 + dy.reshape(1, -1)).clamp(0, H - 1)          # a one-cell shift + the 5x5 window
wx = (xs.reshape(-1, 1) + 1 + dx.reshape(1, -1)).clamp(0, W - 1)
win = (wy * W + wx).int()[None].repeat(N, 1, 1).contiguous()
valid = torch.ones(N, dtype=torch.int32, device='cuda')
hw = (H, W) if '--plain' not in sys.argv else None
f = lambda: ops.window_cross_attention(q, kv[..., :256], kv[..., 256:], win, valid, 4, hw, hw)
for _ in range(3):
    f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f()
b.record(); torch.cuda.synchronize()
print(f'{a.elapsed_time(b) / 20 * 1e3:.1f} us per call of {N} images')
