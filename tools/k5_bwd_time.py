"""K5's backward, gather form (round 6) against the scatter form with fp32 atomics, at the training step's size: 8 images, 6400 queries,
25 window positions, bf16; the inverse index (one torch.sort) timed separately.   python tools/k5_bwd_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geoformer_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


torch.manual_seed(0)
N, h, w = 8, 80, 80
L = S = h * w
q, km, vm, d = (torch.randn(N, L, 256, device='cuda', dtype=torch.bfloat16) for _ in range(4))
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
dy, dx = torch.meshgrid(torch.arange(-2, 3), torch.arange(-2, 3), indexing='ij')
yy, xx = ys.flatten()[:, None] + 1 + dy.flatten()[None], xs.flatten()[:, None] - 1 + dx.flatten()[None]
win = torch.where((yy >= 0) & (yy < h) & (xx >= 0) & (xx < w), yy * w + xx, -1).to(torch.int32)[None].repeat(N, 1, 1).cuda().contiguous()
idx = ops.window_inverse_index(win, S)
print(f'inverse index (torch.sort of {win.numel()} keys): {timeit(lambda: ops.window_inverse_index(win, S)):.0f} us')
print(f'gather form:  {timeit(lambda: ops.window_cross_attention_backward_gather(q, km, vm, d, win, idx)):.0f} us per call')
print(f'scatter form: {timeit(lambda: ops.window_cross_attention_backward(q, km, vm, d, win)):.0f} us per call (+ the casts of its fp32 maps)')
