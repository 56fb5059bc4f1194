"""K10's weight-gradient kernel against the library's (aten::convolution_backward, weights only) at the training backbone's shapes, bf16,
16 images:   python tools/k10_wgrad_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geoformer_amd import fused, miopen
miopen.use_shipped_find_db()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


torch.manual_seed(0)
for (cx, cin, cy, cout, hw) in ((128, 128, 128, 128, 320), (224, 196, 224, 196, 160), (256, 256, 256, 256, 80), (224, 196, 128, 128, 320), (256, 256, 224, 196, 160),
                                (224, 196, 224, 196, 320), (256, 256, 256, 256, 160)):
    x = torch.randn(16, cx, hw, hw, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(16, cy, hw, hw, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(cout, cin, 3, 3, device='cuda', dtype=torch.bfloat16)
    xs, dys = x[:, :cin].contiguous(memory_format=torch.channels_last), dy[:, :cout].contiguous(memory_format=torch.channels_last)
    lib = lambda: torch.ops.aten.convolution_backward(dys, xs, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    own = lambda: fused.conv3x3_wgrad(x, dy, cin, cout)
    a, b = lib().float(), own()
    tl, to = timeit(lib), timeit(own)
    fl = 18.0 * 16 * hw * hw * cin * cout
    print(f'{cin}->{cout} at {hw}x{hw}: library {tl:.3f} ms, own {to:.3f} ms ({fl / to * 1e-9:.0f} TFLOP/s on the real channels), '
          f'rel. difference to the library {float((a - b).norm() / b.norm()):.1e}')
