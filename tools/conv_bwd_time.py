"""Where the training step's convolution backward goes (MIOpen, bf16 NCHW, the backbone's 3x3 shapes at batch 8 = 16 images of 640x640):
forward, backward-data, backward-weights, and backward-data computed as a FORWARD convolution with the flipped / transposed weights.
   python tools/conv_bwd_time.py [images=16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geoformer_amd import miopen as gf_miopen
gf_miopen.use_shipped_find_db()
import torch
import torch.nn.functional as F
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
dt = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


SHAPES = ((128, 128, 320, 1), (128, 196, 320, 2), (196, 196, 160, 1), (196, 256, 160, 2), (256, 256, 80, 1), (196, 128, 320, 1))
if '--padded' in sys.argv:                      # the same layers with the 196-channel sides padded to 224 (what the inference backbone does)
    SHAPES = ((128, 224, 320, 2), (224, 224, 160, 1), (224, 256, 160, 2), (224, 128, 320, 1))
for (ci, co, h, s) in SHAPES:
    x = torch.randn(N, ci, h, h, device='cuda', dtype=dt)
    w = torch.randn(co, ci, 3, 3, device='cuda', dtype=dt) * 0.03
    y = F.conv2d(x, w, None, s, 1)
    gy = torch.randn_like(y)
    t_f = timeit(lambda: F.conv2d(x, w, None, s, 1))
    t_d = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    line = f'{N}x{ci}->{co} {h}x{h} stride {s}: forward {t_f:.3f} ms | backward-data {t_d:.3f} | backward-weights {t_w:.3f}'
    if s == 1:
        wt = w.flip(2, 3).transpose(0, 1).contiguous()
        gx_ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        gx = F.conv2d(gy, wt, None, 1, 1)
        err = float((gx.float() - gx_ref.float()).abs().max() / gx_ref.float().abs().max())
        t_df = timeit(lambda: F.conv2d(gy, wt, None, 1, 1))
        line += f' | backward-data as a forward convolution {t_df:.3f} (rel. diff {err:.1e})'
    print(line, flush=True)
