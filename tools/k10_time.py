"""K10 (gf_conv3x3_nhwc) against MIOpen conv + gf_bias_act at the backbone's 3x3 shapes: correctness and time.
python tools/k10_time.py [fp16|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geoformer_amd import miopen as gf_miopen
gf_miopen.use_shipped_find_db()
import torch
import torch.nn.functional as F
from geoformer_amd import ops, fused

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == 'bf16') else torch.float16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


torch.manual_seed(0)
for (N, CI, CO, H, W) in ((2, 128, 128, 37, 70), (16, 128, 128, 320, 320), (16, 224, 224, 320, 320), (16, 224, 128, 320, 320),
                          (16, 224, 224, 160, 160), (16, 256, 256, 80, 80), (16, 256, 224, 160, 160)):
    x = torch.randn(N, CI, H, W, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(CO, CI, 3, 3, device='cuda', dtype=dt) * 0.03)
    wcl = w.contiguous(memory_format=torch.channels_last)
    b = torch.randn(CO, device='cuda')
    z = torch.randn(N, CO, H, W, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
    ws = fused.pack_conv3x3_stream(w)
    ref = torch.relu(F.conv2d(x.float(), w.float(), None, 1, 1) + b[None, :, None, None] + z.float())
    y = fused.conv3x3(x, ws, CO, b, z, ops.ACT_RELU)
    torch.cuda.synchronize()
    err = float((y.float() - ref).abs().max())
    y0 = fused.conv3x3(x, ws, CO)
    err0 = float((y0.float() - F.conv2d(x.float(), w.float(), None, 1, 1)).abs().max())
    t_k7 = timeit(lambda: fused.conv3x3(x, ws, CO, b, z, ops.ACT_RELU))
    t_sep = timeit(lambda: ops.bias_act_(F.conv2d(x, wcl, None, 1, 1), b, z, ops.ACT_RELU))
    t_conv = timeit(lambda: F.conv2d(x, wcl, None, 1, 1))
    fl = 2.0 * N * H * W * CI * CO * 9
    print(f'{N}x{CI}->{CO}x{H}x{W}: err {err:.3g} / plain {err0:.3g} | K10 {t_k7:.3f} ms ({fl / t_k7 / 1e9:.0f} TFLOP/s) | '
          f'MIOpen conv {t_conv:.3f} + bias_act = {t_sep:.3f} ms', flush=True)

# the 196-channel level: plain seven-chunk call (PAD16 where the output is padded too) against the remainder form (GF_CONV_REM8)
for (N, CI, CO, H, W) in ((16, 224, 224, 160, 160), (16, 224, 128, 320, 320)):
    x = torch.randn(N, CI, H, W, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(CO, CI, 3, 3, device='cuda', dtype=dt) * 0.03)
    w[:, 196:] = 0
    pad16 = CO == 224
    if pad16:
        w[196:] = 0
    b = torch.randn(CO, device='cuda')
    z = torch.randn(N, CO, H, W, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
    ws, wr = fused.pack_conv3x3_stream(w), fused.pack_conv3x3_stream(w, rem8=True)
    t_plain = timeit(lambda: fused.conv3x3(x, ws, CO, b, z, ops.ACT_RELU, pad16=pad16))
    t_rem = timeit(lambda: fused.conv3x3(x, wr, CO, b, z, ops.ACT_RELU, pad16=pad16, rem8=True))
    print(f'{N}x196(224)->{CO}x{H}x{W}: seven chunks {t_plain:.3f} ms | six chunks + remainder {t_rem:.3f} ms ({100 * (1 - t_rem / t_plain):.1f} % less)', flush=True)

# stride 2 (GF_CONV_S2) against the library convolution + gf_bias_act
for (N, CI, CO, H) in ((16, 128, 224, 320), (16, 224, 256, 160)):
    x = torch.randn(N, CI, H, H, device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(CO, CI, 3, 3, device='cuda', dtype=dt) * 0.03)
    pad16 = CO == 224
    if pad16:
        w[196:] = 0
    wcl = w.contiguous(memory_format=torch.channels_last)
    b = torch.randn(CO, device='cuda')
    ws = fused.pack_conv3x3_stream(w, s2=True)
    t_k = timeit(lambda: fused.conv3x3(x, ws, CO, b, None, ops.ACT_RELU, pad16=pad16, stride=2))
    t_sep = timeit(lambda: ops.bias_act_(F.conv2d(x, wcl, None, 2, 1), b, None, ops.ACT_RELU))
    fl = 2.0 * N * (H // 2) ** 2 * CI * CO * 9
    print(f'stride 2 {N}x{CI}->{CO}x{H}x{H}: K10 {t_k:.3f} ms ({fl / t_k / 1e9:.0f} TFLOP/s) | MIOpen conv + bias_act {t_sep:.3f} ms', flush=True)
