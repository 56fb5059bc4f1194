"""Phase timeline of the fused lateral (gf_conv1x1_upsample_add_nhwc = K3 with EPI_UPADD); needs a build with -DK3_TRACE=1
(`// hipcc-flags: -DK3_TRACE=1` as the first line of k3_linear.hip).  python tools/k3_upadd_trace.py"""
import sys, ctypes, time
import numpy as np, torch
sys.path.insert(0, '.')
from geoformer_amd import ops, _lib
L = _lib.lib()
fn = getattr(ctypes.CDLL(_lib.LIB_PATH), 'gf_debug_k3_trace', None)
for (cin, cout, H, h) in ((128, 224, 320, 160), (224, 256, 160, 80)):
    x = torch.randn(16, cin, H, H, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    lo = torch.randn(16, cout, h, h, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, device='cuda', dtype=torch.float16) * 0.05
    for _ in range(3):
        ops.conv1x1_upsample_add(x, w, lo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.conv1x1_upsample_add(x, w, lo)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    by = 16 * H * H * (cin + cout) * 2 + 16 * h * h * cout * 2
    print(f'{cin}->{cout} at {H}x{H}: {ms * 1e3:.0f} us, {by / ms / 1e6:.0f} GB/s algorithmic')
    from geoformer_amd import fused
    if hasattr(fused, 'lateral_supported') and fused.lateral_supported(cin, cout):
        wf = fused.pack_lateral_frags(w)
        for _ in range(3):
            fused.lateral_upsample_add(x, wf, cout, lo)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fused.lateral_upsample_add(x, wf, cout, lo)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f'   K12 streaming form: {ms * 1e3:.0f} us, {by / ms / 1e6:.0f} GB/s algorithmic')
        f12 = getattr(ctypes.CDLL(_lib.LIB_PATH), 'gf_debug_k12_trace', None)
        if f12 is not None:
            b12 = np.zeros(256 * 8 * 8, dtype=np.int64)
            f12(b12.ctypes.data_as(ctypes.c_void_p))
            t12 = b12.reshape(256, 8, 8)[:, :, :5]
            print('   K12 cycles per wave over its 50 tiles: product / merge (waits for the taps) / requests / stores / drain, median', np.median(t12.reshape(-1, 5), axis=0).astype(int).tolist())
    if fn is not None:
        buf = np.zeros(1024 * 4 * 16, dtype=np.int64)
        fn(buf.ctypes.data_as(ctypes.c_void_p))
        t = buf.reshape(1024, 4, 16)
        d = t[:, 0, :] - t[:, 0, :1]
        print('   median per-phase cycles (100 MHz ticks x ?) from WG start: K steps', np.median(d[:, 1:5], axis=0).astype(int).tolist(),
              'k end', int(np.median(d[:, 10])), 'epilogue: before barrier / after / slab 0 done / slab 1 done', np.median(d[:, 11:15], axis=0).astype(int).tolist())
