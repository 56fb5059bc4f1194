"""Phase timeline of gf_conv3x3_nhwc (needs HIPCC_EXTRA='-DK10_TRACE=1' python -m geoformer_amd.build): median s_memtime
offsets (10 ns ticks) per tile: 0 tile start | 1 first block + patch landed | 2.. chunk c done | 10 epilogue start | 11 tile done.
python tools/k10_trace.py [cin cout H]"""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geoformer_amd import fused, _lib, ops
CI, CO, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 128, 320)
N = 16
x = torch.randn(N, CI, H, H, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
w = torch.randn(CO, CI, 3, 3, device='cuda', dtype=torch.float16) * 0.03
z = torch.randn(N, CO, H, H, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
b = torch.randn(CO, device='cuda')
ws = fused.pack_conv3x3_stream(w)
fn = ctypes.CDLL(_lib.LIB_PATH).gf_debug_k10_trace
fn2 = ctypes.CDLL(_lib.LIB_PATH).gf_debug_k10_trace2
for res in (z, None):
    for _ in range(3):
        fused.conv3x3(x, ws, CO, b, res, ops.ACT_RELU)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8 * 16, dtype=np.int64)
    fn(buf.ctypes.data_as(ctypes.c_void_p))
    T4 = buf.reshape(256, 8, 8, 16)
    t = T4[:, :, 0]
    nch = CI // 32
    cols = [0, 1] + list(range(2, 2 + nch)) + [10, 11]
    print(f'{CI}->{CO} {H}x{H} residual={res is not None}: slots {cols}')
    for w in range(8):
        d = T4[:, 4, w, [12, 13, 14, 15, 2, 10, 11]] - T4[:, 4, 0, :1]
        print(f'  tile 4 wave {w}: turn0 reached/vm-waited/barrier passed/dma issued, chunk0 done, epilogue start, end:', np.median(d, axis=0).astype(int).tolist())
    for it in (0, 1, 4, 7):
        d = t[:, it, cols] - t[:, it, :1]
        print(f'  tile {it}: median offsets', np.median(d, axis=0).astype(int).tolist(), ' start-after-kernel-begin', int(np.median(t[:, it, 0]) - t[:, 0, 0].min()))
    b2 = np.zeros(256 * 8 * 32, dtype=np.int64)
    fn2(b2.ctypes.data_as(ctypes.c_void_p))
    S = b2.reshape(256, 8, 32)
    for w in (0, 4):
        d = S[:, w, :] - S[:, 0, :1]
        m = np.median(d, axis=0).astype(int)
        print(f'  tile 4 chunk 1 wave {w}: sub-step starts', m[:19].tolist())
        print(f'      sub-step lengths', np.diff(m[:19]).tolist(), '| turns: wait done', m[20:23].tolist(), 'barrier passed', m[24:27].tolist())
