"""Per-workgroup phase timeline of K3 (needs a build with -DK3_TRACE=1: put `#define K3_TRACE 1` first in\nk3_linear.hip and rebuild).  Prints clock64() offsets of the K steps and the epilogue phases."""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from geoformer_amd import ops, _lib
L = _lib.lib()
M = 102400
dev = 'cuda'
x = torch.randn(M, 256, device=dev, dtype=torch.float16)
w = torch.randn(256, 256, device=dev, dtype=torch.float16) * 0.05
for _ in range(3):
    ops.linear(x, w)
torch.cuda.synchronize()
buf = np.zeros(1024 * 4 * 16, dtype=np.int64)
fn = ctypes.CDLL(_lib.LIB_PATH).gf_debug_k3_trace
fn(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(1024, 4, 16)[:800]
t0 = t[:, :, 0].min()
print('clock range (all WGs):', (t[:, :, :15].max() - t0))
for wg in (0, 1, 100, 255, 256, 511, 512, 600, 799):
    r = t[wg, 0] - t0
    print(wg, 'start', r[0], 'steps', (r[1:5] - r[0]).tolist(), 'kend', r[10] - r[0], 'epi', (r[11:15] - r[0]).tolist())
d = t[:, 0, :] - t[:, 0, :1]
print('median per-phase (from WG start):', np.median(d, axis=0).astype(int).tolist())
print('start spread:', np.percentile(t[:, 0, 0] - t0, [0, 25, 50, 75, 100]).astype(int).tolist())
print('end spread:', np.percentile(t[:, 0, 14] - t0, [0, 25, 50, 75, 100]).astype(int).tolist())
