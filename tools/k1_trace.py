"""Per-workgroup phase timeline of the panel-persistent k1_conf (needs a build with `#define K1_TRACE 1` first in
k1_dual_softmax.hip).  Stamps of each workgroup's SECOND unit: 0 unit start, 1 after the A-fragment loads and
the first prefetch are issued, then per tile 2+4t staged (LDS written, barrier), 3+4t MFMAs done, 4+4t epilogue issued."""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from geoformer_amd import ops, _lib
STATS = len(sys.argv) > 1 and sys.argv[1] == 'stats'      # python tools/k1_trace.py stats  (needs -DK1_TRACE=2): pass A
thr = float(sys.argv[1]) if len(sys.argv) > 1 and not STATS else 0.0
N, L, S, C = 8, 6400, 6400, 256
f0 = (torch.randn(N, L, C, device='cuda') * 1.3).half()
f1 = (f0[:, torch.randperm(S, device='cuda')].float() + 0.4 * torch.randn(N, S, C, device='cuda')).half()
for _ in range(3):
    ops.dual_softmax_match(f0, f1, 0.1, thr, (80, 80), (80, 80), 8.0)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 32, dtype=np.int64)
ctypes.CDLL(_lib.LIB_PATH).gf_debug_k1_trace(buf.ctypes.data_as(ctypes.c_void_p))
if STATS:
    t = buf.reshape(512, 4, 32)[:, 0, :25]
    d = t - t[:, :1]
    m = np.median(d, axis=0)
    print('pass A, second unit of every workgroup, median clocks from unit start (wave 0):')
    for tl in range(4):
        b = 1 + 6 * tl
        print(f'tile {tl}: top {int(m[b])} | LDS write + barrier {int(m[b+1]-m[b])} | prefetch issue + MFMAs {int(m[b+2]-m[b+1])} | maxima / rescale {int(m[b+3]-m[b+2])} | exponentials, sums {int(m[b+4]-m[b+3])} | barrier {int(m[b+5]-m[b+4])} | to next top {int(m[b+6]-m[b+5]) if b + 6 < 25 else -1}')
    sys.exit(0)
t = buf.reshape(512, 4, 32)[:, 0, :22]
d = t - t[:, :1]
print('median stamps from unit start:', np.median(d, axis=0).astype(int).tolist())
print('p90:', np.percentile(d, 90, axis=0).astype(int).tolist())
ph = np.diff(np.median(d, axis=0))
print('per tile: stage/wait', ph[1::4][:5].astype(int).tolist() if False else '', )
m = np.median(d, axis=0)
for tl in range(5):
    print(f'tile {tl}: barrier+stage {int(m[2+4*tl]-m[1+4*tl] if tl else m[2]-m[1])}  mfma {int(m[3+4*tl]-m[2+4*tl])}  epilogue {int(m[4+4*tl]-m[3+4*tl])}')
