import sys, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from geoformer_amd import ops, _lib
N, L, S, C = 8, 6400, 6400, 256
f0 = (torch.randn(N, L, C, device='cuda') * 1.3).half()
f1 = (f0[:, torch.randperm(S, device='cuda')].float() + 0.4 * torch.randn(N, S, C, device='cuda')).half()
for _ in range(3):
    ops.dual_softmax_match(f0, f1, 0.1, 0.2, (80, 80), (80, 80), 8.0)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 32, dtype=np.int64)
ctypes.CDLL(_lib.LIB_PATH).gf_debug_k1_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(512, 4, 32)[:256, 0, :32]
d = t - t[:, :1]
m = np.median(d, axis=0)
print('stamps', m.astype(int).tolist())
for tl in range(1, 5):
    b = 1 + 6 * tl
    print(f'tile {tl}: step(tl,0) {int(m[b] - m[b - 2])} | wait+barrier {int(m[b + 1] - m[b])} | dma issue + segment A + rescale {int(m[b + 3] - m[b + 1])} | segment B {int(m[b + 2] - m[b + 3])} | tail {int(m[b + 4] - m[b + 2])}')
