"""Phase timeline of K4 attn_self (needs a build with -DK4_TRACE=1, e.g. HIPCC_EXTRA=-DK4_TRACE=1).  Stamps of tiles 4..7 of every
workgroup (wave 0): 0 tile top, 1 own LDS-DMA pieces landed, 2 barrier passed, 3 S of query block 0 issued, 4 its softmax done,
5 its P.V issued, 6 end of the tile (query block 1 done)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import ops, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1195
N, L = 16, 6400
q = torch.randn(N, L, 256, device='cuda').half()
kv = torch.randn(N, L, 512, device='cuda').half()
idx = torch.stack([torch.randperm(L, device='cuda')[:L].sort()[0] for _ in range(N)]).int()
nk = torch.full((N,), K, device='cuda', dtype=torch.int32)
for _ in range(3):
    ops.self_attention_gathered(q, kv[..., :256], kv[..., 256:], idx, nk)
torch.cuda.synchronize()
buf = np.zeros(2048 * 32, dtype=np.int64)
ctypes.CDLL(_lib.LIB_PATH).gf_debug_k4_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(2048, 4, 8)[:1600, :, :7]
names = ['wait own DMA', 'barrier', 'request next + S(qb0)', 'softmax(qb0)', 'P.V(qb0) issue', 'query block 1']
if os.environ.get('GF_K4_FORM') in ('pipe2', 'head'):      # attn_self_pipe2's stamps
    names = ['wait own DMA', 'barrier', 'request + tokens + reference test', 'fragment reads, S(t+1) | exp 0..7', 'P.V | exp 8..15', 'max of S(t+1)']
for tl in range(4):
    d = np.diff(t[:, tl, :], axis=1)
    print(f'tile {4 + tl}: ' + ' | '.join(f'{n} {int(np.median(d[:, i]))}' for i, n in enumerate(names)) + f' | tile total {int(np.median(t[:, tl, 6] - t[:, tl, 0]))}', end='')
    if tl < 3:
        print(f' | to next top {int(np.median(t[:, tl + 1, 0] - t[:, tl, 6]))}')
    else:
        print()
