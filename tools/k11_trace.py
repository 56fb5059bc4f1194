"""Phase timeline of the fused fine-level layer (needs a -DK11_TRACE=1 build: HIPCC_EXTRA='-DK11_TRACE=1' is honoured by
geoformer_amd/build.py).  Median s_memtime offsets of the phase boundaries of one window group, per wave half."""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import geoformer_oracle as O
from geoformer_amd import _lib
from geoformer_amd.model.modules import LoFTREncoderLayer
Nw = int(sys.argv[1]) if len(sys.argv) > 1 else 37120
pfx = 'loftr_fine.layers.0.'
W = O.make_weights()
layer = LoFTREncoderLayer(128, 8, 'linear', 'relu')
layer.load_state_dict({k[len(pfx):]: v for k, v in W.items() if k.startswith(pfx)})
layer = layer.cuda()
x = (torch.randn(Nw, 25, 128, device='cuda') * 0.8).half()
y = (torch.randn(Nw, 25, 128, device='cuda') * 0.8).half()
fn = ctypes.CDLL(_lib.LIB_PATH).gf_debug_k11_trace
for name, src in (('self', x), ('cross', y)):
    for _ in range(3):
        layer(x, src)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 4 * 8, dtype=np.int64)
    fn(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(256, 8, 4, 8)
    print(name, ': slots 0 group start | 1 tile landed | 2 k,v,state done | 3 q + attention done | 4 merge + LN1 done | 5 mlp done | 6 stored')
    for git in (0, 1, 2):
        for half, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
            d = t[:, sl, git, :7] - t[:, sl, git, :1]
            print(f'  group iteration {git} {half}: median offsets', np.median(d.reshape(-1, 7), axis=0).astype(int).tolist())
    per = t[:, 0, 1, 0] - t[:, 0, 0, 0]
    print('  group period (wave 0, iteration 0 -> 1): median', int(np.median(per)))
