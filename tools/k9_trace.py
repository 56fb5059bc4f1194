"""Phase timeline of the fused encoder kernels (needs a -DK9_TRACE=1 build: HIPCC_EXTRA='-DK9_TRACE=1' is honoured by
geoformer_amd/build.py).  Prints the median s_memtime offsets (100 MHz ticks x clock ratio: raw counter units) of the
phase boundaries per workgroup, for enc_layer and enc_kv_state at 16 images of 80x80 tokens."""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import geoformer_oracle as O
from geoformer_amd import fused, _lib
from geoformer_amd.model.modules import LoFTREncoderLayer
N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 6400
pfx = 'loftr_coarse.layers.0.'
W = O.make_weights()
layer = LoFTREncoderLayer(256, 8, 'linear', 'relu')
layer.load_state_dict({k[len(pfx):]: v for k, v in W.items() if k.startswith(pfx)})
layer = layer.cuda()
x = (torch.randn(N, L, 256, device='cuda') * 0.7).half()
w = layer.weights(torch.float16)
fn = ctypes.CDLL(_lib.LIB_PATH).gf_debug_k9_trace


def grab():
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 4 * 16, dtype=np.int64)
    fn(buf.ctypes.data_as(ctypes.c_void_p))
    return buf.reshape(4096, 4, 16)[:N * 50]


state = fused.encoder_kv_state(x, w['stream_kv'])
for _ in range(2):
    fused.encoder_kv_state(x, w['stream_kv'])
t = grab()
d = t[:, 0, :8] - t[:, 0, :1]
print('enc_kv_state  phases: 0 start | 1 tile+block0 in LDS | 2 k projected | 3 phi(k) packed | 4 v projected | 5 state | 6 reduced | 7 stored')
print('  median offsets:', np.median(d, axis=0).astype(int).tolist())
print('  wave0 total p10/p50/p90:', np.percentile(d[:, 7], [10, 50, 90]).astype(int).tolist(), ' kernel span', int(t[:, :, 7].max() - t[:, :, 0].min()))
for _ in range(2):
    fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L)
t = grab()
d = t[:, 0, :10] - t[:, 0, :1]
print('enc_layer  phases: 0 start | 1 prologue done | 2 q done | 3 attention done | 4 merge done | 5 LN1+pack | 6 slice 0 done | 7 slices done | 8 LN2 | 9 stored')
print('  median offsets:', np.median(d, axis=0).astype(int).tolist())
print('  slice 0 detail (x half done | m half done | activation packed), offsets from LN1+pack:', (np.median(t[:, 0, 10:13] - t[:, 0, 5:6], axis=0)).astype(int).tolist())
print('  slice 1 detail (same three), offsets from slice 0 done:', (np.median(t[:, 0, 13:16] - t[:, 0, 6:7], axis=0)).astype(int).tolist())
print('  wave0 total p10/p50/p90:', np.percentile(d[:, 9], [10, 50, 90]).astype(int).tolist(), ' kernel span', int(t[:, :, 9].max() - t[:, :, 0].min()))
starts = np.sort(t[:, 0, 0] - t[:, :, 0].min())
print('  WG start times (sorted) at 0/256/512/768/799:', [int(starts[min(i, len(starts) - 1)]) for i in (0, 256, 512, 768, 799)])
