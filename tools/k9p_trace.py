"""Phase timeline of the wave-pair encoder kernel (needs a -DK9P_TRACE=1 build: tools/variant.sh k9p_trace k9_encoder_fused.hip -DK9P_TRACE=1,
run with GF_LIB_PATH=tools/ab/k9p_trace.so).  Prints median s_memtime offsets (shader cycles) of the phase boundaries."""
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
fn = ctypes.CDLL(_lib.LIB_PATH).gf_debug_k9p_trace
NAMES = ['start', 'prologue', 'q', 'attention+exchange', 'merge', 'stage 0 (W1x0+LN1, W1m0, act)', 'st0: W1x0+LN1', 'st0: W1m0+act', 'stages 1-3', '-',
         'last W2', 'LN2 stats', 'stored']
ORDER = [0, 1, 2, 3, 4, 6, 7, 5, 8, 10, 11, 12]


def grab():
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8 * 24, dtype=np.int64)
    fn(buf.ctypes.data_as(ctypes.c_void_p))
    return buf.reshape(1024, 8, 24)[:min(1024, N * 50)]


state = fused.encoder_kv_state(x, w['stream_kv'])
for _ in range(3):
    fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L)
t = grab()
d = t[:, :, :18] - t[:, :1, :1]                      # offsets from wave 0's start
med = np.median(d.reshape(-1, 18), axis=0)
print('phase boundaries, median over workgroups and waves (cycles from the workgroup start), and the phase lengths:')
prev = 0
for i in ORDER:
    print(f'  {i:2d} {NAMES[i]:34s} {int(med[i]):7d}   (+{int(med[i] - prev)})')
    prev = med[i]
print('prologue detail: requests issued', int(med[13]), '| parameters parked', int(med[14]), '| tile + block 0 landed', int(med[15]), '| barrier', int(med[1]))
print('finish detail: LN2 stats', int(med[11]), '| rows in place', int(med[17]), '| stores issued', int(med[12]))
print('per wave (median over workgroups) end of q / merge / slices / stored:')
for wv in range(8):
    m = np.median(d[:, wv], axis=0)
    print(f'  wave {wv}: {int(m[2])} {int(m[4])} {int(m[10])} {int(m[12])}')
starts = np.sort(t[:, 0, 0] - t[:, 0, 0].min())
print('WG start times (sorted) at 0/255/256/511/512/767/768:', [int(starts[min(i, len(starts) - 1)]) for i in (0, 255, 256, 511, 512, 767, 768)])
print('wave 0 total p10/p50/p90:', np.percentile(d[:, 0, 12], [10, 50, 90]).astype(int).tolist())
