"""Times the fused fine-level layer (K11) against the K3 / K2 chain at the nominal-load size:
python tools/fine_layer_time.py [windows] [reps]   (2 x 2320 x 8 = 37120 windows per 'self' call of an 8-pair batch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch
import geoformer_oracle as O
from geoformer_amd import ops
from geoformer_amd.model.modules import LoFTREncoderLayer
Nw = int(sys.argv[1]) if len(sys.argv) > 1 else 37120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pfx = 'loftr_fine.layers.0.'
W = O.make_weights()
layer = LoFTREncoderLayer(128, 8, 'linear', 'relu')
layer.load_state_dict({k[len(pfx):]: v for k, v in W.items() if k.startswith(pfx)})
layer = layer.cuda()
x = (torch.randn(Nw, 25, 128, device='cuda') * 0.8).half()
y = (torch.randn(Nw, 25, 128, device='cuda') * 0.8).half()
w = layer.weights(torch.float16)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def chain(x, s):
    q = ops.linear(x, w['q']); kv = ops.linear(s, w['kv'])
    return layer.finish(x, ops.linear_attention(q, kv[..., :128], kv[..., 128:], 8))


t_self, t_cross = timeit(lambda: layer(x, x)), timeit(lambda: layer(x, y))
t_chain = timeit(lambda: chain(x, x))
fl = Nw * 25 * (8 * 128 * 128 + 8 * 128 * 16 + 8 * 128 * 128 + 4 * 128 * 128)
print(f'{Nw} windows: fused self {t_self:.1f} us ({fl / t_self * 1e-6:.0f} TFLOP/s)  fused cross {t_cross:.1f} us  | K3+K2 chain {t_chain:.1f} us')
