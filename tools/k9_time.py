"""Times the fused encoder kernels at the bench shapes: python tools/k9_time.py [images] [reps]
(16 images of 80x80 tokens = one coarse 'self' layer call of an 8-pair batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import torch
import geoformer_oracle as O
from geoformer_amd import fused, ops
from geoformer_amd.model.modules import LoFTREncoderLayer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = 6400
pfx = 'loftr_coarse.layers.0.'
W = O.make_weights()
layer = LoFTREncoderLayer(256, 8, 'linear', 'relu')
layer.load_state_dict({k[len(pfx):]: v for k, v in W.items() if k.startswith(pfx)})
layer = layer.cuda()
x = (torch.randn(N, L, 256, device='cuda') * 0.7).half()
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


state = fused.encoder_kv_state(x, w['stream_kv'])
t_kv = timeit(lambda: fused.encoder_kv_state(x, w['stream_kv']))
t_layer = timeit(lambda: fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L))
t_fin = timeit(lambda: fused.encoder_layer(x, w['stream_finish'], w['ln'], 1e-5, 1e-5, 0, msg=x))
tok = N * L
fl_layer = tok * (2 * 256 * 256 * 2 + 2 * 256 * 33 + 8 * 256 * 256 + 4 * 256 * 256)
fl_kv = tok * (4 * 256 * 256 + 2 * 256 * 32)


def old():
    q = ops.linear(x, w['q']); kv = ops.linear(x, w['kv'])
    m = ops.linear_attention(q, kv[..., :256], kv[..., 256:], 8)
    m = ops.linear(m, w['merge'], epilogue=ops.EPI_LN, ln=w['n1'], eps=1e-5)
    h = ops.linear(x, w['w1'], a2=m, epilogue=ops.EPI_RELU)
    return ops.linear(h, w['w2'], epilogue=ops.EPI_LN_RES, ln=w['n2'], eps=1e-5, residual=x)


t_tail = timeit(lambda: fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, tail_stream=w['stream_kv'], tail_first=0))
t_tail_half = timeit(lambda: fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, tail_stream=w['stream_kv'], tail_first=N // 2))
print(f'{N} images: layer + state tail (all images) {t_tail:.1f} us [separate: {t_layer + t_kv:.1f}], tail on the second half {t_tail_half:.1f} us '
      f'[separate: {t_layer:.1f} + {N // 2}-image state pass]')
t_old = timeit(old)
print(f'{N} images: enc_kv_state {t_kv:.1f} us ({fl_kv / t_kv * 1e-6:.0f} TFLOP/s)  enc_layer {t_layer:.1f} us ({fl_layer / t_layer * 1e-6:.0f} TFLOP/s)  '
      f'finish-only {t_fin:.1f} us  | K3+K2 chain {t_old:.1f} us')
