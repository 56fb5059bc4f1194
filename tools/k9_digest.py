"""Digest of the fused encoder layer's outputs on seeded inputs (bit-identity checks across builds) + its time:
   python tools/k9_digest.py [images=16]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import torch
import geoformer_oracle as O
from geoformer_amd import fused
from geoformer_amd.model.modules import LoFTREncoderLayer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = 6400
pfx = 'loftr_coarse.layers.0.'
W = O.make_weights()
out = []
for dt in (torch.float16, torch.bfloat16):
    layer = LoFTREncoderLayer(256, 8, 'linear', 'relu')
    layer.load_state_dict({k[len(pfx):]: v for k, v in W.items() if k.startswith(pfx)})
    layer = layer.cuda()
    g = torch.Generator(device='cuda').manual_seed(11)
    x = (torch.randn(N, L, 256, device='cuda', generator=g) * 0.7).to(dt)
    qm = torch.rand(N, L, device='cuda', generator=g) > 0.05
    w = layer.weights(dt)
    state = fused.encoder_kv_state(x, w['stream_kv'])
    y = fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L)
    ym = fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, q_mask=qm)
    yt, st2 = fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, tail_stream=w['stream_kv'], tail_first=0)
    torch.cuda.synchronize()
    h = lambda t: hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:12]
    out.append(f'{str(dt)[6:]}: layer {h(y)} masked {h(ym)} with-tail {h(yt)} tail-state {h(st2)}')
    if dt == torch.float16:
        for _ in range(3):
            fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20):
            fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L)
        b.record(); torch.cuda.synchronize()
        t1 = a.elapsed_time(b) / 20 * 1e3
        a.record()
        for _ in range(20):
            fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, tail_stream=w['stream_kv'], tail_first=0)
        b.record(); torch.cuda.synchronize()
        out.append(f'{N} images: enc_layer {t1:.1f} us, with state tail {a.elapsed_time(b) / 20 * 1e3:.1f} us')
print('\n'.join(out))
