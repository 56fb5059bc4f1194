"""How close the fused encoder layer is to the storage oracle (oracle/geoformer_oracle.py:encoder_layer_fused): share of fp16
outputs that differ at all, that differ by more than one ulp, and the mean difference - to compare two builds
(GF_LIB_PATH=<other .so> python tools/k9_parity.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch
import geoformer_oracle as O
from geoformer_amd.model.modules import LoFTREncoderLayer
PFX = 'loftr_coarse.layers.2.'
W = O.make_weights()
m = LoFTREncoderLayer(256, 8, 'linear', 'relu')
m.load_state_dict({k[len(PFX):]: v for k, v in W.items() if k.startswith(PFX)})
m = m.cuda()
st = torch.float16
g = torch.Generator().manual_seed(7)
N, L = 2, 1280
x = O.rt(torch.randn(N, L, 256, generator=g) * 0.7, st)
src = O.rt(torch.randn(N, L, 256, generator=g) * 0.7, st)
ref = O.encoder_layer_fused(W, PFX, x, src, 8, st, None, None).float()
with torch.no_grad():
    got = m(x.cuda().to(st), src.cuda().to(st), None, None).float().cpu()
d = (got - ref).abs()
ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14)).log2().floor().exp2() * 2.0 ** -10
print(f'differ: {float((d > 0).float().mean()):.4%}  > 1 ulp: {float((d > 1.01 * ulp).float().mean()):.4%}  mean |diff| {float(d.mean()):.3e}  max {float(d.max()):.3e}')
