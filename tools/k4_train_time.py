"""K4's training kernels (csrc/k4_attention_train.hip) against the library's fused attention (F.scaled_dot_product_attention) at the shapes of
the training step's Geo 'self' layers: one image, 6400 queries, S inlier keys, 4 heads of 64, bf16.   python tools/k4_train_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from geoformer_amd.train import hip_autograd as HA


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    for L, S in ((6400, 600), (6400, 1195), (6400, 2400), (6400, 4800), (4800, 1000)):
        q, k, v = (torch.randn(1, n, 256, device='cuda', dtype=dt, requires_grad=True) for n in (L, S, S))
        dout = torch.randn(1, L, 256, device='cuda', dtype=dt)

        def lib_f():
            return F.scaled_dot_product_attention(q.view(1, L, 4, 64).transpose(1, 2), k.view(1, S, 4, 64).transpose(1, 2),
                                                  v.view(1, S, 4, 64).transpose(1, 2)).transpose(1, 2).reshape(1, L, 256)

        def own_f():
            return HA.full_attention(q, k, v, 4)

        res = {}
        for name, f in (('library', lib_f), ('own', own_f)):
            with torch.no_grad():
                tf = timeit(f)
            out = f()
            tb = timeit(lambda: torch.autograd.grad(out, (q, k, v), dout, retain_graph=True))
            res[name] = (tf, tb)
        fl = 4.0 * L * S * 256
        print(f'{str(dt)[6:]} L={L} S={S}: forward library {res["library"][0]:.0f} us, own {res["own"][0]:.0f} us ({fl / res["own"][0] * 1e-6:.0f} TFLOP/s) | '
              f'backward library {res["library"][1]:.0f} us, own {res["own"][1]:.0f} us ({3.5 * fl / res["own"][1] * 1e-6:.0f} TFLOP/s)')
