"""Does the static part of the forward (backbone -> ... -> second coarse matching, everything before the first host sync)
capture into a hipGraph, and what does replaying it buy over eager launches?  python tools/graph_probe.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geoformer_amd import miopen as gf_miopen
gf_miopen.use_shipped_find_db()
import torch
import bench as B
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
model, _ = B.build_model('fp16', 0.0, 0.0, dev)
i0, i1 = B.synth_pairs(batch, 0, 640, dev)


def static_part(a, b):
    data = {'image0': a, 'image1': b}
    with torch.no_grad():
        out = model.forward_static(data)
    return out


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for _ in range(3):
    static_part(i0, i1)
torch.cuda.synchronize()
t_eager = timeit(lambda: static_part(i0, i1))
print(f'eager static part: {t_eager:.2f} ms per {batch} pairs')
s0, s1 = i0.clone(), i1.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = static_part(s0, s1)
    torch.cuda.synchronize()
    t_graph = timeit(g.replay)
    print(f'graph replay:      {t_graph:.2f} ms per {batch} pairs; counts after replay {out["_coarse_dev"]["counts"].tolist()[:3]}')
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:400])
