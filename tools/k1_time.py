import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geoformer_amd import ops
N, L, S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 6400, 6400, 256
THR = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2      # 0.0 selects the dense-candidate variant bench.py runs
for dt in (torch.float16, torch.float32):
    f0 = (torch.randn(N, L, C, device='cuda') * 1.3).to(dt)
    f1 = (f0[:, torch.randperm(S, device='cuda')].float() + 0.4 * torch.randn(N, S, C, device='cuda')).to(dt)
    for _ in range(3):
        out = ops.dual_softmax_match(f0, f1, 0.1, THR, (80, 80), (80, 80), 8.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.dual_softmax_match(f0, f1, 0.1, THR, (80, 80), (80, 80), 8.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(dt, 'N', N, 'ms/call', ms, 'M', int(out['counts'][0]), 'conf GB/s', N * L * S * 4 / ms / 1e6)
