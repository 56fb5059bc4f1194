"""The linear layers' weight gradient (k_train.hip: wgrad_kernel + wgrad_reduce) at the training step's shapes; run once per variant library
(GF_LIB_PATH=tools/ab/wg384.so ...) to compare chunk targets.   python tools/wgrad_chunks_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geoformer_amd import ops

torch.manual_seed(0)
out = []
for T, co, ci in ((51200, 256, 256), (102400, 256, 256), (51200, 512, 512), (51200, 256, 512), (81264, 256, 256), (1049000, 128, 128), (6400, 256, 256)):
    dy = torch.randn(T, co, device='cuda', dtype=torch.bfloat16)
    x = torch.randn(T, ci, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear_wgrad(dy, x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        ops.linear_wgrad(dy, x)
    torch.cuda.synchronize()
    out.append(f'{T}x{co}x{ci}: {(time.perf_counter() - t) / 30 * 1e6:.1f} us')
print(os.environ.get('GF_LIB_PATH', 'default'), ' | '.join(out))
