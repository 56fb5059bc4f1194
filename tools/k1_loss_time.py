"""Times the fused coarse loss (forward: statistics + per-positive terms; backward: two panel sweeps) at 8 pairs of
6400 x 6400, against torch autograd on the materialised confidence matrix (1 pair, scaled)."""
import sys, torch
sys.path.insert(0, '.')
from geoformer_amd import ops
N, L, S, C, T = 8, 6400, 6400, 256, 0.1
f0 = (torch.randn(N, L, C, device='cuda') * 0.75).half()
f1 = (f0.float()[:, torch.randperm(S, device='cuda')] + 0.6 * torch.randn(N, S, C, device='cuda')).half()
pi = torch.arange(0, L, 2, device='cuda').repeat(N)
pb = torch.arange(N, device='cuda').repeat_interleave(L // 2)
pj = torch.randint(0, S, (pi.numel(),), device='cuda')
def ev(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def hip():
    h0, h1 = f0.clone().requires_grad_(True), f1.clone().requires_grad_(True)
    loss, _ = ops.coarse_focal_loss(h0, h1, pb, pi, pj, T)
    loss.backward()
def auto():
    a0, a1 = f0[:1].float().requires_grad_(True), f1[:1].float().requires_grad_(True)
    sim = torch.einsum('nlc,nsc->nls', a0 / C ** .5, a1 / C ** .5) / T
    conf = torch.softmax(sim, 1) * torch.softmax(sim, 2)
    p = torch.clamp(conf, 1e-6, 1 - 1e-6)[pb[:L // 2] * 0, pi[:L // 2], pj[:L // 2]]
    (-0.25 * (1 - p) ** 2 * p.log()).sum().backward()
t_hip, t_auto = ev(hip), ev(auto)
flops = N * 2 * (2 * L * S * C * 2 + 2.0 * L * S * C) + N * 2.0 * L * S * C     # 2 sweeps x (sim + grad GEMM) + statistics
print(f'fused HIP loss fwd+bwd, {N} pairs: {t_hip:.2f} ms ({t_hip / N:.3f} ms/pair, {flops / t_hip * 1e-9:.0f} TFLOP/s); '
      f'torch autograd, 1 pair: {t_auto:.2f} ms')
