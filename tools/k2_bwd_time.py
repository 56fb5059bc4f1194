"""K2's backward (gf_linear_attention_backward) at the training step's shape: 8 images, 6400 x 6400 tokens, 8 heads of 32, bf16.
   python tools/k2_bwd_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geoformer_amd import ops
torch.manual_seed(0)
for N, L in ((8, 6400), (4, 4800), (16, 6400)):
    q, k, v, d = (torch.randn(N, L, 256, device='cuda', dtype=torch.bfloat16) for _ in range(4))
    for _ in range(3):
        ops.linear_attention_backward(q, k, v, d, 8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        ops.linear_attention_backward(q, k, v, d, 8)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 20 * 1e6
    print(f'{N} x {L}: {us:.0f} us per call ({7 * N * L * 256 * 2 / us * 1e-3:.0f} GB/s of operand traffic)')
