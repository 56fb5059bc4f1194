"""The fine level's window linear attention backward (gf_window_linear_attention_backward) at the training step's size: 41,858 windows of
25 tokens x 128 channels, bf16.   python tools/window_la_bwd_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geoformer_amd import ops
torch.manual_seed(0)
for Nw in (41858, 20000):
    q, k, v, d = (torch.randn(Nw, 25, 128, device='cuda', dtype=torch.bfloat16) for _ in range(4))
    for _ in range(3):
        ops.window_linear_attention_backward(q, k, v, d)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        ops.window_linear_attention_backward(q, k, v, d)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 20 * 1e6
    print(f'{Nw} windows: {us:.0f} us per call ({7 * Nw * 25 * 128 * 2 / us * 1e-3:.0f} GB/s of operand traffic)')
