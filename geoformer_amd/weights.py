"""Deterministic random-init weights for benchmarks and smoke runs (no checkpoint is available
offline).  Every tensor of a GeoFormer state dict is filled from its NAME and shape only:
hash-uniform values (murmur3 finaliser of the element index salted with crc32(name)) scaled with
the Xavier-uniform bound for matrices/convolutions, near-identity affine terms for Layer/BatchNorm,
well-conditioned BatchNorm running statistics.  The same recipe is stated independently in
oracle/geoformer_oracle.py (tests check the two agree bit for bit)."""
import math
import zlib

import numpy as np
import torch


def _hash_uniform(n: int, salt: int) -> np.ndarray:
    x = (np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(16))) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x = ((x ^ (x >> np.uint64(13))) * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return x.astype(np.float64) / 4294967296.0


@torch.no_grad()
def deterministic_init_(module: torch.nn.Module) -> torch.nn.Module:
    sd = module.state_dict()
    for name, t in sd.items():
        if not t.dtype.is_floating_point:
            t.zero_()
            continue
        u = torch.from_numpy(_hash_uniform(t.numel(), zlib.crc32(name.encode()))) * 2.0 - 1.0
        if name.endswith('running_var'):
            val = 1.0 + 0.1 * u.abs()
        elif name.endswith('running_mean'):
            val = 0.02 * u
        elif t.dim() == 1 and name.endswith('weight'):
            val = 1.0 + 0.05 * u
        elif t.dim() == 1:
            val = 0.02 * u
        else:
            rf = t[0][0].numel() if t.dim() > 2 else 1
            val = u * math.sqrt(6.0 / (t.shape[1] * rf + t.shape[0] * rf))
        t.copy_(val.to(torch.float32).view_as(t).to(t.dtype))
    module.load_state_dict(sd)
    return module
