"""Timing of gf_ransac_homography_v2 on the bench's load (8 pairs x ~2300 coarse matches, shift-by-one-cell correspondences
with a share of outliers):  python tools/ransac_time.py [matches per pair] [outlier fraction]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from geoformer_amd import ops      # noqa: E402

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
N = 8
rng = np.random.default_rng(0)
p0 = np.stack([rng.integers(0, 79, N * n_per) * 8, rng.integers(0, 79, N * n_per) * 8], 1)
p1 = p0 + 8
out = rng.random(N * n_per) < frac
near = (~out) & (rng.random(N * n_per) < 0.05)          # a few inliers one cell off (distance 8 px = the threshold): LM has work to do
p1[near, 0] += 8
p1[out] = np.stack([rng.integers(0, 80, out.sum()) * 8, rng.integers(0, 80, out.sum()) * 8], 1)
mk0, mk1 = torch.tensor(p0, dtype=torch.float32, device='cuda'), torch.tensor(p1, dtype=torch.float32, device='cuda')
counts = torch.tensor([N * n_per] + [n_per] * N, dtype=torch.int32, device='cuda')
for lm in (10, 0):
    for _ in range(3):
        rs = ops.ransac_homography(mk0, mk1, counts, N, 8, lm_iters=lm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        rs = ops.ransac_homography(mk0, mk1, counts, N, 8, lm_iters=lm)
    torch.cuda.synchronize()
    print(f'lm_iters {lm:2d}: {(time.perf_counter() - t) / 50 * 1e6:7.1f} us per call ({N} pairs x {n_per} matches), valid {rs["valid"].tolist()}, '
          f'inliers {int(rs["keep"].sum())}')
print(rs['M'][0])
