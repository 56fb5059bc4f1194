"""Fixture of the outcome-level parity test at 260 pairs (tests/test_outcome_parity_gpu.py::test_hpatches_protocol_auc_260_pairs): the fp32
ORACLE's side of the synthetic HPatches protocol - per pair (sequence s = 0..51, pair k = 1..5 of oracle/golden_inputs.py:
hpatches_like_features) the mean corner error of the homography the C RANSAC (oracle/ransac_oracle.c, 3 px, sub-pixel keypoints) estimates
from the oracle's matches, and the number of matches.  A failed estimation (fewer than 4 matches / no model) is recorded as +inf, which
cal_error_auc counts as "above every threshold" (hpatches_helper.py:36-56 counts failed pairs the same way).

Test infrastructure: this file runs the oracle (CPU, ~2 s per pair) and writes tests/golden/g18_outcome_oracle_260.npz; nothing of the product
is involved.  The live 65-pair test recomputes the first 13 sequences on the spot and checks them against this file.
    python oracle/gen_outcome_golden.py [sequences=52]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
import geoformer_oracle as O
import golden_inputs as GI
import ransac_oracle as RO
from geoformer_amd import matcher as MT


def oracle_row(W, geo_cfg, s, k):
    data = {'image0': torch.zeros(1, 1, 480, 640), 'image1': torch.zeros(1, 1, 480, 608)}
    f0, f1, H = GI.hpatches_like_features(s, k)
    with torch.no_grad():
        ref = O.geoformer_forward(W, data, None, geo_cfg, RO.make_homography_fn(), None, (f0, f1))
    k0, k1 = ref['mkpts0_f'].numpy(), ref['mkpts1_f'].numpy()
    Hp = None
    if len(k0) >= 4:
        Hp, _ = RO.find_homography_subpixel(k0, k1, 3.0)
    return (MT.corner_error(Hp, H, 640, 480) if Hp is not None else float('inf')), len(k0)


def main():
    seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 52
    W, geo_cfg = O.make_weights(), O.default_geo_config()
    rows = []
    t0 = time.time()
    for s in range(seqs):
        for k in range(1, 6):
            e, n = oracle_row(W, geo_cfg, s, k)
            rows.append((s, k, e, n))
        print(f'sequence {s} done ({time.time() - t0:.0f} s)', file=sys.stderr, flush=True)
    a = np.array(rows, dtype=np.float64)
    out = os.path.join(ROOT, 'tests', 'golden', 'g18_outcome_oracle_260.npz')
    np.savez(out, seq=a[:, 0].astype(np.int32), pair=a[:, 1].astype(np.int32), err=a[:, 2], nmatch=a[:, 3].astype(np.int32))
    fin = a[np.isfinite(a[:, 2])]
    print(f'{len(a)} pairs -> {out}; failed {int((~np.isfinite(a[:, 2])).sum())}, mean corner error {fin[:, 2].mean():.4f} px, '
          f'AUC@1/3/5/10 {np.round(MT.cal_error_auc(a[:, 2], (1, 3, 5, 10)), 5).tolist()}, matches per pair {a[:, 3].mean():.0f}')


if __name__ == '__main__':
    main()
