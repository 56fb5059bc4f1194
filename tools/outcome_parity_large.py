"""The outcome-level parity measurement of tests/test_outcome_parity_gpu.py on a LARGER synthetic HPatches-protocol set (one-off, not part of the
suite): python tools/outcome_parity_large.py [sequences=52]  ->  AUC@1/3/5/10 of the fp32 oracle and of the product in fp32 / fp16 / bf16, the deltas
and their per-pair statistics (standard error of the mean corner-error difference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_outcome_parity_gpu as T
from geoformer_amd import matcher as MT

T.SEQS = int(sys.argv[1]) if len(sys.argv) > 1 else 52
ref = T._oracle_side()
keys = sorted(ref)
er = np.array([ref[k][0] for k in keys]); nr = np.array([ref[k][1] for k in keys])
auc_r = MT.cal_error_auc(er, T.THRES)
print(f'{len(keys)} synthetic pairs ({T.SEQS} sequences x {T.PAIRS}); oracle (fp32): AUC@1/3/5/10 {np.round(auc_r, 5).tolist()}, mean corner error {np.nanmean(er):.4f} px, failed pairs {int(np.isnan(er).sum())}, '
      f'matches per pair {nr.mean():.0f} (min {nr.min()})', flush=True)
for prec in ('fp32', 'fp16', 'bf16'):
    got = T._product_side(prec)
    eg = np.array([got[k][0] for k in keys])
    auc_g = MT.cal_error_auc(eg, T.THRES)
    d = (eg - er)[~(np.isnan(eg) | np.isnan(er))]
    print(f'{prec}: AUC {np.round(auc_g, 5).tolist()}  dAUC {np.round(auc_g - auc_r, 5).tolist()}  corner-error difference: mean {d.mean():+.2e} px '
          f'(standard error {d.std(ddof=1) / np.sqrt(len(d)):.1e}), mean |d| {np.abs(d).mean():.2e}, max |d| {np.abs(d).max():.2e}, pairs with |d| > 0.01 px: {(np.abs(d) > 0.01).sum()}',
          flush=True)
