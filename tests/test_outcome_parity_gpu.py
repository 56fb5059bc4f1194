"""Outcome-level parity of the benched 16-bit modes (VERDICT r04 #2a): what the knife-edge match differences of fp16 / bf16 storage
do to the metric the reference is judged by - the HPatches homography-estimation AUC (hpatches_helper.py:13-56 `cal_error_auc`,
:185-239 the pair loop: matches -> findHomography(RANSAC, 3 px) -> mean corner error -> AUC@1/3/5/10; README.md:117).

No checkpoint and no HPatches data exist offline, so the protocol runs on SYNTHETIC sequences: 13 sequences x 5 pairs = 65 pairs of
planted feature maps of a planar scene (oracle/golden_inputs.py:hpatches_like_features: a continuous random field sampled at the
cells of image 0 and at H^-1 of the cells of image 1, 480x640 against 480x608 - 60x80 and 60x76 coarse grids, the shape class of
data_io.py:16-26), ground-truth homographies of HPatches-like strength (corner displacements up to 6k / 8k px for pair k = 1..5).
Both sides see the same maps and go through the same evaluation arithmetic (geoformer_amd.matcher.cal_error_auc / corner_error, pinned
by tests/golden/g12):
  product:  GeoFormer.forward_features in fp16 / bf16 storage, device RANSAC inside GeoModule, device RANSAC (3 px, sub-pixel
            keypoints) for the final homography - matcher.estimate_homography, as matcher.eval_hpatches does;
  oracle:   the fp32 restatement of the reference (oracle.geoformer_forward) with the C RANSAC inside, the C RANSAC's sub-pixel entry for
            the final homography.
RANSAC parity with OpenCV is UNPINNED (oracle/ransac_oracle.c header); both sides use the build's own algorithm, so the statement is:
"storage round-off moves the AUC by this much", not "this is the reference's AUC".

MEASURED (MI355X, round 5, gpurun_out/r05l_pytest.log; AUC@1/3/5/10 of the oracle 0.618 / 0.871 / 0.922 / 0.961, 192 matches per pair):
  fp32 parity mode   dAUC@3 = -4.4e-4 (dAUC@1 -1.2e-3; |corner-error difference| per pair 1.5e-3 px mean, 0.04 px max): NOT zero although both sides
                     compute in fp32 - on 3 of the 65 pairs 1-5 of 120-500 matches differ.  These planted maps hold mathematically TIED
                     candidates (a cell of image 1 that lands half-way between two cells of image 0 is equally similar to both; confidences of
                     such cells sit at 0.2-0.25, i.e. AT the threshold), and a tie is decided by the last bit of an fp32 sum whose order differs
                     between the MFMA kernels and torch's CPU GEMM.  The reference-generated goldens (g5, g10, g11) have no such ties and are
                     reproduced bit for bit.  This is the noise floor of the protocol on this data; north_star's 1e-3 holds;
  fp16 storage       dAUC@3 = -0.25e-3 ... -1.5e-3 over this round's arithmetic variants (dAUC@1 -2e-3 ... -4e-3; per pair 7e-3 ... 1e-2 px mean): up to three times that floor - the fine
                     arg-max flips and threshold-edge coarse matches of 16-bit storage.  north_star's 1e-3 is NOT met by the fp16 storage mode on
                     this 65-pair set in every variant; the gate below is the measured value + margin, so that a regression shows;
  bf16 storage       dAUC@3 = -2.5e-3 ... +1.7e-3 (per pair 4e-2 px mean, 0.7 px max: larger, sign-symmetric).
On 260 pairs (tools/outcome_parity_large.py 52, one-off; profiles/r05_outcome_parity_260_pairs.txt): dAUC@3 fp32 -2.0e-4, fp16 -6.2e-4, bf16 -6.4e-4 - the 65-pair
figures scatter around these with the sampling noise of the set; at that sample size the fp16 mode is inside north_star's 1e-3."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O
import golden_inputs as GI
import ransac_oracle as RO

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SEQS, PAIRS = 13, 5                        # 65 pairs (the protocol: image 1 against images 2..6 of every sequence)
THRES = (1, 3, 5, 10)
_oracle_rows = {}


def _oracle_side():
    """(corner error, #matches) per pair from the fp32 oracle; computed once for both storage modes."""
    if _oracle_rows:
        return _oracle_rows
    from geoformer_amd import matcher as MT
    W = O.make_weights()
    geo_cfg = O.default_geo_config()
    data = {'image0': torch.zeros(1, 1, 480, 640), 'image1': torch.zeros(1, 1, 480, 608)}
    for s in range(SEQS):
        for k in range(1, PAIRS + 1):
            f0, f1, H = GI.hpatches_like_features(s, k)
            ref = O.geoformer_forward(W, dict(data), None, geo_cfg, RO.make_homography_fn(), None, (f0, f1))
            k0, k1 = ref['mkpts0_f'].numpy(), ref['mkpts1_f'].numpy()
            Hp, _ = RO.find_homography_subpixel(k0, k1, 3.0)
            _oracle_rows[(s, k)] = (MT.corner_error(Hp, H, 640, 480) if Hp is not None else float('nan'), len(k0))
    return _oracle_rows


def _product_side(precision, seqs=None):
    from geoformer_amd import matcher as MT
    from test_e2e_gpu import build, to_dev
    st = {'fp16': torch.float16, 'bf16': torch.bfloat16, 'fp32': torch.float32}[precision]
    m = build(0.2, 0.1, precision)
    m.geo_module.homography_fn = None      # device RANSAC
    data = to_dev({'image0': torch.zeros(1, 1, 480, 640), 'image1': torch.zeros(1, 1, 480, 608)})
    rows = {}
    for s in range(SEQS if seqs is None else seqs):
        for k in range(1, PAIRS + 1):
            (c0, f0), (c1, f1), H = GI.hpatches_like_features(s, k)
            with torch.no_grad():
                out = m.forward_features(dict(data), *(t.to(DEV).to(st) for t in (c0, f0, c1, f1)))
            matches = torch.cat([out['mkpts0_f'], out['mkpts1_f']], 1).float().cpu().numpy()
            Hp = None
            if len(matches) >= 4:
                Hp, _ = MT.estimate_homography(matches, 3.0, DEV)
            rows[(s, k)] = (MT.corner_error(Hp, H, 640, 480) if Hp is not None else (float('nan') if seqs is None else float('inf')), len(matches))
    return rows


def _auc_terms(err, t):
    """Per-pair contribution to AUC@t (the area under recall-vs-error up to t, normalised): max(0, 1 - e / t); their mean is the AUC up to the
    trapezoid rule's discretisation - used for the PAIRED standard error of an AUC difference."""
    return np.clip(1.0 - np.asarray(err, dtype=float) / t, 0.0, None)


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_hpatches_protocol_auc_260_pairs(precision):
    """The outcome-level parity at the sample size where north_star's 1e-3 can be told from sampling noise (VERDICT r05 #2c): 52 sequences x 5
    pairs.  Oracle side = the committed fixture tests/golden/g18_outcome_oracle_260.npz (oracle/gen_outcome_golden.py: the fp32 oracle + C RANSAC
    on the CPU, ~2 s per pair; the 65-pair test above recomputes its first 13 sequences live and checks them against it); product side = 260
    forwards on the GPU.  A pair without an estimate (fewer than 4 matches / no model) counts as an error above every threshold on both sides.
    Gate: |dAUC@3| <= 1e-3 (fp16: north_star's own number; measured -6.2e-4 in round 5, r06: see the printed line), with the PAIRED standard
    error of the difference printed beside it."""
    import os
    from geoformer_amd import matcher as MT
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g18_outcome_oracle_260.npz'))
    assert len(G['err']) == 260
    got = _product_side(precision, seqs=52)
    keys = [(int(s), int(k)) for s, k in zip(G['seq'], G['pair'])]
    er, nr = G['err'].astype(float), G['nmatch']
    eg = np.array([got[k][0] for k in keys]); ng = np.array([got[k][1] for k in keys])
    auc_r, auc_g = MT.cal_error_auc(er, THRES), MT.cal_error_auc(eg, THRES)
    both = np.isfinite(er) & np.isfinite(eg)
    d = eg[both] - er[both]
    se = [float(np.std(_auc_terms(eg, t) - _auc_terms(er, t), ddof=1) / np.sqrt(len(er))) for t in THRES]
    print(f'HPatches-protocol outcome parity at 260 pairs, {precision} product vs the fp32 oracle fixture: failed pairs oracle {int((~np.isfinite(er)).sum())} / '
          f'product {int((~np.isfinite(eg)).sum())}; matches per pair {nr.mean():.0f} / {ng.mean():.0f}')
    print(f'  AUC@1/3/5/10 oracle  {np.round(auc_r, 5).tolist()}')
    print(f'  AUC@1/3/5/10 product {np.round(auc_g, 5).tolist()}')
    print(f'  dAUC                 {np.round(auc_g - auc_r, 5).tolist()}   paired standard error {np.round(se, 5).tolist()}')
    print(f'  corner-error difference over {int(both.sum())} pairs: mean {d.mean():+.2e} px (standard error {d.std(ddof=1) / np.sqrt(len(d)):.1e}), '
          f'mean |d| {np.abs(d).mean():.2e}, max |d| {np.abs(d).max():.2e}, pairs with |d| > 0.01 px: {int((np.abs(d) > 0.01).sum())}')
    assert int((~np.isfinite(er)).sum()) <= 2 and int((~np.isfinite(eg)).sum()) <= int((~np.isfinite(er)).sum()) + 1
    assert abs(auc_g[1] - auc_r[1]) <= GATE_260[precision], (precision, 'dAUC@3', float(auc_g[1] - auc_r[1]))
    assert np.abs(auc_g - auc_r).max() <= {'fp16': 4e-3, 'bf16': 9e-3}[precision], (precision, (auc_g - auc_r).tolist())


# north_star: "reproducing the reference's HPatches AUC within 1e-3".  MEASURED (MI355X, round 6, profiles/r06_outcome_parity_260.txt):
#   fp16  dAUC@1/3/5/10 = +1.8e-3 / +3.4e-4 / +1.9e-4 / +0.9e-4, paired standard error 1.1e-3 / 4.4e-4 / 2.6e-4 / 1.3e-4: inside 1e-3 at @3 and above
#   bf16  dAUC@1/3/5/10 = -2.0e-3 / -1.5e-3 / -2.4e-3 / -3.1e-3, paired standard error 2.6e-3 / 1.4e-3 / 2.0e-3 / 2.9e-3: one pair of the 260 moves by
#         8 px; the difference is ~1 standard error from zero - 8 significant bits do not resolve 1e-3 at this sample size, the gate is 2 sigma
GATE_260 = {'fp16': 1e-3, 'bf16': 3e-3}


@pytest.mark.parametrize('precision', ['fp32', 'fp16', 'bf16'])
def test_hpatches_protocol_auc_product_vs_fp32_oracle(precision):
    from geoformer_amd import matcher as MT
    ref, got = _oracle_side(), _product_side(precision)
    keys = sorted(ref)
    assert len(keys) >= 64
    er = np.array([ref[k][0] for k in keys]); eg = np.array([got[k][0] for k in keys])
    nr = np.array([ref[k][1] for k in keys]); ng = np.array([got[k][1] for k in keys])
    auc_r, auc_g = MT.cal_error_auc(er, THRES), MT.cal_error_auc(eg, THRES)
    ok = ~(np.isnan(er) | np.isnan(eg))
    print(f'HPatches-protocol outcome parity, {precision} product vs fp32 oracle, {len(keys)} synthetic pairs '
          f'(matches per pair: oracle {nr.mean():.0f}, product {ng.mean():.0f}; failed: oracle {int(np.isnan(er).sum())}, product {int(np.isnan(eg).sum())})')
    print(f'  AUC@1/3/5/10 oracle  {np.round(auc_r, 5).tolist()}')
    print(f'  AUC@1/3/5/10 product {np.round(auc_g, 5).tolist()}')
    print(f'  dAUC                 {np.round(auc_g - auc_r, 5).tolist()}   corner error: oracle mean {np.nanmean(er):.4f} px, '
          f'|product - oracle| mean {np.abs(eg - er)[ok].mean():.2e} max {np.abs(eg - er)[ok].max():.2e} px')
    # the workload is a real one: matches on every pair, errors inside the AUC's range on most of them
    assert nr.min() >= 30 and ng.min() >= 30 and np.nanmedian(er) < 3.0 and 0.2 < auc_r[1] < 0.999
    assert int(np.isnan(er).sum()) == int(np.isnan(eg).sum()) == 0
    worst = np.argsort(-np.abs(np.where(ok, eg - er, 0.0)))[:5]
    print('  largest per-pair differences (sequence, pair: oracle / product error px, matches): ' +
          '; '.join(f'{keys[i]}: {er[i]:.3f} / {eg[i]:.3f}, {nr[i]} / {ng[i]}' for i in worst))
    if precision == 'fp32':
        # the live oracle of this box against the committed fixture of the 260-pair test (its first 13 sequences): the same pairs give the same
        # errors up to the knife-edge flips another host's fp32 summation order can cause (thread count changes mconf by <= 6e-5)
        import os
        G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g18_outcome_oracle_260.npz'))
        fix = {(int(a), int(b)): float(c) for a, b, c in zip(G['seq'], G['pair'], G['err'])}
        ef = np.array([fix[k] for k in keys])
        same = np.abs(ef - er) <= 1e-3
        print(f'  live oracle vs fixture g18: {int(same.sum())} of {len(keys)} pairs within 1e-3 px; AUC@3 {MT.cal_error_auc(ef, THRES)[1]:.5f} (fixture) / {auc_r[1]:.5f} (live)')
        assert same.sum() >= len(keys) - 5 and abs(MT.cal_error_auc(ef, THRES)[1] - auc_r[1]) <= 5e-4
    if precision == 'fp32':                 # the parity mode: the protocol's own noise floor on maps with tied candidates (docstring)
        assert abs(auc_g[1] - auc_r[1]) <= 1e-3, (precision, 'dAUC@3', float(auc_g[1] - auc_r[1]))      # north_star's number
        assert np.abs(eg - er)[ok].mean() <= 5e-3 and (np.abs(ng - nr) <= np.maximum(3, 0.02 * nr)).all()
        return
    gate = {'fp16': 2.5e-3, 'bf16': 5e-3}[precision]              # measured -1.5e-3 / +1.7e-3 at worst (docstring) + margin
    assert abs(auc_g[1] - auc_r[1]) <= gate, (precision, 'dAUC@3', float(auc_g[1] - auc_r[1]))
    # (AUC@1 is the touchiest: a 1 px threshold against 0.4 px mean errors; measured up to -4e-3 in fp16, -1.2e-2 in bf16)
    assert np.abs(auc_g - auc_r).max() <= {'fp16': 7.5e-3, 'bf16': 2.5e-2}[precision], (precision, (auc_g - auc_r).tolist())
