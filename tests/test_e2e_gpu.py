"""End-to-end parity of geoformer_amd.GeoFormer against the reference's golden vectors (fp32 parity mode: coarse indices
bit-exact) and, for the fp16 storage mode the bench runs, against the oracle's storage mode (the reference's arithmetic
with fp16 round trips at the kernels' rounding points, oracle/geoformer_oracle.py:geoformer_forward_storage; device
RANSAC on this side, its C statement on that side): coarse indices bit-exact except for counted knife-edge matches."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O
import golden_inputs as GI
import ransac_oracle as RO

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build(coarse_thr, fine_thr, precision='fp32'):
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.geo_config import get_cfg_model
    gc = get_cfg_model(); gc.update(coarse_thr=coarse_thr, fine_thr=fine_thr, precision=precision)
    m = GeoFormer(get_default_cfg(), gc).eval()
    m.load_state_dict(O.make_weights())
    return m.to(DEV)


def replay(G):
    calls = iter(range(int(G['ransac_ncalls'])))

    def fn(a, b):
        i = next(calls)
        np.testing.assert_array_equal(a, G[f'ransac{i}_kp0']); np.testing.assert_array_equal(b, G[f'ransac{i}_kp1'])
        return (G[f'ransac{i}_M'].copy() if G[f'ransac{i}_valid'] else None), G[f'ransac{i}_mask'].copy()
    return fn


def to_dev(d):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), np.asarray(b), rtol=rtol, atol=atol)


# BASELINE.md section 3's gate for the fp32 parity mode: mconf within 1e-4 (ABSOLUTE: confidences live in [0, 1]) of the reference's
# own run, after 14 exact-fp32 transformer layers and two dual-softmax stages whose summation orders differ from torch's
MCONF_ATOL_FP32 = 1e-4


def mconf_close_fp32(got, want, what):
    g, w = got.detach().float().cpu().numpy(), np.asarray(want, dtype=np.float32)
    d = np.abs(g - w)
    rel = d / np.maximum(np.abs(w), 1e-12)
    print(f'{what}: fp32 mconf on {len(w)} matches: max |d| {d.max() if len(d) else 0.0:.2e} (gate {MCONF_ATOL_FP32:g} absolute), '
          f'max relative {rel.max() if len(d) else 0.0:.2e}')
    assert len(g) == len(w) and (len(d) == 0 or float(d.max()) <= MCONF_ATOL_FP32), (what, float(d.max()))


@pytest.mark.parametrize('name', list(GI.g10_cases().keys()))
def test_golden_fp32(golden, name):
    """Reference outputs, injected reference homographies: coarse and fine indices bit-exact."""
    G, case = golden(name), GI.g10_cases()[name]
    m = build(case['coarse_thr'], case['fine_thr'])
    m.geo_module.homography_fn = replay(G)
    data = to_dev(case['data'])
    with torch.no_grad():
        if case['feats'] is None:
            out = m(data)
        else:
            (c0, f0), (c1, f1) = case['feats']
            out = m.forward_features(data, c0.to(DEV), f0.to(DEV), c1.to(DEV), f1.to(DEV))
    for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids'):
        np.testing.assert_array_equal(out[k].cpu().numpy(), G['out_' + k])
    for k in ('mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f'):
        close(out[k], G['out_' + k], 1e-6, 1e-5)
    mconf_close_fp32(out['mconf'], G['out_mconf'], name)
    close(out['mconf'], G['out_mconf'], 1e-3, 1e-6)
    close(out['conf_matrix'], G['out_conf_matrix'], 5e-3, 1e-7)
    close(out['dect_conf_matrix'], G['out_dect_conf_matrix'], 5e-3, 1e-7)
    close(out['fine_matrix'][:12], G['out_fine_matrix_head'], 5e-3, 1e-7)
    for k in ('loftr_f0', 'loftr_f1', 'geo_f0', 'geo_f1'):
        close(out['_feat_dev'][k][..., ::4], G['mid_' + k], 1e-3, 5e-4)
    assert int(out['W']) == 5


def test_640_digest_fp32(golden):
    """BASELINE size (80x80 grids): coarse ids and fine keypoints bit-identical to the reference run."""
    G, case = golden('g11_e2e_640_digest'), GI.g11_inputs()
    m = build(case['coarse_thr'], case['fine_thr'])
    n = int(G['ransac0_n'])
    mask = np.unpackbits(G['ransac0_mask'])[:n].astype(np.uint8)[:, None]
    m.geo_module.homography_fn = lambda a, b: (G['ransac0_M'].copy(), mask)
    (c0, f0), (c1, f1) = case['feats']
    with torch.no_grad():
        out = m.forward_features(to_dev(case['data']), c0.to(DEV), f0.to(DEV), c1.to(DEV), f1.to(DEV))
    assert len(out['b_ids']) == int(G['M']) and len(out['mkpts0_f']) == int(G['Mf'])
    np.testing.assert_array_equal(out['i_ids'].cpu().numpy(), G['i_ids'].astype(np.int64))
    np.testing.assert_array_equal(out['j_ids'].cpu().numpy(), G['j_ids'].astype(np.int64))
    np.testing.assert_array_equal(GI.digest(out['b_ids'].cpu(), out['i_ids'].cpu(), out['j_ids'].cpu()), G['coarse_ids_digest'])
    np.testing.assert_array_equal(GI.digest(out['mkpts0_f'].cpu(), out['mkpts1_f'].cpu()), G['fine_kpts_digest'])
    mconf_close_fp32(out['mconf'][:64], G['mconf_head'], '640 digest')
    close(out['mconf'][:64], G['mconf_head'], 2e-3, 1e-6)


def test_device_ransac_vs_oracle():
    """No injection: device RANSAC in the product, its C statement in the oracle (fp32 parity mode)."""
    case = GI.g10_cases()['g10b_e2e_planted_n2']
    m = build(case['coarse_thr'], case['fine_thr'], 'fp32')
    (c0, f0), (c1, f1) = case['feats']
    with torch.no_grad():
        out = m.forward_features(to_dev(case['data']), *(t.to(DEV) for t in (c0, f0, c1, f1)))
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=case['coarse_thr'], fine_thr=case['fine_thr'])
    ref = O.geoformer_forward(O.make_weights(), dict(case['data']), None, geo_cfg, RO.make_homography_fn(), None,
                              ((c0, f0), (c1, f1)))
    rs = out['_geo_dev']['ransac']
    assert [int(v) for v in rs['valid']] == [1, 1]
    for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids'):
        np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k].numpy())
    close(out['mkpts0_f'], ref['mkpts0_f'], 1e-6, 1e-5); close(out['mkpts1_f'], ref['mkpts1_f'], 1e-6, 1e-5)
    mconf_close_fp32(out['mconf'], ref['mconf'].numpy(), 'device RANSAC vs oracle')
    close(out['mconf'], ref['mconf'], 1e-3, 1e-6)


# ------------------------------------------------------------------------------------------------------------
# fp16 storage mode (what bench.py runs) against the oracle's storage mode
# ------------------------------------------------------------------------------------------------------------
# Noise floor of the mode.  The oracle and the kernels evaluate the SAME fp16-storage arithmetic but sum in different
# orders; a sum that lands within ~1e-7 of an fp16 rounding boundary rounds to the other neighbour (5e-4 relative) and
# that difference is carried through 14 layers into an exponential with 1/temperature = 10.  Measured at 640x640
# (1206 matches): the two confidence matrices differ by 0.8 % on average, 4.6 % at the 99th percentile.  A match whose
# confidence is within KNIFE_EDGE of a decision boundary (threshold 0.2, row / column maximum) can therefore fall on either
# side; everything else must agree bit for bit.  At 640x640 7 of 1206 matches differ, every one of them at the threshold
# (oracle confidences 0.1962 .. 0.2014).
KNIFE_EDGE = 3e-2
FINE_EDGE = 5e-2          # the same for the fine threshold: the 25x25 matrices carry two more fp16 layers
# the fine arg-max: two candidates of a match's 25 x 25 matrix closer than this (relative) can swap.  The fine matrices of the two sides
# agree far better than the coarse ones (median relative difference 2e-6 in fp16 - they are fp32 products of nearly identical windows),
# so the band is narrow: 1e-3 is ~10x the 99th percentile of that difference; bf16: 8x wider, as everywhere
FINE_ARGMAX_EDGE = 1e-3
# Round 5 (VERDICT r04 #2b): the band edge of a case is MEASURED on that case - the 99.9th percentile of the relative
# difference of the two confidence matrices over the decision-relevant entries (oracle confidence >= thr / 2; >= 1e-2 in the dense
# mode) - and capped by the mode's EDGE below; a kernel whose round-off grows widens nothing: the measured percentile has its own
# ceiling NOISE_CEIL = the largest value measured on the MI355X over all cases of the mode + a quarter to a third (gpurun_out/r05*_pytest.log:
# fp16 0.002 ... 0.082, bf16 0.025 ... 0.111 - the largest on the 60-match g10c case, where the percentile is nearly a maximum; the large values belong to the `g11` / HPatches-shaped planted maps whose confidences sit
# close to the threshold, the bench's own maps measure 0.006 ... 0.020 in fp16 and 0.075 in bf16).  fp16 keeps EDGE = 3 % (every
# difference ever seen sat inside it); bf16 was "8 x fp16" = 24 %, which made the band 40 % of all matches - its measured noise is no
# larger than fp16's p99.9 on the same maps (0.085 / 0.099 against 0.082 / 0.081), so its edge is min(measured, 12.5 %): B = 147 ...
# 190 instead of 445 ... 760.  On top of the band rule every case carries a CAP on the number of differing matches: the count measured
# on the MI355X plus a small margin (the counts move by a few with every change of an fp32 summation order), == 0 where the case has
# always been bit-exact.
EDGE = {'fp16': KNIFE_EDGE, 'bf16': 0.125}
NOISE_CEIL = {KNIFE_EDGE: 0.10, 0.125: 0.15}        # keyed by the mode's EDGE (compare_with_storage_oracle gets the edge, not the mode)
FINE_WIDEN = {KNIFE_EDGE: 1.0, 0.125: 8.0}           # the fine level's bands in bf16: 8 x fp16's (8 instead of 11 significant bits), as before
# the two confidence matrices (entries > 1e-3): (max, mean) relative difference - 14 layers of storage round-off feed an
# exponential with 1 / temperature = 10; the maxima measured on the MI355X over all cases (fp16: 0.223 / 8.3e-3 on the HPatches-shaped maps,
# 0.053 / 4.1e-3 on the bench's; bf16: 0.246 / 0.027) + a third
CONF_TOL = {'fp16': (0.3, 1.2e-2), 'bf16': (0.4, 4e-2)}


def measured_edge(oc, rc, thr, q=0.999):
    """The q-quantile of |product - oracle| / oracle over the decision-relevant entries of the oracle's matrix, and the quantiles
    printed beside it."""
    floor = 0.5 * thr if thr > 0 else 1e-2
    rel = ((oc - rc).abs() / rc.clamp_min(1e-12))[rc >= floor].double()
    if rel.numel() == 0:
        return 0.0, (0.0, 0.0, 0.0, 0.0)
    qs = torch.quantile(rel, torch.tensor([0.5, 0.99, q, 1.0], dtype=torch.float64))
    return float(qs[2]), tuple(float(v) for v in qs)


from parity_band import decision_band, knife_bound          # noqa: E402  (tests/parity_band.py: the data-derived bound)


def _kept(fine_matrix, fine_thr):
    """Indices (into the coarse match list) of the matches FineMatching2 keeps: the global maximum of the match's 25x25
    matrix exceeds fine_thr (fine_matching2.py:73-91 - the global arg-max is its row's and its column's maximum)."""
    if fine_matrix.shape[0] == 0:
        return torch.zeros(0, dtype=torch.long)
    return torch.where(fine_matrix.float().cpu().flatten(1).max(1)[0] > fine_thr)[0]


def compare_fine_on_common(out, ref, fine_thr, what, fine_edge=FINE_EDGE, argmax_edge=FINE_ARGMAX_EDGE):
    """Fine level on the coarse matches BOTH sides have (always - also when knife-edge matches differ): m_bids equal,
    fine keypoints identical except where the oracle's own 25 x 25 matrix has two candidates within `fine_edge` of each other
    (closer than `argmax_edge`; integer window offsets around exact coarse positions: a flipped arg-max moves a keypoint by >= 1 px;
    at most half of that band may flip - the rule of tests/parity_band.py), fine confidences agreeing to storage-noise level, kept/dropped
    decisions differing only at the fine threshold."""
    key = lambda d: list(zip(d['b_ids'].tolist(), d['i_ids'].tolist(), d['j_ids'].tolist()))      # noqa: E731
    ko, kr = key(out), key(ref)
    pos_r = {k: n for n, k in enumerate(kr)}
    fo, fr = out['fine_matrix'].float().cpu(), ref['fine_matrix']
    assert fo.shape[0] == len(ko) and fr.shape[0] == len(kr), (what, fo.shape, len(ko), fr.shape, len(kr))
    kept_o, kept_r = _kept(fo, fine_thr), _kept(fr, fine_thr)
    # each side's fine outputs are its kept coarse matches in coarse order (fine_matching2.py:88-91)
    assert len(kept_o) == len(out['mkpts0_f']) and len(kept_r) == len(ref['mkpts0_f']), (what, len(kept_o), len(out['mkpts0_f']))
    np.testing.assert_array_equal(out['m_bids'].cpu().numpy(), out['b_ids'].cpu()[kept_o].numpy())
    slot_r = {int(c): n for n, c in enumerate(kept_r.tolist())}
    pos_o = {k: n for n, k in enumerate(ko)}
    kept_o_set = set(kept_o.tolist())

    def at_fine_threshold(c_r):                        # the oracle's own best entry of that match sits at fine_thr
        v = float(fr[c_r].max())
        return abs(v - fine_thr) <= fine_edge * max(v, fine_thr)
    pairs, flipped = [], 0
    for n_o, c_o in enumerate(kept_o.tolist()):
        c_r = pos_r.get(ko[c_o])
        if c_r is None:
            continue                                   # a knife-edge coarse match of this side only
        if c_r in slot_r:
            pairs.append((n_o, slot_r[c_r]))
        else:                                          # kept here, dropped there
            flipped += 1
            assert at_fine_threshold(c_r), (what, 'kept here only', c_r, float(fr[c_r].max()))
    for c_r in kept_r.tolist():                        # kept there, dropped here
        c_o = pos_o.get(kr[c_r])
        if c_o is not None and c_o not in kept_o_set:
            flipped += 1
            assert at_fine_threshold(c_r), (what, 'kept there only', c_r, float(fr[c_r].max()))
    # band of the fine threshold: common matches whose best fine confidence (the oracle's) sits within FINE_EDGE of fine_thr
    common_r = [c_r for c_r in range(len(kr)) if kr[c_r] in pos_o]
    fine_band = sum(1 for c_r in common_r if at_fine_threshold(c_r)) if fr.shape[0] else 0
    assert flipped <= knife_bound(fine_band), (what, flipped, fine_band)
    if not pairs:
        return 0, flipped
    io = torch.tensor([p[0] for p in pairs]); ir = torch.tensor([p[1] for p in pairs])
    np.testing.assert_array_equal(out['m_bids'].cpu()[io].numpy(), ref['m_bids'][ir].numpy())
    same0 = (out['mkpts0_f'].cpu()[io] - ref['mkpts0_f'][ir]).abs().max(1)[0] < 1e-3
    same1 = (out['mkpts1_f'].cpu()[io] - ref['mkpts1_f'][ir]).abs().max(1)[0] < 1e-3
    frac = float((same0 & same1).float().mean())
    # arg-max band: common kept matches whose two best fine confidences (the oracle's) are within fine_edge of each other
    top2 = fr[kept_r[ir]].flatten(1).topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1]) / top2[:, 0]
    contested = gap < argmax_edge
    moved = ~(same0 & same1)
    assert bool(contested[moved].all()), (what, 'a fine keypoint moved although its arg-max is uncontested', int((moved & ~contested).sum()),
                                          float(gap[moved].max()))
    assert int(moved.sum()) <= knife_bound(int(contested.sum())), (what, int(moved.sum()), int(contested.sum()))
    mo, mr = out['mconf'].float().cpu()[io][same0 & same1], ref['mconf'][ir][same0 & same1]
    rel = (mo - mr).abs() / mr.clamp_min(1e-6)
    assert float(rel.median()) < 2e-2 and float(rel.mean()) < 5e-2, (what, float(rel.median()), float(rel.mean()))
    print(f'{what}: fine level on {len(pairs)} common matches: {100 * frac:.2f} % identical keypoints ({int(moved.sum())} moved of '
          f'{int(contested.sum())} with a contested arg-max, largest gap of a moved one {float(gap[moved].max()) if bool(moved.any()) else 0.0:.1e}), '
          f'{flipped} kept/dropped flips of {fine_band} in the fine band, '
          f'mconf rel. median {float(rel.median()):.1e}')
    return len(pairs), flipped


def compare_with_storage_oracle(out, ref, thr, what, fine_thr=0.1, edge=KNIFE_EDGE, conf_tol=CONF_TOL['fp16'], max_diff=None):
    """Coarse ids bit-exact, or: every match present on one side only lies in the decision band of the oracle's own confidence
    matrix (flip distance < the case's measured edge, capped by `edge` = the mode's EDGE), at most half of the band's population
    differs, and at most `max_diff` matches differ (the count measured on the MI355X + margin; 0 = the case is bit-exact).  The fine
    level is compared on the common matches in either case.  Returns (number of differences, band population)."""
    widen = FINE_WIDEN.get(edge, edge / KNIFE_EDGE)      # the fine level's bands
    a = set(zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()))
    r = set(zip(ref['b_ids'].tolist(), ref['i_ids'].tolist(), ref['j_ids'].tolist()))
    diff = sorted(a ^ r)
    oc, rc = out['conf_matrix'].float().cpu(), ref['conf_matrix']
    e_meas, (q50, q99, q999, qmax) = measured_edge(oc, rc, thr)
    e_used = min(max(e_meas, 1e-4), edge)
    band = decision_band(rc, thr, e_used)
    B = len(band)
    print(f'{what}: {len(r)} coarse matches, {len(diff)} differences (cap {max_diff}), band population B = {B} -> bound {knife_bound(B)}; '
          f'edge measured {e_meas:.4f} (cap {edge:g}, used {e_used:.4f}; noise ceiling {NOISE_CEIL.get(edge, 1.5 * edge):g}); relevant-entry relative difference '
          f'p50 {q50:.2e} p99 {q99:.2e} p99.9 {q999:.2e} max {qmax:.2e}')
    assert e_meas <= NOISE_CEIL.get(edge, 1.5 * edge), (what, 'the measured noise left the mode\'s ceiling', e_meas, NOISE_CEIL.get(edge))
    for k in diff:
        assert k in band, (what, k, 'differs outside the decision band')
    assert len(diff) <= knife_bound(B), (what, len(a), len(r), len(diff), B)
    if max_diff is not None:
        assert len(diff) <= max_diff, (what, 'more differing matches than measured + margin', len(diff), max_diff)
    if not diff:
        for k in ('b_ids', 'i_ids', 'j_ids'):
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k].numpy())      # same order too
    else:                                              # the common matches keep their relative (torch.where) order
        ka = [k for k in zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()) if k in r]
        kr = [k for k in zip(ref['b_ids'].tolist(), ref['i_ids'].tolist(), ref['j_ids'].tolist()) if k in a]
        assert ka == kr, what
    compare_fine_on_common(out, ref, fine_thr, what, FINE_EDGE * widen, FINE_ARGMAX_EDGE * widen)
    # the confidence matrix: 14 layers of 16-bit round-off noise feed an exponential with 1/temperature = 10
    big = rc > 1e-3
    rel = ((oc - rc).abs() / rc.clamp_min(1e-12))[big]
    print(f'{what}: confidence matrix, entries > 1e-3: relative difference max {float(rel.max()):.3f} mean {float(rel.mean()):.2e} '
          f'(tolerance {conf_tol[0]:g} / {conf_tol[1]:g})')
    assert float(rel.max()) < conf_tol[0] and float(rel.mean()) < conf_tol[1], (what, float(rel.max()), float(rel.mean()))
    return len(diff), B


def run_fp16(case, feats=None, data=None, precision='fp16'):
    st = {'fp16': torch.float16, 'bf16': torch.bfloat16}[precision]
    m = build(case['coarse_thr'], case['fine_thr'], precision)
    m.geo_module.homography_fn = None            # device RANSAC
    (c0, f0), (c1, f1) = feats or case['feats']
    data = data or case['data']
    with torch.no_grad():
        out = m.forward_features(to_dev(data), c0.to(DEV).to(st), f0.to(DEV).to(st), c1.to(DEV).to(st), f1.to(DEV).to(st))
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=case['coarse_thr'], fine_thr=case['fine_thr'])
    ref = O.geoformer_forward_storage(O.make_weights(), dict(data), st, None, geo_cfg, RO.make_homography_fn(),
                                      ((c0, f0), (c1, f1)))
    return out, ref


def test_no_coarse_match_branch():
    """M == 0: FinePreprocess / FineMatching2 early returns (fine_preprocess.py:35-38, fine_matching2.py:34-42)."""
    case = GI.g10_cases()['g10b_e2e_planted_n2']
    m = build(0.999999, 0.1)
    (c0, f0), (c1, f1) = case['feats']
    with torch.no_grad():
        out = m.forward_features(to_dev(case['data']), c0.to(DEV), f0.to(DEV), c1.to(DEV), f1.to(DEV))
    assert out['b_ids'].numel() == 0 and out['mkpts0_c'].shape == (0, 2)
    assert out['fine_matrix'].shape == (0, 25, 25)
    assert out['mkpts0_f'].shape == (0, 2) and out['mkpts1_f'].shape == (0, 2) and out['mconf'].numel() == 0
    assert out['conf_matrix'].shape == (2, 80, 80) and out['dect_conf_matrix'].shape == (2, 80, 80)
    # no model for any sample -> GeoModule leaves the position-encoded features untouched by the cross layers
    assert [int(v) for v in out['_geo_dev']['valid']] == [0, 0]


@pytest.mark.parametrize('name', ['g10b_e2e_planted_n2', 'g10c_e2e_planted_unequal', 'g10d_e2e_planted_masked'])
def test_fp16_mode_ids_bit_exact_vs_storage_oracle(golden, name):
    """The golden cases (N = 2; unequal shapes; padding masks + per-image scales + forced match) in the benched fp16 mode."""
    case = GI.g10_cases()[name]
    out, ref = run_fp16(case)
    assert len(ref['b_ids']) > 20
    compare_with_storage_oracle(out, ref, case['coarse_thr'], name, fine_thr=case['fine_thr'], max_diff=0)      # always bit-exact so far
    assert out['mkpts0_f'].dtype == torch.float32 and out['conf_matrix'].dtype == torch.float32
    # and the fp32 REFERENCE run stays the sanity anchor: the same matches up to fp16 resolution
    G = golden(name)
    a = set(zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()))
    b = set(zip(G['out_b_ids'].tolist(), G['out_i_ids'].tolist(), G['out_j_ids'].tolist()))
    assert len(a & b) >= 0.9 * max(len(a), len(b)), (len(a), len(b), len(a & b))


# share of the reference's fp32 coarse matches (g11: 1209) the 16-bit modes reproduce at 640 x 640; measured on MI355X in round 6: fp16 1205 of
# 1209 common (0.9967; 1 only here, 4 only there), bf16 1199 (0.9917; 8 / 10).  Gates = measured minus a margin of about as many matches again.
OVERLAP_640 = {'fp16': 0.992, 'bf16': 0.982}


@pytest.mark.parametrize('precision', ['fp16', 'bf16'])
def test_640_16bit_mode_vs_storage_oracle(golden, precision):
    """BASELINE size (80x80 grids, L = S = 6400) in the fast modes: panel K1, fused encoder layers, flash self-attention,
    device RANSAC - coarse ids against the oracle's storage mode (fp16: the bench's mode; bf16: what BASELINE configs[1] / [3]
    name), differences confined to the decision band and to half its population."""
    G, case = golden('g11_e2e_640_digest'), GI.g11_inputs()
    out, ref = run_fp16(case, precision=precision)
    assert len(ref['b_ids']) > 1000
    compare_with_storage_oracle(out, ref, case['coarse_thr'], f'640 {precision}', fine_thr=case['fine_thr'], edge=EDGE[precision],
                                conf_tol=CONF_TOL[precision], max_diff={'fp16': 14, 'bf16': 30}[precision])     # measured 6-9 / 18-20
    # ... and against the REFERENCE's own fp32 run of this pair (g11: 1209 coarse matches): the share of its matches the 16-bit mode reproduces,
    # printed and gated at the measured value minus a margin (VERDICT r05 #2c; measured on MI355X, round 6: see OVERLAP_640)
    a = set(zip(out['i_ids'].tolist(), out['j_ids'].tolist()))
    b = set(zip(G['i_ids'].astype(np.int64).tolist(), G['j_ids'].astype(np.int64).tolist()))
    overlap = len(a & b) / max(len(a), len(b))
    print(f'640x640 {precision}: {len(a)} coarse matches, the reference (fp32) {len(b)}, common {len(a & b)} = {overlap:.4f} of the larger set; '
          f'only here {len(a - b)}, only there {len(b - a)}')
    assert overlap >= OVERLAP_640[precision], (precision, len(a), len(b), len(a & b))


@pytest.mark.parametrize('mode', ['nominal', 'dense', 'nominal_bf16'])
def test_bench_shape_batch8_vs_storage_oracle(mode):
    """The bench's own shape AND LOAD: ONE forward of N = 8 planted 640x640 pairs built by bench.planted_features - the maps
    `value` is quoted on (M ~ 2300 coarse matches and K ~ 1200 inlier cells per pair at the reference's thresholds) - in 16-bit
    storage (8-pair K1 launches, 16-image K9 launches, two-blocks-per-wave K4, K11 at tens of thousands of windows), with
    thresholds 0.2 / 0.1 (`nominal`: the headline workload, all eight batch slots; `nominal_bf16`: the same in bf16, slot 5) and
    0 / 0 (`dense`: K1's dense-candidate path, slots 0, 3 and 7), against the storage oracle run pair by pair (batch elements
    are independent: full_model.py:39-123 has no cross-sample term)."""
    import bench
    thr, fthr = (0.0, 0.0) if mode == 'dense' else (0.2, 0.1)
    precision = 'bf16' if mode == 'nominal_bf16' else 'fp16'
    st = {'fp16': torch.float16, 'bf16': torch.bfloat16}[precision]
    c0, f0, c1, f1 = bench.planted_features(8, 60000, 80)                 # fp32 on the host: the oracle's inputs
    data = {'image0': torch.zeros(8, 1, 640, 640), 'image1': torch.zeros(8, 1, 640, 640)}
    m = build(thr, fthr, precision)
    with torch.no_grad():
        out = m.forward_features(to_dev(data), c0.to(DEV).to(st), f0.to(DEV).to(st), c1.to(DEV).to(st), f1.to(DEV).to(st))
    assert sorted(set(out['b_ids'].tolist())) == list(range(8))
    assert [int(v) for v in out['_geo_dev']['valid']] == [1] * 8
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=thr, fine_thr=fthr)
    W = O.make_weights()
    ob, fb = out['b_ids'].cpu(), out['m_bids'].cpu()
    total = knife = band = 0
    slots = {'nominal': range(8), 'dense': (0, 3, 7), 'nominal_bf16': (5,)}[mode]
    for b in slots:
        one = {'image0': data['image0'][b:b + 1], 'image1': data['image1'][b:b + 1]}
        ref = O.geoformer_forward_storage(W, one, st, None, geo_cfg, RO.make_homography_fn(),
                                          ((c0[b:b + 1].contiguous(), f0[b:b + 1].contiguous()), (c1[b:b + 1].contiguous(), f1[b:b + 1].contiguous())))
        sel, fsel = ob == b, fb == b
        sub = {k: out[k][sel.to(out[k].device)] for k in ('b_ids', 'i_ids', 'j_ids', 'fine_matrix')}
        sub['b_ids'] = sub['b_ids'] * 0
        sub.update({k: out[k][fsel.to(out[k].device)] for k in ('mkpts0_f', 'mkpts1_f', 'mconf')})
        sub['m_bids'] = out['m_bids'][fsel.to(out['m_bids'].device)] * 0
        sub['conf_matrix'] = out['conf_matrix'][b:b + 1]
        total += len(ref['b_ids'])
        d, B = compare_with_storage_oracle(sub, ref, thr, f'bench-shape {mode} slot {b}', fine_thr=fthr, edge=EDGE[precision],
                                           conf_tol=CONF_TOL[precision],
                                           max_diff={'nominal': 12, 'dense': 0, 'nominal_bf16': 38}[mode])   # measured 2-7 / 0 / 25-26 per slot
        knife, band = knife + d, band + B
    print(f'bench-shape {mode}: {total} coarse matches over the checked slots ({total / len(slots):.0f} per pair), {knife} differences, '
          f'band population {band}')
    assert total > len(slots) * (2000 if thr > 0 else 1000)               # the benched load (M ~ 2300 per pair at 0.2 / 0.1)


@pytest.mark.parametrize('precision', ['fp32', 'fp16', 'bf16'])
def test_hpatches_shaped_unequal_pair(precision):
    """BASELINE configs[1] shape class (eval_configs/geoformer.yml:7-11, data_io.py:16-26: shorter side 480, both sides
    floored to x8): a 480x640 image against a 480x608 one, N = 1 - 60x80 and 60x76 coarse grids (L = 4800, S = 4560:
    neither a multiple of the 128-token tiles), every stage on unequal shapes."""
    feats = GI.planted_features(1, 60, 80, 60, 76, 1201)
    data = {'image0': torch.zeros(1, 1, 480, 640), 'image1': torch.zeros(1, 1, 480, 608)}
    case = {'coarse_thr': 0.2, 'fine_thr': 0.1}
    if precision != 'fp32':
        out, ref = run_fp16(case, feats, data, precision=precision)
        compare_with_storage_oracle(out, ref, 0.2, f'hpatches-shaped {precision}', edge=EDGE[precision], conf_tol=CONF_TOL[precision],
                                    max_diff={'fp16': 12, 'bf16': 36}[precision])                               # measured 7 / 23-25
    else:
        m = build(0.2, 0.1, 'fp32')
        (c0, f0), (c1, f1) = feats
        with torch.no_grad():
            out = m.forward_features(to_dev(data), c0.to(DEV), f0.to(DEV), c1.to(DEV), f1.to(DEV))
        geo_cfg = O.default_geo_config()
        ref = O.geoformer_forward(O.make_weights(), dict(data), None, geo_cfg, RO.make_homography_fn(), None, feats)
        for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids'):
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k].numpy())
        close(out['mkpts0_f'], ref['mkpts0_f'], 1e-6, 1e-5); close(out['mkpts1_f'], ref['mkpts1_f'], 1e-6, 1e-5)
        mconf_close_fp32(out['mconf'], ref['mconf'].numpy(), 'hpatches-shaped fp32')
        close(out['mconf'], ref['mconf'], 2e-3, 1e-6)
    assert len(ref['b_ids']) > 500 and int(out['_geo_dev']['valid'][0]) == 1
    assert tuple(out['conf_matrix'].shape) == (1, 4800, 4560)


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_megadepth_style_batch_inference(precision):
    """BASELINE configs[3] data contract through the HIP path at N = 2: padding masks, per-image scales on BOTH samples
    (the case the reference's own GeoModule cannot run: SURVEY App. A.8 - per-sample scale is what it evidently
    intends), 'dataset_name' present.  fp32: ids bit-exact against the oracle; fp16: against its storage mode."""
    feats = GI.planted_features(2, 8, 10, 8, 10, 1301)
    m0 = torch.ones(2, 8, 10, dtype=torch.bool); m0[0, 7:] = False; m0[1, :, 9:] = False
    m1 = torch.ones(2, 8, 10, dtype=torch.bool); m1[0, :, 8:] = False; m1[1, 6:] = False
    data = {'image0': torch.zeros(2, 1, 64, 80), 'image1': torch.zeros(2, 1, 64, 80), 'mask0': m0, 'mask1': m1,
            'scale0': torch.tensor([[1.5, 1.25], [1.0, 2.0]]), 'scale1': torch.tensor([[1.0, 1.75], [1.25, 1.25]]),
            'dataset_name': ['megadepth', 'megadepth']}
    case = {'coarse_thr': 0.2, 'fine_thr': 0.1}
    if precision == 'fp16':
        out, ref = run_fp16(case, feats, data)
        compare_with_storage_oracle(out, ref, 0.2, 'megadepth-style fp16', max_diff=0)                             # always bit-exact so far
    else:
        m = build(0.2, 0.1, 'fp32')
        (c0, f0), (c1, f1) = feats
        with torch.no_grad():
            out = m.forward_features(to_dev(data), c0.to(DEV), f0.to(DEV), c1.to(DEV), f1.to(DEV))
        ref = O.geoformer_forward(O.make_weights(), dict(data), None, O.default_geo_config(), RO.make_homography_fn(), None, feats)
        for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids'):
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k].numpy())
        for k in ('mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f'):
            close(out[k], ref[k], 1e-6, 1e-5)
    assert len(ref['b_ids']) > 40 and sorted(set(ref['b_ids'].tolist())) == [0, 1]
    assert [int(v) for v in out['_geo_dev']['valid']] == [1, 1]


@pytest.mark.parametrize('name', ['g10b_e2e_planted_n2', 'g10c_e2e_planted_unequal'])
def test_bf16_mode_vs_storage_oracle(golden, name):
    """BASELINE configs[1] / [3] name bf16: the same path in bfloat16 storage (v_mfma_f32_32x32x16_bf16, fp32 accumulation)
    against the oracle's storage mode with bfloat16 round trips, by the same band rule with the 8x wider edge of an 8-bit
    mantissa; the matches stay those of the reference's fp32 run up to that resolution.  (Full-size bf16 cases:
    test_640_16bit_mode_vs_storage_oracle[bf16], test_hpatches_shaped_unequal_pair[bf16],
    test_bench_shape_batch8_vs_storage_oracle[nominal_bf16].)"""
    case = GI.g10_cases()[name]
    out, ref = run_fp16(case, precision='bf16')
    assert len(ref['b_ids']) > 20
    compare_with_storage_oracle(out, ref, case['coarse_thr'], f'{name} bf16', fine_thr=case['fine_thr'], edge=EDGE['bf16'],
                                conf_tol=CONF_TOL['bf16'], max_diff=3)                                          # measured 0-1 of 60-119
    assert out['conf_matrix'].dtype == torch.float32 and out['_feat_dev']['geo_f0'].dtype == torch.bfloat16
    G = golden(name)
    a = set(zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()))
    g = set(zip(G['out_b_ids'].tolist(), G['out_i_ids'].tolist(), G['out_j_ids'].tolist()))
    assert len(a & g) >= 0.8 * max(len(a), len(g)), (len(a), len(g), len(a & g))


def test_graph_replay_equals_eager():
    """GeoFormer.enable_graphs(): the static part (backbone .. second coarse matching) replayed from a captured hipGraph
    against the eager matching path on the replay's own backbone features, bit for bit, also on a second, different input
    (static-buffer reuse).  The backbone's feature maps are compared BIT FOR BIT as well since round 4: no convolution of the
    16-bit backbone goes to MIOpen any more (K10 incl. its stride-2 form, the K3 engine's 1x1 forms, the stem kernel), so nothing
    can pick another algorithm under capture (rounds 2-3 allowed 5 % of the maximum here; ADVICE r02 / VERDICT r03 weak 11)."""
    from geoformer_amd import miopen
    miopen.use_shipped_find_db()
    m = build(0.0, 0.0, 'fp16')
    pairs = [[t.to(DEV) for t in GI.textured_pair(128, 160, 900 + k)] for k in range(2)]
    keys = ('b_ids', 'i_ids', 'j_ids', 'mkpts0_f', 'mkpts1_f', 'mconf', 'conf_matrix', 'dect_conf_matrix')
    with torch.no_grad():
        m.enable_graphs()
        graphed = []
        for i0, i1 in pairs:
            out = m({'image0': i0, 'image1': i1})
            graphed.append(({k: out[k].clone() for k in keys}, tuple(t.clone() for t in out['_backbone_feats'])))
        assert len(m._graphs) == 1
        m.enable_graphs(False)
        for (i0, i1), (got, feats) in zip(pairs, graphed):
            ref = m.forward_features({'image0': i0, 'image1': i1}, *feats)
            assert len(ref['b_ids']) > 5          # (a sanity bound only: ~10-12 matches on this small pair)
            for k in keys:
                assert torch.equal(got[k], ref[k]), k
            c_eager, f_eager = m._backbone(torch.cat([i0, i1], 0))
            for name, fe, fg in (('feat_c0', c_eager[:1], feats[0]), ('feat_f0', f_eager[:1], feats[1]),
                                 ('feat_c1', c_eager[1:], feats[2]), ('feat_f1', f_eager[1:], feats[3])):
                assert fe.shape == fg.shape and torch.equal(fe, fg), name
    assert not torch.equal(graphed[0][0]['conf_matrix'], graphed[1][0]['conf_matrix'])


def test_graphs_follow_weight_reload_and_threshold_change():
    """captured hipGraphs read the packed weight caches by address and have the thresholds baked in as
    kernel arguments.  enable_graphs -> forward -> load_state_dict(other weights) -> forward must equal the eager path
    on the NEW weights (the captures are dropped), a changed coarse threshold must take effect (it is part of the graph
    key), and the small outputs of a graphed forward must survive the next forward (they are copies)."""
    from geoformer_amd import miopen
    miopen.use_shipped_find_db()
    m = build(0.0, 0.0, 'fp16')
    i0, i1 = [t.to(DEV) for t in GI.textured_pair(128, 160, 950)]
    j0, j1 = [t.to(DEV) for t in GI.textured_pair(128, 160, 951)]
    keys = ('b_ids', 'i_ids', 'j_ids', 'mkpts0_f', 'mkpts1_f', 'mconf', 'conf_matrix')
    with torch.no_grad():
        m.enable_graphs()
        first = m({'image0': i0, 'image1': i1})
        held = {k: first[k].clone() for k in keys if k != 'conf_matrix'}
        m({'image0': j0, 'image1': j1})                                   # same shape: replays into the same static buffers
        for k in held:
            assert torch.equal(first[k], held[k]), k                      # copies, not views of the capacity arrays
        assert 'conf_matrix' in first['_aliases_static_buffers']
        m.load_state_dict(O.make_weights(gain=0.9))
        m.to(DEV)
        assert m._graphs == {}
        got = m({'image0': i0, 'image1': i1})
        got = ({k: got[k].clone() for k in keys}, tuple(t.clone() for t in got['_backbone_feats']))
        m.coarse_matching.thr = 0.02
        thr_run = m({'image0': i0, 'image1': i1})
        n_thr = len(thr_run['b_ids'])
        assert len(m._graphs) == 2
        m.enable_graphs(False)
        ref = m.forward_features({'image0': i0, 'image1': i1}, *thr_run['_backbone_feats'])
        assert len(ref['b_ids']) == n_thr
        m.coarse_matching.thr = 0.0
        ref = m.forward_features({'image0': i0, 'image1': i1}, *got[1])
        for k in keys:
            assert torch.equal(got[0][k], ref[k]), k
    assert len(got[0]['b_ids']) > 0
