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


def knife_edge_margin(conf, b, i, j, thr):
    """Relative distance of conf[b,i,j] to the nearest decision boundary of get_coarse_match (coarse_matching.py:161-178):
    the threshold, the best other entry of its row, the best other entry of its column."""
    c = float(conf[b, i, j])
    row, col = conf[b, i].clone(), conf[b, :, j].clone()
    row[j], col[i] = -1, -1
    gaps = [abs(c - float(row.max())), abs(c - float(col.max()))]
    if thr > 0:
        gaps.append(abs(c - thr))
    return min(gaps) / max(c, float(row.max()), float(col.max()), thr, 1e-30)


def compare_with_storage_oracle(out, ref, thr, what, max_knife=3):
    """Coarse ids bit-exact, or: every match present on one side only sits on a decision boundary of the oracle's own
    confidence matrix (margin < KNIFE_EDGE) and there are at most `max_knife` of them.  Returns their count."""
    a = set(zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()))
    r = set(zip(ref['b_ids'].tolist(), ref['i_ids'].tolist(), ref['j_ids'].tolist()))
    diff = sorted(a ^ r)
    assert len(diff) <= max_knife, (what, len(a), len(r), len(diff))
    for (b, i, j) in diff:
        mg = knife_edge_margin(ref['conf_matrix'], b, i, j, thr)
        assert mg < KNIFE_EDGE, (what, (b, i, j), mg)
    print(f'{what}: {len(r)} coarse matches, {len(diff)} knife-edge differences')
    if not diff:
        for k in ('b_ids', 'i_ids', 'j_ids'):
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k].numpy())      # same order too
        # fine level: a coarse match is dropped there when its best 25x25 entry is below fine_thr - the same kind of
        # boundary; counts agree to 1 %, and when they agree the fine keypoints (integer window offsets around exact
        # coarse positions: a flipped arg-max moves one by >= 1 px) are identical for >= 97 % of the matches
        nf, rf = len(out['mkpts0_f']), len(ref['mkpts0_f'])
        assert abs(nf - rf) <= max(1, 0.01 * rf), (what, nf, rf)
        if nf == rf:
            np.testing.assert_array_equal(out['m_bids'].cpu().numpy(), ref['m_bids'].numpy())
            same = (out['mkpts0_f'].cpu() - ref['mkpts0_f']).abs().max(1)[0] < 1e-3
            assert same.float().mean() > 0.97, (what, float(same.float().mean()))
    # the confidence matrix: 14 layers of fp16 round-off noise feed an exponential with 1/temperature = 10
    oc, rc = out['conf_matrix'].float().cpu(), ref['conf_matrix']
    big = rc > 1e-3
    rel = ((oc - rc).abs() / rc.clamp_min(1e-12))[big]
    assert float(rel.max()) < 0.3 and float(rel.mean()) < 2e-2, (what, float(rel.max()), float(rel.mean()))
    return len(diff)


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
    compare_with_storage_oracle(out, ref, case['coarse_thr'], name, max_knife=0)
    assert out['mkpts0_f'].dtype == torch.float32 and out['conf_matrix'].dtype == torch.float32
    # and the fp32 REFERENCE run stays the sanity anchor: the same matches up to fp16 resolution
    G = golden(name)
    a = set(zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()))
    b = set(zip(G['out_b_ids'].tolist(), G['out_i_ids'].tolist(), G['out_j_ids'].tolist()))
    assert len(a & b) >= 0.9 * max(len(a), len(b)), (len(a), len(b), len(a & b))


def test_640_fp16_mode_vs_storage_oracle(golden):
    """BASELINE size (80x80 grids, L = S = 6400) in the fast mode: panel K1, fused encoder layers, flash self-attention,
    device RANSAC - coarse ids against the oracle's storage mode, knife-edge matches counted."""
    G, case = golden('g11_e2e_640_digest'), GI.g11_inputs()
    out, ref = run_fp16(case)
    assert len(ref['b_ids']) > 1000
    n_knife = compare_with_storage_oracle(out, ref, case['coarse_thr'], '640', max_knife=18)      # <= 1.5 % of ~1200
    print(f'640 fp16: {len(ref["b_ids"])} coarse matches, {n_knife} knife-edge differences')
    a = set(zip(out['i_ids'].tolist(), out['j_ids'].tolist()))
    b = set(zip(G['i_ids'].astype(np.int64).tolist(), G['j_ids'].astype(np.int64).tolist()))
    assert len(a & b) >= 0.95 * max(len(a), len(b)), (len(a), len(b), len(a & b))     # vs the reference's own fp32 run


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_hpatches_shaped_unequal_pair(precision):
    """BASELINE configs[1] shape class (eval_configs/geoformer.yml:7-11, data_io.py:16-26: shorter side 480, both sides
    floored to x8): a 480x640 image against a 480x608 one, N = 1 - 60x80 and 60x76 coarse grids (L = 4800, S = 4560:
    neither a multiple of the 128-token tiles), every stage on unequal shapes."""
    feats = GI.planted_features(1, 60, 80, 60, 76, 1201)
    data = {'image0': torch.zeros(1, 1, 480, 640), 'image1': torch.zeros(1, 1, 480, 608)}
    case = {'coarse_thr': 0.2, 'fine_thr': 0.1}
    if precision == 'fp16':
        out, ref = run_fp16(case, feats, data)
        compare_with_storage_oracle(out, ref, 0.2, 'hpatches-shaped fp16', max_knife=16)       # <= 1.5 % of ~1100
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
        compare_with_storage_oracle(out, ref, 0.2, 'megadepth-style fp16', max_knife=0)
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
    against the oracle's storage mode with bfloat16 round trips.  bf16 keeps 8 significant bits (fp16: 11), so its noise
    floor - and with it the band of knife-edge matches - is 8x wider; everything outside that band must agree, and the
    matches stay those of the reference's fp32 run up to that resolution."""
    case = GI.g10_cases()[name]
    out, ref = run_fp16(case, precision='bf16')
    a = set(zip(out['b_ids'].tolist(), out['i_ids'].tolist(), out['j_ids'].tolist()))
    r = set(zip(ref['b_ids'].tolist(), ref['i_ids'].tolist(), ref['j_ids'].tolist()))
    diff = sorted(a ^ r)
    assert len(r) > 20 and len(diff) <= max(2, 0.05 * len(r)), (name, len(a), len(r), len(diff))
    for (b, i, j) in diff:
        assert knife_edge_margin(ref['conf_matrix'], b, i, j, case['coarse_thr']) < 8 * KNIFE_EDGE, (name, (b, i, j))
    print(f'{name} bf16: {len(r)} coarse matches, {len(diff)} knife-edge differences')
    assert out['conf_matrix'].dtype == torch.float32 and out['_feat_dev']['geo_f0'].dtype == torch.bfloat16
    G = golden(name)
    g = set(zip(G['out_b_ids'].tolist(), G['out_i_ids'].tolist(), G['out_j_ids'].tolist()))
    assert len(a & g) >= 0.8 * max(len(a), len(g)), (len(a), len(g), len(a & g))


def test_graph_replay_equals_eager():
    """GeoFormer.enable_graphs(): the static part (backbone .. second coarse matching) replayed from a captured hipGraph
    against the eager matching path on the replay's own backbone features, bit for bit, also on a second, different input
    (static-buffer reuse).  (The backbone itself is compared to fp16 resolution only: MIOpen may pick another convolution
    algorithm under capture for shapes outside the shipped find-db - 1-ulp different feature maps.)"""
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
            assert len(ref['b_ids']) > 5          # (a sanity bound only: ~10-12 matches on this small pair, the count moves with the
                                                  # backbone's last bit, which MIOpen's algorithm choice under capture can change)
            for k in keys:
                assert torch.equal(got[k], ref[k]), k
            c0_eager = m._backbone(torch.cat([i0, i1], 0))[0][:1]
            assert float((c0_eager.float() - feats[0].float()).abs().max()) < 0.05 * float(c0_eager.float().abs().max())
    assert not torch.equal(graphed[0][0]['conf_matrix'], graphed[1][0]['conf_matrix'])
