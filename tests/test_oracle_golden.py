"""Pins oracle/geoformer_oracle.py against golden vectors produced by the reference itself
(oracle/gen_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O
import golden_inputs as GI

torch.set_num_threads(4)
T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def exact(a, b):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_array_equal(a, np.asarray(b))


def tensors_of(d):
    out = []
    for v in d.values():
        if isinstance(v, torch.Tensor):
            out.append(v)
        elif isinstance(v, dict):
            out.extend(tensors_of(v))
        elif isinstance(v, (tuple, list)):
            out.extend(tensors_of({i: x for i, x in enumerate(v)}))
    return out


@pytest.fixture(scope='module')
def W():
    return O.make_weights()


def test_weight_schema_matches_reference_state_dict(W):
    assert len(W) == 253                               # 253 tensors
    nparam = sum(v.numel() for k, v in W.items() if 'running' not in k and 'num_batches' not in k)
    assert nparam == 14187504


def test_input_digests(golden):
    exact(GI.digest(GI.g1_inputs()['x']), golden('g1_position_encoding')['input_digest'])
    exact(GI.digest(*tensors_of(GI.g2_inputs())), golden('g2_linear_attention')['input_digest'])
    exact(GI.digest(*tensors_of(GI.g5_inputs())), golden('g5_coarse_matching')['input_digest'])
    exact(GI.digest(*tensors_of(GI.g9_inputs())), golden('g9_fine_matching')['input_digest'])
    for name, case in GI.g10_cases().items():
        ts = tensors_of(case['data']) + (tensors_of({'f': case['feats']}) if case['feats'] else [])
        exact(GI.digest(*ts), golden(name)['input_digest'])


def test_g1_position_encoding(golden):
    G, I = golden('g1_position_encoding'), GI.g1_inputs()
    for tag, fix in (('bug', False), ('fix', True)):
        tab = O.position_encoding_table(256, 256, 256, fix)
        close(tab[:, :4, :5], G[f'table_{tag}_4x5'], 1e-6, 1e-7)
        close(tab[:, I['sample_ys'], I['sample_xs']], G[f'table_{tag}_samples'], 1e-6, 1e-7)
        close(O.add_position_encoding(I['x'], fix), G[f'out_{tag}'], 1e-6, 1e-7)


def test_g2_linear_attention(golden):
    G, I = golden('g2_linear_attention'), GI.g2_inputs()
    close(O.linear_attention(I['q'], I['k'], I['v']), G['out_nomask'])
    close(O.linear_attention(I['q'], I['k'], I['v'], I['q_mask'], I['kv_mask']), G['out_mask'])
    close(O.linear_attention(I['qf'], I['kf'], I['vf']), G['out_fine'])


def test_g3_loftr_layer_and_schedule(golden, W):
    G, I = golden('g3_loftr_layer'), GI.g3_inputs()
    p = 'loftr_coarse.layers.0.'
    close(O.encoder_layer(W, p, I['x'], I['src'], 8, 'loftr'), G['out_cross'], 1e-4, 1e-5)
    close(O.encoder_layer(W, p, I['x'], I['src'], 8, 'loftr', I['x_mask'], I['src_mask']), G['out_cross_masked'], 1e-4, 1e-5)
    close(O.encoder_layer(W, p, I['x'], I['x'], 8, 'loftr'), G['out_self'], 1e-4, 1e-5)
    close(O.encoder_layer(W, 'loftr_fine.layers.1.', I['xf'], I['sf'], 8, 'loftr'), G['out_fine'], 1e-4, 1e-5)
    names = ['self', 'cross'] * 4
    a, b = O.local_feature_transformer(W, 'loftr_coarse.', names, 8, I['f0'][:1], I['f1'][:1])
    close(a, G['sched_f0'], 1e-4, 1e-4); close(b, G['sched_f1'], 1e-4, 1e-4)
    a, b = O.local_feature_transformer(W, 'loftr_coarse.', names, 8, I['f0'], I['f1'], I['m0'], I['m1'])
    close(a, G['sched_f0_masked'], 1e-4, 1e-4); close(b, G['sched_f1_masked'], 1e-4, 1e-4)


def test_g4_geo_layer(golden, W):
    G, I = golden('g4_geo_layer'), GI.g4_inputs()
    p = 'geo_module.des_transformer.layers.1.'
    close(O.encoder_layer(W, p, I['x_self'], I['src_self'], 4, 'geo'), G['out_self'], 1e-4, 1e-5)
    close(O.encoder_layer(W, p, I['x_cross'], I['src_cross'], 4, 'geo', None, I['kv_mask']), G['out_cross'], 1e-4, 1e-5)
    close(O.full_attention(I['qa'], I['ka'], I['va']), G['attn_nomask'])
    close(O.full_attention(I['qa'], I['ka'], I['va'], None, I['kam']), G['attn_mask'])


@pytest.mark.parametrize('tag', ['plain', 'masked', 'forced', 'ties'])
def test_g5_coarse_matching(golden, tag):
    G, I = golden('g5_coarse_matching'), GI.g5_inputs()
    c = I[tag]
    hw0, hw1 = I['hw0'], I['hw1']
    d = {'hw0_i': torch.tensor([hw0[0] * 8, hw0[1] * 8]), 'hw1_i': torch.tensor([hw1[0] * 8, hw1[1] * 8]),
         'hw0_c': torch.tensor(hw0), 'hw1_c': torch.tensor(hw1)}
    d.update({k: v for k, v in c.items() if k not in ('f0', 'f1')})
    m0 = c['mask0'].flatten(-2) if 'mask0' in c else None
    m1 = c['mask1'].flatten(-2) if 'mask1' in c else None
    conf = O.dual_softmax(c['f0'], c['f1'], 0.1, m0, m1)
    close(conf, G[f'{tag}_conf_matrix'], 1e-5, 1e-9)
    # extraction is pinned on the reference's own confidence matrix: index-exact
    out = O.coarse_match(T(G[f'{tag}_conf_matrix']), d, I['thr'])
    for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids', 'mkpts0_c', 'mkpts1_c', 'mconf'):
        exact(out[k], G[f'{tag}_{k}'])
    out2 = O.coarse_match(conf, d, I['thr'])            # and end-to-end on its own matrix
    for k in ('b_ids', 'i_ids', 'j_ids'):
        exact(out2[k], G[f'{tag}_{k}'])


def test_g6_window_geometry(golden):
    G, I = golden('g6_window_geometry'), GI.g6_inputs()
    H0, W0, H1, W1 = I['dims']
    grid = O.map_keypoints(H0, W0, 8)
    exact(grid, G['grid'])
    for tag, Hm in I['H'].items():
        wp = O.warp_points(grid, T(Hm).float())
        close(wp, G[f'{tag}_warped'], 1e-6, 1e-5)
        kps, mask = O.make_windows(T(G[f'{tag}_warped']), (H1, W1), 5, 8)
        exact(kps, G[f'{tag}_kps']); exact(mask, G[f'{tag}_mask'])
        exact(O.sample_windows(kps, I['fmap'][0], 8), G[f'{tag}_gather'])


def _g7_batch(I):
    h, w = I['h'], I['w']
    return {'image0': torch.zeros(2, 1, h * 8, w * 8), 'image1': torch.zeros(2, 1, h * 8, w * 8),
            'hw0_i': torch.tensor([h * 8, w * 8]), 'hw0_c': torch.tensor([h, w]),
            'mkpts0_c': I['mkpts0_c'], 'mkpts1_c': I['mkpts1_c'], 'm_bids': I['m_bids']}


@pytest.mark.parametrize('tag', ['shift', 'nohomo', 'persp'])
def test_g7_geo_module(golden, W, tag):
    G, I = golden('g7_geo_module'), GI.g7_inputs()

    def replay(a, b):
        return (G[f'{tag}_M'].copy() if G[f'{tag}_valid'] else None), G[f'{tag}_mask'].copy()
    o0, o1 = O.geo_module(W, I['c0'], I['c1'], _g7_batch(I), O.default_geo_config(), replay)
    sub = slice(None) if tag == 'shift' else slice(None, None, 4)
    close(o0[..., sub], G[f'{tag}_out0'], 1e-4, 1e-4); close(o1[..., sub], G[f'{tag}_out1'], 1e-4, 1e-4)


def test_g8_fine_preprocess(golden, W):
    G, I = golden('g8_fine_preprocess'), GI.g8_inputs()
    d = {'hw0_f': torch.tensor(I['hw0_f']), 'hw0_c': torch.tensor(I['hw0_c']), 'hw1_c': torch.tensor(I['hw1_c']),
         'b_ids': I['b_ids'], 'i_ids': I['i_ids'], 'j_ids': I['j_ids']}
    u0, u1 = O.fine_preprocess(W, I['feat_f0'], I['feat_f1'], I['feat_c0'], I['feat_c1'], d)
    close(u0, G['out0'], 1e-5, 1e-5); close(u1, G['out1'], 1e-5, 1e-5)
    d.update(b_ids=I['b_ids'][:0], i_ids=I['i_ids'][:0], j_ids=I['j_ids'][:0])
    e0, e1 = O.fine_preprocess(W, I['feat_f0'], I['feat_f1'], I['feat_c0'], I['feat_c1'], d)
    exact(np.array(e0.shape + e1.shape), G['empty_shape'])


@pytest.mark.parametrize('tag', ['plain', 'scaled'])
def test_g9_fine_matching(golden, tag):
    G, I = golden('g9_fine_matching'), GI.g9_inputs()
    d = {'hw0_i': torch.tensor(I['hw0_i']), 'hw0_c': torch.tensor(I['hw0_c']), 'hw0_f': torch.tensor(I['hw0_f']),
         'b_ids': I['b_ids'], 'mkpts0_c': I['mkpts0_c'], 'mkpts1_c': I['mkpts1_c']}
    if tag == 'scaled':
        d.update(scale0=I['scale0'], scale1=I['scale1'])
    out = O.fine_match(I['f0'], I['f1'], d, I['temperature'], I['thr'])
    close(out['fine_matrix'], G[f'{tag}_fine_matrix'], 1e-5, 1e-9)
    exact(out['m_bids'], G[f'{tag}_m_bids'])
    close(out['mkpts0_f'], G[f'{tag}_mkpts0_f'], 1e-6, 1e-5); close(out['mkpts1_f'], G[f'{tag}_mkpts1_f'], 1e-6, 1e-5)
    close(out['mconf'], G[f'{tag}_mconf'], 1e-5, 1e-8)
    assert len(out['mconf']) < I['f0'].shape[0]          # at least one match fails fine_thr


def replay_ransac(G):
    calls = iter(range(int(G['ransac_ncalls'])))

    def fn(a, b):
        i = next(calls)
        exact(a, G[f'ransac{i}_kp0']); exact(b, G[f'ransac{i}_kp1'])
        return (G[f'ransac{i}_M'].copy() if G[f'ransac{i}_valid'] else None), G[f'ransac{i}_mask'].copy()
    return fn


@pytest.mark.parametrize('name', list(GI.g10_cases().keys()))
def test_g10_end_to_end(golden, W, name):
    G, case = golden(name), GI.g10_cases()[name]
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=case['coarse_thr'], fine_thr=case['fine_thr'])
    rec = {}
    out = O.geoformer_forward(W, dict(case['data']), None, geo_cfg, replay_ransac(G), rec, case['feats'])
    for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids'):
        exact(out[k], G['out_' + k])
    for k in ('mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f'):
        close(out[k], G['out_' + k], 1e-6, 1e-5)
    close(out['mconf'], G['out_mconf'], 1e-4, 1e-6)
    close(out['conf_matrix'], G['out_conf_matrix'], 2e-3, 1e-7)
    close(out['dect_conf_matrix'], G['out_dect_conf_matrix'], 2e-3, 1e-7)
    close(out['fine_matrix'][:12], G['out_fine_matrix_head'], 2e-3, 1e-7)
    for k in ('loftr_f0', 'loftr_f1', 'geo_f0', 'geo_f1'):
        close(rec[k][..., ::4], G['mid_' + k], 1e-4, 1e-4)


def test_g11_end_to_end_640_digest(golden, W):
    G, case = golden('g11_e2e_640_digest'), GI.g11_inputs()
    exact(GI.digest(case['feats'][0][0][:, :, :4], case['feats'][1][1][:, :, :4]), G['input_digest'])
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=case['coarse_thr'], fine_thr=case['fine_thr'])
    n = int(G['ransac0_n'])
    mask = np.unpackbits(G['ransac0_mask'])[:n].astype(np.uint8)[:, None]

    def fn(a, b):
        assert len(a) == n
        return (G['ransac0_M'].copy() if G['ransac0_valid'] else None), mask
    torch.set_num_threads(8)
    out = O.geoformer_forward(W, dict(case['data']), None, geo_cfg, fn, None, case['feats'])
    assert len(out['b_ids']) == int(G['M']) and len(out['mkpts0_f']) == int(G['Mf'])
    exact(out['i_ids'], G['i_ids'].astype(np.int64)); exact(out['j_ids'], G['j_ids'].astype(np.int64))
    exact(GI.digest(out['b_ids'], out['i_ids'], out['j_ids']), G['coarse_ids_digest'])
    exact(GI.digest(out['mkpts0_f'], out['mkpts1_f']), G['fine_kpts_digest'])
    close(out['mconf'][:64], G['mconf_head'], 1e-3, 1e-6)
    close(out['conf_matrix'][0, :64].sum(-1), G['conf_rowsum_head'], 1e-3, 1e-6)


@pytest.mark.parametrize('name', ['g10b_e2e_planted_n2', 'g10c_e2e_planted_unequal', 'g10d_e2e_planted_masked'])
def test_storage_oracle_fp32_equals_goldens(golden, W, name):
    """geoformer_forward_storage is the checker of the benched 16-bit modes (tests/test_e2e_gpu.py); with st = float32 its
    round trips are identities and it must reproduce the REFERENCE's own run: this ties the builder-derived storage
    oracle (project-then-gather windows, flash self-attention tiles, row-group-bias FinePreprocess, fused layer order)
    to the golden vectors.  Ids bit-exact; mconf to 2e-5 (the re-associated sums)."""
    G, case = golden(name), GI.g10_cases()[name]
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=case['coarse_thr'], fine_thr=case['fine_thr'])
    out = O.geoformer_forward_storage(W, dict(case['data']), torch.float32, None, geo_cfg, replay_ransac(G), case['feats'])
    for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids'):
        exact(out[k], G['out_' + k])
    for k in ('mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f'):
        close(out[k], G['out_' + k], 1e-6, 1e-5)
    close(out['mconf'], G['out_mconf'], 1e-4, 2e-5)
    close(out['conf_matrix'], G['out_conf_matrix'], 2e-3, 1e-7)
    close(out['dect_conf_matrix'], G['out_dect_conf_matrix'], 2e-3, 1e-7)
    close(out['fine_matrix'][:12], G['out_fine_matrix_head'], 2e-3, 1e-7)


def test_storage_oracle_fp32_equals_g11_digest(golden, W):
    """The same at the BASELINE size (80x80 grids): coarse ids and fine keypoints bit-identical to the reference run."""
    G, case = golden('g11_e2e_640_digest'), GI.g11_inputs()
    geo_cfg = O.default_geo_config(); geo_cfg.update(coarse_thr=case['coarse_thr'], fine_thr=case['fine_thr'])
    n = int(G['ransac0_n'])
    mask = np.unpackbits(G['ransac0_mask'])[:n].astype(np.uint8)[:, None]
    torch.set_num_threads(8)
    out = O.geoformer_forward_storage(W, dict(case['data']), torch.float32, None, geo_cfg,
                                      lambda a, b: (G['ransac0_M'].copy(), mask), case['feats'])
    assert len(out['b_ids']) == int(G['M']) and len(out['mkpts0_f']) == int(G['Mf'])
    exact(out['i_ids'], G['i_ids'].astype(np.int64)); exact(out['j_ids'], G['j_ids'].astype(np.int64))
    exact(GI.digest(out['b_ids'], out['i_ids'], out['j_ids']), G['coarse_ids_digest'])
    exact(GI.digest(out['mkpts0_f'], out['mkpts1_f']), G['fine_kpts_digest'])
    close(out['mconf'][:64], G['mconf_head'], 1e-3, 1e-6)


def test_g12_eval_helpers(golden):
    """The product's evaluation helpers (pure numpy, run on CPU) against the reference's own."""
    from geoformer_amd import matcher as MT
    G = golden('g12_eval_helpers')
    close(MT.cal_error_auc(G['errors'], list(G['thresholds'])), G['auc'], 1e-12, 1e-12)
    close(MT.cal_error_auc([], list(G['thresholds'])), G['auc_empty'], 0, 0)
    close(MT.cal_reproj_dists(G['p1'], G['p2'], G['H']), G['reproj'], 1e-12, 1e-12)
    for w, h, imsize, df, f, wt, ht, sx, sy in G['resize']:
        got = MT.resize_im(int(w), int(h), imsize=int(imsize), dfactor=int(df), value_to_scale=min if f == 0 else max)
        assert got[0] == int(wt) and got[1] == int(ht) and got[2] == (sx, sy), (w, h, imsize, got)
