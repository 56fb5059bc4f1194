"""Caller-side wrappers on the GPU: image loading/resizing, match_pairs return
conventions, device RANSAC on sub-pixel matches, HPatches-protocol evaluation loop on a synthetic
two-sequence dataset.

The uint8 gray conversion and resize (restatements of OpenCV's published fixed-point paths) are pinned by hand-derived
vectors in tests/test_matcher_cpu.py; here: everything downstream of the resized tensors, and the evaluation loop's
arithmetic on a sequence with planted homographies whose corner errors and AUC are computed by hand."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_seq(root, name, H, seed, w=200, h=168):
    from PIL import Image
    rng = np.random.default_rng(seed)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    for k in range(1, 7):
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, f'{k}.ppm'))
        if k > 1:
            np.savetxt(os.path.join(d, f'H_1_{k}'), H)


def test_estimate_homography_subpixel():
    from geoformer_amd import matcher as MT
    rng = np.random.default_rng(0)
    H = np.array([[1.05, 0.02, -7.3], [-0.04, 0.97, 11.2], [1e-4, -2e-4, 1.0]])
    p = rng.uniform(0, 480, (600, 2))
    q = np.c_[p, np.ones(600)] @ H.T
    q = q[:, :2] / q[:, 2:] + rng.normal(0, 0.4, (600, 2))
    q[:150] = rng.uniform(0, 480, (150, 2))                       # 25 % outliers
    Hp, inl = MT.estimate_homography(np.c_[p, q], 3.0)
    assert Hp is not None and inl[150:].mean() > 0.95 and inl[:150].mean() < 0.1
    assert MT.corner_error(Hp, H, 480, 480) < 1.0
    Hn, _ = MT.estimate_homography(np.c_[p[:3], q[:3]], 3.0)
    assert Hn is None                                             # fewer than 4 matches


def test_matcher_and_hpatches_protocol(tmp_path):
    from geoformer_amd import matcher as MT
    from geoformer_amd.weights import deterministic_init_
    root = str(tmp_path)
    _write_seq(root, 'i_synth', np.eye(3), 1)
    _write_seq(root, 'v_synth', np.array([[1., 0, 4], [0, 1, -3], [0, 0, 1]]), 2)
    m = MT.GeoFormerMatcher(imsize=160, match_threshold=0.0, no_match_upscale=True)
    deterministic_init_(m.model)
    m.model.fine_matching.thr = 0.0
    im1 = os.path.join(root, 'i_synth', '1.ppm')
    gray, scale = m.load_im(im1)
    assert gray.shape == (1, 1, 160, 184) and gray.is_cuda and 0.0 <= float(gray.min()) and float(gray.max()) <= 1.0
    assert scale == (200 / 184, 168 / 160)                        # shorter side -> 160, floored to x8 (data_io.py:16-26)
    res = m.match_pairs(im1, os.path.join(root, 'i_synth', '2.ppm'))
    matches, k1, k2, scores, upscale = res
    assert matches.shape[1] == 4 and len(k1) == len(k2) == len(scores) == len(matches) and upscale.shape == (4,)
    m2 = MT.GeoFormerMatcher(imsize=160, match_threshold=0.0, no_match_upscale=False)
    assert len(m2.match_pairs(im1, im1)) == 4
    out = MT.eval_hpatches(m, root, log=lambda s: None)
    assert len(out['auc_a']) == 4 and len(out['auc_i']) == 4 and len(out['auc_v']) == 4
    assert 0 <= out['failed'] <= 10 and out['match_time'] > 0


def test_command_line(tmp_path, capsys):
    """`python -m geoformer_amd.matcher match|hpatches` (inference.py / eval_Hpatches.py counterparts)."""
    from geoformer_amd import matcher as MT
    root = str(tmp_path / 'hp')
    _write_seq(root, 'v_cli', np.array([[1., 0, 2], [0, 1, 1], [0, 0, 1]]), 3)
    out = str(tmp_path / 'm.npz')
    MT.main(['match', os.path.join(root, 'v_cli', '1.ppm'), os.path.join(root, 'v_cli', '2.ppm'), '--imsize', '160',
             '--match-threshold', '0.0', '--out', out])
    z = np.load(out)
    assert z['matches'].shape[1] == 4 and len(z['scores']) == len(z['matches'])
    MT.main(['hpatches', root, '--imsize', '160', '--match-threshold', '0.0', '--max-seqs', '1'])
    assert 'auc_a' in capsys.readouterr().out


def test_match_pairs_against_oracle_on_the_same_resized_tensors(tmp_path):
    """match_pairs' numbers, not just its shapes: the matcher (fp32 mode, oracle weights) on two image files of different
    sizes against the oracle run on the SAME resized tensors (`load_im`'s output): keypoints of the common matches equal,
    `upscale` = the two resize ratios in (w0, h0, w1, h1) order, and the upscaled form = ratios x the resized form
    (eval_tool/immatch/modules/geoformer.py:40-75: a swapped scale[0..3] or a wrong keypoint scale fails here)."""
    import sys
    from PIL import Image
    import geoformer_oracle as O
    import golden_inputs as GI
    import ransac_oracle as RO
    from geoformer_amd import matcher as MT
    i0, i1 = GI.textured_pair(168, 200, 77)
    big = torch.nn.functional.interpolate(i1, size=(210, 280), mode='bilinear', align_corners=False)    # other size and aspect
    p0, p1 = str(tmp_path / 'a.png'), str(tmp_path / 'b.png')
    Image.fromarray((i0[0, 0] * 255).round().byte().numpy()).save(p0)
    Image.fromarray((big[0, 0] * 255).round().byte().numpy()).save(p1)
    W = O.make_weights()

    def make(noms):
        m = MT.GeoFormerMatcher(imsize=160, match_threshold=0.0, no_match_upscale=noms, precision='fp32')
        m.model.load_state_dict({k: v.clone() for k, v in W.items()})
        m.model.fine_matching.thr = 0.0
        return m
    m = make(True)
    g0, s0 = m.load_im(p0)
    g1, s1 = m.load_im(p1)
    assert tuple(g0.shape) == (1, 1, 160, 184) and tuple(g1.shape) == (1, 1, 160, 208)         # shorter side -> 160, floored to x8
    assert s0 == (200 / 184, 168 / 160) and s1 == (280 / 208, 210 / 160)
    matches, k0, k1, scores, upscale = m.match_pairs(p0, p1)
    np.testing.assert_allclose(upscale, np.array(s0 + s1))
    cfg = dict(O.default_geo_config(), coarse_thr=0.0, fine_thr=0.0)
    ref = O.geoformer_forward(W, {'image0': g0.cpu(), 'image1': g1.cpu()}, None, cfg, RO.make_homography_fn())
    want = {tuple(np.round(r, 3)) for r in torch.cat([ref['mkpts0_f'], ref['mkpts1_f']], 1).numpy().tolist()}
    got = {tuple(np.round(r, 3)) for r in matches.tolist()}
    assert len(want) > 50 and len(got & want) >= 0.9 * max(len(got), len(want)), (len(got), len(want), len(got & want))
    np.testing.assert_array_equal(matches[:, :2], k0); np.testing.assert_array_equal(matches[:, 2:], k1)
    assert len(scores) == len(matches) and float(scores.min()) >= 0 and float(scores.max()) <= 1
    # the other return convention: keypoints scaled back to the original images
    up, u0, u1, sc = make(False).match_pairs(p0, p1)
    np.testing.assert_allclose(up, matches * np.array(s0 + s1)[None], rtol=1e-6)
    np.testing.assert_allclose(u0, k0 * np.array(s0)[None], rtol=1e-6); np.testing.assert_allclose(u1, k1 * np.array(s1)[None], rtol=1e-6)
    assert float(u0[:, 0].max()) <= 200 and float(u1[:, 0].max()) <= 280 and float(u1[:, 0].max()) > 208 * 0.9


def test_eval_hpatches_known_answer(tmp_path):
    """The loop's arithmetic (hpatches_helper.py:185-239) on a planted sequence: a stub matcher returns noise-free matches in
    RESIZED coordinates that follow the ground-truth homography moved into those coordinates, shifted by a known offset e_k
    per pair.  Then H_pred = T(e_k) H_gt' and, H_gt being affine, every warped corner is off by exactly |e_k|: corner errors
    (0, 0.5, 2, 4, 20) px.  By hand (errors sorted, recall 0.2 per pair, trapezoids up to the threshold, divided by it):
        AUC@1  = (0.15 + 0.4 * 0.5) / 1                                  = 0.35
        AUC@3  = (0.15 + 0.5 * 1.5 + 0.6 * 1) / 3                        = 0.5
        AUC@5  = (0.15 + 0.75 + 0.7 * 2 + 0.8 * 1) / 5                   = 0.62
        AUC@10 = (0.15 + 0.75 + 1.4 + 0.8 * 6) / 10                      = 0.71
    and 'correct' = the fractions <= 1, 3, 5, 10 px = 0.4, 0.6, 0.8, 0.8.  A GT homography moved with the scale entries swapped,
    or corners taken at the original instead of the resized size, moves these numbers."""
    from PIL import Image
    from geoformer_amd import matcher as MT
    root = str(tmp_path)
    d = os.path.join(root, 'v_planted')
    os.makedirs(d)
    w, h = 200, 160
    H_gt = {k: np.array([[1.0 + 0.02 * k, 0.03, 5.0 * k], [-0.02, 0.97, -3.0 * k], [0, 0, 1.0]]) for k in range(2, 7)}
    for k in range(1, 7):
        Image.fromarray(np.zeros((h, w, 3), np.uint8)).save(os.path.join(d, f'{k}.ppm'))
        if k > 1:
            np.savetxt(os.path.join(d, f'H_1_{k}'), H_gt[k])
    scale = np.array([1.25, 0.8, 2.0, 1.6])                       # (w1, h1, w2, h2) original / resized
    offs = {2: (0.0, 0.0), 3: (0.3, -0.4), 4: (-1.2, 1.6), 5: (2.4, 3.2), 6: (12.0, -16.0)}     # |e| = 0, 0.5, 2, 4, 20

    class Stub:
        device = 'cuda:0'

        def __call__(self, im1, im2):
            k = int(os.path.basename(im2).split('.')[0])
            Hr = np.linalg.inv(MT.scale_homography(scale[2], scale[3])) @ H_gt[k] @ MT.scale_homography(scale[0], scale[1])
            gx, gy = np.meshgrid(np.linspace(4, w / scale[0] - 4, 12), np.linspace(4, h / scale[1] - 4, 10))
            p = np.stack([gx.ravel(), gy.ravel()], 1)
            q = np.c_[p, np.ones(len(p))] @ Hr.T
            q = q[:, :2] / q[:, 2:] + np.array(offs[k])[None]
            m = np.c_[p, q]
            return m, p, q, np.ones(len(p)), scale
    out = MT.eval_hpatches(Stub(), root, ransac_thres=3, log=lambda s: None)
    assert out['pairs'] == 5 and out['failed'] == 0
    np.testing.assert_allclose(out['auc_a'], [0.35, 0.5, 0.62, 0.71], atol=2e-3)
    np.testing.assert_allclose(out['auc_v'], out['auc_a'])
    np.testing.assert_allclose(out['correct_a'], [0.4, 0.6, 0.8, 0.8])
    np.testing.assert_array_equal(out['auc_i'], np.zeros(4))
