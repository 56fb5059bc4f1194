"""Caller-side wrappers on the GPU: image loading/resizing, match_pairs return
conventions, device RANSAC on sub-pixel matches, HPatches-protocol evaluation loop on a synthetic
two-sequence dataset."""
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
