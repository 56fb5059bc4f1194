"""The build's own RANSAC (oracle/ransac_oracle.c): planted-homography recovery on synthetic point sets.
OpenCV parity is unpinned (see the C header); these are the statistical checks SURVEY §8c asks for."""
import numpy as np
import pytest

import ransac_oracle as R


def planted(n, n_out, H, seed, w=640, h=640):
    rng = np.random.default_rng(seed)
    p0 = np.stack([rng.integers(0, w // 8, n) * 8, rng.integers(0, h // 8, n) * 8], 1).astype(np.int64)
    q = np.c_[p0, np.ones(n)] @ H.T
    p1 = np.floor(q[:, :2] / q[:, 2:3] / 8).astype(np.int64) * 8          # quantised to coarse cells
    out = rng.choice(n, n_out, replace=False)
    p1[out] = np.stack([rng.integers(0, w // 8, n_out) * 8, rng.integers(0, h // 8, n_out) * 8], 1)
    truth = np.ones(n, bool); truth[out] = False
    return p0, p1, truth


@pytest.mark.parametrize('H', [np.array([[1., 0, 8], [0, 1, 8], [0, 0, 1]]),
                               np.array([[0.93, -0.21, 44.3], [0.18, 1.07, -9.6], [0, 0, 1]]),
                               np.array([[1.12, 0.08, -21.0], [-0.05, 0.9, 37.5], [2e-4, -1.3e-4, 1]])])
@pytest.mark.parametrize('outlier_frac', [0.0, 0.3, 0.6])
def test_planted_homography_recovered(H, outlier_frac):
    n = 400
    p0, p1, truth = planted(n, int(n * outlier_frac), H, seed=int(outlier_frac * 10) + 1)
    M, mask = R.find_homography(p0, p1)
    assert M is not None
    got = mask[:, 0].astype(bool)
    # every planted inlier is found (quantisation error < 8 px), few outliers slip in by chance
    assert (got & truth).sum() >= 0.9 * truth.sum()
    assert (got & ~truth).sum() <= 0.05 * max(1, (~truth).sum()) + 2
    # corner transfer error of the recovered model is within the cell quantisation
    corners = np.array([[0, 0, 1], [640, 0, 1], [0, 640, 1], [640, 640, 1.]])
    a = corners @ M.T; b = corners @ H.T
    err = np.linalg.norm(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:], axis=1)
    assert err.max() < 12.0


def test_gates():
    p0 = np.arange(16).reshape(8, 2) * 8
    M, mask = R.find_homography(p0, p0)              # 8 matches: RANSAC is not attempted (geo_module.py:46)
    assert M is None and mask.sum() == 0
    rng = np.random.default_rng(0)
    p0 = rng.integers(0, 80, (9, 2)) * 8
    M, mask = R.find_homography(p0, p0 + 8)
    assert M is not None and mask.sum() == 9
    np.testing.assert_allclose(M, [[1, 0, 8], [0, 1, 8], [0, 0, 1]], atol=1e-9)
    # degenerate: all points identical -> no model
    p = np.zeros((20, 2), np.int64)
    M, mask = R.find_homography(p, p)
    assert M is None


def test_deterministic_and_sample_dependent():
    p0, p1, _ = planted(300, 120, np.array([[1., 0, 8], [0, 1, 8], [0, 0, 1]]), 3)
    a = R.find_homography(p0, p1, sample=0)
    b = R.find_homography(p0, p1, sample=0)
    np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])


def test_lm_iters_zero_reproduces_the_pre_refinement_model(golden):
    """g17 was written by the C oracle of commit a02ec81 (rounds 1-2, before the Levenberg-Marquardt refinement;
    oracle/gen_ransac_golden.py builds it from the git history).  Today's oracle with lm_iters = 0 must return the same
    inlier mask bit for bit and the same M; with the default 10 LM steps the mask is still the best hypothesis' (unchanged)
    and M moves by less than the cell quantisation."""
    from gen_ransac_golden import point_sets
    G = golden('g17_ransac_lm0')
    for b, (p0, p1) in enumerate(point_sets()):
        M0, mask0 = R.find_homography(p0, p1, sample=b, lm_iters=0)
        n = int(G[f's{b}_n'])
        want_mask = np.unpackbits(G[f's{b}_mask'])[:n]
        assert (M0 is not None) == bool(G[f's{b}_valid']), b
        if M0 is None:
            continue
        np.testing.assert_array_equal(mask0[:, 0], want_mask)
        np.testing.assert_allclose(M0, G[f's{b}_M'], rtol=1e-12, atol=1e-12)
        M10, mask10 = R.find_homography(p0, p1, sample=b)
        np.testing.assert_array_equal(mask10[:, 0], want_mask)
        corners = np.array([[0, 0, 1], [640, 0, 1], [0, 640, 1], [640, 640, 1.]])
        a, c = corners @ M10.T, corners @ M0.T
        assert np.linalg.norm(a[:, :2] / a[:, 2:] - c[:, :2] / c[:, 2:], axis=1).max() < 8.0
