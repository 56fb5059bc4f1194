"""CPU unit test of tests/parity_band.py: the decision band of get_coarse_match (coarse_matching.py:161-178) on hand-built
confidence matrices - which entries count, with which flip distance."""
import torch

from parity_band import decision_band, knife_bound


def test_decision_band_on_a_hand_built_matrix():
    c = torch.full((1, 6, 6), 0.01)
    c[0, 0, 0] = 0.90                      # a safe match: far from every boundary
    c[0, 1, 1] = 0.203                     # a match 1.5 % above the threshold
    c[0, 2, 2] = 0.196                     # a non-match 2 % below the threshold, mutual maximum otherwise
    c[0, 3, 3] = 0.50; c[0, 3, 4] = 0.495  # a match whose row runner-up is 1 % behind ...
    c[0, 5, 5] = 0.15                      # a mutual maximum 25 % below the threshold: not in the band
    band = decision_band(c, 0.2, 0.03)
    assert set(band) == {(0, 1, 1), (0, 2, 2), (0, 3, 3), (0, 3, 4)}
    assert abs(band[(0, 1, 1)] - 0.003 / 0.203) < 1e-6
    assert abs(band[(0, 2, 2)] - 0.004 / 0.2) < 1e-6
    assert abs(band[(0, 3, 3)] - 0.005 / 0.5) < 1e-6          # smallest slack of the match: the row runner-up
    assert abs(band[(0, 3, 4)] - 0.005 / 0.5) < 1e-6          # the runner-up: 1 % short of taking the row (its column it owns)
    # a non-match that fails TWO conditions needs BOTH repaired: 1 % behind in its row but 10 % behind in its column -> out
    c[0, 4, 4] = 0.60
    band = decision_band(c, 0.2, 0.03)
    assert (0, 3, 4) not in band and (0, 3, 3) in band and (0, 4, 4) not in band
    # thr = 0 (the dense-candidate runs): no threshold condition
    band0 = decision_band(c, 0.0, 0.03)
    assert (0, 1, 1) not in band0 and (0, 2, 2) not in band0 and (0, 3, 3) in band0
    assert [knife_bound(b) for b in (0, 1, 2, 3, 10)] == [0, 1, 1, 2, 5]


def test_decision_band_counts_scale_with_the_edge():
    g = torch.Generator().manual_seed(3)
    c = torch.softmax(torch.randn(2, 64, 64, generator=g) * 3, 2) * torch.softmax(torch.randn(2, 64, 64, generator=g) * 3, 1)
    b1, b2 = decision_band(c, 0.0, 0.03), decision_band(c, 0.0, 0.24)
    assert set(b1) <= set(b2) and len(b2) > len(b1)
    assert all(0 <= d < 0.03 for d in b1.values())
