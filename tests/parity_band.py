"""The data-derived bound of the 16-bit modes' end-to-end parity tests (tests/test_e2e_gpu.py)."""
import torch


# The bound on the number of differing matches is DERIVED FROM THE DATA of each case, not measured and padded.
# get_coarse_match (coarse_matching.py:161-178) accepts (i, j) iff conf > thr, conf is its row's maximum and conf is its column's
# maximum.  The FLIP DISTANCE of an entry of the oracle's own matrix is how far (relative) the matrix would have to move for that
# decision to come out the other way: for a match the SMALLEST slack of its three conditions, for a non-match the LARGEST violation
# among the conditions it fails (all of them have to be repaired).  The BAND is the set of entries with flip distance < edge;
# B = |band| counts the decisions the storage mode's noise floor can touch.  A correct kernel may differ from the oracle only inside
# the band, and since a band member's noise is as likely to push it away from its boundary as across it, at most about half of the
# band may flip: |diff| <= ceil(B / 2).  Both numbers are printed; DESIGN.md section 4 quotes |diff| / B per case.
def decision_band(conf, thr, edge):
    """{(b, i, j): flip distance} for every entry of conf [N, L, S] (the oracle's matrix) whose flip distance is < edge."""
    band = {}
    for b in range(conf.shape[0]):
        c = conf[b].float()
        r2, c2 = c.topk(2, dim=1), c.topk(2, dim=0)
        rmax, rsec, rarg = r2.values[:, 0], r2.values[:, 1], r2.indices[:, 0]
        cmax, csec, carg = c2.values[0], c2.values[1], c2.indices[0]
        # only entries within `edge` of BOTH maxima (and of the threshold) can have a flip distance < edge
        cand = (c >= (1 - edge) * rmax[:, None]) & (c >= (1 - edge) * cmax[None, :])
        if thr > 0:
            cand &= c >= (1 - edge) * thr
        i, j = torch.where(cand)
        v = c[i, j]
        ro = torch.where(rarg[i] == j, rsec[i], rmax[i])                # best OTHER entry of the row / of the column
        co = torch.where(carg[j] == i, csec[j], cmax[j])
        slack = [(v - ro) / torch.maximum(v, ro).clamp_min(1e-30), (v - co) / torch.maximum(v, co).clamp_min(1e-30)]
        if thr > 0:
            slack.append((v - thr) / torch.maximum(v, torch.full_like(v, thr)))
        sl = torch.stack(slack)                                         # [2 or 3, n]: > 0 = condition holds
        is_match = (sl > 0).all(0)
        viol = (-sl).clamp_min(0).max(0).values                         # non-match: every failed condition must be repaired
        dist = torch.where(is_match, sl.min(0).values, viol)
        for ii, jj, d in zip(i[dist < edge].tolist(), j[dist < edge].tolist(), dist[dist < edge].tolist()):
            band[(b, ii, jj)] = d
    return band


def knife_bound(B):
    return (B + 1) // 2
