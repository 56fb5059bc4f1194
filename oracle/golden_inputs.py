"""Seeded input builders shared by oracle/gen_golden.py (which feeds them to the reference) and by
tests/ (which feed the same tensors to the oracle and to the HIP path).

TEST INFRASTRUCTURE.  Pure torch-CPU; imports nothing from the reference.  Every fixture stores a
digest of the inputs it was generated with, so a change in torch's CPU generator would be detected
(tests/test_oracle_golden.py::test_input_digests) instead of silently invalidating the fixtures.
"""
import hashlib

import numpy as np
import torch


def gen(seed):
    return torch.Generator().manual_seed(seed)


def digest(*tensors) -> np.ndarray:
    h = hashlib.sha256()
    for t in tensors:
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def planted_features(N, h0, w0, h1, w1, seed, sigma=0.5, noise=0.5):
    """Stand-in backbone outputs with planted correspondences: map1 = map0 shifted by one coarse
    cell (+noise).  Returns ((c0, f0), (c1, f1)); c* [N,256,h,w], f* [N,128,4h,4w]."""
    g = gen(seed)
    H, W = max(h0, h1) + 1, max(w0, w1) + 1
    big = torch.randn(N, 256, H, W, generator=g) * sigma
    c0 = big[:, :, :h0, :w0].contiguous()
    c1 = (big[:, :, 1:1 + h1, 1:1 + w1] + noise * torch.randn(N, 256, h1, w1, generator=g)).contiguous()
    bigf = torch.randn(N, 128, 4 * H, 4 * W, generator=g)
    f0 = bigf[:, :, :4 * h0, :4 * w0].contiguous()
    f1 = (bigf[:, :, 4:4 + 4 * h1, 4:4 + 4 * w1]
          + noise * torch.randn(N, 128, 4 * h1, 4 * w1, generator=g)).contiguous()
    return (c0, f0), (c1, f1)


def textured_pair(h, w, seed, noise=0.3):
    """image1 = image0 shifted by 8 px; low-frequency texture + pixel noise, in [0,1]."""
    g = gen(seed)
    base = torch.rand(1, 1, h // 8 + 2, w // 8 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h + 8, w + 8), mode='bicubic', align_corners=True)
    img = (img + noise * torch.rand(1, 1, h + 8, w + 8, generator=g)).clamp(0, 1)
    return img[:, :, :h, :w].contiguous(), img[:, :, 8:, 8:].contiguous()


# ------------------------------------------------------------------------------------------
def g1_inputs():
    return {'x': torch.randn(1, 256, 4, 5, generator=gen(11)),
            'sample_ys': np.array([0, 1, 7, 79, 128, 255, 200, 3]),
            'sample_xs': np.array([0, 2, 9, 79, 64, 255, 17, 250])}


def g2_inputs():
    g = gen(21)
    d = {'q': torch.randn(2, 24, 8, 32, generator=g), 'k': torch.randn(2, 40, 8, 32, generator=g),
         'v': torch.randn(2, 40, 8, 32, generator=g)}
    qm = torch.ones(2, 24, dtype=torch.bool); qm[1, 18:] = False
    km = torch.ones(2, 40, dtype=torch.bool); km[0, 33:] = False; km[1, 29:] = False
    d.update(q_mask=qm, kv_mask=km, qf=torch.randn(3, 25, 8, 16, generator=g),
             kf=torch.randn(3, 25, 8, 16, generator=g), vf=torch.randn(3, 25, 8, 16, generator=g))
    return d


def g3_inputs():
    g = gen(31)
    d = {'x': torch.randn(2, 24, 256, generator=g), 'src': torch.randn(2, 40, 256, generator=g)}
    m = g2_inputs()
    d.update(x_mask=m['q_mask'], src_mask=m['kv_mask'], xf=torch.randn(4, 25, 128, generator=g),
             sf=torch.randn(4, 25, 128, generator=g), f0=torch.randn(2, 30, 256, generator=g),
             f1=torch.randn(2, 35, 256, generator=g))
    m0 = torch.ones(2, 30, dtype=torch.bool); m0[1, 24:] = False
    m1 = torch.ones(2, 35, dtype=torch.bool); m1[1, 28:] = False
    d.update(m0=m0, m1=m1)
    return d


def g4_inputs():
    g = gen(41)
    d = {'x_self': torch.randn(1, 40, 256, generator=g), 'src_self': torch.randn(1, 13, 256, generator=g),
         'x_cross': torch.randn(40, 1, 256, generator=g), 'src_cross': torch.randn(40, 25, 256, generator=g)}
    kvm = torch.rand(40, 25, generator=g) > 0.3
    kvm[7] = False; kvm[8, 1:] = False
    d.update(kv_mask=kvm, qa=torch.randn(3, 9, 4, 64, generator=g), ka=torch.randn(3, 25, 4, 64, generator=g),
             va=torch.randn(3, 25, 4, 64, generator=g))
    kam = torch.rand(3, 25, generator=g) > 0.4; kam[2] = False
    d['kam'] = kam
    return d


def g5_inputs():
    """Coarse-matching cases on a 6x8 vs 7x9 grid; f1 holds noisy copies of 36 rows of f0 at
    permuted positions, so the confidence matrix is sharp but graded."""
    g = gen(51)
    L, S = 48, 63
    f0 = torch.randn(2, L, 256, generator=g) * 1.3
    perm = torch.stack([torch.randperm(S, generator=g) for _ in range(2)])
    f1 = torch.randn(2, S, 256, generator=g) * 1.3
    for b in range(2):
        f1[b, perm[b, :36]] = f0[b, :36] + 0.45 * torch.randn(36, 256, generator=g)
    m0 = torch.ones(2, 6, 8, dtype=torch.bool); m0[1, 5:] = False; m0[1, :, 6:] = False
    m1 = torch.ones(2, 7, 9, dtype=torch.bool); m1[0, :, 8:] = False; m1[1, 6:] = False
    sc0 = torch.tensor([[1.5, 1.25], [2.0, 1.0]]); sc1 = torch.tensor([[1.0, 1.75], [1.1, 0.9]])
    f0e = f0.clone(); f0e[1] = torch.randn(L, 256, generator=g) * 0.05   # sample 1 flat -> no match
    f0t = f0.clone(); f1t = f1.clone()
    f1t[0, 5] = f1t[0, perm[0, 3]]          # duplicated column -> exact tie inside row 3
    f0t[0, 40] = f0t[0, 7]                  # duplicated row    -> exact tie inside a column
    return {'hw0': (6, 8), 'hw1': (7, 9), 'thr': 0.2,
            'plain': dict(f0=f0, f1=f1), 'masked': dict(f0=f0, f1=f1, mask0=m0, mask1=m1, scale0=sc0, scale1=sc1),
            'forced': dict(f0=f0e, f1=f1, dataset_name='x'), 'ties': dict(f0=f0t, f1=f1t)}


G6_HOMOGRAPHIES = {
    'identity': np.eye(3),
    'shift': np.array([[1., 0, -8], [0, 1, -8], [0, 0, 1]]),
    'affine': np.array([[0.93, -0.21, 14.3], [0.18, 1.07, -9.6], [0, 0, 1]]),
    'persp': np.array([[1.12, 0.08, -21.0], [-0.05, 0.9, 17.5], [9e-4, -1.3e-3, 1]]),
}


def g6_inputs():
    return {'dims': (64, 80, 56, 72), 'fmap': torch.randn(1, 16, 7, 9, generator=gen(61)), 'H': G6_HOMOGRAPHIES}


def g7_inputs():
    """GeoModule inputs on a 6x8 grid, batch 2: sample 0 has 14 matches (11 follow the planted
    one-cell shift, 3 are outliers), sample 1 has 5 matches (<= 8: RANSAC is not attempted)."""
    h, w = 6, 8
    (c0, _), (c1, _) = planted_features(2, h, w, h, w, 72)
    g = gen(71)
    cells0 = torch.randperm((h - 1) * (w - 1), generator=g)[:14]
    y0, x0 = cells0 // (w - 1), cells0 % (w - 1)
    mk0 = torch.stack([x0, y0], 1).float() * 8
    mk1 = mk0 + 8
    mk1[3] = torch.tensor([0., 40.]); mk1[9] = torch.tensor([56., 0.]); mk1[12] = torch.tensor([40., 8.])
    mk0b = torch.tensor([[0., 0.], [8., 16.], [24., 8.], [56., 40.], [40., 32.]]); mk1b = mk0b.flip(0).contiguous()
    return {'h': h, 'w': w, 'c0': c0, 'c1': c1, 'mkpts0_c': torch.cat([mk0, mk0b]), 'mkpts1_c': torch.cat([mk1, mk1b]),
            'm_bids': torch.cat([torch.zeros(14, dtype=torch.long), torch.ones(5, dtype=torch.long)]),
            'H_shift': np.array([[1., 0, 8], [0, 1, 8], [0, 0, 1]]), 'H_persp': G6_HOMOGRAPHIES['persp']}


def g8_inputs():
    g = gen(81)
    return {'feat_f0': torch.randn(2, 128, 32, 40, generator=g), 'feat_f1': torch.randn(2, 128, 28, 36, generator=g),
            'feat_c0': torch.randn(2, 80, 256, generator=g), 'feat_c1': torch.randn(2, 63, 256, generator=g),
            'b_ids': torch.tensor([0, 0, 0, 1, 1, 1]), 'i_ids': torch.tensor([0, 9, 79, 3, 44, 70]),
            'j_ids': torch.tensor([62, 0, 31, 8, 9, 54]), 'hw0_f': (32, 40), 'hw0_c': (8, 10), 'hw1_c': (7, 9)}


def g9_inputs():
    g = gen(91)
    Mn = 10
    a = torch.randn(Mn, 25, 128, generator=g) * 1.2
    b = torch.randn(Mn, 25, 128, generator=g) * 1.2
    for m in range(Mn):
        tgt = int(torch.randint(0, 25, (1,), generator=g)); src = int(torch.randint(0, 25, (1,), generator=g))
        if m % 3:
            b[m, tgt] = a[m, src] * 2.0 + 0.2 * torch.randn(128, generator=g)
        else:         # every third match: unplanted and weak -> flat 25x25 confidence, fails fine_thr
            b[m] *= 0.2
    return {'f0': a, 'f1': b, 'b_ids': torch.tensor([0, 0, 0, 1, 1, 1, 1, 1, 1, 1]),
            'mkpts0_c': torch.randint(0, 10, (Mn, 2), generator=g).float() * 8,
            'mkpts1_c': torch.randint(0, 9, (Mn, 2), generator=g).float() * 8,
            'hw0_i': (64, 80), 'hw0_c': (8, 10), 'hw0_f': (32, 40),
            'scale0': torch.tensor([[1.5, 1.25], [2.0, 1.0]]), 'scale1': torch.tensor([[1.0, 1.75], [1.1, 0.9]]),
            'temperature': 0.1, 'thr': 0.1}


def g10_cases():
    """End-to-end cases.  'feats' None = real backbone on a textured pair."""
    cases = {}
    i0, i1 = textured_pair(64, 80, 101)
    cases['g10a_e2e_backbone'] = dict(data={'image0': i0, 'image1': i1}, feats=None, coarse_thr=0.0, fine_thr=0.0)
    cases['g10b_e2e_planted_n2'] = dict(
        data={'image0': torch.zeros(2, 1, 64, 80), 'image1': torch.zeros(2, 1, 64, 80)},
        feats=planted_features(2, 8, 10, 8, 10, 102), coarse_thr=0.2, fine_thr=0.1)
    cases['g10c_e2e_planted_unequal'] = dict(
        data={'image0': torch.zeros(1, 1, 64, 80), 'image1': torch.zeros(1, 1, 56, 72)},
        feats=planted_features(1, 8, 10, 7, 9, 103), coarse_thr=0.2, fine_thr=0.1)
    m0 = torch.ones(1, 8, 10, dtype=torch.bool); m0[:, 7:] = False
    m1 = torch.ones(1, 8, 10, dtype=torch.bool); m1[:, :, 8:] = False
    cases['g10d_e2e_planted_masked'] = dict(
        data={'image0': torch.zeros(1, 1, 64, 80), 'image1': torch.zeros(1, 1, 64, 80), 'mask0': m0, 'mask1': m1,
              'scale0': torch.tensor([[1.5, 1.25]]), 'scale1': torch.tensor([[1.0, 1.75]]), 'dataset_name': ['x']},
        feats=planted_features(1, 8, 10, 8, 10, 104), coarse_thr=0.2, fine_thr=0.1)
    return cases


def g11_inputs():
    return {'feats': planted_features(1, 80, 80, 80, 80, 111), 'coarse_thr': 0.2, 'fine_thr': 0.1,
            'data': {'image0': torch.zeros(1, 1, 640, 640), 'image1': torch.zeros(1, 1, 640, 640)}}


# ------------------------------------------------------------------------------------------ training (f3)
def _homography(seed, n, hw, amp=6.0):
    """Mild random homographies: identity + corner-scale perturbation (pixels)."""
    g = gen(seed)
    H, W = hw
    out = []
    for _ in range(n):
        m = torch.eye(3)
        m[:2, :2] += (torch.rand(2, 2, generator=g) - 0.5) * 0.08
        m[:2, 2] = (torch.rand(2, generator=g) - 0.5) * 2 * amp
        m[2, :2] = (torch.rand(2, generator=g) - 0.5) * 2e-4
        out.append(m)
    return torch.stack(out)


def g13_inputs():
    H01 = _homography(131, 2, (64, 96))
    base = {'image0': torch.zeros(2, 1, 64, 96), 'image1': torch.zeros(2, 1, 64, 96), 'H_0to1': H01,
            'H_1to0': torch.inverse(H01), 'dataset_name': ['oxford', 'oxford'], 'pair_names': ['a', 'b']}
    scaled = dict(base, scale0=torch.tensor([[1.0, 1.0], [1.5, 1.25]]), scale1=torch.tensor([[1.0, 1.0], [1.25, 1.5]]))
    g = gen(132)
    M = 7
    cells0 = torch.randint(0, 8 * 12, (M,), generator=g)
    mk0 = torch.stack([cells0 % 12, cells0 // 12], 1).float() * 8
    H1 = H01[:1]
    w = torch.cat([mk0, torch.ones(M, 1)], 1) @ H1[0].T
    mk1 = ((w[:, :2] / w[:, 2:]) / 8).round().clamp(min=0) * 8
    fine = {'image0': torch.zeros(1, 1, 64, 96), 'image1': torch.zeros(1, 1, 64, 96), 'H_0to1': H1, 'H_1to0': torch.inverse(H1),
            'mkpts0_c': mk0, 'mkpts1_c': mk1, 'b_ids': torch.zeros(M, dtype=torch.long), 'W': torch.tensor(5),
            'hw0_i': torch.tensor([64, 96]), 'hw0_c': torch.tensor([8, 12]), 'hw0_f': torch.tensor([32, 48]),
            'dataset_name': ['oxford']}
    return {'coarse': base, 'coarse_scaled': scaled, 'fine': fine}


def g14_inputs():
    g = gen(141)
    N, L, S, M = 2, 40, 48, 6
    conf = torch.rand(N, L, S, generator=g) ** 3
    dect = torch.rand(N, L, S, generator=g) ** 3
    gt = torch.zeros(N, L, S)
    for b in range(N):
        rows = torch.randperm(L, generator=g)[:11]
        cols = torch.randperm(S, generator=g)[:11]
        gt[b, rows, cols] = 1
    fine = torch.rand(M, 25, 25, generator=g) ** 2
    fgt = torch.zeros(M, 25, 25, dtype=torch.bool)
    fgt[torch.arange(4), torch.randint(0, 25, (4,), generator=g), torch.randint(0, 25, (4,), generator=g)] = True
    m0 = torch.ones(N, 5, 8, dtype=torch.bool); m0[1, 4:] = False
    m1 = torch.ones(N, 6, 8, dtype=torch.bool); m1[0, :, 6:] = False
    return {'conf_matrix': conf, 'dect_conf_matrix': dect, 'conf_matrix_gt': gt, 'fine_matrix': fine,
            'conf_matrix_fine_gt': fgt, 'mask0': m0, 'mask1': m1}


def g15_inputs():
    """One training pair: textured image and its 8-px shift (GT homography = that translation)."""
    i0, i1 = textured_pair(64, 80, 151)
    H01 = torch.tensor([[[1., 0., -8.], [0., 1., -8.], [0., 0., 1.]]])
    return {'coarse_thr': 0.0, 'fine_thr': 0.0,
            'data': {'image0': i0, 'image1': i1, 'H_0to1': H01, 'H_1to0': torch.inverse(H01), 'dataset_name': ['oxford'],
                     'pair_names': ['p']}}


def g16_inputs():
    """Depth + pose supervision (MegaDepth / ScanNet branch): two views of a slanted plane with a known relative pose."""
    g = gen(161)
    N, H, W = 2, 64, 96
    K = torch.tensor([[[80., 0., 48.], [0., 80., 32.], [0., 0., 1.]]]).repeat(N, 1, 1)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    depth0 = (4.0 + 0.01 * xs + 0.02 * ys)[None].repeat(N, 1, 1) + 0.05 * torch.rand(N, H, W, generator=g)
    depth0[:, :3, :5] = 0                                              # holes
    T = torch.eye(4)[None].repeat(N, 1, 1)
    ang = torch.tensor([0.03, -0.05])
    T[:, 0, 0] = torch.cos(ang); T[:, 0, 2] = torch.sin(ang); T[:, 2, 0] = -torch.sin(ang); T[:, 2, 2] = torch.cos(ang)
    T[:, :3, 3] = torch.tensor([[0.15, -0.05, 0.1], [-0.2, 0.03, -0.05]])
    Tinv = torch.inverse(T)
    depth1 = (4.1 + 0.012 * xs + 0.018 * ys)[None].repeat(N, 1, 1) + 0.05 * torch.rand(N, H, W, generator=g)
    kpts = torch.rand(N, 50, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    coarse = {'image0': torch.zeros(N, 1, H, W), 'image1': torch.zeros(N, 1, H, W), 'depth0': depth0, 'depth1': depth1,
              'T_0to1': T, 'T_1to0': Tinv, 'K0': K, 'K1': K.clone(), 'dataset_name': ['megadepth'] * N, 'pair_names': ['a', 'b']}
    M = 9
    cells0 = torch.randint(12, 8 * 12 - 12, (M,), generator=g)
    mk0 = torch.stack([cells0 % 12, cells0 // 12], 1).float() * 8
    mk1 = (mk0 + torch.tensor([8., 0.])).clamp(max=88)
    fine = {'image0': torch.zeros(1, 1, H, W), 'image1': torch.zeros(1, 1, H, W), 'depth0': depth0[:1], 'depth1': depth1[:1],
            'T_0to1': T[:1], 'T_1to0': Tinv[:1], 'K0': K[:1], 'K1': K[:1].clone(), 'mkpts0_c': mk0, 'mkpts1_c': mk1,
            'b_ids': torch.zeros(M, dtype=torch.long), 'W': torch.tensor(5), 'hw0_i': torch.tensor([H, W]),
            'hw0_c': torch.tensor([H // 8, W // 8]), 'hw0_f': torch.tensor([H // 2, W // 2]), 'dataset_name': ['megadepth']}
    return {'kpts': kpts, 'coarse': coarse, 'fine': fine}


# ------------------------------------------------------------------------------------------ outcome-level parity
def hpatches_like_homography(seed, k, w, h):
    """Ground-truth homography of pair `k` (1..5, as HPatches' H_1_2 .. H_1_6) of synthetic sequence `seed`: the four image corners move
    by up to a * k pixels (the four-point parametrisation, solved exactly), a = 6 for even seeds and 8 for odd ones - the strength
    grows along the sequence as HPatches' does.  (a = 2 was tried: a nearly pure sub-cell translation puts EVERY cell of image 1
    half-way between two cells of image 0 at k = 3, 4 - two equal candidates per cell, confidences <= 0.25 - and such pairs keep 8 - 20
    matches; one flipped match then moves the homography by 0.2 px.  No real pair looks like that; with a >= 4 every pair keeps >= 40.)"""
    import numpy as np
    rng = np.random.default_rng(1000 * seed + k)
    amp = (6.0 if seed % 2 == 0 else 8.0) * k
    src = np.array([[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]], dtype=np.float64)
    dst = src + rng.uniform(-amp, amp, (4, 2))
    A, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        A += [[x, y, 1, 0, 0, 0, -u * x, -u * y], [0, 0, 0, x, y, 1, -v * x, -v * y]]
        b += [u, v]
    hvec = np.linalg.solve(np.array(A), np.array(b))
    return np.append(hvec, 1.0).reshape(3, 3)


def _field_sample(lat, xy, spacing, margin):
    """Bilinear sample of the lattice field lat [C, Hl, Wl] (node (r, c) sits at pixel (c * spacing - margin, r * spacing - margin)) at
    pixel positions xy [h, w, 2] -> [C, h, w]."""
    C, Hl, Wl = lat.shape
    gx = (xy[..., 0] + margin) / spacing / (Wl - 1) * 2 - 1
    gy = (xy[..., 1] + margin) / spacing / (Hl - 1) * 2 - 1
    grid = torch.stack([gx, gy], -1)[None].float()
    return torch.nn.functional.grid_sample(lat[None], grid, mode='bilinear', padding_mode='border', align_corners=True)[0]


def hpatches_like_features(seed, k, h0=60, w0=80, h1=60, w1=76, noise=0.25, margin=128, sc=8, sf=2):
    """Planted feature maps of a synthetic HPatches-protocol pair (480x640 against 480x608 by default, data_io.py:16-26's shape
    class): ONE continuous random field per level, defined over image-0 pixel coordinates (a lattice of white noise every 8 px for
    the coarse level / every 2 px for the fine level, bilinearly interpolated); image 0's maps sample it at their own cell
    positions, image 1's maps at H^-1 of theirs, plus noise - the maps a perfect backbone would hand over for a planar scene.
    Returns ((c0, f0), (c1, f1)) [1, 256, h, w] / [1, 128, 4h, 4w] fp32 and H (3x3, image 0 -> image 1 pixels)."""
    import numpy as np
    g = gen(7000 + 10 * seed + k)
    H = hpatches_like_homography(seed, k, 8 * w0, 8 * h0)
    Hinv = torch.from_numpy(np.linalg.inv(H)).double()

    def positions(h, w, step, warp):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64) * step, torch.arange(w, dtype=torch.float64) * step, indexing='ij')
        p = torch.stack([xs, ys, torch.ones_like(xs)], -1)
        if warp:
            p = p @ Hinv.T
        return (p[..., :2] / p[..., 2:]).float()
    Wl, Hl = 8 * max(w0, w1) + 2 * margin, 8 * max(h0, h1) + 2 * margin
    latc = torch.randn(256, Hl // sc + 1, Wl // sc + 1, generator=g) * 0.5
    latf = torch.randn(128, Hl // sf + 1, Wl // sf + 1, generator=g)
    c0 = _field_sample(latc, positions(h0, w0, 8, False), sc, margin)
    c1 = _field_sample(latc, positions(h1, w1, 8, True), sc, margin) + noise * torch.randn(256, h1, w1, generator=g)
    f0 = _field_sample(latf, positions(4 * h0, 4 * w0, 2, False), sf, margin)
    f1 = _field_sample(latf, positions(4 * h1, 4 * w1, 2, True), sf, margin) + noise * torch.randn(128, 4 * h1, 4 * w1, generator=g)
    return (c0[None].contiguous(), f0[None].contiguous()), (c1[None].contiguous(), f1[None].contiguous()), H
