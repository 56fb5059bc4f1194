"""Ground-truth labels for the training step, torch ops under no_grad on the batch's device.

What the reference computes (model/loftr_src/loftr/utils/supervision.py:23-115 coarse, :270-387 fine;
model/loftr_src/loftr/utils/geometry.py:5-54 depth warp) is restated here around two small objects:

  _CellGrid    one image's coarse grid: pixel centres of its cells (per-sample scale, padded cells parked at the
               origin), and the conversion pixel -> flat cell index with an out-of-range sink at cell 0;
  _Projector   the ground-truth map between the two images of the batch: homographies (`H_0to1`, `H_1to0`) or
               depth + relative pose (`depth*`, `T_*`, `K*`).

`spvs_coarse` labels a coarse cell pair (i, j) positive when i projects into j and j projects back into i
(cell 0 excluded: it is where everything out of range lands).  `spvs_fine2` labels, per predicted coarse match,
the one pair of 5x5 window positions whose projected distance is the smallest, if it lies in (0, 3] pixels.

Deviation (documented in DESIGN.md): the fine labels use the homography - or the depth maps and pose - of each
match's own sample; the reference passes the whole [N,3,3] stack (or flattens all windows into one depth-warp
call), which only works for one pair per GPU.  Identical for N = 1.
"""
import torch


def _pixel_lattice(h, w, device):
    """[h*w, 2] float (x, y) of an h x w lattice in row-major order."""
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing='ij')
    return torch.stack([xs, ys], -1).reshape(h * w, 2)


def warp_points_batch(points, homographies):
    """points [B,l,2] through homographies [B,3,3] / [1,3,3] / [3,3] (utils/homography.py:86-105); a homogeneous
    coordinate that is exactly 0 is replaced by 1e-6."""
    B, l = points.shape[:2]
    Hm = homographies if homographies.dim() == 3 else homographies[None]
    if Hm.shape[0] != B:
        Hm = Hm.expand(B, 3, 3) if Hm.shape[0] == 1 else Hm.repeat(B, 1, 1)
    homog = torch.cat([points, points.new_ones(B, l, 1)], dim=-1)
    out = torch.bmm(Hm.to(homog.dtype), homog.transpose(1, 2)).transpose(1, 2)
    den = out[..., 2:].clone()
    den[den == 0] = 1e-6
    return out[..., :2] / den


def _inv3x3(m):
    """Inverse of [..., 3, 3] matrices in closed form (adjugate / determinant): on the GPU `Tensor.inverse()` of a batch of intrinsics goes
    through the batched LU of the solver library with a host synchronisation - 5.4 ms per call, ten calls per training step (round 6: 54 of
    the step's 190 ms of host time).  Same values to a few ulp (the intrinsics are upper triangular and well conditioned)."""
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    A, B, C = e * i - f * h, c * h - b * i, b * f - c * e
    D, E, F = f * g - d * i, a * i - c * g, c * d - a * f
    G, H, I = d * h - e * g, b * g - a * h, a * e - b * d
    det = a * A + b * D + c * G
    return torch.stack([torch.stack([A, B, C], -1), torch.stack([D, E, F], -1), torch.stack([G, H, I], -1)], -2) / det[..., None, None]


@torch.no_grad()
def warp_kpts(kpts0, depth0, depth1, T_0to1, K0, K1):
    """Depth + pose projection of kpts0 [N,l,2] into image 1 (geometry.py:5-54).  Returns (valid [N,l], warped):
    valid = source depth known & projection inside image 1 & depths agree within 20 %."""
    n = kpts0.shape[0]
    rows = torch.arange(n, device=kpts0.device)[:, None]
    px = kpts0.round().long()
    z0 = depth0[rows, px[..., 1].clamp(0, depth0.shape[1] - 1), px[..., 0].clamp(0, depth0.shape[2] - 1)]
    rays = torch.cat([kpts0, torch.ones_like(kpts0[..., :1])], dim=-1) * z0[..., None]
    k0_inv = _inv3x3(K0) if K0.is_cuda else K0.inverse()          # (the CPU path keeps the reference's call: geometry.py:27)
    in_cam1 = T_0to1[:, :3, :3] @ (k0_inv @ rays.transpose(2, 1)) + T_0to1[:, :3, 3:4]          # (slices, not `[3]`: a list index is a host tensor copied to the device - a synchronisation)
    z_proj = in_cam1[:, 2, :]
    img = (K1 @ in_cam1).transpose(2, 1)
    warped = img[..., :2] / (img[..., 2:3] + 1e-4)
    h1, w1 = depth1.shape[1:3]
    inside = (warped[..., 0] > 0) * (warped[..., 0] < w1 - 1) * (warped[..., 1] > 0) * (warped[..., 1] < h1 - 1)
    tgt = warped.long()
    tgt = tgt.masked_fill(~inside[..., None], 0)              # (masked_fill, not `tgt[~inside] = 0`: boolean-mask assignment synchronises with the device)
    z1 = depth1[rows, tgt[..., 1], tgt[..., 0]]
    agree = ((z1 - z_proj) / z1).abs() < 0.2
    return (z0 != 0) * inside * agree, warped


class _Projector:
    def __init__(self, data):
        self.d = data
        self.by_depth = 'depth0' in data

    def forward(self, pts):        # image 0 -> image 1
        d = self.d
        if self.by_depth:
            return warp_kpts(pts, d['depth0'], d['depth1'], d['T_0to1'], d['K0'], d['K1'])
        return None, warp_points_batch(pts, d['H_0to1'])

    def backward(self, pts):       # image 1 -> image 0
        d = self.d
        if self.by_depth:
            return warp_kpts(pts, d['depth1'], d['depth0'], d['T_1to0'], d['K1'], d['K0'])
        return None, warp_points_batch(pts, d['H_1to0'])


class _CellGrid:
    def __init__(self, n, h, w, step, mask, device):
        self.h, self.w = h, w
        self.step = step                                   # pixels per cell: scalar, or [N,1,2] with per-image scale
        self.centres = step * _pixel_lattice(h, w, device)[None].repeat(n, 1, 1)
        if mask is not None:
            self.centres.masked_fill_(~mask.flatten(-2).bool()[..., None], 0)

    def cell_of(self, pts):
        """Flat index of the cell nearest to each pixel position; positions outside the grid -> 0."""
        c = (pts / self.step).round().long()
        flat = c[..., 0] + c[..., 1] * self.w
        outside = (c[..., 0] < 0) | (c[..., 0] >= self.w) | (c[..., 1] < 0) | (c[..., 1] >= self.h)
        flat.masked_fill_(outside, 0)
        return flat


@torch.no_grad()
def spvs_coarse(data, resolution=(8, 2)):
    """Writes conf_matrix_gt [N,L,S], spv_b_ids / spv_i_ids / spv_j_ids, spv_w_pt0_i, spv_pt1_i, spv_num_gt."""
    dev = data['image0'].device
    n, _, H0, W0 = data['image0'].shape
    _, _, H1, W1 = data['image1'].shape
    c = resolution[0]
    scaled = 'scale0' in data
    g0 = _CellGrid(n, H0 // c, W0 // c, c * data['scale0'][:, None] if scaled else c, data.get('mask0'), dev)
    g1 = _CellGrid(n, H1 // c, W1 // c, c * data['scale1'][:, None] if scaled else c, data.get('mask1'), dev)
    proj = _Projector(data)
    _, into1 = proj.forward(g0.centres)
    _, into0 = proj.backward(g1.centres)
    j_of_i = g1.cell_of(into1)                              # [N, L]: where cell i of image 0 lands in image 1
    i_of_j = g0.cell_of(into0)                              # [N, S]
    L = g0.h * g0.w
    mutual = torch.gather(i_of_j, 1, j_of_i) == torch.arange(L, device=dev)[None]
    mutual[:, 0] = False
    b_ids, i_ids = mutual.nonzero(as_tuple=True)
    j_ids = j_of_i[b_ids, i_ids]
    gt = torch.zeros(n, L, g1.h * g1.w, device=dev)
    gt[b_ids, i_ids, j_ids] = 1
    data['spv_num_gt'] = int(b_ids.numel())                # the real count: the stand-in below is not a label
    if b_ids.numel() == 0:                                  # one stand-in match keeps the fine level alive
        b_ids = i_ids = j_ids = torch.zeros(1, dtype=torch.long, device=dev)
    data.update(conf_matrix_gt=gt, spv_b_ids=b_ids, spv_i_ids=i_ids, spv_j_ids=j_ids, spv_w_pt0_i=into1,
                spv_pt1_i=g1.centres)


@torch.no_grad()
def spvs_fine2(data, resolution=(8, 2)):
    """Writes conf_matrix_fine_gt [M, W*W, W*W] (bool)."""
    dev = data['image0'].device
    W = int(data['W'])
    WW = W * W
    M = data['mkpts0_c'].shape[0]
    if M == 0:
        data['conf_matrix_fine_gt'] = torch.zeros(0, WW, WW, dtype=torch.bool, device=dev)
        return
    b = data['b_ids']
    offsets = _pixel_lattice(W, W, dev)[None].repeat(M, 1, 1) - W // 2          # window positions around a centre
    hi, hc, hf = data['hw0_i'][0], data['hw0_c'][0], data['hw0_f'][0]
    coarse_px, fine_per_coarse, fine_px = hi // hc, hf // hc, hi // hf

    def window_pixels(centres_c, key):
        """Input-image pixel positions of the W x W fine window around each coarse keypoint."""
        s = data[key][b] if key in data else None
        on_fine_grid = centres_c / (coarse_px * s if s is not None else coarse_px) * fine_per_coarse
        win = on_fine_grid[:, None].repeat(1, WW, 1) + offsets
        return win * ((fine_px * s)[:, None].repeat(1, WW, 1) if s is not None else fine_px)
    p0, p1 = window_pixels(data['mkpts0_c'], 'scale0'), window_pixels(data['mkpts1_c'], 'scale1')
    proj = _Projector(data)
    if proj.by_depth:
        # every match is projected with the depth maps and pose of ITS sample (the reference flattens all windows
        # into one [1, M*WW, 2] call, which is only defined for one pair per GPU; same result for N = 1)
        q = torch.empty_like(p0)
        for s in range(data['depth0'].shape[0]):
            sel = (b == s).nonzero(as_tuple=True)[0]
            if sel.numel() == 0:
                continue
            one = {k: data[k][s:s + 1] for k in ('depth0', 'depth1', 'T_0to1', 'K0', 'K1')}
            ok, qs = warp_kpts(p0[sel].reshape(1, -1, 2), one['depth0'], one['depth1'], one['T_0to1'], one['K0'], one['K1'])
            qs.masked_fill_(~ok[..., None], -100000)
            q[sel] = qs.view(-1, WW, 2)
    else:
        Hm = data['H_0to1']
        q = warp_points_batch(p0, Hm[b] if Hm.dim() == 3 else Hm)
    dist = torch.sqrt(((q[:, :, None] - p1[:, None]) ** 2).sum(-1))            # [M, WW, WW]
    only = torch.zeros(M, WW * WW, dtype=dist.dtype, device=dev)
    only[torch.arange(M, device=dev), dist.view(M, -1).argmin(1)] = 1
    dist = dist * only.view(M, WW, WW)
    data['conf_matrix_fine_gt'] = (dist <= 3) * (dist > 0)
