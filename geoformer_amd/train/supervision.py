"""Ground-truth supervision for the training step, torch ops under no_grad on the batch's device.

  spvs_coarse   model/loftr_src/loftr/utils/supervision.py:23-115   mutual-nearest coarse cells under the GT warp
  spvs_fine2    model/loftr_src/loftr/utils/supervision.py:270-387  25x25 window labels around the predicted matches
  warp_kpts     model/loftr_src/loftr/utils/geometry.py:5-54        depth + pose warp (MegaDepth / ScanNet branch)

Deviation (documented): `spvs_fine2` warps the windows of match m with the homography of ITS sample
(`H[b_ids[m]]`); the reference hands the whole `[N,3,3]` stack to `warp_points_batch`, which only works for one
pair per GPU.  Identical for N = 1.
"""
import torch


def _meshgrid_xy(h, w, device):
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing='ij')
    return torch.stack([xs, ys], -1)            # kornia.create_meshgrid(h, w, False): [..., (x, y)]


def warp_points_batch(points, homographies):
    """utils/homography.py:86-105: points [B,l,2], homographies [B,3,3] or [3,3]; exact-zero w -> 1e-6."""
    B, l = points.shape[:2]
    pts = torch.cat([points, torch.ones(B, l, 1, device=points.device, dtype=points.dtype)], dim=-1)
    if homographies.dim() == 2:
        homographies = homographies[None]
    if homographies.shape[0] != B:
        homographies = homographies.expand(B, 3, 3) if homographies.shape[0] == 1 else homographies.repeat(B, 1, 1)
    w = torch.bmm(homographies.to(pts.dtype), pts.permute(0, 2, 1)).permute(0, 2, 1)
    sc = w[:, :, 2:].clone()
    sc[sc == 0] = 1e-6
    return w[:, :, :2] / sc


@torch.no_grad()
def warp_kpts(kpts0, depth0, depth1, T_0to1, K0, K1):
    """geometry.py:5-54: unproject with depth0, transform, project with K1, depth-consistency check (0.2 rel)."""
    kl = kpts0.round().long()
    n = kpts0.shape[0]
    d0 = torch.stack([depth0[i, kl[i, :, 1].clamp(0, depth0.shape[1] - 1), kl[i, :, 0].clamp(0, depth0.shape[2] - 1)]
                      for i in range(n)], dim=0)
    nonzero = d0 != 0
    h0 = torch.cat([kpts0, torch.ones_like(kpts0[:, :, :1])], dim=-1) * d0[..., None]
    cam = K0.inverse() @ h0.transpose(2, 1)
    w_cam = T_0to1[:, :3, :3] @ cam + T_0to1[:, :3, [3]]
    w_depth = w_cam[:, 2, :]
    wh = (K1 @ w_cam).transpose(2, 1)
    w_kpts0 = wh[:, :, :2] / (wh[:, :, [2]] + 1e-4)
    h, w = depth1.shape[1:3]
    covis = (w_kpts0[:, :, 0] > 0) * (w_kpts0[:, :, 0] < w - 1) * (w_kpts0[:, :, 1] > 0) * (w_kpts0[:, :, 1] < h - 1)
    wl = w_kpts0.long()
    wl[~covis, :] = 0
    d1 = torch.stack([depth1[i, wl[i, :, 1], wl[i, :, 0]] for i in range(n)], dim=0)
    consistent = ((d1 - w_depth) / d1).abs() < 0.2
    return nonzero * covis * consistent, w_kpts0


@torch.no_grad()
def spvs_coarse(data, resolution=(8, 2)):
    """Writes conf_matrix_gt [N,hw0,hw1], spv_b_ids/spv_i_ids/spv_j_ids, spv_w_pt0_i, spv_pt1_i."""
    device = data['image0'].device
    N, _, H0, W0 = data['image0'].shape
    _, _, H1, W1 = data['image1'].shape
    scale = resolution[0]
    scale0 = scale * data['scale0'][:, None] if 'scale0' in data else scale
    scale1 = scale * data['scale1'][:, None] if 'scale0' in data else scale
    h0, w0, h1, w1 = H0 // scale, W0 // scale, H1 // scale, W1 // scale
    grid_pt0_i = scale0 * _meshgrid_xy(h0, w0, device).reshape(1, h0 * w0, 2).repeat(N, 1, 1)
    grid_pt1_i = scale1 * _meshgrid_xy(h1, w1, device).reshape(1, h1 * w1, 2).repeat(N, 1, 1)
    if 'mask0' in data:                                     # zero-padded regions -> (0, 0)
        grid_pt0_i[~data['mask0'].flatten(-2).bool()] = 0
        grid_pt1_i[~data['mask1'].flatten(-2).bool()] = 0
    if 'depth0' in data:
        _, w_pt0_i = warp_kpts(grid_pt0_i, data['depth0'], data['depth1'], data['T_0to1'], data['K0'], data['K1'])
        _, w_pt1_i = warp_kpts(grid_pt1_i, data['depth1'], data['depth0'], data['T_1to0'], data['K1'], data['K0'])
    else:
        w_pt0_i = warp_points_batch(grid_pt0_i, data['H_0to1'])
        w_pt1_i = warp_points_batch(grid_pt1_i, data['H_1to0'])
    w_pt0_c = (w_pt0_i / scale1).round().long()
    w_pt1_c = (w_pt1_i / scale0).round().long()
    nearest_index1 = w_pt0_c[..., 0] + w_pt0_c[..., 1] * w1
    nearest_index0 = w_pt1_c[..., 0] + w_pt1_c[..., 1] * w0

    def oob(pt, w, h):
        return (pt[..., 0] < 0) | (pt[..., 0] >= w) | (pt[..., 1] < 0) | (pt[..., 1] >= h)
    nearest_index1[oob(w_pt0_c, w1, h1)] = 0
    nearest_index0[oob(w_pt1_c, w0, h0)] = 0
    loop_back = torch.gather(nearest_index0, 1, nearest_index1)
    correct = loop_back == torch.arange(h0 * w0, device=device)[None]
    correct[:, 0] = False                                   # the top-left cell is the out-of-bounds sink
    conf_gt = torch.zeros(N, h0 * w0, h1 * w1, device=device)
    b_ids, i_ids = torch.where(correct)
    j_ids = nearest_index1[b_ids, i_ids]
    conf_gt[b_ids, i_ids, j_ids] = 1
    data['spv_num_gt'] = int(len(b_ids))                   # before the dummy below (the loss terms need the real count)
    if len(b_ids) == 0:                                     # keeps the fine level alive; does not touch its loss
        b_ids = i_ids = j_ids = torch.zeros(1, dtype=torch.long, device=device)
    data.update(conf_matrix_gt=conf_gt, spv_b_ids=b_ids, spv_i_ids=i_ids, spv_j_ids=j_ids, spv_w_pt0_i=w_pt0_i,
                spv_pt1_i=grid_pt1_i)


@torch.no_grad()
def spvs_fine2(data, resolution=(8, 2)):
    """Writes conf_matrix_fine_gt [M, W*W, W*W] (bool): for each predicted coarse match the single window-cell pair
    whose GT-warped distance is smallest, if that distance is in (0, 3] pixels."""
    device = data['image0'].device
    W = int(data['W'])
    WW = W * W
    ck0, ck1 = data['mkpts0_c'], data['mkpts1_c']
    M = ck0.shape[0]
    if M == 0:
        data['conf_matrix_fine_gt'] = torch.zeros(0, WW, WW, dtype=torch.bool, device=device)
        return
    grid_w = _meshgrid_xy(W, W, device).reshape(1, WW, 2).repeat(M, 1, 1) - W // 2
    b = data['b_ids']
    cs = data['hw0_i'][0] // data['hw0_c'][0]
    cs0 = cs * data['scale0'][b] if 'scale0' in data else cs
    cs1 = cs * data['scale1'][b] if 'scale1' in data else cs
    c2f = data['hw0_f'][0] // data['hw0_c'][0]
    kpts0 = (ck0 / cs0 * c2f)[:, None].repeat(1, WW, 1) + grid_w
    kpts1 = (ck1 / cs1 * c2f)[:, None].repeat(1, WW, 1) + grid_w
    fs = data['hw0_i'][0] // data['hw0_f'][0]
    fs0 = (fs * data['scale0'][b])[:, None].repeat(1, WW, 1) if 'scale0' in data else fs
    fs1 = (fs * data['scale1'][b])[:, None].repeat(1, WW, 1) if 'scale1' in data else fs
    kpts0_raw, kpts1_raw = kpts0 * fs0, kpts1 * fs1
    if 'depth0' in data:
        flat0 = kpts0_raw.reshape(1, M * WW, 2)
        if data['depth0'].shape[0] != 1:
            raise NotImplementedError('depth-based fine supervision follows the reference: one pair per GPU')
        mk0, w_pt0_i = warp_kpts(flat0, data['depth0'], data['depth1'], data['T_0to1'], data['K0'], data['K1'])
        w_pt0_i[~mk0] = -100000
        w_pt0_i = w_pt0_i.view(M, WW, 2)
    else:
        H = data['H_0to1']
        w_pt0_i = warp_points_batch(kpts0_raw, H[b] if H.dim() == 3 else H)
    dis = torch.sqrt(((w_pt0_i[:, :, None] - kpts1_raw[:, None]) ** 2).sum(-1))         # [M, WW, WW]
    best = dis.view(M, -1).argmin(1)
    keep = torch.zeros(M, WW * WW, dtype=dis.dtype, device=device)
    keep[torch.arange(M, device=device), best] = 1
    dis = dis * keep.view(M, WW, WW)
    data['conf_matrix_fine_gt'] = (dis <= 3) * (dis > 0)
