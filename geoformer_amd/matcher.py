"""Caller-side counterpart of the reference's inference / evaluation wrappers:

  * `resize_im`, `load_gray_scale_tensor`   eval_tool/immatch/utils/data_io.py:16-26, :48-62
  * `GeoFormerMatcher`                      inference.py:12-99 and eval_tool/immatch/modules/geoformer.py:12-99
                                            (same constructor arguments, `match_pairs` return convention)
  * `cal_error_auc`, `cal_reproj_dists`, `eval_hpatches`
                                            eval_tool/immatch/utils/hpatches_helper.py:13-34, :94-317

OpenCV and torchvision are not available offline: images are read with PIL, converted to gray and resized as UINT8
images by a restatement of OpenCV's published 8-bit paths (`cv2_gray_u8`: the 14-bit fixed-point BGR->gray of
cv2.imread(..., IMREAD_GRAYSCALE); `cv2_resize_linear_u8`: cv2.resize's default INTER_LINEAR for 8-bit images, 11-bit
fixed-point coefficients, half-pixel centres, round-half-up) - byte work, bit for bit; OpenCV itself is absent from the
build container, so the restatement is pinned by hand-derived vectors (tests/test_matcher_cpu.py), not by OpenCV's own
output.  The homography of the HPatches metric is estimated from the fine matches by the device RANSAC
(`ops.ransac_homography(..., thr=ransac_thres, integer_keypoints=False, min_points=4)`).
"""
import glob
import os
import time
from typing import Optional

import numpy as np
import torch

from . import ops
from .model.cvpr_ds_config import get_default_cfg
from .model.full_model import GeoFormer
from .model.geo_config import get_cfg_model


def resize_im(wo, ho, imsize=None, dfactor=1, value_to_scale=max, aspan=False):
    """Target size: scale so that value_to_scale(w, h) == imsize when it exceeds it (or always with
    aspan), then floor both sides to multiples of dfactor.  Returns (wt, ht, (wo/wt, ho/ht))."""
    wt, ht = wo, ho
    if imsize and imsize > 0 and (aspan or value_to_scale(wo, ho) > imsize):
        s = imsize / value_to_scale(wo, ho)
        ht, wt = int(round(ho * s)), int(round(wo * s))
    wt, ht = int(wt // dfactor * dfactor), int(ht // dfactor * dfactor)
    return wt, ht, (wo / wt, ho / ht)


def cv2_gray_u8(rgb):
    """[H,W,3] uint8 RGB -> [H,W] uint8 as cv2.imread(path, cv2.IMREAD_GRAYSCALE) converts a colour file (data_io.py:50):
    (R*4899 + G*9617 + B*1868 + 2^13) >> 14 (OpenCV imgcodecs' icvCvt_BGR2Gray_8u_C3C1R, 14-bit coefficients)."""
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def _cv_round(x):
    """cvRound: round half to even (lrint) - what saturate_cast<short>(float) does."""
    return np.rint(x)


def _linear_coeffs(ssize, dsize, clamp_index):
    """Source index and the two 11-bit weights of every destination position (imgproc/resize.cpp, resize generic path:
    fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; weights saturate_cast<short>({1 - fx, fx} * 2048)).
    Horizontally an out-of-range sx is clamped together with fx = 0; vertically the ROW indices are clamped instead."""
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_index:
        lo, hi = s < 0, s >= ssize - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, ssize - 1, s))
    w1 = np.clip(_cv_round(f * np.float32(2048)), -32768, 32767).astype(np.int64)
    w0 = np.clip(_cv_round((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
    return s, w0, w1


def cv2_resize_linear_u8(src, wt, ht):
    """cv2.resize(src, (wt, ht)) for a single-channel uint8 image with the default INTER_LINEAR (data_io.py:53), OpenCV's
    fixed-point path: horizontal pass into 19-bit integers (pixel x 11-bit weight), vertical pass
    dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2  (VResizeLinear<uchar, int, short>).  An exact 2x
    decimation in both directions is INTER_AREA instead (resize.cpp: 'INTER_LINEAR && is_area_fast && iscale == 2'):
    (a + b + c + d + 2) >> 2."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    hs, ws = src.shape
    if (ws, hs) == (wt, ht):
        return src.copy()
    if ws == 2 * wt and hs == 2 * ht:
        t = src.astype(np.int64)
        return ((t[0::2, 0::2] + t[0::2, 1::2] + t[1::2, 0::2] + t[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _linear_coeffs(ws, wt, clamp_index=True)
    sy, b0, b1 = _linear_coeffs(hs, ht, clamp_index=False)
    t = src.astype(np.int64)
    rows = t[:, sx] * a0[None, :] + t[:, np.minimum(sx + 1, ws - 1)] * a1[None, :]            # [hs, wt]
    r0, r1 = rows[np.clip(sy, 0, hs - 1)], rows[np.clip(sy + 1, 0, hs - 1)]                  # [ht, wt]
    out = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def load_gray_image(im_path):
    """[H,W] uint8 like cv2.imread(im_path, cv2.IMREAD_GRAYSCALE) (data_io.py:50).
    PNG / PPM / BMP colour files (HPatches is PPM): OpenCV decodes BGR and applies its fixed-point BGR2GRAY - `cv2_gray_u8`,
    byte for byte.  JPEG (the FIRE / ISC evaluation images): OpenCV hands IMREAD_GRAYSCALE to libjpeg as
    out_color_space = JCS_GRAYSCALE, i.e. the decoder emits the Y plane of the file's YCbCr data itself and no RGB image ever
    exists; PIL's `draft('L', ...)` configures libjpeg the same way, so that path is used for JPEG files (bit-equal to the Y
    channel of a YCbCr decode: tests/test_matcher_cpu.py).  Limits: a CMYK / RGB-encoded (Adobe) JPEG falls back to the
    formula on the decoded RGB, and libjpeg-turbo vs libjpeg IDCT differences between the two installations are not ours to pin."""
    from PIL import Image
    im = Image.open(im_path)
    if im.format in ('JPEG', 'MPO') and im.mode in ('RGB', 'L'):
        im.draft('L', im.size)                                       # libjpeg JCS_GRAYSCALE, what OpenCV asks for
        if im.mode == 'L':
            return np.array(im, dtype=np.uint8)
    if im.mode in ('RGB', 'RGBA', 'P', 'CMYK', 'YCbCr'):
        return cv2_gray_u8(np.array(im.convert('RGB'), dtype=np.uint8))
    return np.array(im.convert('L'), dtype=np.uint8)


def load_gray_scale_tensor(im_path, device, imsize=None, dfactor=8, value_to_scale=min, aspan=False):
    """[1,1,H,W] float in [0,1] + (wo/wt, ho/ht); H, W multiples of dfactor (load_gray_scale_tensor_cv, data_io.py:48-62:
    gray uint8 image -> cv2.resize on the uint8 image -> to_tensor)."""
    im = load_gray_image(im_path)
    ho, wo = im.shape
    wt, ht, scale = resize_im(wo, ho, imsize=imsize, dfactor=dfactor, value_to_scale=value_to_scale, aspan=aspan)
    im = cv2_resize_linear_u8(im, wt, ht)
    t = torch.from_numpy(im).to(device=device, dtype=torch.float32)[None, None]
    return t / 255.0, scale


class GeoFormerMatcher:
    def __init__(self, imsize, match_threshold, no_match_upscale=False, ckpt=None, device='cuda', precision='fp32',
                 miopen_search=False):
        """precision / miopen_search are additions: 'fp16' is the fast mode; miopen_search=True lets MIOpen search its
        convolution algorithms once per new image shape (seconds each, ~25 % faster backbone afterwards: 6.8 -> 5.2 ms
        per 480x640 pair) - worth it when a dataset repeats a few shapes."""
        from . import miopen
        miopen.use_shipped_find_db()              # no-op if the caller configured MIOpen already
        if miopen_search:
            torch.backends.cudnn.benchmark = True
        self.device, self.imsize = device, imsize
        self.match_threshold, self.no_match_upscale = match_threshold, no_match_upscale
        conf = get_default_cfg()
        conf['match_coarse']['thr'] = match_threshold
        gcfg = get_cfg_model()
        gcfg['coarse_thr'] = match_threshold
        gcfg['precision'] = precision
        self.model = GeoFormer(conf, gcfg)
        self.ckpt_name = 'random'
        if ckpt is not None:
            sd = torch.load(ckpt, map_location='cpu')
            sd = sd.get('state_dict', sd)
            self.model.load_state_dict(sd, strict=False)
            self.ckpt_name = os.path.splitext(os.path.basename(ckpt))[0]
        self.model = self.model.eval().to(device)
        self.name = f'GeoFormer_{self.ckpt_name}' + ('_noms' if no_match_upscale else '')

    def load_im(self, im_path):
        return load_gray_scale_tensor(im_path, self.device, imsize=self.imsize, dfactor=8, value_to_scale=min)

    def match_inputs_(self, gray1, gray2):
        with torch.no_grad():
            batch = self.model({'image0': gray1, 'image1': gray2})
        kpts1, kpts2 = batch['mkpts0_f'].cpu().numpy(), batch['mkpts1_f'].cpu().numpy()
        scores = batch['mconf'].cpu().numpy()
        return np.concatenate([kpts1, kpts2], axis=1), kpts1, kpts2, scores

    def match_pairs(self, im1_path, im2_path):
        gray1, sc1 = self.load_im(im1_path)
        gray2, sc2 = self.load_im(im2_path)
        upscale = np.array([sc1 + sc2])
        matches, kpts1, kpts2, scores = self.match_inputs_(gray1, gray2)
        if self.no_match_upscale:
            return matches, kpts1, kpts2, scores, upscale.squeeze(0)
        return upscale * matches, sc1 * kpts1, sc2 * kpts2, scores

    __call__ = match_pairs


# ---------------------------------------------------------------------------------------------
# HPatches homography metric
# ---------------------------------------------------------------------------------------------
def cal_error_auc(errors, thresholds):
    """Area under the recall-vs-error curve up to each threshold, normalised by the threshold."""
    errors = np.asarray(errors, dtype=float)
    if errors.size == 0:
        return np.zeros(len(thresholds))
    n = errors.size
    err = np.concatenate([[0.0], np.sort(errors)])
    rec = np.arange(n + 1) / n
    out = []
    for t in thresholds:
        k = np.searchsorted(err, t)
        x = np.concatenate([err[:k], [t]])
        y = np.concatenate([rec[:k], [rec[k - 1]]])
        out.append(float(np.sum((y[1:] + y[:-1]) * 0.5 * np.diff(x))) / t)      # trapezoid rule
    return np.array(out, dtype=float)


def cal_reproj_dists(p1s, p2s, homography):
    p = np.concatenate([p1s, np.ones((len(p1s), 1))], axis=1) @ np.asarray(homography).T
    return np.sqrt(((p2s - p[:, :2] / p[:, 2:]) ** 2).sum(1))


def scale_homography(sw, sh):
    return np.array([[sw, 0, 0], [0, sh, 0], [0, 0, 1.0]])


def estimate_homography(matches, thr, device='cuda'):
    """RANSAC homography from [n,4] matches on the device; returns (H | None, inlier mask)."""
    if len(matches) < 4:
        return None, np.zeros(len(matches), bool)
    m = torch.as_tensor(matches, dtype=torch.float32, device=device)
    counts = torch.tensor([len(m), len(m)], dtype=torch.int32, device=device)
    rs = ops.ransac_homography(m[:, :2].contiguous(), m[:, 2:4].contiguous(), counts, 1, 1.0, thr=thr,
                               integer_keypoints=False, min_points=4)
    if int(rs['valid'][0]) == 0:
        return None, np.zeros(len(matches), bool)
    return rs['M'][0].cpu().numpy(), rs['keep'][:len(m)].cpu().numpy().astype(bool)


def corner_error(H_pred, H_gt, w, h):
    corners = np.array([[0, 0, 1], [0, h - 1, 1], [w - 1, 0, 1], [w - 1, h - 1, 1.0]])
    a = corners @ np.asarray(H_gt).T
    b = corners @ np.asarray(H_pred).T
    return float(np.mean(np.linalg.norm(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:], axis=1)))


def hpatches_pairs(data_root, max_seqs: Optional[int] = None):
    """The protocol's pair list: every sequence (reverse lexical order, as hpatches_helper.eval_hpatches walks
    them), image 1 against images 2..6 -> [(sequence name, im1, im2, H_1_k path)]."""
    seqs = sorted(glob.glob(os.path.join(data_root, '*')))[::-1]
    if max_seqs:
        seqs = seqs[:max_seqs]
    return [(os.path.basename(seq), os.path.join(seq, '1.ppm'), os.path.join(seq, f'{k}.ppm'), os.path.join(seq, f'H_1_{k}'))
            for seq in seqs for k in range(2, 7)]


def eval_hpatches(matcher, data_root, ransac_thres=3, thres=(1, 3, 5, 10), scale_H=True, max_seqs: Optional[int] = None,
                  log=print):
    """Homography-estimation AUC over HPatches sequences (pairs 1 -> 2..6), the protocol of
    hpatches_helper.eval_hpatches with task='homography', h_solver='cv'.

    Pairs are independent: under torch.distributed (one process per GPU) the pair list is cut into contiguous
    per-rank blocks by `shard.run_sharded` and every rank ends up with the summaries of ALL pairs, so the metric
    is identical on every rank and to a single-process run."""
    from PIL import Image
    from .shard import run_sharded

    def match_block(block):
        out = []
        for sname, im1, im2, hfile in block:
            H_gt = np.loadtxt(hfile)
            scale = np.ones(4)
            t0 = time.time()
            res = matcher(im1, im2)
            dt = time.time() - t0
            matches = res[0]
            if scale_H and len(res) > 4:       # matches stay in resized coordinates: move the GT homography there
                scale = res[4]
                H_gt = np.linalg.inv(scale_homography(scale[2], scale[3])) @ H_gt @ scale_homography(scale[0], scale[1])
            H_pred, _ = estimate_homography(matches, ransac_thres, matcher.device)
            if H_pred is None:
                d = float('nan')
            else:
                w, h = Image.open(im1).size
                d = corner_error(H_pred, H_gt, w / scale[0], h / scale[1])
            out.append((sname, d, len(matches), dt))
        return out
    rows = run_sharded(hpatches_pairs(data_root, max_seqs), match_block, batch=5)
    dists = {'a': [], 'i': [], 'v': []}
    for sname, d, _, _ in rows:
        dists['a'].append(d)
        dists[sname[0] if sname[0] in 'iv' else 'v'].append(d)
    out = {}
    for key, ds in dists.items():
        ds = np.asarray(ds, dtype=float)
        out['correct_' + key] = np.mean([[float(d <= t) for t in thres] for d in ds], axis=0) if len(ds) else np.zeros(len(thres))
        out['auc_' + key] = cal_error_auc(ds, thres)
    out.update(failed=int(sum(np.isnan(r[1]) for r in rows)), mean_matches=float(np.mean([r[2] for r in rows])) if rows else 0.0,
               match_time=float(np.mean([r[3] for r in rows])) if rows else 0.0, pairs=len(rows))
    log(f"Hest Correct: a={out['correct_a']} i={out['correct_i']} v={out['correct_v']}")
    log(f"Hest AUC: a={out['auc_a']} i={out['auc_i']} v={out['auc_v']}")
    return out


# ---------------------------------------------------------------------------------------------
# command line: the counterparts of `python inference.py` and `python eval_Hpatches.py`
#   python -m geoformer_amd.matcher match im1 im2 [--ckpt saved_ckpt/geoformer.ckpt] [--out matches.npz]
#   python -m geoformer_amd.matcher hpatches /path/to/hpatches-sequences-release [--ckpt ...]
# ---------------------------------------------------------------------------------------------
def main(argv=None):
    """Defaults follow the reference per sub-command: `match` = inference.py:107 (imsize 640, matches scaled back to the
    original images); `hpatches` = eval_configs/geoformer.yml:7-11 with eval_Hpatches.py:96-100 (imsize 480, match
    threshold 0.2, no_match_upscale True -> the ground-truth homography is moved into resized coordinates, RANSAC
    threshold 3).  Under `python -m torch.distributed.run --nproc-per-node N -m geoformer_amd.matcher hpatches ...` the
    pair list is sharded over the N GPUs (shard.run_sharded)."""
    import argparse
    ap = argparse.ArgumentParser(prog='python -m geoformer_amd.matcher')
    sub = ap.add_subparsers(dest='cmd', required=True)
    m = sub.add_parser('match', help='match one image pair (inference.py)')
    m.add_argument('im1'), m.add_argument('im2'), m.add_argument('--out', default=None)
    m.add_argument('--imsize', type=int, default=640)
    m.add_argument('--no-match-upscale', action='store_true')
    h = sub.add_parser('hpatches', help='homography AUC over HPatches sequences (eval_Hpatches.py)')
    h.add_argument('root'), h.add_argument('--max-seqs', type=int, default=None)
    h.add_argument('--ransac-thres', type=float, default=3.0)
    h.add_argument('--imsize', type=int, default=480)
    h.add_argument('--match-upscale', dest='no_match_upscale', action='store_false',
                   help='scale matches back to the original images instead of moving the GT homography (reference: off)')
    h.set_defaults(no_match_upscale=True)
    for p in (m, h):
        p.add_argument('--ckpt', default=None)
        p.add_argument('--match-threshold', type=float, default=0.2)
        p.add_argument('--precision', choices=('fp32', 'fp16', 'bf16'), default='fp16',
                       help="fp16 / bf16 = the fast modes (16-bit storage, fp32 accumulation; bf16 = BASELINE configs[1]'s wording); "
                            "fp32 = the reference's arithmetic")
    args = ap.parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        torch.cuda.set_device(local)
        torch.distributed.init_process_group('nccl', device_id=torch.device('cuda', local))
    matcher = GeoFormerMatcher(args.imsize, args.match_threshold, args.no_match_upscale, args.ckpt, device=f'cuda:{local}',
                               precision=args.precision)
    if args.cmd == 'match':
        res = matcher(args.im1, args.im2)
        print(f'{matcher.name}: {len(res[0])} matches')
        if args.out:
            np.savez(args.out, matches=res[0], kpts1=res[1], kpts2=res[2], scores=res[3])
    else:
        quiet = int(os.environ.get('RANK', '0')) != 0
        out = eval_hpatches(matcher, args.root, ransac_thres=args.ransac_thres, max_seqs=args.max_seqs,
                            scale_H=matcher.no_match_upscale, log=(lambda s: None) if quiet else print)
        if not quiet:
            print({k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in out.items()})
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
