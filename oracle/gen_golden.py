"""Golden-vector generator: runs the ORIGINAL GeoFormer modules (imported from /root/reference
through oracle/_ref_import.py) on the seeded inputs of oracle/golden_inputs.py and writes
tests/golden/*.npz.

TEST INFRASTRUCTURE.  Runs only in the build container (the reference checkout does not exist on
the GPU box).  Fixtures are data: a digest of the inputs, the injected (M, mask) homographies, and
the reference's outputs.  Weights are never stored - both sides rebuild them from tensor names
with geoformer_oracle.closed_form_fill(); inputs are rebuilt from seeds by golden_inputs.py.

    python oracle/gen_golden.py            # rewrites every fixture
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R          # noqa: E402
import geoformer_oracle as O     # noqa: E402
import golden_inputs as GI       # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
torch.set_num_threads(1)   # bit-stable generation


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f'{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB  {len(arrs)} arrays')


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


def weights_for(module, prefix):
    """Closed-form weights for a sub-module as if it lived at `prefix` in the GeoFormer state dict."""
    named = {prefix + k: v for k, v in module.state_dict().items()}
    O.closed_form_fill(named)
    module.load_state_dict({k[len(prefix):]: v for k, v in named.items()})
    return module.eval()


def robust_dlt_homography(kp0, kp1, thr=8.0):
    """Deterministic stand-in for cv2.findHomography used ONLY to drive the reference while
    generating fixtures: normalised DLT re-fitted three times on its own inliers.  Its outputs are
    recorded in the fixtures, so nothing downstream depends on this implementation."""
    a, b = kp0.astype(np.float64), kp1.astype(np.float64)
    n = len(a)
    keep = np.ones(n, bool)
    Hm = None
    for _ in range(4):
        if keep.sum() < 4:
            return None, np.zeros((n, 1), np.uint8)
        aa, bb = a[keep], b[keep]

        def norm(p):
            c = p.mean(0)
            s = np.sqrt(2.0) / max(np.sqrt(((p - c) ** 2).sum(1)).mean(), 1e-12)
            return (p - c) * s, np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
        an, Ta = norm(aa)
        bn, Tb = norm(bb)
        A = np.zeros((2 * len(aa), 9))
        A[0::2, 0:2] = an; A[0::2, 2] = 1; A[0::2, 6:8] = -bn[:, :1] * an; A[0::2, 8] = -bn[:, 0]
        A[1::2, 3:5] = an; A[1::2, 5] = 1; A[1::2, 6:8] = -bn[:, 1:] * an; A[1::2, 8] = -bn[:, 1]
        _, _, vt = np.linalg.svd(A)
        Hm = np.linalg.inv(Tb) @ vt[-1].reshape(3, 3) @ Ta
        if abs(Hm[2, 2]) < 1e-12:
            return None, np.zeros((n, 1), np.uint8)
        Hm = Hm / Hm[2, 2]
        p = np.c_[a, np.ones(n)] @ Hm.T
        err = np.sqrt(((p[:, :2] / p[:, 2:3] - b) ** 2).sum(1))
        med = np.median(err)
        keep = err < max(thr, min(med, 4 * thr))
    keep = err < thr
    return Hm, keep.astype(np.uint8)[:, None]


class Recorder:
    def __init__(self, fn):
        self.fn, self.calls = fn, []

    def __call__(self, a, b):
        M, mask = self.fn(a, b)
        self.calls.append((np.array(a), np.array(b), None if M is None else np.array(M), np.array(mask)))
        return M, mask

    def arrays(self, prefix='ransac'):
        out = {f'{prefix}_ncalls': np.array(len(self.calls))}
        for i, (a, b, M, mask) in enumerate(self.calls):
            out[f'{prefix}{i}_kp0'] = a; out[f'{prefix}{i}_kp1'] = b
            out[f'{prefix}{i}_valid'] = np.array(M is not None)
            out[f'{prefix}{i}_M'] = np.zeros((3, 3)) if M is None else M
            out[f'{prefix}{i}_mask'] = mask
        return out


def main():
    ns = R.import_reference()
    cfg = copy.deepcopy(dict(ns.cvpr_ds_config.default_cfg))

    # ---------------- G1 position encoding ----------------
    I = GI.g1_inputs()
    arrs = {'input_digest': GI.digest(I['x'])}
    for tag, fix in (('bug', False), ('fix', True)):
        pe = ns.position_encoding.PositionEncodingSine(256, temp_bug_fix=fix)
        arrs[f'table_{tag}_4x5'] = pe.pe[0, :, :4, :5]
        arrs[f'table_{tag}_samples'] = pe.pe[0][:, I['sample_ys'], I['sample_xs']]
        arrs[f'out_{tag}'] = pe(I['x'])
    save('g1_position_encoding', **arrs)

    # ---------------- G2 linear attention ----------------
    I = GI.g2_inputs()
    la = ns.linear_attention.LinearAttention()
    save('g2_linear_attention', input_digest=GI.digest(*tensors_of(I)),
         out_nomask=la(I['q'], I['k'], I['v']), out_mask=la(I['q'], I['k'], I['v'], I['q_mask'], I['kv_mask']),
         out_fine=la(I['qf'], I['kf'], I['vf']))

    # ---------------- G3 LoFTR encoder layer + schedule ----------------
    I = GI.g3_inputs()
    lay = weights_for(ns.loftr_transformer.LoFTREncoderLayer(256, 8, 'linear'), 'loftr_coarse.layers.0.')
    layf = weights_for(ns.loftr_transformer.LoFTREncoderLayer(128, 8, 'linear'), 'loftr_fine.layers.1.')
    lft = weights_for(ns.loftr_transformer.LocalFeatureTransformer(cfg['coarse']), 'loftr_coarse.')
    with torch.no_grad():
        o0, o1 = lft(I['f0'][:1], I['f1'][:1])
        p0, p1 = lft(I['f0'], I['f1'], I['m0'], I['m1'])
        save('g3_loftr_layer', input_digest=GI.digest(*tensors_of(I)), out_cross=lay(I['x'], I['src']),
             out_cross_masked=lay(I['x'], I['src'], I['x_mask'], I['src_mask']), out_self=lay(I['x'], I['x']),
             out_fine=layf(I['xf'], I['sf']), sched_f0=o0, sched_f1=o1, sched_f0_masked=p0, sched_f1_masked=p1)

    # ---------------- G4 Geo encoder layer / full attention ----------------
    I = GI.g4_inputs()
    glay = weights_for(ns.geo_transformer.LoFTREncoderLayer(256, 4, False), 'geo_module.des_transformer.layers.1.')
    fa = ns.geo_attention.FullAttention()
    with torch.no_grad():
        save('g4_geo_layer', input_digest=GI.digest(*tensors_of(I)), out_self=glay(I['x_self'], I['src_self']),
             out_cross=glay(I['x_cross'], I['src_cross'], None, I['kv_mask']),
             attn_nomask=fa(I['qa'], I['ka'], I['va']), attn_mask=fa(I['qa'], I['ka'], I['va'], None, I['kam']))

    # ---------------- G5 coarse matching ----------------
    I = GI.g5_inputs()
    arrs = {'input_digest': GI.digest(*tensors_of(I))}
    mc = copy.deepcopy(cfg['match_coarse']); mc['thr'] = I['thr']
    cm = ns.coarse_matching.CoarseMatching(mc)
    hw0, hw1 = I['hw0'], I['hw1']
    for tag in ('plain', 'masked', 'forced', 'ties'):
        c = I[tag]
        d = {'hw0_i': torch.tensor([hw0[0] * 8, hw0[1] * 8]), 'hw1_i': torch.tensor([hw1[0] * 8, hw1[1] * 8]),
             'hw0_c': torch.tensor(hw0), 'hw1_c': torch.tensor(hw1)}
        d.update({k: v for k, v in c.items() if k not in ('f0', 'f1')})
        m0 = c['mask0'].flatten(-2) if 'mask0' in c else None
        m1 = c['mask1'].flatten(-2) if 'mask1' in c else None
        cm(c['f0'], c['f1'], d, mask_c0=m0, mask_c1=m1)
        for k in ('conf_matrix', 'b_ids', 'i_ids', 'j_ids', 'mkpts0_c', 'mkpts1_c', 'mconf', 'm_bids'):
            arrs[f'{tag}_{k}'] = d[k]
        print(f'   coarse case {tag}: M={len(d["b_ids"])}')
    save('g5_coarse_matching', **arrs)

    # ---------------- G6 window geometry + gather ----------------
    I = GI.g6_inputs()
    H0, W0, H1, W1 = I['dims']
    kp = ns.common_utils.get_map_keypoints(H0, W0, 8)
    arrs = {'input_digest': GI.digest(I['fmap']), 'grid': kp}
    for tag, Hm in I['H'].items():
        wp = ns.homography.warp_points_batch(kp.unsqueeze(0), homographies=torch.from_numpy(Hm).unsqueeze(0).float())[0]
        wins, masks = ns.common_utils.generate_window([wp], (H1, W1), window_size=5, scale=8)
        smp = ns.common_utils.sample_descriptors([wins[0]], I['fmap'], 8)[0]
        arrs[f'{tag}_warped'] = wp; arrs[f'{tag}_kps'] = wins[0].to(torch.int16)
        arrs[f'{tag}_mask'] = masks[0]; arrs[f'{tag}_gather'] = smp
    save('g6_window_geometry', **arrs)

    # ---------------- G7 GeoModule ----------------
    I = GI.g7_inputs()
    gcfg = copy.deepcopy(ns.geo_config.default_cfg)
    gm = ns.geo_module.GeoModule(gcfg, 256)
    sd = {'geo_module.' + k: v for k, v in gm.state_dict().items()}
    O.closed_form_fill(sd)
    gm.load_state_dict({k[len('geo_module.'):]: v for k, v in sd.items()})
    gm.eval()
    h, w = I['h'], I['w']
    batch = {'image0': torch.zeros(2, 1, h * 8, w * 8), 'image1': torch.zeros(2, 1, h * 8, w * 8),
             'hw0_i': torch.tensor([h * 8, w * 8]), 'hw0_c': torch.tensor([h, w]),
             'mkpts0_c': I['mkpts0_c'], 'mkpts1_c': I['mkpts1_c'], 'm_bids': I['m_bids']}

    def planted(Hm):
        def fn(a, b):
            p = np.c_[a.astype(np.float64), np.ones(len(a))] @ Hm.T
            err = np.sqrt(((p[:, :2] / p[:, 2:3] - b) ** 2).sum(1))
            return Hm.copy(), (err < 8.0).astype(np.uint8)[:, None]
        return fn
    arrs = {'input_digest': GI.digest(I['c0'], I['c1'], I['mkpts0_c'], I['mkpts1_c'])}
    for tag, fn in (('shift', planted(I['H_shift'])),
                    ('nohomo', lambda a, b: (None, np.zeros((len(a), 1), np.uint8))),
                    ('persp', lambda a, b: (I['H_persp'].copy(), (np.arange(len(a)) % 3 != 0).astype(np.uint8)[:, None]))):
        rec = Recorder(fn)
        R.set_find_homography(rec)
        with torch.no_grad():
            o0, o1 = gm(I['c0'].clone(), I['c1'].clone(), batch)
        assert len(rec.calls) == 1
        sub = slice(None) if tag == 'shift' else slice(None, None, 4)
        arrs.update({f'{tag}_out0': o0[..., sub], f'{tag}_out1': o1[..., sub], f'{tag}_valid': rec.calls[0][2] is not None,
                     f'{tag}_M': np.zeros((3, 3)) if rec.calls[0][2] is None else rec.calls[0][2],
                     f'{tag}_mask': rec.calls[0][3]})
        print(f'   geo_module {tag}: inliers {int(rec.calls[0][3].sum())} of {len(rec.calls[0][3])}')
    save('g7_geo_module', **arrs)

    # ---------------- G8 fine preprocess ----------------
    I = GI.g8_inputs()
    fp = weights_for(ns.fine_preprocess.FinePreprocess(cfg), 'fine_preprocess.')
    d = {'hw0_f': torch.tensor(I['hw0_f']), 'hw0_c': torch.tensor(I['hw0_c']), 'hw1_c': torch.tensor(I['hw1_c']),
         'b_ids': I['b_ids'], 'i_ids': I['i_ids'], 'j_ids': I['j_ids']}
    with torch.no_grad():
        u0, u1 = fp(I['feat_f0'], I['feat_f1'], I['feat_c0'], I['feat_c1'], d)
        d0 = dict(d); d0.update(b_ids=I['b_ids'][:0], i_ids=I['i_ids'][:0], j_ids=I['j_ids'][:0])
        e0, e1 = fp(I['feat_f0'], I['feat_f1'], I['feat_c0'], I['feat_c1'], d0)
    save('g8_fine_preprocess', input_digest=GI.digest(*tensors_of(I)), out0=u0, out1=u1,
         empty_shape=np.array(e0.shape + e1.shape))

    # ---------------- G9 fine matching ----------------
    I = GI.g9_inputs()
    fm = ns.fine_matching2.FineMatching2(I['temperature'], I['thr'])
    arrs = {'input_digest': GI.digest(*tensors_of(I))}
    Mn = I['f0'].shape[0]
    for tag, extra in (('plain', {}), ('scaled', {'scale0': I['scale0'], 'scale1': I['scale1']})):
        d = {'hw0_i': torch.tensor(I['hw0_i']), 'hw0_c': torch.tensor(I['hw0_c']), 'hw0_f': torch.tensor(I['hw0_f']),
             'image0': torch.zeros(2, 1, *I['hw0_i']), 'b_ids': I['b_ids'], 'mkpts0_c': I['mkpts0_c'].clone(),
             'mkpts1_c': I['mkpts1_c'].clone(), 'mconf': torch.full((Mn,), -1.0), 'm_bids': I['b_ids'].clone()}
        d.update(extra)
        with torch.no_grad():
            fm(I['f0'], I['f1'], d)
        for k in ('fine_matrix', 'mkpts0_f', 'mkpts1_f', 'mconf', 'm_bids'):
            arrs[f'{tag}_{k}'] = d[k]
        print(f'   fine case {tag}: Mf={len(d["mconf"])} of {Mn}')
    save('g9_fine_matching', **arrs)

    # ---------------- G10 / G11 end-to-end ----------------
    def build_model(coarse_thr, fine_thr):
        c = copy.deepcopy(dict(ns.cvpr_ds_config.default_cfg))
        gc = copy.deepcopy(ns.geo_config.default_cfg)
        gc.update(coarse_thr=coarse_thr, fine_thr=fine_thr)
        m = ns.full_model.GeoFormer(c, gc).eval()
        sd = m.state_dict(); O.closed_form_fill(sd); m.load_state_dict(sd)
        return m

    class StubBackbone(torch.nn.Module):
        def __init__(self, a, b):
            super().__init__(); self.a, self.b = a, b; self.calls = 0

        def forward(self, x):
            if x.shape[0] == 2 * self.a[0].shape[0] and self.a[0].shape[2:] == self.b[0].shape[2:]:
                return torch.cat([self.a[0], self.b[0]]), torch.cat([self.a[1], self.b[1]])
            self.calls += 1
            return self.a if self.calls % 2 == 1 else self.b

    def run_model(case):
        model = build_model(case['coarse_thr'], case['fine_thr'])
        rec = Recorder(robust_dlt_homography)
        R.set_find_homography(rec)
        if case['feats'] is not None:
            model.backbone = StubBackbone(*case['feats'])
        inter = {}
        h1 = model.loftr_coarse.register_forward_hook(
            lambda m, i, o: inter.update(loftr_f0=o[0].clone(), loftr_f1=o[1].clone()))
        h2 = model.geo_module.register_forward_hook(
            lambda m, i, o: inter.update(geo_f0=o[0].clone(), geo_f1=o[1].clone()))
        with torch.no_grad():
            out = model(dict(case['data']))
        h1.remove(); h2.remove()
        return out, inter, rec

    SMALL = ('b_ids', 'i_ids', 'j_ids', 'mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f', 'mconf', 'm_bids')
    for name, case in GI.g10_cases().items():
        out, inter, rec = run_model(case)
        arrs = {'input_digest': GI.digest(*tensors_of(case['data']), *(tensors_of({'f': case['feats']}) if case['feats'] else []))}
        arrs.update({'out_' + k: out[k] for k in SMALL})
        arrs.update(out_conf_matrix=out['conf_matrix'], out_dect_conf_matrix=out['dect_conf_matrix'],
                    out_fine_matrix_head=out['fine_matrix'][:12])
        arrs.update({'mid_' + k: v[..., ::4] for k, v in inter.items()})
        arrs.update(rec.arrays())
        print(f'   e2e {name}: M={len(out["b_ids"])} Mf={len(out["mkpts0_f"])} ransac_calls={len(rec.calls)} '
              f'inliers={[int(c[3].sum()) for c in rec.calls]} of {[len(c[3]) for c in rec.calls]}')
        save(name, **arrs)

    case = GI.g11_inputs()
    torch.set_num_threads(8)
    out, inter, rec = run_model(case)
    torch.set_num_threads(1)
    print(f'   e2e 640: M={len(out["b_ids"])} Mf={len(out["mkpts0_f"])} inliers={[int(c[3].sum()) for c in rec.calls]}')
    r = rec.arrays()
    save('g11_e2e_640_digest', input_digest=GI.digest(case['feats'][0][0][:, :, :4], case['feats'][1][1][:, :, :4]),
         coarse_ids_digest=GI.digest(out['b_ids'], out['i_ids'], out['j_ids']),
         fine_kpts_digest=GI.digest(out['mkpts0_f'], out['mkpts1_f']),
         M=np.array(len(out['b_ids'])), Mf=np.array(len(out['mkpts0_f'])),
         i_ids=out['i_ids'].to(torch.int16), j_ids=out['j_ids'].to(torch.int16),
         mconf_sum=out['mconf'].double().sum(), mconf_head=out['mconf'][:64],
         mkpts0_f_head=out['mkpts0_f'][:64], mkpts1_f_head=out['mkpts1_f'][:64],
         conf_rowsum_head=out['conf_matrix'][0, :64].sum(-1),
         ransac0_M=r['ransac0_M'], ransac0_valid=r['ransac0_valid'],
         ransac0_mask=np.packbits(r['ransac0_mask'][:, 0]), ransac0_n=np.array(len(r['ransac0_mask'])))


def eval_helpers():
    """G12: the pure-python helpers of the evaluation harness (cal_error_auc, cal_reproj_dists, resize_im)."""
    ev = R.import_eval_helpers()
    rng = np.random.default_rng(121)
    errs = np.abs(rng.normal(0, 4, 57)); errs[5] = np.nan; errs[11] = 250.0
    thr = [1, 3, 5, 10]
    sizes = [(1024, 768), (640, 480), (800, 533), (479, 641), (2000, 1500), (333, 500)]
    arrs = {'errors': errs, 'thresholds': np.array(thr), 'auc': ev.hpatches_helper.cal_error_auc(errs, thr),
            'auc_empty': ev.hpatches_helper.cal_error_auc([], thr), 'sizes': np.array(sizes)}
    Hm = np.array([[1.1, 0.05, -12.0], [-0.03, 0.95, 7.5], [2e-4, -1e-4, 1.0]])
    p1 = rng.uniform(0, 600, (40, 2)); p2 = rng.uniform(0, 600, (40, 2))
    arrs.update(H=Hm, p1=p1, p2=p2, reproj=ev.hpatches_helper.cal_reproj_dists(p1, p2, Hm))
    rs = []
    for (w, h) in sizes:
        for imsize, df, f in ((480, 8, min), (640, 8, min), (-1, 8, min), (1024, 16, max)):
            wt, ht, sc = ev.data_io.resize_im(w, h, imsize=imsize, dfactor=df, value_to_scale=f)
            rs.append([w, h, imsize, df, 0 if f is min else 1, wt, ht, sc[0], sc[1]])
    arrs['resize'] = np.array(rs, dtype=np.float64)
    save('g12_eval_helpers', **arrs)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'eval':
        eval_helpers()
    else:
        main()
        eval_helpers()
