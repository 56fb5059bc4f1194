"""Which rounding points of the 16-bit storage modes cost the matches?  (VERDICT r05 #2a; a CPU experiment with the storage oracle.)

The storage oracle (oracle/geoformer_oracle.py: geoformer_forward_storage and its stage functions) restates the reference's arithmetic
with a round trip through the storage type wherever the HIP kernels round.  Here every STAGE gets its own storage type, and the stages are
switched to fp32 one at a time (everything else 16-bit), on the synthetic HPatches-protocol set of tests/test_outcome_parity_gpu.py
(13 sequences x 5 pairs of planted maps, 480x640 against 480x608):
   maps      the backbone's maps and the position-encoded maps as stored
   loftr     the eight coarse LoFTR layers (K9: every MFMA operand + the layer output)
   k1a_in    ONLY the storage of the features the first dual-softmax reads (the last LoFTR layer call's output of each image kept fp32)
   geo       the four GeoTransformer layers (K3 / K4 / K5 / K9 finish)
   k1b_in    ONLY the storage of the features the second dual-softmax reads (the last Geo layer's outputs kept fp32)
   fine      FinePreprocess + the two fine-level layers (K7 / K3 / K11) and with them the features K8 reads
Per configuration: dAUC@1 / dAUC@3 against the fp32 oracle (C RANSAC on both sides, 3 px, sub-pixel keypoints), the number of pairs
whose coarse match SET differs from the fp32 run's and the total number of differing coarse matches (symmetric difference of (i, j) after
the second coarse matching; `first`: after the first), and the number of fine matches that moved.
   python tools/rounding_ablation.py [fp16|bf16] [sequences=13] [pairs=5]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
import geoformer_oracle as O
import golden_inputs as GI
import ransac_oracle as RO
from geoformer_amd import matcher as MT

STAGES = ('maps', 'loftr', 'k1a_in', 'geo', 'k1b_in', 'fine')
F32 = torch.float32


def forward_stages(P, data, S, feats, homography_fn):
    """geoformer_forward_storage with one storage type per stage (S[stage] = torch dtype; all float32 = the fp32 oracle's arithmetic in
    this code path)."""
    rt = O.rt
    loftr_cfg, geo_cfg = O.default_loftr_config(), O.default_geo_config()
    thr, temp = geo_cfg['coarse_thr'], loftr_cfg['match_coarse']['dsmax_temperature']
    img0, img1 = data['image0'], data['image1']
    data.update(bs=torch.tensor(img0.size(0)), hw0_i=torch.tensor(img0.shape[2:]), hw1_i=torch.tensor(img1.shape[2:]))
    (c0, ff0), (c1, ff1) = [(rt(c, S['maps']), rt(f, S['maps'])) for c, f in feats]
    data.update(hw0_c=torch.tensor(c0.shape[2:]), hw1_c=torch.tensor(c1.shape[2:]), hw0_f=torch.tensor(ff0.shape[2:]), hw1_f=torch.tensor(ff1.shape[2:]))
    pe0 = rt(O.add_position_encoding(c0, False), S['maps']).flatten(2).transpose(1, 2).contiguous()
    pe1 = rt(O.add_position_encoding(c1, False), S['maps']).flatten(2).transpose(1, 2).contiguous()
    f0, f1 = pe0, pe1
    names, nh, st = loftr_cfg['coarse']['layer_names'], loftr_cfg['coarse']['nhead'], S['loftr']
    k0 = k1 = None
    for idx, name in enumerate(names):
        lp = f'loftr_coarse.layers.{idx}.'
        if name == 'self':
            u0, u1 = O.encoder_layer_fused(P, lp, f0, f0, nh, st, round_out=False), O.encoder_layer_fused(P, lp, f1, f1, nh, st, round_out=False)
            f0, f1 = rt(u0, st), rt(u1, st)
        else:
            u0 = O.encoder_layer_fused(P, lp, f0, f1, nh, st, round_out=False)
            f0 = rt(u0, st)                                                       # image 1 attends to the STORED new f0
            u1 = O.encoder_layer_fused(P, lp, f1, f0, nh, st, round_out=False)
            f1 = rt(u1, st)
        k0, k1 = u0, u1                                                           # the last layer's un-rounded outputs
    a0, a1 = (k0, k1) if S['k1a_in'] == F32 else (f0, f1)
    conf = O.dual_softmax(a0, a1, temp)
    data['conf_matrix'] = conf
    data.update(O.coarse_match(conf, data, thr))
    first = (data['i_ids'].clone(), data['j_ids'].clone())
    g0, g1 = O.geo_module_storage(P, pe0, pe1, tuple(c0.shape[2:]), tuple(c1.shape[2:]), data, geo_cfg, homography_fn, S['geo'], round_last=False)
    g0r, g1r = rt(g0, S['geo']), rt(g1, S['geo'])                                  # as stored (what the fine level gathers)
    b0, b1 = (g0, g1) if S['k1b_in'] == F32 else (g0r, g1r)
    conf = O.dual_softmax(b0, b1, temp)
    data['conf_matrix'] = conf
    data.update(O.coarse_match(conf, data, thr))
    W = loftr_cfg['fine_window_size']
    data['W'] = torch.tensor(W)
    sf = S['fine']
    u0, u1 = O.fine_preprocess_storage(P, ff0, ff1, g0r, g1r, data, sf, W)
    if u0.size(0) != 0:
        nhf = loftr_cfg['fine']['nhead']
        for idx, name in enumerate(loftr_cfg['fine']['layer_names']):
            lp = f'loftr_fine.layers.{idx}.'
            if name == 'self':
                u0, u1 = O.encoder_layer_chain(P, lp, u0, u0, nhf, sf), O.encoder_layer_chain(P, lp, u1, u1, nhf, sf)
            else:
                u0 = O.encoder_layer_chain(P, lp, u0, u1, nhf, sf)
                u1 = O.encoder_layer_chain(P, lp, u1, u0, nhf, sf)
    data.update(O.fine_match(u0, u1, data, geo_cfg['fine_temperature'], geo_cfg['fine_thr']))
    data['_first'] = first
    return data


def main():
    st = {'fp16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else 'fp16']
    seqs = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    W = O.make_weights()
    configs = {'fp32 (reference)': {s: F32 for s in STAGES}, 'all 16-bit': {s: st for s in STAGES}}
    for s in STAGES:
        configs[f'{s} in fp32'] = {k: (F32 if k == s else st) for k in STAGES}
    configs['k1a_in + k1b_in in fp32'] = {k: (F32 if k in ('k1a_in', 'k1b_in') else st) for k in STAGES}
    configs['k1a_in + k1b_in + fine in fp32'] = {k: (F32 if k in ('k1a_in', 'k1b_in', 'fine') else st) for k in STAGES}
    rows = {name: {'err': [], 'sets': [], 'first': [], 'fine': [], 'n': []} for name in configs}
    base = {'image0': torch.zeros(1, 1, 480, 640), 'image1': torch.zeros(1, 1, 480, 608)}
    t0 = time.time()
    for s in range(seqs):
        for k in range(1, npairs + 1):
            f0, f1, H = GI.hpatches_like_features(s, k)
            for name, S in configs.items():
                with torch.no_grad():
                    out = forward_stages(W, dict(base), S, (f0, f1), RO.make_homography_fn())
                k0, k1 = out['mkpts0_f'].numpy(), out['mkpts1_f'].numpy()
                Hp, _ = RO.find_homography_subpixel(k0, k1, 3.0)
                r = rows[name]
                r['err'].append(MT.corner_error(Hp, H, 640, 480) if Hp is not None else float('nan'))
                r['sets'].append(set(zip(out['i_ids'].tolist(), out['j_ids'].tolist())))
                r['first'].append(set(zip(out['_first'][0].tolist(), out['_first'][1].tolist())))
                r['fine'].append({(i, j): (tuple(np.round(a, 3)), tuple(np.round(b, 3))) for i, j, a, b in
                                  zip(out['i_ids'].tolist(), out['j_ids'].tolist(), k0.tolist(), k1.tolist())} if len(k0) == len(out['i_ids']) else {})
                r['n'].append(len(k0))
            print(f'  sequence {s} pair {k} done ({time.time() - t0:.0f} s)', file=sys.stderr, flush=True)
    ref = rows['fp32 (reference)']
    auc_ref = MT.cal_error_auc(np.array(ref['err']), (1, 3, 5, 10))
    npx = len(ref['err'])
    print(f'rounding ablation, {str(st)[6:]} storage, {npx} synthetic HPatches-protocol pairs; fp32 reference AUC@1/3/5/10 = {np.round(auc_ref, 4).tolist()}, '
          f'{np.mean(ref["n"]):.0f} fine matches per pair')
    print(f'{"configuration":34s} {"dAUC@1":>9s} {"dAUC@3":>9s} {"pairs w/ other set":>19s} {"coarse matches differing (first)":>33s} {"fine matches moved":>19s}')
    for name, r in rows.items():
        auc = MT.cal_error_auc(np.array(r['err']), (1, 3, 5, 10))
        dset = [len(a ^ b) for a, b in zip(r['sets'], ref['sets'])]
        dfirst = [len(a ^ b) for a, b in zip(r['first'], ref['first'])]
        moved = sum(sum(1 for key, v in a.items() if key in b and b[key] != v) for a, b in zip(r['fine'], ref['fine']))
        print(f'{name:34s} {auc[0] - auc_ref[0]:+9.5f} {auc[1] - auc_ref[1]:+9.5f} {sum(1 for d in dset if d):19d} {sum(dset):20d} ({sum(dfirst):6d})      {moved:14d}')


if __name__ == '__main__':
    main()
