"""K1 parity: HIP dual-softmax + match extraction vs the oracle and vs the reference's golden vectors."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O
import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _data(hw0, hw1, extra=None):
    d = {'hw0_i': torch.tensor([hw0[0] * 8, hw0[1] * 8]), 'hw1_i': torch.tensor([hw1[0] * 8, hw1[1] * 8]),
         'hw0_c': torch.tensor(hw0), 'hw1_c': torch.tensor(hw1)}
    d.update(extra or {})
    return d


def _run_hip(f0, f1, thr, hw0, hw1, case, dtype):
    from geoformer_amd import ops
    dev = 'cuda:0'
    kw = {}
    if 'mask0' in case:
        kw.update(mask0=case['mask0'].to(dev), mask1=case['mask1'].to(dev))
    if 'scale0' in case:
        kw.update(scale0=case['scale0'], scale1=case['scale1'])
    out = ops.dual_softmax_match(f0.to(dev, dtype), f1.to(dev, dtype), 0.1, thr, hw0, hw1, 8.0,
                                 force_one='dataset_name' in case, **kw)
    torch.cuda.synchronize()
    M = int(out['counts'][0])
    res = {k: out[k][:M].cpu() for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts0_c', 'mkpts1_c')}
    res['conf_matrix'] = out['conf_matrix'].cpu()
    res['counts'] = out['counts'].cpu()
    return res


@pytest.mark.parametrize('tag', ['plain', 'masked', 'forced', 'ties'])
def test_golden_fp32(golden, tag):
    """fp32 path against the REFERENCE's outputs: indices bit-exact, floats to fp32 rounding."""
    G, I = golden('g5_coarse_matching'), GI.g5_inputs()
    c = I[tag]
    out = _run_hip(c['f0'], c['f1'], I['thr'], I['hw0'], I['hw1'], c, torch.float32)
    np.testing.assert_allclose(out['conf_matrix'].numpy(), G[f'{tag}_conf_matrix'], rtol=2e-5, atol=1e-9)
    for k in ('b_ids', 'i_ids', 'j_ids'):
        np.testing.assert_array_equal(out[k].numpy(), G[f'{tag}_{k}'])
    np.testing.assert_array_equal(out['mkpts0_c'].numpy(), G[f'{tag}_mkpts0_c'])
    np.testing.assert_array_equal(out['mkpts1_c'].numpy(), G[f'{tag}_mkpts1_c'])
    np.testing.assert_allclose(out['mconf'].numpy(), G[f'{tag}_mconf'], rtol=2e-5, atol=1e-9)


def _planted(N, L, S, C, seed, noise=0.35, scale=1.3):
    g = torch.Generator().manual_seed(seed)
    f0 = torch.randn(N, L, C, generator=g) * scale
    f1 = torch.randn(N, S, C, generator=g) * scale
    k = min(L, S) * 2 // 3
    for b in range(N):
        pi = torch.randperm(L, generator=g)[:k]
        pj = torch.randperm(S, generator=g)[:k]
        f1[b, pj] = f0[b, pi] + noise * scale * torch.randn(k, C, generator=g)
    return f0, f1


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape', [(1, 300, 500, 256), (2, 1200, 1184, 256), (1, 4800, 4560, 256)])
def test_vs_oracle_ragged(shape, dtype):
    """Ragged sizes (not multiples of the 128 tile), oracle run on the SAME rounded inputs."""
    N, L, S, C = shape
    f0, f1 = _planted(N, L, S, C, seed=L + S)
    f0, f1 = f0.to(dtype).float(), f1.to(dtype).float()          # identical rounded inputs on both sides
    w0, w1 = 20, 16
    hw0, hw1 = (L // w0, w0), (S // w1, w1)
    out = _run_hip(f0, f1, 0.2, hw0, hw1, {}, dtype)
    torch.set_num_threads(8)
    conf = O.dual_softmax(f0, f1, 0.1)
    ref = O.coarse_match(conf, _data(hw0, hw1), 0.2)
    assert len(ref['b_ids']) > min(L, S) // 4
    np.testing.assert_allclose(out['conf_matrix'].numpy(), conf.numpy(), rtol=1e-4, atol=1e-9)
    for k in ('b_ids', 'i_ids', 'j_ids'):
        np.testing.assert_array_equal(out[k].numpy(), ref[k].numpy())
    np.testing.assert_array_equal(out['mkpts0_c'].numpy(), ref['mkpts0_c'].numpy())
    np.testing.assert_array_equal(out['mkpts1_c'].numpy(), ref['mkpts1_c'].numpy())
    np.testing.assert_allclose(out['mconf'].numpy(), ref['mconf'].numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('thr', [0.0, 0.01])
def test_low_threshold_dense_path(thr):
    """thr < 0.05 switches pass B to the in-tile (dense) reduction: same results required."""
    N, L, S, C = 2, 300, 500, 256
    f0, f1 = _planted(N, L, S, C, seed=5)
    hw0, hw1 = (15, 20), (25, 20)
    out = _run_hip(f0, f1, thr, hw0, hw1, {}, torch.float32)
    conf = O.dual_softmax(f0, f1, 0.1)
    ref = O.coarse_match(conf, _data(hw0, hw1), thr)
    assert len(ref['b_ids']) > 100
    for k in ('b_ids', 'i_ids', 'j_ids'):
        np.testing.assert_array_equal(out[k].numpy(), ref[k].numpy())
    np.testing.assert_allclose(out['mconf'].numpy(), ref['mconf'].numpy(), rtol=1e-4, atol=1e-12)


@pytest.mark.parametrize('thr', [0.2, 0.0])
def test_full_size_properties(thr):
    """BASELINE size (L=S=6400, fp16 features; the row-panel-persistent pass B, sparse candidates at thr = 0.2 and the
    dense-candidate variant bench.py runs at thr = 0): size-independent properties of the dual-softmax -
    row/column sums of sqrt-factors, mutual-nearest consistency, ordering, planted recovery."""
    from geoformer_amd import ops
    N, L, S, C = 2, 6400, 6400, 256
    g = torch.Generator().manual_seed(7)
    f0 = torch.randn(N, L, C, generator=g) * 1.3
    perm = torch.stack([torch.randperm(S, generator=g) for _ in range(N)])
    f1 = torch.stack([f0[b][perm[b]] for b in range(N)]) + 0.4 * torch.randn(N, S, C, generator=g)
    out = ops.dual_softmax_match(f0.cuda().half(), f1.cuda().half(), 0.1, thr, (80, 80), (80, 80), 8.0)
    torch.cuda.synchronize()
    M = int(out['counts'][0])
    conf = out['conf_matrix']
    b, i, j = (out[k][:M] for k in ('b_ids', 'i_ids', 'j_ids'))
    assert M > 0.9 * N * L
    # planted permutation recovered: f1[b, j] = f0[b, perm[b, j]]  ->  i == perm[b, j]
    assert torch.equal(perm.cuda()[b, j], i)
    # torch.where order and per-sample counts
    key = b * L + i
    assert torch.all(key[1:] > key[:-1])
    assert int(out['counts'][1:].sum()) == M
    # every reported match is the maximum of its row and column and exceeds thr
    assert torch.equal(conf[b, i].argmax(-1), j)
    assert torch.equal(conf[b, :, j].argmax(0) if False else conf.transpose(1, 2)[b, j].argmax(-1), i)
    assert torch.all(out['mconf'][:M] > thr) and torch.equal(out['mconf'][:M], conf[b, i, j])
    # conf <= 1 and each row / column of conf sums to <= 1 (product of two probabilities)
    assert float(conf.max()) <= 1.0 + 1e-5
    assert float(conf.sum(-1).max()) <= 1.0 + 1e-4 and float(conf.sum(-2).max()) <= 1.0 + 1e-4


@pytest.mark.parametrize('thr', [0.2, 0.0])
def test_full_size_repeatable(thr):
    """The same 8-pair call 25 times (the shape of a bench step: 640 column runs per row panel, 512 persistent workgroups):
    the confidence matrix and the matches must come out bit-identical every time (a guard against ordering bugs in the
    pipelined pass B - tiles read while their LDS-DMA is in flight, buffers reused too early; it does NOT prove their
    absence: a counted wait that was one tile too loose passed it, the fix came from reading the code)."""
    from geoformer_amd import ops
    N, L, S, C = 8, 6400, 6400, 256
    g = torch.Generator().manual_seed(17)
    f0 = (torch.randn(N, L, C, generator=g) * 1.3).cuda().half()
    f1 = (f0[:, torch.randperm(S, generator=g).cuda()].float() + 0.4 * torch.randn(N, S, C, generator=g).cuda()).half()
    first = None
    for _ in range(25):
        out = ops.dual_softmax_match(f0, f1, 0.1, thr, (80, 80), (80, 80), 8.0)
        M = int(out['counts'][0])
        digest = (M, out['conf_matrix'].view(torch.int32).sum(dtype=torch.int64).item(),
                  out['conf_matrix'][:, ::97, ::89].clone(), out['i_ids'][:M].clone(), out['j_ids'][:M].clone(),
                  out['mconf'][:M].clone())
        if first is None:
            first = digest
            continue
        assert digest[0] == first[0] and digest[1] == first[1]
        for a, b in zip(digest[2:], first[2:]):
            assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('thr', [0.0, 0.2])
def test_match_only_equals_contract_mode(dtype, thr):
    """Match-only mode (conf == NULL; inference.py:51-75 never reads the matrices): ids, mconf, keypoints and
    counts BIT-IDENTICAL to the contract mode's on planted correspondences with duplicated rows and columns (exact ties), both
    candidate forms (thr 0: dense, thr 0.2: sparse); gf_dual_softmax_conf_at returns the matrix entries bit for bit."""
    from geoformer_amd import ops
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(31)
    N, h, w = 2, 16, 24                                   # L = S = 384: three row panels, six column tiles
    L = h * w
    f0 = torch.randn(N, L, 256, generator=g) * 1.2
    f1 = torch.stack([f0[b][torch.randperm(L, generator=g)] for b in range(N)]) + 0.3 * torch.randn(N, L, 256, generator=g)
    f1[0, 5] = f1[0, 200]; f1[1, 77] = f1[1, 300]         # duplicated columns: exact ties inside rows
    f0[0, 40] = f0[0, 7]; f0[1, 333] = f0[1, 12]          # duplicated rows: exact ties inside columns
    f0, f1 = f0.to(dev, dtype), f1.to(dev, dtype)
    assert ops.match_only_supported(f0, f1)
    a = ops.dual_softmax_match(f0, f1, 0.1, thr, (h, w), (h, w), 8.0)
    b = ops.dual_softmax_match(f0, f1, 0.1, thr, (h, w), (h, w), 8.0, materialize=False)
    torch.cuda.synchronize()
    assert b['conf_matrix'] is None
    assert torch.equal(a['counts'], b['counts'])
    M = int(a['counts'][0])
    assert M > 100
    for k in ('b_ids', 'i_ids', 'j_ids', 'mconf', 'mkpts0_c', 'mkpts1_c'):
        assert torch.equal(a[k][:M], b[k][:M]), k
    # single entries: every match, and a random sample of the matrix
    got = ops.dual_softmax_conf_at(f0, f1, 0.1, a['b_ids'][:M], a['i_ids'][:M], a['j_ids'][:M])
    assert torch.equal(got, a['mconf'][:M])
    bb = torch.randint(0, N, (4000,), generator=g); ii = torch.randint(0, L, (4000,), generator=g); jj = torch.randint(0, L, (4000,), generator=g)
    got = ops.dual_softmax_conf_at(f0, f1, 0.1, bb, ii, jj)
    want = a['conf_matrix'][bb.to(dev), ii.to(dev), jj.to(dev)]
    assert torch.equal(got, want)
    # the statistics in the workspace are tied to the call that left them - out-of-range indices, other feature
    # tensors (a second CoarseMatching pass on the same workspace) or another temperature give NaN, not silent garbage
    bad = ops.dual_softmax_conf_at(f0, f1, 0.1, torch.tensor([0, N, 0, 0]), torch.tensor([0, 0, L, -1]), torch.tensor([0, 0, 0, 0]))
    assert not torch.isnan(bad[0]) and bool(torch.isnan(bad[1:]).all())
    assert bool(torch.isnan(ops.dual_softmax_conf_at(f0, f1, 0.2, bb[:8], ii[:8], jj[:8])).all())
    f1b = f1.clone()
    assert bool(torch.isnan(ops.dual_softmax_conf_at(f0, f1b, 0.1, bb[:8], ii[:8], jj[:8])).all())
    ops.dual_softmax_match(f0, f1b, 0.1, thr, (h, w), (h, w), 8.0, materialize=False)      # the workspace now belongs to (f0, f1b)
    assert bool(torch.isnan(ops.dual_softmax_conf_at(f0, f1, 0.1, bb[:8], ii[:8], jj[:8])).all())
    assert torch.equal(ops.dual_softmax_conf_at(f0, f1b, 0.1, bb, ii, jj), want)
    # same ADDRESS, other CONTENT (the caching allocator hands a freed feature tensor's memory to the next one of the same
    # shape): the stamp carries a fingerprint of the content, so the stale statistics are refused ...
    f1b.mul_(1.25)
    assert bool(torch.isnan(ops.dual_softmax_conf_at(f0, f1b, 0.1, bb[:8], ii[:8], jj[:8])).all())
    f1b.copy_(f1)
    assert torch.equal(ops.dual_softmax_conf_at(f0, f1b, 0.1, bb, ii, jj), want)          # ... and accepted again with the content back
    # a non-contiguous view is refused (a silent copy would never match the stamped pointers: all-NaN)
    with pytest.raises(ValueError):
        ops.dual_softmax_conf_at(f0.transpose(1, 2).contiguous().transpose(1, 2), f1b, 0.1, bb[:8], ii[:8], jj[:8])
    # outside the configuration the library is built for, the wrapper materialises the matrix as usual
    c = ops.dual_softmax_match(f0[:, :100].contiguous(), f1, 0.1, thr, (10, 10), (h, w), 8.0, materialize=False)
    assert c['conf_matrix'] is not None


def test_model_match_only_flag():
    """geoformer_cfg['materialize_conf'] = False: conf matrices absent (None), every match output identical."""
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.geo_config import get_cfg_model
    dev = 'cuda:0'
    W = O.make_weights()
    outs = []
    for mat in (True, False):
        cfg = get_cfg_model(); cfg.update(precision='fp16', materialize_conf=mat)
        m = GeoFormer(get_default_cfg(), cfg).eval()
        m.load_state_dict(dict(W)); m = m.to(dev)
        (c0, f0), (c1, f1) = GI.planted_features(2, 16, 16, 16, 16, 102)
        data = {'image0': torch.zeros(2, 1, 128, 128, device=dev), 'image1': torch.zeros(2, 1, 128, 128, device=dev)}
        with torch.no_grad():
            outs.append(m.forward_features(data, c0.to(dev), f0.to(dev), c1.to(dev), f1.to(dev)))
    a, b = outs
    assert a['conf_matrix'] is not None and b['conf_matrix'] is None and b['dect_conf_matrix'] is None
    assert len(a['b_ids']) > 50
    for k in ('b_ids', 'i_ids', 'j_ids', 'm_bids', 'mkpts0_c', 'mkpts1_c', 'mkpts0_f', 'mkpts1_f', 'mconf'):
        assert torch.equal(a[k], b[k]), k
