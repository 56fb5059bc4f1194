"""the K3 chain's backward in HIP (csrc/k_train.hip, train/hip_autograd.py) against torch autograd of the
same arithmetic on the same rounded operands, and the mixed-bf16 training step with it against the autocast step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import geoformer_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('T,cout,cin', [(12800, 256, 256), (6400 + 37, 512, 512), (100, 128, 256), (31, 256, 128), (5000, 224, 128), (777, 256, 224), (300, 8, 136)])
def test_linear_wgrad(dtype, T, cout, cin):
    """dW = dY^T X with fp32 accumulation: against the fp64 product of the same 16-bit operands (ragged token counts, all the
    layer widths); a column block of a wider gradient (the torch.cat([x, m]) halves); accumulation."""
    from geoformer_amd import ops
    g = torch.Generator().manual_seed(T)
    dy = torch.randn(T, cout, generator=g).to(DEV, dtype)
    x = torch.randn(T, cin, generator=g).to(DEV, dtype)
    want = (dy.double().t() @ x.double())
    got = ops.linear_wgrad(dy, x)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale + 1e-4 * np.sqrt(T) * 1e-2
    # strided operands (views of wider buffers) into a column block, then accumulated once more
    wide = torch.zeros(cout, cin + 128, device=DEV)
    xw = torch.cat([x, x], 1)
    ops.linear_wgrad(dy, xw[:, cin:], out=wide[:, 128:])
    assert torch.equal(wide[:, 128:], got) and float(wide[:, :128].abs().max()) == 0.0
    ops.linear_wgrad(dy, x, out=wide[:, 128:], accumulate=True)
    torch.testing.assert_close(wide[:, 128:], 2 * got, rtol=1e-6, atol=1e-6)
    # deterministic
    assert torch.equal(ops.linear_wgrad(dy, x), got)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('C', [128, 256, 512])
def test_layernorm_forward_backward(dtype, C):
    from geoformer_amd import ops
    g = torch.Generator().manual_seed(C)
    T = 1000 + C // 64
    y = (torch.randn(T, C, generator=g) * 1.7 + 0.3).to(DEV, dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    dout = torch.randn(T, C, generator=g).to(DEV, dtype)
    out, stats = ops.layernorm_forward(y, gamma, beta, 1e-5)
    yr = y.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(yr, (C,), gr, br, 1e-5)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert float((out.float() - ref.detach()).abs().max()) <= ulp * float(ref.detach().abs().max())
    torch.testing.assert_close(stats[:, 0], y.float().mean(1), rtol=1e-5, atol=1e-5)
    ref.backward(dout.float())
    dy, dg, db = ops.layernorm_backward(dout, y, stats, gamma)
    assert float((dy.float() - yr.grad).abs().max()) <= 2 * ulp * float(yr.grad.abs().max())
    torch.testing.assert_close(dg, gr.grad, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(db, br.grad, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize('kind', ['relu', 'tanh'])
def test_hip_linear_function_against_autograd(kind):
    """HipLinear (two-part input, fused activation) and HipLayerNorm chained as in an encoder layer: outputs and all gradients
    against torch autograd of the same chain on the same bf16-rounded operands (fp32 master weights)."""
    from geoformer_amd.train import hip_autograd as HA
    g = torch.Generator().manual_seed(5)
    T, c = 3000, 256
    x = (torch.randn(2, T // 2, c, generator=g) * 0.8).to(DEV)
    m = (torch.randn(2, T // 2, c, generator=g) * 0.8).to(DEV)
    w1 = (torch.randn(2 * c, 2 * c, generator=g) / np.sqrt(2 * c)).to(DEV)
    w2 = (torch.randn(c, 2 * c, generator=g) / np.sqrt(2 * c)).to(DEV)
    gam = (1 + 0.1 * torch.randn(c, generator=g)).to(DEV); bet = (0.1 * torch.randn(c, generator=g)).to(DEV)
    act = torch.relu if kind == 'relu' else torch.tanh
    dt = torch.bfloat16

    def run(hip):
        ps = [t.clone().requires_grad_(True) for t in (x, m, w1, w2, gam, bet)]
        xx, mm, a, b, ga, be = ps
        if hip:
            h = HA.linear(xx.to(dt), a, mm.to(dt), kind)
            o = HA.layer_norm(HA.linear(h, b), ga, be)
        else:
            h = act(F.linear(torch.cat([xx.to(dt), mm.to(dt)], 2), a.to(dt)))
            o = F.layer_norm(F.linear(h, b.to(dt)).float(), (c,), ga, be).to(dt)
        loss = (o.float() * torch.linspace(-1, 1, c, device=DEV)).sum() / T
        loss.backward()
        return o.detach().float(), [p.grad.float() for p in ps]

    o_ref, g_ref = run(False)
    o_hip, g_hip = run(True)
    assert float((o_hip - o_ref).abs().max()) <= 2 * 2.0 ** -7 * float(o_ref.abs().max())
    for name, a, b in zip(('x', 'm', 'w1', 'w2', 'gamma', 'beta'), g_hip, g_ref):
        rel = float((a - b).norm() / b.norm())
        assert rel < 2e-2, (name, rel)                           # bf16 intermediates round at different points: 1e-2 in norm


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('L,S,masked', [(640, 640, False), (300, 437, True), (130, 64, False)])
def test_linear_attention_backward(dtype, L, S, masked):
    """gf_linear_attention_backward against torch autograd of the training restatement of LinearAttention.forward
    (train/functional.py:linear_attention, fp32 arithmetic) on the same 16-bit q, k, v, dout: dq, dk, dv to 2e-2 in norm and 3 ulp of
    the storage type on the large entries (ragged chunks, padding masks on both sides, row-strided views of a fused projection)."""
    from geoformer_amd import ops
    from geoformer_amd.train import functional as TF
    g = torch.Generator().manual_seed(L + S)
    N, H, D = 2, 8, 32
    C = H * D
    q = (torch.randn(N, L, C, generator=g) * 0.8).to(DEV, dtype)
    kvbuf = (torch.randn(N, S, 2 * C, generator=g) * 0.8).to(DEV, dtype)
    k, v = kvbuf[..., :C], kvbuf[..., C:]                              # strided views (ld = 2 C)
    dout = torch.randn(N, L, C, generator=g).to(DEV, dtype)
    qm = km = None
    if masked:
        qm = torch.ones(N, L, dtype=torch.bool); qm[1, L // 2:] = False
        km = torch.ones(N, S, dtype=torch.bool); km[0, S - 50:] = False; km[1, :17] = False
        qm, km = qm.to(DEV), km.to(DEV)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    out = TF.linear_attention(qf.view(N, L, H, D), kf.view(N, S, H, D), vf.view(N, S, H, D), qm, km)
    out.backward(dout.float().view(N, L, H, D))
    dq, dk, dv = ops.linear_attention_backward(q, k, v, dout, H, qm, km)
    for name, got, want in (('dq', dq, qf.grad), ('dk', dk, kf.grad), ('dv', dv, vf.grad)):
        rel = float((got.float() - want).norm() / want.norm())
        assert rel < 2e-2, (name, rel)
        if masked:
            dead = ~(qm if name == 'dq' else km)
            assert float(got[dead].abs().max()) == 0.0, name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fine_match_backward(dtype):
    """gf_fine_match_backward against autograd of the training restatement of FineMatching2's confidence (train/functional.py:dual_softmax)
    for a random dconf, and the HipFineMatch Function's outputs against functional.fine_match (same matches, keypoints, matrix)."""
    from geoformer_amd import ops
    from geoformer_amd.train import functional as TF
    import golden_inputs as GI
    I = GI.g9_inputs()
    f0, f1 = I['f0'].to(DEV, dtype), I['f1'].to(DEV, dtype)
    g = torch.Generator().manual_seed(3)
    dconf = torch.randn(f0.shape[0], 25, 25, generator=g).to(DEV)
    a, b = f0.float().clone().requires_grad_(True), f1.float().clone().requires_grad_(True)
    TF.dual_softmax(a, b, I['temperature']).backward(dconf)
    df0, df1 = ops.fine_match_backward(f0, f1, I['temperature'], dconf)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    for name, got, want in (('df0', df0, a.grad), ('df1', df1, b.grad)):
        rel = float((got.float() - want).norm() / want.norm())
        assert rel < tol, (name, rel)
    if dtype == torch.float32:
        data = {'hw0_i': torch.tensor(I['hw0_i']), 'hw0_c': torch.tensor(I['hw0_c']), 'hw0_f': torch.tensor(I['hw0_f']), 'b_ids': I['b_ids'].to(DEV),
                'mkpts0_c': I['mkpts0_c'].to(DEV), 'mkpts1_c': I['mkpts1_c'].to(DEV)}
        ref = TF.fine_match(f0, f1, data, I['temperature'], I['thr'])
        TF.set_hip_backward(True)
        try:
            got = TF.fine_match(f0.clone().requires_grad_(True), f1.clone().requires_grad_(True), data, I['temperature'], I['thr'])
        finally:
            TF.set_hip_backward(False)
        torch.testing.assert_close(got['fine_matrix'], ref['fine_matrix'], rtol=2e-5, atol=1e-9)
        assert torch.equal(got['m_bids'], ref['m_bids'])
        torch.testing.assert_close(got['mkpts0_f'], ref['mkpts0_f'].float(), rtol=1e-6, atol=1e-5)
        torch.testing.assert_close(got['mkpts1_f'], ref['mkpts1_f'].float(), rtol=1e-6, atol=1e-5)
        assert got['fine_matrix'].requires_grad and not got['mkpts0_f'].requires_grad


def test_hip_backward_train_step_matches_autocast_step():
    """TrainStep(precision='bf16', hip_backward=True) against the autocast bf16 step on the same batch and weights: loss terms
    within 2 %, parameter gradients of the first coarse term aligned (cosine > 0.97), fp32 parameters, loss decreasing."""
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.geo_config import get_cfg_model
    from geoformer_amd.train import TrainStep, synthetic_homography_batch
    res = {}
    for hip in (False, True):
        g = get_cfg_model()
        g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
        model = GeoFormer(get_default_cfg(), g)
        sd = model.state_dict(); O.closed_form_fill(sd); model.load_state_dict(sd)
        model.cuda()
        step = TrainStep(model, trainer_cfg={'warmup_step': 0, 'canonical_lr': 1e-2, 'gradient_clipping': 0.0}, batch_size=2,
                         fused_coarse_loss=True, precision='bf16', hip_backward=hip)
        batch = synthetic_homography_batch(2, (128, 256), seed=31, device='cuda')
        from geoformer_amd.train.functional import set_hip_backward
        set_hip_backward(hip)
        try:
            step.core(batch)
        finally:
            set_hip_backward(False)
        from geoformer_amd.train.hip_autograd import WEIGHTS
        step.optimizer.zero_grad(set_to_none=True)
        first = batch['loss_d_fused'][0] / batch['loss_d_fused'][1]
        first.backward()
        WEIGHTS.clear()
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        assert all(p.dtype == torch.float32 for p in model.parameters()) and all(v.dtype == torch.float32 for v in grads.values())
        scal = {k: float(v) for k, v in batch['loss_scalars'].items()}
        losses = [float(step(synthetic_homography_batch(2, (128, 256), seed=31, device='cuda'))) for _ in range(3)]
        res[hip] = (scal, grads, losses)
    (s0, g0, l0), (s1, g1, l1) = res[False], res[True]
    print(f'autocast {s0} losses {l0} | hip backward {s1} losses {l1}')
    for k in ('loss_c', 'loss_d'):
        # 4 %: the autocast step alone spreads by 1.7 % from run to run on one box (loss_c 3.085 / 3.091 / 3.116 / 3.136 in four runs of round 6;
        # the library's kernels are not bit-reproducible), so the 2 % of earlier rounds failed one run in a few
        assert s1[k] == pytest.approx(s0[k], rel=4e-2), (k, s0, s1)
    common = [n for n in g0 if n in g1 and n.startswith('loftr_coarse')]
    assert len(common) >= 40
    dot = sum(float((g0[n] * g1[n]).sum()) for n in common)
    na, nb = (sum(float((g[n] ** 2).sum()) for n in common) ** 0.5 for g in (g0, g1))
    print(f'cosine of the coarse-loss gradients (autocast vs HIP backward) {dot / (na * nb):.4f}')
    assert dot / (na * nb) > 0.97
    assert all(np.isfinite(l1)) and l1[-1] < l1[0], l1


def test_hip_functions_follow_a_hand_stepped_optimizer():
    """functional.set_hip_backward(True) WITHOUT TrainStep (nobody clears the weight cache): after every
    optimizer step the HIP linear must multiply by the UPDATED weight - its output tracks F.linear of the current bf16 cast,
    and three SGD steps move the loss exactly as the same three steps through torch autograd do (same bf16 operands)."""
    from geoformer_amd.train import hip_autograd as HA
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(2, 256, 256, generator=g)).to(DEV).to(torch.bfloat16)
    tgt = torch.randn(2, 256, 128, generator=g).to(DEV)
    w0 = (torch.randn(128, 256, generator=g) / 16).to(DEV)

    def run(hip):
        w = torch.nn.Parameter(w0.clone())
        opt = torch.optim.SGD([w], lr=0.05)
        losses = []
        for _ in range(3):
            y = HA.linear(x, w) if hip else F.linear(x, w.to(torch.bfloat16))
            if hip:                                                        # the product uses the weight of THIS step
                torch.testing.assert_close(y.float(), F.linear(x, w.detach().to(torch.bfloat16)).float(), rtol=2e-2, atol=2e-2)
            loss = ((y.float() - tgt) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        return losses, w.detach().clone()
    HA.WEIGHTS.clear()
    l_hip, w_hip = run(True)
    l_ref, w_ref = run(False)
    assert l_hip[2] < l_hip[1] < l_hip[0]
    np.testing.assert_allclose(l_hip, l_ref, rtol=2e-3)
    assert float((w_hip - w_ref).norm() / (w_ref - w0).norm()) < 2e-2


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('Lw', [25, 32, 7])
def test_window_linear_attention_backward(dtype, Lw):
    """The fine level's window linear attention (8 heads of 16, <= 32 tokens, no masks): HipWindowLinearAttention - forward K2's
    window form, backward gf_window_linear_attention_backward - against autograd of the functional restatement
    (linear_attention.py:21-51) on the same 16-bit tensors: output to the storage type's resolution, dq / dk / dv to 2e-2 in norm."""
    from geoformer_amd.train import functional as TF
    from geoformer_amd.train import hip_autograd as HA
    g = torch.Generator().manual_seed(Lw)
    Nw = 300
    q, k, v = ((torch.randn(Nw, Lw, 128, generator=g) * 0.8).to(DEV).to(dtype) for _ in range(3))
    dout = (torch.randn(Nw, Lw, 128, generator=g) * 0.5).to(DEV).to(dtype)
    a = [t.clone().requires_grad_(True) for t in (q, k, v)]
    ref = TF.linear_attention(*(t.view(Nw, Lw, 8, 16) for t in a)).reshape(Nw, Lw, 128)
    ref.backward(dout)
    b = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out = HA.window_linear_attention(*b, 8)
    out.backward(dout)
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    assert float((out.float() - ref.float()).abs().max()) < 4 * ulp * float(ref.float().abs().max())
    for name, x, y in zip('qkv', b, a):
        rel = float((x.grad.float() - y.grad.float()).norm() / y.grad.float().norm())
        assert rel < 2e-2, (name, rel)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_window_cross_attention_backward(dtype):
    """GeoTransformer's 'cross' attention (25 window keys per query, 4 heads of 64; geo_attention.py:72-101) on the projected maps:
    HipWindowCrossAttention - forward K5, backward gf_window_cross_attention_backward (fp32 scatter-adds for the overlapping
    windows) - against autograd of the functional restatement (gather the window rows, full_attention with the window mask) on the
    same tensors; masked window positions, a query without any valid key, repeated cells."""
    from geoformer_amd.train import functional as TF
    from geoformer_amd.train import hip_autograd as HA
    g = torch.Generator().manual_seed(23)
    hq, wq, hk, wk = 20, 24, 18, 26
    L, S, C = hq * wq, hk * wk, 256
    q = (torch.randn(1, L, C, generator=g) * 0.8).to(DEV).to(dtype)
    km = (torch.randn(1, S, C, generator=g) * 0.8).to(DEV).to(dtype)
    vm = (torch.randn(1, S, C, generator=g) * 0.8).to(DEV).to(dtype)
    dout = (torch.randn(1, L, C, generator=g) * 0.5).to(DEV).to(dtype)
    # windows: 5 x 5 neighbourhoods around a shifted position (overlapping between neighbouring queries), some positions outside
    ys, xs = torch.meshgrid(torch.arange(hq), torch.arange(wq), indexing='ij')
    cy, cx = (ys.flatten() * hk) // hq + 1, (xs.flatten() * wk) // wq - 1
    dy, dx = torch.meshgrid(torch.arange(-2, 3), torch.arange(-2, 3), indexing='ij')
    yy, xx = cy[:, None] + dy.flatten()[None], cx[:, None] + dx.flatten()[None]
    ok = (yy >= 0) & (yy < hk) & (xx >= 0) & (xx < wk)
    win = torch.where(ok, yy * wk + xx, -1)
    win[7] = -1                                           # a query without any valid key
    win[11, :5] = win[11, 5]                              # repeated cells inside one window
    win = win.to(torch.int32).to(DEV)[None].contiguous()
    a = [t.clone().requires_grad_(True) for t in (q, km, vm)]
    mask = (win[0] >= 0)
    idx = win[0].clamp_min(0).long()
    kg = a[1][0].index_select(0, idx.reshape(-1)).view(L, 25, 4, 64)
    vg = a[2][0].index_select(0, idx.reshape(-1)).view(L, 25, 4, 64)
    ref = TF.full_attention(a[0][0].view(L, 1, 4, 64), kg, vg, mask).reshape(1, L, C)
    ref.backward(dout)
    b = [t.clone().requires_grad_(True) for t in (q, km, vm)]
    out = HA.window_cross_attention(*b, win, 4)
    out.backward(dout)
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    assert float((out.float() - ref.float()).abs().max()) < tol * max(1.0, float(ref.float().abs().max()))
    assert float(out[0, 7].abs().max()) == 0.0 and float(b[0].grad[0, 7].abs().max()) == 0.0
    for name, x, y in zip(('dq', 'dk', 'dv'), b, a):
        rel = float((x.grad.float() - y.grad.float()).norm() / y.grad.float().norm())
        assert rel < (1e-4 if dtype == torch.float32 else 2e-2), (name, rel)
    # round 6: dk / dv are gathered along the inverse window table (no atomics): a second run gives the same bits, and the scatter form
    # (fp32 atomic adds, kept for A/B) agrees to the rounding of the storage type
    b2 = [t.clone().requires_grad_(True) for t in (q, km, vm)]
    HA.window_cross_attention(*b2, win, 4).backward(dout)
    assert all(torch.equal(x.grad, y.grad) for x, y in zip(b, b2))
    HA._GATHER_K5_BACKWARD[0] = False
    try:
        b3 = [t.clone().requires_grad_(True) for t in (q, km, vm)]
        HA.window_cross_attention(*b3, win, 4).backward(dout)
    finally:
        HA._GATHER_K5_BACKWARD[0] = True
    for x, y in zip(b, b3):
        assert float((x.grad.float() - y.grad.float()).norm() / y.grad.float().norm()) < (1e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('N,L,S', [(1, 6400, 1195), (1, 300, 33), (2, 130, 64), (1, 37, 1), (1, 128, 32)])
def test_full_attention_train_forward_backward(dtype, N, L, S):
    """GeoTransformer's 'self' attention of the training step (all L cells against the S projected inlier rows, 4 heads of 64, no masks;
    geo_attention.py:72-101): HipFullAttention - gf_full_attention_train_forward / gf_full_attention_backward, flash form - against
    autograd of the oracle's explicit full_attention in fp32 on the same 16-bit inputs; ragged query and key tiles, one key, a batch,
    row-strided views (q, k, v as slices of one fused projection).  The backward is bit-reproducible (no atomics)."""
    import geoformer_oracle as O
    from geoformer_amd.train import hip_autograd as HA
    g = torch.Generator().manual_seed(5 + L + S)
    C = 256
    x = (torch.randn(N, L, 3 * C, generator=g) * 1.5).to(DEV).to(dtype)
    q = x[..., :C]                                         # a row-strided view (ld = 768)
    k = (torch.randn(N, S, C, generator=g) * 1.5).to(DEV).to(dtype)
    v = (torch.randn(N, S, C, generator=g)).to(DEV).to(dtype)
    dout = (torch.randn(N, L, C, generator=g) * 0.5).to(DEV).to(dtype)
    a = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    ref = O.full_attention(a[0].view(N, L, 4, 64), a[1].view(N, S, 4, 64), a[2].view(N, S, 4, 64)).reshape(N, L, C)
    ref.backward(dout.float())
    b = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out = HA.full_attention(*b, 4)
    out.backward(dout)
    eps = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
    # forward: P rounded to the storage type in the second product, the output rounded once
    assert float((out.float() - ref).abs().max()) < 3 * eps * max(1.0, float(ref.abs().max()))
    for name, x_, y_ in zip(('dq', 'dk', 'dv'), b, a):
        floor = 1e-2 * float(dout.float().norm())          # one key: dq is exactly zero in the reference, rounding noise here
        rel = float((x_.grad.float() - y_.grad).norm() / y_.grad.norm().clamp_min(floor))
        assert rel < 4 * eps, (name, rel)
    # the saved statistics: base-2 log-sum-exp of the scaled logits
    from geoformer_amd import ops
    _, lse = ops.full_attention_train_forward(q, k, v, 4)
    logits = torch.einsum('nlhd,nshd->nhls', q.float().view(N, L, 4, 64), k.float().view(N, S, 4, 64)) * 0.125
    want = torch.logsumexp(logits, dim=-1) * 1.4426950408889634
    assert float((lse - want).abs().max()) < 2e-3
    # reproducible
    b2 = [t.clone().requires_grad_(True) for t in (q, k, v)]
    HA.full_attention(*b2, 4).backward(dout)
    assert all(torch.equal(x_.grad, y_.grad) for x_, y_ in zip(b, b2))


def test_full_attention_train_no_keys():
    """S = 0 (an image without inliers): the forward writes zeros, every gradient is zero."""
    from geoformer_amd.train import hip_autograd as HA
    q = torch.randn(1, 70, 256, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    k = torch.zeros(1, 0, 256, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    v = torch.zeros(1, 0, 256, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    out = HA.full_attention(q, k, v, 4)
    out.backward(torch.ones_like(out))
    assert float(out.abs().max()) == 0.0 and float(q.grad.abs().max()) == 0.0 and k.grad.shape == (1, 0, 256)


def test_batched_geo_layers_equal_the_image_by_image_form():
    """Round 6: the HIP step runs the Geo layers' per-token operations once per layer on all images of the batch (functional._BATCHED_GEO) and
    only the attention cores per image.  functional.geo_module on fixed feature maps, matches and homographies (the whole step is not
    bit-reproducible upstream: the library's train-mode convolutions), batched against image by image: outputs bit-equal (the per-token
    arithmetic does not depend on the batching), input gradients equal to 16-bit rounding of a handful of elements (K5's backward adds with
    atomics), the Geo layers' parameter gradients aligned (sums over other chunks of rows).  Three samples: one with a homography and 60
    inlier cells, one with 5 matches (no RANSAC: self layers only), one without any match (every layer skipped)."""
    from geoformer_amd.train import functional as TF
    from geoformer_amd.train.hip_autograd import WEIGHTS
    h, w, n = 20, 24, 3
    g = torch.Generator().manual_seed(9)
    W = O.make_weights()
    cells = torch.randperm((h - 2) * (w - 2), generator=g)[:60]
    mk0 = torch.stack([cells % (w - 2), cells // (w - 2)], 1).float() * 8
    mk0b = torch.tensor([[0., 0.], [8., 16.], [24., 8.], [56., 40.], [40., 32.]])
    data0 = {'image0': torch.zeros(n, 1, h * 8, w * 8), 'image1': torch.zeros(n, 1, h * 8, w * 8), 'hw0_i': torch.tensor([h * 8, w * 8]),
             'hw0_c': torch.tensor([h, w]), 'mkpts0_c': torch.cat([mk0, mk0b]).cuda(), 'mkpts1_c': torch.cat([mk0 + 8, mk0b.flip(0)]).cuda(),
             'm_bids': torch.cat([torch.zeros(60, dtype=torch.long), torch.ones(5, dtype=torch.long)]).cuda()}
    M = torch.tensor([[1., 0, 8], [0, 1, 8], [0, 0, 1]], dtype=torch.float64)
    hfn = lambda b, kp0, kp1: (M, torch.ones(len(kp0), dtype=torch.bool, device=kp0.device))
    c0 = torch.randn(n, 256, h, w, generator=g).cuda()
    c1 = torch.randn(n, 256, h, w, generator=g).cuda()
    dout = [torch.randn(n, h * w, 256, generator=g).cuda() for _ in range(2)]
    res = {}
    for batched in (False, True):
        P = {k: v.detach().clone().cuda().requires_grad_(True) for k, v in W.items() if k.startswith('geo_module.')}
        a0, a1 = c0.clone().requires_grad_(True), c1.clone().requires_grad_(True)
        TF._BATCHED_GEO[0] = batched
        TF.set_hip_backward(True)
        try:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                o0, o1 = TF.geo_module(P, a0, a1, dict(data0), O.default_geo_config(), hfn)
            (o0 * dout[0]).sum().add((o1 * dout[1]).sum()).backward()
        finally:
            TF.set_hip_backward(False)
            TF._BATCHED_GEO[0] = True
            WEIGHTS.clear()
        res[batched] = (o0.detach(), o1.detach(), a0.grad, a1.grad, {k: v.grad for k, v in P.items() if v.grad is not None})
    r0, r1 = res[False], res[True]
    assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])
    for k in (2, 3):
        d = (r0[k] - r1[k]).abs().max() / r0[k].abs().max()
        assert float(d) < 2e-2, (k, float(d))
        assert float((r0[k] - r1[k]).norm() / r0[k].norm()) < 1e-2          # measured 3e-3: bf16 gradients, K5's atomic adds
    names = [k for k in r0[4] if k in r1[4]]
    assert len(names) >= 20 and len(names) == len(r0[4])
    dot = sum(float((r0[4][k] * r1[4][k]).sum()) for k in names)
    na, nb = (sum(float((r[4][k] ** 2).sum()) for k in names) ** 0.5 for r in (r0, r1))
    print(f'Geo layers batched against image by image: outputs bit-equal, cosine of the parameter gradients {dot / (na * nb):.6f}, norms {na:.4e} {nb:.4e}')
    assert dot / (na * nb) > 0.9995 and abs(na - nb) < 1e-2 * na
    # the sample without matches: every layer skipped - its rows leave as they came (position encoding added), its gradient is dout
    pe = TF.add_pe(c0[2:3]).flatten(2).transpose(1, 2)[0]
    assert torch.equal(r1[0][2], pe) and torch.equal(r1[2][2], dout[0][2].transpose(0, 1).reshape(256, h, w))
