"""Training step on the MI355X: device RANSAC inside the autograd forward, loss goes down on a fixed batch, and the
HIP inference path picks up the updated weights afterwards."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O

pytestmark = pytest.mark.gpu


def test_train_steps_reduce_loss_and_refresh_inference():
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.geo_config import get_cfg_model
    from geoformer_amd.train import TrainStep, synthetic_homography_batch
    g = get_cfg_model()
    g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
    model = GeoFormer(get_default_cfg(), g)
    sd = model.state_dict(); O.closed_form_fill(sd); model.load_state_dict(sd)
    model.cuda()
    step = TrainStep(model, trainer_cfg={'warmup_step': 0, 'canonical_lr': 2e-2}, batch_size=2)
    losses = []
    for _ in range(4):
        batch = synthetic_homography_batch(2, (128, 160), seed=7, device='cuda')
        losses.append(float(step(batch)))
        assert int(batch['conf_matrix_gt'].sum()) > 100 and len(batch['b_ids']) > 16
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # inference (HIP path) after training: same weights as an oracle forward on the CPU copy
    model.eval()
    pair = synthetic_homography_batch(1, (128, 160), seed=8, device='cuda')
    with torch.no_grad():
        out = model({'image0': pair['image0'], 'image1': pair['image1']})
    W = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    from ransac_oracle import make_homography_fn
    ref = O.geoformer_forward(W, {'image0': pair['image0'].cpu(), 'image1': pair['image1'].cpu()}, None,
                              dict(O.default_geo_config(), coarse_thr=0.0, fine_thr=0.0), make_homography_fn())
    torch.testing.assert_close(out['dect_conf_matrix'].cpu(), ref['dect_conf_matrix'], rtol=5e-3, atol=1e-7)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('weighted,masked', [(False, False), (True, False), (True, True)])
def test_fused_coarse_focal_loss_matches_autograd(dtype, weighted, masked):
    """gf_coarse_loss_forward/backward against torch autograd on the same fp16-rounded features (fp32 math):
    loss and per-positive confidences to 2e-3, gradients to 2e-2 (fp16 MFMA operands, fp32 accumulation)."""
    from geoformer_amd import ops
    torch.manual_seed(11)
    N, L, S, C, T = 2, 256, 384, 256, 0.1
    # planted but noisy correspondences: confidences of the positives spread over (0, 1), i.e. a state in which the
    # gradient is not the near-cancellation of its dense and sparse parts (at p -> 1 those two are equal and opposite
    # and ANY rounding of either dominates their difference)
    base = torch.randn(N, max(L, S), C, device='cuda') * 0.75
    f0 = base[:, :L].clone()
    perm = torch.stack([torch.randperm(S, device='cuda') for _ in range(N)])
    f1 = torch.gather(base[:, :S], 1, perm[..., None].expand(-1, -1, C)) + 0.6 * torch.randn(N, S, C, device='cuda')
    f0, f1 = f0.to(dtype), f1.to(dtype)
    # positives: every third row of f0, matched to where its feature went (when it exists), unique rows and columns
    inv = torch.argsort(perm, dim=1)
    pb = torch.arange(N, device='cuda').repeat_interleave(L // 3)
    pi = torch.arange(0, L - 2, 3, device='cuda').repeat(N)[:pb.numel()]
    pj = inv[pb, pi]
    w = torch.rand(pb.numel(), device='cuda') + 0.5 if weighted else None
    m0 = m1 = None
    if masked:                      # zero-padded borders: the last rows of image0's grid, a block of image1's
        m0 = torch.ones(N, L, dtype=torch.bool, device='cuda'); m0[:, L - 40:] = False
        m1 = torch.ones(N, S, dtype=torch.bool, device='cuda'); m1[0, 100:170] = False; m1[1, :33] = False
        keep = m0[pb, pi] & m1[pb, pj]
        pb, pi, pj, w = pb[keep], pi[keep], pj[keep], w[keep]

    a0 = f0.detach().half().float().requires_grad_(True)
    a1 = f1.detach().half().float().requires_grad_(True)
    sim = torch.einsum('nlc,nsc->nls', a0 / C ** .5, a1 / C ** .5) / T
    if masked:
        sim = sim.masked_fill(~(m0[..., None] & m1[:, None]), -1e9)
    conf = torch.softmax(sim, 1) * torch.softmax(sim, 2)
    p = torch.clamp(conf, 1e-6, 1 - 1e-6)[pb, pi, pj]
    terms = -0.25 * (1 - p) ** 2.0 * p.log()
    ref = (terms * w).sum() if weighted else terms.sum()
    (ref * 0.37).backward()

    h0, h1 = f0.detach().clone().requires_grad_(True), f1.detach().clone().requires_grad_(True)
    loss, pk = ops.coarse_focal_loss(h0, h1, pb, pi, pj, T, 0.25, 2.0, w, m0, m1)
    (loss * 0.37).backward()
    assert 0.02 < float(pk.median()) < 0.98, float(pk.median())
    torch.testing.assert_close(pk, conf[pb, pi, pj].detach(), rtol=2e-3, atol=1e-7)
    torch.testing.assert_close(loss.detach(), ref.detach(), rtol=2e-3, atol=1e-6)
    for got, want in ((h0.grad.float(), a0.grad), (h1.grad.float(), a1.grad)):
        rel = (got - want).norm() / want.norm()
        assert rel < 2e-2, float(rel)
        assert (got - want).abs().max() < 3e-2 * want.abs().max()


@pytest.mark.parametrize('masked', [False, True])
def test_train_step_fused_loss_equals_autograd_path(masked):
    """Same batch, same weights: the step with the fused HIP coarse loss against the step that differentiates the
    materialised confidence matrices (loss terms to 2e-3, parameter gradients to 3e-2 in norm)."""
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.geo_config import get_cfg_model
    from geoformer_amd.train import GeoLoss, forward_train, spvs_coarse, spvs_fine2, synthetic_homography_batch
    g = get_cfg_model()
    g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
    model = GeoFormer(get_default_cfg(), g)
    sd = model.state_dict(); O.closed_form_fill(sd); model.load_state_dict(sd)
    model.cuda().train()
    loss_fn = GeoLoss()
    grads, scalars = [], []
    for fused in (None, loss_fn.fused_params()):
        batch = synthetic_homography_batch(2, (128, 256), seed=21, device='cuda')      # 16 x 32 = 512 coarse cells
        if masked:                  # MegaDepth-style zero padding: bottom rows of image0, right columns of image1
            batch['mask0'] = torch.ones(2, 16, 32, dtype=torch.bool, device='cuda'); batch['mask0'][:, 13:] = False
            batch['mask1'] = torch.ones(2, 16, 32, dtype=torch.bool, device='cuda'); batch['mask1'][:, :, 27:] = False
        spvs_coarse(batch)
        forward_train(model, batch, fused_coarse_loss=fused)
        spvs_fine2(batch)
        model.zero_grad(set_to_none=True)
        loss_fn(batch).backward()
        scalars.append({k: float(v) for k, v in batch['loss_scalars'].items()})
        grads.append({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        assert ('loss_c_fused' in batch) == (fused is not None)
    for k in ('loss_c', 'loss_d', 'loss'):
        assert scalars[1][k] == pytest.approx(scalars[0][k], rel=2e-3), (k, scalars)
    num = sum(((grads[1][n] - grads[0][n]) ** 2).sum() for n in grads[0])
    den = sum((grads[0][n] ** 2).sum() for n in grads[0])
    assert float(torch.sqrt(num / den)) < 3e-2


def test_fused_coarse_focal_loss_full_size():
    """The coarse level's real size (one pair, L = S = 6400): fused HIP loss against autograd."""
    from geoformer_amd import ops
    torch.manual_seed(12)
    N, L, S, C, T = 1, 6400, 6400, 256, 0.1
    f0 = torch.randn(N, L, C, device='cuda') * 0.75
    perm = torch.randperm(S, device='cuda')[None]
    f1 = (f0[:, perm[0]] + 0.6 * torch.randn(N, S, C, device='cuda')).half()
    f0 = f0.half()
    inv = torch.argsort(perm, dim=1)
    pi = torch.arange(0, L, 2, device='cuda')
    pb = torch.zeros_like(pi)
    pj = inv[0, pi]
    a0, a1 = f0.float().requires_grad_(True), f1.float().requires_grad_(True)
    sim = torch.einsum('nlc,nsc->nls', a0 / C ** .5, a1 / C ** .5) / T
    conf = torch.softmax(sim, 1) * torch.softmax(sim, 2)
    p = torch.clamp(conf, 1e-6, 1 - 1e-6)[pb, pi, pj]
    ref = (-0.25 * (1 - p) ** 2.0 * p.log()).mean()
    ref.backward()
    h0, h1 = f0.clone().requires_grad_(True), f1.clone().requires_grad_(True)
    loss, pk = ops.coarse_focal_loss(h0, h1, pb, pi, pj, T)
    (loss / pi.numel()).backward()
    torch.testing.assert_close(loss.detach() / pi.numel(), ref.detach(), rtol=2e-3, atol=1e-7)
    for got, want in ((h0.grad.float(), a0.grad), (h1.grad.float(), a1.grad)):
        assert float((got - want).norm() / want.norm()) < 2e-2
