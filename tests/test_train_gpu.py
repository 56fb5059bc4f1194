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
