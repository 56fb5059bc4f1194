"""GeoLoss (model/loftr_src/losses/loftr_loss.py:195-395): focal / cross-entropy loss on both coarse confidence
matrices (`conf_matrix` after the GeoModule and `dect_conf_matrix` before it) plus the cross-entropy on the fine
25x25 matrices.  Config keys are the lower-cased `LOFTR.LOSS.*` / `LOFTR.MATCH_COARSE.*` of the reference."""
import torch
import torch.nn as nn

DEFAULT_LOSS_CFG = {          # model/loftr_src/config/default.py:48-63
    'coarse_type': 'focal', 'coarse_weight': 1.0, 'focal_alpha': 0.25, 'focal_gamma': 2.0, 'pos_weight': 1.0,
    'neg_weight': 1.0, 'fine_type': 'l2_with_std', 'fine_weight': 1.0, 'fine_correct_thr': 1.0,
}


class GeoLoss(nn.Module):
    def __init__(self, loss_cfg=None, match_type='dual_softmax', sparse_spvs=True):
        super().__init__()
        self.cfg = dict(DEFAULT_LOSS_CFG, **(loss_cfg or {}))
        if match_type != 'dual_softmax':
            raise NotImplementedError('GeoFormer trains with dual-softmax matching only')
        self.sparse_spvs = sparse_spvs
        self.c_pos_w, self.c_neg_w = self.cfg['pos_weight'], self.cfg['neg_weight']

    def compute_coarse_loss(self, conf, conf_gt, weight=None):
        """loftr_loss.py:210-284."""
        pos_mask, neg_mask = conf_gt == 1, conf_gt == 0
        c_pos_w, c_neg_w = self.c_pos_w, self.c_neg_w
        if not pos_mask.any():                       # no GT match at all: a dummy positive with zero weight
            pos_mask[0, 0, 0] = True
            if weight is not None:
                weight[0, 0, 0] = 0.
            c_pos_w = 0.
        if not neg_mask.any():
            neg_mask[0, 0, 0] = True
            if weight is not None:
                weight[0, 0, 0] = 0.
            c_neg_w = 0.
        conf = torch.clamp(conf, 1e-6, 1 - 1e-6)
        if self.cfg['coarse_type'] == 'cross_entropy':
            if self.sparse_spvs:
                raise AssertionError('Sparse Supervision for cross-entropy not implemented!')
            loss_pos, loss_neg = -torch.log(conf[pos_mask]), -torch.log(1 - conf[neg_mask])
            if weight is not None:
                loss_pos, loss_neg = loss_pos * weight[pos_mask], loss_neg * weight[neg_mask]
            return c_pos_w * loss_pos.mean() + c_neg_w * loss_neg.mean()
        if self.cfg['coarse_type'] != 'focal':
            raise ValueError(f"Unknown coarse loss: {self.cfg['coarse_type']}")
        alpha, gamma = self.cfg['focal_alpha'], self.cfg['focal_gamma']
        pos_conf = conf[pos_mask]
        loss_pos = -alpha * torch.pow(1 - pos_conf, gamma) * pos_conf.log()
        if self.sparse_spvs:                         # dual-softmax has no dustbin: unmatched cells are unsupervised
            if weight is not None:
                loss_pos = loss_pos * weight[pos_mask]
            return c_pos_w * loss_pos.mean()
        neg_conf = conf[neg_mask]
        loss_neg = -alpha * torch.pow(neg_conf, gamma) * (1 - neg_conf).log()
        if weight is not None:
            loss_pos, loss_neg = loss_pos * weight[pos_mask], loss_neg * weight[neg_mask]
        return c_pos_w * loss_pos.mean() + c_neg_w * loss_neg.mean()

    def compute_fine_loss(self, conf, conf_gt):
        """loftr_loss.py:286-296."""
        pos_mask, neg_mask = conf_gt == 1, conf_gt == 0
        conf = torch.clamp(conf, 1e-6, 1 - 1e-6)
        loss_pos = (-torch.log(conf[pos_mask])).mean()
        loss_neg = (-torch.log(1 - conf[neg_mask])).mean()
        if torch.isnan(loss_neg):
            return loss_pos
        if torch.isnan(loss_pos):
            return loss_neg
        return self.c_pos_w * loss_pos + self.c_neg_w * loss_neg

    @torch.no_grad()
    def compute_c_weight(self, data):
        if 'mask0' in data:
            return (data['mask0'].flatten(-2)[..., None] * data['mask1'].flatten(-2)[:, None]).float()
        return None

    def fused_params(self):
        """(alpha, gamma) if this loss is the configuration the fused HIP coarse loss implements, else None."""
        return (self.cfg['focal_alpha'], self.cfg['focal_gamma']) if (self.cfg['coarse_type'] == 'focal' and self.sparse_spvs) else None

    def forward(self, data):
        """loftr_loss.py:353-395: loss = (loss_c + loss_d)*coarse_weight + loss_f*fine_weight."""
        if 'loss_c_fused' in data:               # forward_train(..., fused_coarse_loss=...): HIP loss terms (sum, count)
            (sc, nc), (sd, nd) = data['loss_c_fused'], data['loss_d_fused']
            has_gt = float(data.get('spv_num_gt', 1) > 0)
            loss_c, loss_d = self.c_pos_w * has_gt * sc / nc, self.c_pos_w * has_gt * sd / nd
        else:
            w = self.compute_c_weight(data)
            loss_c = self.compute_coarse_loss(data['conf_matrix'], data['conf_matrix_gt'], weight=w)
            loss_d = self.compute_coarse_loss(data['dect_conf_matrix'], data['conf_matrix_gt'], weight=w)
        loss = (loss_c + loss_d) * self.cfg['coarse_weight']
        loss_f = self.compute_fine_loss(data['fine_matrix'], data['conf_matrix_fine_gt'])
        loss = loss + loss_f * self.cfg['fine_weight']
        data.update(loss=loss, loss_scalars={'loss_c': loss_c.detach().cpu(), 'loss_d': loss_d.detach().cpu(),
                                             'loss_f': loss_f.detach().cpu(), 'loss': loss.detach().cpu()})
        return loss
