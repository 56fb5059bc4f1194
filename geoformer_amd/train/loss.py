"""GeoLoss: the training objective of GeoFormer (behaviour of model/loftr_src/losses/loftr_loss.py:195-395).

    loss = coarse_weight * (L_c(conf_matrix) + L_c(dect_conf_matrix)) + fine_weight * L_f(fine_matrix)

L_c is evaluated on the entries of a dual-softmax confidence tensor that the ground truth labels 1 ("positives")
and, for the dense variants only, 0 ("negatives"); L_f is a two-sided cross entropy on the 25x25 fine matrices.
Everything is written per selected entry: a `_Picked` holds the gathered probabilities of one label class together
with their weights and the coefficient of that class, and the variants differ only in the per-entry term applied.

Degenerate batches follow the reference: a label class without any entry is represented by the single entry
[0, 0, 0] with weight 0 and coefficient 0 (the graph stays connected, the value is exactly 0).
Config keys are the lower-cased `LOFTR.LOSS.*` of model/loftr_src/config/default.py:48-63.
"""
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

DEFAULT_LOSS_CFG = {
    'coarse_type': 'focal', 'coarse_weight': 1.0, 'focal_alpha': 0.25, 'focal_gamma': 2.0, 'pos_weight': 1.0,
    'neg_weight': 1.0, 'fine_type': 'l2_with_std', 'fine_weight': 1.0, 'fine_correct_thr': 1.0,
}
P_MIN, P_MAX = 1e-6, 1 - 1e-6          # probabilities are clamped into [P_MIN, P_MAX] before any logarithm


class _Picked(NamedTuple):
    p: torch.Tensor                     # clamped probabilities of the class, in row-major order of their positions
    w: Optional[torch.Tensor]           # per-entry weights (padding masks) or None
    coeff: float                        # pos_weight / neg_weight, 0 when the class was empty

    def mean(self, term):
        v = term(self.p)
        if self.w is not None:
            v = v * self.w
        return self.coeff * v.mean()


def _pick(prob, labels, value, weight, coeff):
    """Entries of `prob` whose label equals `value` (empty class -> the [0,0,0] stand-in, see module docstring)."""
    where = (labels == value).nonzero(as_tuple=True)
    if where[0].numel() == 0:
        where = tuple(torch.zeros(1, dtype=torch.long, device=prob.device) for _ in where)
        return _Picked(prob[where], None if weight is None else torch.zeros(1, device=prob.device, dtype=weight.dtype), 0.0), True
    return _Picked(prob[where], None if weight is None else weight[where], coeff), False


class GeoLoss(nn.Module):
    def __init__(self, loss_cfg=None, match_type='dual_softmax', sparse_spvs=True):
        super().__init__()
        self.cfg = dict(DEFAULT_LOSS_CFG, **(loss_cfg or {}))
        if match_type != 'dual_softmax':
            raise NotImplementedError('GeoFormer trains with dual-softmax matching only')
        if self.cfg['coarse_type'] not in ('focal', 'cross_entropy'):
            raise ValueError(f"Unknown coarse loss: {self.cfg['coarse_type']}")
        self.sparse_spvs = sparse_spvs
        self.c_pos_w, self.c_neg_w = self.cfg['pos_weight'], self.cfg['neg_weight']

    # ---- coarse level --------------------------------------------------------------------------------
    def compute_coarse_loss(self, conf, conf_gt, weight=None):
        """conf, conf_gt [N, L, S]; weight [N, L, S] or None (loftr_loss.py:210-284)."""
        kind = self.cfg['coarse_type']
        if kind == 'cross_entropy' and self.sparse_spvs:
            raise AssertionError('Sparse Supervision for cross-entropy not implemented!')
        prob = conf.clamp(P_MIN, P_MAX)
        pos, no_pos = _pick(prob, conf_gt, 1, weight, self.c_pos_w)
        want_neg = not (kind == 'focal' and self.sparse_spvs)       # dual-softmax has no dustbin: sparse = positives only
        if want_neg:
            if no_pos and weight is not None:       # the stand-in positive also silences entry [0,0,0] as a negative
                weight = weight.clone()
                weight[0, 0, 0] = 0
            neg, _ = _pick(prob, conf_gt, 0, weight, self.c_neg_w)
        if kind == 'cross_entropy':
            return pos.mean(lambda p: -p.log()) + neg.mean(lambda p: -(1 - p).log())
        a, g = self.cfg['focal_alpha'], self.cfg['focal_gamma']
        out = pos.mean(lambda p: -a * torch.pow(1 - p, g) * p.log())
        if want_neg:
            out = out + neg.mean(lambda p: -a * torch.pow(p, g) * (1 - p).log())
        return out

    @torch.no_grad()
    def compute_c_weight(self, data):
        """Outer product of the two padding masks, or None without masks (loftr_loss.py:343-351)."""
        if 'mask0' not in data:
            return None
        return (data['mask0'].flatten(-2)[..., None] * data['mask1'].flatten(-2)[:, None]).float()

    def fused_params(self):
        """(alpha, gamma) if this loss is the configuration the fused HIP coarse loss implements, else None."""
        if self.cfg['coarse_type'] == 'focal' and self.sparse_spvs:
            return self.cfg['focal_alpha'], self.cfg['focal_gamma']
        return None

    # ---- fine level ----------------------------------------------------------------------------------
    def compute_fine_loss(self, conf, conf_gt):
        """Two-sided cross entropy on [M, WW, WW]; a side without entries (NaN mean) drops out (loftr_loss.py:286-296)."""
        prob = conf.clamp(P_MIN, P_MAX)
        hit, miss = conf_gt == 1, conf_gt == 0
        sides = [(self.c_pos_w, (-prob[hit].log()).mean()), (self.c_neg_w, (-(1 - prob[miss]).log()).mean())]
        if torch.isnan(sides[1][1]):
            return sides[0][1]
        if torch.isnan(sides[0][1]):
            return sides[1][1]
        return sides[0][0] * sides[0][1] + sides[1][0] * sides[1][1]

    # ---- total ---------------------------------------------------------------------------------------
    def forward(self, data):
        if 'loss_c_fused' in data:       # forward_train(..., fused_coarse_loss=...): (sum over positives, count) from HIP
            live = float(data.get('spv_num_gt', 1) > 0)
            loss_c, loss_d = (self.c_pos_w * live * s / n for s, n in (data['loss_c_fused'], data['loss_d_fused']))
        else:
            w = self.compute_c_weight(data)
            loss_c, loss_d = (self.compute_coarse_loss(data[k], data['conf_matrix_gt'], weight=w)
                              for k in ('conf_matrix', 'dect_conf_matrix'))
        loss_f = self.compute_fine_loss(data['fine_matrix'], data['conf_matrix_fine_gt'])
        loss = (loss_c + loss_d) * self.cfg['coarse_weight'] + loss_f * self.cfg['fine_weight']
        data['loss'] = loss
        data['loss_scalars'] = {k: v.detach().cpu() for k, v in (('loss_c', loss_c), ('loss_d', loss_d), ('loss_f', loss_f), ('loss', loss))}
        return loss
