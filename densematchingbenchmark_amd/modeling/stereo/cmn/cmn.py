"""AcfNet confidence measurement network: drop-in for cmn/cmn.py:10-92 (fused inference kernel; training branch on the 2-D
convolution / BatchNorm / weight-gradient kernels under autograd)."""
import torch.nn as nn

from .... import ops
from ..layers import train_fn
from ..layers.basic_layers import _versions, fold_batch_norm


class ConfHead(nn.Module):
    """Conv2d(in_planes -> in_planes//3, 3x3, no bias) + BN + ReLU + Conv2d(-> 1, 1x1, no bias), cmn.py:21-32.
    ``state_dict`` keys as in the reference: conf_net.0.0.weight, conf_net.0.1.*, conf_net.1.weight.
    ``forward`` returns the confidence MAP (sigmoid applied): the fused kernel never materialises the logit."""

    def __init__(self, in_planes, batch_norm=True):
        super().__init__()
        self.in_planes = in_planes
        self.sec_in_planes = int(in_planes // 3) if int(in_planes // 3) > 0 else 1
        first = [nn.Conv2d(in_planes, self.sec_in_planes, 3, 1, 1, bias=False)]
        if batch_norm:
            first.append(nn.BatchNorm2d(self.sec_in_planes))
        first.append(nn.ReLU(inplace=True))
        self.conf_net = nn.Sequential(nn.Sequential(*first), nn.Conv2d(self.sec_in_planes, 1, 1, 1, 0, bias=False))
        self.batch_norm = batch_norm
        self._key, self._cache = None, None

    def _prepacked(self):
        conv1 = self.conf_net[0][0]
        bn = self.conf_net[0][1] if self.batch_norm else None
        conv2 = self.conf_net[1]
        parts = [conv1.weight, conv2.weight] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked] if bn else [])
        key = _versions(*parts)
        if key != self._key:
            w1 = conv1.weight.detach()
            scale, shift = fold_batch_norm(bn, None, self.sec_in_planes, w1.device)
            if scale is None:
                import torch
                scale = torch.ones(self.sec_in_planes, device=w1.device)
                shift = torch.zeros(self.sec_in_planes, device=w1.device)
            self._key = key
            self._cache = (ops.pack_conf_head_weights(w1), scale, shift, conv2.weight.detach().reshape(-1).contiguous())
        return self._cache

    def forward(self, cost):
        wp, scale, shift, w2 = self._prepacked()
        if ops.conf_head_composite_applicable(cost, self.sec_in_planes):
            # ``cost`` is AcfNet's learned 4x up-sampling of a quarter-resolution volume (the aggregator left a note on the
            # tensor): head o up-sampling = 16 phase-wise 3x3 convolutions of that volume, a quarter of the multiplications
            src = ops.UpsampleSource.lookup(cost)
            key = _versions(self.conf_net[0][0].weight, src.w8) + (self._key,)
            if getattr(self, "_comp_key", None) != key:
                self._comp_key = key
                self._comp = ops.conf_head_k8s4_pack(self.conf_net[0][0].weight, src.w8, scale, shift)
            return ops.conf_head_from_source(cost, self._comp, scale, shift, w2)
        return ops.conf_head(cost, wp, scale, shift, w2)

    def logits(self, cost):
        """The confidence cost BEFORE the sigmoid, differentiable (training side): cmn.py:34-36."""
        conv1 = self.conf_net[0][0]
        bn = self.conf_net[0][1] if self.batch_norm else None
        gamma = bn.weight if bn is not None and bn.affine else None
        beta = bn.bias if bn is not None and bn.affine else None
        return train_fn.ConfHeadFn.apply(cost, conv1.weight, gamma, beta, self.conf_net[1].weight, self)


class Cmn(nn.Module):
    """``forward(costs, target=None)`` -> ``(cost_vars, confs)`` in eval mode, ``(cost_vars, cm_losses)`` in training mode
    (cmn.py:57-84)."""

    def __init__(self, cfg, in_planes, num, alpha, beta):
        super().__init__()
        self.cfg = cfg.copy()
        batch_norm = self.cfg.model.batch_norm
        self.conf_heads = nn.ModuleList([ConfHead(in_planes, batch_norm) for _ in range(num)])
        self.alpha, self.beta = alpha, beta
        self.loss_evaluator = None

    def get_confidence(self, costs):
        assert len(self.conf_heads) == len(costs), "NUM of confidence heads({}) must be equal to NUM" \
                                                   "of cost volumes({})".format(len(self.conf_heads), len(costs))
        confs = [head(cost) for cost, head in zip(costs, self.conf_heads)]
        cost_vars = [self.alpha * (1 - conf) + self.beta for conf in confs]
        return confs, cost_vars

    def forward(self, costs, target=None):
        if self.training:
            # cmn.py:62-84: confidence costs (logits) -> NLL loss; variance = alpha * (1 - sigmoid(logit)) + beta feeds the focal
            # loss, so the focal loss's d/d variance flows back into the heads.  The sigmoid / affine on the [B, 1, H, W] maps
            # is left to torch (plumbing-sized); everything on the [B, D, H, W] volumes is HIP.
            import torch
            assert len(self.conf_heads) == len(costs)
            conf_costs = [head.logits(cost) for cost, head in zip(costs, self.conf_heads)]
            cost_vars = [self.alpha * (1 - torch.sigmoid(c)) + self.beta for c in conf_costs]
            if self.loss_evaluator is None:
                from ..losses import ConfidenceNllLoss
                node = self.cfg.model.cmn.losses.nll_loss
                self.loss_evaluator = (ConfidenceNllLoss(max_disp=node.max_disp, start_disp=node.get("start_disp", 0),
                                                         weights=node.get("weights", None), sparse=self.cfg.data.sparse), node.weight)
            evaluator, weight = self.loss_evaluator
            return cost_vars, {k: v * weight for k, v in evaluator(conf_costs, target).items()}
        confs, cost_vars = self.get_confidence(costs)
        return cost_vars, confs


def build_cmn(cfg):
    c = cfg.model.cmn
    return Cmn(cfg, c.in_planes, c.num, c.alpha, c.beta)
