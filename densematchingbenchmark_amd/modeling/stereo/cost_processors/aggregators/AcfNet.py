"""AcfNet cost aggregation: drop-in for cost_processors/aggregators/AcfNet.py:8-90."""
import torch.nn as nn

from ..... import ops
from ...layers import train_fn
from ...layers.basic_layers import HeadConv3d, conv3d_bn, conv3d_bn_relu
from ..utils.hourglass import Hourglass
from .PSMNet import PSMAggregator


class AcfAggregator(PSMAggregator):
    """Same trunk wiring as PSMAggregator but dres0/dres1/classif*.0 convs carry a bias (AcfNet.py:31-52 omit
    ``bias=False``) and the up-sampling is three learned ConvTranspose3d(1, 1, 8, 4, 2) (AcfNet.py:55-57,81-83)."""

    def __init__(self, max_disp, in_planes=64, batch_norm=True):
        nn.Module.__init__(self)
        self.max_disp, self.in_planes, self.batch_norm = max_disp, in_planes, batch_norm
        bn = batch_norm
        self.dres0 = nn.Sequential(conv3d_bn_relu(bn, in_planes, 32, 3, 1, 1), conv3d_bn_relu(bn, 32, 32, 3, 1, 1))
        self.dres1 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1), conv3d_bn(bn, 32, 32, 3, 1, 1))
        self.dres2 = Hourglass(in_planes=32, batch_norm=bn)
        self.dres3 = Hourglass(in_planes=32, batch_norm=bn)
        self.dres4 = Hourglass(in_planes=32, batch_norm=bn)
        self.classif1 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1), HeadConv3d(32, bias=False))
        self.classif2 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1), HeadConv3d(32, bias=False))
        self.classif3 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1), HeadConv3d(32, bias=False))
        self.deconv1 = nn.ConvTranspose3d(1, 1, 8, 4, 2, bias=False)
        self.deconv2 = nn.ConvTranspose3d(1, 1, 8, 4, 2, bias=False)
        self.deconv3 = nn.ConvTranspose3d(1, 1, 8, 4, 2, bias=False)

    def forward(self, raw_cost):
        B, C, D, H, W = raw_cost.shape
        if D * 4 != self.max_disp:
            raise ValueError("AcfAggregator up-samples exactly 4x: raw volume has %d planes, max_disp=%d" % (D, self.max_disp))
        with train_fn.carry_scope():
            cost1, cost2, cost3 = self.trunk(raw_cost)
        pairs = ((cost3, self.deconv3), (cost2, self.deconv2), (cost1, self.deconv1))
        if train_fn.wants_grad(self, cost1):   # differentiable up-sampling
            return [train_fn.DeconvK8S4Fn.apply(c.squeeze(1), m.weight) for c, m in pairs]
        # as in PSMAggregator: the up-sampling kernel also regresses the standard soft-argmin of the volume it writes and
        # leaves it as a hint on the tensor (bit-identical to the predictor's own pass, which then is not needed)
        vals = ops.disp_sample_values(self.max_disp, 0, 1)
        out = []
        for c, m in pairs:
            cq = c.squeeze(1)
            # (any W: the kernel's 16-byte stores go to column 4 q of rows of 4 W floats -- always aligned)
            cost, disp = ops.deconv3d_k8s4_c1_soft_argmin(cq, m.weight.detach().view(8, 8, 8), vals, 1.0)
            ops.RegressionHint.attach(cost, vals, 1.0, disp)
            out.append(ops.UpsampleSource.attach(cost, cq, m.weight))   # lets the confidence head work at quarter resolution
        return out
