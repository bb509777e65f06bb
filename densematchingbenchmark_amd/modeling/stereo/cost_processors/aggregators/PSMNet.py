"""PSMNet stacked-hourglass cost aggregation: drop-in for cost_processors/aggregators/PSMNet.py:9-95."""
import torch
import torch.nn as nn

from ..... import ops
from ...layers import train_fn
from ...layers.basic_layers import HeadConv3d, conv3d_bn, conv3d_bn_relu
from ..utils.hourglass import Hourglass


class PSMAggregator(nn.Module):
    """raw_cost [B, in_planes, D/4, H/4, W/4] -> [cost3, cost2, cost1], each [B, max_disp, H, W] (best first).
    25 fused MFMA conv launches + 3 head convs + 3 trilinear up-samplings; all biases absent (PSMNet.py:31-54)."""

    accepts_lazy_cat = True   # dres0[0] is a FusedConv3d: it takes a LazyCatVolume (cost_processors/utils/cat_fms.py)

    def __init__(self, max_disp, in_planes=64, batch_norm=True):
        super().__init__()
        self.max_disp, self.in_planes, self.batch_norm = max_disp, in_planes, batch_norm
        bn = batch_norm
        self.dres0 = nn.Sequential(conv3d_bn_relu(bn, in_planes, 32, 3, 1, 1, bias=False),
                                   conv3d_bn_relu(bn, 32, 32, 3, 1, 1, bias=False))
        self.dres1 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1, bias=False),
                                   conv3d_bn(bn, 32, 32, 3, 1, 1, bias=False))
        self.dres2 = Hourglass(in_planes=32, batch_norm=bn)
        self.dres3 = Hourglass(in_planes=32, batch_norm=bn)
        self.dres4 = Hourglass(in_planes=32, batch_norm=bn)
        self.classif1 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1, bias=False), HeadConv3d(32, bias=False))
        self.classif2 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1, bias=False), HeadConv3d(32, bias=False))
        self.classif3 = nn.Sequential(conv3d_bn_relu(bn, 32, 32, 3, 1, 1, bias=False), HeadConv3d(32, bias=False))

    def trunk(self, raw_cost):
        """PSMNet.py:58-72 at 1/4 resolution: returns (cost1, cost2, cost3), each [B, 1, D/4, H/4, W/4]."""
        cost0 = self.dres0(raw_cost)
        cost0 = self.dres1[1](self.dres1[0](cost0), residual=cost0)      # dres1(cost0) + cost0
        out1, pre1, post1 = self.dres2(cost0, None, None, skip=cost0)    # out1 + cost0 fused into conv6
        out2, pre2, post2 = self.dres3(out1, pre1, post1, skip=cost0)
        out3, pre3, post3 = self.dres4(out2, pre2, post2, skip=cost0)
        cost1 = self.classif1[1](self.classif1[0](out1))
        cost2 = self.classif2[1](self.classif2[0](out2), residual=cost1)  # classif2(out2) + cost1
        cost3 = self.classif3[1](self.classif3[0](out3), residual=cost2)
        return cost1, cost2, cost3

    def _forward_overlapped(self, raw_cost):
        """Eval only: the same launches as trunk() + the up-sampling loop, with classifier branch k (and its up-sampling /
        regression) on a second stream next to hourglass k + 1.  Same kernels on the same operands: identical results."""
        B, C, D, H, W = raw_cost.shape
        size = (self.max_disp, H * 4, W * 4)
        vals = ops.disp_sample_values(self.max_disp, 0, 1)
        main = torch.cuda.current_stream(raw_cost.device)
        side = ops.side_stream(raw_cost.device)

        def upsample(c):
            cost, disp = ops.trilinear_ac_soft_argmin(c.squeeze(1), size, vals, 1.0)
            cost.record_stream(main)            # returned to the caller, who works on its own stream
            disp.record_stream(main)
            return ops.RegressionHint.attach(cost, vals, 1.0, disp)

        cost0 = self.dres0(raw_cost)
        cost0 = self.dres1[1](self.dres1[0](cost0), residual=cost0)
        out1, pre1, post1 = self.dres2(cost0, None, None, skip=cost0)
        fork1 = main.record_event()
        with torch.cuda.stream(side):
            side.wait_event(fork1)
            cost1 = self.classif1[1](self.classif1[0](out1))
            up1 = upsample(cost1)
        out2, pre2, post2 = self.dres3(out1, pre1, post1, skip=cost0)
        fork2 = main.record_event()
        with torch.cuda.stream(side):
            side.wait_event(fork2)
            cost2 = self.classif2[1](self.classif2[0](out2), residual=cost1)
            cost2.record_stream(main)           # allocated on the side stream, read by classif3 on the caller's
            join2 = side.record_event()
            up2 = upsample(cost2)
            done = side.record_event()
        out3, pre3, post3 = self.dres4(out2, pre2, post2, skip=cost0)
        main.wait_event(join2)
        cost3 = self.classif3[1](self.classif3[0](out3), residual=cost2)
        up3 = upsample(cost3)
        main.wait_event(done)
        return [up3, up2, up1]

    def forward(self, raw_cost):
        B, C, D, H, W = raw_cost.shape
        if ops.branch_overlap(raw_cost) and raw_cost.device.type == "cuda" and not train_fn.wants_grad(self, raw_cost):
            return self._forward_overlapped(raw_cost)
        with train_fn.carry_scope():                                     # (shares the model's scope when called from one)
            cost1, cost2, cost3 = self.trunk(raw_cost)
        size = (self.max_disp, H * 4, W * 4)                             # PSMNet.py:75-88, align_corners=True
        # The up-sampling kernel also regresses the standard soft-argmin (alpha 1, samples 0..max_disp-1) of the volume
        # it writes and leaves it as a hint on the tensor: FasterSoftArgmin / SoftArgmin with exactly those parameters
        # return it instead of reading the [B, max_disp, H, W] volume again (bit-identical either way).
        vals = ops.disp_sample_values(self.max_disp, 0, 1)
        up = []
        for c in (cost3, cost2, cost1):
            if train_fn.wants_grad(self, c):   # differentiable through the disparity
                cost, disp = train_fn.UpsampleRegressFn.apply(c.squeeze(1), size, tuple(vals), 1.0)
            else:
                cost, disp = ops.trilinear_ac_soft_argmin(c.squeeze(1), size, vals, 1.0)
            up.append(ops.RegressionHint.attach(cost, vals, 1.0, disp))
        return up
