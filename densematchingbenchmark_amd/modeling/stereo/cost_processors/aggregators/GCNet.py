"""GC-Net aggregator: drop-in for dmb/modeling/stereo/cost_processors/aggregators/GCNet.py:7-120.

Same module tree (layer19 ... layer37) and ``state_dict`` keys.  19 fused MFMA launches: every conv / transposed conv
carries its BatchNorm and ReLU, and the four skip additions of the decoder (``layer34(cost33 + cost29)`` ...) run in
the PRODUCING layer's epilogue (ReLU first, then the add); the only PyTorch ops are the four channel concatenations
that feed the stride-2 layers (plain copies)."""
import torch
import torch.nn as nn

from ...layers.basic_layers import HeadDeconv3d, conv3d_bn_relu, deconv3d_bn_relu


class GCAggregator(nn.Module):
    """raw_cost [B, in_planes, max_disp/2, H/2, W/2] -> [cost [B, max_disp, H, W]]."""

    def __init__(self, max_disp, in_planes=64, batch_norm=True):
        super().__init__()
        self.max_disp, self.in_planes, self.batch_norm = max_disp, in_planes, batch_norm
        self.F = F = in_planes // 2
        self.layer19 = self._make_layer(in_planes, F)
        self.layer20 = self._make_layer(F, F)
        self.layer21 = self._make_layer(in_planes + F, 2 * F, 2)
        self.layer22 = self._make_layer(2 * F, 2 * F)
        self.layer23 = self._make_layer(2 * F, 2 * F)
        self.layer24 = self._make_layer(4 * F, 2 * F, 2)
        self.layer25 = self._make_layer(2 * F, 2 * F)
        self.layer26 = self._make_layer(2 * F, 2 * F)
        self.layer27 = self._make_layer(4 * F, 2 * F, 2)
        self.layer28 = self._make_layer(2 * F, 2 * F)
        self.layer29 = self._make_layer(2 * F, 2 * F)
        self.layer30 = self._make_layer(4 * F, 4 * F, 2)
        self.layer31 = self._make_layer(4 * F, 4 * F)
        self.layer32 = self._make_layer(4 * F, 4 * F)
        self.layer33 = self._make_tlayer(4 * F, 2 * F)
        self.layer34 = self._make_tlayer(2 * F, 2 * F)
        self.layer35 = self._make_tlayer(2 * F, 2 * F)
        self.layer36 = self._make_tlayer(2 * F, F)
        self.layer37 = self._make_tlayer(F, 1, has_bn_relu=False)

    def _make_layer(self, in_planes, out_planes, stride=1):
        return conv3d_bn_relu(self.batch_norm, in_planes, out_planes, kernel_size=3, stride=stride, padding=1,
                              dilation=1, bias=False)

    def _make_tlayer(self, in_planes, out_planes, stride=2, has_bn_relu=True):
        if has_bn_relu:
            return deconv3d_bn_relu(self.batch_norm, in_planes, out_planes, kernel_size=3, stride=stride, padding=1,
                                    output_padding=1, bias=False)
        return HeadDeconv3d(in_planes, out_planes)

    def forward(self, raw_cost):
        c18 = raw_cost
        c20 = self.layer20(self.layer19(c18))
        c21 = self.layer21(torch.cat([c18, c20], dim=1))
        c23 = self.layer23(self.layer22(c21))
        c24 = self.layer24(torch.cat([c21, c23], dim=1))
        c26 = self.layer26(self.layer25(c24))
        c27 = self.layer27(torch.cat([c24, c26], dim=1))
        c29 = self.layer29(self.layer28(c27))
        c30 = self.layer30(torch.cat([c27, c29], dim=1))
        c32 = self.layer32(self.layer31(c30))
        s33 = self.layer33(c32, skip=c29)          # = cost33 + cost29 (GCNet.py:110)
        s34 = self.layer34(s33, skip=c26)          # = cost34 + cost26 (:112)
        s35 = self.layer35(s34, skip=c23)          # = cost35 + cost23 (:114)
        s36 = self.layer36(s35, skip=c20)          # = cost36 + cost20 (:116)
        return [self.layer37(s36).squeeze(dim=1)]
