"""StereoNet cost aggregation: drop-in for cost_processors/aggregators/StereoNet.py:9-55."""
import torch.nn as nn

from ...layers.basic_layers import HeadConv3d, conv3d_bn_relu


class StereoNetAggregator(nn.Module):
    """``num`` x (Conv3d 32->32 + BN + ReLU, bias=True) + Conv3d 32->1 (bias=True); returns ``[cost]`` at the
    volume's own resolution, [B, D, H, W] (no up-sampling, StereoNet.py:42-55)."""

    accepts_lazy_cat = True   # classify[0] is a FusedConv3d: it takes a LazyCatVolume (here: of the difference volume)

    def __init__(self, max_disp, in_planes=32, batch_norm=True, num=4):
        super().__init__()
        self.max_disp, self.in_planes, self.batch_norm, self.num = max_disp, in_planes, batch_norm, num
        self.classify = nn.ModuleList([
            conv3d_bn_relu(batch_norm, in_planes, 32, kernel_size=3, stride=1, padding=1, dilation=1, bias=True)
            for _ in range(num)])
        self.lastconv = HeadConv3d(32, bias=True)

    def forward(self, raw_cost):
        for layer in self.classify:
            raw_cost = layer(raw_cost)
        return [self.lastconv(raw_cost).squeeze(1)]
