"""GC-Net feature backbone: drop-in for dmb/modeling/stereo/backbones/GCNet.py:8-52 (same ``backbone.N`` keys): a 5x5
stride-2 conv + BN + ReLU, eight BasicBlocks and a 3x3 convolution, every one a fused HIP launch; both views run as one
batch."""
import torch
import torch.nn as nn

from .... import ops
from ..layers import train_fn
from ..layers.basic_layers_2d import BasicBlock, conv_bn_relu
from .StereoNet import _HipConv2d


class GCNetBackbone(nn.Module):
    def __init__(self, in_planes, batch_norm=True):
        super().__init__()
        self.in_planes = in_planes
        self.backbone = nn.Sequential(
            conv_bn_relu(batch_norm, in_planes, 32, 5, 2, 2),
            *[BasicBlock(batch_norm, 32, 32, 1, None, 1, 1) for _ in range(8)],
            _HipConv2d(32, 32, kernel_size=3, stride=1, padding=1))

    def forward(self, *input):
        if len(input) != 2:
            raise ValueError('expected input length 2 (got {} length input)'.format(len(input)))
        l_img, r_img = input
        if train_fn.wants_grad(self, l_img, r_img):
            # one view after the other, as the reference does (backbones/GCNet.py:47-51): BatchNorm statistics per call
            return self.backbone(l_img), self.backbone(r_img)
        # shared weights, per-image results: two chains on two streams, or one batch of 2B images (ops.two_view_forward)
        return ops.two_view_forward(self.backbone, l_img, r_img, module=self)
