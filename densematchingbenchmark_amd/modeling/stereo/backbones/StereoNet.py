"""StereoNet feature backbone: drop-in for dmb/modeling/stereo/backbones/StereoNet.py:7-106 (same module tree and
``state_dict`` keys): three 5x5 stride-2 down-sampling convolutions, six BasicBlocks and a 3x3 convolution, each one
fused HIP launch; both views run as one batch."""
import torch
import torch.nn as nn

from .... import ops
from ..layers import train_fn
from ..layers.basic_layers import _versions
from ..layers.basic_layers_2d import BasicBlock


class _HipConv2d(nn.Conv2d):
    """Bare nn.Conv2d (bias, no BN, no activation) on the HIP conv2d kernel; keys identical to nn.Conv2d."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding):
        super().__init__(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=True)
        if padding != kernel_size // 2:
            raise NotImplementedError("HIP conv2d: 'same' padding only")
        self._key, self._cache = None, None

    def _prepacked(self):
        key = _versions(self.weight, self.bias)
        if key != self._key:
            self._key = key
            self._cache = (ops.pack_conv2d_weights(self.weight.detach()), self.bias.detach().float().contiguous())
        return self._cache

    def forward(self, x):
        if train_fn.wants_grad(self, x):
            return train_fn.BareConv2dFn.apply(x, self.weight, self.bias, None, False)
        wp, bias = self._prepacked()
        return ops.conv2d(x, wp, self.out_channels, self.kernel_size[0], self.stride[0], 1, None, bias, None, False)


class DownsampleHead(nn.Module):
    """backbones/StereoNet.py:7-32."""

    def __init__(self, in_planes, out_planes, batch_norm=True):
        super().__init__()
        self.in_planes, self.out_planes, self.batch_norm = in_planes, out_planes, batch_norm
        self.downsample = _HipConv2d(in_planes, out_planes, kernel_size=5, stride=2, padding=2)

    def forward(self, x):
        return self.downsample(x)


class StereoNetBackbone(nn.Module):
    def __init__(self, in_planes=3, batch_norm=True, downsample_num=3, residual_num=6):
        super().__init__()
        self.in_planes, self.batch_norm = in_planes, batch_norm
        self.downsample_num, self.residual_num = downsample_num, residual_num
        self.downsample = nn.ModuleList()
        cin = in_planes
        for _ in range(downsample_num):
            self.downsample.append(DownsampleHead(cin, 32))
            cin = 32
        self.residual_blocks = nn.ModuleList(
            [BasicBlock(batch_norm, 32, 32, stride=1, downsample=None, padding=1, dilation=1) for _ in range(residual_num)])
        self.lastconv = _HipConv2d(32, 32, kernel_size=3, stride=1, padding=1)

    def _forward(self, x):
        for head in self.downsample:
            x = head(x)
        for block in self.residual_blocks:
            x = block(x)
        return self.lastconv(x)

    def forward(self, *input):
        if len(input) != 2:
            raise ValueError('expected input length 2 (got {} length input)'.format(len(input)))
        l_img, r_img = input
        if train_fn.wants_grad(self, l_img, r_img):
            # one view after the other, as the reference does (backbones/StereoNet.py:95-99): BatchNorm statistics per call
            return self._forward(l_img), self._forward(r_img)
        # shared weights: one batch of 2B images (this backbone is a millisecond of small launches: two chains on two streams,
        # ops.two_view_forward, measured 1.05 -> 1.14 ms at 8 pairs of 384x1248 -- they pay for the PSMNet and GC-Net backbones)
        B = l_img.shape[0]
        f = self._forward(torch.cat((l_img, r_img), 0))
        return f[:B], f[B:]
