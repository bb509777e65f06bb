"""2-D conv + BatchNorm (+ReLU) units and BasicBlock of the feature backbones on the fused HIP conv2d kernel.

Mirrors dmb/modeling/stereo/layers/basic_layers.py:31-46 (conv_bn), :105-123 (conv_bn_relu), :219-243 (BasicBlock):
same factory names, argument order and ``state_dict`` keys; torch.nn modules are parameter containers only."""
import torch
import torch.nn as nn

from .... import ops
from . import train_fn
from .basic_layers import _versions, epoch_on_mode_switch, fold_batch_norm

__all__ = ["FusedConv2d", "conv_bn", "conv_bn_relu", "BasicBlock"]


class FusedConv2d(nn.Sequential):
    """Sequential(Conv2d, [BatchNorm2d], [ReLU]) as ONE kernel launch: conv (k 1 or 3, stride 1 or 2, dilation 1, 2,
    4 or 8) + folded BN + optional residual + ReLU; may read / write channel windows of wider tensors."""

    def __init__(self, batch_norm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True,
                 relu=False):
        # basic_layers.py:14-28: padding follows the dilation when dilation > 1
        pad = dilation if dilation > 1 else padding
        if kernel_size not in (1, 3, 5) or stride not in (1, 2) or dilation not in (1, 2, 4, 8) \
                or pad != dilation * (kernel_size // 2) or (dilation > 2 and out_planes > 32) \
                or (kernel_size == 5 and (stride != 2 or dilation != 1 or out_planes > 32)):
            raise NotImplementedError("HIP conv2d: kernel 1|3, stride 1|2, dilation 1|2 (4|8 up to 32 output channels), "
                                      "or kernel 5 with stride 2 (up to 32 output channels); 'same' padding")
        layers = [nn.Conv2d(in_planes, out_planes, kernel_size, stride=stride, padding=pad, dilation=dilation, bias=bias)]
        if batch_norm:
            layers.append(nn.BatchNorm2d(out_planes))
        if relu:
            layers.append(nn.ReLU(inplace=True))
        super().__init__(*layers)
        self.in_planes, self.out_planes = in_planes, out_planes
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.has_bn, self.has_relu = bool(batch_norm), bool(relu)
        self._cache_key, self._cache = None, None

    def train(self, mode=True):
        epoch_on_mode_switch(self, mode)
        return super().train(mode)

    def _prepacked(self):
        conv = self[0]
        bn = self[1] if self.has_bn else None
        parts = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked] if bn is not None else [])
        key = _versions(*parts)
        if key != self._cache_key:
            w = conv.weight.detach()
            scale, shift = fold_batch_norm(bn, conv.bias, self.out_planes, w.device)
            self._cache_key, self._cache = key, (ops.pack_conv2d_weights(w), scale, shift)
        return self._cache

    def forward(self, x, residual=None, relu=None, in_window=None, out=None, out_ch_offset=0, res_ch_offset=0):
        if train_fn.wants_grad(self, x, residual):
            # training / differentiable path: plain tensors (no channel windows), separate launches under torch.autograd
            if in_window is not None or out is not None or res_ch_offset:
                raise ValueError("FusedConv2d: channel windows are an inference-path feature")
            return train_fn.conv2d_unit(self, x, residual, self.has_relu if relu is None else relu)
        wp, scale, shift = self._prepacked()
        return ops.conv2d(x, wp, self.out_planes, self.kernel_size, self.stride, self.dilation, scale, shift, residual,
                          self.has_relu if relu is None else relu, in_window, out, out_ch_offset, res_ch_offset)


def conv_bn(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
    """basic_layers.py:31-46."""
    return FusedConv2d(batchNorm, in_planes, out_planes, kernel_size, stride, padding, dilation, bias, relu=False)


def conv_bn_relu(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
    """basic_layers.py:105-123."""
    return FusedConv2d(batchNorm, in_planes, out_planes, kernel_size, stride, padding, dilation, bias, relu=True)


class BasicBlock(nn.Module):
    """basic_layers.py:219-243: out = conv2(conv1(x)) + (downsample(x) | x), no ReLU after the add.  Two launches
    (three with a down-sampling 1x1 conv): the skip add runs in conv2's epilogue."""
    expansion = 1

    def __init__(self, batchNorm, in_planes, out_planes, stride, downsample, padding, dilation):
        super().__init__()
        self.conv1 = conv_bn_relu(batchNorm, in_planes, out_planes, 3, stride, padding, dilation, bias=False)
        self.conv2 = conv_bn(batchNorm, out_planes, out_planes, 3, 1, padding, dilation, bias=False)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, out=None, out_ch_offset=0, in_window=None):
        """``in_window=(offset, channels)`` reads the block input from a channel window of a wider tensor."""
        if self.downsample is not None:
            skip, roff = self.downsample(x, in_window=in_window), 0
        else:
            skip, roff = x, (in_window[0] if in_window is not None else 0)
        return self.conv2(self.conv1(x, in_window=in_window), residual=skip, res_ch_offset=roff, out=out,
                          out_ch_offset=out_ch_offset)
