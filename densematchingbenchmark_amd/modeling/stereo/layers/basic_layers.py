"""3-D conv + BatchNorm (+ReLU) units of the aggregators, backed by the fused HIP kernels.

Mirrors the factories of dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177 -- same names, same argument
order, same ``state_dict`` keys (``<unit>.0.weight``, ``<unit>.1.running_mean`` ...) -- but every unit is a
``FusedConv3d`` whose forward is ONE kernel launch: convolution with BatchNorm folded into a per-channel
scale/shift, optional residual add and ReLU in the epilogue.  torch.nn modules are kept only as parameter
containers so that reference checkpoints load with ``load_state_dict``.  In training mode (or when an input carries a
gradient) the same units run as autograd Functions on the HIP kernels (train_fn.py).  No CPU fallback.
"""
import torch
import torch.nn as nn

from .... import ops
from . import train_fn

__all__ = ["FusedConv3d", "HeadConv3d", "HeadDeconv3d", "conv3d_bn", "conv3d_bn_relu", "deconv3d_bn", "deconv3d_bn_relu",
           "fold_batch_norm"]


def fold_batch_norm(bn, conv_bias, out_planes, device):
    """Eval-mode BN as y = x * scale + shift, folded in FP64 and rounded once:
    scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ conv_bias * scale)."""
    if bn is None:
        if conv_bias is None:
            return None, None
        return torch.ones(out_planes, dtype=torch.float32, device=device), conv_bias.detach().float().contiguous()
    var = bn.running_var.detach().double()
    mean = bn.running_mean.detach().double()
    gamma = bn.weight.detach().double() if bn.affine else torch.ones_like(var)
    beta = bn.bias.detach().double() if bn.affine else torch.zeros_like(var)
    scale = gamma / torch.sqrt(var + bn.eps)
    shift = beta - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


def _versions(*tensors):
    """Cache key of folded / packed parameters.  BatchNorm's ``num_batches_tracked`` is part of every key that covers running
    statistics: the training-mode kernel rewrites ``running_mean`` / ``running_var`` through raw device pointers (their
    ``_version`` does not move), but every such forward bumps ``num_batches_tracked`` in place."""
    return tuple((t.data_ptr(), t._version, t.device) for t in tensors if t is not None) + (ops.param_epoch(),)


def epoch_on_mode_switch(module, mode):
    """Called from train(mode) of the fused modules: a switch between training and evaluation advances ops' parameter epoch, which
    is part of every cache key above -- in-place updates that leave ``_version`` alone (torch's fused optimizers) are then seen by
    the first eval-mode forward after training."""
    if bool(mode) != module.training:
        ops.bump_param_epoch()


class FusedConv3d(nn.Sequential):
    """Sequential(Conv3d | ConvTranspose3d, [BatchNorm3d], [ReLU]) executed as one fused HIP kernel.

    kernel 3 / padding 1 / stride 1|2 convolutions and kernel 3 / stride 2 / padding 1 / output_padding 1
    transposed convolutions (the only forms the aggregators use).  ``forward(x, residual=None)`` computes
    ``act(BN(conv(x)) + residual)`` where ``act`` is ReLU iff the unit has one or ``relu=True`` is passed
    (hourglass.py:67-70,78-81 apply the ReLU after the skip add)."""

    def __init__(self, batch_norm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True,
                 relu=False, transposed=False, output_padding=0):
        layers = []
        if transposed:
            if (kernel_size, stride, padding, output_padding) != (3, 2, 1, 1):
                raise NotImplementedError("HIP transposed conv: only kernel 3, stride 2, padding 1, output_padding 1")
            layers.append(nn.ConvTranspose3d(in_planes, out_planes, kernel_size, stride=stride, padding=padding,
                                             output_padding=output_padding, bias=bias))
        else:
            if kernel_size != 3 or padding != 1 or dilation != 1 or stride not in (1, 2):
                raise NotImplementedError("HIP conv3d: only kernel 3, padding 1, dilation 1, stride 1 or 2")
            layers.append(nn.Conv3d(in_planes, out_planes, kernel_size, stride=stride, padding=padding,
                                    dilation=dilation, bias=bias))
        if batch_norm:
            layers.append(nn.BatchNorm3d(out_planes))
        if relu:
            layers.append(nn.ReLU(inplace=True))
        super().__init__(*layers)
        self.in_planes, self.out_planes, self.stride = in_planes, out_planes, stride
        self.transposed, self.has_bn, self.has_relu = transposed, bool(batch_norm), bool(relu)
        self._cache_key, self._cache = None, None

    def train(self, mode=True):
        epoch_on_mode_switch(self, mode)
        return super().train(mode)

    def _prepacked(self):
        conv = self[0]
        bn = self[1] if self.has_bn else None
        parts = [conv.weight, conv.bias]
        if bn is not None:
            parts += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
        key = _versions(*parts)
        if key != self._cache_key:
            w = conv.weight.detach()
            wp = ops.pack_deconv3d_weights(w) if self.transposed else ops.pack_conv3d_weights(w)
            scale, shift = fold_batch_norm(bn, conv.bias, self.out_planes, w.device)
            self._cache_key, self._cache = key, (wp, scale, shift)
        return self._cache

    def forward(self, x, residual=None, relu=None, skip=None):
        """``residual`` is added BEFORE the activation (hourglass.py:67-81), ``skip`` AFTER it (GC-Net,
        aggregators/GCNet.py:108-116: ``layer34(cost33 + cost29)`` -- the add runs in layer33's epilogue)."""
        act = self.has_relu if relu is None else relu
        if getattr(x, "kind", None) == "gwc_cat":   # LazyGwcCatVolume: correlation channels 3-D, concat channels as 2-D maps
            G = x.num_groups
            if (residual is None and skip is None and not self.transposed and self.stride == 1 and not self.training
                    and x.shape[1] == self.in_planes and G % 2 == 0 and self.out_planes == 32
                    and ops.catconv_applicable(x.reference_fm, x.target_fm, x.disp_idx, self.out_planes)):
                key = _versions(self[0].weight)
                if getattr(self, "_gwc_key", None) != key:
                    w = self[0].weight.detach()
                    self._gwc_key = key
                    self._gwc = (ops.pack_conv3d_weights(w[:, :G].contiguous()), ops.catconv_pack(w[:, G:].contiguous(), "cat"))
                wp_g, packs = self._gwc
                _, scale, shift = self._prepacked()
                # convolution is linear in its input channels: relu(scale * (conv(gwc) + maps(cat)) + shift), the maps' part
                # entering the 3-D kernel's epilogue as its residual operand, already scaled
                part = ops.catconv_first(x.reference_fm, x.target_fm, len(x.disp_idx), packs, scale, None, False)
                return ops.conv3d_k3(x.correlation_part(), wp_g, self.out_planes, scale, shift, part, 1, act)
            x = x.materialize()
        elif hasattr(x, "materialize"):   # LazyCatVolume: the concatenation volume as a description
            if (residual is None and skip is None and not self.transposed and self.stride == 1
                    and x.shape[1] == self.in_planes
                    and ops.catconv_applicable(x.reference_fm, x.target_fm, x.disp_idx, self.out_planes)):
                if getattr(x, "differentiable", False) or self.training:
                    # training path: the 2-D form in the forward pass, the volume only inside the backward pass
                    return train_fn.cat_conv_unit(self, x, act)
                _, scale, shift = self._prepacked()
                return ops.catconv_first(x.reference_fm, x.target_fm, len(x.disp_idx), self._prepacked_cat(x.kind), scale, shift, act)
            x = x.materialize()
        if skip is not None:
            if residual is not None:
                raise ValueError("FusedConv3d: residual and skip are mutually exclusive")
            residual, act = skip, ("pre" if act else False)
        if train_fn.wants_grad(self, x, residual):
            # training / differentiable path: conv, BatchNorm statistics, epilogue and their backward
            # passes as separate HIP launches under torch.autograd
            return train_fn.conv_unit(self, x, residual, act)
        wp, scale, shift = self._prepacked()
        if self.transposed:
            return ops.deconv3d_k3s2(x, wp, self.out_planes, scale, shift, residual, act)
        if ops.conv3d_mode() == "bf16x6" and ops.conv3d_x6_applicable(x, self.out_planes, self.stride):
            return ops.conv3d_k3_x6(x, self._prepacked_x6(), self.out_planes, scale, shift, residual, act)   # opt-in only
        return ops.conv3d_k3(x, wp, self.out_planes, scale, shift, residual, self.stride, act)

    def _prepacked_cat(self, kind):
        key = _versions(self[0].weight) + (kind,)
        if getattr(self, "_cat_key", None) != key:
            self._cat_key, self._cat = key, ops.catconv_pack(self[0].weight.detach(), kind)
        return self._cat

    def _prepacked_x6(self):
        key = _versions(self[0].weight)
        if getattr(self, "_x6_key", None) != key:
            self._x6_key, self._x6 = key, ops.pack_conv3d_x6_weights(self[0].weight.detach())
        return self._x6


class HeadConv3d(nn.Conv3d):
    """nn.Conv3d(C, 1, 3, 1, 1) classifier head (PSMNet.py:46, StereoNet.py:39) on the single-channel HIP kernel;
    ``forward(x, residual=None)`` fuses the cumulative cost add of PSMNet.py:71-72."""

    def __init__(self, in_planes, bias=False):
        super().__init__(in_planes, 1, kernel_size=3, stride=1, padding=1, bias=bias)
        self._bias_key, self._bias_val = None, 0.0

    def train(self, mode=True):
        epoch_on_mode_switch(self, mode)
        return super().train(mode)

    def forward(self, x, residual=None):
        if hasattr(x, "materialize"):   # a lazy volume reaching a head directly (an aggregator without trunk layers)
            x = x.materialize()
        if train_fn.wants_grad(self, x, residual):
            return train_fn.HeadConvFn.apply(x, self.weight, self.bias, residual)
        return ops.conv3d_k3_c1(x, self.weight.detach(), self.bias_value(), residual)

    def bias_value(self):
        if self.bias is None:
            return 0.0
        key = _versions(self.bias)
        if key != self._bias_key:  # one device->host read per weight load, not per call
            self._bias_key, self._bias_val = key, float(self.bias.detach().cpu()[0])
        return self._bias_val


class HeadDeconv3d(nn.ConvTranspose3d):
    """nn.ConvTranspose3d(C, Co<=32, 3, stride 2, padding 1, output_padding 1, bias) without BN / activation (GC-Net's
    1-channel output layer, aggregators/GCNet.py:63-67) on the MFMA transposed kernel with zero-padded weight rows."""

    def __init__(self, in_planes, out_planes):
        super().__init__(in_planes, out_planes, kernel_size=3, stride=2, padding=1, output_padding=1)
        self._key, self._cache = None, None

    def forward(self, x):
        if train_fn.wants_grad(self, x):
            return train_fn.HeadDeconvFn.apply(x, self.weight, self.bias)
        key = _versions(self.weight, self.bias)
        if key != self._key:
            self._key = key
            self._cache = (ops.pack_deconv3d_weights(self.weight.detach()), self.bias.detach().float().contiguous())
        wp, bias = self._cache
        return ops.deconv3d_k3s2(x, wp, self.out_channels, None, bias, None, False)


def conv3d_bn(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
    """basic_layers.py:68-83."""
    return FusedConv3d(batchNorm, in_planes, out_planes, kernel_size, stride, padding, dilation, bias, relu=False)


def conv3d_bn_relu(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
    """basic_layers.py:160-177."""
    return FusedConv3d(batchNorm, in_planes, out_planes, kernel_size, stride, padding, dilation, bias, relu=True)


def deconv3d_bn(batchNorm, in_planes, out_planes, kernel_size=4, stride=2, padding=1, output_padding=0, bias=True):
    """basic_layers.py:86-100."""
    return FusedConv3d(batchNorm, in_planes, out_planes, kernel_size, stride, padding, 1, bias, relu=False,
                       transposed=True, output_padding=output_padding)


def deconv3d_bn_relu(batchNorm, in_planes, out_planes, kernel_size=4, stride=2, padding=1, output_padding=0, bias=True):
    """basic_layers.py:180-197."""
    return FusedConv3d(batchNorm, in_planes, out_planes, kernel_size, stride, padding, 1, bias, relu=True,
                       transposed=True, output_padding=output_padding)
