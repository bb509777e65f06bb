"""PSMNet feature backbone (2-D CNN + spatial pyramid pooling): drop-in for dmb/modeling/stereo/backbones/PSMNet.py.

Same module tree (firstconv, layer1-4, branch1-4, lastconv) and ``state_dict`` keys; every convolution is one fused
HIP launch (conv + folded BN + skip + ReLU) and the 320-channel SPP concatenation is never copied: layer2's and
layer4's last blocks and the four up-sampled branches write straight into their channel windows of one buffer."""
import torch
import torch.nn as nn

from .... import ops
from ..layers import train_fn
from ..layers.basic_layers_2d import BasicBlock, conv_bn, conv_bn_relu


class PSMNetBackbone(nn.Module):
    def __init__(self, in_planes=3, batch_norm=True):
        super().__init__()
        self.in_planes, self.batch_norm = in_planes, batch_norm
        bn = batch_norm
        self.firstconv = nn.Sequential(conv_bn_relu(bn, in_planes, 32, 3, 2, 1, 1, bias=False),
                                       conv_bn_relu(bn, 32, 32, 3, 1, 1, 1, bias=False),
                                       conv_bn_relu(bn, 32, 32, 3, 1, 1, 1, bias=False))
        self.in_planes = 32
        self.layer1 = self._make_layer(bn, 32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(bn, 64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(bn, 128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(bn, 128, 3, 1, 2, 2)
        for i, k in enumerate((64, 32, 16, 8), start=1):   # PSMNet.py:43-58
            setattr(self, "branch%d" % i, nn.Sequential(nn.AvgPool2d((k, k), stride=(k, k)),
                                                        conv_bn_relu(bn, 128, 32, 1, 1, 0, 1, bias=False)))
        # the reference's lastconv[1] is a bare nn.Conv2d (key 'lastconv.1.weight'), not a conv_bn Sequential
        self.lastconv = nn.Sequential(conv_bn_relu(bn, 320, 128, 3, 1, 1, 1, bias=False), _BareConv1x1(128, 32))

    def _make_layer(self, bn, out_planes, blocks, stride, padding, dilation):
        downsample = None
        if stride != 1 or self.in_planes != out_planes:
            downsample = conv_bn(bn, self.in_planes, out_planes, kernel_size=1, stride=stride, padding=0, dilation=1)
        layers = [BasicBlock(bn, self.in_planes, out_planes, stride, downsample, padding, dilation)]
        self.in_planes = out_planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(bn, self.in_planes, out_planes, 1, None, padding, dilation))
        return nn.Sequential(*layers)

    def _forward(self, x):
        x = self.firstconv(x)
        x = self.layer1(x)
        B, _, H2, W2 = x.shape
        H4, W4 = (H2 - 1) // 2 + 1, (W2 - 1) // 2 + 1
        # PSMNet.py:119-121: cat(output_4_0 [64], output_8 [128], branch4, branch3, branch2, branch1 [32 each])
        feat = torch.empty((B, 320, H4, W4), dtype=torch.float32, device=x.device)
        for blk in self.layer2[:-1]:
            x = blk(x)
        self.layer2[-1](x, out=feat, out_ch_offset=0)                       # output_4_0 -> channels 0..63
        x = self.layer3[0](feat, in_window=(0, 64))                         # reads that window in place
        for blk in self.layer3[1:]:
            x = blk(x)
        for blk in self.layer4[:-1]:
            x = blk(x)
        self.layer4[-1](x, out=feat, out_ch_offset=64)                      # output_8 -> channels 64..191
        for i, off in ((4, 192), (3, 224), (2, 256), (1, 288)):
            branch = getattr(self, "branch%d" % i)
            k = branch[0].kernel_size[0]
            pooled = ops.avgpool2d(feat, k, in_window=(64, 128))
            ops.bilinear_ac(branch[1](pooled), (H4, W4), out=feat, out_ch_offset=off)
        return self.lastconv[1](self.lastconv[0](feat))

    def _forward_train(self, x):
        """The same network on plain tensors under autograd (SURVEY 8-f3 widened to the backbone): PSMNet.py:64-125."""
        x = self.firstconv(x)
        x = self.layer1(x)
        out2 = self.layer2(x)
        x = self.layer3(out2)
        out4 = self.layer4(x)
        H4, W4 = out4.shape[2:]
        ups = []
        for i in (4, 3, 2, 1):
            branch = getattr(self, "branch%d" % i)
            pooled = train_fn.AvgPool2dFn.apply(out4, branch[0].kernel_size[0])
            ups.append(train_fn.BilinearAcFn.apply(branch[1](pooled), (H4, W4)))
        feat = torch.cat([out2, out4] + ups, 1)                             # PSMNet.py:119-121
        return self.lastconv[1](self.lastconv[0](feat))

    def forward(self, *input):
        if len(input) != 2:
            raise ValueError('expected input length 2 (got {} length input)'.format(len(input)))
        l_img, r_img = input
        if train_fn.wants_grad(self, l_img, r_img):
            # one view after the other, as the reference does (PSMNet.py:127-131): in training mode the BatchNorm statistics
            # are those of each call, and the running buffers are updated twice
            return self._forward_train(l_img), self._forward_train(r_img)
        # shared weights (PSMNet.py:127-131), per-image results: the two views as two chains on two streams (or, with
        # ops.set_view_streams(False), as one batch of 2B images) -- see ops.two_view_forward
        return ops.two_view_forward(self._forward, l_img, r_img, module=self)


class _BareConv1x1(nn.Conv2d):
    """nn.Conv2d(C, Co, 1, bias=False) (PSMNet.py:61-62) on the HIP conv2d kernel; keys identical to nn.Conv2d."""

    def __init__(self, in_planes, out_planes):
        super().__init__(in_planes, out_planes, kernel_size=1, padding=0, stride=1, dilation=1, bias=False)
        self._key, self._wp = None, None

    def _prepacked(self):
        from ..layers.basic_layers import _versions
        key = _versions(self.weight)
        if key != self._key:
            self._key, self._wp = key, ops.pack_conv2d_weights(self.weight.detach())
        return self._wp

    def forward(self, x):
        if train_fn.wants_grad(self, x):
            return train_fn.BareConv1x1Fn.apply(x, self.weight)
        return ops.conv2d(x, self._prepacked(), self.out_channels, 1, 1, 1, None, None, None, False)
