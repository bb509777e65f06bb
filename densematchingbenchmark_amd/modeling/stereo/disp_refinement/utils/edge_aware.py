"""StereoNet's edge-aware refinement on the HIP conv2d kernel: drop-in for
dmb/modeling/stereo/disp_refinement/utils/edge_aware.py:8-70 (same module tree and ``state_dict`` keys).

14 launches per block: half-pixel bilinear up-sampling * scale, conv_mix (4 -> 32), six dilated BasicBlocks
(dilation 1, 2, 4, 8, 1, 1; two fused launches each, the skip add in the second one's epilogue) and conv_res
(32 -> 1) whose epilogue adds the up-sampled disparity and applies the final ReLU."""
import torch
import torch.nn as nn

from ..... import ops
from ...layers import train_fn
from ...layers.basic_layers import _versions
from ...layers.basic_layers_2d import BasicBlock, conv_bn_relu


class _ResidualHead(nn.Conv2d):
    """nn.Conv2d(32, 1, 3, padding=1, bias=True) (edge_aware.py:42) + skip + ReLU (edge_aware.py:62-66) in one launch."""

    def __init__(self, in_planes):
        super().__init__(in_planes, 1, kernel_size=3, stride=1, padding=1, bias=True)
        self._key, self._cache = None, None

    def forward(self, x, skip, res_ch_offset=0):
        if train_fn.wants_grad(self, x, skip):
            if res_ch_offset or skip.shape[1] != 1:
                raise ValueError("_ResidualHead training path: the skip is the 1-channel up-sampled disparity")
            return train_fn.BareConv2dFn.apply(x, self.weight, self.bias, skip, True)
        key = _versions(self.weight, self.bias)
        if key != self._key:
            self._key = key
            self._cache = (ops.pack_conv2d_weights(self.weight.detach()), self.bias.detach().float().contiguous())
        wp, bias = self._cache
        return ops.conv2d(x, wp, 1, 3, 1, 1, None, bias, skip, True, res_ch_offset=res_ch_offset)


class EdgeAwareRefinement(nn.Module):
    def __init__(self, in_planes, batch_norm=True):
        super().__init__()
        self.in_planes, self.batch_norm = in_planes, batch_norm
        self.conv_mix = conv_bn_relu(batch_norm, in_planes, 32, kernel_size=3, stride=1, padding=1, dilation=1, bias=True)
        self.dilation_list = [1, 2, 4, 8, 1, 1]
        self.residual_dilation_blocks = nn.ModuleList(
            [BasicBlock(batch_norm, 32, 32, stride=1, downsample=None, padding=1, dilation=d) for d in self.dilation_list])
        self.conv_res = _ResidualHead(32)

    def forward(self, disp, leftImage):
        h, w = leftImage.shape[-2:]
        scale = w / disp.shape[-1]
        B = disp.shape[0]
        if train_fn.wants_grad(self, disp):   # training: plain tensors under autograd (edge_aware.py:45-66)
            up = train_fn.BilinearScaleFn.apply(disp, (h, w), scale)
            feat = self.conv_mix(torch.cat((up, leftImage), 1))
            for block in self.residual_dilation_blocks:
                feat = block(feat)
            return self.conv_res(feat, up)
        # cat(up_disp, leftImage) (edge_aware.py:53): the up-sampling kernel writes channel 0 of the mixed input in place
        mixed = torch.empty((B, 1 + leftImage.shape[1], h, w), dtype=torch.float32, device=disp.device)
        ops.bilinear_scale(disp, (h, w), scale, out=mixed, out_ch_offset=0)
        mixed[:, 1:].copy_(leftImage)
        feat = self.conv_mix(mixed)
        for block in self.residual_dilation_blocks:
            feat = block(feat)
        return self.conv_res(feat, mixed, res_ch_offset=0)   # relu(conv_res(feat) + up_disp)
