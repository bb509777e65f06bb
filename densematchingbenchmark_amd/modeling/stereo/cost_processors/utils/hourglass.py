"""Stacked-hourglass unit of PSMNet/AcfNet: drop-in for cost_processors/utils/hourglass.py:8-86."""
import torch.nn as nn

from ..... import ops
from ...layers import train_fn
from ...layers.basic_layers import conv3d_bn, conv3d_bn_relu, deconv3d_bn


class Hourglass(nn.Module):
    """Six fused conv units; every skip add and ReLU of hourglass.py:62-86 runs in the producing kernel's
    epilogue.  ``forward(x, presqu, postsqu)`` returns ``(out, pre, post)`` like the reference; the extra
    keyword ``skip`` fuses the caller's ``out + cost0`` (PSMNet.py:62,65,68) into conv6."""

    def __init__(self, in_planes, batch_norm=True):
        super().__init__()
        self.batch_norm = batch_norm
        c = in_planes
        self.conv1 = conv3d_bn_relu(batch_norm, c, c * 2, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv2 = conv3d_bn(batch_norm, c * 2, c * 2, kernel_size=3, stride=1, padding=1, bias=False)
        self.conv3 = conv3d_bn_relu(batch_norm, c * 2, c * 2, kernel_size=3, stride=2, padding=1, bias=False)
        self.conv4 = conv3d_bn_relu(batch_norm, c * 2, c * 2, kernel_size=3, stride=1, padding=1, bias=False)
        self.conv5 = deconv3d_bn(batch_norm, c * 2, c * 2, kernel_size=3, padding=1, output_padding=1, stride=2,
                                 bias=False)
        self.conv6 = deconv3d_bn(batch_norm, c * 2, c, kernel_size=3, padding=1, output_padding=1, stride=2,
                                 bias=False)

    def forward(self, x, presqu=None, postsqu=None, skip=None):
        out = self.conv1(x)                                            # hourglass.py:64
        pre = self.conv2(out, residual=postsqu, relu=True)             # :66-70  relu(conv2(out) [+ postsqu])
        out = self.conv3(pre)                                          # :73
        up_res = presqu if presqu is not None else pre
        if (out.shape[-1] % 4 and ops.padded_rows_applicable(out, self.conv5.out_planes)
                and not train_fn.wants_grad(self.conv4, out) and not train_fn.wants_grad(self.conv5, out, up_res)):
            # The deepest level is internal to this module, so its row length is ours to choose: when the rows are not a
            # 16-byte multiple (KITTI: 1248 / 16 = 78 columns) they are padded with zero columns to the next multiple of 4.  A
            # zero column IS the convolution's padding, so conv4 over the padded tensor computes the same outputs (what it
            # writes into the padding columns is cleared again), and the transposed conv5 only writes the real 2 x 78 columns:
            # both layers stay on their 16-byte kernels instead of the dword ones (0.36 -> 0.19 ms and 0.34 -> 0.26 ms there).
            w = out.shape[-1]
            out = self.conv4(ops.copy_window(out, (w + 3) // 4 * 4, 0))
            ops.zero_columns_(out, w)
            wp, scale, shift = self.conv5._prepacked()
            post = ops.deconv3d_k3s2(out, wp, self.conv5.out_planes, scale, shift, up_res, True, out_width=2 * w)
        else:
            out = self.conv4(out)                                      # :75
            post = self.conv5(out, residual=up_res, relu=True)         # :78-81
        out = self.conv6(post, residual=skip)                          # :84 (+ caller's skip)
        return out, pre, post
