"""Drop-in for dmb/modeling/stereo/disp_refinement/StereoNet.py:7-62: cascade of edge-aware refinement blocks."""
import torch.nn as nn

from .... import ops
from ..layers import train_fn
from .utils.edge_aware import EdgeAwareRefinement


class StereoNetRefinement(nn.Module):
    def __init__(self, in_planes, batch_norm=True, num=1):
        super().__init__()
        self.in_planes, self.batch_norm, self.num = in_planes, batch_norm, num
        self.refine_blocks = nn.ModuleList([EdgeAwareRefinement(in_planes, batch_norm) for _ in range(num)])

    def forward(self, disps, left, right, leftImage, rightImage):
        init_disp = disps[-1]
        h, w = leftImage.shape[-2:]
        scale = w / init_disp.shape[-1]
        if train_fn.wants_grad(self, init_disp):
            init_disp = train_fn.BilinearScaleFn.apply(init_disp, (h, w), scale)
        else:
            init_disp = ops.bilinear_scale(init_disp, (h, w), scale)
        refine_disps = [init_disp]
        for block in self.refine_blocks:
            refine_disps.append(block(refine_disps[-1], leftImage))
        refine_disps.reverse()    # better maps first (StereoNet.py:58-59)
        return refine_disps
