"""LocalSoftArgmin: drop-in for disp_predictors/local_soft_argmin.py:5-105."""
import torch.nn as nn

from .... import ops


class LocalSoftArgmin(nn.Module):
    """arg-max over D (bit-exact index path, first maximum wins) + soft-argmin over the 2*radius+1 samples around
    it, in one kernel.  ``forward(..., return_index=True)`` additionally returns the int64 arg-max map."""

    def __init__(self, max_disp, radius, start_disp=0, dilation=1, radius_dilation=1, alpha=1.0, normalize=True):
        super().__init__()
        self.max_disp, self.radius, self.start_disp, self.dilation = max_disp, radius, start_disp, dilation
        self.radius_dilation = radius_dilation
        self.end_disp = start_disp + max_disp - 1
        self.disp_sample_number = (max_disp + dilation - 1) // dilation
        self.alpha, self.normalize = alpha, normalize

    def forward(self, cost_volume, disp_sample=None, return_index=False):
        D = cost_volume.size()[1]
        assert D == self.disp_sample_number, 'Number of disparity sample should be same' \
                                             'with predicted disparity number in cost volume!'
        return ops.local_soft_argmin(cost_volume, self.radius, self.radius_dilation, self.start_disp, self.dilation,
                                     self.alpha, return_index=return_index)

    def __repr__(self):
        s = '{}\n'.format(self.__class__.__name__)
        s += ' ' * 4 + 'Max Disparity: {}\n'.format(self.max_disp)
        s += ' ' * 4 + 'Local disparity sample radius: {}\n'.format(self.radius)
        s += ' ' * 4 + 'Start disparity: {}\n'.format(self.start_disp)
        s += ' ' * 4 + 'Dilation rate: {}\n'.format(self.dilation)
        s += ' ' * 4 + 'Local disparity sample dilation rate: {}\n'.format(self.radius_dilation)
        s += ' ' * 4 + 'Alpha: {}\n'.format(self.alpha)
        s += ' ' * 4 + 'Normalize: {}\n'.format(self.normalize)
        return s

    @property
    def name(self):
        return 'LocalSoftArgmin'
