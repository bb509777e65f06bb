"""SoftArgmin: drop-in for disp_predictors/soft_argmin.py:5-75 on the fused HIP soft-argmin kernel."""
import torch
import torch.nn as nn

from .... import ops
from ..layers import train_fn


class _SoftArgminBase(nn.Module):
    def __init__(self, max_disp=192, start_disp=0, dilation=1, alpha=1.0, normalize=True):
        super().__init__()
        self.max_disp, self.start_disp, self.dilation = max_disp, start_disp, dilation
        self.end_disp = start_disp + max_disp - 1
        self.disp_sample_number = (max_disp + dilation - 1) // dilation
        self.alpha, self.normalize = alpha, normalize

    def _check(self, cost_volume):
        if cost_volume.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(cost_volume.dim()))

    def __repr__(self):
        s = '{}\n'.format(self.__class__.__name__)
        s += ' ' * 4 + 'Max Disparity: {}\n'.format(self.max_disp)
        s += ' ' * 4 + 'Start disparity: {}\n'.format(self.start_disp)
        s += ' ' * 4 + 'Dilation rate: {}\n'.format(self.dilation)
        s += ' ' * 4 + 'Alpha: {}\n'.format(self.alpha)
        s += ' ' * 4 + 'Normalize: {}\n'.format(self.normalize)
        return s


class SoftArgmin(_SoftArgminBase):
    """cost [B, D, H, W] (+ optional per-pixel ``disp_sample`` [B, D, H, W]) -> disparity [B, 1, H, W].
    One pass over the cost volume: alpha scale, soft-max over D and the expectation are fused."""

    def __init__(self, max_disp=192, start_disp=0, dilation=1, alpha=1.0, normalize=True):
        super().__init__(max_disp, start_disp, dilation, alpha, normalize)
        self.disp_sample = torch.linspace(self.start_disp, self.end_disp, self.disp_sample_number)

    def forward(self, cost_volume, disp_sample=None):
        self._check(cost_volume)
        D = cost_volume.shape[1]
        if disp_sample is None:
            assert D == self.disp_sample_number, 'The number of disparity samples should be consistent!'
            vals = self.disp_sample.tolist()
            hint = ops.RegressionHint.lookup(cost_volume, vals, self.alpha, self.normalize)
            if hint is not None:     # the producing kernel already regressed this very tensor with these parameters
                return hint
            if torch.is_grad_enabled() and cost_volume.requires_grad:
                if not self.normalize:   # never return a silently detached result to a caller that asked for gradients
                    raise NotImplementedError("SoftArgmin(normalize=False) has no backward on the HIP path")
                return train_fn.SoftArgminFn.apply(cost_volume, tuple(vals), self.alpha)
            return ops.soft_argmin(cost_volume, vals, self.alpha, self.normalize)
        assert D == disp_sample.shape[1], 'The number of disparity samples should be consistent!'
        if torch.is_grad_enabled() and (cost_volume.requires_grad or disp_sample.requires_grad):
            raise NotImplementedError("SoftArgmin with a per-pixel disp_sample has no backward on the HIP path")
        return ops.soft_argmin_sampled(cost_volume, disp_sample.float().expand_as(cost_volume).contiguous(),
                                       self.alpha, self.normalize)

    @property
    def name(self):
        return 'SoftArgmin'
