"""Difference cost volume: drop-in for dmb/modeling/stereo/cost_processors/utils/dif_fms.py (``DIF_FUNCS``)."""
import torch

from ..... import ops
from ...layers import train_fn


def dif_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None,
            normalize=False, p=1.0):
    """[B, C, H, W] x 2 -> [B, C, D, H, W] (dif_fms.py:7-46; ``normalize``/``p`` are unused there as well)."""
    idx = ops.disp_index_list(max_disp, start_disp, dilation)
    if torch.is_grad_enabled() and (reference_fm.requires_grad or target_fm.requires_grad):
        return train_fn.DifFmsFn.apply(reference_fm.float().contiguous(), target_fm.float().contiguous(), tuple(idx))
    return ops.dif_fms(reference_fm.float(), target_fm.float(), idx)


def fast_dif_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None,
                 normalize=False, p=1.0):
    """dif_fms.py:49-86: reference features (masked where the warped target is not positive) minus the target features
    warped by ``disp_sample`` ([B, D, H, W]) or by the builder's own linspace samples; ``normalize`` reduces the channels
    with a p-norm ([B, D, H, W]).  Same sampling convention as fast_cat_fms (csrc/warp_volume.hip).  Under autograd the two
    feature maps get their gradients from the sampler's adjoint and per-pixel samples theirs from its column derivative (what
    the reference gets from F.grid_sample's backward), with or without ``normalize``."""
    wrt_samples = disp_sample is not None and disp_sample.requires_grad
    needs_grad = torch.is_grad_enabled() and (reference_fm.requires_grad or target_fm.requires_grad or wrt_samples)
    if disp_sample is None:
        disp_sample = ops.fast_disp_samples(max_disp, start_disp, dilation)
    if needs_grad:
        return train_fn.FastFmsFn.apply(reference_fm.float().contiguous(), target_fm.float().contiguous(),
                                        train_fn.fast_samples_for_autograd(disp_sample, reference_fm), True, bool(normalize), p)
    return ops.fast_dif_fms(reference_fm.float(), target_fm.float(), disp_sample, normalize, p)


DIF_FUNCS = dict(
    default=dif_fms,
    fast_mode=fast_dif_fms,
)
